# same box, alternating: option 13 (second bf16 term of the four encoder forward weights as a second K range, one launch each) off / on
ulimit -c 0; export HSA_ENABLE_COREDUMP=0
F="--steps 20 --warmup 5 --no-cpu-baseline --no-fp32-form --no-two-call --no-graph-form"
for o in 0 1 0 1; do
  TTMI_OPTIONS=13=$o python bench.py $F 2>&1 | tail -1 > /tmp/l.json
  python3 -c "import json; d=json.loads(open('/tmp/l.json').read()); print('option13 $o', d['ms_per_step'], d['value'])"
done
