ulimit -c 0; export HSA_ENABLE_COREDUMP=0
F="--workload c5 --steps 6 --warmup 2 --no-cpu-baseline --no-fp32-form --no-two-call --no-graph-form"
for c in 4 8 4 8; do
  timeout -k 10 300 python bench.py $F --loss-chunk $c 2>&1 | tail -1 > /tmp/l.json || break
  python3 -c "import sys,json; d=json.loads(open('/tmp/l.json').read()); print('chunk $c', d['ms_per_step'], d['value'], d['final_loss'])" || { tail -c 400 /tmp/l.json; break; }
done
