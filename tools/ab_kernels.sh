# same-box A/B of per-kernel durations: the round-2 tree kept in ab/r02 (built by hand from commit 899c9a3) against the current tree
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in old new old new; do
  if [ $v = old ]; then D=$R/ab/r02; else D=$R; fi
  O=$R/gpurun_out/ab_$v; rm -rf $O; mkdir -p $O
  (cd $D && rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-two-call $( [ $v = new ] && echo "--no-fp32-form --no-graph-form" ) > $O/run.log 2>&1)
  echo "== $v $(tail -1 $O/run.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
  python3 - "$O/t_kernel_stats.csv" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if any(k in n for k in ("v8_kernel<unsigned short, 3>","v8_kernel<unsigned short, 4>","tn_bf16_v8_kernel<2>","joint_tanh","joint_sum")):
        print("   %-70s avg %8.1f us" % (n.replace("(anonymous namespace)::","")[:70], float(r["AverageNs"])/1e3))
PY
done
