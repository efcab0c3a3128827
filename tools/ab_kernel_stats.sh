#!/bin/bash
# same-box A/B of per-kernel durations between the in-tree libttmi.so ("new") and another build ("base", TTMI_LIB): alternating rocprofv3
# kernel-trace runs of the default bench, average duration of every kernel whose name matches one of the patterns, and the step time.
# usage (GPU box): bash tools/ab_kernel_stats.sh ab/libttmi_base.so "joint_tanh|joint_sum" [rounds] [extra bench.py args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OTHER=$(readlink -f $R/$1); PAT=$2; ROUNDS=${3:-2}; shift; shift; shift
for i in $(seq 1 $ROUNDS); do for v in base new; do
  O=$R/gpurun_out/abk_$v; rm -rf $O; mkdir -p $O
  if [ $v = base ]; then export TTMI_LIB=$OTHER; else export TTMI_LIB=; fi
  (cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-two-call --no-fp32-form --no-graph-form --no-sync-form "$@" > $O/run.log 2>&1)
  echo "== $v step $(grep '^{"metric' $O/run.log | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
  python3 - "$O" "$PAT" <<'PY'
import csv,sys,glob,re
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if re.search(sys.argv[2], n):
        print("   %-80s calls %4s avg %9.1f us" % (n.replace("(anonymous namespace)::","").replace("void ","")[:80], r["Calls"], float(r["AverageNs"])/1e3))
PY
done; done
