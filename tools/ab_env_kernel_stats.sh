#!/bin/bash
# same-box A/B of per-kernel durations of ONE build under two environments: alternating rocprofv3 kernel-trace runs of the default bench
# usage (GPU box): bash tools/ab_env_kernel_stats.sh "VAR=a" "VAR=b" "pattern" [rounds] [extra bench.py args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
EA=$1; EB=$2; PAT=$3; ROUNDS=${4:-2}; shift; shift; shift; shift
for i in $(seq 1 $ROUNDS); do for v in A B; do
  O=$R/gpurun_out/abe_$v; rm -rf $O; mkdir -p $O
  if [ $v = A ]; then E=$EA; else E=$EB; fi
  (cd $R && export $E && rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-two-call --no-fp32-form --no-graph-form --no-sync-form "$@" > $O/run.log 2>&1)
  echo "== $v ($E) step $(grep '^{"metric' $O/run.log | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
  python3 - "$O" "$PAT" <<'PY'
import csv,sys,glob,re
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if re.search(sys.argv[2], n):
        print("   %-80s calls %4s avg %9.1f us" % (n.replace("(anonymous namespace)::","").replace("void ","")[:80], r["Calls"], float(r["AverageNs"])/1e3))
PY
done; done
