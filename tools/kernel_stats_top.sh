#!/bin/bash
# per-kernel totals of one bench run under rocprofv3 --kernel-trace --stats: top N kernels by total time
# usage (GPU box): bash tools/kernel_stats_top.sh <tag> <N> [bench.py args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; N=$2; shift; shift
O=$R/gpurun_out/ks_$TAG; rm -rf $O; mkdir -p $O
(cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python3 bench.py --no-cpu-baseline --no-sync-form "$@" > $O/run.log 2>&1)
grep '^{"metric' $O/run.log | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_per_step", d["ms_per_step"], "steps", d["steps"], "warmup", d["warmup"])'
python3 - "$O" "$N" <<'PY' | tee $R/gpurun_out/ks_$TAG.txt
import csv,sys,glob
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:-float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms over the whole run" % (tot/1e6))
for r in rows[:int(sys.argv[2])]:
    print("%-110s calls %6s total %9.2f ms avg %9.1f us %5.1f%%" % (r["Name"].replace("(anonymous namespace)::","").replace("void ","")[:110], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
