#!/bin/bash
# average duration of the kernels whose names match a pattern, in the default bench step.   usage: tools/micro/kernel_avg.sh "<grep -E pattern>" [bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=$1; shift
O=gpurun_out/prof_kavg; rm -rf $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-two-call --no-fp32-form --no-graph-form --no-sync-form --no-trajectory "$@" > $O.log 2>&1
python3 - $O "$P" <<'PY'
import sys, csv, glob, re
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for row in csv.DictReader(open(f)):
    if re.search(sys.argv[2], row["Name"]):
        print("%8.1f us avg  calls %4d  %s" % (float(row["AverageNs"]) / 1e3, int(row["Calls"]), row["Name"][:100]))
PY
