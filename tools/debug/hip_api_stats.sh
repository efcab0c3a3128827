#!/bin/bash
# HIP API calls of the default bench (which ones block the host: *Synchronize, hipMemcpy*, hipMalloc / hipFree), GPU box
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/hipapi; rm -rf $O; mkdir -p $O
(cd $R && rocprofv3 --hip-trace --stats --output-format csv -d $O -o t -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-two-call --no-fp32-form --no-graph-form --no-sync-form > $O/run.log 2>&1)
python3 - "$O" <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*hip_api_stats.csv",recursive=True)
if not f: f=glob.glob(sys.argv[1]+"/**/*domain_stats.csv",recursive=True)
print(f)
for r in sorted(csv.DictReader(open(f[0])), key=lambda r:-float(r["TotalDurationNs"]))[:25]:
    print("%-40s calls %7s total %9.2f ms avg %9.1f us" % (r["Name"][:40], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3))
PY
