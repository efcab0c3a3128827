#!/bin/bash
# which kernels pay for a CU reservation: per-kernel totals of two traces of the same bench command.   usage: tools/micro/reserve_kernel_diff.sh <r>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_reserve; rm -rf $O; mkdir -p $O
for r in 0 $1; do
  export TTMI_BENCH_RESERVE_CUS=$r
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/r$r -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-two-call --no-fp32-form --no-graph-form --no-sync-form --no-trajectory > $O/r$r.log 2>&1
done
python3 - $O/r0 $O/r$1 <<'PY'
import sys, csv, glob, collections
def load(d):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    out = {}
    for row in csv.DictReader(open(f)):
        out[row["Name"]] = (int(row["Calls"]), float(row["TotalDurationNs"]) / 1e6)
    return out
a, b = load(sys.argv[1]), load(sys.argv[2])
rows = sorted(((b.get(k, (0, 0))[1] - a.get(k, (0, 0))[1], k) for k in set(a) | set(b)), reverse=True)
print("kernel totals over the run (8 steps): reserved minus unreserved, ms")
for dms, k in rows[:14] + rows[-4:]:
    print("%+8.3f  %7.3f -> %7.3f  calls %5d  %s" % (dms, a.get(k, (0, 0))[1], b.get(k, (0, 0))[1], b.get(k, a.get(k))[0], k[:110]))
PY
