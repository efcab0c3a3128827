#!/bin/bash
# bf16x3 mode: per-kernel totals of one step and the distribution of split3_kernel launches by duration.   usage: tools/micro/x3_split_sizes.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/kernel_stats_top.sh x3 34 --precision bf16x3 --steps 3 --warmup 1 --no-two-call --no-fp32-form --no-graph-form --no-trajectory
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/ks_x3/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
for kern in ("split3_kernel", "split3_transpose_kernel", "softmax_fwd_reg_kernel", "softmax_bwd_kernel"):
    d = sorted(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if kern + "(" in r["Kernel_Name"]), reverse=True)
    if not d: continue
    print("%s: %d launches, total %.1f ms; top: %s; >=500us: %d (%.1f ms), 50-500us: %d (%.1f ms), <50us: %d (%.1f ms)" % (
        kern, len(d), sum(d) / 1e3, " ".join("%.0f" % x for x in d[:8]), sum(x >= 500 for x in d), sum(x for x in d if x >= 500) / 1e3,
        sum(50 <= x < 500 for x in d), sum(x for x in d if 50 <= x < 500) / 1e3, sum(x < 50 for x in d), sum(x for x in d if x < 50) / 1e3))
PY
