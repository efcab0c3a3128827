# per-kernel average durations of one C2-shaped encoder layer under the flash kernels' timing switches (TTMI_FLASH_DEBUG bit sets given as
# arguments; results of such runs are wrong by construction - timing experiments only):  tools/exp_attn.sh 0 1 8 16 ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/exp_attn
mkdir -p $O
for dbg in "$@"; do
  rm -rf $O/d$dbg
  TTMI_FLASH_DEBUG=$dbg rocprofv3 --kernel-trace --stats --output-format csv -d $O/d$dbg -o attn -- python3 $R/tools/bench_attn.py > $O/run_$dbg.log 2>&1
  python3 - $dbg <<'PY'
import csv, os, sys, glob
dbg = sys.argv[1]
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/exp_attn/d" + dbg
f = glob.glob(O + "/**/attn_kernel_stats.csv", recursive=True)[0]
out = []
for r in csv.DictReader(open(f)):
    n = r["Name"]
    for k in ("flash_fwd_res", "flash_fwd_rel", "flash_bwd_rel", "attn_dqde", "flash_delta", "attn_fwd2", "attn_bwd2"):
        if k in n:
            out.append("%s %.1f us" % (k, float(r["AverageNs"]) / 1e3))
print("debug %4s: %s" % (dbg, ", ".join(out)), flush=True)
PY
done
