cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/exp_attn
for B in 8 16 32 64; do
  O=$R/gpurun_out/exp_attn/b$B; rm -rf $O
  B=$B rocprofv3 --kernel-trace --stats --output-format csv -d $O -o attn -- python3 $R/tools/bench_attn.py > $O.log 2>&1
  python3 - $B <<'PY'
import csv, os, sys, glob
B = sys.argv[1]
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/exp_attn/b" + B
f = glob.glob(O + "/**/attn_kernel_stats.csv", recursive=True)[0]
out = []
for r in csv.DictReader(open(f)):
    n = r["Name"]
    for k in ("flash_fwd_res", "flash_bwd_rel", "attn_dqde", "flash_delta"):
        if k in n:
            out.append("%s %.1f us" % (k, float(r["AverageNs"]) / 1e3))
print("B %4s: %s" % (B, ", ".join(out)), flush=True)
PY
done
