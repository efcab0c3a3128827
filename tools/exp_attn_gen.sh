# per-kernel average durations of one C2-shaped encoder layer (tools/bench_attn.py) under TTMI_OPTIONS settings given as arguments, e.g.
#   tools/exp_attn_gen.sh 15:2 15:3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/exp_attn_gen
mkdir -p $O
for opt in "$@"; do
  tag=$(echo $opt | tr ':,' '__')
  rm -rf $O/d$tag
  TTMI_OPTIONS=$opt rocprofv3 --kernel-trace --stats --output-format csv -d $O/d$tag -o attn -- python3 $R/tools/bench_attn.py > $O/run_$tag.log 2>&1
  python3 - $tag <<'PY'
import csv, os, sys, glob
tag = sys.argv[1]
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/exp_attn_gen/d" + tag
f = glob.glob(O + "/**/attn_kernel_stats.csv", recursive=True)[0]
out = []
for r in csv.DictReader(open(f)):
    n = r["Name"]
    for k in ("flash_fwd_res", "flash_fwd_rel", "flash_bwd_rel", "attn_dqde", "flash_delta"):
        if k in n:
            out.append("%s %.1f us" % (n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0], float(r["AverageNs"]) / 1e3))
print("options %8s: %s" % (tag, ", ".join(out)), flush=True)
PY
done
