# per-kernel average durations of one C2-shaped encoder layer (tools/bench_attn.py) for the in-tree library and for other builds of it, alternating:
#   tools/exp_attn_lib.sh ab/libttmi_split.so [rounds]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OTHER=$(readlink -f $R/$1); ROUNDS=${2:-2}
O=$R/gpurun_out/exp_attn_lib
mkdir -p $O
for i in $(seq 1 $ROUNDS); do for v in tree other; do
  rm -rf $O/d$v
  if [ $v = other ]; then export TTMI_LIB=$OTHER; else export TTMI_LIB=; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/d$v -o attn -- python3 $R/tools/bench_attn.py > $O/run_$v.log 2>&1
  python3 - $v <<'PY'
import csv, os, sys, glob
tag = sys.argv[1]
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/exp_attn_lib/d" + tag
f = glob.glob(O + "/**/attn_kernel_stats.csv", recursive=True)[0]
out = []
for r in csv.DictReader(open(f)):
    n = r["Name"]
    for k in ("flash_fwd_res", "flash_bwd_rel", "attn_dqde", "flash_delta"):
        if k in n:
            out.append("%s %.1f us" % (n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0], float(r["AverageNs"]) / 1e3))
print("%6s: %s" % (tag, ", ".join(out)), flush=True)
PY
done; done
