# kernel-trace stats of one C2-shaped encoder layer (tools/bench_attn.py): per-kernel average durations
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_attn
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o attn -- python3 $R/tools/bench_attn.py > $O/run.log 2>&1
tail -1 $O/run.log
python3 - <<'PY'
import csv, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof_attn"
rows=list(csv.DictReader(open(O+"/attn_kernel_stats.csv")))
for r in rows[:22]:
    print("%-90s calls %4s avg %9.1f us  total %8.2f ms" % (r["Name"].replace("(anonymous namespace)::","")[:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
