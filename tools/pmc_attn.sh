cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_attn
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $O/a -- python3 $R/tools/bench_attn.py > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA --output-format csv -d $O/b -- python3 $R/tools/bench_attn.py > $O/b.log 2>&1
cd $R
for d in a b; do python3 tools/pmc_summary.py $O/$d PMC flash_bwd_rel flash_fwd_rel flash_fwd_res | sort | uniq -c | sort -k2 | awk '{print}' | head -60; done > $O/summary.txt
python3 - <<'PY'
import csv, glob, collections, os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/pmc_attn"
for d in "ab":
    acc=collections.defaultdict(list)
    for f in glob.glob(O+"/%s/**/*counter_collection.csv"%d, recursive=True):
        for r in csv.DictReader(open(f)):
            n=r["Kernel_Name"]
            for k in ("flash_bwd_rel2","flash_fwd_res","attn_dqde"):
                if k in n: acc[(k,r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in sorted(acc.items()): print(k, "n=%d mean=%.4e"%(len(v), sum(v)/len(v)))
PY
