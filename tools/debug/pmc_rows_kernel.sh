#!/bin/bash
# where the waves of x3_rows_nt64_kernel spend their cycles (SQ counters, microbenchmark shapes), GPU box
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_rows; rm -rf $O; mkdir -p $O
(cd $R && rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/a -- python3 tools/debug/bench_generic_gemm.py > $O/a.log 2>&1) &&
(cd $R && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU --output-format csv -d $O/b -- python3 tools/debug/bench_generic_gemm.py > $O/b.log 2>&1) &&
(cd $R && rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/c -- python3 tools/debug/bench_generic_gemm.py > $O/c.log 2>&1)
python3 - "$O" <<'PY'
import csv,glob,sys,collections
for sub in "abc":
    acc=collections.defaultdict(list)
    for f in glob.glob(sys.argv[1]+"/"+sub+"/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            if "x3_rows_nt64_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        # launches alternate plain / beta forms: print the two groups by launch parity
        print("%-26s n=%3d  first-form mean %.4e   (all: min %.3e max %.3e)"%(k,len(v),sum(v[:len(v)//2])/max(1,len(v)//2),min(v),max(v)))
PY
