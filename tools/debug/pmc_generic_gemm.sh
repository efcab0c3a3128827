#!/bin/bash
# HBM-side bytes of the bf16x3 attention-core kernels on the microbenchmark shapes: separate FETCH_SIZE / WRITE_SIZE passes (GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_gg; rm -rf $O; mkdir -p $O
(cd $R && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 tools/debug/bench_generic_gemm.py > $O/fetch.log 2>&1) &&
(cd $R && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 tools/debug/bench_generic_gemm.py > $O/write.log 2>&1) &&
(cd $R && rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/tcc -- python3 tools/debug/bench_generic_gemm.py > $O/tcc.log 2>&1)
for k in x3_panel64 x3_rows_nt64; do
  python3 $R/tools/pmc_summary.py $O/fetch FETCH $k | sort | uniq -c | sort -rn | head -6
  python3 $R/tools/pmc_summary.py $O/write WRITE $k | sort | uniq -c | sort -rn | head -6
  python3 $R/tools/pmc_summary.py $O/tcc TCC $k | sort | awk 'NR%23==1' | head -12
done
