#!/bin/bash
# FETCH_SIZE of the joint's forward / dgrad launches under the two tile walks of gemm_nt_bf16_v8_kernel (TTMI_TILE_WALK=0 / 1), GPU box
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in 0 1; do
  O=$R/gpurun_out/pmc_walk$w; rm -rf $O; mkdir -p $O
  (cd $R && export TTMI_TILE_WALK=$w && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-two-call --no-fp32-form --no-graph-form --no-sync-form > $O/run.log 2>&1)
  echo "== TTMI_TILE_WALK=$w"
  python3 $R/tools/pmc_summary.py $O FETCH "gemm_nt_bf16_v8_kernel<unsigned short, 3>" "gemm_nt_bf16_v8_kernel<unsigned short, 4>" | sort | uniq -c | sort -rn | head -4
done
