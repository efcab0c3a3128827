#!/bin/bash
# One round's committed profile set, on the GPU box (gpurun): kernel-trace stats of the default bench, the exclusive-time attribution of
# one step, and three separate --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ + GRBM: TCC slots do not fit both sizes in one pass).
# usage: tools/profile_round.sh <round tag, e.g. r02> <commit>
set -e
TAG=$1; COMMIT=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT profiles
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-two-call --no-fp32-form --no-graph-form --no-sync-form > $OUT/trace.log 2>&1
python3 tools/prof_summary.py $OUT/trace profiles/${TAG}_bench_kernel_stats.csv "bench.py --steps 4 --warmup 2 --no-two-call (6 steps, default = exp-domain loss form), C2, $TAG build $COMMIT" > /dev/null
python3 tools/kernel_exclusive.py $OUT/trace > profiles/${TAG}_bench_step_attribution.txt
python3 tools/gpu_busy.py $OUT/trace >> profiles/${TAG}_bench_step_attribution.txt
# the reference's own call sequence (model(inputs, targets) + RNNTLoss) in a trace of its own
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace2 -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fp32-form --no-graph-form --no-sync-form --loss-form two-call > $OUT/trace2.log 2>&1
python3 tools/prof_summary.py $OUT/trace2 profiles/${TAG}_bench_two_call_kernel_stats.csv "bench.py --steps 4 --warmup 2 --loss-form two-call (6 steps), C2, $TAG build $COMMIT" > /dev/null
python3 tools/kernel_exclusive.py $OUT/trace2 > profiles/${TAG}_bench_two_call_step_attribution.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-two-call --no-fp32-form --no-graph-form --no-sync-form > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-two-call --no-fp32-form --no-graph-form --no-sync-form > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-two-call --no-fp32-form --no-graph-form --no-sync-form > $OUT/sq.log 2>&1
F=profiles/${TAG}_pmc_joint_kernels.txt
echo "# rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE> --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline ($TAG build $COMMIT; three separate passes; per-launch values, KB as reported)" > $F
for k in "gemm_nt_bf16_v8_kernel<unsigned short, 3>" "gemm_nt_bf16_v8_kernel<unsigned short, 4>" "gemm_tn_bf16_v8_kernel<2>" "rnnt_prep_exp_kernel" "rnnt_scale_exp_kernel" "gemm_nt_bf16_v8_kernel<unsigned short, 1>" "gemm_tn_bf16_v8_kernel<1>" "rnnt_lse_kernel" "rnnt_grad_kernel" "flash_bwd_rel2_kernel<0>" "flash_fwd_res_kernel<0>" "attn_dqde_kernel" "rnnt_lattice_lds_kernel" "ln_bwd_fused_kernel<2>" "ln_fwd_kernel"; do
  python3 tools/pmc_summary.py $OUT/fetch FETCH "$k" | sort | awk 'NR%4==1' >> $F
  python3 tools/pmc_summary.py $OUT/write WRITE "$k" | sort | awk 'NR%4==1' >> $F
done
python3 tools/pmc_summary.py $OUT/sq SQ "gemm_nt_bf16_v8_kernel<unsigned short, 3>" "gemm_nt_bf16_v8_kernel<unsigned short, 4>" "gemm_tn_bf16_v8_kernel<2>" "gemm_nt_bf16_v8_kernel<unsigned short, 1>" "gemm_tn_bf16_v8_kernel<1>" "flash_bwd_rel2_kernel<0>" "flash_fwd_res_kernel<0>" "attn_dqde_kernel" | sort >> $F
python3 tools/update_pmc_json.py $OUT/fetch $OUT/write $OUT/sq $COMMIT $TAG > $OUT/pmc_json.log
cp profiles/${TAG}_* profiles/pmc_joint_projection.json profiles/pmc_secondary_kernels.json gpurun_out/ 2>/dev/null || true
tail -3 $OUT/pmc_json.log
