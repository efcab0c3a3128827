#!/bin/bash
# The bench / diagnostic logs a round commits under profiles/ (GPU box, after tools/profile_round.sh): one line per workload, the greedy decoder,
# the lattice kernels alone, the bf16 loss-error decomposition, one encoder layer's attention kernels.   usage: tools/round_logs.sh <tag>
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
python3 bench.py --steps 50 --warmup 10 > $O/${TAG}_bench_c2.log 2>&1 && tail -c 300 $O/${TAG}_bench_c2.log && echo
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_c2_20steps.log 2>&1 && echo "driver-form run done"
# the other BASELINE workloads: sustained figures (10 warm-up + 50 timed steps, median / p10 / p90 in the line) ...
for w in c4-band c4-chunk c5; do
  python3 bench.py --workload $w --steps 50 --warmup 10 --no-cpu-baseline --no-fp32-form --no-graph-form --no-sync-form > $O/${TAG}_bench_$w.log 2>&1 && echo "$w done"
done
# ... and their kernel statistics / step attribution (one trace per workload)
for w in c4-band c5; do
  P=$O/prof_${TAG}_$w; rm -rf $P
  rocprofv3 --kernel-trace --stats --output-format csv -d $P -- python3 bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-two-call --no-fp32-form --no-graph-form --no-sync-form > $P.log 2>&1
  python3 tools/prof_summary.py $P profiles/${TAG}_bench_${w}_kernel_stats.csv "bench.py --workload $w --steps 4 --warmup 2 (6 steps), $TAG build" > /dev/null
  python3 tools/kernel_exclusive.py $P > profiles/${TAG}_bench_${w}_step_attribution.txt && echo "$w profile done"
done
cp profiles/${TAG}_bench_c*_kernel_stats.csv profiles/${TAG}_bench_c*_step_attribution.txt $O/ 2>/dev/null
python3 bench.py --mode decode > $O/${TAG}_decode.log 2>&1 && echo "decode done"
python3 tools/bench_lattice.py > $O/${TAG}_lattice_bench.log 2>&1 && echo "lattice done"
python3 tools/debug/bf16_loss_error.py --steps 50 > $O/${TAG}_bf16_loss_error.log 2>&1 && echo "bf16 diag done"
bash tools/prof_attn.sh > $O/${TAG}_attention_layer_kernels.txt 2>&1 && echo "attn done"
# exact-f32 NT kernels (decode / fp32 mode) against each other, and the per-kernel picture of one batched greedy-decode pass
python3 tools/micro/f32_gemm_ab.py > $O/${TAG}_f32_gemm_kernels.txt 2>&1 && echo "f32 gemm table done"
P=$O/prof_${TAG}_decode_batch; rm -rf $P
rocprofv3 --kernel-trace --stats --output-format csv -d $P -- python3 tools/debug/decode_batch_only.py > $P.log 2>&1
python3 tools/prof_summary.py $P $O/${TAG}_decode_batch_kernel_stats.csv "tools/debug/decode_batch_only.py (3 passes of decode_batch, 32 utterances), $TAG build" > /dev/null && echo "decode trace done"

