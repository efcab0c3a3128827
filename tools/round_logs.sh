#!/bin/bash
# The bench / diagnostic logs a round commits under profiles/ (GPU box, after tools/profile_round.sh): one line per workload, the greedy decoder,
# the lattice kernels alone, the bf16 loss-error decomposition, one encoder layer's attention kernels.   usage: tools/round_logs.sh <tag>
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
python3 bench.py > $O/${TAG}_bench_c2.log 2>&1 && tail -c 300 $O/${TAG}_bench_c2.log && echo
for w in c4-band c4-chunk c5; do
  python3 bench.py --workload $w --no-cpu-baseline --no-fp32-form --no-graph-form > $O/${TAG}_bench_$w.log 2>&1 && echo "$w done"
done
python3 bench.py --mode decode > $O/${TAG}_decode.log 2>&1 && echo "decode done"
python3 tools/bench_lattice.py > $O/${TAG}_lattice_bench.log 2>&1 && echo "lattice done"
python3 tools/debug/bf16_loss_error.py --steps 50 > $O/${TAG}_bf16_loss_error.log 2>&1 && echo "bf16 diag done"
bash tools/prof_attn.sh > $O/${TAG}_attention_layer_kernels.txt 2>&1 && echo "attn done"
