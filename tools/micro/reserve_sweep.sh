#!/bin/bash
# the CU reservation's price on one GPU, alternating runs on one box.   usage: tools/micro/reserve_sweep.sh "0 4 8" [rounds]
cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-two-call --no-fp32-form --no-graph-form --no-sync-form --no-trajectory"
for r in $(seq 1 ${2:-2}); do
  for c in $1; do
    TTMI_BENCH_RESERVE_CUS=$c $B 2>/dev/null | grep '^{' | tail -1 | python3 -c "import sys, json; j = json.loads(sys.stdin.read()); print('reserve $c step %.3f ms' % j['ms_per_step'])"
  done
done
