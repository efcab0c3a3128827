#!/bin/bash
# under a reservation: grouped weight gradients against immediate (forked, K-split) ones, alternating.   usage: tools/micro/reserve_nogroup_ab.sh <r> [rounds]
cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-two-call --no-fp32-form --no-graph-form --no-sync-form --no-trajectory"
for r in $(seq 1 ${2:-2}); do
  for o in "" "--no-grouped-wgrads"; do
    for c in 0 $1; do
      TTMI_BENCH_RESERVE_CUS=$c $B $o 2>/dev/null | grep '^{' | tail -1 | python3 -c "import sys, json; j = json.loads(sys.stdin.read()); print('reserve $c [$o] step %.3f ms' % j['ms_per_step'])"
    done
  done
done
