#!/bin/bash
# reservation r with and without an option, alternating.   usage: tools/micro/reserve_opt_ab.sh <r> "<TTMI_OPTIONS value>" [rounds]
cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-two-call --no-fp32-form --no-graph-form --no-sync-form --no-trajectory"
for r in $(seq 1 ${3:-2}); do
  for o in "" "$2"; do
    TTMI_OPTIONS=$o TTMI_BENCH_RESERVE_CUS=$1 $B 2>/dev/null | grep '^{' | tail -1 | python3 -c "import sys, json; j = json.loads(sys.stdin.read()); print('reserve $1 options [$o] step %.3f ms' % j['ms_per_step'])"
  done
done
