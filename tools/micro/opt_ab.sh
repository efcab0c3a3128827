#!/bin/bash
# default bench step with and without a TTMI_OPTIONS setting, alternating on one box.   usage: tools/micro/opt_ab.sh "<options>" [rounds]
cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-two-call --no-fp32-form --no-graph-form --no-sync-form --no-trajectory"
for r in $(seq 1 ${2:-3}); do
  for o in "" "$1"; do
    TTMI_OPTIONS=$o $B 2>/dev/null | grep '^{' | tail -1 | python3 -c "import sys, json; j = json.loads(sys.stdin.read()); print('options [$o] step %.3f ms' % j['ms_per_step'])"
  done
done
