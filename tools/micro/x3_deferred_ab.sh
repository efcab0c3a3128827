#!/bin/bash
# bf16x3 mode: the two-call path with materialised f32 logits (default) against Transducer.loss (fused, chunked joint + loss; loss gradient as bf16 planes, or not)
cd $GRAFT_REPO_ROOT
B="python3 bench.py --precision bf16x3 --steps 6 --warmup 2 --no-cpu-baseline --no-two-call --no-fp32-form --no-graph-form --no-sync-form --no-trajectory"
for r in 1 2; do
  for cfg in "|--loss-form two-call" "TTMI_X3_SPLIT_GRAD=1|--loss-form fused" "TTMI_X3_SPLIT_GRAD=0|--loss-form fused" "TTMI_X3_SPLIT_GRAD=1|--loss-form fused --loss-chunk 32"; do
    e=${cfg%%|*}; a=${cfg##*|}
    env $e $B $a 2>/dev/null | grep '^{' | tail -1 | python3 -c "import sys, json; j = json.loads(sys.stdin.read()); print('[$e] $a: %.2f ms per step' % j['ms_per_step'])"
  done
done
