#!/bin/bash
# A/B on one box, alternating: where and how the label encoder's two passes run beside the audio encoder.   usage: tools/micro/label_overlap_ab.sh [rounds]
cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-two-call --no-fp32-form --no-graph-form --no-sync-form --no-trajectory"
for r in $(seq 1 ${1:-2}); do
  for cfg in "base|" "serial|TTMI_BENCH_OVERLAP_LABEL=0" "side-low|TTMI_SIDE_STREAM_PRIORITY=1" "side-high|TTMI_SIDE_STREAM_PRIORITY=-1" "no-value|TTMI_LABEL_VALUE_PRECISION=off" "no-value-serial|TTMI_LABEL_VALUE_PRECISION=off TTMI_BENCH_OVERLAP_LABEL=0"; do
    n=${cfg%%|*}; e=${cfg##*|}
    ms=$(env $e $B 2>/dev/null | grep '^{' | tail -1 | python3 -c "import sys, json; j = json.loads(sys.stdin.read()); print('%.3f ms (median %.3f)' % (j['ms_per_step'], j['two_call_form']['spread']['median'] if 'spread' in j.get('two_call_form', {}) else j['ms_per_step']))")
    echo "round $r  $n: $ms"
  done
done
