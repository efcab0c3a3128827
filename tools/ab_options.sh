#!/bin/bash
# same-box A/B of a measurement switch: default options against TTMI_OPTIONS="$1" (e.g. 15=1), alternating runs of the default bench.
# usage (GPU box): bash tools/ab_options.sh 15=1 [extra bench.py args]
OPT=$1; shift
show() { python3 -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1])
s = j.get('ms_per_step_spread') or {}
print('%-10s step %.3f ms (median %s) | joint fwd %.3f ms | attention bwd %s ms' % (sys.argv[1], j['ms_per_step'], s.get('median'), j['roofline']['kernel_ms'], (j.get('roofline_attn') or {}).get('kernel_ms')))" "$1"; }
for i in 1 2 3; do
  TTMI_OPTIONS= python3 bench.py --no-cpu-baseline --no-fp32-form --no-graph-form --no-two-call --steps 20 --warmup 5 "$@" 2>/dev/null | show default
  TTMI_OPTIONS=$OPT python3 bench.py --no-cpu-baseline --no-fp32-form --no-graph-form --no-two-call --steps 20 --warmup 5 "$@" 2>/dev/null | show "$OPT"
done
