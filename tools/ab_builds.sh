#!/bin/bash
# same-box A/B of two BUILDS of the library: the in-tree libttmi.so against another one (TTMI_LIB), alternating runs of the default bench.
# usage (GPU box): bash tools/ab_builds.sh transformer-transducer_amd/ttmi/libttmi_prev.so [extra bench.py args]
OTHER=$(readlink -f $1); shift
show() { python3 -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1])
s = j.get('ms_per_step_spread') or {}
print('%-10s step %.3f ms (median %s) | joint fwd %.3f ms | attention bwd %s ms' % (sys.argv[1], j['ms_per_step'], s.get('median'), j['roofline']['kernel_ms'], (j.get('roofline_attn') or {}).get('kernel_ms')))" "$1"; }
for i in 1 2 3; do
  TTMI_LIB= python3 bench.py --no-cpu-baseline --no-fp32-form --no-graph-form --no-two-call --steps 20 --warmup 5 "$@" 2>/dev/null | show in-tree
  TTMI_LIB=$OTHER python3 bench.py --no-cpu-baseline --no-fp32-form --no-graph-form --no-two-call --steps 20 --warmup 5 "$@" 2>/dev/null | show other
done
