#!/bin/bash
# same-box A/B of two builds of libttmi.so: ab/libttmi_prev.so (built from another commit) against the tree's library, alternating runs.
# usage (GPU box): bash tools/ab_bench.sh [extra bench.py args]      prints ms/step | joint projection ms | loss op ms
set -e
L=transformer-transducer_amd/ttmi/libttmi.so
cp $L /tmp/libttmi_new.so
show() { python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1])
r = j['roofline'] if j['roofline']['bound'] == 'mfma' else j['roofline_joint']
l = j.get('roofline_loss') or j['roofline']
print('%s  step %.3f ms | joint fwd %.3f ms | loss op %s ms' % (sys.argv[1], j['ms_per_step'], r['kernel_ms'], l.get('kernel_ms')))" "$1"; }
for i in 1 2 3; do
  cp ab/libttmi_prev.so $L; python bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>/dev/null | show prev
  cp /tmp/libttmi_new.so $L; python bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>/dev/null | show new
done
