#!/bin/bash
# same-box A/B of two builds of libttmi.so: ab/libttmi_prev.so (built from another commit) against the tree's library, alternating runs.
# usage (GPU box): bash tools/ab_bench.sh [extra bench.py args]
set -e
L=transformer-transducer_amd/ttmi/libttmi.so
cp $L /tmp/libttmi_new.so
for i in 1 2 3; do
  cp ab/libttmi_prev.so $L; echo -n "prev "; python bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'
  cp /tmp/libttmi_new.so $L; echo -n "new  "; python bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done
