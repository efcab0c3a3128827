#!/usr/bin/env python3
"""Host-side profile of bench.py's training step (cProfile over 5 steps after warm-up): where the CPU time per step goes.
GPU box only.  usage: python tools/host_profile.py [bench.py args]"""
import cProfile
import os
import pstats
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = [os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "5", "--warmup", "2"] + sys.argv[1:]
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(sys.argv[0], run_name="__main__")
finally:
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(35)
