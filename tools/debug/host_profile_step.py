"""cProfile of the host side of the default bench step (GPU box): where the Python / torch time of one training step goes.
usage: python tools/debug/host_profile_step.py [steps]"""
import cProfile
import io
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--steps", sys.argv[1] if len(sys.argv) > 1 else "20", "--warmup", "3", "--no-cpu-baseline", "--no-two-call", "--no-fp32-form", "--no-graph-form",
            "--no-sync-form"]
import bench  # noqa: E402

pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue()[:9000])
