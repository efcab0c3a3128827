"""Host time spent inside each C entry point of libttmi.so over the default bench's steps (GPU box): a timing proxy around the CDLL.
usage: python tools/debug/host_time_per_entry.py [steps]"""
import collections
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "transformer-transducer_amd"))
import ttmi  # noqa: E402

real = ttmi.lib()
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])


class Proxy:
    def __getattr__(self, name):
        fn = getattr(real, name)
        if not name.startswith("ttmi_") or name.endswith("_floats") or name.endswith("_floats_prec") or name in ("ttmi_last_error",):
            return fn

        class W:
            def __call__(self, *a):
                t0 = time.perf_counter()
                r = fn(*a)
                dt = time.perf_counter() - t0
                e = acc[name]
                e[0] += 1
                e[1] += dt
                e[2] = max(e[2], dt)
                return r

            def __setattr__(self, k, v):
                setattr(fn, k, v)

            def __getattr__(self, k):
                return getattr(fn, k)
        return W()


ttmi._lib = Proxy()
steps = sys.argv[1] if len(sys.argv) > 1 else "20"
sys.argv = ["bench.py", "--steps", steps, "--warmup", "5", "--no-cpu-baseline", "--no-two-call", "--no-fp32-form", "--no-graph-form", "--no-sync-form"]
import bench  # noqa: E402
bench.main()
n = int(steps) + 5 + 3
print("host time inside the library's entry points, per step (~%d steps in the run):" % n)
tot = 0.0
for k, (c, t, m) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%-34s calls/step %6.1f  host %7.3f ms/step  avg %7.1f us  max %8.1f us" % (k, c / n, t / n * 1e3, t / c * 1e6, m * 1e6))
    tot += t
print("total %.2f ms/step" % (sum(v[1] for v in acc.values()) / n * 1e3))
