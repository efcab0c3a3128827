#!/usr/bin/env python3
"""From a rocprofv3 kernel trace (kernel_trace.csv): per kernel name, over the last full training step (between two optimiser kernels),
 - calls, total duration,
 - EXCLUSIVE time: time during which it is the only kernel on the device (what removing it would save at most),
 - SHARED time: time it runs beside other kernels, split evenly among the kernels running at that moment.
The sum of exclusive + shared over all kernels = the busy (union) time of the step, so the table attributes the step's GPU time.
usage: kernel_exclusive.py <rocprof dir> [step index from the end, default 1]"""
import csv
import glob
import sys
from collections import defaultdict

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "sgd_kernel" in r[2] or "adam_kernel" in r[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
a, b = marks[-k - 1], marks[-k]
seg = rows[a + 1:b + 1]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    if n.startswith("at::native"):
        n = "torch:" + n.split("<")[0].split("::")[-1]
    return n[:70]


events = []
for i, (s, e, n) in enumerate(seg):
    events.append((s, 1, i))
    events.append((e, 0, i))
events.sort()
active = set()
excl, shared, total, calls = defaultdict(float), defaultdict(float), defaultdict(float), defaultdict(int)
prev = events[0][0]
for t, kind, i in events:
    dt = t - prev
    if dt > 0 and active:
        if len(active) == 1:
            excl[short(seg[next(iter(active))][2])] += dt
        else:
            for j in active:
                shared[short(seg[j][2])] += dt / len(active)
    prev = t
    if kind == 1:
        active.add(i)
    else:
        active.discard(i)
for s, e, n in seg:
    total[short(n)] += e - s
    calls[short(n)] += 1
span = seg[-1][1] - seg[0][0]
busy = sum(excl.values()) + sum(shared.values())
print("step: %d kernels, span %.2f ms, busy %.2f ms, sum of durations %.2f ms" % (len(seg), span / 1e6, busy / 1e6, sum(total.values()) / 1e6))
print("%-72s %6s %9s %9s %9s %9s" % ("kernel", "calls", "total_ms", "excl_ms", "shared_ms", "attrib_ms"))
for n in sorted(total, key=lambda n: -(excl[n] + shared[n])):
    print("%-72s %6d %9.3f %9.3f %9.3f %9.3f" % (n, calls[n], total[n] / 1e6, excl[n] / 1e6, shared[n] / 1e6, (excl[n] + shared[n]) / 1e6))
