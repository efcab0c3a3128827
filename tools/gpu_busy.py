#!/usr/bin/env python3
"""From a rocprofv3 kernel trace: union of kernel intervals (GPU busy time), sum of durations and wall span of the last N
dispatches-windows between sgd_kernel launches (= training steps).  usage: gpu_busy.py <rocprof dir>"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "sgd_kernel" in r[2] or "adam_kernel" in r[2]]
for a, b in zip(marks[:-1], marks[1:]):
    seg = rows[a + 1:b + 1]
    span = seg[-1][1] - seg[0][0]
    tot = sum(e - s for s, e, _ in seg)
    busy, cur_s, cur_e = 0, None, None
    for s, e, _ in seg:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    gaps = []
    cur_e, prev_n = None, ""
    for s, e, n in seg:
        if cur_e is not None and s > cur_e:
            gaps.append((s - cur_e, prev_n[:28] + " -> " + n))
        if cur_e is None or e > cur_e:
            cur_e, prev_n = e, n
    gaps.sort(reverse=True)
    print("step: %d kernels, span %.2f ms, busy (union) %.2f ms, idle %.2f ms, sum of durations %.2f ms" %
          (len(seg), span / 1e6, busy / 1e6, (span - busy) / 1e6, tot / 1e6))
    print("   idle gaps: n=%d, >20us: %d (%.2f ms), largest: %s" % (len(gaps), sum(1 for g, _ in gaps if g > 20000),
          sum(g for g, _ in gaps if g > 20000) / 1e6, "; ".join("%.0fus %s" % (g / 1e3, n[:75]) for g, n in gaps[:8])))
