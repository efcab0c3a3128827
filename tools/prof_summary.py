#!/usr/bin/env python3
"""Trim a rocprofv3 kernel_stats.csv into a short committed summary (kernel names shortened)."""
import csv
import glob
import sys

src = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(src)))
with open(sys.argv[2], "w") as o:
    o.write("# %s\n" % sys.argv[3])
    o.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
    for r in rows:
        n = r["Name"].replace("(anonymous namespace)::", "")
        if n.startswith("void at::native") or n.startswith("at::native"):
            n = "torch:" + n.split("<")[0].split("::")[-1] + "<" + (n.split("<", 1)[1][:60] if "<" in n else "")
        if len(n) > 120:
            n = n[:120] + "..."
        o.write('"%s",%s,%s,%s,%s,%s,%s\n' % (n, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]))
print(open(sys.argv[2]).read()[:6000])
