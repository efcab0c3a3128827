#!/usr/bin/env python3
"""Per-launch values of one rocprofv3 --pmc counter for the kernels whose name contains any of the given substrings.
usage: pmc_summary.py <rocprof output dir> <label> <substr> [<substr> ...]   (appends to stdout; GPU box only)"""
import csv
import glob
import sys

d, label, subs = sys.argv[1], sys.argv[2], sys.argv[3:]
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if any(s in n for s in subs):
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            print("%-6s %-60s grid %10s  %s %.4e  dur %.1f us" % (label, n.replace("(anonymous namespace)::", "")[:60], r["Grid_Size"],
                                                                 r["Counter_Name"], float(r["Counter_Value"]), dur))
