#!/usr/bin/env python3
"""Median duration per kernel from a rocprofv3 --kernel-trace CSV directory, grouped by
consecutive frames of one workload.   python tools/kernel_split.py <dir>"""
import csv
import glob
import sys
from collections import OrderedDict

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
frames, cur = [], None
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if n.startswith("__amd"):
        continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if n.startswith("trace_kernel"):
        cur = OrderedDict()
        frames.append(cur)
    if cur is not None:
        cur[n] = d
for fr in frames:
    print("  ".join("%s %.3f" % (k, v) for k, v in fr.items()), " total %.3f" % sum(fr.values()))
