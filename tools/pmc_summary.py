#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output: one line per kernel dispatch with every counter.

    python tools/pmc_summary.py <dir given to rocprofv3 -d> [kernel substring]
"""
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    key = sys.argv[2] if len(sys.argv) > 2 else ""
    files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
    rows = {}
    for f in files:
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                name = r["Kernel_Name"].split("(")[0]
                if key and key not in name:
                    continue
                k = (int(r["Dispatch_Id"]), name)
                d = rows.setdefault(k, {})
                d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                d["_dur_ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
                d["_vgpr"] = r.get("VGPR_Count", "")
    for (disp, name), d in sorted(rows.items()):
        print("%s dispatch %d: %s" % (name, disp, " ".join(
            "%s=%s" % (c, ("%.6g" % v) if isinstance(v, float) else v) for c, v in sorted(d.items()))))


main()
