#!/usr/bin/env python3
"""Kernels of one traced iteration whose duration is far above the median of their (name, grid) group -- how a launch that
waits for resources (an empty CU, LDS) beside another stream's kernels shows up.  usage: outliers.py <dir with *_kernel_trace.csv> [factor]"""
import csv, glob, statistics, sys
from collections import defaultdict


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]


def main():
    d = sys.argv[1]
    fac = float(sys.argv[2]) if len(sys.argv) > 2 else 2.5
    f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "pack_kernel" in r["Kernel_Name"]]
    lo, hi = (marks[-2], marks[-1]) if len(marks) >= 2 else (0, len(rows))
    t0 = int(rows[lo]["Start_Timestamp"])
    groups = defaultdict(list)
    for r in rows:                                   # medians over ALL iterations
        key = (short(r["Kernel_Name"]), r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""))
        groups[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    med = {k: statistics.median(v) for k, v in groups.items()}
    print("%10s %9s %9s  q  kernel" % ("start_us", "dur_us", "median"))
    for r in rows[lo:hi]:
        key = (short(r["Kernel_Name"]), r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""))
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if dur > fac * med[key] and dur > 20:
            print("%10.1f %9.1f %9.1f  %s  %s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, dur, med[key], r.get("Queue_Id", "?"), key[0][-44:], key[1]))


if __name__ == "__main__":
    main()
