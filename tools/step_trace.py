#!/usr/bin/env python3
"""Per-launch durations of one decoder step from a rocprofv3 --kernel-trace run of tools/step_group_run.py.
usage: step_trace.py <rocprof output dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if any(k in r["Kernel_Name"] for k in ("skf_kernel", "sk_kernel", "attn_fwd", "step_prep"))]
n = len(sel)
mid = sel[n // 2: n // 2 + 12]
prev = None
for r in mid:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print("%-28s grid %6s x %s  wg %4s  dur %6.2f us  start-to-start %6.2f" % (
        name[:28], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Grid_Size_Y"], r["Workgroup_Size_X"], (e - s) / 1e3,
        ((s - prev) / 1e3) if prev else 0))
    prev = s
