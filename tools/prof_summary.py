#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats --output-format csv run into a small text table
(committed under profiles/).  usage: prof_summary.py <dir with *_kernel_stats.csv> <out.txt> [iters]"""
import collections
import csv
import glob
import sys


def main():
    d, out = sys.argv[1], sys.argv[2]
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    f = glob.glob(d + "/**/*_kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    lines = ["# rocprofv3 --kernel-trace --stats summary (%s), %d iterations profiled" % (f.split("/")[-1], iters),
             "# total GPU kernel time %.3f ms (%.3f ms / iteration)" % (tot / 1e6, tot / 1e6 / iters),
             "%-70s %8s %12s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "%")]
    for r in rows[:40]:
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0][:70]
        lines.append("%-70s %8s %12.1f %10.2f %7.2f" % (name, r["Calls"], float(r["TotalDurationNs"]) / 1e3,
                                                       float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    tr = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)
    if tr:
        g = collections.defaultdict(list)
        for r in csv.DictReader(open(tr[0])):
            n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
            g[(n, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1))].append(
                (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        lines.append("")
        lines.append("# per (kernel, workgroups) : calls/iter, median us, total us/iter")
        for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1]))[:40]:
            v.sort()
            lines.append("%-50s wg=%-6d %7.1f %9.2f %10.1f" % (k[0], k[1], len(v) / iters, v[len(v) // 2], sum(v) / iters))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
