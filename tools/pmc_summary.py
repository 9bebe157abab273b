#!/usr/bin/env python3
"""Per-kernel averages of the counters in one or more rocprofv3 --pmc output dirs -> text table.
usage: pmc_summary.py <out.txt> <title> <dir> [<dir> ...]"""
import collections, csv, glob, sys


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]


def main():
    out, title, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines = ["# " + title, "# per-dispatch averages (counter totals over the launch)"]
    for k in sorted(agg, key=lambda k: -sum(sum(v) for v in agg[k].values())):
        c = agg[k]
        n = max(len(v) for v in c.values())
        if n < 20:
            continue
        lines.append("%-46s dispatches %d" % (k, n))
        for name in sorted(c):
            lines.append("    %-32s %14.1f" % (name, sum(c[name]) / len(c[name])))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CU_CYCLES" in c:
            b = sum(c["SQ_BUSY_CU_CYCLES"]) / len(c["SQ_BUSY_CU_CYCLES"])
            m = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
            if b > 0:
                lines.append("    -> MFMA busy / (4 SIMDs x CU busy)   %13.1f %%" % (100.0 * m / (4 * b)))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
