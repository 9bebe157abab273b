"""Diagnosis: split a rocprofv3 kernel trace (kernel_trace.csv) at its largest idle gap and compare the two halves:
per-kernel mean duration, busy time per queue, wall time.  (Used to find out why the second model of a process ran slower.)"""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", "?")) for r in rows))
# split at the biggest gap between consecutive kernel starts
gaps = [(ev[i + 1][0] - ev[i][0], i) for i in range(len(ev) - 1)]
g, i = max(gaps)
halves = [ev[:i + 1], ev[i + 1:]]
for h, name in zip(halves, ("first", "second")):
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    h = h[-n:]                         # the last n kernels of the half: steady state
    wall = h[-1][1] - h[0][0]
    busy = collections.Counter(); cnt = collections.Counter(); q = collections.Counter()
    for s, e, k, qq in h:
        busy[k] += e - s; cnt[k] += 1; q[qq] += e - s
    print("== %s half: %d kernels, wall %.3f ms, sum of kernel time %.3f ms, queues %s" % (name, len(h), wall / 1e6, sum(busy.values()) / 1e6,
          {k: round(v / 1e6, 2) for k, v in q.items()}))
    for k, v in busy.most_common(12):
        print("   %-60s n=%5d  mean %.2f us  total %.3f ms" % (k, cnt[k], v / cnt[k] / 1e3, v / 1e6))
