#!/usr/bin/env python3
"""Text timeline of ONE training iteration from a rocprofv3 --kernel-trace csv: which kernels ran on which HIP
stream (queue), when, and how much of the iteration each stream / the GPU was busy.  Iterations are delimited by
pack_kernel (the weight re-pack at the head of a forward).   usage: timeline.py <dir with *_kernel_trace.csv> [iteration index from the end, default 1] [from_us to_us: list every kernel in the window]"""
import csv
import glob
import sys


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]


def union(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def main():
    d = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
    rows = [dict(name=short(r["Kernel_Name"]), q=r.get("Queue_Id", "0"), s=int(r["Start_Timestamp"]), e=int(r["End_Timestamp"]))
            for r in csv.DictReader(open(f))]
    rows.sort(key=lambda r: r["s"])
    # an iteration starts with the weight re-pack (one pack_kernel per optimizer step, first thing of the next forward); the
    # update itself is several clip_adam launches on two streams (train.ClipAdam(overlap=True)), so it cannot delimit
    # -- the one on the MAIN queue only: since round 5 the decoder's tiles are re-packed early on the optimizer's side stream
    # (model.pack_early), a second pack_kernel per iteration that must not delimit
    from collections import Counter
    main_queue = Counter(r["q"] for r in rows).most_common(1)[0][0]
    marks = [i for i, r in enumerate(rows) if r["name"].startswith("pack_kernel") and r["q"] == main_queue]
    if len(marks) > back + 1:
        lo, hi = marks[-back - 1], marks[-back]
    else:
        adam = [i for i, r in enumerate(rows) if r["name"].startswith("clip_adam")]
        lo, hi = adam[-back - 1] + 1, adam[-back] + 1
    it = rows[lo:hi]
    t0, t1 = it[0]["s"], max(r["e"] for r in it)
    qs = sorted({r["q"] for r in it}, key=lambda q: -sum(r["e"] - r["s"] for r in it if r["q"] == q))
    print("# NOTE: under rocprofv3 the enqueueing thread is slower than the GPU in places (a launch costs it ~10 us): idle gaps on the main\n"
          "# queue inside the loops -- the ~200-250 us one in the reverse-time loop in particular -- are host starvation that the\n"
          "# un-profiled run does not have (docs/EXPERIMENTS.md round 5: per-step events, 63-65 us per step, no hole)")
    print("iteration span %.3f ms, %d kernels, queues %s" % ((t1 - t0) / 1e6, len(it), qs))
    for q in qs:
        iv = [(r["s"], r["e"]) for r in it if r["q"] == q]
        print("  queue %s: %d kernels, busy %.3f ms" % (q, len(iv), union(iv) / 1e6))
    print("  GPU busy (union) %.3f ms; both-streams-busy %.3f ms" % (
        union([(r["s"], r["e"]) for r in it]) / 1e6,
        (sum(union([(r["s"], r["e"]) for r in it if r["q"] == q]) for q in qs) - union([(r["s"], r["e"]) for r in it])) / 1e6))
    # merged runs per queue
    print("\n%9s %9s %5s  %-3s %s" % ("start_us", "dur_us", "n", "q", "kernels"))
    runs = []
    for q in qs:
        cur = None
        for r in (r for r in it if r["q"] == q):
            key = r["name"]
            if cur and (key in cur["names"] or len(cur["names"]) < 6) and r["s"] - cur["e"] < 30000 and cur["n"] < 400 and \
                    (key in cur["names"] or cur["n"] < 6 * 1):
                cur["names"].setdefault(key, 0); cur["names"][key] += 1; cur["e"] = max(cur["e"], r["e"]); cur["n"] += 1
                cur["busy"] += r["e"] - r["s"]
            else:
                cur = dict(q=q, s=r["s"], e=r["e"], n=1, names={key: 1}, busy=r["e"] - r["s"])
                runs.append(cur)
    if len(sys.argv) > 4:          # detail window [a, b) in us: every kernel
        a, b = float(sys.argv[3]), float(sys.argv[4])
        print("\ndetail %g..%g us" % (a, b))
        for r in it:
            ts = (r["s"] - t0) / 1e3
            if a <= ts < b:
                print("%9.1f %8.1f  q%d  %s" % (ts, (r["e"] - r["s"]) / 1e3, qs.index(r["q"]), r["name"]))
        return
    for r in sorted(runs, key=lambda r: r["s"]):
        print("%9.1f %9.1f %5d  %-3s %s" % ((r["s"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3, r["n"], qs.index(r["q"]),
                                            ", ".join("%s x%d" % kv for kv in r["names"].items())))
    # the two decoder recurrences as blocks: forward = first to last attention-forward launch (+ the skinny launch behind it),
    # reverse-time = the skinny launch in front of the first attention-backward launch to the one behind the last
    main_q = qs[0]
    mk = [r for r in it if r["q"] == main_q]
    for tag, key in (("forward decoder loop", "attn_fwd"), ("reverse-time decoder loop", "attn_bwd_split")):
        idx = [i for i, r in enumerate(mk) if r["name"].startswith(key)]
        if len(idx) >= 2:
            a = mk[max(idx[0] - 1, 0)]["s"]
            b = mk[min(idx[-1] + 1, len(mk) - 1)]["e"]
            gaps = [(mk[i + 1]["s"] - mk[i]["e"]) / 1e3 for i in range(max(idx[0] - 1, 0), min(idx[-1] + 1, len(mk) - 1))]
            big = [g for g in gaps if g > 20]
            print("\n%s (main stream): %d steps in %.1f us = %.1f us per step; idle gaps > 20 us inside the block: %s"
                  % (tag, len(idx), (b - a) / 1e3, (b - a) / 1e3 / len(idx), ["%.0f" % g for g in big] or "none"))


if __name__ == "__main__":
    main()
