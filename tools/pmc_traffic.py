#!/usr/bin/env python3
"""HBM traffic of the decoder-step launch group from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
tools/step_group_run.py.  FETCH_SIZE is doubled (gfx950: it reports 1/2 of a wide coalesced stream,
MI355X_MICROARCH.md section HBM); WRITE_SIZE is taken as reported (uncalibrated).  Units: the counters are KB.
usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json>"""
import collections, csv, glob, json, sys

# launches of one xg_step_fwd (round 2): three multi-job skinny launches -- [POS gate || p || zero], [attention (two
# workgroups per video) || cell 1 || S2'], [cell 2 || h1 copy] -- all instances of skf_kernel (4- and 8-wave variants)
STEP = {"skf_kernel": 3}


def per_kernel(d, counter):
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        for k in STEP:
            if k in r["Kernel_Name"]:
                agg[k].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


def main():
    fd, wd, out = sys.argv[1:4]
    fetch, n1 = per_kernel(fd, "FETCH_SIZE")
    write, n2 = per_kernel(wd, "WRITE_SIZE")
    rows, tot_f, tot_w = {}, 0.0, 0.0
    for k, mult in STEP.items():
        f = fetch.get(k, 0.0) * 1024 * 2.0 * mult
        w = write.get(k, 0.0) * 1024 * mult
        rows[k] = dict(launches_per_step=mult, fetch_bytes_corrected=round(f), write_bytes=round(w), dispatches_seen=n1.get(k, 0))
        tot_f += f; tot_w += w
    res = dict(source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE over tools/step_group_run.py (config 2, B=128)",
               correction="FETCH_SIZE x2 (gfx950 wide-load undercount), WRITE_SIZE as reported",
               per_kernel=rows, fetch_bytes_per_step=round(tot_f), write_bytes_per_step=round(tot_w),
               hbm_bytes_per_step=round(tot_f + tot_w))
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
