#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs into per-kernel, per-launch HBM traffic.

usage: aggregate_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch.  Per /opt/skills/guides/MI355X_MICROARCH.md the
gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so wide coalesced reads are doubled before they are
compared with a byte count (an upper bound for narrow accesses); WRITE_SIZE is exact for 16-B stores.
"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        name = r["Kernel_Name"].split("(")[0]
        acc[name][0] += float(r["Counter_Value"])
        acc[name][1] += 1
    return acc


def main():
    f, w, out = sys.argv[1:4]
    fe, wr = per_kernel(f, "FETCH_SIZE"), per_kernel(w, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fe) | set(wr)):
        if not k.startswith("mpst::"):
            continue
        fk = fe[k][0] / max(fe[k][1], 1)
        wk = wr[k][0] / max(wr[k][1], 1)
        res[k] = {"launches": max(fe[k][1], wr[k][1]), "FETCH_SIZE_KiB": round(fk, 2), "WRITE_SIZE_KiB": round(wk, 2),
                  "hbm_bytes_per_launch_corrected": int(2 * fk * 1024 + wk * 1024)}
    json.dump({"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `python3 bench.py --steps 1 "
                       "--warmup 1 --no-cpu-baseline`; per-launch averages; hbm_bytes = 2*FETCH_SIZE + WRITE_SIZE (gfx950 "
                       "correction of the MI355X guide).", "kernels": res}, open(out, "w"), indent=1)
    for k, v in res.items():
        print(f"{k:28s} {v['launches']:5d} launches  {v['hbm_bytes_per_launch_corrected'] / 1e6:8.3f} MB/launch")


if __name__ == "__main__":
    main()
