#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs (one pass per counter group, as gpurun requires) into per-kernel,
per-launch figures.

usage: aggregate_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <sq_counter_collection.csv|-> <out.json> [note]

HBM traffic: FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch.  Per /opt/skills/guides/MI355X_MICROARCH.md the
gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so wide coalesced reads are doubled before they are compared with a
byte count (an upper bound for narrow accesses); WRITE_SIZE is exact for 16-B stores.

MFMA: SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles in which a SIMD's matrix pipe is busy, summed over the chip
(v_mfma_f64_16x16x4_f64 = 64 cycles: 131072 of them in one k_bond_fused launch at N=4096 give 8.39e6, as counted);
SQ_INSTS_VALU_MFMA_MOPS_F64 * 512 = fp64 MFMA flops.  mfma_util = busy cycles / (1024 SIMDs x kernel duration x 2.4 GHz):
the fraction of the chip's matrix-pipe time the launch used, priced at the maximum clock (GRBM_GUI_ACTIVE is also
recorded, but its window is wider than the kernel on short dispatches).
"""
import collections
import csv
import json
import sys

SIMDS, CLOCK_HZ = 1024, 2.4e9


def per_kernel(path, counters):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    dur = collections.defaultdict(lambda: [0.0, 0])
    seen = set()
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").split("<")[0]
        if r.get("Counter_Name") in counters:
            a = acc[name][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
        key = (name, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            dur[name][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            dur[name][1] += 1
    return acc, dur


def main():
    f, w, q, out = sys.argv[1:5]
    note = sys.argv[5] if len(sys.argv) > 5 else ""
    fe, _ = per_kernel(f, {"FETCH_SIZE"})
    wr, _ = per_kernel(w, {"WRITE_SIZE"})
    sq, dur = per_kernel(q, {"SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F64", "GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"}) if q != "-" else ({}, {})
    res = {}
    for k in sorted(set(fe) | set(wr) | set(sq)):
        if not k.startswith("mpst::"):
            continue
        avg = lambda d, c: d[k][c][0] / max(d[k][c][1], 1) if k in d and c in d[k] else None
        fk, wk = avg(fe, "FETCH_SIZE") or 0.0, avg(wr, "WRITE_SIZE") or 0.0
        e = {"launches": max(fe[k]["FETCH_SIZE"][1] if k in fe else 0, wr[k]["WRITE_SIZE"][1] if k in wr else 0),
             "FETCH_SIZE_KiB": round(fk, 2), "WRITE_SIZE_KiB": round(wk, 2), "hbm_bytes_per_launch_corrected": int(2 * fk * 1024 + wk * 1024)}
        if k in sq:
            d_ns = dur[k][0] / max(dur[k][1], 1)
            busy = avg(sq, "SQ_VALU_MFMA_BUSY_CYCLES") or 0.0
            e.update({"avg_duration_us_pmc_pass": round(d_ns / 1e3, 2), "mfma_busy_cycles_per_launch": round(busy, 1),
                      "mfma_f64_flops_per_launch": round((avg(sq, "SQ_INSTS_VALU_MFMA_MOPS_F64") or 0.0) * 512, 1),
                      "mfma_util": round(busy / (SIMDS * d_ns * 1e-9 * CLOCK_HZ), 5),
                      "GRBM_GUI_ACTIVE_per_launch": round(avg(sq, "GRBM_GUI_ACTIVE") or 0.0, 1),
                      "SQ_WAVE_CYCLES_per_launch": round(avg(sq, "SQ_WAVE_CYCLES") or 0.0, 1)})
        res[k] = e
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import csrc_sha256                 # digest of the kernel sources the passes were run on
    json.dump({"csrc_sha256": csrc_sha256(), "note": "rocprofv3 --kernel-trace --pmc, three separate passes (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES "
                       "SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE) of the command in brackets (profiles/collect_profiles.sh)" + note + "; "
                       "per-launch averages; hbm_bytes = 2*FETCH_SIZE + WRITE_SIZE (gfx950 correction of the MI355X guide); "
                       "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel duration x 2.4 GHz).", "kernels": res}, open(out, "w"), indent=1)
    for k, v in res.items():
        print(f"{k:24s} {v['launches']:5d} launches  {v['hbm_bytes_per_launch_corrected'] / 1e6:8.3f} MB/launch  mfma_util {v.get('mfma_util')}")


if __name__ == "__main__":
    main()
