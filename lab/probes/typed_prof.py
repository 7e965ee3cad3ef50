"""Probe (not a test): per-kernel-class device time of the element-typed sweep at a given shape.
usage: python lab/probes/typed_prof.py N T chi d dtype [C] [sweeps]"""
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mpstime_jl_amd as mt
from oracle import ref_numpy as R

DT = {"f64": np.float64, "f32": np.float32, "c128": np.complex128, "c64": np.complex64}


def main():
    N, T, chi, d = map(int, sys.argv[1:5])
    dt = DT[sys.argv[5]]
    C = int(sys.argv[6]) if len(sys.argv) > 6 else 1
    nsw = int(sys.argv[7]) if len(sys.argv) > 7 else 2
    rng = np.random.default_rng(1)
    X = rng.uniform(-1, 1, (N, T))
    cx = np.dtype(dt).kind == "c"
    phi = (R.fourier_encode(X, d) if cx else R.legendre_encode(X, d)).astype(dt)
    lab = np.sort(np.arange(N) % C).astype(np.int32)
    W = mt.generate_startingMPS(4, T, d, C, 1234, dt)
    eng = mt.SweepEngine(0)
    eng.set_options(chi_max=chi, eta=0.01)
    eng.set_dataset(0, phi, lab, C)
    eng.set_mps(W)
    eng.build_caches()
    t0 = time.time()
    for _ in range(2):
        st = eng.sweep()
    print(f"warmup: {time.time() - t0:.2f}s, last sweep {st['seconds'] * 1e3:.1f} ms, max chi {st['max_chi']}", flush=True)
    ts = []
    for _ in range(nsw):
        ts.append(eng.sweep()["seconds"])
    print(f"{sys.argv[5]} N={N} T={T} chi={chi} d={d} C={C}: {np.mean(ts) * 1e3:.2f} ms/sweep = {np.mean(ts) / (2 * (T - 1)) * 1e6:.1f} us/bond, info {eng.info()}", flush=True)
    eng.set_profile(0x7FF)
    eng.sweep()
    prof = eng.get_profile()
    tot = sum(v[0] for v in prof.values())
    for k, (us, cnt) in prof.items():
        if cnt:
            print(f"  {k:22s} {us / 1e3:9.2f} ms  {cnt:6d} launches  {us / cnt:9.1f} us each  {100 * us / tot:5.1f} %")
    eng.close()


if __name__ == "__main__":
    main()
