"""Throughput of the device imputation engine on a trained-shape random MPS (bring-up / DESIGN numbers)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mpstime_jl_amd as mt
from oracle import ref_numpy as R

for (N, T, d, chi, frac) in [(1024, 100, 4, 32, 0.5), (4096, 100, 4, 32, 0.5), (1024, 200, 8, 64, 0.5)]:
    rng = np.random.default_rng(0)
    W = R.random_mps(T, d, chi, 1, rng)
    xs = -1.0 + 1e-4 * np.arange(20001)
    grid_phi = R.legendre_encode(xs, d)
    X = rng.uniform(-0.95, 0.95, (N, T))
    phi = R.legendre_encode(X, d)
    m = np.zeros((N, T), dtype=np.uint8)
    nm = int(round(T * frac))
    for i in range(N):
        s = rng.integers(0, T - nm + 1)
        m[i, s:s + nm] = 1
    eng = mt.SweepEngine(0)
    eng.set_options(chi_max=chi)
    eng.set_dataset(1, phi, np.zeros(N, dtype=np.int32), 1)
    eng.set_mps(W)
    eng.impute(1, m[:64], xs, grid_phi, 0, True) if False else None
    t0 = time.perf_counter()
    x, e, secs = eng.impute(1, m, xs, grid_phi, 0, True)
    wall = time.perf_counter() - t0
    sites = int(m.sum())
    print(f"N={N} T={T} d={d} chi={chi}: {sites} missing sites, device {secs * 1e3:.1f} ms, wall {wall * 1e3:.1f} ms, "
          f"{sites / secs / 1e6:.2f} M site-imputations/s, {2.0 * sites * 20001 * (d * d + d) / secs / 1e12:.2f} TFLOP/s on the density grid")
    eng.close()
