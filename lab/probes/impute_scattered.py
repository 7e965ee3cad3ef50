"""Probe: imputation with SCATTERED missing points (20 % of every instance) instead of a block - known sites between missing ones
carry a full environment matrix, so the known-site step of k_imp_right dominates.  usage: impute_scattered.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import mpstime_jl_amd as mt
import bench
for cx, d, chi, N, T in ((True, 8, 64, 2048, 200), (False, 12, 40, 2048, 100)):
    rng = np.random.default_rng(5)
    W = bench.random_chain(T, d, chi, np.random.default_rng(7))
    if not cx:
        W = [np.ascontiguousarray(w.real) for w in W]
    enc = mt.model_encoding("Fourier" if cx else "Legendre")
    xs = -1.0 + 1e-4 * np.arange(20001)
    X = rng.uniform(-0.95, 0.95, (N, T))
    phi = np.ascontiguousarray(enc.encode(X, d), dtype=np.complex128 if cx else np.float64)
    gphi = np.ascontiguousarray(enc.encode(xs, d), dtype=phi.dtype)
    m = (rng.uniform(size=(N, T)) < 0.2).astype(np.uint8)
    lab = np.zeros(N, dtype=np.int32)
    eng = mt.SweepEngine(0)
    eng.impute_model(W, phi[:8], lab[:8], m[:8], xs, gphi, 0, True, compute="f32")
    x, e, secs = eng.impute_model(W, phi, lab, m, xs, gphi, 0, True, compute="f32")
    pr, pl = eng.impute_phases()
    print(("fourier" if cx else "legendre"), d, chi, N, T, f"20 % scattered: {secs * 1e3:.1f} ms (right {pr * 1e3:.1f}, left {pl * 1e3:.1f}), {int(m.sum()) / secs / 1e6:.2f} M site-imputations/s", flush=True)
    eng.close()
