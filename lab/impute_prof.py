"""One imputation pass per element type at configs[4]'s bond shape, for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mpstime_jl_amd as mt
sys.argv = [sys.argv[0]]
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
N, T, d, chi = 2048, 200, 8, 64
rng = np.random.default_rng(1)
xs = -1.0 + 1e-4 * np.arange(20001)
m = np.zeros((N, T), dtype=np.uint8)
for i in range(N):
    s0 = rng.integers(0, T - T // 2 + 1)
    m[i, s0:s0 + T // 2] = 1
lab = np.zeros(N, dtype=np.int32)
X = rng.uniform(-0.95, 0.95, (N, T))
eng = mt.SweepEngine(0)
Wc = bench.random_chain(T, d, chi, np.random.default_rng(7))
for cx in (False, True):
    enc = mt.model_encoding("Fourier" if cx else "Legendre")
    W = Wc if cx else [np.ascontiguousarray(w.real) for w in Wc]
    gphi = enc.encode(xs, d)
    phi = enc.encode(X, d)
    for compute in ("f64", "f32"):
        if cx and compute == "f64":
            continue
        x, e, secs = eng.impute_model(W, phi, lab, m, xs, gphi, 0, True, compute=compute)
        print(cx, compute, secs, flush=True)
eng.close()
