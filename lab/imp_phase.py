import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mpstime_jl_amd as mt
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
for (N, T, d, chi, cx, compute) in [(1024, 200, 8, 64, False, "f64"), (4096, 100, 4, 32, False, "f64"), (1024, 200, 8, 64, True, "f32")]:
    rng = np.random.default_rng(1)
    xs = -1.0 + 1e-4 * np.arange(20001)
    m = np.zeros((N, T), dtype=np.uint8)
    for i in range(N):
        s0 = rng.integers(0, T - T // 2 + 1)
        m[i, s0:s0 + T // 2] = 1
    m[0, :] = 0; m[0, T // 4: T // 4 + T // 2] = 1
    X = rng.uniform(-0.95, 0.95, (N, T))
    W = bench.random_chain(T, d, chi, np.random.default_rng(7))
    if not cx:
        W = [np.ascontiguousarray(w.real) for w in W]
    enc = mt.model_encoding("Fourier" if cx else "Legendre")
    eng = mt.SweepEngine(0)
    x, e, secs = eng.impute_model(W, enc.encode(X, d), np.zeros(N, dtype=np.int32), m, xs, enc.encode(xs, d), 0, True, compute=compute)
    ph = eng.impute_phases()
    t = e[0, :8] * 0.01      # us (100 MHz)
    names = ["(hdr)", "U+rho", "density", "scan", "select+wmad", "project", "LW(+known sites, rescale)", "-"]
    print(f"N={N} T={T} chi={chi} d={d} cx={cx} {compute}: total {secs*1e3:.1f} ms (right {ph[0]*1e3:.1f}, left {ph[1]*1e3:.1f}); instance 0, us per missing site:",
          {n: round(v / (T // 2), 1) for n, v in zip(names, t)}, "sum per instance (ms)", round(t.sum() / 1e3, 2))
    eng.close()
