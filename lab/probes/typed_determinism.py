"""Probe: repeated element-typed fits (large-bond path: XCD-local complex reduction, blocked real solver) give identical bits."""
import sys, hashlib
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import mpstime_jl_amd as mt
from oracle import ref_numpy as R
def run(dt, N, T, chi, d, nsw):
    rng = np.random.default_rng(1)
    X = rng.uniform(-1, 1, (N, T))
    cx = np.dtype(dt).kind == "c"
    phi = (R.fourier_encode(X, d) if cx else R.legendre_encode(X, d)).astype(dt)
    lab = np.zeros(N, dtype=np.int32)
    W = mt.generate_startingMPS(4, T, d, 1, 1234, dt)
    eng = mt.SweepEngine(0)
    eng.set_options(chi_max=chi, eta=0.01)
    eng.set_dataset(0, phi, lab, 1)
    eng.set_mps(W)
    eng.build_caches()
    for _ in range(nsw):
        eng.sweep()
    out = eng.get_mps()
    info = eng.info()
    eng.close()
    h = hashlib.sha256()
    for w in out:
        h.update(np.ascontiguousarray(w).tobytes())
    return h.hexdigest()[:16], info["library_eig_fallbacks"], info["persistent_tridiag_aborts"]
for dt, shape in ((np.complex64, (2048, 24, 64, 8)), (np.complex64, (1024, 16, 32, 8)), (np.float32, (2048, 24, 64, 8)), (np.complex128, (512, 12, 20, 6))):
    hs = [run(dt, *shape, 2) for _ in range(4)]
    print(np.dtype(dt).name, shape, hs, "IDENTICAL" if len(set(h[0] for h in hs)) == 1 else "DIFFER")
