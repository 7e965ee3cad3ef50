"""Per-kernel device time of the large-bond path (k_yhat_gen, k_grad, reduce, eigensolver) at full bond dimension."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import mpstime_jl_amd as mt
from oracle import ref_numpy as R
N = int(os.environ.get("PN", 8192)); T = 10; d = 8; chi = 64
rng = np.random.default_rng(0)
X = rng.uniform(-0.9, 0.9, (N, T))
phi = R.legendre_encode(X, d)
lab = (np.arange(N) % 2).astype(np.int32)
W = R.random_mps(T, d, chi, 2, rng)
eng = mt.SweepEngine(0)
eng.set_options(chi_max=chi, eta=0.01)
eng.set_dataset(0, phi, np.sort(lab), 2)
eng.set_mps(W); eng.build_caches()
eng.sweep()
eng.set_profile(0x7FF)
eng.sweep()
pr = eng.get_profile()
print("N", N, {k: round(v[0] / max(v[1], 1), 1) for k, v in pr.items() if v[1]})
