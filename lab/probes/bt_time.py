"""Device time of the blocked eigensolver inside a sweep-like loop (bond_step at full bond dimension)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import mpstime_jl_amd as mt
from oracle import ref_numpy as R
N, T, d, chi = 2048, 12, 8, 64
rng = np.random.default_rng(0)
X = rng.uniform(-0.9, 0.9, (N, T))
phi = R.legendre_encode(X, d)
W = R.random_mps(T, d, chi, 1, rng)
eng = mt.SweepEngine(0)
eng.set_options(chi_max=chi, eta=0.01)
eng.set_dataset(0, phi, np.zeros(N, dtype=np.int32), 1)
eng.set_mps(W); eng.build_caches()
eng.sweep()
eng.set_profile(0x7FF)
eng.sweep()
pr = eng.get_profile()
print("BT_G", os.environ.get("MPST_BT_G"), {k: round(v[0] / max(v[1], 1), 1) for k, v in pr.items() if v[1]}, eng.info())
