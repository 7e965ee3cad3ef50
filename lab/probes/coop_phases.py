"""Per-phase time of the persistent tridiagonalisation (library built with EXTRA=-DMPST_COOP_PROF): sums over the
n - 1 steps of one full-size solve as seen by thread 0 of workgroup 0."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import mpstime_jl_amd as mt
from oracle import ref_numpy as R
N, T, d, chi = 2048, 10, 8, 64
rng = np.random.default_rng(0)
X = rng.uniform(-0.9, 0.9, (N, T))
phi = R.legendre_encode(X, d)
W = R.random_mps(T, d, chi, 1, rng)
eng = mt.SweepEngine(0)
eng.set_options(chi_max=chi, eta=0.01)
eng.set_dataset(0, phi, np.zeros(N, dtype=np.int32), 1)
eng.set_mps(W); eng.build_caches()
eng.sweep()
for lid in (8, 7, 6, 5, 4, 3):          # leftwards from the right end: bonds (6,7) ... are at full size (512 x 512)
    eng.bond_step(lid, True)
    us = list(eng.eig_phases().values())
    names = ["wait(poll)", "loads", "(a) alpha", "(b) reflector", "(c) rows"]
    print("lid", lid, {k: round(float(u), 1) for k, u in zip(names, us[:5])}, "arrive", round(0.01 * float(us[5]), 1), "total", round(float(sum(us[:5])) + 0.01 * float(us[5]), 1))
