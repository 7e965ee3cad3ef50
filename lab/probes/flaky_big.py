"""Repeat the dchi168 sweep of tests/test_gpu_bigbond.py and count the runs whose result differs from the oracle's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import mpstime_jl_amd as mt
from oracle import ref_numpy as R
from tests.helpers import load_engine, make_problem

N, T, d, chi0, chimax, C, loss, bbopt = (40, 3, 12, 10, 14, 1, "KLD", "TSGO")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ds, W0 = make_problem(N, T, d, chi0, C, seed=N + d)
opts = R.SweepOptions(nsweeps=1, chi_max=chimax, eta=0.05, loss_grad=loss, bbopt=bbopt)
Wo = [t.copy() for t in W0]
R.sweep(Wo, ds, opts)
yo = R.contract_mps(Wo, ds.phi)
eng = mt.SweepEngine(0)
bad = 0
for r in range(reps):
    load_engine(eng, ds, W0, opts)
    eng.build_caches()
    st = eng.sweep()
    yg = R.contract_mps(eng.get_mps(), ds.phi)
    err = np.abs(yg - yo).max() / np.abs(yo).max()
    if err > 1e-8:
        bad += 1
        print("rep", r, "err", err, "info", eng.info(), "chi", eng.get_chi()[0], flush=True)
print("bad", bad, "of", reps, "env", {k: v for k, v in os.environ.items() if k.startswith("MPST_")})
eng.close()
