"""Large-bond eigensolver with and without the subspace solver in front: per-bond agreement with the C restatement and time.
usage: python lab/probes/subspace_probe.py [N T chi d]   (env MPST_NO_SUBSPACE=1 switches it off)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import mpstime_jl_amd as mt
from oracle import ref_numpy as R
from tests.helpers import make_problem, bond_of

N, T, chi, d = [int(x) for x in sys.argv[1:5]] if len(sys.argv) > 4 else (512, 10, 32, 8)
ds, W = make_problem(N, T, d, 4, 2, seed=3)
opts = R.SweepOptions(chi_max=chi, eta=0.05)
eng = mt.SweepEngine(0)
eng.set_options(chi_max=chi, eta=0.05)
eng.set_dataset(0, ds.phi, ds.label_index, 2)
eng.set_mps(W)
eng.build_caches()
Wo = [t.copy() for t in W]
LE, RE = R.construct_caches(Wo, ds.phi, True)
worst = dict(S=0.0, bond=0.0)
for q in range(2 * (T - 1)):
    lid, gl = bond_of(q, T)
    if q == T - 1:
        LE, RE = R.construct_caches(Wo, ds.phi, False)
    eng.set_mps(Wo)
    eng.build_caches()
    tr = {}
    R.bond_step(Wo, LE, RE, lid, ds, opts, gl, tr)
    got = eng.bond_step(lid, gl)
    nk = min(got["chi"], tr["chi"])
    eS = np.abs(got["S"][:nk] - tr["S"][:nk]).max() / tr["S"][0]
    Wg = eng.get_mps()
    bt_g, sh = R.flatten_bt(Wg[lid], Wg[lid + 1]); bt_o, _ = R.flatten_bt(Wo[lid], Wo[lid + 1])
    eB = np.abs(bt_g - bt_o).max() / np.abs(bt_o).max() if got["chi"] == tr["chi"] else float("inf")
    worst["S"] = max(worst["S"], eS); worst["bond"] = max(worst["bond"], eB)
    info = eng.info()
    print(q, lid, int(gl), "chi", got["chi"], tr["chi"], "errS %.2e errB %.2e" % (eS, eB), "ss", info["subspace_attempted"], info["subspace_accepted"], "lib", info["library_eig_fallbacks"], flush=True)
print("worst", worst)
# free-running timing
eng.set_mps(W); eng.build_caches()
for _ in range(2):
    eng.sweep()
t0 = time.perf_counter()
for _ in range(3):
    st = eng.sweep()
dt = (time.perf_counter() - t0) / 3
print("ms per sweep", 1e3 * dt, "per bond us", 1e6 * dt / (2 * (T - 1)), eng.info())
eng.set_profile(0x7FF); eng.sweep(); print(eng.get_profile()); eng.close()
