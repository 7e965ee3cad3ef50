"""How far do the graph path and the per-bond path drift apart in one sweep (gauge-invariant overlaps), for several starts?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import mpstime_jl_amd as mt
from oracle import ref_numpy as R
from tests.helpers import bond_of
N, T, d, chi = 4096, 100, 4, 32
full = bench.make_inputs(N, T, d)
sub = slice(0, N, 37)
for seed in (1234, 1, 2, 3):
    W0 = mt.generate_startingMPS(4, T, d, 2, seed)
    eng = mt.SweepEngine(0)
    eng.set_options(chi_max=chi, eta=0.01)
    eng.set_dataset(0, full.phi, full.label_index, 2)
    eng.set_mps(W0)
    eng.build_caches()
    for _ in range(3):
        eng.sweep()
    Ws = eng.get_mps()
    out = {}
    for path in ("sweep", "steps", "steps2"):
        eng.set_mps(Ws)
        eng.build_caches()
        if path == "sweep":
            eng.sweep()
        else:
            for q in range(2 * (T - 1)):
                eng.bond_step(*bond_of(q, T))
        out[path] = R.contract_mps(eng.get_mps(), full.phi[sub])
    dev = np.abs(out["sweep"] - out["steps"]).max() / np.abs(out["steps"]).max()
    dev2 = np.abs(out["steps2"] - out["steps"]).max() / np.abs(out["steps"]).max()
    print(f"seed {seed}: graph vs steps {dev:.2e}; steps vs steps again {dev2:.2e}", flush=True)
    eng.close()
