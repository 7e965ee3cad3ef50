import sys
import numpy as np
sys.path.insert(0, ".")
import mpstime_jl_amd as mt
from oracle import ref_complex as RC
for dt in (np.complex128, np.complex64, np.float32):
    for T in (20, 100, 200):
        ds, W = RC.make_problem(64, T, 4, 4, 2, seed=1, dtype=dt)
        eng = mt.SweepEngine(0)
        eng.set_options(chi_max=8, eta=0.05)
        eng.set_dataset(0, ds.phi, ds.label_index, 2)
        eng.set_mps(W)
        eng.build_caches()
        wide = np.complex128 if np.dtype(dt).kind == "c" else np.float64
        ds64, W64 = RC.cast_problem(ds, W, wide)
        ko = RC.mse_loss_acc(W64, ds64)[1]
        k0 = eng.eval(0)[1]
        tr = eng.bond_step(T - 2, True)
        LE, RE = RC.construct_caches(W64, ds64.phi, True)
        to = {}
        RC.bond_step(W64, LE, RE, T - 2, ds64, RC.SweepOptions(chi_max=8, eta=0.05), True, to)
        print(np.dtype(dt).name, T, "kld", k0, ko, "| bond loss", tr["loss"], to["loss"], "grad", tr["grad_norm"], to["grad_norm"], flush=True)
        eng.close()
