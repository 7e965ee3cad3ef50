import sys
import numpy as np
sys.path.insert(0, ".")
import mpstime_jl_amd as mt
import bench
from oracle import ref_complex as RC
for (N, T, d, chi, C) in ((64, 200, 8, 64, 1), (64, 50, 8, 64, 1), (64, 200, 4, 64, 1), (64, 200, 8, 16, 1), (64, 200, 8, 64, 2), (8192, 20, 8, 64, 1)):
    dt = np.dtype(np.complex64)
    full = bench.typed_inputs(N, T, d, C, True)
    W0 = mt.generate_startingMPS(4, T, d, C, 1234, dt)
    eng = mt.SweepEngine(0)
    eng.set_options(chi_max=chi, eta=0.01)
    eng.set_dataset(0, full.phi, full.label_index, C, dtype=dt)
    eng.set_mps(W0)
    eng.build_caches()
    ds = RC.EncodedSet(full.phi.astype(np.complex64).astype(np.complex128), full.label_index, np.bincount(full.label_index).astype(np.int64))
    ko = RC.mse_loss_acc([t.astype(np.complex128) for t in W0], ds)[1]
    print((N, T, d, chi, C), "kld", eng.eval(0)[1], ko, flush=True)
    eng.close()
