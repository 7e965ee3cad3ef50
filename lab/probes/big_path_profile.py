"""Two sweeps of the large-bond path (chi=64, d=8: 512 x 512 bond tensors) for a rocprofv3 kernel trace.
T is short (most bonds are at full dimension from the third site on); the full-size figures are bench.py's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import mpstime_jl_amd as mt
from oracle import ref_numpy as R
N, T, d, chi = 8192, 24, 8, 64
rng = np.random.default_rng(0)
X = rng.uniform(-0.9, 0.9, (N, T))
phi = R.legendre_encode(X, d)
lab = np.sort((np.arange(N) % 2).astype(np.int32))
W = R.random_mps(T, d, chi, 2, rng)
eng = mt.SweepEngine(0)
eng.set_options(chi_max=chi, eta=0.01)
eng.set_dataset(0, phi, lab, 2)
eng.set_mps(W); eng.build_caches()
for _ in range(3):
    eng.sweep()
print("chi", eng.get_chi().tolist(), eng.info())
