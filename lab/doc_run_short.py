import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mpstime_jl_amd as mt
rng = np.random.default_rng(7)
X1, _ = mt.trendy_sine(100, 120, period=(12.0, 15.0), slope=[-3.0, 0.0, 3.0], sigma=0.1, rng=rng)
X2, _ = mt.trendy_sine(100, 120, period=(16.0, 19.0), slope=[-3.0, 0.0, 3.0], sigma=0.1, rng=rng)
X = np.concatenate([X1, X2]); y = np.concatenate([np.zeros(120, dtype=np.int64), np.ones(120, dtype=np.int64)])
opts = mt.MPSOptions(verbosity=-1, chi_max=37, d=8, eta=0.0234, nsweeps=2)
t0 = time.perf_counter()
W, info, _ = mt.fitMPS(X, y, None, None, opts)
print("wall", time.perf_counter() - t0, info["time_taken"])
