"""The fit the reference's docs log (docs/src/hyperparameters.md:65-74): noisy trendy sine, 240 training series (a 5-fold
CV split of 300), T=100, chi_max=37, d=8, eta=0.0234, default nsweeps: 'finished in 128.25s (train=126.38s)' there."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mpstime_jl_amd as mt
rng = np.random.default_rng(7)
X1, _ = mt.trendy_sine(100, 150, period=(12.0, 15.0), slope=[-3.0, 0.0, 3.0], sigma=0.1, rng=rng)
X2, _ = mt.trendy_sine(100, 150, period=(16.0, 19.0), slope=[-3.0, 0.0, 3.0], sigma=0.1, rng=rng)
X = np.concatenate([X1, X2]); y = np.concatenate([np.zeros(150, dtype=np.int64), np.ones(150, dtype=np.int64)])
perm = rng.permutation(300)
tr, te = perm[:240], perm[240:]
opts = mt.MPSOptions(verbosity=-1, chi_max=37, d=8, eta=0.023357214690901226)
for rep in range(2):
    t0 = time.perf_counter()
    W, info, _ = mt.fitMPS(X[tr], y[tr], X[te], y[te], opts)
    dt = time.perf_counter() - t0
    print(f"fit {rep}: {dt:.2f} s wall for nsweeps={opts.nsweeps}; sweep seconds {np.round(info['time_taken'], 3).tolist() if 'time_taken' in info else ''}; "
          f"train acc {info['train_acc'][-1]:.3f} test acc {info['test_acc'][-1]:.3f}", flush=True)
