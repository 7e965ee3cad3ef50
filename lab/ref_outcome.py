"""Outcome-level comparison with the reference's own fit: the trained MPS the reference serialised (ECG200 train set,
default MPSOptions) against fits of this engine on the same training data with the same options from its own random
starts (Julia's RNG stream cannot be reproduced): training KLD / MSE / accuracy after nsweeps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mpstime_jl_amd as mt

z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "ref_ecg200_trained_mps.npz"))
T = z["pstates"].shape[1]
Wref = [z[f"W_{j}"] for j in range(T)]
cd = z["class_distribution"]
y = np.repeat(np.arange(len(cd)), cd)
X = z["original_data"]
eng = mt.SweepEngine(0)
eng.set_options(chi_max=25, eta=0.01, cutoff=1e-10)
eng.set_dataset(0, z["pstates"], y.astype(np.int32), 2)
eng.set_mps(Wref)
print("reference MPS on its training data: MSE %.5f KLD %.5f acc %.4f" % eng.eval(0)[:3], "chi", [w.shape[2] for w in Wref][:12], "...")
eng.close()
for nsw in (5, 10):
    for seed in (1234, 1, 2, 3):
        opts = mt.MPSOptions(verbosity=-1, nsweeps=nsw, init_rng=seed)
        tr, info, _ = mt.fitMPS(X, y, None, None, opts)
        print("nsweeps", nsw, "seed", seed, "train KLD", [round(v, 3) for v in info["train_KL_div"][-3:]], "acc", info["train_acc"][-1], flush=True)
