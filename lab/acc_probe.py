import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
import mpstime_jl_amd as mt
import bench
full = bench.make_inputs(4096, 100, 4)
for seed in (1234, 1, 2):
    W0 = mt.generate_startingMPS(4, 100, 4, 2, seed)
    eng = mt.SweepEngine(0)
    eng.set_options(chi_max=32, eta=0.01)
    eng.set_dataset(0, full.phi, full.label_index, 2)
    eng.set_mps(W0); eng.build_caches()
    out = []
    for s in range(3):
        st = eng.sweep()
        mse, kld, acc, _ = eng.eval(0)
        out.append((round(kld, 3), round(acc, 4), st["eig_fallbacks"]))
    print("seed", seed, out, flush=True)
    eng.close()
