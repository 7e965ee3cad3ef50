"""Training KLD / accuracy per sweep at the headline shape for several starting seeds: how wide is the plateau band?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import mpstime_jl_amd as mt
full = bench.make_inputs(4096, 100, 4)
for seed in (1234, 1, 2):
    W0 = mt.generate_startingMPS(4, 100, 4, 2, seed)
    eng = mt.SweepEngine(0)
    eng.set_options(chi_max=32, eta=0.01, cutoff=1e-10)
    eng.set_dataset(0, full.phi, full.label_index, 2)
    eng.set_mps(W0)
    eng.build_caches()
    row = []
    for s in range(8):
        eng.sweep()
        _, kld, acc, _ = eng.eval(0)
        row.append(f"{kld:.2f}/{acc:.4f}")
    print(f"seed {seed}:", " ".join(row), flush=True)
    eng.close()
