import sys; sys.path.insert(0,'.')
import numpy as np
import mpstime_jl_amd as mt
from oracle.c_oracle import COracle
import bench
full = bench.make_inputs(4096, 100, 4); W0 = mt.generate_startingMPS(4, 100, 4, 2, 1234)
co = COracle(W0, full.phi, full.label_index, full.class_distribution, 32, eta=0.01, rebuild_caches=False)
co.build_caches(); ref = co.sweep(record=True, max_bonds=12)["bonds_rec"]
for alg in (0,1):
    eng = mt.SweepEngine(0); eng.set_options(chi_max=32, eta=0.01, svd_alg=alg); eng.set_dataset(0, full.phi, full.label_index, 2); eng.set_mps(W0); eng.build_caches()
    for k,lid in enumerate(range(98, 86, -1)):
        tr = eng.bond_step(lid, True)
        So = ref[k]["S"]; Sg = tr["S"]
        print(alg, k, lid, tr["chi"], ref[k]["chi"], "oracle P tail:", (So[-3:]**2), "engine P tail:", (Sg[-3:]**2), "n_spec", len(Sg), len(So))
    eng.close()
