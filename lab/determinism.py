"""Run the same sweeps several times from the same start and demand identical bits (every reduction in the engine has a
fixed order, so any difference is a race).  Configs: headline (fused chain, hipGraph), N=8192 (256 partitions), the
plain-stream profiling path, the blocked eigensolver (d*chi > 128)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mpstime_jl_amd as mt
import bench


F = int(sys.argv[1]) if len(sys.argv) > 1 else 1


def run(name, phi, labels, C, chi, d, nsweeps, reps, prof=0, **opt):
    reps *= F
    T = phi.shape[1]
    W0 = mt.generate_startingMPS(4, T, d, C, 1234)
    eng = mt.SweepEngine(0)
    eng.set_options(chi_max=chi, eta=0.01, cutoff=1e-10, **opt)
    eng.set_dataset(0, phi, labels, C)
    ref = None
    bad = 0
    t0 = time.time()
    for r in range(reps):
        eng.set_mps(W0)
        eng.build_caches()
        if prof:
            eng.set_profile(prof)
        for s in range(nsweeps):
            eng.sweep()
        W = eng.get_mps()
        if ref is None:
            ref = W
        elif not all(np.array_equal(a, b) for a, b in zip(ref, W)):
            bad += 1
            print("  rep", r, "differs: max", max(float(np.abs(a - b).max()) if a.shape == b.shape else -1.0 for a, b in zip(ref, W)), flush=True)
    print(f"{name}: {reps} x {nsweeps} sweeps, differing repetitions: {bad}, info {eng.info()}, {time.time() - t0:.1f} s", flush=True)
    eng.close()
    return bad


bad = 0
full = bench.make_inputs(4096, 100, 4)
bad += run("headline chi32 (graph)", full.phi, full.label_index, 2, 32, 4, 3, 12)
bad += run("headline chi32 update_iters=2 MSE", full.phi, full.label_index, 2, 32, 4, 2, 6, update_iters=2, loss="MSE")
bad += run("headline chi32 (plain stream, profiling)", full.phi, full.label_index, 2, 32, 4, 2, 6, prof=0x7FF)
full8 = bench.make_inputs(8192, 100, 4)
bad += run("N=8192 chi32 (sliced pair, 512-series shares)", full8.phi, full8.label_index, 2, 32, 4, 2, 8)
full16 = bench.make_inputs(16384, 60, 4)
bad += run("N=16384 chi32 (persistent k_bond_fused pair)", full16.phi, full16.label_index, 2, 32, 4, 2, 4)
full2 = bench.make_inputs(240, 60, 8)
bad += run("d8 chi37 (blocked eigensolver)", full2.phi, full2.label_index, 2, 37, 8, 1, 8)
full3 = bench.make_inputs(512, 40, 8)
bad += run("d8 chi64 (blocked eigensolver, n=512)", full3.phi, full3.label_index, 2, 64, 8, 1, 4)
print("TOTAL differing repetitions:", bad)
