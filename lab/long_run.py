"""Soak: many sweeps on config 3 / config 2 / the ECG200 fixture; report fallbacks, loss, accuracy (scratch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mpstime_jl_amd as mt
import bench

def run(name, phi, labels, C, chi, d, nsweeps, seed=1234):
    T = phi.shape[1]
    eng = mt.SweepEngine(0)
    eng.set_options(chi_max=chi, eta=0.01, cutoff=1e-10)
    eng.set_dataset(0, phi, labels, C)
    eng.set_mps(mt.generate_startingMPS(4, T, d, C, seed)); eng.build_caches()
    fb = 0; t0 = time.time(); hist = []
    for s in range(nsweeps):
        st = eng.sweep(); fb += st["eig_fallbacks"]
        if s % 5 == 4 or s == nsweeps - 1:
            mse, kld, acc, _ = eng.eval(0); hist.append((s + 1, round(kld, 4), round(acc, 4)))
    eng.normalize()
    W = eng.get_mps()
    finite = all(np.isfinite(t).all() for t in W)
    print(name, "sweeps", nsweeps, "fallbacks", fb, "finite", finite, "chi_max", int(eng.get_chi()[0].max()), "hist", hist, "%.1fs" % (time.time() - t0))
    eng.close()

full = bench.make_inputs(4096, 100, 4)
run("config3 chi32", full.phi, full.label_index, 2, 32, 4, 30)
run("config2 chi16", full.phi, full.label_index, 2, 16, 4, 20)
z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "ref_ecg200_trained_mps.npz"))
lab = np.repeat(np.arange(2), z["class_distribution"])
run("ecg200 d5 chi25", z["pstates"], lab, 2, 25, 5, 20, seed=7)
full2 = bench.make_inputs(240, 60, 8)
run("d8 chi37 (blocked eigensolver)", full2.phi, full2.label_index, 2, 37, 8, 6)
print("library fallbacks of the blocked solver are reported by eng.info() inside run() if needed")
