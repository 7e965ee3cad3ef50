import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) + "")
sys.argv = ["x", "--quick"]
import numpy as np
import mpstime_jl_amd as mt
from oracle import ref_numpy as R
exec(open(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) + "/lab/probes/impute_bench.py").read().split("res = []")[0].split("ap = argparse")[0])
exec("def random_chain" + open(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) + "/lab/probes/impute_bench.py").read().split("def random_chain")[1].split("def deviation")[0])
eng = mt.SweepEngine(0)
for (N, T, d, chi) in [(4096, 100, 4, 32), (1024, 200, 8, 64), (2048, 100, 12, 40)]:
    for cx in (False, True):
        W, xs, gphi, phi, m = problem(N, T, d, chi, cx)
        lab = np.zeros(N, dtype=np.int32)
        for compute in ("f64", "f32"):
            if cx and compute == "f64" and chi > 48: continue
            eng.impute_model(W, phi[:8], lab[:8], m[:8], xs, gphi, 0, True, compute=compute)
            x, e, secs = eng.impute_model(W, phi, lab, m, xs, gphi, 0, True, compute=compute)
            pr, pl = eng.impute_phases()
            print(N, T, d, chi, "cx" if cx else "re", compute, f"total {secs*1e3:.1f} ms  right {pr*1e3:.1f}  left {pl*1e3:.1f}  closed_form {eng.impute_info()['closed_form_densities']}", flush=True)
