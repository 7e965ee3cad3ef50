"""Random shapes and batch sizes through mpst_sweep_batch against separate sweeps, bit for bit (not collected by pytest; run by hand on a
GPU box: python tests/fuzz_batch.py [cases] [seed]).  Beyond 8 fits the eigensolver takes several eigenpairs per workgroup
(k_eig_trivec_bm); the results must not know."""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import mpstime_jl_amd as mt             # noqa: E402
from tests.helpers import make_problem  # noqa: E402


def main(cases=8, seed=0):
    rng = np.random.default_rng(seed)
    bad = 0
    for case in range(cases):
        d = int(rng.choice([2, 4, 8]))
        chi = int(rng.choice([c for c in (6, 12, 16, 32, 64) if d * c <= 128]))
        K = int(rng.integers(9, 41))
        N = int(rng.choice([128, 512, 1024]))
        T = int(rng.integers(6, 20))
        C = int(rng.integers(2, 4))
        probs = [make_problem(N, T, d, min(4, chi), C, seed=1000 * case + k) for k in range(K)]

        def fresh(k):
            e = mt.SweepEngine(0)
            e.set_batch_hint(K)
            e.set_options(chi_max=chi, eta=[0.05, 0.02, 0.1][k % 3], cutoff=1e-10)
            ds, W = probs[k]
            e.set_dataset(0, ds.phi, ds.label_index, C)
            e.set_mps(W)
            e.build_caches()
            return e

        solo = [fresh(k) for k in range(K)]
        bat = [fresh(k) for k in range(K)]
        ok = True
        try:
            for sweep in range(2):
                for e in solo:
                    e.sweep()
                st = mt.sweep_batch(bat)
                ok &= all(s["eig_fallbacks"] == 0 for s in st)
                for a, b in zip(solo, bat):
                    ok &= all(np.array_equal(ta, tb) for ta, tb in zip(a.get_mps(), b.get_mps()))
        finally:
            for e in solo + bat:
                e.close()
        print("ok  " if ok else "FAIL", f"case {case}: K={K} N={N} T={T} d={d} chi={chi} C={C}")
        bad += 0 if ok else 1
    print("failures:", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
