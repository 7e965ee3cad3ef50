"""Random shapes through the imputation engine against oracle/impute_numpy.py (not collected by pytest; run by hand on a GPU box:
python tests/fuzz_impute.py [cases] [seed]).  Exercises the batched sweep (k_imp_leftb), the vector recursion of the environment
pass and their fall-backs on shapes no fixed test has: chi = 1, d = 2 ... 16, partial workgroups, every missing pattern."""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import mpstime_jl_amd as mt                                     # noqa: E402
from tests.test_gpu_impute_model import _check, _complex_mps    # noqa: E402
from oracle import ref_numpy as R                               # noqa: E402


def main(cases=40, seed=0):
    rng = np.random.default_rng(seed)
    eng = mt.SweepEngine(0)
    bad = 0
    try:
        for case in range(cases):
            cx = bool(rng.integers(0, 2))
            N = int(rng.integers(1, 50))
            T = int(rng.integers(2, 18))
            d = int(rng.choice([2, 3, 4, 5, 8, 12, 16]))
            chi = int(rng.choice([1, 2, 3, 7, 16, 17, 33, 40]))
            C = int(rng.integers(1, 4))
            order = int(rng.integers(0, 2))
            compute = "f64"
            ngrid = int(rng.choice([201, 1201]))
            W = _complex_mps(T, d, chi, C, rng) if cx else R.random_mps(T, d, chi, C, rng)
            xs = -1.0 + (2.0 / (ngrid - 1)) * np.arange(ngrid)
            enc = (lambda x: R.fourier_encode(x, d)) if cx else (lambda x: R.legendre_encode(x, d))
            X = rng.uniform(-0.95, 0.95, (N, T))
            y = rng.integers(0, C, N).astype(np.int32)
            p = float(rng.choice([0.1, 0.4, 0.8]))
            m = (rng.uniform(size=(N, T)) < p).astype(np.uint8)
            kind = int(rng.integers(0, 4))
            if kind == 1:                     # a block
                m[:] = 0
                a = int(rng.integers(0, T))
                m[:, a:min(T, a + max(1, T // 2))] = 1
            elif kind == 2:
                m[0] = 1
            grid_phi, phi = enc(xs), enc(X)
            tag = f"case {case}: cx={cx} N={N} T={T} d={d} chi={chi} C={C} order={order} kind={kind} p={p} ngrid={ngrid}"
            try:
                x_med, e_med, _ = eng.impute_model(W, phi, y, m, xs, grid_phi, 0, True, order=order, compute=compute)
                info = eng.impute_info()
                _check(W, xs, grid_phi, phi, y, m, x_med, e_med, "median", ["forwards", "backwards"][order], max_flips=max(2, N // 8))
                print("ok  ", tag, info)
            except AssertionError as e:
                bad += 1
                print("FAIL", tag, str(e)[:300])
    finally:
        eng.close()
    print("failures:", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
