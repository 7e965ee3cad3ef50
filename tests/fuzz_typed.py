"""Random shapes through the element-typed sweep (fp32 / complex64 / complex128, and Float64 through the typed kernels) against
oracle/ref_complex.py, teacher forced (tests/test_gpu_typed.py's comparison and tolerances); not collected by pytest; run by hand on a GPU
box: python tests/fuzz_typed.py [seed] [cases].

Tolerances scale with the bond's conditioning: a series whose overlap is a fraction 1 / kappa of the typical one carries kappa times the
rounding of its chain products into the KLD gradient through 1 / yhat (the reference does not clamp).  tests/test_gpu_typed.py's table is
for kappa of a few hundred; random models reach 1e6 (fp32 then resolves the gradient to 10 %: the first runs of this fuzzer 'failed' there
and nowhere else).  A remaining FAIL with update_iters > 1 is to be read against the same case in Float64 (MPST_TYPED=1): the second
optimiser step of a bond can land next to a vanishing overlap that the start state's kappa does not show - seed 6's (257, 3, 5, 3) does: the
two-site tensor is off by 1e-2 in fp32 and by 5e-12 in fp64, the same 1e5 roundings in both."""
import os
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import mpstime_jl_amd as mt                                                  # noqa: E402
from oracle import ref_complex as RC                                        # noqa: E402
from oracle import ref_numpy as R                                           # noqa: E402
from tests.helpers import bond_of                                           # noqa: E402
from tests.test_gpu_typed import DT, TOL, caches_around, problem, two_site  # noqa: E402


def one(dtype, N, T, d, chimax, C, loss, bbopt, iters, sep, seed, bal):
    ds, W = problem(N, T, d, min(3, chimax), C, seed, DT[dtype], balanced=bal)
    opts = RC.SweepOptions(chi_max=chimax, eta=0.05 if bbopt == "TSGO" else 0.01, update_iters=iters, loss_grad=loss, bbopt=bbopt,
                           train_classes_separately=sep)
    tol = TOL["f32" if dtype in ("float32", "complex64") else "f64"]
    eng = mt.SweepEngine(0)
    try:
        eng.set_options(chi_max=opts.chi_max, eta=opts.eta, cutoff=opts.cutoff, update_iters=opts.update_iters, loss=opts.loss_grad, bbopt=opts.bbopt,
                        rescale=opts.rescale, train_classes_separately=sep)
        eng.set_dataset(0, ds.phi.astype(DT[dtype]), ds.label_index, C)
        flips = 0
        for q in range(2 * (T - 1)):
            lid, gl = bond_of(q, T)
            ls = lid + 1 if gl else lid
            eng.set_mps([t.astype(DT[dtype]) for t in W], label_site=ls)
            eng.build_caches()
            assert eng.info()["typed_kernels"]
            LE, RE = caches_around(W, ds.phi, ls)
            y = np.abs(R.contract_mps(W, ds.phi))
            y = y[np.arange(N), ds.label_index] if y.ndim == 2 else y
            kappa = float(np.median(y) / max(y.min(), 1e-300))
            # (kappa is a proxy: the cancellation INSIDE an overlap is not in it; every further optimiser step of a bond starts from the
            # previous step's rounded tensor)
            scale = (max(1.0, kappa / 30.0) if loss == "KLD" else 1.0) * iters
            tr = {}
            RC.bond_step(W, LE, RE, lid, ds, opts, gl, tr)
            got = eng.bond_step(lid, gl)
            assert abs(got["loss"] - tr["loss"]) / max(1.0, abs(tr["loss"])) < tol["loss"] * scale, ("loss", q, kappa)
            assert abs(got["grad_norm"] - tr["grad_norm"]) / tr["grad_norm"] < tol["grad"] * scale, ("grad", q, kappa)
            nk = min(got["chi"], tr["chi"])
            assert np.abs(got["S"][:nk] - tr["S"][:nk]).max() / tr["S"][0] < tol["S"] * scale, ("S", q, kappa)
            if got["chi"] != tr["chi"]:
                flips += 1
                continue
            Wg = eng.get_mps()
            a, b = two_site(Wg[lid], Wg[lid + 1]), two_site(W[lid], W[lid + 1])
            assert np.abs(a - b).max() / np.abs(b).max() < tol["bond"] * scale, ("bond", q, kappa)
        assert flips <= 1, flips
    finally:
        eng.close()


def main(seed=0, cases=12):
    rng = np.random.default_rng(seed)
    bad = 0
    for case in range(cases):
        dtype = str(rng.choice(["complex128", "complex64", "float32", "float64"]))
        d = int(rng.choice([2, 3, 4, 5, 8]))
        chimax = int(rng.choice([c for c in (3, 5, 8, 12, 16) if d * c <= 64]))
        N = int(rng.choice([40, 96, 130, 257]))
        T = int(rng.integers(2, 8))
        C = int(rng.integers(1, 4))
        loss = str(rng.choice(["KLD", "KLD", "MSE"]))
        bbopt = "GD" if loss == "MSE" else str(rng.choice(["TSGO", "GD"]))
        iters = int(rng.integers(1, 3))
        sep = bool(rng.integers(0, 2)) and loss == "KLD"
        bal = bool(rng.integers(0, 2))
        if dtype == "float64":
            os.environ["MPST_TYPED"] = "1"
        try:
            one(dtype, N, T, d, chimax, C, loss, bbopt, iters, sep, 100 + case, bal)
            print("ok  ", dtype, N, T, d, chimax, C, loss, bbopt, iters, sep, flush=True)
        except BaseException as e:
            bad += 1
            print("FAIL", dtype, N, T, d, chimax, C, loss, bbopt, iters, sep, repr(e)[:300], flush=True)
        finally:
            os.environ.pop("MPST_TYPED", None)
    print("failures:", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(*(int(x) for x in sys.argv[1:3])) else 0)
