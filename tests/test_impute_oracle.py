"""The imputation oracle (oracle/impute_numpy.py, a line-by-line restatement of src/Imputation/MPS_methods.jl) against an
independent definition of the same conditional densities: density-matrix environments over the full chain with every
other missing site traced out, no preconditioning, no orthogonalisation."""
import numpy as np
import pytest

from oracle import impute_numpy as I
from oracle import ref_numpy as R


def _setup(T=7, d=3, chi=4, seed=0, ngrid=201):
    rng = np.random.default_rng(seed)
    W = R.random_mps(T, d, chi, 2, rng)
    mps = I.expand_label_index(W)[1]
    xs = np.linspace(-1.0, 1.0, ngrid)
    grid_phi = R.legendre_encode(xs, d)
    x = rng.uniform(-1, 1, T)
    enc = R.legendre_encode(x, d)
    return mps, xs, grid_phi, enc, rng


@pytest.mark.parametrize("missing", [[2], [0, 1], [3, 4, 5], [1, 3, 6], [0, 2, 4, 6], list(range(7))])
@pytest.mark.parametrize("order", ["forwards", "backwards"])
def test_median_path_equals_brute_force_conditionals(missing, order):
    mps, xs, grid_phi, enc, rng = _setup(seed=len(missing))
    T = len(mps)
    xo, eo = I.impute(mps, enc, missing, xs, grid_phi, "median", order)
    known = np.ones(T, dtype=bool)
    known[missing] = False
    fixed = {}
    seq = sorted(missing) if order == "forwards" else sorted(missing, reverse=True)
    pos = {j: k for k, j in enumerate(sorted(missing))}
    for j in seq:
        p = I.brute_force_conditional(mps, enc, known, j, fixed, grid_phi)
        cdf = I.cumul_trapz_even(xs, p)
        k = int(np.argmin(np.abs(cdf / cdf[-1] - 0.5)))
        assert abs(xs[k] - xo[pos[j]]) <= (xs[1] - xs[0]) * 1.0000001, (j, xs[k], xo[pos[j]])
        kk = int(np.argmin(np.abs(xs - xo[pos[j]])))
        wm = I.weighted_median(np.abs(xs - xs[kk]), p / cdf[-1])
        assert abs(wm - eo[pos[j]]) <= 1e-9
        fixed[j] = grid_phi[kk]                    # condition on what the oracle chose (scale is irrelevant)


def test_mode_and_quantile_paths():
    mps, xs, grid_phi, enc, rng = _setup(T=6, seed=5)
    missing = [1, 2, 4]
    xm, _ = I.impute(mps, enc, missing, xs, grid_phi, "mode")
    known = np.ones(6, dtype=bool)
    known[missing] = False
    fixed = {}
    for k, j in enumerate(missing):
        p = I.brute_force_conditional(mps, enc, known, j, fixed, grid_phi)
        assert xs[int(np.argmax(p))] == xm[k]
        fixed[j] = grid_phi[int(np.argmax(p))]
    u = rng.uniform(0, 1, 3)
    xq, _ = I.impute(mps, enc, missing, xs, grid_phi, "quantile", u=u)
    fixed = {}
    for k, j in enumerate(missing):
        p = I.brute_force_conditional(mps, enc, known, j, fixed, grid_phi)
        cdf = I.cumul_trapz_even(xs, p)
        kk = int(np.argmin(np.abs(cdf / cdf[-1] - u[k])))
        assert abs(xs[kk] - xq[k]) <= (xs[1] - xs[0]) * 1.0000001
        fixed[j] = grid_phi[int(np.argmin(np.abs(xs - xq[k])))]


@pytest.mark.parametrize("order", ["forwards", "backwards"])
def test_mean_path_equals_brute_force_moments(order):
    mps, xs, grid_phi, enc, rng = _setup(T=6, seed=9, ngrid=401)
    d = grid_phi.shape[1]
    missing = [0, 2, 3, 5]
    f = lambda x: R.legendre_encode(x, d)
    xo, so = I.impute(mps, enc, missing, xs, grid_phi, "mean", order, True, None, encode=f)
    known = np.ones(6, dtype=bool)
    known[missing] = False
    fixed = {}
    seq = missing if order == "forwards" else missing[::-1]
    for j in seq:
        p = I.brute_force_conditional(mps, enc, known, j, fixed, grid_phi)
        # the reference's estimator (rectangle sums over a trapezoid normalisation, sampling_utils.jl:82-93) on the
        # independently computed density
        dx = xs[1] - xs[0]
        Z = dx * (np.sum(p) - 0.5 * (p[0] + p[-1]))
        ex = np.sum(xs * p) * dx / Z
        k = missing.index(j)
        assert abs(ex - xo[k]) < 1e-10
        assert abs(np.sqrt(np.sum((xs - ex) ** 2 * p) * dx / Z) - so[k]) < 1e-10
        fixed[j] = f(xo[k])


def test_rejection_sampling_keeps_the_first_sample_inside_the_window():
    mps, xs, grid_phi, enc, rng = _setup(T=6, seed=11, ngrid=401)
    missing = [1, 2, 4]
    u = rng.uniform(0, 1, (3, 8))
    thr = 0.7
    xr, wr = I.impute(mps, enc, missing, xs, grid_phi, "ITS", "forwards", True, u, rejection_threshold=thr, max_trials=8)
    known = np.ones(6, dtype=bool)
    known[missing] = False
    fixed = {}
    for k, j in enumerate(missing):
        p = I.brute_force_conditional(mps, enc, known, j, fixed, grid_phi)
        cdf = I.cumul_trapz_even(xs, p)
        cdf /= cdf[-1]
        med = xs[int(np.argmin(np.abs(cdf - 0.5)))]
        draws = [xs[int(np.argmin(np.abs(cdf - t)))] for t in u[k]]
        ok = [abs(x - med) < thr * wr[k] for x in draws]
        expect = draws[ok.index(True)] if any(ok) else draws[-1]
        assert abs(expect - xr[k]) <= (xs[1] - xs[0]) * 1.0000001
        fixed[j] = grid_phi[int(np.argmin(np.abs(xs - xr[k])))]
    # a threshold nothing can miss is plain inverse-transform sampling of the first number
    x0, _ = I.impute(mps, enc, missing, xs, grid_phi, "ITS", "forwards", True, u, rejection_threshold=1e9, max_trials=8)
    x1, _ = I.impute(mps, enc, missing, xs, grid_phi, "quantile", "forwards", True, u[:, 0])
    assert np.array_equal(x0, x1)


def test_weighted_median_matches_definition():
    rng = np.random.default_rng(1)
    v = rng.uniform(0, 1, 101)
    w = rng.uniform(0, 1, 101)
    m = I.weighted_median(v, w)
    o = np.argsort(v)
    cw = np.cumsum(w[o])
    assert m == v[o][np.argmax(cw > w.sum() / 2)]
    w2 = np.zeros(5)
    w2[3] = 1.0
    assert I.weighted_median(np.arange(5.0), w2) == 3.0


# ---- the committed imputation instance (tests/golden/make_golden_complex_impute.py) and, when a maintainer has written it, the reference's own answer ----
import os

_IFIX = os.path.join(os.path.dirname(__file__), "golden", "impute_median_c1.npz")
_IREF = os.path.join(os.path.dirname(__file__), "golden", "juliaref_impute_median_c1.npz")


def _instance():
    g = np.load(_IFIX)
    T = len(g["x"])
    return g, [g[f"mps_{j}"] for j in range(T)]


def test_imputation_fixture_is_reproduced():
    g, mps = _instance()
    for order in ("forwards", "backwards"):
        x, err = I.impute(mps, g["enc"], g["missing"], g["xs"], g["grid_phi"], "median", order, True)
        assert np.array_equal(x, g[f"x_{order}"]) and np.allclose(err, g[f"wmad_{order}"], rtol=0, atol=1e-12)
        assert len(x) == len(g["missing"]) and np.all(np.abs(x) <= 1.0)


@pytest.mark.skipif(not os.path.exists(_IREF), reason="no tests/golden/juliaref_impute_median_c1.npz: a maintainer with Julia writes it with "
                                                      "tests/golden/make_reference_goldens.jl (the reference's impute_median on this instance)")
def test_imputation_oracle_against_reference_vectors():
    """impute_median(...; get_wmad = true) of the REFERENCE (MPS_methods.jl:201-230) on the fixture's class MPS, series and grid: the same
    grid values (exact: they are grid points) and the same weighted median absolute deviations, both imputation orders."""
    g, mps = _instance()
    ref = np.load(_IREF)
    for order in ("forwards", "backwards"):
        x, err = I.impute(mps, g["enc"], g["missing"], g["xs"], g["grid_phi"], "median", order, True)
        assert np.allclose(x, ref[f"x_{order}"], rtol=0, atol=1e-12)
        assert np.allclose(err, ref[f"wmad_{order}"], rtol=0, atol=1e-9)
