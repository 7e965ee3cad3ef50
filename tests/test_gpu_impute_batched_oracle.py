"""The batched imputation sweep (k_imp_leftb: sixteen instances per workgroup, closed-form densities - what `bench.py --workload impute`
times) DIRECTLY against the NumPy restatement of src/Imputation/MPS_methods.jl:42-177 (oracle/impute_numpy.py): all five methods, both
orders, complex (Fourier) and real (Legendre) models; fp64 chain: the oracle's grid values; fp32 chain: within stated grid steps at the
first site that differs.  Then BASELINE configs[4] at FULL size (N = 8192, T = 200, chi = 64, d = 8, complex64 model, fp32 chain, the
reference's 20 001-value grid, imputation.jl:90-107): properties on all instances, 16 sampled instances against the oracle.  And the two
fuzzers (tests/fuzz_impute.py, tests/fuzz_batch.py) as seeded, bounded cases."""
import numpy as np
import pytest

from oracle import impute_numpy as I
from tests.test_gpu_impute_model import _check, _problem

pytestmark = pytest.mark.gpu

METHODS = ["median", "mode", "quantile", "mean", "its_reject"]


def _oracle(W, phi, y, i, sites, xs, grid_phi, method, order, u, enc):
    classes = I.expand_label_index(W)
    ui = None if u is None else (u[i, sites] if order == "forwards" else u[i, sites][::-1])
    if method == "its_reject":
        return I.impute(classes[y[i]], phi[i], sites, xs, grid_phi, "quantile", order, True, ui, rejection_threshold=1.0, max_trials=3)
    if method == "mean":
        return I.impute(classes[y[i]], phi[i], sites, xs, grid_phi, "mean", order, True, None, encode=enc)
    if method == "quantile":
        return I.impute(classes[y[i]], phi[i], sites, xs, grid_phi, "quantile", order, False, None if ui is None else ui[:, :1])
    return I.impute(classes[y[i]], phi[i], sites, xs, grid_phi, method, order, method == "median", None)


@pytest.mark.parametrize("order", ["forwards", "backwards"])
@pytest.mark.parametrize("compute", ["f64", "f32"])
@pytest.mark.parametrize("cx,d,chi", [(True, 8, 20), (False, 4, 33)], ids=["fourier_d8", "legendre_d4_chi33"])
def test_batched_sweep_against_the_oracle(engine_cls, compute, cx, d, chi, order):
    N, T, C = 21, 10, 3             # a full workgroup of 16 and a partial one; ragged missing patterns (_problem)
    W, xs, enc, grid_phi, X, y, phi, m, rng = _problem(N, T, d, chi, C, seed=4000 + d + chi, ngrid=2001, cx=cx)
    u = rng.uniform(0.02, 0.98, (N, T, 3))
    o = ["forwards", "backwards"].index(order)
    eng = engine_cls(0)
    try:
        runs = {"median": eng.impute_model(W, phi, y, m, xs, grid_phi, 0, True, order=o, compute=compute)[:2],
                "mode": eng.impute_model(W, phi, y, m, xs, grid_phi, 1, False, order=o, compute=compute)[:2],
                "quantile": eng.impute_model(W, phi, y, m, xs, grid_phi, 2, False, u[:, :, :1], order=o, compute=compute)[:2],
                "mean": eng.impute_model(W, phi, y, m, xs, grid_phi, 3, True, order=o, compute=compute)[:2],
                "its_reject": eng.impute_model(W, phi, y, m, xs, grid_phi, 4, True, u, max_trials=3, rejection_threshold=1.0, order=o,
                                               compute=compute)[:2]}
        info = eng.impute_info()
        assert info["closed_form_densities"] and info["batched_sweep"], info
    finally:
        eng.close()
    dx = xs[1] - xs[0]
    f64 = compute == "f64"
    for method in METHODS:
        xg, eg = runs[method]
        assert np.all(xg[m == 0] == 0.0) and np.all(np.isfinite(xg))
        flips = 0
        for i in range(N):
            sites = np.flatnonzero(m[i])
            if len(sites) == 0:
                continue
            xo, eo = _oracle(W, phi, y, i, sites, xs, grid_phi, method, order, u, enc)
            diff = np.abs(xg[i, sites] - xo)
            if order == "backwards":
                diff = diff[::-1]           # in the order the sites were imputed
            if method == "mean":
                # no selection: a smooth functional of the density (fp32 chain: products of up to 33 terms)
                assert diff.max() < (1e-9 if f64 else 5e-4), (method, i, diff.max())
                continue
            if np.any(diff > 1e-12):
                first = int(np.argmax(diff > 1e-12))
                # fp64 chain: a cumulative sum within rounding of a threshold moves the choice by one grid step, and the conditioning
                # carries the difference on; fp32 chain: a few steps of the 2001-value grid at the first site that differs
                assert diff[first] <= (1.0000001 if f64 else 4.0000001) * dx, (method, i, first, diff[first] / dx)
                assert np.all(diff[:first] <= 1e-12)
                flips += 1
            elif f64 and method in ("median", "its_reject") and eg is not None:
                assert np.abs(eg[i, sites] - eo).max() <= dx * 1.0000001, (method, i)
        assert flips <= (3 if f64 else N // 2), (method, flips)


def test_configs4_full_size_pass(engine_cls):
    """BASELINE configs[4]: N = 8192 instances, T = 200, chi = 64, d = 8, complex64 model (Fourier), fp32 chain, a 50 % block missing
    per instance, the reference's 20 001-value grid - the pass `bench.py --workload impute` times.  On ALL instances: finite values
    inside the grid range at every missing site, zeros at the known ones, uncertainties finite and non-negative; 16 sampled instances
    against the NumPy restatement: the first imputed site of every sampled instance within 3 grid steps, at least 85 % of their sites
    on the oracle's grid value (fp32 chain, flat conditionals of a random model: test_config5_element_type_and_shape has the same bar)."""
    import bench
    import mpstime_jl_amd as mt
    N, T, d, chi = 8192, 200, 8, 64
    rng = np.random.default_rng(100)
    W = bench.random_chain(T, d, chi, np.random.default_rng(7))
    enc = mt.model_encoding("Fourier")
    xs = -1.0 + 1e-4 * np.arange(20001)
    gphi = np.ascontiguousarray(enc.encode(xs, d), dtype=np.complex128)
    X = rng.uniform(-0.95, 0.95, (N, T))
    phi = np.ascontiguousarray(enc.encode(X, d), dtype=np.complex128)
    m = np.zeros((N, T), dtype=np.uint8)
    for i in range(N):
        s0 = rng.integers(0, T - T // 2 + 1)
        m[i, s0:s0 + T // 2] = 1
    lab = np.zeros(N, dtype=np.int32)
    eng = engine_cls(0)
    try:
        x, e, _ = eng.impute_model(W, phi, lab, m, xs, gphi, 0, True, compute="f32")
        info = eng.impute_info()
    finally:
        eng.close()
    assert info["closed_form_densities"] and info["batched_sweep"], info
    mask = m.astype(bool)
    assert np.all(np.isfinite(x)) and np.all(np.isfinite(e))
    assert np.all(x[~mask] == 0.0) and np.all(e[~mask] == 0.0)
    assert x[mask].min() >= xs[0] and x[mask].max() <= xs[-1]
    assert np.all(e[mask] >= 0.0) and e[mask].max() <= xs[-1] - xs[0]
    # every imputed value is a grid value
    k = np.rint((x[mask] - xs[0]) / 1e-4)
    assert np.abs(x[mask] - (xs[0] + 1e-4 * k)).max() < 1e-9
    cls = [w.reshape(w.shape[:3]) for w in W]
    dx = xs[1] - xs[0]
    same = tot = 0
    for i in np.random.default_rng(3).choice(N, 16, replace=False):
        sites = np.flatnonzero(m[i])
        xo, _ = I.impute(cls, phi[i], sites, xs, gphi, "median")
        diff = np.abs(x[i, sites] - xo)
        assert diff[0] <= 3 * dx, (i, diff[0] / dx)
        same += int(np.sum(diff <= 1e-12))
        tot += len(sites)
    assert same >= 0.85 * tot, (same, tot)


@pytest.mark.parametrize("seed", [0, 1])
def test_fuzz_impute_seeded(seed):
    """tests/fuzz_impute.py, bounded: 20 random (element type, N, T, d, chi, C, order, missing pattern, grid) cases per seed through the
    imputation engine against the oracle."""
    from tests import fuzz_impute
    assert fuzz_impute.main(cases=20, seed=seed) == 0


def test_fuzz_batch_seeded():
    """tests/fuzz_batch.py, bounded: random shapes and batch sizes (9..40 fits) through mpst_sweep_batch against separate sweeps, bit for bit."""
    from tests import fuzz_batch
    assert fuzz_batch.main(cases=2, seed=0) == 0
