"""The device-side imputation engine (mpst_impute, through the C ABI) against the NumPy restatement of
src/Imputation/MPS_methods.jl, instance by instance: same imputed grid value at every missing site (a grid step of
slack where the cumulative density sits within rounding of the target between two neighbouring grid values), same
weighted median absolute deviation."""
import numpy as np
import pytest

import mpstime_jl_amd as mt
from oracle import impute_numpy as I
from oracle import ref_numpy as R

pytestmark = pytest.mark.gpu


def _problem(N, T, d, chi, C, seed, ngrid=2001):
    rng = np.random.default_rng(seed)
    W = R.random_mps(T, d, chi, C, rng)
    xs = -1.0 + (2.0 / (ngrid - 1)) * np.arange(ngrid)
    grid_phi = R.legendre_encode(xs, d)
    X = rng.uniform(-0.95, 0.95, (N, T))
    y = np.sort(rng.integers(0, C, N)).astype(np.int32)
    phi = R.legendre_encode(X, d)
    return W, xs, grid_phi, X, y, phi, rng


def _masks(N, T, rng, kind):
    m = np.zeros((N, T), dtype=np.uint8)
    for i in range(N):
        if kind == "block":
            n = int(rng.integers(1, T))
            s = int(rng.integers(0, T - n + 1))
            m[i, s:s + n] = 1
        elif kind == "scatter":
            m[i] = rng.uniform(size=T) < 0.4
            if m[i].sum() == 0:
                m[i, rng.integers(0, T)] = 1
        elif kind == "all":
            m[i] = 1
        elif kind == "none_some":
            if i % 3:
                m[i, rng.integers(0, T)] = 1
    return m


def _compare(W, xs, grid_phi, phi, y, m, x_g, e_g, method, u=None, wmad=True, order="forwards", **kw):
    classes = I.expand_label_index(W)
    dx = xs[1] - xs[0]
    nflip = 0
    for i in range(len(y)):
        sites = np.flatnonzero(m[i])
        if len(sites) == 0:
            assert np.all(x_g[i] == 0.0)
            continue
        ui = None if u is None else u[i, sites]
        xo, eo = I.impute(classes[y[i]], phi[i], sites, xs, grid_phi, method, order, wmad, ui, **kw)
        diff = np.abs(x_g[i, sites] - xo)
        if order == "backwards":
            diff = diff[::-1]                   # in the order the sites were imputed
        if np.any(diff > 1e-12):
            # a flip between neighbouring grid values changes the conditioning of the later sites: compare up to the first
            first = int(np.argmax(diff > 1e-12))
            assert diff[first] <= dx * 1.0000001, (i, sites, first, x_g[i, sites], xo)
            assert np.all(diff[:first] <= 1e-12)
            nflip += 1
            continue
        if (method == "median" and wmad) or kw.get("rejection_threshold") is not None:
            assert np.abs(e_g[i, sites] - eo).max() <= dx * 1.0000001, (i, e_g[i, sites], eo)
            assert np.mean(np.abs(e_g[i, sites] - eo) <= 1e-9) > 0.9
        assert np.all(x_g[i][m[i] == 0] == 0.0)
    return nflip


@pytest.mark.parametrize("kind", ["block", "scatter", "all", "none_some"])
@pytest.mark.parametrize("cfg", [(24, 12, 4, 8, 2), (18, 9, 3, 5, 3), (10, 16, 8, 20, 1)], ids=["d4chi8", "d3chi5C3", "d8chi20"])
def test_median_matches_the_oracle(engine_cls, cfg, kind):
    N, T, d, chi, C = cfg
    W, xs, grid_phi, X, y, phi, rng = _problem(N, T, d, chi, C, seed=N + T)
    m = _masks(N, T, rng, kind)
    eng = engine_cls(0)
    try:
        eng.set_options(chi_max=chi)
        eng.set_dataset(1, phi, y, C)
        eng.set_mps(W)
        x_g, e_g, secs = eng.impute(1, m, xs, grid_phi, 0, True)
    finally:
        eng.close()
    assert _compare(W, xs, grid_phi, phi, y, m, x_g, e_g, "median") <= max(1, N // 8)


def test_mode_and_inverse_transform_sampling(engine_cls):
    N, T, d, chi, C = 20, 10, 4, 6, 2
    W, xs, grid_phi, X, y, phi, rng = _problem(N, T, d, chi, C, seed=77)
    m = _masks(N, T, rng, "scatter")
    u = rng.uniform(0.02, 0.98, (N, T))
    eng = engine_cls(0)
    try:
        eng.set_options(chi_max=chi)
        eng.set_dataset(1, phi, y, C)
        eng.set_mps(W)
        x_mode, _, _ = eng.impute(1, m, xs, grid_phi, 1, False)
        x_its, _, _ = eng.impute(1, m, xs, grid_phi, 2, False, u)
    finally:
        eng.close()
    assert _compare(W, xs, grid_phi, phi, y, m, x_mode, None, "mode") <= 2
    assert _compare(W, xs, grid_phi, phi, y, m, x_its, None, "quantile", u=u) <= 2


@pytest.mark.parametrize("kind", ["block", "scatter", "all"])
@pytest.mark.parametrize("cfg", [(24, 12, 4, 8, 2), (12, 9, 3, 5, 3), (10, 16, 8, 20, 1)], ids=["d4chi8", "d3chi5C3", "d8chi20"])
def test_backwards_order(engine_cls, cfg, kind):
    """impute_order = :backwards: the last missing site first, each conditioned on the ones to its right."""
    N, T, d, chi, C = cfg
    W, xs, grid_phi, X, y, phi, rng = _problem(N, T, d, chi, C, seed=3 * N + T)
    m = _masks(N, T, rng, kind)
    u = rng.uniform(0.02, 0.98, (N, T))
    eng = engine_cls(0)
    try:
        eng.set_options(chi_max=chi)
        eng.set_dataset(1, phi, y, C)
        eng.set_mps(W)
        x_g, e_g, _ = eng.impute(1, m, xs, grid_phi, 0, True, order=1)
        x_mode, _, _ = eng.impute(1, m, xs, grid_phi, 1, False, order=1)
        x_its, _, _ = eng.impute(1, m, xs, grid_phi, 2, False, u, order=1)
        x_fwd, _, _ = eng.impute(1, m, xs, grid_phi, 0, True, order=0)
    finally:
        eng.close()
    assert _compare(W, xs, grid_phi, phi, y, m, x_g, e_g, "median", order="backwards") <= max(1, N // 8)
    assert _compare(W, xs, grid_phi, phi, y, m, x_mode, None, "mode", order="backwards") <= 2
    # the k-th uniform number belongs to the k-th site imputed: last missing site first
    nflip = 0
    classes = I.expand_label_index(W)
    for i in range(N):
        sites = np.flatnonzero(m[i])
        xo, _ = I.impute(classes[y[i]], phi[i], sites, xs, grid_phi, "quantile", "backwards", False, u[i, sites][::-1])
        nflip += int(np.any(np.abs(x_its[i, sites] - xo) > 1e-12))
        assert np.abs(x_its[i, sites] - xo)[-1] <= (xs[1] - xs[0]) * 1.0000001
    assert nflip <= 2
    if kind != "none_some":
        multi = m.sum(axis=1) > 1
        assert np.any(np.abs(x_g - x_fwd)[multi] > 1e-6)           # the two orders are different estimators


@pytest.mark.parametrize("order", [0, 1], ids=["forwards", "backwards"])
@pytest.mark.parametrize("norm", [False, True], ids=["legendre_no_norm", "legendre"])
def test_mean_and_standard_deviation(engine_cls, order, norm):
    """impute_mean: expectation value and standard deviation of the conditional density; the chain is re-conditioned on
    the encoding of the expectation value (evaluated on the device), so the values are continuous: compared to 1e-9."""
    N, T, d, chi, C = 16, 11, 5, 9, 2
    rng = np.random.default_rng(123)
    W = R.random_mps(T, d, chi, C, rng)
    ngrid = 1501
    xs = -1.0 + (2.0 / (ngrid - 1)) * np.arange(ngrid)
    enc = lambda x: R.legendre_encode(x, d, norm=norm)
    grid_phi = enc(xs)
    X = rng.uniform(-0.95, 0.95, (N, T))
    y = np.sort(rng.integers(0, C, N)).astype(np.int32)
    phi = enc(X)
    m = _masks(N, T, rng, "scatter")
    eng = engine_cls(0)
    try:
        eng.set_options(chi_max=chi)
        eng.set_dataset(1, phi, y, C)
        eng.set_mps(W)
        x_g, e_g, _ = eng.impute(1, m, xs, grid_phi, 3, True, order=order, mean_basis=0 if norm else 1)
        x_n, e_n, _ = eng.impute(1, m, xs, grid_phi, 3, False, order=order, mean_basis=0 if norm else 1)
    finally:
        eng.close()
    classes = I.expand_label_index(W)
    for i in range(N):
        sites = np.flatnonzero(m[i])
        xo, eo = I.impute(classes[y[i]], phi[i], sites, xs, grid_phi, "mean", ["forwards", "backwards"][order], True, None, encode=enc)
        assert np.abs(x_g[i, sites] - xo).max() < 1e-9, (i, x_g[i, sites], xo)
        assert np.abs(e_g[i, sites] - eo).max() < 1e-9
        assert np.all(x_g[i][m[i] == 0] == 0.0)
    assert np.array_equal(x_g, x_n) and np.all(e_n == 0.0)


@pytest.mark.parametrize("order", [0, 1], ids=["forwards", "backwards"])
def test_inverse_transform_sampling_with_rejection(engine_cls, order):
    """get_sample_from_rdm with rejection_threshold: up to max_trials samples, the first within threshold * WMAD of the
    median is kept.  A tight threshold makes most sites use several trials; a huge one must reproduce plain ITS."""
    N, T, d, chi, C = 20, 10, 4, 6, 2
    W, xs, grid_phi, X, y, phi, rng = _problem(N, T, d, chi, C, seed=91)
    m = _masks(N, T, rng, "scatter")
    K = 6
    u = rng.uniform(0.02, 0.98, (N, T, K))
    oname = ["forwards", "backwards"][order]
    eng = engine_cls(0)
    try:
        eng.set_options(chi_max=chi)
        eng.set_dataset(1, phi, y, C)
        eng.set_mps(W)
        x_r, e_r, _ = eng.impute(1, m, xs, grid_phi, 4, True, u, order=order, max_trials=K, rejection_threshold=0.8)
        x_big, _, _ = eng.impute(1, m, xs, grid_phi, 4, True, u, order=order, max_trials=K, rejection_threshold=1e9)
        x_its, _, _ = eng.impute(1, m, xs, grid_phi, 2, False, np.ascontiguousarray(u[:, :, 0]), order=order)
    finally:
        eng.close()
    assert np.array_equal(x_big, x_its)
    classes = I.expand_label_index(W)
    nflip = 0
    used_more = 0
    for i in range(N):
        sites = np.flatnonzero(m[i])
        ui = u[i, sites] if order == 0 else u[i, sites][::-1]
        xo, eo = I.impute(classes[y[i]], phi[i], sites, xs, grid_phi, "ITS", oname, True, ui, rejection_threshold=0.8, max_trials=K)
        x1, _ = I.impute(classes[y[i]], phi[i], sites, xs, grid_phi, "ITS", oname, True, ui[:, :1])
        used_more += int(np.any(np.abs(xo - x1) > 1e-9))
        if np.any(np.abs(x_r[i, sites] - xo) > 1e-12):
            nflip += 1
            continue
        assert np.abs(e_r[i, sites] - eo).max() <= (xs[1] - xs[0]) * 1.0000001
    assert nflip <= 2 and used_more >= N // 2


def test_full_grid_on_the_reference_trained_mps(engine_cls):
    """The reference's own trained ECG200 MPS (d=5, chi=25, T=96) with the reference's default grid (dx = 1e-4: 20 001
    candidate values), block-missing masks (the reference's default MAR mechanism) of 20-80 % on a subset of its training
    series, through the package API and against the oracle."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_ecg200_trained_mps.npz"))
    T = z["pstates"].shape[1]
    W = [z[f"W_{j}"] for j in range(T)]
    cd = z["class_distribution"]
    y = np.repeat(np.arange(len(cd)), cd)
    X = z["original_data"]
    opts = mt.MPSOptions(verbosity=-1)
    td = mt.EncodedTimeSeriesSet(z["pstates"], y, y.astype(np.int32), X, cd)
    trained = mt.TrainedMPS(W, opts, td)
    imp = mt.init_imputation_problem(trained, X, y, verbosity=0)
    assert len(imp.x_guess_range.xvals) == 20001
    rng = np.random.default_rng(3)
    rows = np.array([0, 5, 17, 40, 69, 75, 90, 99])
    mask = np.zeros((len(rows), T), dtype=bool)
    for k in range(len(rows)):
        _, idx = mt.mar(X[rows[k]], [0.2, 0.5, 0.8][k % 3], rng)
        mask[k, idx] = True
    ts, err, secs = mt.impute_dataset(imp, mask, "median", rows=rows, invert_transform=False, return_seconds=True)
    from mpstime_jl_amd.imputation import _scaled_instances
    enc, norms, raw, full, scaled, oob = _scaled_instances(imp, rows, mask)
    phi = enc.encode(scaled, opts.d)
    classes = I.expand_label_index(W)
    nflip = 0
    for k in range(len(rows)):
        sites = np.flatnonzero(mask[k])
        xo, eo = I.impute(classes[y[rows[k]]], phi[k], sites, imp.x_guess_range.xvals, imp.x_guess_range.xvals_enc, "median")
        diff = np.abs(ts[k, sites] - xo)
        if np.any(diff > 1e-12):
            first = int(np.argmax(diff > 1e-12))
            assert diff[first] <= 1.0000001e-4 and np.all(diff[:first] <= 1e-12)
            nflip += 1
        else:
            assert np.abs(err[k, sites] - eo).max() <= 1.0000001e-4
        assert np.array_equal(ts[k][~mask[k]], scaled[k][~mask[k]])
    assert nflip <= 2
    # original units: the known values come back exactly where the out-of-bounds rescale did not touch them, the
    # imputed block stays within the range the training data spans, and the error estimate is positive
    ts_o, err_o = mt.impute_dataset(imp, mask, "median", rows=rows)
    assert np.allclose(ts_o[~mask], X[rows][~mask], rtol=1e-9, atol=1e-9)
    assert np.all(np.isfinite(ts_o)) and np.all(err_o[mask] >= 0)
    # the model imputes its own training data far better than a flat baseline does
    flat = np.abs(np.mean(X) - X[rows])[mask].mean()
    assert np.abs(ts_o - X[rows])[mask].mean() < 0.6 * flat
    # MPS_impute: the reference's per-instance entry point and its metrics
    cls = int(y[rows[0]])
    inst = int(np.flatnonzero(np.flatnonzero(y == cls) == rows[0])[0])
    t1, e1, target, metrics = mt.MPS_impute(imp, cls, inst, np.flatnonzero(mask[0]), "median")
    assert np.allclose(t1[0], ts_o[0]) and set(metrics[0]) == {"MAE", "MAPE", "NN_MAE", "NN_MAPE"}
    assert abs(metrics[0]["MAE"] - np.abs(ts_o[0] - X[rows[0]])[mask[0]].mean()) < 1e-12
