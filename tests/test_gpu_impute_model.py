"""mpst_impute_model_run: the imputation engine on a model handed over in one call - complex models (the reference's
Fourier basis) and the fp32 chain arithmetic - against the NumPy restatement of src/Imputation/MPS_methods.jl
(oracle/impute_numpy.py, complex-aware: dag(...) on the known states and on the re-conditioning state)."""
import numpy as np
import pytest

from oracle import impute_numpy as I
from oracle import ref_numpy as R

pytestmark = pytest.mark.gpu


def _complex_mps(T, d, chi, C, rng):
    Wr = R.random_mps(T, d, chi, C, rng)
    Wi = R.random_mps(T, d, chi, C, rng)
    return [a + 1j * b for a, b in zip(Wr, Wi)]


def _problem(N, T, d, chi, C, seed, ngrid, cx):
    rng = np.random.default_rng(seed)
    W = _complex_mps(T, d, chi, C, rng) if cx else R.random_mps(T, d, chi, C, rng)
    xs = -1.0 + (2.0 / (ngrid - 1)) * np.arange(ngrid)
    enc = (lambda x: R.fourier_encode(x, d)) if cx else (lambda x: R.legendre_encode(x, d))
    X = rng.uniform(-0.95, 0.95, (N, T))
    y = rng.integers(0, C, N).astype(np.int32)              # any order: the model-run entry does not need sorted classes
    m = (rng.uniform(size=(N, T)) < 0.4).astype(np.uint8)
    m[0] = 1
    m[1] = 0
    m[2, :] = 0
    m[2, T // 2] = 1
    return W, xs, enc, enc(xs), X, y, enc(X), m, rng


def _check(W, xs, grid_phi, phi, y, m, x_g, e_g, method, order="forwards", wmad=True, u=None, max_flips=2, **kw):
    classes = I.expand_label_index(W)
    dx = xs[1] - xs[0]
    nflip = 0
    for i in range(len(y)):
        sites = np.flatnonzero(m[i])
        assert np.all(x_g[i][m[i] == 0] == 0.0)
        if len(sites) == 0:
            continue
        ui = None if u is None else (u[i, sites] if order == "forwards" else u[i, sites][::-1])
        xo, eo = I.impute(classes[y[i]], phi[i], sites, xs, grid_phi, method, order, wmad, ui, **kw)
        diff = np.abs(x_g[i, sites] - xo)
        if order == "backwards":
            diff = diff[::-1]
        if np.any(diff > 1e-12):
            first = int(np.argmax(diff > 1e-12))
            assert diff[first] <= dx * 1.0000001, (i, sites, x_g[i, sites], xo)
            assert np.all(diff[:first] <= 1e-12)
            nflip += 1
            continue
        if method == "median" and wmad:
            assert np.abs(e_g[i, sites] - eo).max() <= dx * 1.0000001
    assert nflip <= max_flips, nflip


@pytest.mark.parametrize("order", ["forwards", "backwards"])
@pytest.mark.parametrize("cfg", [(14, 9, 3, 5, 2), (10, 12, 4, 9, 3), (6, 10, 8, 20, 1), (3, 5, 2, 3, 2), (35, 7, 16, 6, 2)],
                         ids=["d3chi5", "d4chi9C3", "d8chi20", "d2chi3_N3", "d16chi6_N35"])
def test_complex_fourier_model_fp64(engine_cls, cfg, order):
    N, T, d, chi, C = cfg
    W, xs, enc, grid_phi, X, y, phi, m, rng = _problem(N, T, d, chi, C, seed=5 * N + T, ngrid=1201, cx=True)
    o = ["forwards", "backwards"].index(order)
    u = rng.uniform(0.02, 0.98, (N, T))
    eng = engine_cls(0)
    try:
        x_med, e_med, _ = eng.impute_model(W, phi, y, m, xs, grid_phi, 0, True, order=o)
        x_mode, _, _ = eng.impute_model(W, phi, y, m, xs, grid_phi, 1, False, order=o)
        x_its, _, _ = eng.impute_model(W, phi, y, m, xs, grid_phi, 2, False, u, order=o)
        x_mean, e_mean, _ = eng.impute_model(W, phi, y, m, xs, grid_phi, 3, True, order=o)
        assert eng.impute_info()["closed_form_densities"]          # Fourier states on a uniform grid: recognised
    finally:
        eng.close()
    _check(W, xs, grid_phi, phi, y, m, x_med, e_med, "median", order)
    _check(W, xs, grid_phi, phi, y, m, x_mode, None, "mode", order)
    _check(W, xs, grid_phi, phi, y, m, x_its, None, "quantile", order, u=u)
    classes = I.expand_label_index(W)
    for i in range(N):
        sites = np.flatnonzero(m[i])
        if len(sites) == 0:
            continue
        xo, eo = I.impute(classes[y[i]], phi[i], sites, xs, grid_phi, "mean", order, True, None, encode=enc)
        assert np.abs(x_mean[i, sites] - xo).max() < 1e-9 and np.abs(e_mean[i, sites] - eo).max() < 1e-9


@pytest.mark.parametrize("compute", ["f64", "f32"])
@pytest.mark.parametrize("cx,d", [(True, 8), (False, 4), (False, 12)], ids=["fourier_d8", "legendre_d4", "legendre_d12"])
def test_closed_form_densities_equal_the_table_path(engine_cls, compute, cx, d, monkeypatch):
    """Fourier (complex models) or Legendre (real models) grid states on a uniform grid: densities and cumulative sums from 2d
    coefficients (k_imp_left<..., TRIG>: geometric series / Euler-Maclaurin) against the same kernel streaming the table of
    grid states (MPST_IMP_NO_TRIG=1), every method, on the reference's 20 001-value grid; a grid that is not uniform, or states
    that are not the basis, keep the table path."""
    N, T, chi, C = 12, 16, 12, 2
    W, xs, enc, grid_phi, X, y, phi, m, rng = _problem(N, T, d, chi, C, seed=77, ngrid=20001, cx=cx)
    u = rng.uniform(0.02, 0.98, (N, T, 3))
    eng = engine_cls(0)
    try:
        runs = {}
        for tag in ("trig", "table"):
            if tag == "table":
                monkeypatch.setenv("MPST_IMP_NO_TRIG", "1")
            runs[tag] = [eng.impute_model(W, phi, y, m, xs, grid_phi, 0, True, compute=compute)[:2],
                         eng.impute_model(W, phi, y, m, xs, grid_phi, 1, False, compute=compute)[:2],
                         eng.impute_model(W, phi, y, m, xs, grid_phi, 2, False, u[:, :, :1], compute=compute)[:2],
                         eng.impute_model(W, phi, y, m, xs, grid_phi, 3, True, compute=compute)[:2],
                         eng.impute_model(W, phi, y, m, xs, grid_phi, 4, True, u, max_trials=3, rejection_threshold=1.0, compute=compute)[:2]]
            assert eng.impute_info()["closed_form_densities"] == (tag == "trig")
        monkeypatch.delenv("MPST_IMP_NO_TRIG")
        xs2 = xs.copy()
        xs2[5000] += 1e-6
        eng.impute_model(W, phi, y, m, xs2, grid_phi, 0, True, compute=compute)
        assert not eng.impute_info()["closed_form_densities"]
        gp2 = grid_phi.copy()
        gp2[:, 3] = np.conj(gp2[:, 3]) if cx else -gp2[:, 3]
        eng.impute_model(W, phi, y, m, xs, gp2, 0, True, compute=compute)
        assert not eng.impute_info()["closed_form_densities"]
    finally:
        eng.close()
    mask = m.astype(bool)
    dx = xs[1] - xs[0]
    for k, ((xa, ea), (xb, eb)) in enumerate(zip(runs["trig"], runs["table"])):
        if k == 3:          # mean: sums over the grid in another order
            # (fp32 chain: the closed-form sweep forms its products on the matrix pipe, the table kernel with FMAs - other sums)
            tol = 1e-11 if compute == "f64" else 2e-4
            assert np.abs(xa - xb)[mask].max() < tol and np.abs(ea - eb)[mask].max() < tol
            continue
        # selections: the same grid value, up to a cumulative sum that lands within rounding of a threshold (then one step, and
        # the conditioning carries the difference on: such instances are counted, not compared further)
        bad = [i for i in range(N) if not np.array_equal(xa[i], xb[i])]
        assert len(bad) <= 1, (k, bad)
        for i in bad:
            first = int(np.argmax(xa[i] != xb[i]))
            assert abs(xa[i, first] - xb[i, first]) <= dx * 1.0000001
        good = [i for i in range(N) if i not in bad]
        if ea is not None and k in (0, 4):
            assert np.abs(ea[good] - eb[good]).max() <= dx * 1.0000001


def test_real_model_run_is_the_context_path(engine_cls):
    """Same kernels, same operands: handing the model over in one call gives bit-identical results to the data set + MPS
    of a context (class-sorted there, any order here)."""
    N, T, d, chi, C = 18, 11, 4, 8, 2
    W, xs, enc, grid_phi, X, y, phi, m, rng = _problem(N, T, d, chi, C, seed=7, ngrid=2001, cx=False)
    order = np.argsort(y, kind="stable")
    eng = engine_cls(0)
    try:
        x1, e1, _ = eng.impute_model(W, phi, y, m, xs, grid_phi, 0, True)
        eng.set_options(chi_max=chi)
        eng.set_dataset(1, phi[order], y[order], C)
        eng.set_mps(W)
        x2, e2, _ = eng.impute(1, m[order], xs, grid_phi, 0, True)
    finally:
        eng.close()
    assert np.array_equal(x1[order], x2) and np.array_equal(e1[order], e2)
    _check(W, xs, grid_phi, phi, y, m, x1, e1, "median")


@pytest.mark.parametrize("cx", [False, True], ids=["real", "complex"])
def test_fp32_chain_arithmetic(engine_cls, cx):
    """compute = fp32: the imputed values are the fp64 ones up to the grid value the density's quantile falls on - the
    fp32 rounding of the chain (relative 1e-6 on rho) moves a quantile by much less than a grid step of 1e-3, so almost every
    site agrees exactly and the rest by one step (before the difference propagates through the conditioning)."""
    N, T, d, chi, C = 24, 14, 4, 10, 2
    W, xs, enc, grid_phi, X, y, phi, m, rng = _problem(N, T, d, chi, C, seed=31, ngrid=2001, cx=cx)
    eng = engine_cls(0)
    try:
        x64, e64, _ = eng.impute_model(W, phi, y, m, xs, grid_phi, 0, True, compute="f64")
        x32, e32, _ = eng.impute_model(W, phi, y, m, xs, grid_phi, 0, True, compute="f32")
        xm64, s64, _ = eng.impute_model(W, phi, y, m, xs, grid_phi, 3, True, compute="f64")
        xm32, s32, _ = eng.impute_model(W, phi, y, m, xs, grid_phi, 3, True, compute="f32")
    finally:
        eng.close()
    _check(W, xs, grid_phi, phi, y, m, x64, e64, "median")
    mask = m.astype(bool)
    dx = xs[1] - xs[0]
    diff = np.abs(x32 - x64)[mask]
    assert np.mean(diff == 0.0) > 0.9 and np.quantile(diff, 0.99) <= 3 * dx and diff.max() < 0.05
    assert np.abs(xm32 - xm64)[mask].max() < 2e-4 and np.abs(s32 - s64)[mask].max() < 2e-4
    assert not np.array_equal(xm32, xm64)                      # it really is a different arithmetic


def test_model_run_error_paths(engine_cls):
    import mpstime_jl_amd as mt
    N, T, d, chi, C = 4, 6, 3, 4, 2
    W, xs, enc, grid_phi, X, y, phi, m, rng = _problem(N, T, d, chi, C, seed=1, ngrid=101, cx=True)
    eng = engine_cls(0)
    try:
        with pytest.raises(mt.MPSTError, match="Fourier"):
            eng.impute_model(W, phi, y, m, xs, grid_phi, 3, True, mean_basis=1)
        bad = y.copy()
        bad[0] = C
        with pytest.raises(mt.MPSTError, match="label_idx"):
            eng.impute_model(W, phi, bad, m, xs, grid_phi, 0, True)
        with pytest.raises(mt.MPSTError, match="chi_max <= 128"):
            Wh = R.random_mps(10, 16, 200, 1, rng)
            eng.impute_model(Wh, R.legendre_encode(X[:, :1].repeat(10, 1), 16), np.zeros(N, dtype=np.int32), np.ones((N, 10), dtype=np.uint8),
                             xs, R.legendre_encode(xs, 16), 0, True)
    finally:
        eng.close()


def test_host_api_fourier_model_and_fp32(engine_cls):
    """init_imputation_problem / impute_dataset / MPS_impute with the reference's Fourier encoding (a complex MPS, as its
    legacy path trains) and with compute="f32"."""
    import mpstime_jl_amd as mt
    from mpstime_jl_amd.imputation import _scaled_instances
    rng = np.random.default_rng(17)
    T, d, chi, C, Ntr, Nte = 16, 5, 8, 2, 30, 10
    W = _complex_mps(T, d, chi, C, rng)
    opts = mt.MPSOptions(encoding="Fourier", d=d, chi_max=chi, verbosity=-1)
    assert opts.dtype == "ComplexF64"
    t = np.linspace(0, 1, T)
    ytr = np.sort(rng.integers(0, C, Ntr))
    Xtr = np.sin(2 * np.pi * (t[None, :] * (1 + ytr[:, None]) + rng.uniform(size=(Ntr, 1)))) + 0.1 * rng.normal(size=(Ntr, T))
    yte = rng.integers(0, C, Nte)
    Xte = np.sin(2 * np.pi * (t[None, :] * (1 + yte[:, None]) + rng.uniform(size=(Nte, 1)))) + 0.1 * rng.normal(size=(Nte, T))
    td = mt.EncodedTimeSeriesSet(None, ytr, ytr.astype(np.int32), Xtr, np.bincount(ytr, minlength=C))
    trained = mt.TrainedMPS(W, opts, td)
    imp = mt.init_imputation_problem(trained, Xte, yte, dx=1e-3, verbosity=0)
    assert np.iscomplexobj(imp.x_guess_range.xvals_enc)
    mask = rng.uniform(size=(Nte, T)) < 0.35
    ts, err = mt.impute_dataset(imp, mask, "median", invert_transform=False)
    enc, norms, raw, full, scaled, oob = _scaled_instances(imp, np.arange(Nte), mask)
    phi = enc.encode(scaled, d)
    classes = I.expand_label_index(W)
    nflip = 0
    for k in range(Nte):
        sites = np.flatnonzero(mask[k])
        if len(sites) == 0:
            continue
        xo, eo = I.impute(classes[yte[k]], phi[k], sites, imp.x_guess_range.xvals, imp.x_guess_range.xvals_enc, "median")
        diff = np.abs(ts[k, sites] - xo)
        if np.any(diff > 1e-12):
            first = int(np.argmax(diff > 1e-12))
            assert diff[first] <= 1.0000001e-3 and np.all(diff[:first] <= 1e-12)
            nflip += 1
        assert np.array_equal(ts[k][~mask[k]], scaled[k][~mask[k]])
    assert nflip <= 1
    ts32, _ = mt.impute_dataset(imp, mask, "median", invert_transform=False, compute="f32")
    assert np.mean(ts32 == ts) > 0.9 and np.abs(ts32 - ts).max() < 0.05
    ts_m, sd = mt.impute_dataset(imp, mask, "mean", impute_order="backwards")
    assert np.all(np.isfinite(ts_m)) and np.all(sd[mask] >= 0)
    cls = int(yte[0])
    inst = int(np.flatnonzero(np.flatnonzero(yte == cls) == 0)[0])
    t1, e1, target, metrics = mt.MPS_impute(imp, cls, inst, np.flatnonzero(mask[0]), "median", impute_order="backwards")
    assert np.all(np.isfinite(t1[0])) and "MAE" in metrics[0]


@pytest.mark.parametrize("cx,compute,chi,d", [(True, "f64", 50, 2), (False, "f64", 72, 3), (True, "f32", 72, 3), (False, "f64", 128, 2)],
                         ids=["complex_f64_chi50", "real_f64_chi72", "complex_f32_chi72", "real_f64_chi128"])
def test_bond_dimensions_beyond_the_lds_kernel(engine_cls, cx, compute, chi, d):
    """chi above 64 (complex fp64: 48): the environment pass works out of global scratch (k_imp_right_big)."""
    N, C = 6, 1
    T = {2: 16, 3: 12}[d]
    rng = np.random.default_rng(chi)
    W = _complex_mps(T, d, chi, C, rng) if cx else R.random_mps(T, d, chi, C, rng)
    assert max(t.shape[2] for t in W) == min(chi, d ** (T // 2))
    xs = -1.0 + (2.0 / 800) * np.arange(801)
    enc = (lambda x: R.fourier_encode(x, d)) if cx else (lambda x: R.legendre_encode(x, d))
    X = rng.uniform(-0.9, 0.9, (N, T))
    y = np.zeros(N, dtype=np.int32)
    m = (rng.uniform(size=(N, T)) < 0.5).astype(np.uint8)
    m[0] = 1
    eng = engine_cls(0)
    try:
        x_g, e_g, _ = eng.impute_model(W, enc(X), y, m, xs, enc(xs), 0, True, compute=compute)
        x_b, e_b, _ = eng.impute_model(W, enc(X), y, m, xs, enc(xs), 0, True, compute=compute, order=1)
    finally:
        eng.close()
    if compute == "f64":
        _check(W, xs, enc(xs), enc(X), y, m, x_g, e_g, "median", max_flips=1)
        _check(W, xs, enc(xs), enc(X), y, m, x_b, e_b, "median", "backwards", max_flips=1)
    else:
        classes = I.expand_label_index(W)
        same = []
        for i in range(N):
            sites = np.flatnonzero(m[i])
            xo, _ = I.impute(classes[0], enc(X)[i], sites, xs, enc(xs), "median")
            same.append(np.abs(x_g[i, sites] - xo) < 1e-12)
        assert np.mean(np.concatenate(same)) > 0.8


def test_config5_element_type_and_shape(engine_cls):
    """BASELINE configs[4]'s own element type x bond shape - complex Fourier model, d = 8, chi = 64, the reference's full
    20 001-point grid (imputation.jl:90-107) - at a size the NumPy restatement finishes in seconds.  fp64 on the device
    (the global-scratch environment kernel: chi > 48) must reproduce the oracle's grid values exactly; the fp32 chain
    arithmetic (k_imp_left<float, true, 1> / k_imp_right<float, true>, what `bench.py --workload impute` times) lands on the
    oracle's grid value at almost every site and within a few grid steps at all but the sites where an earlier one-step
    difference has changed the conditioning (random chains have flat, multi-modal conditionals: the worst case)."""
    N, T, d, chi, C = 16, 24, 8, 64, 1
    W, xs, enc, grid_phi, X, y, phi, m, rng = _problem(N, T, d, chi, C, seed=77, ngrid=20001, cx=True)
    eng = engine_cls(0)
    try:
        x64, e64, _ = eng.impute_model(W, phi, y, m, xs, grid_phi, 0, True, compute="f64")
        x32, e32, _ = eng.impute_model(W, phi, y, m, xs, grid_phi, 0, True, compute="f32")
    finally:
        eng.close()
    _check(W, xs, grid_phi, phi, y, m, x64, e64, "median", max_flips=3)
    classes = I.expand_label_index(W)
    dx = xs[1] - xs[0]
    same = tot = first_same = first_tot = 0
    for i in range(N):
        sites = np.flatnonzero(m[i])
        if len(sites) == 0:
            continue
        xo, _ = I.impute(classes[y[i]], phi[i], sites, xs, grid_phi, "median", "forwards", True, None)
        diff = np.abs(x32[i, sites] - xo)
        same += int(np.sum(diff <= 1e-12))
        tot += len(sites)
        # the first imputed site of an instance is conditioned on known values only: no propagated difference there
        first_same += int(diff[0] <= 3 * dx)
        first_tot += 1
    assert first_same == first_tot, (first_same, first_tot)
    assert same >= 0.85 * tot, (same, tot)


@pytest.mark.parametrize("basis", ["Stoudenmire", "Sahand", "Uniform"])
def test_mean_with_the_other_closed_form_bases(engine_cls, basis):
    """impute_mean re-conditions on the encoding of E[x] (sampling_utils.jl:66-96), which the device evaluates from the
    basis' closed form: angle_encode (bases.jl:7-21), sahand_encode (:45-68), uniform_encode (:2-4) next to Legendre and
    Fourier.  Mean and standard deviation against the NumPy restatement with the host encoder as `encode`."""
    import mpstime_jl_amd as mt
    cx = basis != "Uniform"
    d = {"Stoudenmire": 2, "Sahand": 4, "Uniform": 3}[basis]
    code = {"Stoudenmire": 3, "Sahand": 4, "Uniform": 5}[basis]
    fn = {"Stoudenmire": mt.angle_encode, "Sahand": mt.sahand_encode, "Uniform": mt.uniform_encode}[basis]
    enc = lambda x: fn(np.asarray(x, dtype=np.float64), d)
    N, T, chi, C, ngrid = 9, 8, 5, 2, 1001
    rng = np.random.default_rng(17)
    W = _complex_mps(T, d, chi, C, rng) if cx else R.random_mps(T, d, chi, C, rng)
    xs = np.arange(ngrid) / (ngrid - 1.0)                     # these bases live on [0, 1]
    X = rng.uniform(0.03, 0.97, (N, T))
    y = rng.integers(0, C, N).astype(np.int32)
    m = (rng.uniform(size=(N, T)) < 0.4).astype(np.uint8)
    m[0] = 1
    m[1] = 0
    grid_phi, phi = enc(xs), enc(X)
    eng = engine_cls(0)
    try:
        x_mean, e_mean, _ = eng.impute_model(W, phi, y, m, xs, grid_phi, 3, True, mean_basis=code)
        if basis == "Stoudenmire":
            with pytest.raises(mt.MPSTError, match="Stoudenmire"):
                eng.impute_model(W, phi, y, m, xs, grid_phi, 3, True, mean_basis=0)       # a real basis for a complex model
    finally:
        eng.close()
    classes = I.expand_label_index(W)
    for i in range(N):
        sites = np.flatnonzero(m[i])
        if len(sites) == 0:
            continue
        xo, eo = I.impute(classes[y[i]], phi[i], sites, xs, grid_phi, "mean", "forwards", True, None, encode=lambda v: enc(v))
        assert np.abs(x_mean[i, sites] - xo).max() < 1e-9 and np.abs(e_mean[i, sites] - eo).max() < 1e-9


def test_host_api_sahand_model_mean(engine_cls):
    """impute_dataset(..., "mean") through the host mirror for a Sahand-encoded (complex) model: the basis code the host
    passes down selects the device's closed form of sahand_encode for the re-conditioning state."""
    import mpstime_jl_amd as mt
    from mpstime_jl_amd.imputation import _scaled_instances
    rng = np.random.default_rng(23)
    T, d, chi, C, Ntr, Nte = 12, 4, 6, 2, 24, 8
    W = _complex_mps(T, d, chi, C, rng)
    opts = mt.MPSOptions(encoding="Sahand", d=d, chi_max=chi, verbosity=-1)
    t = np.linspace(0, 1, T)
    ytr = np.sort(rng.integers(0, C, Ntr))
    Xtr = np.sin(2 * np.pi * (t[None, :] * (1 + ytr[:, None]) + rng.uniform(size=(Ntr, 1)))) + 0.1 * rng.normal(size=(Ntr, T))
    yte = rng.integers(0, C, Nte)
    Xte = np.sin(2 * np.pi * (t[None, :] * (1 + yte[:, None]) + rng.uniform(size=(Nte, 1)))) + 0.1 * rng.normal(size=(Nte, T))
    td = mt.EncodedTimeSeriesSet(None, ytr, ytr.astype(np.int32), Xtr, np.bincount(ytr, minlength=C))
    imp = mt.init_imputation_problem(mt.TrainedMPS(W, opts, td), Xte, yte, dx=2e-3, verbosity=0)
    mask = rng.uniform(size=(Nte, T)) < 0.35
    ts, sd = mt.impute_dataset(imp, mask, "mean", invert_transform=False)
    enc, norms, raw, full, scaled, oob = _scaled_instances(imp, np.arange(Nte), mask)
    phi = enc.encode(scaled, d)
    classes = I.expand_label_index(W)
    xs, gphi = imp.x_guess_range.xvals, imp.x_guess_range.xvals_enc
    for k in range(Nte):
        sites = np.flatnonzero(mask[k])
        if len(sites) == 0:
            continue
        xo, eo = I.impute(classes[yte[k]], phi[k], sites, xs, gphi, "mean", "forwards", True, None, encode=lambda v: enc.encode(np.asarray(v), d))
        assert np.abs(ts[k, sites] - xo).max() < 1e-9 and np.abs(sd[k, sites] - eo).max() < 1e-9


@pytest.mark.parametrize("cx", [False, True], ids=["legendre", "fourier"])
def test_closed_form_on_a_partial_grid(engine_cls, cx, monkeypatch):
    """The closed forms do not assume the grid spans the basis' whole range: a uniform grid over [-0.7, 0.85] (1 501 values),
    d = 6 (Legendre) / d = 5 (Fourier), median + WMAD and inverse-transform sampling, against the table path."""
    N, T, chi, C = 10, 12, 7, 1
    d = 5 if cx else 6
    rng = np.random.default_rng(3)
    W = _complex_mps(T, d, chi, C, rng) if cx else R.random_mps(T, d, chi, C, rng)
    xs = -0.7 + (1.55 / 1500) * np.arange(1501)
    enc = (lambda x: R.fourier_encode(x, d)) if cx else (lambda x: R.legendre_encode(x, d))
    X = rng.uniform(-0.6, 0.8, (N, T))
    y = np.zeros(N, dtype=np.int32)
    m = (rng.uniform(size=(N, T)) < 0.5).astype(np.uint8)
    u = rng.uniform(0.05, 0.95, (N, T, 1))
    phi, grid_phi = enc(X), enc(xs)
    eng = engine_cls(0)
    try:
        a = [eng.impute_model(W, phi, y, m, xs, grid_phi, 0, True)[:2], eng.impute_model(W, phi, y, m, xs, grid_phi, 2, False, u)[:2]]
        assert eng.impute_info()["closed_form_densities"]
        monkeypatch.setenv("MPST_IMP_NO_TRIG", "1")
        b = [eng.impute_model(W, phi, y, m, xs, grid_phi, 0, True)[:2], eng.impute_model(W, phi, y, m, xs, grid_phi, 2, False, u)[:2]]
        assert not eng.impute_info()["closed_form_densities"]
    finally:
        eng.close()
    dx = xs[1] - xs[0]
    for (xa, ea), (xb, eb) in zip(a, b):
        bad = [i for i in range(N) if not np.array_equal(xa[i], xb[i])]
        assert len(bad) <= 1, bad
        for i in bad:
            first = int(np.argmax(xa[i] != xb[i]))
            assert abs(xa[i, first] - xb[i, first]) <= dx * 1.0000001


@pytest.mark.parametrize("d,ngrid", [(12, 101), (16, 101), (16, 41)])
def test_coarse_legendre_grids_take_the_table_path(engine_cls, d, ngrid):
    """The closed-form prefix sums of the Legendre densities are an Euler-Maclaurin expansion truncated after the h^3 term; on a coarse grid
    the next term moves the cumulative trapezoid by whole grid steps (relative cdf error 5e-5 at d = 12 / 101 points, 6e-4 at d = 16 /
    101, 8e-2 at d = 16 / 41).  Such grids must be recognised and served by the table path: median + WMAD against the NumPy restatement
    of the reference's cumulative trapezoid (src/Imputation/sampling_utils.jl:162-199)."""
    N, T, chi, C = 8, 10, 6, 1
    W, xs, enc, grid_phi, X, y, phi, m, rng = _problem(N, T, d, chi, C, seed=100 + d + ngrid, ngrid=ngrid, cx=False)
    eng = engine_cls(0)
    try:
        x_g, e_g, _ = eng.impute_model(W, phi, y, m, xs, grid_phi, 0, True)
        assert not eng.impute_info()["closed_form_densities"]
    finally:
        eng.close()
    _check(W, xs, grid_phi, phi, y, m, x_g, e_g, "median")


@pytest.mark.parametrize("order", [0, 1], ids=["forwards", "backwards"])
@pytest.mark.parametrize("compute", ["f64", "f32"])
@pytest.mark.parametrize("cx,d,chi", [(True, 8, 20), (True, 12, 9), (False, 12, 16), (False, 4, 33)],
                         ids=["fourier_d8", "fourier_d12", "legendre_d12", "legendre_d4_chi33"])
def test_batched_sweep_equals_the_one_instance_kernel(engine_cls, compute, cx, d, chi, order, monkeypatch):
    """k_imp_leftb (sixteen instances per workgroup: the products with the site tensor and the environment on the matrix pipe, a wave
    per instance for the selections) against k_imp_left<.., TRIG> (a workgroup per instance, MPST_IMP_NO_BATCH=1): 37 instances = two
    full workgroups and a partial one, ragged missing patterns (one instance without a missing site, one with all of them), three
    classes at the label site, every method, both orders.  fp64 chain: the same grid values up to a cumulative sum that lands within
    rounding of a threshold; fp32 chain: the two kernels add their products in different orders, so one grid step."""
    N, T, C = 37, 14, 3
    W, xs, enc, grid_phi, X, y, phi, m, rng = _problem(N, T, d, chi, C, seed=1000 + d + chi, ngrid=20001, cx=cx)
    u = rng.uniform(0.02, 0.98, (N, T, 3))
    eng = engine_cls(0)
    try:
        runs = {}
        for tag in ("batched", "single"):
            if tag == "single":
                monkeypatch.setenv("MPST_IMP_NO_BATCH", "1")
            runs[tag] = [eng.impute_model(W, phi, y, m, xs, grid_phi, 0, True, order=order, compute=compute)[:2],
                         eng.impute_model(W, phi, y, m, xs, grid_phi, 1, False, order=order, compute=compute)[:2],
                         eng.impute_model(W, phi, y, m, xs, grid_phi, 2, False, u[:, :, :1], order=order, compute=compute)[:2],
                         eng.impute_model(W, phi, y, m, xs, grid_phi, 3, True, order=order, compute=compute)[:2],
                         eng.impute_model(W, phi, y, m, xs, grid_phi, 4, True, u, max_trials=3, rejection_threshold=1.0, order=order,
                                          compute=compute)[:2]]
            info = eng.impute_info()
            assert info["closed_form_densities"] and info["batched_sweep"] == (tag == "batched")
    finally:
        eng.close()
    mask = m.astype(bool)
    dx = xs[1] - xs[0]
    f64 = compute == "f64"
    for k, ((xa, ea), (xb, eb)) in enumerate(zip(runs["batched"], runs["single"])):
        assert np.all(xa[~mask] == 0.0) and np.all(np.isfinite(xa))
        if k == 3:          # mean: sums over the grid in another order
            tol = 1e-10 if f64 else 2e-4
            assert np.abs(xa - xb)[mask].max() < tol and np.abs(ea - eb)[mask].max() < tol
            continue
        bad = [i for i in range(N) if not np.array_equal(xa[i], xb[i])]
        if f64:
            assert len(bad) <= 2, (k, bad)
            for i in bad:
                first = int(np.argmax(xa[i] != xb[i]))
                assert abs(xa[i, first] - xb[i, first]) <= dx * 1.0000001
        else:
            # (the first site that differs: by a few of the 20 001 grid steps - fp32 products of up to 33 terms in two orders, a
            # density as flat as a random model's; the conditioning carries the difference on)
            for i in bad:
                first = int(np.argmax(xa[i] != xb[i]))
                assert abs(xa[i, first] - xb[i, first]) <= 10 * dx * 1.0000001, (k, i, xa[i, first], xb[i, first])
        good = [i for i in range(N) if i not in bad]
        if ea is not None and k in (0, 4) and f64:
            assert np.abs(ea[good] - eb[good]).max() <= dx * 1.0000001
