"""Device-side preprocessing + encoding (mpst_encode_dataset) against the host restatement of
src/utils.jl:161-275 and src/Encodings/bases.jl:70-108."""
import numpy as np
import pytest

import mpstime_jl_amd as mt
from oracle import ref_numpy as R

pytestmark = pytest.mark.gpu


def _data(N, T, seed, spread=1.0):
    rng = np.random.default_rng(seed)
    X = spread * rng.standard_normal((N, T)).cumsum(axis=1) / 3.0
    y = np.sort(rng.integers(0, 3, N))
    y[:3] = 0
    y[-3:] = 2
    return X, np.sort(y)


@pytest.mark.parametrize("basis,d", [("Legendre_No_Norm", 4), ("Legendre", 7), ("Legendre_Norm", 5)])
@pytest.mark.parametrize("sig,mm", [(True, True), (False, True), (True, False)])
def test_train_encoding_matches_oracle(engine_cls, basis, d, sig, mm):
    X, y = _data(97, 41, 5)
    if not mm:
        X = 1.0 / (1.0 + np.exp(-X)) if not sig else X      # without minmax the data must already sit in [0, 1]
    eng = engine_cls(0)
    norms, sec = eng.encode_dataset(0, X, y, 3, basis=basis, d=d, sigmoid_transform=sig, minmax=mm)
    phi = eng.get_encoded(0)
    Xs, norms_o = R.transform_train_data(X, sigmoid_transform=sig, minmax=mm)
    ref = R.legendre_encode(Xs, d, norm=(basis == "Legendre_Norm"))
    assert phi.shape == ref.shape
    assert np.max(np.abs(phi - ref)) < 1e-13
    if mm:
        assert abs(norms.minmax[0] - norms_o[1][0]) < 1e-15 and abs(norms.minmax[1] - norms_o[1][1]) < 1e-15
    assert sec > 0
    eng.close()


def test_test_set_out_of_bounds_rescale_matches_oracle(engine_cls):
    Xtr, ytr = _data(64, 30, 1)
    Xte, yte = _data(50, 30, 2, spread=2.5)       # wider than the training data: some series leave [0, 1]
    eng = engine_cls(0)
    norms, _ = eng.encode_dataset(0, Xtr, ytr, 3, d=4)
    oob, _ = eng.encode_dataset(1, Xte, yte, 3, d=4, norms=norms)
    phi = eng.get_encoded(1)
    _, norms_o = R.transform_train_data(Xtr)
    Xs, oob_o = R.transform_test_data(Xte, norms_o)
    ref = R.legendre_encode(Xs, 4)
    assert np.max(np.abs(phi - ref)) < 1e-13
    assert len(oob) == len(oob_o) and len(oob) > 0
    for a, b in zip(oob, oob_o):
        assert a[0] == b[0] and abs(a[1] - b[1]) < 1e-14 and abs(a[2] - b[2]) < 1e-14
    eng.close()


def test_encoded_on_device_trains_like_uploaded(engine_cls):
    """A sweep on device-encoded data equals the sweep on the same values uploaded through mpst_set_dataset."""
    X, y = _data(80, 12, 9)
    opts = dict(chi_max=6, eta=0.05)
    W = R.random_mps(12, 4, 3, 3, np.random.default_rng(3))
    eng = engine_cls(0)
    eng.encode_dataset(0, X, y, 3, d=4)
    phi = eng.get_encoded(0)
    eng.set_options(**opts); eng.set_mps(W); eng.build_caches(); eng.sweep()
    a = eng.eval(0)
    eng.close()
    eng = engine_cls(0)
    eng.set_options(**opts); eng.set_dataset(0, phi, y, 3); eng.set_mps(W); eng.build_caches(); eng.sweep()
    b = eng.eval(0)
    eng.close()
    assert a[:3] == b[:3]


def test_unsupported_basis_is_reported(engine_cls):
    eng = engine_cls(0)
    with pytest.raises(mt.MPSTError):
        eng.encode_dataset(0, np.zeros((4, 5)), np.zeros(4, dtype=np.int32), 1, basis="Fourier", d=4)
    eng.close()


def test_fitmps_with_device_encoding_matches_host_encoding():
    """fitMPS(device_encode=True): same encoded sets (1e-13) and, from them, the same first logged losses."""
    rng = np.random.default_rng(11)
    Xtr, _ = mt.trendy_sine(24, 60, sigma=0.1, rng=rng)
    ytr = np.arange(60) % 2
    Xte, _ = mt.trendy_sine(24, 20, sigma=0.1, rng=rng)
    yte = np.arange(20) % 2
    opts = mt.MPSOptions(d=4, chi_max=8, nsweeps=1, verbosity=-1)
    a, info_a, te_a = mt.fitMPS(Xtr, ytr, Xte, yte, opts)
    b, info_b, te_b = mt.fitMPS(Xtr, ytr, Xte, yte, opts, device_encode=True)
    assert np.max(np.abs(a.train_data.phi - b.train_data.phi)) < 1e-13
    assert np.max(np.abs(te_a.phi - te_b.phi)) < 1e-13
    assert np.array_equal(a.train_data.label_index, b.train_data.label_index)
    assert np.array_equal(a.train_data.original_data, b.train_data.original_data)
    assert abs(info_a["train_KL_div"][0] - info_b["train_KL_div"][0]) < 1e-10
    assert abs(info_a["test_KL_div"][0] - info_b["test_KL_div"][0]) < 1e-10


@pytest.mark.parametrize("shape", [(97, 41), (64, 30), (1, 2), (3, 3), (500, 100)])
def test_robust_sigmoid_fit_on_the_device(engine_cls, shape):
    """Normalization.fit(RobustSigmoid, X_train) (utils.jl:174): median and inter-quartile range of ALL training values
    from a device radix sort; type-7 quantiles written as NumPy's _lerp, so the host restatement is matched to the bit
    (odd and even counts, the tiny cases where the quartiles interpolate between the only two values)."""
    N, T = shape
    rng = np.random.default_rng(N * 1000 + T)
    X = rng.normal(size=(N, T)) * 3.0 + 0.5
    y = np.zeros(N, dtype=np.int32)
    eng = engine_cls(0)
    try:
        norms, sec = eng.encode_dataset(0, X, y, 1, d=3)
        med = float(np.median(X))
        q75, q25 = np.percentile(X, [75.0, 25.0])
        assert norms.sigmoid[0] == med and norms.sigmoid[1] == float(q75 - q25)
        # a fit handed in (e.g. computed over all shards) takes the other branch and gives the same encoding
        phi = eng.get_encoded(0)
        norms2, _ = eng.encode_dataset(0, X, y, 1, d=3, sigmoid_fit=norms.sigmoid)
        assert norms2.sigmoid == norms.sigmoid and np.array_equal(eng.get_encoded(0), phi)
        with pytest.raises(mt.MPSTError, match="iqr"):
            eng.encode_dataset(0, np.ones((4, 5)), np.zeros(4, dtype=np.int32), 1, d=3)      # constant data: iqr = 0
    finally:
        eng.close()


@pytest.mark.parametrize("d", [2, 5, 8])
def test_fourier_and_legendre_values_on_the_device(engine_cls, d):
    """mpst_encode_values: fourier_encode (bases.jl:23-42: cispi(f x) / sqrt(d), f = 0, 1, -1, 2, -2, ...) and the Legendre
    bases on the device without a data set - the states mpst_impute_model_run takes - against the host restatement."""
    rng = np.random.default_rng(d)
    X = rng.uniform(-1.0, 1.0, (37, 23))
    eng = engine_cls(0)
    try:
        phi_f, sec = eng.encode_values(X, "Fourier", d)
        phi_l, _ = eng.encode_values(X, "Legendre_No_Norm", d)
        phi_n, _ = eng.encode_values(X, "Legendre_Norm", d)
        # with the preprocessing of a training set in front (RobustSigmoid fitted on the device, MinMax)
        Xr = rng.normal(size=(37, 23)) * 2.0
        phi_p, _ = eng.encode_values(Xr, "Fourier", d, sigmoid_transform=True, minmax=True)
    finally:
        eng.close()
    assert phi_f.dtype == np.complex128 and sec > 0
    assert np.abs(phi_f - R.fourier_encode(X, d)).max() < 1e-14
    assert np.abs(phi_l - R.legendre_encode(X, d)).max() < 1e-13
    assert np.abs(phi_n - R.legendre_encode(X, d, norm=True)).max() < 1e-13
    Xs, _ = R.transform_train_data(Xr, sigmoid_transform=True, minmax=True)
    assert np.abs(phi_p - R.fourier_encode(Xs, d)).max() < 1e-12


def test_encode_values_test_set_path_equals_the_data_set_path(engine_cls):
    """mpst_encode_values with a training fit handed in (a test set: train-fitted sigmoid / min-max) gives the states
    mpst_encode_dataset stores for the same data (out-of-bounds rescale off in both)."""
    Xtr, ytr = _data(64, 30, 1)
    Xte, yte = _data(40, 30, 2, spread=1.2)
    eng = engine_cls(0)
    try:
        norms, _ = eng.encode_dataset(0, Xtr, ytr, 3, d=4)
        eng.encode_dataset(1, Xte, yte, 3, d=4, norms=norms, rescale_out_of_bounds=False)
        ref = eng.get_encoded(1)
        phi, _ = eng.encode_values(Xte, "Legendre_No_Norm", 4, norms=norms, minmax=True)
        assert np.array_equal(phi, ref)
        # the scaled values themselves from the second Legendre state (sqrt(3/2) x), then Fourier on the host
        xs = ref[:, :, 1] / np.sqrt(1.5)
        phi_f, _ = eng.encode_values(Xte, "Fourier", 4, norms=norms, minmax=True)
        assert np.abs(phi_f - R.fourier_encode(xs, 4)).max() < 1e-12
    finally:
        eng.close()



@pytest.mark.parametrize("basis,d", [("Stoudenmire", 2), ("Sahand", 4), ("Sahand", 8), ("Uniform", 3)])
def test_the_other_closed_form_bases_on_the_device(engine_cls, basis, d):
    """angle_encode (bases.jl:7-21), sahand_encode (:45-68), uniform_encode (:2-4) through mpst_encode_values against the host
    encoders - raw values in the bases' [0, 1] domain, and behind the training-set preprocessing."""
    import mpstime_jl_amd as mt
    fn = {"Stoudenmire": mt.angle_encode, "Sahand": mt.sahand_encode, "Uniform": mt.uniform_encode}[basis]
    rng = np.random.default_rng(d + len(basis))
    X = rng.uniform(0.0, 1.0, (29, 17))
    X[0, :4] = [0.0, 1.0, 0.5, 0.25]                          # interval boundaries of the Sahand basis
    Xr = rng.normal(size=(29, 17)) * 2.0
    eng = engine_cls(0)
    try:
        phi, _ = eng.encode_values(X, basis, d)
        phi_p, _ = eng.encode_values(Xr, basis, d, sigmoid_transform=True, minmax=True)
        if basis == "Stoudenmire":
            with pytest.raises(mt.MPSTError, match="d = 2"):
                eng.encode_values(X, basis, 3)
        if basis == "Sahand":
            with pytest.raises(mt.MPSTError, match="even"):
                eng.encode_values(X, basis, 3)
    finally:
        eng.close()
    ref = fn(X, d)
    assert phi.dtype == ref.dtype and np.abs(phi - ref).max() < 1e-14
    opts = mt.MPSOptions(d=d, encoding=basis, verbosity=-1)
    Xs, _ = mt.transform_train_data(Xr, opts, mt.model_encoding(basis).range)
    assert np.abs(phi_p - fn(Xs, d)).max() < 1e-12
