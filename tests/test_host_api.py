"""Host-side mirror of the reference API: MPSOptions defaults and validation, encodings,
preprocessing, data-set encoding, starting MPS, batch sharding (all CPU)."""
import math

import numpy as np
import pytest

import mpstime_jl_amd as mt
from mpstime_jl_amd.options import engine_options


def test_mpsoptions_defaults_match_reference():
    # src/Structs/options.jl:106-143
    o = mt.MPSOptions()
    ref = dict(verbosity=1, nsweeps=10, chi_max=25, eta=0.01, d=5, encoding="Legendre_No_Norm", projected_basis=False,
               aux_basis_dim=2, cutoff=1e-10, update_iters=1, dtype="Float64", loss_grad="KLD", bbopt="TSGO",
               track_cost=False, rescale=(False, True), train_classes_separately=False,
               encode_classes_separately=False, return_encoding_meta_info=False, minmax=True, exit_early=False,
               sigmoid_transform=True, init_rng=1234, chi_init=4, log_level=3, data_bounds=(0.0, 1.0),
               use_legacy_ITensor=False, svd_alg="divide_and_conquer")
    assert o.asdict() == ref and len(ref) == 27
    assert mt.MPSOptions(encoding=":Fourier").dtype == "ComplexF64"       # options.jl:117
    assert mt.MPSOptions().set(chi_max=7).chi_max == 7                    # _set_options


def test_engine_option_resolution_and_rejections():
    e = engine_options(mt.MPSOptions(loss_grad=":MSE", bbopt=":GD"))
    assert e["loss"] == "MSE" and e["bbopt"] == "GD"
    with pytest.raises(RuntimeError, match="Optim/OptimKit based solvers currently unimplemented"):
        engine_options(mt.MPSOptions(bbopt="Optim"))                      # loss_functions.jl:166-170
    with pytest.raises(RuntimeError, match="legacy"):
        engine_options(mt.MPSOptions(loss_grad="Mixed"))
    # complex encodings and reduced precision are accepted (element-typed kernels); unknown element types are not
    from mpstime_jl_amd import options as O
    assert engine_options(mt.MPSOptions(encoding="Fourier"))["loss"] == "KLD"
    assert O.numpy_dtype(mt.MPSOptions(encoding="Fourier").dtype) == np.complex128
    assert O.numpy_dtype("Float32") == np.float32 and O.numpy_dtype("ComplexF32") == np.complex64
    with pytest.raises(ValueError, match="dtype"):
        engine_options(mt.MPSOptions(dtype="BigFloat"))
    with pytest.raises(ValueError):
        mt.MPSOptions(encoding="nonsense")


@pytest.mark.parametrize("name", ["Legendre", "Legendre_No_Norm", "Legendre_Norm", "Fourier", "Stoudenmire", "Sahand",
                                  "Uniform"])
def test_model_symbolic_encoding_round_trip(name):
    # test/basis_tests.jl:3-8
    enc = mt.model_encoding(name)
    assert mt.model_encoding(mt.symbolic_encoding(enc)).name == enc.name


def test_legendre_basis_is_orthonormal_and_matches_closed_forms():
    # bases.jl:77-92: phi_k = sqrt((2k+1)/2) P_k;  Gauss-Legendre quadrature is exact here
    xs, ws = np.polynomial.legendre.leggauss(16)
    phi = mt.legendre_encode_no_norm(xs, 6)
    gram = (phi * ws[:, None]).T @ phi
    assert np.allclose(gram, np.eye(6), atol=1e-13)
    x = np.array([-0.7, 0.0, 0.3, 1.0])
    assert np.allclose(mt.legendre_encode_no_norm(x, 3)[:, 2], math.sqrt(2.5) * 0.5 * (3 * x ** 2 - 1))
    assert np.allclose(mt.legendre_encode(x, 4), mt.legendre_encode_no_norm(x, 4) / math.sqrt(math.sqrt(4.5) * 4))


def test_fourier_basis():
    assert mt.get_fourier_freqs(5) == [0, 1, -1, 2, -2] and mt.get_fourier_freqs(4) == [0, 1, -1, 2]
    f = mt.fourier_encode(np.array([0.25]), 3)[0]
    assert np.allclose(f, np.array([1, np.exp(0.25j * np.pi), np.exp(-0.25j * np.pi)]) / math.sqrt(3))
    assert abs(np.vdot(f, f) - 1) < 1e-14


def test_transform_data_ranges_and_out_of_bounds_rescale():
    rng = np.random.default_rng(0)
    Xtr = rng.normal(size=(30, 20))
    Xte = np.concatenate([rng.normal(size=(5, 20)), 10 * rng.normal(size=(2, 20))])   # two series far out of range
    opts = mt.MPSOptions()
    enc = mt.model_encoding("Legendre")
    a, b, norms, oob = mt.transform_data(Xtr, Xte, opts, enc.range)
    assert a.min() == -1.0 and a.max() == 1.0                  # MinMax over the whole training matrix
    assert b.min() >= -1.0 - 1e-12 and b.max() <= 1.0 + 1e-12  # utils.jl:243-266
    assert len(oob) >= 1
    med, iqr = norms.sigmoid
    assert med == np.median(Xtr) and iqr > 0
    # data_bounds shrink the interval (utils.jl:186-191)
    a2, _, _, _ = mt.transform_data(Xtr, Xte[:0], mt.MPSOptions(data_bounds=(0.1, 0.9)), enc.range)
    assert abs(a2.min() + 0.8) < 1e-12 and abs(a2.max() - 0.8) < 1e-12


def test_encode_dataset_sorts_stably_and_counts_classes():
    X = np.linspace(-1, 1, 6 * 4).reshape(6, 4)
    y = np.array([3, 1, 3, 1, 2, 1])
    enc = mt.model_encoding("Legendre")
    ets = mt.encode_dataset(X, X, y, enc, 3, {1: 0, 2: 1, 3: 2})
    assert list(ets.labels) == [1, 1, 1, 2, 3, 3]
    assert np.array_equal(ets.original_data, X[[1, 3, 5, 4, 0, 2]])          # stable sortperm (encodings.jl:43)
    assert list(ets.label_index) == [0, 0, 0, 1, 2, 2] and list(ets.class_distribution) == [3, 1, 2]
    assert ets.phi.shape == (6, 4, 3)
    with pytest.raises(ValueError, match="rescaled between"):
        mt.encode_dataset(X * 2, X * 2, y, enc, 3, {1: 0, 2: 1, 3: 2})         # encodings.jl:115-119


def test_starting_mps_is_normalised_and_left_canonical():
    W = mt.generate_startingMPS(4, 9, 3, 2, init_rng=7)
    # bond dimensions are capped by d^j near the ends, as ITensors.random_mps does
    assert W[-1].shape == (3, 3, 1, 2) and W[0].shape == (1, 3, 3) and W[4].shape == (4, 3, 4)
    for t in W[:-1]:
        m = t.reshape(-1, t.shape[2])
        assert np.allclose(m.T @ m, np.eye(m.shape[1]), atol=1e-13)
    assert abs(np.linalg.norm(W[-1]) - 1) < 1e-14
    W2 = mt.generate_startingMPS(4, 9, 3, 2, init_rng=7)
    assert all(np.array_equal(a, b) for a, b in zip(W, W2))


def test_batch_shards_partition_every_class():
    X = np.random.default_rng(1).uniform(-1, 1, (23, 5))
    y = np.array([0] * 9 + [1] * 3 + [2] * 11)
    ets = mt.encode_dataset(X, X, y, mt.model_encoding("Legendre"), 2, {0: 0, 1: 1, 2: 2})
    for world in (1, 2, 3, 8):
        parts = [mt.split_encoded(ets, r, world) for r in range(world)]
        assert all(np.array_equal(g, [9, 3, 11]) for _, g in parts)
        assert sum(len(p) for p, _ in parts) == 23
        assert np.array_equal(sum(p.class_distribution for p, _ in parts), [9, 3, 11])
        for p, _ in parts:
            assert np.all(np.diff(p.label_index) >= 0)                          # shards stay class-sorted
        got = np.sort(np.concatenate([p.original_data[:, 0] for p, _ in parts]))
        assert np.array_equal(got, np.sort(ets.original_data[:, 0]))


def test_fitmps_input_validation_needs_no_gpu():
    X = np.random.default_rng(2).normal(size=(10, 6))
    with pytest.raises(AssertionError, match="training labels"):
        mt.fitMPS(X, np.zeros(9, dtype=int))
    with pytest.raises(ValueError, match="not present in the training set"):
        mt.fitMPS(X, np.zeros(10, dtype=int), X[:2], np.array([0, 5]))
    with pytest.raises(RuntimeError, match="complex valued encoding"):
        mt.fitMPS(X, np.zeros(10, dtype=int), opts=mt.MPSOptions(encoding="Fourier", dtype="Float64"))
    with pytest.raises(ValueError, match="Custom"):
        mt.fitMPS(X, np.zeros(10, dtype=int), custom_encoding=mt.model_encoding("Legendre"))


def test_shard_split_rejects_empty_shards_on_every_rank():
    """A rank without series would leave the collectives of the others hanging: the split refuses, identically on all ranks."""
    import numpy as np
    import pytest
    phi = np.zeros((3, 4, 2))
    ets = mt.EncodedTimeSeriesSet(phi, np.array([1, 1, 2]), np.array([0, 0, 1]), np.zeros((3, 0)), np.array([2, 1]))
    for r in range(8):
        with pytest.raises(ValueError, match="would hold no series"):
            mt.split_encoded(ets, r, 8)
    loc, gc = mt.split_encoded(ets, 1, 2)
    assert len(loc) >= 1 and list(gc) == [2, 1]


def test_per_sweep_loss_and_optimiser_and_save_load(tmp_path):
    """loss_grad / bbopt may be one value per sweep (RealRealHighDimension.jl:691-713); TrainedMPS round-trips through
    save_trained_mps / load_trained_mps (test/save_load.jl:17-24)."""
    import numpy as np
    import pytest
    from mpstime_jl_amd.options import engine_options
    o = mt.MPSOptions(nsweeps=3, loss_grad=["KLD", ":MSE", "KLD"], bbopt=("TSGO", "GD", "TSGO"))
    assert [engine_options(o, k)["loss"] for k in range(3)] == ["KLD", "MSE", "KLD"]
    assert [engine_options(o, k)["bbopt"] for k in range(3)] == ["TSGO", "GD", "TSGO"]
    with pytest.raises(AssertionError, match="length nsweeps"):
        mt.MPSOptions(nsweeps=2, loss_grad=["KLD"])
    W = mt.generate_startingMPS(3, 5, 2, 2, 0)
    rng = np.random.default_rng(0)
    td = mt.EncodedTimeSeriesSet(rng.uniform(size=(4, 5, 2)), np.array([0, 0, 1, 1]), np.array([0, 0, 1, 1], dtype=np.int32),
                                 rng.uniform(size=(4, 5)), np.array([2, 2]))
    tm = mt.TrainedMPS(W, o, td)
    mt.save_trained_mps(tmp_path / "m.npz", tm)
    back = mt.load_trained_mps(tmp_path / "m.npz")
    assert back == tm and back.opts == o and np.array_equal(back.train_data.phi, td.phi)


def test_imputation_host_helpers():
    """Host side of the imputation API: invert_test_transform undoes transform_test_data (utils.jl:202-334), the block
    MAR mechanism (missing_data_mechanisms.jl:140-147), kNN_impute (imputation.jl:215-254), the grid of candidate values."""
    import numpy as np
    from mpstime_jl_amd.encodings import model_encoding, transform_test_data, transform_train_data
    rng = np.random.default_rng(0)
    Xtr = rng.normal(0.3, 2.0, (40, 12))
    Xte = rng.normal(0.3, 2.5, (7, 12))
    Xte[2, 5], Xte[4, 1] = 40.0, -40.0           # beyond anything in the training data: these series need the out-of-bounds rescale
    opts = mt.MPSOptions(d=4, verbosity=-1)
    enc = model_encoding(opts.encoding)
    _, norms = transform_train_data(Xtr, opts, enc.range)
    Xs, oob = transform_test_data(Xte, norms, opts, enc.range)
    assert len(oob) > 0 and Xs.min() >= -1 and Xs.max() <= 1
    back = mt.invert_test_transform(Xs, oob, norms, opts, enc.range)
    ok = np.ones_like(Xte, dtype=bool)
    ok[2, 5] = ok[4, 1] = False                  # saturated by the sigmoid: not invertible to 1e-9, everything else is
    assert np.allclose(back[ok], Xte[ok], rtol=1e-7, atol=1e-7)
    xc, idx = mt.mar(np.arange(20.0), 0.25, np.random.default_rng(1))
    assert len(idx) == 5 and np.all(np.diff(idx) == 1) and np.all(np.isnan(xc[idx])) and np.isfinite(np.delete(xc, idx)).all()
    with pytest.raises(ValueError):
        mt.mar(np.arange(4.0), 1.5)
    W = mt.generate_startingMPS(3, 12, 4, 2, 0)
    ytr = np.array([0] * 20 + [1] * 20)
    Xs_tr, _ = transform_train_data(Xtr, opts, enc.range)
    td = mt.encode_dataset(Xtr, Xs_tr, ytr, enc, 4, {0: 0, 1: 1})
    imp = mt.init_imputation_problem(mt.TrainedMPS(W, opts, td), Xte, np.array([0, 1, 0, 1, 0, 1, 0]), dx=1e-3, verbosity=0)
    assert len(imp.x_guess_range.xvals) == 2001 and abs(imp.x_guess_range.xvals[-1] - 1.0) < 1e-12
    assert imp.x_guess_range.xvals_enc.shape == (2001, 4) and imp.class_map == {0: 0, 1: 1}
    nn = mt.kNN_impute(imp, 1, 0, [3, 4, 5], k=2)
    known = np.setdiff1d(np.arange(12), [3, 4, 5])
    d2 = np.mean((Xtr[20:][:, known] - Xte[1][known]) ** 2, axis=1)
    assert np.array_equal(nn[0], Xtr[20:][np.argmin(d2)]) and len(nn) == 2
    with pytest.raises(ValueError, match="Invalid method"):
        mt.impute_dataset(imp, np.zeros((7, 12), dtype=bool), "bogus")
