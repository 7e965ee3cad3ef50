"""Public API on the GPU: fitMPS from raw data, classify, error paths of the C ABI
(structural properties of test/classification.jl:22-24)."""
import numpy as np
import pytest

import mpstime_jl_amd as mt

pytestmark = pytest.mark.gpu


def _toy(n=60, T=20, seed=3):
    rng = np.random.default_rng(seed)
    X1, _ = mt.trendy_sine(T, n // 2, period=(12.0, 15.0), slope=[-3.0, 0.0, 3.0], sigma=0.1, rng=rng)
    X2, _ = mt.trendy_sine(T, n // 2, period=(16.0, 19.0), slope=[-3.0, 0.0, 3.0], sigma=0.1, rng=rng)
    X = np.concatenate([X1, X2])
    y = np.concatenate([np.zeros(n // 2, dtype=int), np.ones(n // 2, dtype=int)])
    p = rng.permutation(n)
    return X[p], y[p]


def test_fitmps_classify_structural_properties():
    Xtr, ytr = _toy(80, 20, 3)
    Xte, yte = _toy(30, 20, 4)
    opts = mt.MPSOptions(d=3, chi_max=8, nsweeps=3, eta=0.05, verbosity=-1)
    mps, info, test_states = mt.fitMPS(Xtr, ytr, Xte, yte, opts)
    mps2, info2, _ = mt.fitMPS(Xtr, ytr, opts=opts)
    # (b) training is identical with or without test data (classification.jl:23)
    assert mps == mps2 and info["train_KL_div"] == info2["train_KL_div"]
    assert "test_acc" not in info2
    # (a) classify(mps, test_states) == classify(mps, X_test)[sortperm(y_test)] (classification.jl:22)
    p_states = mt.classify(mps, test_states)
    p_raw = mt.classify(mps, Xte)
    assert np.array_equal(p_states, p_raw[np.argsort(yte, kind="stable")])
    assert np.mean(p_states == np.sort(yte)) == info["test_acc"][-1]
    assert info["train_KL_div"][-2] < info["train_KL_div"][0]
    # unsupervised overload: one class, label index of dimension 1 (RealRealHighDimension.jl:416)
    mps1, info1, _ = mt.fitMPS(Xtr, opts=opts.set(nsweeps=1))
    assert mps1.mps[-1].shape[3] == 1 and info1["train_acc"][-1] == 1.0


def test_abi_error_paths(engine_cls):
    e = engine_cls(0)
    try:
        with pytest.raises(mt.MPSTError, match="set_options"):
            e.build_caches()
        e.set_options(chi_max=4)
        phi = np.random.default_rng(0).uniform(-1, 1, (8, 4, 2))
        with pytest.raises(mt.MPSTError, match="sorted by class"):
            e.set_dataset(0, phi, np.array([0, 1, 0, 1, 0, 1, 0, 1]), 2)
        e.set_dataset(0, phi, np.array([0, 0, 0, 0, 1, 1, 1, 1]), 2)
        W = mt.generate_startingMPS(2, 4, 2, 2, 0)
        e.set_mps(W)
        e.build_caches()
        e.bond_step(2, True)
        e.build_caches()                        # label now on site 2: environments on both sides of it
        with pytest.raises(mt.MPSTError, match="no Loss_Grad_MSE method"):
            e.set_options(chi_max=4, loss="MSE", train_classes_separately=True)
        with pytest.raises(mt.MPSTError):
            e.set_options(chi_max=64)           # exceeds the capacity fixed by set_mps
    finally:
        e.close()


def test_rccl_single_rank_communicator_is_a_no_op(engine_cls):
    """The all-reduce leg of the sharded sweep (mpst_comm_init + ncclAllReduce on the engine's stream)
    with a 1-rank communicator must give the same bits as the engine without a communicator."""
    import ctypes as C
    from oracle import ref_numpy as R
    from tests.helpers import load_engine, make_problem
    ds, W = make_problem(96, 10, 3, 3, 2, seed=21, balanced=False)
    opts = R.SweepOptions(chi_max=6, eta=0.05)
    out = []
    for use_comm in (False, True):
        eng = engine_cls(0)
        load_engine(eng, ds, W, opts)
        if use_comm:
            uid = (C.c_uint8 * 128)()
            assert eng.lib.mpst_comm_unique_id(uid) == 0
            eng._chk(eng.lib.mpst_comm_init(eng.ctx, uid, 1, 0))
        eng.build_caches()
        eng.sweep()
        eng.sweep()
        out.append((eng.get_mps(), eng.eval(0)))
        eng.close()
    for a, b in zip(out[0][0], out[1][0]):
        assert np.array_equal(a, b)
    assert out[0][1][:3] == out[1][1][:3] and np.array_equal(out[0][1][3], out[1][1][3])


def test_exit_early_stops_at_perfect_training_accuracy():
    """exit_early (RealRealHighDimension.jl:847): the sweep loop ends as soon as the logged training accuracy is 1."""
    Xtr, ytr = _toy(40, 16, 5)
    opts = mt.MPSOptions(d=3, chi_max=8, nsweeps=12, eta=0.1, verbosity=-1, exit_early=True)
    mps, info, _ = mt.fitMPS(Xtr, ytr, opts=opts)
    accs = info["train_acc"]
    if 1.0 in accs[:-1]:
        first = accs.index(1.0)
        # entries: initial, one per executed sweep, final (after normalize!)
        assert len(accs) == first + 2 and len(accs) < opts.nsweeps + 2
    else:
        assert len(accs) == opts.nsweeps + 2
    # TrainedMPS equality compares options and tensors elementwise (src/Structs/operations.jl:4-36)
    mps_b, _, _ = mt.fitMPS(Xtr, ytr, opts=opts)
    assert mps == mps_b
    mps_c, _, _ = mt.fitMPS(Xtr, ytr, opts=opts.set(eta=0.05))
    assert mps != mps_c


def test_fitmps_with_per_sweep_loss_and_track_cost(capsys):
    """loss_grad / bbopt given per sweep (RealRealHighDimension.jl:691-713) and track_cost's printed losses."""
    Xtr, ytr = _toy(60, 12, 7)
    base = mt.MPSOptions(d=3, chi_max=6, nsweeps=2, eta=0.05, verbosity=-1)
    a, info_a, _ = mt.fitMPS(Xtr, ytr, opts=base)
    b, info_b, _ = mt.fitMPS(Xtr, ytr, opts=base.set(loss_grad=["KLD", "KLD"], bbopt=["TSGO", "TSGO"]))
    assert a.mps[0].shape == b.mps[0].shape and all(np.array_equal(x, y) for x, y in zip(a.mps, b.mps))
    c, info_c, _ = mt.fitMPS(Xtr, ytr, opts=base.set(loss_grad=["KLD", "MSE"], bbopt=["TSGO", "GD"]))
    assert info_c["train_KL_div"][1] == info_a["train_KL_div"][1]          # first sweep identical
    assert info_c["train_KL_div"][2] != info_a["train_KL_div"][2]          # second one used the other loss / optimiser
    capsys.readouterr()
    res, _, _ = mt.fitMPS(Xtr, ytr, opts=base.set(nsweeps=1, verbosity=1, track_cost=True))
    out = capsys.readouterr().out
    assert out.count("Loss before step 1:") == 2 * 11 and out.count("Loss at site") == 2 * 11
    assert "Loss at site 11*12:" in out and "Loss at site 1*2:" in out
    # the order of the reference's output (RealRealHighDimension.jl:729-811): the backward half's eleven bonds, the two mid-sweep lines,
    # the forward half's eleven bonds
    lines = [ln for ln in out.splitlines() if ln.startswith(("Loss at site", "Backward sweep finished", "Starting forward sweep"))]
    assert lines[0].startswith("Loss at site 11*12:") and lines[10].startswith("Loss at site 1*2:")
    assert lines[11] == "Backward sweep finished." and lines[12].startswith("Starting forward sweep: [1/1]")
    assert lines[13].startswith("Loss at site 1*2:") and lines[23].startswith("Loss at site 11*12:") and len(lines) == 24
    # the returned model still carries the encoded training set (not the loss trace), and survives save / load
    assert isinstance(res.train_data, mt.EncodedTimeSeriesSet) and res.train_data.phi.shape[0] == 60
    import tempfile, os
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "m.npz")
        mt.save_trained_mps(path, res)
        assert mt.load_trained_mps(path) == res
    # per-sweep optimiser names are printed per sweep
    mt.fitMPS(Xtr, ytr, opts=base.set(verbosity=0, bbopt=["TSGO", "GD"]))
    out = capsys.readouterr().out
    assert out.count('"TSGO" algorithm') == 1 and out.count('"GD" algorithm') == 1
