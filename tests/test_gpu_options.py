"""Option paths of apply_update / decomposeBT that the main parity cases leave at their defaults, and the
call-order / recovery rules of the C ABI.  Every case goes through the C ABI and is compared with the oracle.

  rescale = (before, after)   loss_functions.jl:109 (normalize!(BT_init)) and :177 (normalize!(BT_new))
  svd_alg = "recursive"       RealRealHighDimension.jl:756,798 -> the one-sided Jacobi solver inside a sweep
  chi_max > 32 with d*chi_max <= 128: more than 32 eigenpairs from the tridiagonal path
"""
import numpy as np
import pytest

import mpstime_jl_amd as mt
from oracle import ref_numpy as R
from tests.helpers import load_engine, make_problem

pytestmark = pytest.mark.gpu


@pytest.fixture()
def eng(engine_cls):
    e = engine_cls(0)
    yield e
    e.close()


def _sweep_bond_by_bond(eng, ds, W0, opts, tol_S=1e-9, tol_y=1e-9, tol_g=1e-11, **kw):
    T = ds.phi.shape[1]
    load_engine(eng, ds, W0, opts, **kw)
    eng.build_caches()
    W = [t.copy() for t in W0]
    LE, RE = R.construct_caches(W, ds.phi, True)
    for going_left, order in ((True, range(T - 2, -1, -1)), (False, range(0, T - 1))):
        if not going_left:
            LE, RE = R.construct_caches(W, ds.phi, False)
        for lid in order:
            tr_o = {}
            R.bond_step(W, LE, RE, lid, ds, opts, going_left, tr_o)
            tr_g = eng.bond_step(lid, going_left)
            assert tr_g["chi"] == tr_o["chi"], (lid, going_left)
            assert abs(tr_g["loss"] - tr_o["loss"]) <= 1e-11 * max(1.0, abs(tr_o["loss"]))
            assert abs(tr_g["grad_norm"] - tr_o["grad_norm"]) <= tol_g * tr_o["grad_norm"]
            if not opts.rescale[1]:     # the oracle records ||bt_new|| after normalize!, the engine before it
                assert abs(tr_g["bt_new_norm"] - tr_o["bt_new_norm"]) <= 1e-11 * tr_o["bt_new_norm"]
            So = np.asarray(tr_o["S"])
            assert np.abs(np.asarray(tr_g["S"])[:len(So)] - So).max() <= tol_S * So[0]
            yo, yg = R.contract_mps(W, ds.phi), R.contract_mps(eng.get_mps(), ds.phi)
            assert np.abs(yg - yo).max() <= tol_y * np.abs(yo).max()
    return W


@pytest.mark.parametrize("rescale", [(True, True), (True, False), (False, False)])
@pytest.mark.parametrize("loss,bbopt", [("KLD", "TSGO"), ("MSE", "GD")])
def test_rescale_flags(eng, rescale, loss, bbopt):
    """Both normalisations of apply_update switched independently (the default is (False, True))."""
    ds, W0 = make_problem(52, 6, 3, 3, 2, seed=31, balanced=False)
    opts = R.SweepOptions(nsweeps=1, chi_max=7, eta=0.05, loss_grad=loss, bbopt=bbopt, rescale=rescale, update_iters=2)
    W = _sweep_bond_by_bond(eng, ds, W0, opts)
    if not rescale[1]:
        # without the final normalisation the norm of the state is whatever the gradient steps left
        assert abs(R.mps_norm(W) - 1.0) > 1e-6
    assert abs(R.mps_norm(eng.get_mps()) - R.mps_norm(W)) < 1e-10 * R.mps_norm(W)


def test_rescale_before_full_sweep_matches_oracle(eng):
    """mpst_sweep itself (not the bond-by-bond hook) with rescale=(True, True): the path on which bond k+1's tensor is
    NOT assembled by bond k's environment kernel."""
    ds, W0 = make_problem(64, 7, 3, 3, 2, seed=33)
    opts = R.SweepOptions(nsweeps=2, chi_max=8, eta=0.04, rescale=(True, True))
    Wo, info = R.fit(W0, ds, None, opts)
    load_engine(eng, ds, W0, opts)
    eng.build_caches()
    for k in range(2):
        eng.sweep()
        assert abs(eng.eval(0)[1] - info["train_KL_div"][k + 1]) <= 1e-6 * max(1.0, abs(info["train_KL_div"][k + 1]))


def test_recursive_svd_inside_a_sweep(eng):
    """svd_alg="recursive" -> MPST_SVD_JACOBI: every bond of a sweep decomposed by the one-sided Jacobi solver."""
    ds, W0 = make_problem(48, 6, 4, 3, 2, seed=35)
    opts = R.SweepOptions(nsweeps=1, chi_max=10, eta=0.05)
    assert mt.options.engine_options(mt.MPSOptions(svd_alg="recursive", d=4, chi_max=10))["svd_alg"] == 1
    T = ds.phi.shape[1]
    load_engine(eng, ds, W0, opts, svd_alg=1)
    eng.build_caches()
    W = [t.copy() for t in W0]
    LE, RE = R.construct_caches(W, ds.phi, True)
    used = 0
    for going_left, order in ((True, range(T - 2, -1, -1)), (False, range(0, T - 1))):
        if not going_left:
            LE, RE = R.construct_caches(W, ds.phi, False)
        for lid in order:
            tr_o = {}
            R.bond_step(W, LE, RE, lid, ds, opts, going_left, tr_o)
            tr_g = eng.bond_step(lid, going_left)
            used += tr_g["eig_sweeps"] > 0
            assert tr_g["chi"] == tr_o["chi"]
            So = np.asarray(tr_o["S"])
            assert np.abs(np.asarray(tr_g["S"])[:len(So)] - So).max() <= 1e-9 * So[0]
            yo, yg = R.contract_mps(W, ds.phi), R.contract_mps(eng.get_mps(), ds.phi)
            assert np.abs(yg - yo).max() <= 1e-9 * np.abs(yo).max()
    assert used == 2 * (T - 1)          # the Jacobi path really ran on every bond
    # and the same through mpst_sweep (graph replay) from the start
    load_engine(eng, ds, W0, opts, svd_alg=1)
    eng.build_caches()
    st = eng.sweep()
    Wo = [t.copy() for t in W0]
    R.sweep(Wo, ds, opts)
    assert st["eig_sweeps_total"] > 0
    yo, yg = R.contract_mps(Wo, ds.phi), R.contract_mps(eng.get_mps(), ds.phi)
    assert np.abs(yg - yo).max() <= 1e-8 * np.abs(yo).max()


@pytest.mark.parametrize("d,chi_max,T", [(3, 40, 9), (2, 64, 15), (2, 50, 14)])
def test_more_than_32_kept_states(eng, d, chi_max, T):
    """chi_max in (32, 64] with d*chi_max <= 128: the reference's tutorial range chi_max=(20,40)
    (docs/src/hyperparameters.md:44-45) at small d.  The chain is long enough for the bond dimension to get there
    (it is capped by d^j near the ends); the gradient tolerance allows for the 1/yhat weights of a random start."""
    ds, W0 = make_problem(200, T, d, chi_max - 3, 2, seed=41)
    opts = R.SweepOptions(nsweeps=1, chi_max=chi_max, eta=0.05, cutoff=1e-14)
    W = _sweep_bond_by_bond(eng, ds, W0, opts, tol_S=1e-9, tol_y=1e-8, tol_g=1e-9)
    assert max(t.shape[2] for t in W) > 32          # the case really keeps more than 32 states


def test_context_recovers_after_a_failed_decomposition(eng):
    """MPST_ERR_SVD is the class of failure tune() retries on (hyperparameters/tuning.jl:73-86): the same context must
    train again after new inputs, and the per-sweep diagnostics must not accumulate across sweeps."""
    ds, W0 = make_problem(40, 5, 3, 3, 2, seed=43)
    opts = R.SweepOptions(nsweeps=1, chi_max=6, eta=0.05)
    load_engine(eng, ds, W0, opts)
    bad = [t.copy() for t in W0]
    bad[2][...] = np.nan
    eng.set_mps(bad)
    eng.build_caches()
    with pytest.raises(mt.SVDError):
        eng.sweep()
    with pytest.raises(mt.MPSTError, match="mpst_build_caches"):
        eng.sweep()                                   # state after a failure is unspecified: caches must be rebuilt
    eng.set_mps(W0)
    eng.build_caches()
    st1 = eng.sweep()
    st2 = eng.sweep()
    assert st1["eig_fallbacks"] == 0 and st2["eig_fallbacks"] == 0
    Wo = [t.copy() for t in W0]
    R.sweep(Wo, ds, opts)
    R.sweep(Wo, ds, opts)
    yo, yg = R.contract_mps(Wo, ds.phi), R.contract_mps(eng.get_mps(), ds.phi)
    assert np.abs(yg - yo).max() <= 1e-8 * np.abs(yo).max()
    # same for the bond-step hook
    eng.set_mps(bad)
    eng.build_caches()
    with pytest.raises(mt.SVDError):
        eng.bond_step(3, True)
    eng.set_mps(W0)
    eng.build_caches()
    eng.bond_step(3, True)


def test_call_order_is_enforced(eng):
    """Loading a TEST set must not touch the training caches; loading a TRAIN set or an MPS invalidates them."""
    ds, W0 = make_problem(48, 5, 3, 3, 2, seed=45)
    test, _ = make_problem(17, 5, 3, 3, 2, seed=46, balanced=False)
    opts = R.SweepOptions(nsweeps=1, chi_max=6, eta=0.05)
    load_engine(eng, ds, W0, opts)
    with pytest.raises(mt.MPSTError, match="mpst_build_caches"):
        eng.sweep()
    eng.build_caches()
    eng.set_dataset(1, test.phi, test.label_index, 2)           # after build_caches: caches stay valid
    eng.sweep()
    a = eng.get_mps()
    acc_test = eng.eval(1)[2]
    e2 = type(eng)(0)
    try:
        load_engine(e2, ds, W0, opts, test=test)
        e2.build_caches()
        e2.sweep()
        assert all(np.array_equal(x, y) for x, y in zip(a, e2.get_mps()))
        assert acc_test == e2.eval(1)[2]
    finally:
        e2.close()
    eng.set_dataset(0, ds.phi, ds.label_index, 2)               # new training data: caches are gone
    with pytest.raises(mt.MPSTError, match="mpst_build_caches"):
        eng.sweep()
    eng.set_mps(W0)
    eng.build_caches()
    eng.bond_step(3, True)                                       # label now on site 3, not on the last site
    with pytest.raises(mt.MPSTError, match="last site"):
        eng.sweep()
    with pytest.raises(mt.MPSTError, match="does not hold the label"):
        eng.bond_step(0, True)


def test_track_cost_records_the_losses_the_reference_prints(eng):
    """opts.track_cost (loss_functions.jl:50-52,80-82,181-184): per bond the loss before every optimiser step and the loss at
    the updated, normalised bond tensor; checked against the oracle's loss function on the oracle's own trajectory."""
    ds, W0 = make_problem(60, 6, 3, 3, 2, seed=51, balanced=False)
    for loss, rescale in (("KLD", (False, True)), ("MSE", (False, False))):
        opts = R.SweepOptions(nsweeps=1, chi_max=6, eta=0.05, update_iters=2, loss_grad=loss, rescale=rescale)
        load_engine(eng, ds, W0, opts, track_cost=True)
        eng.build_caches()
        eng.sweep()
        tr = eng.loss_trace()
        T = ds.phi.shape[1]
        assert tr.shape == (2 * (T - 1), 3)
        W = [t.copy() for t in W0]
        LE, RE = R.construct_caches(W, ds.phi, True)
        q = 0
        lg = R.LOSS_GRADS[loss]
        for going_left, order in ((True, range(T - 2, -1, -1)), (False, range(0, T - 1))):
            for lid in order:
                bt, shape4 = R.flatten_bt(W[lid], W[lid + 1])
                l0, g0 = lg(bt, LE, RE, ds, lid, lid + 1, False)
                bt1 = bt - opts.eta * g0 / np.linalg.norm(g0)
                l1, g1 = lg(bt1, LE, RE, ds, lid, lid + 1, False)
                bt2 = bt1 - opts.eta * g1 / np.linalg.norm(g1)
                if rescale[1]:
                    bt2 = bt2 / np.linalg.norm(bt2)
                l2, _ = lg(bt2, LE, RE, ds, lid, lid + 1, False)
                assert np.allclose(tr[q], [l0, l1, l2], rtol=1e-9, atol=1e-12), (q, tr[q], l0, l1, l2)
                R.bond_step(W, LE, RE, lid, ds, opts, going_left, {})
                q += 1
    # a sweep without track_cost leaves the trace alone and costs no extra launches
    load_engine(eng, ds, W0, opts)
    eng.build_caches()
    eng.sweep()
    assert np.all(eng.loss_trace() == 0.0)
