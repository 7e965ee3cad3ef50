"""Parity of the HIP sweep engine (through the C ABI) with the CPU oracle on the same seeded inputs.

Tolerances (fp64): per-bond loss / ||grad|| / ||bt|| 1e-11 relative; singular values 1e-9
relative to the largest; gauge-invariant overlaps <W|phi_i> of the whole MPS with every series
1e-9 relative to the largest overlap; bond dimensions, labels, predictions and confusion counts exact.
"""
import numpy as np
import pytest

from oracle import ref_numpy as R
from tests.helpers import load_engine, make_problem

pytestmark = pytest.mark.gpu

RTOL_SCALAR = 1e-11
TOL_S = 1e-9
TOL_BT = 1e-9


@pytest.fixture()
def eng(engine_cls):
    e = engine_cls(0)
    yield e
    e.close()


def _check_bond(tr_o, tr_g):
    assert tr_g["chi"] == tr_o["chi"]
    assert abs(tr_g["loss"] - tr_o["loss"]) <= RTOL_SCALAR * max(1.0, abs(tr_o["loss"]))
    assert abs(tr_g["grad_norm"] - tr_o["grad_norm"]) <= RTOL_SCALAR * tr_o["grad_norm"]
    So = np.asarray(tr_o["S"])
    Sg = np.asarray(tr_g["S"])[: len(So)]
    assert np.abs(Sg - So).max() <= TOL_S * So[0]


CASES = [
    # N, T, d, chi_init, chi_max, C, loss, bbopt, sep, iters, balanced
    (48, 6, 2, 2, 4, 2, "KLD", "TSGO", False, 1, True),
    (40, 5, 3, 3, 6, 3, "KLD", "TSGO", True, 1, False),
    (37, 4, 4, 4, 8, 2, "MSE", "TSGO", False, 1, False),
    (64, 6, 4, 4, 16, 2, "KLD", "GD", False, 3, True),
    (33, 3, 3, 2, 5, 1, "KLD", "TSGO", False, 2, True),
    (20, 2, 3, 1, 5, 2, "KLD", "TSGO", False, 1, True),
    (50, 7, 2, 2, 7, 4, "MSE", "GD", False, 2, False),
    # edges: the maximum number of classes (MAX_C = 16), fewer series than one tile, the largest d*chi the
    # eigensolver holds (7*18 = 126 <= 128) with a non-power-of-two d, and chi_max = 32 with d = 4 (128)
    (70, 4, 2, 2, 6, 16, "KLD", "TSGO", False, 1, False),
    (3, 5, 3, 2, 6, 2, "KLD", "TSGO", False, 1, True),
    (60, 4, 7, 6, 18, 2, "KLD", "TSGO", False, 1, True),
    (96, 4, 4, 8, 32, 3, "MSE", "TSGO", False, 1, False),
    # d > 8 (two site-vector pieces per loader lane in k_grad_s, one bond entry per gradient block row) and d = 16
    (45, 4, 12, 5, 10, 2, "KLD", "TSGO", False, 1, False),
    (30, 3, 16, 4, 8, 3, "MSE", "GD", False, 2, True),
    # more than 32 eigenpairs wanted from the 128 x 128 solver (d = 2, chi_max = 64: 64 eigenvector workgroups, 64-column
    # eigenvector block in k_eig_fin)
    (80, 14, 2, 48, 64, 2, "KLD", "TSGO", False, 1, True),
]


@pytest.mark.parametrize("case", CASES, ids=[f"case{i}" for i in range(len(CASES))])
def test_sweep_bond_by_bond(eng, case):
    N, T, d, chi0, chimax, C, loss, bbopt, sep, iters, bal = case
    ds, W0 = make_problem(N, T, d, chi0, C, seed=N + T, balanced=bal)
    opts = R.SweepOptions(nsweeps=1, chi_max=chimax, eta=0.05, update_iters=iters, loss_grad=loss, bbopt=bbopt,
                          train_classes_separately=sep)
    load_engine(eng, ds, W0, opts)
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
            _check_bond(tr_o, tr_g)
            # gauge-invariant state after the split: the overlaps of the whole MPS with every series
            yo = R.contract_mps(W, ds.phi)
            yg = R.contract_mps(eng.get_mps(), ds.phi)
            assert np.abs(yg - yo).max() <= TOL_BT * np.abs(yo).max()
    # evaluation read-out (summary.jl:33-114)
    mse, kld, acc, conf = eng.eval(0)
    mo, ko, ao, co = R.mse_loss_acc(W, ds, conf=True)
    assert abs(mse - mo) < 1e-9 and abs(kld - ko) < 1e-9 * max(1, abs(ko)) and acc == ao
    assert np.array_equal(conf, co)


def test_full_fit_loss_curve(eng):
    """nsweeps of mpst_sweep vs the oracle's fit: per-sweep train KLD/MSE/acc, predictions."""
    N, T, d, chi0, chimax, C = 96, 8, 3, 3, 9, 2
    ds, W0 = make_problem(N, T, d, chi0, C, seed=5)
    test, _ = make_problem(31, T, d, chi0, C, seed=6, balanced=False)
    opts = R.SweepOptions(nsweeps=3, chi_max=chimax, eta=0.03)
    Wo, info = R.fit(W0, ds, test, opts)
    load_engine(eng, ds, W0, opts, test=test)
    eng.build_caches()
    curve = [eng.eval(0)]
    for _ in range(opts.nsweeps):
        eng.sweep()
        curve.append(eng.eval(0))
    eng.normalize()
    curve.append(eng.eval(0))
    for k, (mse, kld, acc, _) in enumerate(curve):
        assert abs(kld - info["train_KL_div"][k]) <= 1e-6 * max(1.0, abs(info["train_KL_div"][k]))
        assert abs(mse - info["train_loss"][k]) <= 1e-6
        assert acc == info["train_acc"][k]
    tm, tk, ta, tconf = eng.eval(1)
    assert abs(tk - info["test_KL_div"][-1]) <= 1e-6 * max(1.0, abs(info["test_KL_div"][-1]))
    assert np.array_equal(tconf, info["test_conf"][-1])
    pred, yh = eng.classify(1, return_overlaps=True)
    assert np.array_equal(pred, R.classify(Wo, test.phi))
    assert abs(R.mps_norm(eng.get_mps()) - 1.0) < 1e-12
    chi, ls = eng.get_chi()
    assert ls == T - 1 and list(chi) == [1] + [t.shape[2] for t in Wo]


def test_rebuild_caches_is_bit_identical(engine_cls):
    """the two full cache rebuilds per sweep recompute what is already stored."""
    ds, W0 = make_problem(64, 6, 3, 3, 2, seed=11)
    opts = R.SweepOptions(nsweeps=2, chi_max=6, eta=0.04)
    out = []
    for rebuild in (False, True):
        e = engine_cls(0)
        load_engine(e, ds, W0, opts, rebuild_caches=rebuild)
        e.build_caches()
        for _ in range(2):
            e.sweep()
        out.append(e.get_mps())
        e.close()
    for a, b in zip(*out):
        assert np.array_equal(a, b)
