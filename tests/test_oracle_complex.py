"""oracle/ref_complex.py (the restatement of the reference's complex / legacy-engine semantics) checked without a GPU:
  * its real specialisation reproduces oracle/ref_numpy.py's per-sample restatement of the array engine - the relation the
    reference's own test asserts between its two engines (test/classification.jl:24);
  * its gradient is the conjugate Wirtinger derivative of the loss written as a full-chain contraction (independent
    formulation, finite differences), for KLD, MSE and train_classes_separately;
  * complex SVD split: left * right reproduces the truncated bond tensor, orthonormality of the unlabelled site.
"""
import numpy as np
import pytest

from oracle import ref_complex as RC
from oracle import ref_numpy as R
from tests.helpers import make_problem


@pytest.mark.parametrize("loss,sep", [("KLD", False), ("KLD", True), ("MSE", False)])
def test_real_specialisation_equals_the_array_engine_restatement(loss, sep):
    ds, W = make_problem(40, 5, 3, 3, 2, seed=3, balanced=False)
    LE, RE = R.construct_caches(W, ds.phi, going_left=True)
    lid = 3
    bt, shape4 = R.flatten_bt(W[lid], W[lid + 1])
    l1, g1 = R.LOSS_GRADS[loss](bt, LE, RE, ds, lid, lid + 1, sep)
    bt5 = R.unflatten_bt(bt, shape4)
    l2, g2 = RC.loss_grad(bt5, LE[lid - 1], None, ds, lid, lid + 1, loss, sep)
    assert abs(l1 - l2) < 1e-12 * max(1, abs(l1))
    assert np.abs(R.unflatten_bt(g1, shape4) - g2).max() < 1e-12 * np.abs(g2).max()


@pytest.mark.parametrize("loss,sep", [("KLD", False), ("KLD", True), ("MSE", False)])
@pytest.mark.parametrize("lid", [0, 2, 4])
def test_gradient_is_the_conjugate_wirtinger_derivative_of_the_full_chain_loss(loss, sep, lid):
    ds, W = RC.make_problem(30, 6, 3, 3, 2, seed=7, dtype=np.complex128, balanced=False)
    T = len(W)
    # put the label on site lid + 1 by sweeping left from the end (oracle's own sweep code)
    opts = RC.SweepOptions(chi_max=6, eta=0.05)
    LE, RE = RC.construct_caches(W, ds.phi, going_left=True)
    for j in range(T - 2, lid, -1):
        RC.bond_step(W, LE, RE, j, ds, opts, True)
    bt, shape4 = RC.flatten_bt(W[lid], W[lid + 1])
    bt5 = RC.unflatten_bt(bt, shape4)
    LEp = LE[lid - 1] if lid > 0 else None
    REp = RE[lid + 2] if lid + 1 < T - 1 else None
    l0, g = RC.loss_grad(bt5, LEp, REp, ds, lid, lid + 1, loss, sep)
    f0, _ = RC.full_chain_loss(W, bt5, lid, ds, loss, sep)
    assert abs(l0 - f0) < 1e-12 * max(1, abs(f0))
    rng = np.random.default_rng(0)
    for _ in range(3):
        dB = rng.standard_normal(bt5.shape) + 1j * rng.standard_normal(bt5.shape)
        eps = 1e-6
        fp, _ = RC.full_chain_loss(W, bt5 + eps * dB, lid, ds, loss, sep)
        fm, _ = RC.full_chain_loss(W, bt5 - eps * dB, lid, ds, loss, sep)
        fd = (fp - fm) / (2 * eps)
        # dL = 2 Re <g, dB> for g = dL/d conj(B).  The reference's KLD gradient IS that derivative, -conj(phi~/yhat)
        # (legacy loss_functions.jl:482); its MSE gradient (yhat - y) conj(phi~) (:616) is twice it - the same factor conventions
        # as its real engine (d(-log y^2)/dB = -2 phi/y is accumulated as -phi/y, d(0.5 (y-t)^2)/dB = (y-t) phi as is).
        an = (2 if loss == "KLD" else 1) * np.real(np.sum(np.conj(g) * dB))
        assert abs(fd - an) < 1e-6 * max(1.0, abs(an)), (fd, an)


def test_complex_split_reproduces_the_truncated_bond_tensor():
    ds, W = RC.make_problem(20, 5, 3, 3, 2, seed=1, dtype=np.complex128)
    bt, shape4 = RC.flatten_bt(W[3], W[4])
    bt5 = RC.unflatten_bt(bt, shape4)
    for gl in (True, False):
        left, right, S = RC.decompose_bt(bt5, 64, 0.0, going_left=gl)
        if gl:
            rec = np.einsum("askc,ktb->satbc", left, right)
            m = right.reshape(right.shape[0], -1)
            assert np.abs(m @ m.conj().T - np.eye(len(S))).max() < 1e-12
        else:
            rec = np.einsum("ask,ktbc->satbc", left, right)
            m = left.reshape(-1, left.shape[2])
            assert np.abs(m.conj().T @ m - np.eye(len(S))).max() < 1e-12
        assert np.abs(rec - bt5).max() < 1e-12


def test_reduced_precision_problem_keeps_its_element_type():
    ds, W = RC.make_problem(16, 4, 2, 2, 2, seed=2, dtype=np.complex64)
    assert ds.phi.dtype == np.complex64 and all(t.dtype == np.complex64 for t in W)
    opts = RC.SweepOptions(chi_max=4, eta=0.05)
    RC.sweep(W, ds, opts)
    assert all(t.dtype == np.complex64 for t in W)
    with pytest.raises(ValueError, match="complex valued encoding"):
        RC.cast_problem(ds, W, np.float32)


# ---- the committed complex fixture (tests/golden/make_golden_complex_impute.py) and, when a maintainer has written it, the reference's own run on it ----
import os

_CFIX = os.path.join(os.path.dirname(__file__), "golden", "complex_kld_c2.npz")
_CREF = os.path.join(os.path.dirname(__file__), "golden", "juliaref_complex_kld_c2.npz")


def _complex_fixture_trajectory():
    g = np.load(_CFIX)
    T = g["phi"].shape[1]
    ds = R.EncodedSet(g["phi"], g["label_index"], g["class_distribution"])
    W = [g[f"W0_{j}"].copy() for j in range(T)]
    chimax, iters, nsw, sep = [int(x) for x in g["opts"]]
    opts = RC.SweepOptions(nsweeps=nsw, chi_max=chimax, eta=float(g["eta"]), update_iters=iters, loss_grad=str(g["loss"]), bbopt=str(g["bbopt"]),
                           train_classes_separately=bool(sep))
    bonds, klds = [], [R.mse_loss_acc(W, ds)[1]]
    for _ in range(nsw):
        rec = []
        RC.sweep(W, ds, opts, record=rec)
        bonds += rec
        klds.append(R.mse_loss_acc(W, ds)[1])
    return g, bonds, klds


def test_complex_fixture_is_reproduced():
    g, bonds, klds = _complex_fixture_trajectory()
    assert np.iscomplexobj(g["phi"]) and np.iscomplexobj(g["W0_0"])
    assert np.array_equal([b["chi"] for b in bonds], g["bond_chi"])
    assert np.allclose([b["loss"] for b in bonds], g["bond_loss"], rtol=1e-10)
    assert np.allclose([b["grad_norm"] for b in bonds], g["bond_grad_norm"], rtol=1e-9)
    assert np.allclose(klds, g["train_KL_div"], rtol=1e-9)


@pytest.mark.skipif(not os.path.exists(_CREF), reason="no tests/golden/juliaref_complex_kld_c2.npz: a maintainer with Julia writes it with "
                                                      "tests/golden/make_reference_goldens.jl (the reference's legacy ITensor engine on this fixture)")
def test_complex_oracle_against_reference_vectors():
    """fitMPS_IT's sweep body (use_legacy_ITensor = true, ComplexF64) on the fixture's inputs: per-bond loss, ||grad||, chi, kept singular values."""
    g, bonds, klds = _complex_fixture_trajectory()
    ref = np.load(_CREF)
    assert np.array_equal([b["chi"] for b in bonds], ref["bond_chi"])
    assert np.allclose([b["loss"] for b in bonds], ref["bond_loss"], rtol=1e-8, atol=1e-11)
    assert np.allclose([b["grad_norm"] for b in bonds], ref["bond_grad_norm"], rtol=1e-8)
    for i, b in enumerate(bonds):
        assert np.allclose(b["S"], ref["bond_S"][i, :len(b["S"])], rtol=0, atol=1e-8 * b["S"][0])
    assert np.allclose(klds, ref["train_KL_div"], rtol=1e-6)
