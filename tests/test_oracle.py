"""CPU checks of the oracle itself: the NumPy restatement against the committed golden fixtures,
against the independent autograd formulation, and against the C restatement; the NDTensors
truncation rule on hand-worked spectra; structural properties the reference's tests pin
(test/classification.jl:22-24)."""
import glob
import os

import numpy as np
import pytest

from oracle import naive
from oracle import ref_numpy as R
from tests.helpers import make_problem

GOLDEN = [p for p in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
          if not os.path.basename(p).startswith(("ref_", "juliaref_", "complex_", "impute_"))]      # ref_*: reference-produced tensors (test_reference_fixture.py)


def load_golden(path):
    g = np.load(path)
    T = g["phi"].shape[1]
    W0 = [g[f"W0_{j}"] for j in range(T)]
    ds = R.EncodedSet(g["phi"], g["label_index"], g["class_distribution"])
    chimax, iters, nsw, sep = [int(x) for x in g["opts"]]
    opts = R.SweepOptions(nsweeps=nsw, chi_max=chimax, eta=float(g["eta"]), update_iters=iters,
                          loss_grad=str(g["loss"]), bbopt=str(g["bbopt"]), train_classes_separately=bool(sep))
    return g, ds, W0, opts


JULIAREF = [p for p in GOLDEN if os.path.exists(os.path.join(os.path.dirname(p), "juliaref_" + os.path.basename(p)))]


@pytest.mark.skipif(not JULIAREF, reason="no tests/golden/juliaref_*.npz: a maintainer with Julia writes them with "
                                         "tests/golden/make_reference_goldens.jl (the pin of the sweep trajectory on the reference)")
@pytest.mark.parametrize("path", JULIAREF, ids=[os.path.basename(p)[:-4] for p in JULIAREF])
def test_oracle_against_reference_vectors(path):
    """Per-bond {loss, ||grad||, chi, kept singular values} and the per-sweep KLD of the JULIA REFERENCE on the fixture's inputs."""
    g, ds, W0, opts = load_golden(path)
    ref = np.load(os.path.join(os.path.dirname(path), "juliaref_" + os.path.basename(path)))
    rec = []
    Wf, info = R.fit(W0, ds, None, opts, record=rec)
    bonds = [b for sw in rec for b in sw]
    assert len(bonds) == len(ref["bond_loss"])
    assert np.array_equal([b["chi"] for b in bonds], ref["bond_chi"])
    assert np.allclose([b["loss"] for b in bonds], ref["bond_loss"], rtol=1e-8, atol=1e-11)
    assert np.allclose([b["grad_norm"] for b in bonds], ref["bond_grad_norm"], rtol=1e-8)
    for i, b in enumerate(bonds):
        assert np.allclose(b["S"], ref["bond_S"][i, :len(b["S"])], rtol=0, atol=1e-8 * b["S"][0])
    assert np.allclose(info["train_KL_div"][:len(ref["train_KL_div"])], ref["train_KL_div"], rtol=1e-6)


def test_golden_fixtures_present():
    assert len(GOLDEN) >= 6


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_numpy_oracle_reproduces_golden(path):
    g, ds, W0, opts = load_golden(path)
    if ds.N * ds.phi.shape[1] > 5000:
        opts.nsweeps = 1        # keep the CPU suite short; the first sweep of the fixture is compared
    rec = []
    Wf, info = R.fit(W0, ds, None, opts, record=rec)
    bonds = [b for sw in rec for b in sw]
    nb = len(bonds)
    assert np.array_equal([b["chi"] for b in bonds], g["bond_chi"][:nb])
    assert np.allclose([b["loss"] for b in bonds], g["bond_loss"][:nb], rtol=1e-9, atol=1e-12)
    assert np.allclose([b["grad_norm"] for b in bonds], g["bond_grad_norm"][:nb], rtol=1e-9)
    for i, b in enumerate(bonds):
        assert np.allclose(b["S"], g["bond_S"][i, :len(b["S"])], rtol=0, atol=1e-9)
    k = opts.nsweeps + 1
    assert np.allclose(info["train_KL_div"][:k], g["train_KL_div"][:k], rtol=1e-8)
    if opts.nsweeps == int(g["opts"][2]):
        assert np.array_equal(R.classify(Wf, ds.phi), g["pred"])
        assert np.allclose(R.contract_mps(Wf, ds.phi), g["overlaps"], atol=1e-8 * np.abs(g["overlaps"]).max())


@pytest.mark.parametrize("path", [p for p in GOLDEN if "config1" not in p],
                         ids=[os.path.basename(p)[:-4] for p in GOLDEN if "config1" not in p])
def test_c_oracle_matches_golden(path):
    from oracle.c_oracle import COracle
    g, ds, W0, opts = load_golden(path)
    co = COracle(W0, ds.phi, ds.label_index, ds.class_distribution, opts.chi_max, eta=opts.eta,
                 update_iters=opts.update_iters, loss=opts.loss_grad, bbopt=opts.bbopt,
                 train_classes_separately=opts.train_classes_separately)
    co.build_caches()
    bonds = []
    for _ in range(opts.nsweeps):
        bonds += co.sweep(record=True)["bonds_rec"]
    assert np.array_equal([b["chi"] for b in bonds], g["bond_chi"])
    assert np.allclose([b["loss"] for b in bonds], g["bond_loss"], rtol=1e-9, atol=1e-12)
    assert np.allclose([b["grad_norm"] for b in bonds], g["bond_grad_norm"], rtol=1e-9)
    Wc = R.normalize_mps(co.get_mps())
    assert np.allclose(R.contract_mps(Wc, ds.phi), g["overlaps"], atol=1e-8 * np.abs(g["overlaps"]).max())


@pytest.mark.parametrize("loss,sep", [("KLD", False), ("KLD", True), ("MSE", False)])
def test_array_path_equals_autograd_definition(loss, sep):
    """The fused one-sample-late loop (loss_functions.jl:248-262) is the same maths as the
    definition differentiated by autograd (the reference's array-vs-legacy check, classification.jl:24)."""
    ds, W = make_problem(30, 5, 3, 3, 3, seed=9, balanced=False)
    LE, RE = R.construct_caches(W, ds.phi, True)
    opts = R.SweepOptions(chi_max=6, eta=0.05, loss_grad=loss, train_classes_separately=sep)
    for lid in range(3, -1, -1):
        bt, shape4 = R.flatten_bt(W[lid], W[lid + 1])
        l, gr = R.LOSS_GRADS[loss](bt, LE, RE, ds, lid, lid + 1, sep)
        l2, g2, _ = naive.loss_and_grad(W, R.unflatten_bt(bt, shape4), lid, ds.phi, ds.label_index,
                                         ds.class_distribution, loss, sep)
        f = 0.5 if loss == "KLD" else 1.0
        assert abs(l - l2) < 1e-12 * max(1, abs(l2))
        assert np.abs(R.unflatten_bt(gr, shape4) - f * g2).max() < 1e-12 * np.abs(g2).max()
        R.bond_step(W, LE, RE, lid, ds, opts, True)


def test_truncation_rule_hand_cases():
    """NDTensors truncate! with relative cutoff."""
    S = np.sqrt([0.5, 0.3, 0.15, 0.05])
    assert R.truncate_spectrum(S, 10, 0.0) == 4
    assert R.truncate_spectrum(S, 2, 0.0) == 2                 # maxdim first
    assert R.truncate_spectrum(S, 10, 0.05) == 3               # 0.05 <= 0.05*1
    assert R.truncate_spectrum(S, 10, 0.0499) == 4
    assert R.truncate_spectrum(S, 10, 0.2) == 2                # 0.05 + 0.15 <= 0.2
    assert R.truncate_spectrum(S, 3, 0.1) == 3                 # the weight cut by maxdim counts: 0.05+0.15 > 0.1
    assert R.truncate_spectrum(S, 10, 1.0) == 1                # mindim = 1
    assert R.truncate_spectrum(np.array([2.0]), 10, 0.5) == 1
    assert R.truncate_spectrum(np.array([1.0, 0.0, 0.0]), 10, 1e-10) == 1


def test_split_preserves_bond_tensor_and_canonical_form():
    ds, W = make_problem(20, 4, 3, 3, 2, seed=2)
    bt, shape4 = R.flatten_bt(W[2], W[3])
    bt5 = R.unflatten_bt(bt / np.linalg.norm(bt), shape4)
    for going_left in (True, False):
        l, r, S = R.decompose_bt(bt5, 2, 3, chi_max=100, cutoff=0.0, going_left=going_left)
        bt_back, _ = R.flatten_bt(l, r)
        assert np.allclose(R.unflatten_bt(bt_back, shape4), bt5, atol=1e-13)
        iso = r if going_left else l
        m = iso.reshape(iso.shape[0], -1) if going_left else iso.reshape(-1, iso.shape[2]).T
        assert np.allclose(m @ m.T, np.eye(m.shape[0]), atol=1e-13)
        assert abs(np.sum(S ** 2) - 1.0) < 1e-13


def test_predictions_invariant_to_test_data_and_normalisation():
    """classification.jl:23: supplying test data does not change the trained MPS; normalize! leaves
    predictions unchanged (only the overall scale of the overlaps moves)."""
    ds, W0 = make_problem(40, 5, 3, 3, 2, seed=4)
    test, _ = make_problem(11, 5, 3, 3, 2, seed=5)
    opts = R.SweepOptions(nsweeps=2, chi_max=6, eta=0.05)
    Wa, ia = R.fit(W0, ds, None, opts)
    Wb, ib = R.fit(W0, ds, test, opts)
    for a, b in zip(Wa, Wb):
        assert np.array_equal(a, b)
    assert len(ib["test_acc"]) == opts.nsweeps + 2 and np.isnan(ib["time_taken"][-1]) and ib["time_taken"][0] == 0.0
    assert abs(R.mps_norm(Wa) - 1.0) < 1e-13


def test_cache_rebuild_is_recomputation():
    """after a half-sweep the environments written bond by bond equal a full rebuild."""
    ds, W = make_problem(16, 5, 2, 2, 2, seed=8)
    opts = R.SweepOptions(chi_max=4, eta=0.05)
    LE, RE = R.construct_caches(W, ds.phi, True)
    for j in range(3, -1, -1):
        R.bond_step(W, LE, RE, j, ds, opts, True)
    _, RE2 = R.construct_caches(W, ds.phi, False)
    for j in range(1, 5):
        assert np.allclose(RE[j], RE2[j], rtol=1e-13, atol=1e-15)
