"""The HIP engine on the two fixtures of tests/golden/make_golden_complex_impute.py: the ComplexF64 sweep the reference trains through its
legacy ITensor engine (src/legacy_itensor/RealRealLegacyITensor.jl:280-365), bond by bond, free running; and the one imputation
instance (src/Imputation/MPS_methods.jl:201-230, both orders).  A maintainer's juliaref_<name>.npz (tests/golden/make_reference_goldens.jl)
pins the same numbers on the reference; here they are the oracle's."""
import os

import numpy as np
import pytest

import mpstime_jl_amd as mt
from oracle import ref_numpy as R

pytestmark = pytest.mark.gpu
HERE = os.path.join(os.path.dirname(__file__), "golden")


def test_engine_reproduces_the_complex_fixture_bond_by_bond():
    g = np.load(os.path.join(HERE, "complex_kld_c2.npz"))
    phi = g["phi"]
    T = phi.shape[1]
    chimax, iters, nsw, sep = [int(x) for x in g["opts"]]
    eng = mt.SweepEngine(0)
    try:
        eng.set_options(chi_max=chimax, eta=float(g["eta"]), cutoff=1e-10, update_iters=iters, loss=str(g["loss"]), bbopt=str(g["bbopt"]),
                        rescale=(False, True), train_classes_separately=bool(sep))
        eng.set_dataset(0, phi, g["label_index"], len(g["class_distribution"]))
        eng.set_mps([g[f"W0_{j}"] for j in range(T)])
        eng.build_caches()
        assert eng.info()["typed_kernels"]
        klds = [eng.eval(0)[1]]
        k = 0
        for _ in range(nsw):
            for going_left, order in ((True, range(T - 2, -1, -1)), (False, range(0, T - 1))):
                for lid in order:
                    tr = eng.bond_step(lid, going_left)
                    assert int(g["bond_lid"][k]) == lid and bool(g["bond_left"][k]) == going_left
                    assert tr["chi"] == int(g["bond_chi"][k]), k
                    assert abs(tr["loss"] - g["bond_loss"][k]) <= 1e-7 * max(1.0, abs(g["bond_loss"][k])), k
                    assert abs(tr["grad_norm"] - g["bond_grad_norm"][k]) <= 1e-6 * g["bond_grad_norm"][k], k
                    So = g["bond_S"][k, :tr["chi"]]
                    assert np.abs(tr["S"][:tr["chi"]] - So).max() <= 1e-7 * So[0], k
                    k += 1
            klds.append(eng.eval(0)[1])
        assert np.allclose(klds, g["train_KL_div"], rtol=1e-6)
        W = eng.get_mps()
        ds = R.EncodedSet(phi, g["label_index"], g["class_distribution"])
        # overlaps are gauge invariant up to nothing at all (the label site carries the phases of the split)
        assert np.abs(np.abs(R.contract_mps(W, ds.phi)) - np.abs(g["overlaps"])).max() <= 1e-6 * np.abs(g["overlaps"]).max()
        chi, _ = eng.get_chi()
        assert np.array_equal(chi, g["final_chi"])
    finally:
        eng.close()


@pytest.mark.parametrize("compute", ["f64", "f32"])
def test_engine_reproduces_the_imputation_instance(compute):
    g = np.load(os.path.join(HERE, "impute_median_c1.npz"))
    T = len(g["x"])
    W = [g[f"mps_{j}"] for j in range(T)]
    W[-1] = W[-1][..., None]                    # a one-class label index on the last site
    m = np.zeros((1, T), dtype=np.uint8)
    m[0, g["missing"]] = 1
    eng = mt.SweepEngine(0)
    try:
        for oi, order in enumerate(("forwards", "backwards")):
            x, err, _ = eng.impute_model(W, g["enc"][None], np.zeros(1, dtype=np.int32), m, g["xs"], g["grid_phi"], 0, True, order=oi, compute=compute)
            got, want = x[0, g["missing"]], g[f"x_{order}"]
            step = g["xs"][1] - g["xs"][0]
            if compute == "f64":
                assert np.array_equal(got, want), (order, got, want)
                assert np.allclose(err[0, g["missing"]], g[f"wmad_{order}"], rtol=0, atol=1e-9)
            else:       # fp32 chain contractions: a median may land on a neighbouring grid value (and then everything after it shifts a little)
                assert np.abs(got - want).max() <= 20 * step, (order, got, want)
    finally:
        eng.close()
