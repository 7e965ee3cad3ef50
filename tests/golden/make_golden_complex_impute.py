#!/usr/bin/env python3
"""Two more golden fixtures, written like tests/golden/make_golden.py's:

  complex_kld_c2.npz       a ComplexF64 sweep (Fourier basis, d = 3): what the reference trains through its legacy ITensor engine
                           (`use_legacy_ITensor = true`, src/legacy_itensor/RealRealLegacyITensor.jl:147-420 - its array engine raises
                           on complex encodings, RealRealHighDimension.jl:461-466).  Same fields as the real fixtures; from
                           oracle/ref_complex.py.
  impute_median_c1.npz     ONE imputation instance (src/Imputation/MPS_methods.jl:201-230 impute_median, get_wmad = true): a class
                           MPS, the encoded series, the missing sites, the grid - and the imputed values + weighted median absolute
                           deviations for both imputation orders; from oracle/impute_numpy.py.

Both come from this repository's CPU restatements (PARITY UNPINNED against the reference itself: no Julia in the build image);
tests/golden/make_reference_goldens.jl runs the REFERENCE on the same inputs and writes juliaref_<name>.npz beside them.

Run:  python tests/golden/make_golden_complex_impute.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import impute_numpy as I  # noqa: E402
from oracle import ref_complex as RC  # noqa: E402
from oracle import ref_numpy as R  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def complex_sweep():
    N, T, d, chi0, chimax, C, nsw, eta = 48, 6, 3, 3, 9, 2, 2, 0.05
    ds, W0 = RC.make_problem(N, T, d, chi0, C, seed=321, dtype=np.complex128, encoding="fourier", balanced=True)
    opts = RC.SweepOptions(nsweeps=nsw, chi_max=chimax, eta=eta, update_iters=1, loss_grad="KLD", bbopt="TSGO",
                           train_classes_separately=False)
    W = [t.copy() for t in W0]
    bonds = []
    klds = [R.mse_loss_acc(W, ds)[1]]
    for _ in range(nsw):
        rec = []
        RC.sweep(W, ds, opts, record=rec)
        bonds += rec
        klds.append(R.mse_loss_acc(W, ds)[1])
    smax = max(len(b["S"]) for b in bonds)
    S = np.zeros((len(bonds), smax))
    for i, b in enumerate(bonds):
        S[i, :len(b["S"])] = b["S"]
    out = dict(phi=ds.phi, label_index=ds.label_index, class_distribution=ds.class_distribution,
               opts=np.array([chimax, 1, nsw, 0], dtype=np.int64), eta=eta, loss="KLD", bbopt="TSGO",
               bond_loss=np.array([b["loss"] for b in bonds]), bond_grad_norm=np.array([b["grad_norm"] for b in bonds]),
               bond_bt_norm=np.array([b["bt_new_norm"] for b in bonds]), bond_chi=np.array([b["chi"] for b in bonds]),
               bond_lid=np.array([b["lid"] for b in bonds]), bond_left=np.array([b["going_left"] for b in bonds]),
               bond_S=S, train_KL_div=np.array(klds), overlaps=R.contract_mps(W, ds.phi),
               final_chi=np.array([1] + [t.shape[2] for t in W]))
    for j, t in enumerate(W0):
        out[f"W0_{j}"] = t
    np.savez_compressed(os.path.join(HERE, "complex_kld_c2.npz"), **out)
    print(f"complex_kld_c2: {len(bonds)} bonds, KLD {klds}")


def impute_instance():
    T, d, chi, ngrid = 12, 4, 6, 2001
    rng = np.random.default_rng(77)
    W = R.random_mps(T, d, chi, 2, rng)
    mps = I.expand_label_index(W)[1]
    xs = -1.0 + 1e-3 * np.arange(ngrid)                 # the reference's grid: range(-1, 1; step = dx)  (imputation.jl:88-99)
    grid_phi = R.legendre_encode(xs, d)
    x = rng.uniform(-0.9, 0.9, T)
    enc = R.legendre_encode(x, d)
    missing = np.array([0, 3, 4, 5, 8, 11])
    out = dict(x=x, enc=enc, missing=missing, xs=xs, grid_phi=grid_phi, d=d)
    for j, t in enumerate(mps):
        out[f"mps_{j}"] = t
    for order in ("forwards", "backwards"):
        xi, err = I.impute(mps, enc, missing, xs, grid_phi, "median", order, True)
        out[f"x_{order}"] = np.asarray(xi)
        out[f"wmad_{order}"] = np.asarray(err)
    np.savez_compressed(os.path.join(HERE, "impute_median_c1.npz"), **out)
    print("impute_median_c1:", out["x_forwards"], out["wmad_forwards"])


if __name__ == "__main__":
    complex_sweep()
    impute_instance()
