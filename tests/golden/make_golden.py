#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz.

The Julia/ITensors reference cannot run in the build container, so these vectors
come from the CPU restatement oracle/ref_numpy.py and are cross-validated, bond by bond, against
the independent autograd formulation oracle/naive.py before they are written (the array-engine vs
legacy-engine idea of the reference's own test/classification.jl:24).  PARITY UNPINNED against the
reference itself.

Each fixture holds: the encoded inputs, the initial MPS, the options, and per bond
{loss, ||grad||_F, ||bt_new||_F, chi_new, kept singular values}, per sweep {train MSE, KLD, acc},
the final overlaps <W|phi_i> and predictions.

Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import naive  # noqa: E402
from oracle import ref_numpy as R  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# name: (N, T, d, chi_init, chi_max, C, loss, bbopt, train_sep, update_iters, nsweeps, eta, balanced)
CASES = {
    "kld_tsgo_c2": (64, 6, 4, 4, 12, 2, "KLD", "TSGO", False, 1, 2, 0.05, True),
    "kld_sep_c3_ragged": (45, 5, 3, 3, 7, 3, "KLD", "TSGO", True, 1, 2, 0.05, False),
    "mse_gd_c2_iters3": (40, 4, 2, 2, 4, 2, "MSE", "GD", False, 3, 2, 0.1, False),
    "kld_c1_unsupervised": (33, 3, 4, 2, 6, 1, "KLD", "TSGO", False, 1, 2, 0.05, True),
    "two_site_mps": (24, 2, 3, 1, 5, 2, "KLD", "TSGO", False, 2, 2, 0.05, True),
    "config1_trendy_sine": (200, 50, 2, 4, 4, 2, "KLD", "TSGO", False, 1, 2, 0.01, True),   # BASELINE configs[0]
}


def inputs(name, N, T, d, chi0, C, balanced, seed):
    rng = np.random.default_rng(seed)
    if name == "config1_trendy_sine":
        X, y = R.trendy_sine_dataset(N, T, rng)
        Xs, _ = R.transform_train_data(X)
        ds = R.encode_dataset(X, Xs, y, lambda x: R.legendre_encode(x, d), (-1, 1))
    else:
        X = rng.uniform(-1, 1, (N, T))
        y = (np.arange(N) % C) if balanced else np.concatenate([np.arange(C), rng.integers(0, C, N - C)])
        rng.shuffle(y)
        ds = R.encode_dataset(X, X, y, lambda x: R.legendre_encode(x, d), (-1, 1))
    W0 = R.random_mps(T, d, chi0, C, np.random.default_rng(seed + 1000))
    return ds, W0


def cross_validate(W0, ds, opts):
    """array-path formula vs autograd definition at every bond of the first backward half-sweep."""
    W = [t.copy() for t in W0]
    LE, RE = R.construct_caches(W, ds.phi, True)
    worst = 0.0
    for lid in range(len(W) - 2, -1, -1):
        bt, shape4 = R.flatten_bt(W[lid], W[lid + 1])
        l, g = R.LOSS_GRADS[opts.loss_grad](bt, LE, RE, ds, lid, lid + 1, opts.train_classes_separately)
        l2, g2, _ = naive.loss_and_grad(W, R.unflatten_bt(bt, shape4), lid, ds.phi, ds.label_index,
                                         ds.class_distribution, opts.loss_grad, opts.train_classes_separately)
        f = 0.5 if opts.loss_grad == "KLD" else 1.0
        worst = max(worst, abs(l - l2) / max(1.0, abs(l2)),
                    np.abs(R.unflatten_bt(g, shape4) - f * g2).max() / np.abs(g2).max())
        R.bond_step(W, LE, RE, lid, ds, opts, True)
    return worst


def main():
    for k, (name, cfg) in enumerate(CASES.items()):
        N, T, d, chi0, chimax, C, loss, bbopt, sep, iters, nsw, eta, bal = cfg
        ds, W0 = inputs(name, N, T, d, chi0, C, bal, seed=100 + k)
        opts = R.SweepOptions(nsweeps=nsw, chi_max=chimax, eta=eta, update_iters=iters, loss_grad=loss, bbopt=bbopt,
                              train_classes_separately=sep)
        worst = cross_validate(W0, ds, opts)
        assert worst < 1e-12, (name, worst)
        rec = []
        Wf, info = R.fit(W0, ds, None, opts, record=rec)
        bonds = [b for sw in rec for b in sw]
        smax = max(len(b["S"]) for b in bonds)
        S = np.zeros((len(bonds), smax))
        for i, b in enumerate(bonds):
            S[i, :len(b["S"])] = b["S"]
        out = dict(
            phi=ds.phi, label_index=ds.label_index, class_distribution=ds.class_distribution,
            opts=np.array([chimax, iters, nsw, int(sep)], dtype=np.int64), eta=eta, loss=loss, bbopt=bbopt,
            bond_loss=np.array([b["loss"] for b in bonds]), bond_grad_norm=np.array([b["grad_norm"] for b in bonds]),
            bond_bt_norm=np.array([b["bt_new_norm"] for b in bonds]), bond_chi=np.array([b["chi"] for b in bonds]),
            bond_lid=np.array([b["lid"] for b in bonds]), bond_left=np.array([b["going_left"] for b in bonds]),
            bond_S=S, train_loss=np.array(info["train_loss"]), train_KL_div=np.array(info["train_KL_div"]),
            train_acc=np.array(info["train_acc"]), overlaps=R.contract_mps(Wf, ds.phi),
            pred=R.classify(Wf, ds.phi), final_chi=np.array([1] + [t.shape[2] for t in Wf]))
        for j, t in enumerate(W0):
            out[f"W0_{j}"] = t
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(f"{name}: cross-check {worst:.1e}, train KLD {info['train_KL_div']}, {len(bonds)} bonds")


if __name__ == "__main__":
    main()
