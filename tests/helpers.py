"""Shared builders for parity tests: seeded synthetic inputs in the oracle's conventions."""
import numpy as np

from oracle import ref_numpy as R


def make_problem(N, T, d, chi_init, C, seed=0, balanced=True, encoding="legendre"):
    """Seeded data set + initial MPS.  Labels are ragged when balanced=False."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(-1, 1, (N, T))
    if balanced:
        y = np.arange(N) % C
    else:
        y = rng.integers(0, C, N)
        y[:C] = np.arange(C)          # every class present
    rng.shuffle(y)
    ds = R.encode_dataset(X, X, y, lambda x: R.legendre_encode(x, d), (-1, 1))
    W = R.random_mps(T, d, chi_init, C, np.random.default_rng(seed + 1000))
    return ds, W


def load_engine(eng, ds, W, opts: R.SweepOptions, test=None, **kw):
    C = len(ds.class_distribution)
    eng.set_options(chi_max=opts.chi_max, eta=opts.eta, cutoff=opts.cutoff, update_iters=opts.update_iters,
                    loss=opts.loss_grad, bbopt=opts.bbopt, rescale=opts.rescale,
                    train_classes_separately=opts.train_classes_separately, **kw)
    eng.set_dataset(0, ds.phi, ds.label_index, C)
    if test is not None:
        eng.set_dataset(1, test.phi, test.label_index, C)
    eng.set_mps(W)


def bond_matrix(Wl, Wr):
    """Gauge-invariant two-site tensor (s_l, a, s_r, b, c)."""
    bt, shape4 = R.flatten_bt(Wl, Wr)
    return R.unflatten_bt(bt, shape4)
