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


def bond_of(q, T):
    """Bond q of a sweep in sweep order (backward half-sweep first): (lid, going_left)."""
    going_left = q < T - 1
    return ((T - 2 - q) if going_left else (q - (T - 1))), going_left


def teacher_forced_sweep(eng, co, phi, T, nbonds=None, overlap_every=9, sub=slice(None), first=0):
    """Every bond update of a sweep compared with the C oracle, each starting from the ORACLE's state
    (set_mps + build_caches): free-running trajectories diverge chaotically (oracle/sensitivity_study.py),
    one update from a common state is well conditioned.  Returns the worst relative deviations and the
    number of bonds whose kept dimension differs by one because a singular value sits on the cutoff."""
    worst = dict(loss=0.0, grad=0.0, S=0.0, overlap=0.0)
    chi_flips = 0
    nb = 2 * (T - 1) - first if nbonds is None else nbonds
    for q in range(first, first + nb):
        lid, going_left = bond_of(q, T)
        eng.set_mps(co.get_mps())
        eng.build_caches()
        ref = co.sweep(max_bonds=1, first_bond=q, record=True)["bonds_rec"][0]
        tr = eng.bond_step(lid, going_left)
        worst["loss"] = max(worst["loss"], abs(tr["loss"] - ref["loss"]) / max(1.0, abs(ref["loss"])))
        worst["grad"] = max(worst["grad"], abs(tr["grad_norm"] - ref["grad_norm"]) / ref["grad_norm"])
        nk = min(tr["chi"], ref["chi"])
        worst["S"] = max(worst["S"], np.abs(tr["S"][:nk] - ref["S"][:nk]).max() / ref["S"][0])
        if tr["chi"] != ref["chi"]:
            # only a singular value sitting on the cutoff may be decided differently
            P = ref["S"] ** 2 / np.sum(ref["S"] ** 2)
            lo, hi = sorted((tr["chi"], ref["chi"]))
            tail = P[lo:].sum()
            assert hi - lo == 1 and abs(tail - 1e-10) < 1e-3 * 1e-10, (q, lid, tr["chi"], ref["chi"], tail)
            chi_flips += 1
        elif q % overlap_every == 0:
            yo = R.contract_mps(co.get_mps(), phi[sub])
            yg = R.contract_mps(eng.get_mps(), phi[sub])
            worst["overlap"] = max(worst["overlap"], np.abs(yo - yg).max() / np.abs(yo).max())
    return worst, chi_flips


def teacher_forced_segment(eng, make_oracle, W0, phi, T, first, count, sub=slice(None), overlap_every=5):
    """Bonds [first, first+count) of the FIRST sweep compared one by one with the C oracle.  The engine free-runs from
    W0 up to bond `first` (cheap on the GPU); the oracle is then initialised from the engine's state (label site
    wherever the sweep has carried it, environments rebuilt on both sides of it) and from there on every update of both
    starts from the oracle's state, as in teacher_forced_sweep."""
    eng.set_mps(W0)
    eng.build_caches()
    for q in range(first):
        eng.bond_step(*bond_of(q, T))
    co = make_oracle(eng.get_mps() if first else W0)
    co.build_caches(around_label=True)
    return teacher_forced_sweep(eng, co, phi, T, nbonds=count, overlap_every=overlap_every, sub=sub, first=first)
