"""Step 0 of the subspace-eigensolver question: run real sweeps with the NumPy restatements
(oracle/, test infrastructure) on the bench's generators and dump, per bond, the updated bond matrix M (rows x cols as
decomposeBT sees it) and the row space known BEFORE the solve (the label-carrying site, rows = (bond, class)).
usage: python lab/subspace/dump_grams.py N T chi d sweeps out.npz [f64|real1|fourier] [every]
  f64: two classes, Legendre (headline generator); real1: one class, Legendre; fourier: one class, complex128 through
  oracle/ref_complex.py (BASELINE configs[4]'s training shape); every: keep every k-th bond."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ref_numpy as R, ref_complex as RC
import bench

N, T, chi, d, sweeps, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
mode = sys.argv[7] if len(sys.argv) > 7 else "f64"
every = int(sys.argv[8]) if len(sys.argv) > 8 else 1
C = 2 if mode == "f64" else 1
ds_mt = bench.make_inputs(N, T, d) if mode == "f64" else bench.typed_inputs(N, T, d, C, mode == "fourier")
ds = R.EncodedSet(np.asarray(ds_mt.phi), np.asarray(ds_mt.label_index), np.asarray(ds_mt.class_distribution), None, None)
import mpstime_jl_amd as mt
W = mt.generate_startingMPS(4, T, d, C, 1234, np.complex128 if mode == "fourier" else np.float64)   # as bench.py
Rm = RC if mode == "fourier" else R
opts = R.SweepOptions(chi_max=chi, eta=0.01, cutoff=1e-10, update_iters=1)
rec, state, cnt = [], {}, [0]
orig_dec, orig_flat = Rm.decompose_bt, Rm.flatten_bt

def dec_hook(bt5, *a, **k):
    going_left = k.get("going_left", a[-1] if Rm is R else a[2] if len(a) > 2 else True)
    d_l, Da, d_r, Db, Cc = bt5.shape
    if going_left:
        M = bt5.transpose(1, 4, 0, 2, 3).reshape(Da * Cc * d_l, d_r * Db)
    else:
        M = bt5.transpose(3, 4, 2, 0, 1).reshape(Db * Cc * d_r, d_l * Da)
    res = orig_dec(bt5, *a, **k)
    if cnt[0] % every == 0:
        rec.append(dict(M=M.copy(), gl=going_left, S=np.asarray(res[2]).copy(), sweep=state["sweep"], R0=state["R0"], lid=state["lid"]))
    cnt[0] += 1
    return res

def flat_hook(Wl, Wr):
    if Wr.ndim == 4:   # going left: label on the right site (b, s_r, r, c); columns of M = (s_r, r)
        Db, dd, Dr, Cc = Wr.shape
        state["R0"] = Wr.transpose(0, 3, 1, 2).reshape(Db * Cc, dd * Dr).copy()
    else:              # going right: label on the left site (a, s, k, c); columns of M = (s_l, a)
        Da, dd, Dk, Cc = Wl.shape
        state["R0"] = Wl.transpose(2, 3, 1, 0).reshape(Dk * Cc, dd * Da).copy()
    return orig_flat(Wl, Wr)

orig_bs = Rm.bond_step
def bs_hook(W, LE, RE, lid, *a, **k):
    state["lid"] = lid
    return orig_bs(W, LE, RE, lid, *a, **k)
Rm.decompose_bt, Rm.flatten_bt, Rm.bond_step = dec_hook, flat_hook, bs_hook
LE = RE = None
t0 = time.time()
for s in range(sweeps):
    state["sweep"] = s
    LE, RE = Rm.sweep(W, ds, opts, LE, RE)
    print("sweep", s, time.time() - t0, flush=True)
np.savez_compressed(out, **{f"M{i}": r["M"] for i, r in enumerate(rec)}, **{f"R{i}": r["R0"] for i, r in enumerate(rec)},
                    **{f"S{i}": r["S"] for i, r in enumerate(rec)},
                    meta=np.array([[r["lid"], int(r["gl"]), r["sweep"]] for r in rec]))
