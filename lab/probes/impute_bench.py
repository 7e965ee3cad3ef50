"""Throughput of the device imputation engine for the four element types (real / complex model x fp64 / fp32 chain
arithmetic) and the fp32 tolerance study of BASELINE configs[4] (imputation, N=8192, T=200, chi=64, d=8, Fourier).
Random MPSs are brought to left-canonical form first (a random chain of 200 sites otherwise over/underflows the
unscaled NumPy restatement; the engine rescales at every site).  A model TRAINED with the sweep engine (real Legendre,
the only kind the array path trains) gives the study its meaningful half: error against the held-out truth in both
precisions.  Writes a JSON summary when --out is given."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import mpstime_jl_amd as mt
from oracle import ref_numpy as R
from oracle import impute_numpy as I

ap = argparse.ArgumentParser()
ap.add_argument("--out", default="")
ap.add_argument("--quick", action="store_true")
ap.add_argument("--oracle-sample", type=int, default=4)
a = ap.parse_args()


def random_chain(T, d, chi, cx, rng):
    """Gaussian site tensors in left-canonical form, the carried factor renormalised at every site (its norm alone
    overflows over 200 sites); label index (one class) on the last site."""
    dims = [1] + [int(min(chi, d ** min(j, T - j, 30))) for j in range(1, T)] + [1]
    t = []
    for j in range(T):
        w = rng.standard_normal((dims[j], d, dims[j + 1]))
        if cx:
            w = w + 1j * rng.standard_normal(w.shape)
        t.append(w)
    for j in range(T - 1):
        l, d_, r = t[j].shape
        q, rr = np.linalg.qr(t[j].reshape(l * d_, r))
        t[j] = q.reshape(l, d_, q.shape[1])
        t[j + 1] = np.einsum("kb,bsc->ksc", rr / np.linalg.norm(rr), t[j + 1])
    t[-1] = t[-1] / np.linalg.norm(t[-1])
    assert all(np.all(np.isfinite(x)) for x in t)
    return [w if k < T - 1 else w[..., None] for k, w in enumerate(t)]


def block_mask(N, T, frac, rng):
    m = np.zeros((N, T), dtype=np.uint8)
    nm = int(round(T * frac))
    for i in range(N):
        s = rng.integers(0, T - nm + 1)
        m[i, s:s + nm] = 1
    return m


def problem(N, T, d, chi, cx, frac=0.5, seed=0):
    rng = np.random.default_rng(seed)
    W = random_chain(T, d, chi, cx, rng)
    xs = -1.0 + 1e-4 * np.arange(20001)
    enc = (lambda x: R.fourier_encode(x, d)) if cx else (lambda x: R.legendre_encode(x, d))
    X = rng.uniform(-0.95, 0.95, (N, T))
    return W, xs, enc(xs), enc(X), block_mask(N, T, frac, rng)


def deviation(x, ref, mk):
    dv = np.abs(x - ref)[mk]
    return dict(identical_frac=float(np.mean(dv < 1e-12)), within_1_step=float(np.mean(dv <= 1.0000001e-4)),
                within_10_steps=float(np.mean(dv <= 1.0000001e-3)), p99=float(np.quantile(dv, 0.99)), max=float(dv.max()), mean=float(dv.mean()))


res = []
eng = mt.SweepEngine(0)
cases = [(4096, 100, 4, 32), (1024, 200, 8, 64)] if a.quick else [(4096, 100, 4, 32), (1024, 200, 8, 64), (8192, 200, 8, 64)]
for (N, T, d, chi) in cases:
    for cx in (False, True):
        W, xs, gphi, phi, m = problem(N, T, d, chi, cx)
        lab = np.zeros(N, dtype=np.int32)
        sites = int(m.sum())
        mk = m.astype(bool)
        ref = None
        for compute in ("f64", "f32"):
            if False:
                continue
            eng.impute_model(W, phi[:8], lab[:8], m[:8], xs, gphi, 0, True, compute=compute)      # warm-up (attributes, allocations)
            t0 = time.perf_counter()
            x, e, secs = eng.impute_model(W, phi, lab, m, xs, gphi, 0, True, compute=compute)
            wall = time.perf_counter() - t0
            row = dict(kind="random", N=N, T=T, d=d, chi=chi, complex=cx, compute=compute, missing_sites=sites, device_ms=secs * 1e3,
                       wall_ms=wall * 1e3, site_imputations_per_s=sites / secs)
            if compute == "f64":
                ref = x
            elif ref is not None:
                row["vs_f64"] = deviation(x, ref, mk)
            if cx and chi > 48 and a.oracle_sample:
                # no fp64 device run at this size: a sample of instances against the NumPy complex128 restatement
                cls = [w.reshape(w.shape[:3]) for w in W]
                sel = np.zeros_like(mk)
                xo_all = np.zeros_like(x)
                for i in range(a.oracle_sample):
                    s_ = np.flatnonzero(m[i])
                    xo, _ = I.impute(cls, phi[i], s_, xs, gphi, "median")
                    xo_all[i, s_] = xo
                    sel[i, s_] = True
                row["vs_numpy_c128_sample"] = dict(instances=a.oracle_sample, **deviation(x, xo_all, sel))
            print(json.dumps(row), flush=True)
            res.append(row)

# a trained model: two classes of noisy sinusoids, 3 sweeps of the sweep engine, imputation of a 50 % block of held-out series
rng = np.random.default_rng(5)
T, d, chi, Ntr, Nte = 100, 4, 32, 2048, 1024
t = np.linspace(0, 1, T)


def make(Nn):
    y = rng.integers(0, 2, Nn)
    X = np.sin(2 * np.pi * ((2 + y[:, None]) * t[None, :] + rng.uniform(size=(Nn, 1)))) + 0.1 * rng.normal(size=(Nn, T))
    return X, y


Xtr, ytr = make(Ntr)
Xte, yte = make(Nte)
opts = mt.MPSOptions(d=d, chi_max=chi, nsweeps=3, verbosity=-1, sigmoid_transform=False)
trained, info, _ = mt.fitMPS(Xtr, ytr, Xte, yte, opts)
imp = mt.init_imputation_problem(trained, Xte, yte, verbosity=0)
mask = block_mask(Nte, T, 0.5, rng).astype(bool)
out = {}
for compute in ("f64", "f32"):
    ts, err, secs = mt.impute_dataset(imp, mask, "median", compute=compute, return_seconds=True)
    out[compute] = ts
    row = dict(kind="trained", N=Nte, T=T, d=d, chi=chi, complex=False, compute=compute, missing_sites=int(mask.sum()), device_ms=secs * 1e3,
               site_imputations_per_s=int(mask.sum()) / secs, test_accuracy=float(info["test_acc"][-1]),
               mae_vs_truth=float(np.abs(ts - Xte)[mask].mean()), mae_flat_baseline=float(np.abs(np.mean(Xtr) - Xte)[mask].mean()))
    if compute == "f32":
        dv = np.abs(out["f32"] - out["f64"])[mask]
        row["vs_f64_original_units"] = dict(identical_frac=float(np.mean(dv < 1e-12)), p50=float(np.quantile(dv, 0.5)), p99=float(np.quantile(dv, 0.99)),
                                            max=float(dv.max()), mean=float(dv.mean()))
    print(json.dumps(row), flush=True)
    res.append(row)
eng.close()
if a.out:
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
