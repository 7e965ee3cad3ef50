#!/usr/bin/env python3
"""bench.py - full sweeps/sec of the MI355X sweep engine on BASELINE.json's headline config.

Workload (BASELINE.json configs[2], the one the metric is quoted on): synthetic two-class
"noisy trendy sine" (docs/src/classification.md:20-43 of the reference), N=4096 series of
length T=100, RobustSigmoid+MinMax preprocessing, Legendre d=4 encoding, chi_max=32, KLD loss,
TSGO eta=0.01, update_iters=1, rescale=(false,true), cutoff=1e-10, fp64.  One "step" = one full
sweep (src/Training/RealRealHighDimension.jl:727-808: 2(T-1) bond updates AND the two cache
rebuilds per sweep, :770,:804 - redundant recomputation with bit-identical results, which
--no-rebuild-caches leaves out; the line carries that figure as `rebuild_caches_off`).

With --gpus N>1 (launched by torch.distributed.run) the N=4096 series are sharded over the
ranks (strong scaling) and the bond gradient is all-reduced with RCCL once per optimiser step.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6   # MI355X FP64 matrix peak (AMD spec)
PEAK_HBM_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md
PEAK_FP32_MFMA_TFLOPS = 157.3  # f32-input MFMA (v_mfma_f32_16x16x4_f32), same guide


def make_inputs(N, T, d, seed=1):
    import mpstime_jl_amd as mt
    rng = np.random.default_rng(seed)
    half = N // 2
    X1, _ = mt.trendy_sine(T, half, period=(12.0, 15.0), slope=[-3.0, 0.0, 3.0], sigma=0.1, rng=rng)
    X2, _ = mt.trendy_sine(T, N - half, period=(16.0, 19.0), slope=[-3.0, 0.0, 3.0], sigma=0.1, rng=rng)
    X = np.concatenate([X1, X2])
    y = np.concatenate([np.ones(half, dtype=np.int64), 2 * np.ones(N - half, dtype=np.int64)])
    opts = mt.MPSOptions(d=d, encoding="Legendre", verbosity=-1)
    enc = mt.model_encoding(opts.encoding)
    Xs, _ = mt.transform_train_data(X, opts, enc.range)
    return mt.encode_dataset(X, Xs, y, enc, d, {1: 0, 2: 1})


def kernel_model(N, d, chi, C, info):
    """Algorithmic flops / bytes per launch of each kernel class at steady state (all bulk bonds at chi_max); formulas
    from SURVEY.md 8(d), stated again in DESIGN.md.  `info` = SweepEngine.info(): which launch chain is in use."""
    X = Y = d * chi
    m, n = chi * C * d, d * chi
    model = {
        "yhat": ("mfma", 2.0 * N * X * Y),                 # Z = X B_c, rowdot with Y
        "grad": ("mfma", 2.0 * N * X * Y),                 # G_c = X^T diag(w) Y
        # SVD of the (chi C d) x (d chi) bond matrix, done as Gram matrix + symmetric eigensolver:
        "gram": ("mfma", 2.0 * m * n * n),
        "eig_tri": ("mfma", 4.0 / 3.0 * n ** 3),            # Householder tridiagonalisation (dsytd2 count)
        "eig_vec": ("mfma", 4.0 * n * n * chi),             # back-transformation of chi vectors (+ O(n chi) bisection/twisted)
        "eig_fin": ("mfma", 8.0 * n * chi * chi),           # Gram + Loewdin update of the kept vectors
        "split": ("mfma", 2.0 * m * n * chi),
        "bt_assemble": ("mfma", 2.0 * C * X * chi * Y),
        "env": ("hbm", 8.0 * N * (chi + d + chi)),         # read env row + site vector, write new env row
    }
    # unfused chain: k_grad_reduce reads C * nsplit partial blocks and writes the gradient; k_grad_norm + k_update read it
    # twice and read + write the bond tensor.  The profile slot holds two scopes per optimiser step (reduce | norm + update),
    # so the per-scope average is half of the sum.  nsplit as in csrc/mpst_internal.h: grad_nsplit.
    dm = d * chi
    nb = ((dm + 63) // 64) ** 2
    nsplit = min(max(1, info.get("nchunks", 1)), max(1, -(-512 // (nb * C))))
    # bytes per scope: k_grad_reduce reads nsplit partial blocks per (class, output block) and writes the gradient;
    # k_grad_norm + k_update read the gradient twice and read + write the bond tensor: (nsplit + 1) and 4 passes over
    # C X Y doubles - the average of the two scopes (the slot's launches_per_sweep counts both)
    model["grad_reduce+update"] = ("hbm", 0.5 * 8.0 * ((nsplit + 1.0) + 4.0) * C * X * Y)
    if info.get("fused") and info.get("sliced_bond_gemms"):
        # k_yhat_s / k_grad_s: one GEMM each (2 N X Y), no partial gradients per workgroup; the shares of a gradient block
        # (grad_shares of them) meet inside k_grad_s
        model.pop("grad_reduce+update", None)
    elif info.get("fused"):
        P = info["nparts"]
        model["grad"] = ("mfma", 4.0 * N * X * Y)           # k_bond_fused: yhat AND the gradient partials
        model["grad_reduce+update"] = ("hbm", 8.0 * (P + C) * X * Y)   # k_fused_reduce: read P partials, write the gradient
    if info.get("fused"):
        model["env"] = ("hbm", 8.0 * N * (chi + d + chi))  # k_env_split (+ 2mn chi flops of the back-split, + next bond tensor)
    if info.get("four_launch_chain"):
        # k_bond_tail: the next bond's yhat GEMM (2 N X Y) + projection and environment rows (6 N X chi) + back-split and next tensor; the
        # verification + polish every workgroup repeats is not algorithmic work
        model["env"] = ("mfma", 2.0 * N * X * Y + 6.0 * N * X * chi + 2.0 * m * n * chi + 2.0 * C * X * chi * Y)
        model.pop("yhat", None)
        model.pop("eig_fin", None)
    if info.get("eig_merged") and not info.get("large_bond"):
        # k_eig_trivec: every eigenvector workgroup repeats the tridiagonalisation; the algorithmic count has it once
        model["eig_tri"] = ("mfma", 4.0 / 3.0 * n ** 3 + 4.0 * n * n * chi)
        model.pop("eig_vec", None)
    if info.get("large_bond"):
        model["eig_tri"] = ("mfma", 4.0 / 3.0 * n ** 3 + 4.0 * n ** 3)   # (dense symmetric eigensolver count: reduction + back-transformation)
    return model


def csrc_sha256():
    """Digest of the kernel sources of this tree: ties quoted profile figures to the code they were measured on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "mpstime.jl_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "mpstime.jl_amd", "csrc", "*.h")) +
                    glob.glob(os.path.join(ROOT, "mpstime.jl_amd", "csrc", "*.inl"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def host_cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def random_chain(T, d, chi, rng):
    """Complex Gaussian site tensors in left-canonical form (the carried factor renormalised at every site), one class,
    label index on the last site: the shape of a Fourier-encoded MPS out of the reference's legacy trainer."""
    dims = [1] + [int(min(chi, d ** min(j, T - j, 30))) for j in range(1, T)] + [1]
    t = [rng.standard_normal((dims[j], d, dims[j + 1])) + 1j * rng.standard_normal((dims[j], d, dims[j + 1])) for j in range(T)]
    for j in range(T - 1):
        l, d_, r = t[j].shape
        q, rr = np.linalg.qr(t[j].reshape(l * d_, r))
        t[j] = q.reshape(l, d_, q.shape[1])
        t[j + 1] = np.einsum("kb,bsc->ksc", rr / np.linalg.norm(rr), t[j + 1])
    t[-1] = t[-1] / np.linalg.norm(t[-1])
    return [w if k < T - 1 else w[..., None] for k, w in enumerate(t)]


def impute_workload(args, mt, torch, dist, world, rank, dev_index, host_reduce):
    """BASELINE configs[4]: median imputation (+ WMAD) of a 50 % block of every instance on the reference's 20 001-value grid.
    Instances are independent: every rank imputes its own N instances (weak scaling, no collective on the data path).  A
    step is one pass over the rank's instances; the time of a step is the device time between the first and the last
    kernel of the pass (operands resident), the PCIe-inclusive wall time is reported beside it."""
    defaults = ap_defaults()
    N = args.N if args.N != defaults["N"] else 8192
    T = args.T if args.T != defaults["T"] else 200
    chi = args.chi if args.chi != defaults["chi"] else 64
    d = args.d if args.d != defaults["d"] else 8
    rng = np.random.default_rng(100 + rank)
    W = random_chain(T, d, chi, np.random.default_rng(7))
    enc = mt.model_encoding("Fourier")
    xs = -1.0 + 1e-4 * np.arange(20001)
    gphi = np.ascontiguousarray(enc.encode(xs, d), dtype=np.complex128)
    X = rng.uniform(-0.95, 0.95, (N, T))
    phi = np.ascontiguousarray(enc.encode(X, d), dtype=np.complex128)
    m = np.zeros((N, T), dtype=np.uint8)
    for i in range(N):
        s0 = rng.integers(0, T - T // 2 + 1)
        m[i, s0:s0 + T // 2] = 1
    lab = np.zeros(N, dtype=np.int32)
    sites = int(m.sum())
    eng = mt.SweepEngine(dev_index)
    for _ in range(args.warmup):
        eng.impute_model(W, phi, lab, m, xs, gphi, 0, True, compute="f32")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dev_s = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x, e, secs = eng.impute_model(W, phi, lab, m, xs, gphi, 0, True, compute="f32")
        dev_s += secs
    eng_phases = eng.impute_phases()
    closed_form = eng.impute_info()["closed_form_densities"]
    batched = eng.impute_info()["batched_sweep"]
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
        dev_s, wall = host_reduce([dev_s, wall], dist.ReduceOp.MAX)
    eng.close()
    ms_step = 1e3 * dev_s / args.steps
    value = world * sites * args.steps / dev_s
    t_env, t_den = eng_phases            # seconds of the last pass on this rank
    ngrid = len(xs)
    known = N * T - sites
    # k_imp_right (fp32 MFMA).  ALGORITHMIC count (the reference's arithmetic, MPS_methods.jl:42-99: every site of the chain enters the
    # environment as full complex matrix products, four real products each): per missing site d x 2 complex chi^3 products, per
    # known site 2.
    flops_env_alg = 8.0 * chi ** 3 * (2.0 * d * sites + 2.0 * known)
    # EXECUTED count - what the kernel issues to the matrix pipe, and what `achieved` / `frac` are computed from: complex products in
    # the 3M form (three real products: 6 chi^3 flops); the pass walks from the far end of the chain to the LAST missing site it meets
    # and stops there (no product at that site, none beyond it: the vector pass of k_imp_left owns those sites); the known sites it
    # meets BEFORE its first missing site are a vector recursion r <- M_j r on the vector ALUs (8 d chi^2 flops a site, not matrix work)
    prod_sites, vec_sites = 0.0, 0.0
    for i in range(N):
        nz = np.flatnonzero(m[i])
        if nz.size == 0:
            continue
        first, last = int(nz[0]), int(nz[-1])          # the pass runs from site T-1 down to `first`
        vec_sites += T - 1 - last
        inner = m[i, first + 1:last + 1]               # the sites it multiplies through: (first, last]
        prod_sites += 2.0 * d * float(inner.sum()) + 2.0 * float((inner == 0).sum())
    flops_env = 6.0 * chi ** 3 * prod_sites
    flops_env_vec = 8.0 * d * chi ** 2 * vec_sites
    # k_imp_left: the density on the grid, ngrid x (d^2 complex MACs + d squares) fp64 VALU flops per missing site, and its
    # streams (write p, read p, write S, ~2 reads of S for the selections)
    flops_den = sites * ngrid * (8.0 * d * d + 4.0 * d)
    bytes_den = sites * ngrid * 8.0 * 5.0
    dom_env = t_env >= t_den
    roof_env = {"kernel": "k_imp_right<float, complex>", "bound": "mfma", "achieved": flops_env / t_env / 1e12, "peak": PEAK_FP32_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": flops_env / t_env / 1e12 / PEAK_FP32_MFMA_TFLOPS, "traffic": None, "avg_ms": 1e3 * t_env,
                "flops": "executed: 3M complex products (6 chi^3 flops), only the sites between an instance's first and last missing site; the known "
                         "sites ahead of them are a vector recursion on the vector ALUs (%.3g flops, not counted)" % flops_env_vec,
                "algorithmic": {"flops": flops_env_alg, "achieved": flops_env_alg / t_env / 1e12, "frac": flops_env_alg / t_env / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                                "note": "the reference's arithmetic: 4M complex products (8 chi^3 flops), every site of the chain a matrix product; a "
                                        "figure of merit for the formulation, NOT a fraction of the matrix pipe's peak (it can exceed what the pipe executes)"}}
    roof_den = {"kernel": "k_imp_left<float, complex>", "bound": "hbm", "achieved": bytes_den / t_den / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": bytes_den / t_den / 1e9 / PEAK_HBM_GBS, "traffic": None, "avg_ms": 1e3 * t_den,
                "note": f"its density loop does {flops_den / t_den / 1e12:.2f} TFLOP/s of fp64 VALU work (vector peak 78.6) at one workgroup per CU; the p / prefix-sum streams are the HBM figure"}
    if closed_form:
        # Fourier states on a uniform grid: densities and cumulative sums come from 2d coefficients, no table, no p / S streams.  What
        # is left per instance is a dependent chain of T matrix-vector products (L W_j: the site tensors come from the L2) plus, per
        # missing site, one read of its chi x chi environment (the only HBM stream) and O(log ngrid) closed-form evaluations
        bytes_den = sites * chi * chi * 8.0 + N * T * d * 8.0
        roof_den = {"kernel": "k_imp_left<float, complex, closed-form densities>", "bound": "hbm", "achieved": bytes_den / t_den / 1e9, "peak": PEAK_HBM_GBS,
                    "unit": "GB/s", "frac": bytes_den / t_den / 1e9 / PEAK_HBM_GBS, "traffic": None, "avg_ms": 1e3 * t_den,
                    "note": "latency chain: one workgroup per instance walks its T sites (matrix-vector products against site tensors served by the L2, "
                            "then the selections on 2d Fourier coefficients); bytes = the environments of the missing sites + the encoded series"}
        if batched:
            roof_den["kernel"] = "k_imp_leftb<float, complex, [Re; Im] rows> (16 instances per workgroup, closed-form densities)"
            roof_den["note"] = ("a workgroup walks the chain with 16 instances in step: L W_j for all of them and LW R_b per missing instance on the "
                                "matrix pipe (operands straight from memory), a wave per instance for the closed-form selections; bytes = the "
                                "environments of the missing sites (read once: the stream that bounds the kernel's product phase) + the encoded series")
    line = {"metric": "site-imputations/sec (imputation engine, BASELINE configs[4])", "value": value, "unit": "site-imputations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "c64 model; f32 chain contractions (MFMA f32 16x16x4), f64 densities",
            "data": "synthetic",
            "config": {"workload": f"median imputation + WMAD of a 50 % block, N={N} instances per GPU, T={T}, chi={chi}, d={d} Fourier "
                                   f"(complex random canonical MPS), 20001-value grid", "parallelism": f"instances sharded over {world} GPU(s), no collective"},
            "roofline": roof_env if dom_env else roof_den, "roofline_other_kernel": roof_den if dom_env else roof_env,
            "closed_form_densities": bool(closed_form), "batched_sweep": bool(batched),
            "wall_ms_per_step_incl_pcie_and_host_packing": 1e3 * wall / args.steps}
    if rank == 0 and world == 1:
        # side figure: a real model on the reference's default basis (Legendre, d = 12, chi = 40: the size of its imputation
        # examples), same missing pattern and grid; its densities run in closed form as well (Legendre series, Euler-Maclaurin)
        try:
            N2, T2, d2, chi2 = 2048, 100, 12, 40
            W2 = [np.ascontiguousarray(w.real) for w in random_chain(T2, d2, chi2, np.random.default_rng(11))]
            for j in range(T2 - 1):                       # left-canonical again after dropping the imaginary parts
                l_, dd_, r_ = W2[j].shape[:3]
                q_, rr_ = np.linalg.qr(W2[j].reshape(l_ * dd_, r_))
                W2[j] = q_.reshape(l_, dd_, q_.shape[1])
                W2[j + 1] = np.einsum("kb,bsc...->ksc...", rr_ / np.linalg.norm(rr_), W2[j + 1])
            W2[-1] = W2[-1] / np.linalg.norm(W2[-1])
            enc2 = mt.model_encoding("Legendre")
            rng2 = np.random.default_rng(12)
            X2 = rng2.uniform(-0.95, 0.95, (N2, T2))
            phi2 = np.ascontiguousarray(enc2.encode(X2, d2), dtype=np.float64)
            g2 = np.ascontiguousarray(enc2.encode(xs, d2), dtype=np.float64)
            m2 = np.zeros((N2, T2), dtype=np.uint8)
            for i in range(N2):
                s0 = rng2.integers(0, T2 - T2 // 2 + 1)
                m2[i, s0:s0 + T2 // 2] = 1
            eng2 = mt.SweepEngine(dev_index)
            lab2 = np.zeros(N2, dtype=np.int32)
            eng2.impute_model(W2, phi2[:8], lab2[:8], m2[:8], xs, g2, 0, True, compute="f32")
            _, _, s2 = eng2.impute_model(W2, phi2, lab2, m2, xs, g2, 0, True, compute="f32")
            line["legendre_side"] = {"config": f"real model, Legendre d={d2}, chi={chi2}, N={N2}, T={T2}, 50 % block, fp32 chain", "value": int(m2.sum()) / s2,
                                     "unit": "site-imputations/s", "ms": 1e3 * s2, "closed_form_densities": eng2.impute_info()["closed_form_densities"]}
            eng2.close()
        except Exception as e:                            # a side figure never takes the line down
            line["legendre_side"] = {"error": str(e)}
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import impute_numpy as I         # the checker, timed as the CPU baseline on a bounded sample
        cls = [w.reshape(w.shape[:3]) for w in W]
        t1 = time.perf_counter()
        done = 0
        agree = []
        for i in range(min(N, 64)):
            s_ = np.flatnonzero(m[i])
            xo, _ = I.impute(cls, phi[i], s_, xs, gphi, "median")
            agree.append(np.mean(np.abs(x[i, s_] - xo) < 1e-12))
            done += len(s_)
            if time.perf_counter() - t1 > 15.0:
                break
        dt = time.perf_counter() - t1
        line["cpu_baseline"] = {"value": done / dt, "unit": "site-imputations/s", "cores": 1, "kind": "port",
                                "sample": f"{done} missing sites of the same instances with the NumPy complex128 restatement "
                                          f"(oracle/impute_numpy.py) in {dt:.1f} s on {host_cpu_model()}",
                                "fp32_chain_vs_numpy_c128_identical_frac": float(np.mean(agree))}
    return line


NP_DT = {"f64": "float64", "f32": "float32", "c128": "complex128", "c64": "complex64"}


def typed_inputs(N, T, d, C, cx, seed=1):
    """Synthetic series of the headline generator, preprocessed as fitMPS does and encoded with the Fourier basis (complex element
    types, BASELINE configs[4]) or Legendre (real).  C = 1: one class, else two."""
    import mpstime_jl_amd as mt
    rng = np.random.default_rng(seed)
    half = N // 2 if C > 1 else N
    X1, _ = mt.trendy_sine(T, half, period=(12.0, 15.0), slope=[-3.0, 0.0, 3.0], sigma=0.1, rng=rng)
    if C > 1:
        X2, _ = mt.trendy_sine(T, N - half, period=(16.0, 19.0), slope=[-3.0, 0.0, 3.0], sigma=0.1, rng=rng)
        X = np.concatenate([X1, X2])
        y = np.concatenate([np.ones(half, dtype=np.int64), 2 * np.ones(N - half, dtype=np.int64)])
        keys = {1: 0, 2: 1}
    else:
        X, y, keys = X1, np.zeros(N, dtype=np.int64), {0: 0}
    opts = mt.MPSOptions(d=d, encoding="Fourier" if cx else "Legendre", verbosity=-1)
    enc = mt.model_encoding(opts.encoding)
    Xs, _ = mt.transform_train_data(X, opts, enc.range)
    return mt.encode_dataset(X, Xs, y, enc, d, keys)


def typed_kernel_model(N, d, chi, C, cx, esz):
    """Algorithmic flops / bytes per launch of the element-typed chain (csrc/mpst_typed.hip) at steady state; SURVEY 8(d)'s
    formulas with the GEMM terms x 4 for complex elements.  (bound, amount, peak)."""
    z = 4.0 if cx else 1.0
    X = Y = d * chi
    m, n = chi * C * d, d * chi
    ne = (2 if cx else 1) * n                     # order of the real symmetric problem the fp64 eigensolver gets
    f32peak = PEAK_FP32_MFMA_TFLOPS if esz in (4, 8) and esz // (2 if cx else 1) == 4 else PEAK_FP64_MFMA_TFLOPS
    return {
        "yhat": ("mfma", z * 2.0 * N * X * Y, f32peak),
        "grad": ("mfma", z * 2.0 * N * X * Y, f32peak),
        "env": ("mfma", z * 2.0 * N * X * chi, f32peak),
        "gram": ("mfma", z * 2.0 * m * n * n, PEAK_FP64_MFMA_TFLOPS),
        "split": ("mfma", z * 2.0 * m * n * chi, PEAK_FP64_MFMA_TFLOPS),
        "bt_assemble": ("mfma", z * 2.0 * C * X * chi * Y, PEAK_FP64_MFMA_TFLOPS),
        # complex: native Hermitian reduction (zhetd2 count = 4 x the real one at order n) + back-transformation of chi complex vectors
        "eig_tri": ("mfma", (16.0 / 3.0 * n ** 3 + 16.0 * n * n * chi) if cx else (4.0 / 3.0 * n ** 3 + 4.0 * n * n * chi), PEAK_FP64_MFMA_TFLOPS),
        "eig_fin": ("mfma", 8.0 * ne * chi * chi, PEAK_FP64_MFMA_TFLOPS),
        "grad_reduce+update": ("hbm", 0.5 * esz * 6.0 * C * X * Y, PEAK_HBM_GBS),
    }


def typed_workload(args, mt, torch, rank, dev_index):
    """Side line (never the headline `value` of BASELINE's metric): the training sweep in another element type - fp32, complex64,
    complex128 - through csrc/mpst_typed.hip; defaults to BASELINE configs[4]'s training shape.  Besides throughput and the
    per-kernel roofline it carries a tolerance study: from the state the timed sweeps end in, the same bond updates are run by
    the engine in this type and in its double-precision counterpart, and the gauge-invariant per-bond quantities compared."""
    dt = np.dtype(NP_DT[args.dtype])
    cx = dt.kind == "c"
    f32 = dt.itemsize // (2 if cx else 1) == 4
    defaults = ap_defaults()
    if all(getattr(args, k) == defaults[k] for k in defaults):
        args.N, args.T, args.chi, args.d = 8192, 200, 64, 8            # BASELINE configs[4]'s training shape
    N, T, d, chi = args.N, args.T, args.d, args.chi
    C = args.classes if args.classes else (1 if cx else 2)
    full = typed_inputs(N, T, d, C, cx)
    W0 = mt.generate_startingMPS(4, T, d, C, 1234, dt)
    eng = mt.SweepEngine(dev_index)
    eng.set_options(chi_max=chi, eta=0.01, cutoff=1e-10, update_iters=1, loss="KLD", bbopt="TSGO", rescale=(False, True))
    eng.set_dataset(0, full.phi, full.label_index, C, dtype=dt)
    eng.set_mps(W0)
    eng.build_caches()
    for _ in range(args.warmup):
        eng.sweep()
    info = eng.info()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dev_s = 0.0
    for _ in range(args.steps):
        dev_s += eng.sweep()["seconds"]
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    mse, kld, acc, _ = eng.eval(0)
    chi_now, _ = eng.get_chi()
    Wend = eng.get_mps()
    info = eng.info()              # after the timed sweeps: the counters of the eigensolver paths (subspace attempted / accepted, fall-backs)
    eng.set_profile(0x7FF)
    eng.sweep()
    breakdown = eng.get_profile()
    eng.set_profile(0)
    model = typed_kernel_model(N, d, chi, C, cx, dt.itemsize)
    kernels = {}
    for k, (us, cnt) in breakdown.items():
        if not cnt:
            continue
        e = {"us_per_sweep": us, "launches_per_sweep": cnt, "avg_us": us / cnt}
        if k in model:
            bound, amt, peak = model[k]
            ach = amt / (us / cnt * 1e-6) / (1e12 if bound == "mfma" else 1e9)
            e.update({"bound": bound, "achieved": ach, "unit": "TFLOP/s" if bound == "mfma" else "GB/s", "frac": ach / peak})
        kernels[k] = e
    dominant = max(kernels, key=lambda k: kernels[k]["us_per_sweep"])
    dk = kernels[dominant]
    # ---- tolerance study: the same bond updates in this type and in double precision, from a common state -----------------
    study = None
    if f32:
        wide = np.dtype(np.complex128 if cx else np.float64)
        ref = mt.SweepEngine(dev_index)
        try:
            ref.set_options(chi_max=chi, eta=0.01, cutoff=1e-10, update_iters=1, loss="KLD", bbopt="TSGO", rescale=(False, True))
            ref.set_dataset(0, full.phi.astype(dt).astype(wide), full.label_index, C, dtype=wide)     # the fp32-rounded inputs, held in fp64
            worst = {"loss": 0.0, "grad_norm": 0.0, "singular_values_rel_sigma1": 0.0, "chi_differs": 0}
            Wc = [t.astype(wide) for t in Wend]
            nb = min(args.study_bonds, T - 1)
            for e in (eng, ref):
                e.set_mps(Wc)
                e.build_caches()
            for q in range(nb):                       # the first nb bonds of a backward half-sweep, free running in both
                lid = T - 2 - q
                a, b = eng.bond_step(lid, True), ref.bond_step(lid, True)
                worst["loss"] = max(worst["loss"], abs(a["loss"] - b["loss"]) / max(1.0, abs(b["loss"])))
                worst["grad_norm"] = max(worst["grad_norm"], abs(a["grad_norm"] - b["grad_norm"]) / b["grad_norm"])
                nk = min(a["chi"], b["chi"])
                worst["singular_values_rel_sigma1"] = max(worst["singular_values_rel_sigma1"], float(np.abs(a["S"][:nk] - b["S"][:nk]).max() / b["S"][0]))
                worst["chi_differs"] += int(a["chi"] != b["chi"])
                if a["chi"] != b["chi"]:
                    break
                e_mps = eng.get_mps()
                ref.set_mps([t.astype(wide) for t in e_mps])      # teacher forcing: both continue from the fp32 engine's state
                ref.build_caches()
            study = {"bonds": nb, "reference": f"the same engine in {wide.name} on the fp32-rounded inputs and state", "max_rel_deviation": worst,
                     "note": "Gram matrix, eigensolver, losses, gradient reduction and the optimiser step are fp64 in every element type"}
        finally:
            ref.close()
    eng.close()
    if rank != 0:
        return None
    sweeps_per_s = args.steps / elapsed
    survey = {"flops_per_sweep_real": 2 * (T - 1) * (4.0 * N * (d * chi) ** 2 + 2.0 * N * d * chi * chi) + 4.0 * N * T * d * chi * chi}
    survey["flops_per_sweep_this_type"] = survey["flops_per_sweep_real"] * (4.0 if cx else 1.0)
    peak = PEAK_FP32_MFMA_TFLOPS if f32 else PEAK_FP64_MFMA_TFLOPS
    survey["mfma_roofline_sweeps_per_s"] = peak * 1e12 / survey["flops_per_sweep_this_type"]
    survey["frac_of_mfma_roofline"] = sweeps_per_s / survey["mfma_roofline_sweeps_per_s"]
    return {
        "metric": "full sweeps/sec (side line: element type %s)" % dt.name, "value": sweeps_per_s, "unit": "sweeps/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"training sweep N={N} T={T} chi={chi} d={d} C={C} {'Fourier' if cx else 'Legendre'} encoding, element type {dt.name} "
                               "(BASELINE configs[4] training shape)" , "N": N, "T": T, "chi_max": chi, "d": d, "classes": C,
                   "loss": "KLD", "optimiser": "TSGO", "eta": 0.01, "cutoff": 1e-10, "update_iters": 1, "rebuild_caches": False,
                   "semantics": "legacy ITensor engine of the reference (src/legacy_itensor/loss_functions.jl:433-640)" if cx else "array engine"},
        "roofline": {"kernel": dominant, "bound": dk.get("bound", "mfma"), "achieved": dk.get("achieved"), "peak": PEAK_FP64_MFMA_TFLOPS if dominant.startswith("eig") else peak,
                     "unit": dk.get("unit", "TFLOP/s"), "frac": dk.get("frac"), "traffic": None,
                     "avg_us": dk["avg_us"], "note": "dominant kernel class of one profiled sweep (HIP events on the engine's stream); the fp64 eigensolver of the "
                     "Gram matrix (complex: Hermitian reduction to a real tridiagonal matrix with complex reflectors, eigenvectors returned in the 2n embedding the split reads) is a dependent chain of Householder steps, not a throughput kernel"},
        "kernels": kernels,
        "sweep_vs_survey_8d": survey,
        "tolerance_study": study,
        "device_seconds_per_sweep": dev_s / args.steps, "bonds_per_sweep": 2 * (T - 1), "us_per_bond": 1e6 * dev_s / args.steps / (2 * (T - 1)),
        "chain": info, "max_chi": int(max(chi_now)), "train_kld_after": kld, "train_acc_after": acc,
        "csrc_sha256": csrc_sha256(),
    }


def tolerance_study(args, mt):
    """BASELINE configs[4]: "fp32 with tolerance study".  Two FREE-RUNNING fits from the same starting MPS and the same data, one in the
    narrow element type (float32 / complex64) and one in its double-precision counterpart, through the whole public path (fitMPS ->
    engine sweeps -> imputation): per-sweep train KLD / accuracy, test accuracy, bond dimensions, and the median imputation of a
    held-out 50 % block from both models against the truth.  The reference's own statement of what spread is tolerable:
    docs/src/imputation.md:62-64 (results vary by 1-2 % between machines)."""
    cx = args.dtype in ("c64", "c128")
    defaults = ap_defaults()
    if all(getattr(args, k) == defaults[k] for k in defaults):
        args.N, args.T, args.chi, args.d = 2048, 100, 32, 8
    N, T, d, chi = args.N, args.T, args.d, args.chi
    nsw = 5
    rng = np.random.default_rng(11)
    Nte = max(64, N // 4)
    def gen(n):
        h = n // 2
        X1, _ = mt.trendy_sine(T, h, period=(12.0, 15.0), slope=[-3.0, 0.0, 3.0], sigma=0.1, rng=rng)
        X2, _ = mt.trendy_sine(T, n - h, period=(16.0, 19.0), slope=[-3.0, 0.0, 3.0], sigma=0.1, rng=rng)
        return np.concatenate([X1, X2]), np.concatenate([np.ones(h, dtype=np.int64), 2 * np.ones(n - h, dtype=np.int64)])
    Xtr, ytr = gen(N)
    Xte, yte = gen(Nte)
    mask = np.zeros((Nte, T), dtype=bool)
    nm = T // 2
    for i in range(Nte):
        s0 = int(rng.integers(0, T - nm + 1))
        mask[i, s0:s0 + nm] = True
    names = {"f32": ("Float32", "Float64"), "c64": ("ComplexF32", "ComplexF64")}[args.dtype]
    fits = {}
    wide_np = np.complex128 if cx else np.float64
    # control: the wide type once more from a start that differs by 1e-12 (relative, Gaussian) - the spread two legitimate double-precision
    # runs show after the same number of sweeps (a DMRG sweep amplifies rounding-level differences: oracle/sensitivity_study.py)
    W0 = mt.generate_startingMPS(4, T, d, 2, 1234, wide_np)
    prng = np.random.default_rng(99)
    W0p = [t * (1.0 + 1e-12 * prng.standard_normal(t.shape)) for t in W0]
    for name in names + ("control",):
        dname = names[1] if name == "control" else name
        opts = mt.MPSOptions(d=d, chi_max=chi, nsweeps=nsw, eta=0.01, encoding="Fourier" if cx else "Legendre", dtype=dname, verbosity=-1,
                             chi_init=4, init_rng=1234, exit_early=False)
        t0 = time.perf_counter()
        trained, info, _ = mt.fitMPS(Xtr, ytr, Xte, yte, opts, W=(W0p if name == "control" else [t.astype(mt.options.numpy_dtype(dname)) for t in W0]))
        fit_s = time.perf_counter() - t0
        imp = mt.init_imputation_problem(trained, Xte, yte, verbosity=0)
        compute = "f32" if name in ("Float32", "ComplexF32") else "f64"
        ts, err, secs = mt.impute_dataset(imp, mask, "median", compute=compute, return_seconds=True)
        chis = [int(t.shape[2]) for t in trained.mps[:-1]]
        fits[name] = dict(info=info, chis=chis, imputed=ts, fit_s=fit_s, impute_device_s=secs)
    n_, w_ = names
    a, b, ctl = fits[n_], fits[w_], fits["control"]
    def col(k):
        return [float(x) for x in a["info"][k]], [float(x) for x in b["info"][k]]
    kn, kw = col("train_KL_div")
    an, aw = col("train_acc")
    tn, tw = col("test_acc")
    mae_n = float(np.abs(a["imputed"] - Xte)[mask].mean())
    mae_w = float(np.abs(b["imputed"] - Xte)[mask].mean())
    mae_c = float(np.abs(ctl["imputed"] - Xte)[mask].mean())
    dvc = np.abs(ctl["imputed"] - b["imputed"])[mask]
    kc = [float(x) for x in ctl["info"]["train_KL_div"]]
    tc = [float(x) for x in ctl["info"]["test_acc"]]
    dv = np.abs(a["imputed"] - b["imputed"])[mask]
    flat = float(np.abs(np.mean(Xtr) - Xte)[mask].mean())
    span = float(Xte.max() - Xte.min())
    rows = [{"after_sweep": i, "train_KL_div": {n_: kn[i], w_: kw[i], "abs_diff": abs(kn[i] - kw[i]), "control_abs_diff": abs(kc[i] - kw[i])},
             "train_acc": {n_: an[i], w_: aw[i]}, "test_acc": {n_: tn[i], w_: tw[i], "control": tc[i]}} for i in range(min(len(kn), nsw + 1))]
    return {
        "metric": f"tolerance study {n_} vs {w_}: free-running {nsw}-sweep fits + median imputation (N={N}, T={T}, chi={chi}, d={d}) - a side study, "
                  "NOT the headline metric", "value": None, "unit": None, "n_gpus": 1, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"fitMPS(encoding={'Fourier' if cx else 'Legendre'}, d={d}, chi_max={chi}, nsweeps={nsw}, eta=0.01) on {N} noisy trendy sines "
                               f"(2 classes), {Nte} held-out series, a contiguous 50 % block of each imputed (median) from the trained model; "
                               "both fits start from generate_startingMPS(chi_init=4, init_rng=1234) and see the same encoded data"},
        "per_sweep": rows,
        "bond_dimensions": {"equal": a["chis"] == b["chis"], "max_abs_diff": int(np.abs(np.array(a["chis"]) - np.array(b["chis"])).max()),
                            "sum": {n_: int(sum(a["chis"])), w_: int(sum(b["chis"]))}},
        "control": {"what": f"{w_} again from a starting MPS perturbed by 1e-12 (relative): the spread of two legitimate double-precision runs",
                    "final_train_KL_div_rel_diff": abs(kc[-1] - kw[-1]) / max(1.0, abs(kw[-1])), "mae_vs_truth": mae_c, "mae_rel_diff": abs(mae_c - mae_w) / mae_w,
                    "imputations_mean_abs_diff_over_span": float(dvc.mean()) / span,
                    "bond_dimensions_equal": ctl["chis"] == b["chis"]},
        "imputation": {"missing_sites": int(mask.sum()), "mae_vs_truth": {n_: mae_n, w_: mae_w, "rel_diff": abs(mae_n - mae_w) / mae_w},
                       "mae_flat_mean_baseline": flat, "data_span": span,
                       "narrow_vs_wide_imputations": {"mean_abs": float(dv.mean()), "p50": float(np.quantile(dv, 0.5)), "p99": float(np.quantile(dv, 0.99)),
                                                      "max": float(dv.max()), "mean_abs_over_span": float(dv.mean()) / span}},
        "final": {"train_KL_div_rel_diff": abs(kn[-1] - kw[-1]) / max(1.0, abs(kw[-1])), "test_acc": {n_: tn[-1], w_: tw[-1]}},
        "seconds": {n_: {"fit": a["fit_s"], "impute_device": a["impute_device_s"]}, w_: {"fit": b["fit_s"], "impute_device": b["impute_device_s"]}},
        "reference_statement": "docs/src/imputation.md:62-64: the reference expects 1-2 % variation between machines",
        "note": "free-running trajectories of a DMRG sweep are chaotic (oracle/sensitivity_study.py): the per-sweep differences are those of two "
                "legitimate runs, not rounding errors of one; what is compared is the QUALITY of the two models",
    }


def ap_defaults():
    return {"N": 4096, "T": 100, "chi": 32, "d": 4}


def main():
    # Exactly one line may reach stdout: native libraries (the RCCL init banner, rocm-smi notices) write to fd 1
    # behind Python's back, so fd 1 is pointed at stderr for the whole run and the JSON line goes to the saved fd.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--T", type=int, default=100)
    ap.add_argument("--chi", type=int, default=32)
    ap.add_argument("--d", type=int, default=4)
    ap.add_argument("--rebuild-caches", action="store_true", help="(default since round 6; kept for old command lines) construct_caches twice per sweep, as the reference does")
    ap.add_argument("--no-rebuild-caches", action="store_true", help="leave the reference's two cache rebuilds per sweep out of the timed sweeps (bit-identical results)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-bonds", type=int, default=24)
    ap.add_argument("--cpu-full-sweep", action="store_true", help="(default since round 5; kept for old command lines) time ONE complete sweep of the CPU restatement")
    ap.add_argument("--pmc-file", default="", help="counter summary (profiles/aggregate_pmc.py) to quote HBM traffic / MFMA utilisation from; default: the "
                                                   "newest profiles/rNN_pmc_counters.json whose csrc digest matches this tree")
    ap.add_argument("--cpu-sample-only", action="store_true", help="cpu_baseline from the bounded bond sample only (extrapolated), skip the complete CPU sweep (~20 s)")
    ap.add_argument("--concurrent", type=int, default=8,
                    help="extra figure (never `value`): aggregate sweeps/s of this many INDEPENDENT fits sharing the GPU, one context "
                         "and stream each (hyper-parameter search / CV folds); 0 = skip")
    ap.add_argument("--fits-per-gpu", type=int, default=32,
                    help="with --gpus N > 1: independent fits per GPU of the `independent_fits` side figure (32: 8x a single fit per GPU)")
    ap.add_argument("--no-sharded-extra", action="store_true",
                    help="with --gpus N > 1: skip the configs[3] (N = 32768 sharded) side measurement")
    ap.add_argument("--sharded-N", type=int, default=32768)
    ap.add_argument("--workload", choices=["sweep", "impute"], default="sweep",
                    help="sweep: the headline training sweep (BASELINE configs[2]).  impute: BASELINE configs[4], the imputation engine "
                         "on a complex (Fourier) model with fp32 chain arithmetic; defaults N=8192 per GPU, T=200, chi=64, d=8")
    ap.add_argument("--dtype", choices=["f64", "f32", "c128", "c64"], default="f64",
                    help="element type of the sweep.  f64: the headline (BASELINE's metric).  Anything else: a SIDE line through the element-typed "
                         "kernels (csrc/mpst_typed.hip), by default at BASELINE configs[4]'s training shape N=8192 T=200 chi=64 d=8, Fourier encoding "
                         "for the complex types, with a tolerance study against the double-precision type")
    ap.add_argument("--classes", type=int, default=0, help="--dtype side line: number of classes (default 1 for complex types, 2 for f32)")
    ap.add_argument("--study-bonds", type=int, default=6, help="--dtype side line: bonds of the tolerance study")
    ap.add_argument("--study", action="store_true",
                    help="with --dtype f32|c64: the tolerance study BASELINE configs[4] names - free-running 5-sweep fits from one start in the "
                         "narrow and the wide element type (N=2048, T=100, chi=32, d=8), per-sweep KLD / accuracy, bond dimensions, and median "
                         "imputation of a held-out 50 %% block from both models; never the headline value")
    ap.add_argument("--allreduce", choices=["auto", "rccl", "oneshot"], default="auto",
                    help="collective of the sharded sweep: RCCL, the one-shot direct-write kernel, or whichever one trial sweep shows faster")
    args = ap.parse_args()
    args.rebuild_caches = not args.no_rebuild_caches       # `value` is the like-for-like sweep: RealRealHighDimension.jl:770,804 are inside the timed :727-808

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, rendezvous on 127.0.0.1) as
        # a CHILD process and hand its single JSON line through.  Nothing in this process has touched the GPU yet
        # (no torch / HIP import above), and it never replaces itself with another program.
        import socket
        import subprocess
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
        os.write(real_stdout, proc.stdout)
        sys.exit(proc.returncode)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    # MPST_BENCH_SHARE_GPU=1 (test mode for 1-GPU boxes): every rank on GPU local_rank mod #GPUs, host-side group on gloo,
    # no RCCL (it refuses two ranks on one device) - the sharded sweep then runs on the one-shot all-reduce alone
    share = os.environ.get("MPST_BENCH_SHARE_GPU") == "1"
    if share:
        # ranks that share a GPU also share its CUs: the all-reduce's spinning workgroups of all ranks together must leave room for
        # a peer's next compute kernel (csrc/mpst_allreduce.hip: AR_WG).  Before the library is loaded: it reads the value once.
        os.environ.setdefault("MPST_AR_WG", "4")
        # ... and time-share it: with 8 processes on one GPU an optimiser step takes ~10 ms and a rank can fall seconds behind
        # its peers (profiles/r04_oneshot_shared_gpu.txt); the 10 s default of the bounded spins is for a GPU per rank
        os.environ.setdefault("MPST_AR_TIMEOUT_S", "180")
    dev_index = local_rank % max(torch.cuda.device_count(), 1) if share else local_rank
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(dev_index)
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    hdev = "cpu" if share else "cuda"

    def host_reduce(vals, op):
        t = torch.tensor(vals, dtype=torch.float64, device=hdev)
        dist.all_reduce(t, op=op)
        return t.tolist()

    import mpstime_jl_amd as mt

    def note(msg):
        """Preflight / decision log: stderr of rank 0 (stdout carries the one JSON line)."""
        if rank == 0:
            sys.stderr.write(f"[bench] {msg}\n")
            sys.stderr.flush()

    if world > 1:
        cl = mt.comm_library()       # bound at run time: the copy torch has loaded, never a second one (include/mpstime_hip.h)
        note(f"{world} ranks, rank 0 on cuda:{dev_index}" + (f" (all ranks share one GPU: MPST_BENCH_SHARE_GPU=1, host group on gloo, MPST_AR_WG={os.environ.get('MPST_AR_WG')})" if share else ""))
        note(f"librccl bound by libmpstime_hip.so: {cl['library']}, version {cl['version']} (built against {cl['built_against']})")
        try:
            note(f"torch.cuda.nccl.version() = {torch.cuda.nccl.version()}")
        except Exception as e:
            note(f"torch.cuda.nccl.version() unavailable: {e}")
    if args.dtype != "f64":
        if world > 1:
            note("--dtype side line runs on one GPU")
            sys.exit(2)
        line = tolerance_study(args, mt) if args.study else typed_workload(args, mt, torch, rank, dev_index)
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
        return
    if args.workload == "impute":
        line = impute_workload(args, mt, torch, dist, world, rank, dev_index, host_reduce)
        if rank == 0:
            os.write(real_stdout, (json.dumps(line) + "\n").encode())
        if world > 1:
            dist.destroy_process_group()
        return
    N, T, d, chi, C = args.N, args.T, args.d, args.chi, 2
    full = make_inputs(N, T, d)
    W0 = mt.generate_startingMPS(4, T, d, C, 1234)

    eng = mt.SweepEngine(dev_index)
    eng.set_options(chi_max=chi, eta=0.01, cutoff=1e-10, update_iters=1, loss="KLD", bbopt="TSGO",
                    rescale=(False, True), rebuild_caches=args.rebuild_caches)
    sh = None
    if world > 1:
        sh = mt.Shard(rank, world, rccl=not share)
        local, gcounts = sh.split(full)
        sh.attach(eng)                   # RCCL communicator (the baseline collective)
        eng.set_dataset(0, local.phi, local.label_index, C, gcounts)
    else:
        eng.set_dataset(0, full.phi, full.label_index, C)
    # side measurement (not part of the timed sweeps): device-side preprocessing + encoding of the same raw matrix
    # (SURVEY 8f row 2), into the unused test slot; algorithmic bytes = 8 N T (1 + d)
    encode_info = None
    if world == 1:
        try:
            eng.encode_dataset(1, full.original_data, full.label_index, C, basis="Legendre_No_Norm", d=d)      # warm-up
            _, enc_s = eng.encode_dataset(1, full.original_data, full.label_index, C, basis="Legendre_No_Norm", d=d)
            err = float(np.max(np.abs(eng.get_encoded(1) - full.phi)))
            nbytes = 8.0 * N * T * (1 + d)
            encode_info = {"kernels": "k_enc_range + k_enc_range_final + k_encode", "device_us": 1e6 * enc_s, "algorithmic_bytes": nbytes,
                           "achieved_GBs": nbytes / enc_s / 1e9, "frac_of_hbm_peak": nbytes / enc_s / 1e9 / PEAK_HBM_GBS,
                           "max_abs_diff_vs_host_encoding": err}
            eng.set_dataset(1, np.zeros((0, T, d)), np.zeros(0, dtype=np.int32), C)
        except Exception as e:
            encode_info = {"error": str(e)}

    eng.set_mps(W0)
    allreduce = {"path": "none"}
    if world > 1:
        # the collective of the sharded sweep: RCCL's all-reduce and, next to it, the one-shot direct-write kernel over
        # peer-mapped inboxes.  Each candidate runs two trial sweeps (bounded spins: a failing path costs about a
        # second, not the run); all ranks agree on the outcome over the host-side group and keep the faster working one.
        paths = [] if share else ["rccl"]
        if args.allreduce != "rccl":
            ok = 1.0
            try:
                sh.attach_oneshot(eng)
            except Exception as e:
                ok = 0.0
                allreduce["oneshot_error"] = str(e)
            if host_reduce([ok], dist.ReduceOp.MIN)[0] > 0:
                paths.append("oneshot")
        times = {}
        for name in paths:
            if "oneshot" in paths:
                sh.select(eng, name == "oneshot")
            eng.set_mps(W0)
            eng.build_caches()
            good, dt = 1.0, 1e9
            try:
                eng.sweep()
                torch.cuda.synchronize()
                tq = time.perf_counter()
                eng.sweep()
                torch.cuda.synchronize()
                dt = time.perf_counter() - tq
            except Exception as e:
                good = 0.0
                allreduce[name + "_error"] = str(e)
            # every rank holds the whole MPS and applies the same all-reduced update: after the trial sweeps the replicas must
            # be identical - a collective that delivers wrong or differently ordered sums shows here, not in the results
            fp = 0.0
            if good > 0:
                try:
                    fp = float(int(mt.mps_content_digest(eng.get_mps())[:12], 16))      # 48 bits: exact in a double
                except Exception as e:
                    good = 0.0
                    allreduce[name + "_error"] = str(e)
            good = host_reduce([good], dist.ReduceOp.MIN)[0]
            if good > 0 and host_reduce([fp], dist.ReduceOp.MIN)[0] != host_reduce([fp], dist.ReduceOp.MAX)[0]:
                good = 0.0
                allreduce[name + "_error"] = "the ranks' MPS replicas differ after the trial sweeps"
            dt = host_reduce([dt], dist.ReduceOp.MAX)[0]
            times[name] = dt if good > 0 else None
        allreduce["trial_sweep_s"] = times
        working = [n for n in paths if times.get(n) is not None]
        for n in paths:
            note(f"collective '{n}': " + (f"trial sweep {1e3 * times[n]:.2f} ms" if times.get(n) is not None
                                           else f"FAILED ({allreduce.get(n + '_error', 'a peer failed')})"))
        if "oneshot_error" in allreduce and "oneshot" not in paths:
            note(f"collective 'oneshot' not available: {allreduce['oneshot_error']}")
        if not working:
            note(f"no working collective: {allreduce}")
            sys.exit(3)
        pick = args.allreduce if args.allreduce in working else min(working, key=lambda n: times[n])
        why = ("requested with --allreduce" if args.allreduce in working else
               "the only candidate" + (" (RCCL refuses two ranks on one device)" if share else "") if len(paths) == 1 else
               "the only one that worked" if len(working) == 1 else "the faster trial sweep")
        note(f"picked '{pick}': {why}")
        if "oneshot" in paths:
            sh.select(eng, pick == "oneshot")
        allreduce["path"] = pick
        allreduce["picked_because"] = why
        allreduce["librccl"] = mt.comm_library()
        eng.set_mps(W0)
    eng.build_caches()

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()


    # warmup: chi grows to chi_max during the first sweep
    for w in range(args.warmup):
        eng.sweep()
    info = eng.info()

    # ---- the timed region: K sweeps exactly as fitMPS runs them (profiling off: replayed from the captured hipGraph on
    # one GPU, plain stream with the RCCL all-reduce on several) ----------------------------------------------------
    sync()
    t0 = time.perf_counter()
    dev_s = 0.0
    fallbacks = 0
    for _ in range(args.steps):
        st = eng.sweep()
        dev_s += st["seconds"]
        fallbacks += st["eig_fallbacks"]
    sync()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if dist is not None:
        elapsed = host_reduce([elapsed], dist.ReduceOp.MAX)[0]
    mse, kld, acc, _ = eng.eval(0)
    chi_now, _ = eng.get_chi()
    # every rank holds the whole MPS and applied the same all-reduced updates: after the TIMED sweeps the replicas must still be
    # identical bit for bit (content digest per rank, compared over the host-side group)
    replicas = None
    if world > 1:
        dg = mt.mps_content_digest(eng.get_mps())
        fp = float(int(dg[:12], 16))                       # 48 bits: exact in a double
        lo, hi = host_reduce([fp], dist.ReduceOp.MIN)[0], host_reduce([fp], dist.ReduceOp.MAX)[0]
        replicas = {"identical_after_timed_sweeps": bool(lo == hi), "digest_rank0": dg[:16]}
        note(f"MPS replicas after the timed sweeps: {'identical on all ranks' if lo == hi else 'DIFFER'} (rank 0 digest {dg[:16]})")

    # ---- separate, untimed passes for the roofline: HIP events on the engine's own stream around every launch of every
    # kernel class (one sweep), then around the dominant class only over the same K sweeps as the timed region ---------
    eng.set_profile(0x7FF)
    eng.sweep()
    breakdown = eng.get_profile()
    dominant = max(breakdown, key=lambda k: breakdown[k][0]) if breakdown else "eig_tri"
    kidx = list(mt._lib.KERNEL_CLASSES).index(dominant)
    eng.set_profile(1 << kidx)
    for _ in range(args.steps):
        eng.sweep()
    prof = eng.get_profile()
    eng.set_profile(0)

    out = None
    if rank == 0:
        model = kernel_model(N / world, d, chi, C, info)
        us_tot, cnt = prof[dominant]
        avg_us = us_tot / max(cnt, 1)
        bound, alg = model.get(dominant, ("mfma", 0.0))
        if bound == "mfma":
            achieved, peak, unit = alg / (avg_us * 1e-6) / 1e12, PEAK_FP64_MFMA_TFLOPS, "TFLOP/s"
        else:
            achieved, peak, unit = alg / (avg_us * 1e-6) / 1e9, PEAK_HBM_GBS, "GB/s"
        kernels = {}
        for k, (us, c) in breakdown.items():
            if c == 0:
                continue
            b, a = model.get(k, ("mfma", 0.0))
            per = us / c
            kernels[k] = {"avg_us": round(per, 3), "launches_per_sweep": c, "share": round(us / max(sum(v[0] for v in breakdown.values()), 1e-9), 4),
                          "bound": b, "achieved": round(a / (per * 1e-6) / (1e12 if b == "mfma" else 1e9), 4),
                          "unit": "TFLOP/s" if b == "mfma" else "GB/s"}
        # HBM traffic / MFMA utilisation: PMC counters cannot be collected from inside this process, so the per-launch
        # figures of the committed rocprofv3 --pmc passes over this same command are QUOTED - under their own key, and only
        # when that profile was taken with the kernel sources of this tree (csrc_sha256; profiles/aggregate_pmc.py;
        # FETCH_SIZE doubled per the MI355X guide's gfx950 correction)
        traffic, traffic_src = None, None
        fused, b2 = info.get("fused"), info.get("sliced_bond_gemms")
        knames = {"eig_tri": ("mpst::k_bt_coop" if info.get("large_bond") else "mpst::k_eig_trivec" if info.get("eig_merged") else "mpst::k_eig_tri"), "eig_vec": "mpst::k_eig_vec", "eig_fin": "mpst::k_eig_fin",
                  "yhat": "mpst::k_yhat_s" if b2 else "mpst::k_yhat",
                  "grad": "mpst::k_grad_s" if b2 else ("mpst::k_bond_fused" if fused else "mpst::k_grad"),
                  "gram": "mpst::k_gram_upd" if fused else "mpst::k_gram",
                  "split": "mpst::k_split", "env": ("mpst::k_bond_tail" if info.get("four_launch_chain") else "mpst::k_env_split") if fused else "mpst::k_env",
                  "grad_reduce+update": "mpst::k_fused_reduce" if fused else "mpst::k_grad_reduce"}
        # --pmc-file, else the newest committed profiles/r*_pmc_counters.json taken on THIS tree's kernel sources
        import glob
        cands = [args.pmc_file] if args.pmc_file else sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_counters.json")), reverse=True)
        pmc_file, pmc, pmc_quoted = None, {}, None
        for cf in cands:
            try:
                pj = json.load(open(cf if os.path.isabs(cf) else os.path.join(ROOT, cf) if not os.path.exists(cf) else cf))
                if pj.get("csrc_sha256") == csrc_sha256():
                    pmc, pmc_file = pj["kernels"], os.path.relpath(cf, ROOT) if os.path.isabs(cf) else cf
                    break
            except Exception:
                pass
        headline = world == 1 and (N, T, chi, d) == (4096, 100, 32, 4)
        kname = knames.get(dominant)
        if headline and pmc:
            pmc_quoted = {"source": pmc_file, "csrc_sha256": csrc_sha256(), "kernels": {}}
            for k, kn in knames.items():
                hit = [v for name, v in pmc.items() if name == kn or name.startswith(kn + "<") or name.startswith("void " + kn)]
                if k in kernels and hit:
                    pmc_quoted["kernels"][k] = {"mfma_util": hit[0].get("mfma_util"),
                                                "hbm_bytes_per_launch": hit[0].get("hbm_bytes_per_launch_corrected")}
            if dominant in pmc_quoted["kernels"]:
                traffic, traffic_src = pmc_quoted["kernels"][dominant]["hbm_bytes_per_launch"], pmc_file
        out = {
            "metric": ("full sweeps/sec (N=4096,T=100,chi=32,d=4)" if (N, T, chi, d) == (4096, 100, 32, 4)
                       else f"full sweeps/sec (N={N},T={T},chi={chi},d={d}) - NOT the headline configuration"),
            "value": args.steps / elapsed, "unit": "sweeps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"DMRG training sweep, noisy trendy sine 2-class, N={N}, T={T}, chi_max={chi}, d={d} Legendre, "
                                   f"KLD+TSGO eta=0.01, fp64; " + ("configs[2] of BASELINE.json" if (N, T, chi, d) == (4096, 100, 32, 4)
                                                                   else "a side measurement, not the configuration the metric is quoted on"),
                       "N": N, "T": T, "chi_max": chi, "d": d, "C": C, "rebuild_caches": bool(args.rebuild_caches),
                       "parallelism": f"batch-shard x{world}" if world > 1 else "single GPU",
                       "bond_dims_max": int(chi_now.max())},
            "device_ms_per_step": 1e3 * dev_s / args.steps,
            "train_KL_div_after": kld, "train_acc_after": acc,
            "allreduce": dict(allreduce, nranks_seen=int(info.get("ranks", 1)), us_per_optimiser_step=(breakdown["allreduce"][0] / max(breakdown["allreduce"][1], 1)
                                                                 if breakdown.get("allreduce", (0, 0))[1] else None), replicas=replicas),
            "eig_fallbacks_total": fallbacks, "eig_phases_us_last_bond": eng.eig_phases(), "launch_chain": info,
            "tail_phases_us": eng.tail_phases() if info.get("four_launch_chain") else None,
            "timed_region": "K sweeps with profiling off" + (" (hipGraph replay)" if info.get("graph") else " (plain stream)") +
                            "; per-kernel figures from separate event-instrumented sweeps after it",
            # the whole SVD (gram + eig_tri + eig_vec + eig_fin) against SURVEY 8(d)'s dense-SVD count 4mn^2 + 8n^3
            "svd_group": {"algorithmic_flops_dense_svd": 4.0 * (chi * C * d) * (d * chi) ** 2 + 8.0 * (d * chi) ** 3,
                          "avg_us": round(sum(breakdown[k][0] / max(breakdown[k][1], 1) for k in ("gram", "eig_tri", "eig_vec", "eig_fin")
                                              if k in breakdown), 3)},
            # SURVEY 8(d)'s per-sweep counts (all bonds at chi_max), for the region that was TIMED (rebuild_caches off: the two
            # construct_caches passes the reference repeats are left out, A.6) and with them
            "algorithmic_per_sweep": (lambda nb, w: (lambda byt, flo, rb, rf: {
                "timed_region": {"bytes": byt + (rb if args.rebuild_caches else 0.0), "flops": flo + (rf if args.rebuild_caches else 0.0)},
                "survey_8d_with_cache_rebuilds": {"bytes": byt + rb, "flops": flo + rf},
                "whole_sweep_frac_of_fp64_mfma_peak": (flo + (rf if args.rebuild_caches else 0.0)) / (elapsed / args.steps) / 1e12 / PEAK_FP64_MFMA_TFLOPS,
                "whole_sweep_frac_of_hbm_peak": (byt + (rb if args.rebuild_caches else 0.0)) / (elapsed / args.steps) / 1e9 / PEAK_HBM_GBS,
                "note": "N is the per-rank batch; bond terms: w N (3 chi + 2 d) + 2 w C d^2 chi^2 bytes, 4 N (d chi)^2 + 2 N d chi^2 + "
                        "4 (chi C d)(d chi)^2 + 8 (d chi)^3 flops; rebuild terms: 2 w N T (d + chi) bytes, 4 N T d chi^2 flops"})(
                    nb * (w * (N / world) * (3 * chi + 2 * d) + 2.0 * w * C * d * d * chi * chi),
                    nb * (4.0 * (N / world) * (d * chi) ** 2 + 2.0 * (N / world) * d * chi * chi + 4.0 * (chi * C * d) * (d * chi) ** 2 + 8.0 * (d * chi) ** 3),
                    2.0 * w * (N / world) * T * (d + chi), 4.0 * (N / world) * T * d * chi * chi))(2 * (T - 1), 8),
            "roofline": {"kernel": dominant, "kernel_symbol": kname, "bound": bound, "achieved": achieved, "peak": peak, "unit": unit,
                         "frac": achieved / peak, "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                         "avg_launch_us": avg_us, "launches": cnt, "algorithmic_per_launch": alg,
                         "in_kernel_us_last_launch": eng.eig_phases()["tridiag"] if dominant == "eig_tri" else None},
            "kernels": kernels,
            "pmc_quoted": pmc_quoted,
            "device_encode": encode_info,
        }

    # ---- extra with several ranks: BASELINE configs[3] - N = 32768 series sharded over the ranks (weak-scaling shape: the
    # per-rank batch that makes the sharded kernels matter) - with the all-reduce time per optimiser step and the Amdahl
    # bound from THIS run's own per-kernel profile.  Never `value`.
    if world > 1 and not args.no_sharded_extra:
        sharded = None
        eng2 = None
        if out is not None:
            # the headline figures are on stderr before the side measurement starts: whatever happens in it, they are logged
            note("headline (before the configs[3] side measurement): " + json.dumps({k: out[k] for k in ("metric", "value", "unit", "n_gpus", "ms_per_step")}))
        try:
            # the headline engine is done (nothing below uses it with several ranks): its stream must not stay around as a second
            # user queue per process - 8 ranks x 2 queues oversubscribe the hardware queue slots of a SHARED GPU and every
            # optimiser step then waits for a 10 ms scheduling quantum (measured: 10.0 ms per all-reduce, round 4)
            eng.close()
            eng = None
            N2 = args.sharded_N
            full2 = make_inputs(N2, T, d)
            eng2 = mt.SweepEngine(dev_index)
            eng2.set_options(chi_max=chi, eta=0.01, cutoff=1e-10, update_iters=1, loss="KLD", bbopt="TSGO", rescale=(False, True))
            sh2 = mt.Shard(rank, world, rccl=not share)
            local2, gc2 = sh2.split(full2)
            sh2.attach(eng2)
            eng2.set_dataset(0, local2.phi, local2.label_index, C, gc2)
            eng2.set_mps(W0)
            if allreduce.get("path") == "oneshot" or share:
                sh2.attach_oneshot(eng2)
                sh2.select(eng2, True)
            eng2.build_caches()
            setup_ok = 1.0
        except Exception as e:
            setup_ok = 0.0
            sharded = {"error": f"setup: {e}"}
        # every rank learns whether every rank got this far: nobody waits in a collective for a rank that has given up
        if host_reduce([setup_ok], dist.ReduceOp.MIN)[0] <= 0:
            if sharded is None:
                sharded = {"error": "setup failed on another rank"}
            note(f"configs[3] side measurement skipped: {sharded['error']}")
            raise_skip = True
        else:
            raise_skip = False
        try:
            if raise_skip:
                raise RuntimeError(sharded["error"])
            for _ in range(2):
                eng2.sweep()
            sync()
            tq = time.perf_counter()
            nsw2 = 3
            for _ in range(nsw2):
                eng2.sweep()
            sync()
            dt2 = host_reduce([time.perf_counter() - tq], dist.ReduceOp.MAX)[0]
            eng2.set_profile(0x7FF)
            eng2.sweep()
            bd2 = eng2.get_profile()
            eng2.set_profile(0)
            eng2.close()
            nb = 2 * (T - 1)
            per = {k: v[0] / nb for k, v in bd2.items() if v[1]}          # us per bond
            shard_keys = ("yhat", "grad", "env")                          # N-proportional: they shrink with the shard
            t_sh = sum(per.get(k, 0.0) for k in shard_keys)
            t_ar = per.get("allreduce", 0.0)
            t_rep = sum(v for k, v in per.items() if k not in shard_keys and k != "allreduce")
            sharded = {"N": N2, "config": "configs[3] of BASELINE.json", "value": nsw2 / dt2, "unit": "sweeps/s",
                       "ms_per_step": 1e3 * dt2 / nsw2, "per_rank_series": int(local2.phi.shape[0]),
                       "allreduce_us_per_optimiser_step": (bd2["allreduce"][0] / bd2["allreduce"][1]) if bd2.get("allreduce", (0, 0))[1] else None,
                       "us_per_bond": {"replicated (gram, eigensolver, split)": t_rep, "sharded (yhat, grad, env) at N/ranks": t_sh,
                                       "allreduce": t_ar},
                       # one GPU would spend about ranks * t_sh on the sharded kernels (they are N-proportional at this size)
                       "amdahl": {"speedup_vs_1gpu_estimate": (t_rep + world * t_sh) / max(t_rep + t_sh + t_ar, 1e-9),
                                  "ceiling_infinite_ranks": (t_rep + world * t_sh) / max(t_rep + t_ar, 1e-9),
                                  "note": "from this run's event profile; the per-bond eigensolver is replicated on every rank"}}
            note(f"configs[3] side measurement: N={N2} over {world} ranks: {sharded['value']:.2f} sweeps/s, all-reduce "
                 f"{sharded['allreduce_us_per_optimiser_step']} us per optimiser step, Amdahl estimate {sharded['amdahl']['speedup_vs_1gpu_estimate']:.2f}x")
        except Exception as e:
            if not raise_skip:
                sharded = {"error": str(e)}
                note(f"configs[3] side measurement failed: {e}")
            try:
                eng2.close()
            except Exception:
                pass
        if out is not None:
            out["sharded_n32768"] = sharded

    # ---- extra with several ranks: what DOES scale on a node - independent fits (hyper-parameter candidates, CV folds, restarts: the
    # reference's @distributed loops) dealt over the GPUs, K per GPU in one launch chain each (mpst_sweep_batch), no collective at all.
    # Every fit is a complete headline-shape fit on the whole data set.  Never `value`.
    if world > 1 and args.concurrent > 1:
        indep = None
        engs = []
        try:
            K = max(1, args.fits_per_gpu)
            for k in range(K):
                e2 = mt.SweepEngine(dev_index)
                e2.set_batch_hint(K)
                e2.set_options(chi_max=chi, eta=0.01, cutoff=1e-10, update_iters=1, loss="KLD", bbopt="TSGO", rescale=(False, True))
                e2.set_dataset(0, full.phi, full.label_index, C)
                e2.set_mps(W0)
                e2.build_caches()
                engs.append(e2)
            for _ in range(2):
                mt.sweep_batch(engs)                # capture + warm; bond dimensions reach chi_max
            okk = 1.0
        except Exception as e:
            okk = 0.0
            indep = {"error": str(e)}
        if host_reduce([okk], dist.ReduceOp.MIN)[0] > 0:
            nsw = 3
            sync()
            ti0 = time.perf_counter()
            for _ in range(nsw):
                mt.sweep_batch(engs)
            torch.cuda.synchronize()
            ti = host_reduce([time.perf_counter() - ti0], dist.ReduceOp.MAX)[0]
            indep = {"fits_per_gpu": K, "gpus": world, "sweeps_each": nsw, "aggregate_sweeps_per_s": world * K * nsw / ti,
                     "ratio_to_single_fit_single_gpu": None, "collective": "none",
                     "note": "independent fits of the headline shape dealt over the ranks, K per GPU in one launch chain (mpst_sweep_batch; one process "
                             "drives several GPUs with mpst_sweep_batch_multi); a side figure, never the headline metric"}
        elif indep is None:
            indep = {"error": "setup failed on another rank"}
        for e2 in engs:
            try:
                e2.close()
            except Exception:
                pass
        if out is not None:
            out["independent_fits"] = indep

    # ---- with several ranks the headline line says, as first-class keys, what the sharded sweep can and cannot do: its measured
    # value, the Amdahl ceiling from THIS run's own per-kernel profile of the headline shape (the per-bond eigensolver, Gram update and
    # split are replicated on every rank; only the bond GEMMs and the environment update shrink with the shard), and the thing that
    # does scale on a node: independent fits.
    if world > 1 and out is not None:
        nbh = 2 * (T - 1)
        perh = {k: v[0] / nbh for k, v in breakdown.items() if v[1]}
        sk = ("yhat", "grad", "env")
        h_sh = sum(perh.get(k, 0.0) for k in sk)
        h_ar = perh.get("allreduce", 0.0)
        h_rep = sum(v for k, v in perh.items() if k not in sk and k != "allreduce")
        # the sharded kernels at N/ranks series are launch-latency bound at this size: one GPU spends at most ranks x their time, at least
        # their time - both readings are printed
        out["multi_gpu"] = {
            "sharded_sweeps_per_s": out["value"],
            "us_per_bond": {"replicated (gram, eigensolver, split)": h_rep, "sharded (yhat, grad, env) at N/ranks": h_sh, "allreduce": h_ar},
            "amdahl": {"speedup_vs_1gpu_if_sharded_kernels_are_N_proportional": (h_rep + world * h_sh) / max(h_rep + h_sh + h_ar, 1e-9),
                       "speedup_vs_1gpu_if_they_are_latency_bound": (h_rep + h_sh) / max(h_rep + h_sh + h_ar, 1e-9),
                       "ceiling_infinite_ranks": (h_rep + world * h_sh) / max(h_rep + h_ar, 1e-9),
                       "note": "from this run's event profile of the headline shape; at N = 4096 one GPU's four-launch chain (no all-reduce, "
                               "fused tail) is faster than any sharded run: the sharded sweep is for batches that do not fit one GPU's time budget"},
            "sharded_n32768": ({k: sharded.get(k) for k in ("value", "unit", "allreduce_us_per_optimiser_step", "amdahl", "error") if k in sharded}
                               if isinstance(out.get("sharded_n32768"), dict) else None),
            "independent_fits_aggregate_sweeps_per_s": (out.get("independent_fits") or {}).get("aggregate_sweeps_per_s"),
            "independent_fits_per_gpu": (out.get("independent_fits") or {}).get("fits_per_gpu"),
            "what_scales": "independent fits (hyper-parameter candidates, folds, restarts) dealt over the GPUs with no collective; a single fit of "
                           "the headline shape does not, and the line says by how much"}

    # ---- extra: the same K sweeps WITHOUT the reference's two cache rebuilds per sweep (RealRealHighDimension.jl:770,804): they recompute
    # what the sweep has just left in the caches (bit-identical results; tests/test_gpu_parity.py::test_rebuild_caches_is_bit_identical)
    if rank == 0 and world == 1:
        other = not args.rebuild_caches
        key = "rebuild_caches_on" if other else "rebuild_caches_off"
        try:
            eng.set_options(chi_max=chi, eta=0.01, cutoff=1e-10, update_iters=1, loss="KLD", bbopt="TSGO", rescale=(False, True), rebuild_caches=other)
            eng.sweep()                             # capture + warm
            torch.cuda.synchronize()
            tr0 = time.perf_counter()
            for _ in range(args.steps):
                eng.sweep()
            torch.cuda.synchronize()
            trb = time.perf_counter() - tr0
            out[key] = {"value": args.steps / trb, "unit": "sweeps/s", "ms_per_step": 1e3 * trb / args.steps,
                        "note": ("construct_caches twice per sweep as the reference does" if other else
                                 "the two construct_caches passes per sweep left out") + "; same results bit for bit"}
            eng.set_options(chi_max=chi, eta=0.01, cutoff=1e-10, update_iters=1, loss="KLD", bbopt="TSGO", rescale=(False, True), rebuild_caches=args.rebuild_caches)
        except Exception as e:
            out[key] = {"error": str(e)}

    # ---- extra: K independent fits sharing the GPU.  One fit is bound by the latency chain of its per-bond eigensolver
    # (one workgroup of 256 CUs busy for 3/4 of a bond), so independent fits - the reference farms hyper-parameter
    # candidates and CV folds out with @distributed - overlap almost perfectly.  Reported next to `value`, never as it.
    if rank == 0 and world == 1 and args.concurrent > 1:
        try:
            import threading
            K = args.concurrent
            engs = []
            for k in range(K):
                e2 = mt.SweepEngine(dev_index)
                e2.set_batch_hint(K)
                e2.set_options(chi_max=chi, eta=0.01, cutoff=1e-10, update_iters=1, loss="KLD", bbopt="TSGO", rescale=(False, True))
                e2.set_dataset(0, full.phi, full.label_index, C)
                e2.set_mps(eng.get_mps())
                e2.build_caches()
                e2.sweep()                          # capture + warm
                engs.append(e2)
            nsw = 3
            def run(e):
                for _ in range(nsw):
                    e.sweep()
            torch.cuda.synchronize()
            tc0 = time.perf_counter()
            ths = [threading.Thread(target=run, args=(e,)) for e in engs]
            [t.start() for t in ths]
            [t.join() for t in ths]
            torch.cuda.synchronize()
            tc = time.perf_counter() - tc0
            threads_rate = K * nsw / tc
            # the same K fits advanced by ONE launch chain (mpst_sweep_batch: every launch carries all K fits)
            mt.sweep_batch(engs)                    # capture + warm
            torch.cuda.synchronize()
            tb0 = time.perf_counter()
            for _ in range(nsw):
                mt.sweep_batch(engs)
            torch.cuda.synchronize()
            tb = time.perf_counter() - tb0
            out["concurrent_fits"] = {"fits": K, "sweeps_each": nsw, "aggregate_sweeps_per_s": K * nsw / tb,
                                      "ratio_to_single_fit": (K * nsw / tb) / out["value"], "ms_per_batched_sweep": 1e3 * tb / nsw,
                                      "host_threads": {"aggregate_sweeps_per_s": threads_rate, "ratio_to_single_fit": threads_rate / out["value"],
                                                       "note": "K contexts driven from K host threads (own stream and hipGraph each): bound by the dispatch rate"},
                                      "note": "K independent fits of one shape in one launch chain (mpst_sweep_batch); results bit-identical to separate "
                                              "sweeps (tests/test_gpu_sweep_paths.py); a side figure, never the headline metric"}
            # beyond 8 fits the eigensolver's workgroups outnumber the CUs: every workgroup then takes several eigenpairs instead of
            # repeating the reduction in rounds (k_eig_trivec_bm) - the same chain with 32 fits
            K2 = 32
            if K < K2:
                for e2 in engs:
                    e2.close()
                engs = []
                for k in range(K2):
                    e2 = mt.SweepEngine(dev_index)
                    e2.set_batch_hint(K2)
                    e2.set_options(chi_max=chi, eta=0.01, cutoff=1e-10, update_iters=1, loss="KLD", bbopt="TSGO", rescale=(False, True))
                    e2.set_dataset(0, full.phi, full.label_index, C)
                    e2.set_mps(eng.get_mps())
                    e2.build_caches()
                    engs.append(e2)
                mt.sweep_batch(engs)                # capture + warm
                torch.cuda.synchronize()
                tb0 = time.perf_counter()
                for _ in range(nsw):
                    mt.sweep_batch(engs)
                torch.cuda.synchronize()
                tb2 = time.perf_counter() - tb0
                out["concurrent_fits"]["fits_32"] = {"fits": K2, "aggregate_sweeps_per_s": K2 * nsw / tb2,
                                                     "ratio_to_single_fit": (K2 * nsw / tb2) / out["value"], "ms_per_batched_sweep": 1e3 * tb2 / nsw}
            for e2 in engs:
                e2.close()
        except Exception as e:
            out.setdefault("concurrent_fits", {})["error"] = str(e)

    # ---- CPU baseline: the C restatement of the reference loop structure, bounded sample --------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            from oracle.c_oracle import COracle
            Wnow = eng.get_mps()                      # steady-state MPS (all bulk bonds at chi_max)
            co = COracle(Wnow, full.phi, full.label_index, full.class_distribution, chi, eta=0.01, rebuild_caches=True,
                         native=True)
            t_c0 = time.perf_counter()
            co.build_caches()                         # construct_caches: one of the two rebuilds per sweep
            t_cache = time.perf_counter() - t_c0
            skip = 6                                  # leave the narrow edge bonds out of the per-bond average
            co.sweep(max_bonds=skip, first_bond=0)
            r = co.sweep(max_bonds=args.cpu_bonds, first_bond=skip)
            per_bond = r["seconds"] / max(r["bonds"], 1)
            sweep_s = per_bond * 2 * (T - 1) + 2 * t_cache
            full_sweep_s = None
            if not args.cpu_sample_only:              # ONE whole sweep on the host (all 2(T-1) bonds + both cache rebuilds), timed, no extrapolation
                co2 = COracle(Wnow, full.phi, full.label_index, full.class_distribution, chi, eta=0.01, rebuild_caches=True, native=True)
                t_f0 = time.perf_counter()
                co2.build_caches()
                co2.sweep()
                full_sweep_s = time.perf_counter() - t_f0
            # (i) the SVD with every host core (as OpenBLAS would run it for Julia), (ii) the
            # NumPy / SciPy-gesdd oracle on the same inputs as a second data point
            import scipy.linalg
            A = np.random.default_rng(0).standard_normal((chi * C * d, d * chi))
            t_s0 = time.perf_counter()
            for _ in range(20):
                scipy.linalg.svd(A, full_matrices=False, lapack_driver="gesdd")
            svd_ms = 1e3 * (time.perf_counter() - t_s0) / 20
            np_per_bond = None
            try:
                from oracle import ref_numpy as R
                Wn = [t.copy() for t in Wnow]
                dsn = R.EncodedSet(full.phi, full.label_index, full.class_distribution)
                opts_n = R.SweepOptions(nsweeps=1, chi_max=chi, eta=0.01)
                nb_np = 3
                # run the bonds in sweep order from the right end so that the oracle's caches are valid
                Wn = [t.copy() for t in Wnow]
                LEn, REn = R.construct_caches(Wn, dsn.phi, True)
                t_n0 = time.perf_counter()
                for q in range(skip + nb_np):
                    R.bond_step(Wn, LEn, REn, T - 2 - q, dsn, opts_n, True, {})
                np_per_bond = (time.perf_counter() - t_n0) / (skip + nb_np)
            except Exception:
                pass
            port = ("oracle/mps_oracle.c (-O3 -march=native -ffast-math, 1 thread for the per-series loops, SciPy OpenBLAS dgesdd); "
                    "the Julia reference itself cannot run here")
            extrap = {"value": 1.0 / sweep_s, "unit": "sweeps/s", "seconds_sampled": r["seconds"] + t_cache,
                      "sample": f"{r['bonds']} steady-state bulk bond updates ({per_bond:.3f} s each) + one construct_caches ({t_cache:.2f} s), "
                                f"extrapolated to 2(T-1)={2 * (T - 1)} bonds + 2 cache rebuilds"}
            timed = full_sweep_s is not None
            out["cpu_baseline"] = {
                "value": 1.0 / (full_sweep_s if timed else sweep_s), "unit": "sweeps/s", "cores": 1, "kind": "port",
                "sample": (f"{port}: ONE complete sweep from the steady-state MPS - all {2 * (T - 1)} bond updates + both cache rebuilds - "
                           f"TIMED on this host ({full_sweep_s:.1f} s), no extrapolation") if timed else f"{port}: {extrap['sample']}",
                "host_cpus": os.cpu_count(), "host_cpu_model": host_cpu_model(),
                "seconds_sampled": full_sweep_s if timed else extrap["seconds_sampled"],
                "full_sweep_timed": None if not timed else {"seconds": full_sweep_s, "value": 1.0 / full_sweep_s, "unit": "sweeps/s"},
                "extrapolated_from_bond_sample": extrap,
                "svd_gesdd_all_cores_ms": svd_ms,
                "numpy_oracle": None if np_per_bond is None else {
                    "value": 1.0 / (np_per_bond * 2 * (T - 1)), "unit": "sweeps/s",
                    "sample": f"oracle/ref_numpy.py (vectorised NumPy + SciPy gesdd, {os.cpu_count()} BLAS threads available): "
                              f"{skip + 3} bond updates from the right end ({np_per_bond:.3f} s each) extrapolated to {2 * (T - 1)}"}}
        except Exception as e:      # the baseline is a reported extra, never a reason to lose the bench line
            out["cpu_baseline"] = {"value": None, "unit": "sweeps/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if eng is not None:
        eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
