"""Host-side input producers of the sweep: preprocessing and encodings.

Mirrors src/utils.jl:161-295 (RobustSigmoid + MinMax) and src/Encodings (encode_dataset,
Encodings/encodings.jl:33-156; bases, Encodings/bases.jl).  These run once per fit on the
host in the reference as well; the data-driven bases (SLTD, Sahand-Legendre, split bases,
projected bases) are out of scope of the sweep engine and raise.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import numpy as np

from .options import MPSOptions, encoding_info


# ---------------------------------------------------------------------------------------
# containers (src/Structs/structs.jl:12-33)
# ---------------------------------------------------------------------------------------
@dataclass
class EncodedTimeSeriesSet:
    """timeseries (as one array), original_data, class_distribution - structs.jl:27-33.
    ``phi[i, t, :]`` is PState i's pstate[t]; ``labels``/``label_index`` are PState.label /
    PState.label_index (0-based here)."""

    phi: np.ndarray
    labels: np.ndarray
    label_index: np.ndarray
    original_data: np.ndarray
    class_distribution: np.ndarray

    def __len__(self):
        return 0 if self.phi.ndim != 3 else self.phi.shape[0]

    @staticmethod
    def empty():
        return EncodedTimeSeriesSet(np.zeros((0, 0, 0)), np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int32),
                                    np.zeros((0, 0)), np.zeros(0, dtype=np.int64))

    def isempty(self):
        return len(self) == 0


# ---------------------------------------------------------------------------------------
# bases (src/Encodings/bases.jl)
# ---------------------------------------------------------------------------------------
def _legendre_table(x, d):
    """P_0..P_{d-1} by Bonnet recursion, scaled to sqrt((2k+1)/2) P_k - LegendrePolynomials'
    Pl(x, k; norm=Val(:normalized)) as used at bases.jl:77-79."""
    x = np.asarray(x, dtype=np.float64)
    out = np.empty(x.shape + (d,))
    p0 = np.ones_like(x)
    out[..., 0] = p0
    if d > 1:
        p1 = x.copy()
        out[..., 1] = p1
        for k in range(1, d - 1):
            p2 = ((2 * k + 1) * x * p1 - k * p0) / (k + 1)
            out[..., k + 1] = p2
            p0, p1 = p1, p2
    out *= np.sqrt((2.0 * np.arange(d) + 1.0) / 2.0)
    return out


def legendre_encode(x, d, norm=True):
    """bases.jl:81-92.  norm=True divides by sqrt(Pl(1,d;normalized)*d) (:86-89)."""
    ls = _legendre_table(x, d)
    if norm:
        ls = ls / math.sqrt(math.sqrt((2 * d + 1) / 2.0) * d)
    return ls


def legendre_encode_no_norm(x, d):
    """bases.jl:108."""
    return legendre_encode(x, d, norm=False)


def get_fourier_freqs(d):
    """bases.jl:27-34: 0, 1, -1, 2, -2, ... truncated to d terms."""
    hb = math.ceil((d - 1.0) / 2.0)
    fr = [0]
    for i in range(1, hb + 1):
        fr += [i, -i]
    return fr[:d]


def fourier_encode(x, d):
    """bases.jl:23-42: cispi(k x)/sqrt(d)."""
    x = np.asarray(x, dtype=np.float64)
    k = np.asarray(get_fourier_freqs(d), dtype=np.float64)
    return np.exp(1j * np.pi * x[..., None] * k) / math.sqrt(d)


def angle_encode(x, d=2, periods=0.25):
    """Stoudenmire angle encoding, bases.jl:7-21 (d must be 2)."""
    if d != 2:
        raise ValueError("Stoudenmire Angle encoding only supports d = 2!")
    x = np.asarray(x, dtype=np.float64)
    s1 = np.exp(1j * np.pi * 1.5 * x) * np.cos(np.pi * 2 * periods * x)
    s2 = np.exp(-1j * np.pi * 1.5 * x) * np.sin(np.pi * 2 * periods * x)
    return np.stack([s1, s2], axis=-1)


def sahand_encode(x, d):
    """bases.jl:45-68."""
    if d % 2:
        raise ValueError("Sahand encoding only supports even dimension")
    x = np.asarray(x, dtype=np.float64)
    dx = 2.0 / d
    out = np.zeros(x.shape + (d,), dtype=np.complex128)
    for i in range(1, d + 1):
        interval = math.ceil(i / 2)
        startx = (interval - 1) * dx
        inside = (startx <= x) & (x <= interval * dx)
        if i % 2:
            s = np.exp(1j * np.pi * 1.5 * x / dx) * np.cos(np.pi * 0.5 * (x - startx) / dx)
        else:
            s = np.exp(-1j * np.pi * 1.5 * x / dx) * np.sin(np.pi * 0.5 * (x - startx) / dx)
        out[..., i - 1] = np.where(inside, s, 0.0)
    return out


def uniform_encode(x, d):
    """bases.jl:2-4."""
    x = np.asarray(x, dtype=np.float64)
    return np.full(x.shape + (d,), 1.0 / d)


@dataclass(frozen=True)
class Encoding:
    """Basis (src/Encodings/basis_structs.jl): name, encode(x, d), complex flag, input range."""

    name: str
    iscomplex: bool
    range: tuple
    encode: object = None


def model_encoding(symbol, custom: Optional[Encoding] = None) -> Encoding:
    """options.jl:243-279."""
    canon, iscomplex, rng, data_driven = encoding_info(symbol)
    if canon == "Custom":
        if custom is None:
            raise ValueError("To use a custom encoding, pass custom_encoding")
        return custom
    if data_driven:
        raise NotImplementedError(f"encoding {canon} is data-driven (host-side, one-off) and outside the sweep engine's scope")
    fn = {"Legendre_No_Norm": legendre_encode_no_norm, "Legendre_Norm": legendre_encode, "Fourier": fourier_encode,
          "Stoudenmire": angle_encode, "Sahand": sahand_encode, "Uniform": uniform_encode}[canon]
    return Encoding(canon, iscomplex, rng, fn)


def symbolic_encoding(enc: Encoding) -> str:
    """Inverse of model_encoding (options.jl:281-296); basis_tests.jl:8 pins the round trip."""
    return enc.name


# ---------------------------------------------------------------------------------------
# preprocessing (src/utils.jl:161-295).  Series are ROWS here ((N, T)); both normalisations
# are fitted over the whole matrix, so the reference's transposed orientation is immaterial.
# ---------------------------------------------------------------------------------------
@dataclass
class Norms:
    sigmoid: Optional[tuple] = None   # (median, iqr)  - Normalization.jl RobustSigmoid
    minmax: Optional[tuple] = None    # (min, max)


def _robust_sigmoid(X, med, iqr):
    return 1.0 / (1.0 + np.exp(-(X - med) / (iqr / 1.35)))


def transform_train_data(X_train, opts: MPSOptions, enc_range):
    """utils.jl:161-199."""
    Xs = np.array(X_train, dtype=np.float64, copy=True)
    norms = Norms()
    if opts.sigmoid_transform:
        med = float(np.median(Xs))
        q75, q25 = np.percentile(Xs, [75.0, 25.0])
        norms.sigmoid = (med, float(q75 - q25))
        Xs = _robust_sigmoid(Xs, *norms.sigmoid)
    if opts.minmax:
        lo, hi = float(Xs.min()), float(Xs.max())
        norms.minmax = (lo, hi)
        Xs = (Xs - lo) / (hi - lo)
        lb, ub = opts.data_bounds
        Xs = Xs * (ub - lb) + lb
    a, b = enc_range
    return (b - a) * Xs + a, norms


def transform_test_data(X_test, norms: Norms, opts: MPSOptions, enc_range, rescale_out_of_bounds=True):
    """utils.jl:202-275: train-fitted transforms, then per-series out-of-bounds rescale (:243-266)."""
    Xs = np.array(X_test, dtype=np.float64, copy=True)
    if Xs.size == 0:
        return Xs, []
    if norms.sigmoid is not None:
        Xs = _robust_sigmoid(Xs, *norms.sigmoid)
    if norms.minmax is not None:
        lo, hi = norms.minmax
        Xs = (Xs - lo) / (hi - lo)
    if opts.minmax:
        lb, ub = opts.data_bounds
        Xs = Xs * (ub - lb) + lb
    oob = []
    if rescale_out_of_bounds:
        for i in range(Xs.shape[0]):
            ts = Xs[i]
            tr = [i, 0.0, 1.0]
            lo, hi = float(ts.min()), float(ts.max())
            if lo < 0:
                ts -= lo
                hi = float(ts.max())
                tr[1] = lo
            if hi > 1:
                ts /= hi
                tr[2] = hi
            if tr[1:] != [0.0, 1.0]:
                oob.append(tr)
    a, b = enc_range
    return (b - a) * Xs + a, oob


def transform_data(X_train, X_test, opts: MPSOptions, enc_range):
    """utils.jl:287-295."""
    Xtr, norms = transform_train_data(X_train, opts, enc_range)
    Xte, oob = transform_test_data(X_test, norms, opts, enc_range)
    return Xtr, Xte, norms, oob


# ---------------------------------------------------------------------------------------
# encode_dataset (Encodings/encodings.jl:33-156)
# ---------------------------------------------------------------------------------------
def encode_dataset(X_orig, X_scaled, y, enc: Encoding, d, class_keys) -> EncodedTimeSeriesSet:
    """Stable sort by class (:43), range check (:115-119), encode every value, class
    distribution in class-key order (:151-152)."""
    y = np.asarray(y)
    if X_scaled.shape[0] == 0:
        return EncodedTimeSeriesSet.empty()
    order = np.argsort(y, kind="stable")
    Xo, Xs, ys = np.asarray(X_orig)[order], X_scaled[order], y[order]
    a, b = enc.range
    if not np.all((a <= Xs) & (Xs <= b)):
        raise ValueError(f"Data must be rescaled between {a} and {b} before a {enc.name} encoding.")
    phi = enc.encode(Xs, d)
    label_index = np.array([class_keys[v] for v in ys.tolist()], dtype=np.int32)
    _, counts = np.unique(ys, return_counts=True)
    return EncodedTimeSeriesSet(phi, ys, label_index, np.array(Xo, dtype=np.float64), counts.astype(np.int64))
