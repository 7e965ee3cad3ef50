"""MPSOptions - the reference's option struct with the same field names, defaults and
validation (src/Structs/options.jl:11-39, defaults :106-143, symbol mapping :243-327).

Julia Symbols become Python strings (``:Legendre`` -> ``"Legendre"``); a leading colon is
accepted and stripped so option tables can be pasted from Julia code.
"""
from __future__ import annotations

from dataclasses import dataclass, fields, replace
from typing import Tuple


def _sym(x) -> str:
    return str(x).lstrip(":")


# model_encoding (options.jl:243-279): name -> (canonical name, is complex, range, time dependent / data driven)
_ENCODINGS = {
    "legendre": ("Legendre_No_Norm", False, (-1.0, 1.0), False),          # :Legendre == :Legendre_No_Norm (:245-246)
    "legendre_no_norm": ("Legendre_No_Norm", False, (-1.0, 1.0), False),
    "legendre_norm": ("Legendre_Norm", False, (-1.0, 1.0), False),
    "fourier": ("Fourier", True, (-1.0, 1.0), False),
    "stoudenmire": ("Stoudenmire", True, (0.0, 1.0), False),
    "sahand": ("Sahand", True, (0.0, 1.0), False),
    "uniform": ("Uniform", False, (0.0, 1.0), False),
    "sltd": ("SLTD", False, (-1.0, 1.0), True),
    "sahand_legendre_time_dependent": ("SLTD", False, (-1.0, 1.0), True),
    "sahand_legendre": ("Sahand_Legendre", False, (-1.0, 1.0), True),
    "custom": ("Custom", False, None, False),
}


def encoding_info(name):
    key = _sym(name).lower()
    if key not in _ENCODINGS:
        raise ValueError(f"Unknown encoding {name!r}")
    return _ENCODINGS[key]


@dataclass(frozen=True)
class MPSOptions:
    """Keyword-for-keyword mirror of ``MPSOptions(; ...)`` (options.jl:106-143)."""

    verbosity: int = 1
    nsweeps: int = 10
    chi_max: int = 25
    eta: float = 0.01
    d: int = 5
    encoding: str = "Legendre_No_Norm"
    projected_basis: bool = False
    aux_basis_dim: int = 2
    cutoff: float = 1e-10
    update_iters: int = 1
    dtype: str = ""                # "" -> Float64, or ComplexF64 when the encoding is complex (:117)
    loss_grad: str = "KLD"
    bbopt: str = "TSGO"
    track_cost: bool = False
    rescale: Tuple[bool, bool] = (False, True)
    train_classes_separately: bool = False
    encode_classes_separately: bool = False
    return_encoding_meta_info: bool = False
    minmax: bool = True
    exit_early: bool = False
    sigmoid_transform: bool = True
    init_rng: int = 1234
    chi_init: int = 4
    log_level: int = 3
    data_bounds: Tuple[float, float] = (0.0, 1.0)
    use_legacy_ITensor: bool = False
    svd_alg: str = "divide_and_conquer"

    def __post_init__(self):
        object.__setattr__(self, "encoding", _sym(self.encoding))
        # a Symbol, or one per sweep (the reference's Options accepts arrays of length nsweeps, RealRealHighDimension.jl:691-713)
        for name in ("loss_grad", "bbopt"):
            v = getattr(self, name)
            if isinstance(v, (list, tuple)):
                if len(v) != self.nsweeps:
                    raise AssertionError(f"{name} must be one value or a sequence of length nsweeps")
                object.__setattr__(self, name, tuple(_sym(x) for x in v))
            else:
                object.__setattr__(self, name, _sym(v))
        canon, iscomplex, _, _ = encoding_info(self.encoding)
        if not self.dtype:
            object.__setattr__(self, "dtype", "ComplexF64" if iscomplex else "Float64")
        object.__setattr__(self, "rescale", (bool(self.rescale[0]), bool(self.rescale[1])))
        object.__setattr__(self, "data_bounds", (float(self.data_bounds[0]), float(self.data_bounds[1])))

    # _set_options (options.jl:373-384): functional update
    def set(self, **kw) -> "MPSOptions":
        return replace(self, **kw)

    def asdict(self):
        return {f.name: getattr(self, f.name) for f in fields(self)}


def safe_options(opts) -> MPSOptions:
    """safe_options (options.jl:398-408): accept MPSOptions (or a dict of its fields)."""
    if isinstance(opts, MPSOptions):
        return opts
    if isinstance(opts, dict):
        return MPSOptions(**opts)
    raise TypeError("opts must be an MPSOptions")


_NUMPY_DTYPES = {"Float64": "float64", "Float32": "float32", "ComplexF64": "complex128", "ComplexF32": "complex64"}


def numpy_dtype(name):
    """opts.dtype (a Julia DataType name, options.jl:117) as the NumPy dtype the engine is given (mpst_set_dataset's dtype).
    The reference's array engine takes Float64 only and sends everything else through its legacy ITensor engine; here the
    element-typed kernels (csrc/mpst_typed.hip) train all four."""
    import numpy as np
    key = str(name).replace("Complex{Float64}", "ComplexF64").replace("Complex{Float32}", "ComplexF32")
    if key not in _NUMPY_DTYPES:
        raise ValueError(f"dtype {name!r} is not one of {sorted(_NUMPY_DTYPES)}")
    return np.dtype(_NUMPY_DTYPES[key])


def engine_options(opts: MPSOptions, sweep: int = 0) -> dict:
    """Resolve the symbols sweep number `sweep` consumes (model_loss_func :318-327, model_bbopt :298-311)
    and reject what the array engine - like the reference's own array path - does not implement."""
    lg = opts.loss_grad[sweep] if isinstance(opts.loss_grad, tuple) else opts.loss_grad
    bbo = opts.bbopt[sweep] if isinstance(opts.bbopt, tuple) else opts.bbopt
    opts = replace(opts, loss_grad=lg, bbopt=bbo)
    loss = opts.loss_grad.upper()
    if loss not in ("KLD", "MSE"):
        if loss == "MIXED":
            raise RuntimeError("loss_grad=:Mixed is only implemented in the legacy ITensor path "
                               "(set use_legacy_ITensor=true in the Julia package)")
        raise ValueError(f"Unknown loss function {opts.loss_grad!r}")
    bb = opts.bbopt.upper()
    if bb in ("OPTIM", "OPTIMKIT"):
        # loss_functions.jl:166-170, same text
        raise RuntimeError("Optim/OptimKit based solvers currently unimplemented for this version, "
                           "set 'use_legacy_ITensor=true' in MPSOptions to enable")
    if bb not in ("TSGO", "GD"):
        raise ValueError(f"Unknown Black Box Optimiser {opts.bbopt!r}, options are [CustomGD, Optim, OptimKit]")
    if opts.use_legacy_ITensor:
        raise RuntimeError("use_legacy_ITensor=true selects the reference's ITensor engine, which this package does not ship")
    numpy_dtype(opts.dtype)          # Float64, Float32, ComplexF64, ComplexF32: the element types the engine trains in
    alg = {"divide_and_conquer": 0, "qr_iteration": 0, "recursive": 1, "jacobi": 1}.get(opts.svd_alg)
    if alg is None:
        raise ValueError(f"Unknown svd_alg {opts.svd_alg!r}")
    return dict(chi_max=opts.chi_max, eta=opts.eta, cutoff=opts.cutoff, update_iters=opts.update_iters, loss=loss,
                bbopt=bb, rescale=opts.rescale, train_classes_separately=opts.train_classes_separately, svd_alg=alg)
