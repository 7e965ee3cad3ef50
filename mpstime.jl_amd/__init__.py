"""mpstime.jl_amd - MI355X-native sweep engine behind MPSTime.jl's fitMPS / MPSOptions /
Encodings API.  The hot path lives in libmpstime_hip.so (csrc/); everything here is the
host-side mirror of the reference's interface for that path."""
from . import _lib
from ._lib import MPSTError, SVDError
from .engine import SweepEngine, comm_library, sweep_batch, sweep_batch_multi
from .options import MPSOptions, safe_options
from .encodings import (EncodedTimeSeriesSet, Encoding, encode_dataset, model_encoding, symbolic_encoding,
                        transform_data, transform_train_data, transform_test_data, legendre_encode,
                        legendre_encode_no_norm, fourier_encode, get_fourier_freqs, angle_encode, sahand_encode,
                        uniform_encode)
from .training import (TrainedMPS, fitMPS, fit_encoded, classify, generate_startingMPS, trendy_sine, save_trained_mps,
                       load_trained_mps, mps_content_digest)
from .distributed import Shard, split_encoded
from .imputation import (ImputationProblem, init_imputation_problem, MPS_impute, impute_dataset, kNN_impute, mar,
                         invert_test_transform)
from .jld2 import JLD2File, read_jld2, load_trained_mps_jld2
from . import options

__all__ = ["SweepEngine", "comm_library", "sweep_batch", "sweep_batch_multi", "MPSOptions", "safe_options", "EncodedTimeSeriesSet", "Encoding", "encode_dataset",
           "model_encoding", "symbolic_encoding", "transform_data", "TrainedMPS", "fitMPS", "fit_encoded", "classify",
           "generate_startingMPS", "trendy_sine", "Shard", "split_encoded", "MPSTError", "SVDError", "ImputationProblem",
           "init_imputation_problem", "save_trained_mps", "load_trained_mps", "mps_content_digest", "MPS_impute", "impute_dataset", "kNN_impute", "mar", "invert_test_transform",
           "JLD2File", "read_jld2", "load_trained_mps_jld2"]
