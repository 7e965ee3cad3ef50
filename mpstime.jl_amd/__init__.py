"""mpstime.jl_amd - MI355X-native sweep engine behind MPSTime.jl's fitMPS API."""
from . import _lib
from .engine import SweepEngine
from ._lib import MPSTError, SVDError
