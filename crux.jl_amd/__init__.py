"""crux.jl_amd -- MI355X-native actor-learner hot path behind the Crux.jl solver interface.

The directory name contains a dot, so import it through the `crux_jl_amd` shim at the repository root:

    import crux_jl_amd as crux

Everything here talks to libcruxhip.so (hand-written HIP for gfx950) through the C ABI in include/cruxhip.h.
"""
from . import _lib
from ._lib import CruxError, LIB_PATH
from .api import *  # noqa: F401,F403
from . import api, dist
from . import logging as logging_   # LoggerParams, log, TBLogger, readtb, log_* (src/logging.jl); named logging_ to leave the stdlib name alone
from .logging import LoggerParams, TBLogger, readtb, aggregate_info, log
