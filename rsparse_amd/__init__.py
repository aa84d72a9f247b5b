"""rsparse_amd -- MI355X-native WRMF / ALS solver behind rsparse's WRMF operator boundary.

Only the hot path of the reference (R/model_WRMF.R + inst/include/wrmf_{implicit,explicit}.hpp):
  rsparse_amd.WRMF            host-side mirror of the R6 class
  rsparse_amd.als             als_implicit()/als_explicit() wrappers over the stateless C ABI
  rsparse_amd.engine          device-resident, row-sharded driver (one process per GPU)
  rsparse_amd.synth           synthetic interaction matrices for the BASELINE configs
The numerics live in csrc/ (HIP, gfx950) and are reached through include/rsparse_wrmf_hip.h.
"""
from . import _lib  # noqa: F401


def __getattr__(name):
    if name == "WRMF":
        from .wrmf import WRMF
        return WRMF
    raise AttributeError(name)
