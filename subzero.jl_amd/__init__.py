"""subzero.jl_amd — MI355X (gfx950) engine for the per-timestep collision / forcing / rigid-body
path of Subzero.jl, behind the reference's own process API.

  csrc/      hand-written HIP kernels + the C-ABI (include/subzero_hip.h) -> libsubzero_hip.so
  capi.py    ctypes binding of the C-ABI
  host.py    World: the reference's timestep_* / floe_*_interaction! functions on top of it
  floe.py    host-side floe setup (the Floe constructor's derived columns)
  fields.py  synthetic floe fields and sub-floe points for the benchmark configurations
  tiles.py   spatial tile decomposition + ghost-floe halo exchange for multi-GPU runs
"""
from . import capi, floe
from .capi import SzError
from .host import World

__all__ = ["World", "SzError", "capi", "floe"]
