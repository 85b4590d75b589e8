"""opm-autodiff_amd — MI355X-native Newton-iteration hot path for OPM Flow (assembly + ILU0/BiCGStab).

The directory name carries a hyphen (mandated layout), so import it with
``importlib.import_module("opm-autodiff_amd")``.  The compute path lives in ``libopmhip.so`` (HIP, gfx950),
reached through the C-ABI declared in ``include/opmhip.h``; ``capi`` is its ctypes binding.
"""
from . import capi, decks, equil, fluid, grid, mmio, newton, ras, thpres, transmissibility, wells  # noqa: F401
