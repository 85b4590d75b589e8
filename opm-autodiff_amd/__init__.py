"""opm-autodiff_amd — MI355X-native Newton-iteration hot path for OPM Flow (assembly + ILU0/BiCGStab).

The directory name carries a hyphen (mandated layout), so import it with
``importlib.import_module("opm-autodiff_amd")``.
"""
from . import mmio  # noqa: F401
