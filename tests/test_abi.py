"""CPU-side checks of the drop-in boundary: the shared library loads and exports every symbol that
include/opmhip.h declares, struct sizes agree, and (without a GPU) create fails loudly instead of falling back."""
import ctypes as C

import pytest


def test_library_exports_every_declared_symbol(pkg):
    L = pkg.capi.lib()
    names = pkg.capi.declared_symbols()
    assert len(names) >= 12
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert L.opmhip_abi_version() == 7


def test_default_config_matches_flow_defaults(pkg):
    cfg = pkg.capi.Config()
    pkg.capi.lib().opmhip_default_config(C.byref(cfg))
    # linalg/FlowLinearSolverParameters.hpp:142-154
    assert (cfg.maxit, cfg.tolerance, cfg.ilu_relaxation) == (200, 1e-2, 0.9)
    assert cfg.reorder == pkg.capi.REORDER["graph_coloring"] and cfg.zero_diag_fix == 1
    # --linear-solver-configuration=ilu0, --cpr-reuse-setup=3 (FlowLinearSolverParameters.hpp:212-214)
    assert cfg.preconditioner == pkg.capi.PRECONDITIONER["ilu0"] and cfg.cpr_reuse_setup == 3 and cfg.cpr_async_setup == 0 and cfg.chain_length == 0


def test_no_gpu_means_loud_failure(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.capi.OpmHipError) as e:
        pkg.capi.HipSolver()
    assert e.value.code in (pkg.capi.NO_DEVICE, pkg.capi.DEVICE_ERROR)


def test_bad_config_rejected(pkg):
    L = pkg.capi.lib()
    cfg = pkg.capi.Config()
    L.opmhip_default_config(C.byref(cfg))
    cfg.abi_version = 99
    h = C.c_void_p()
    assert L.opmhip_create(C.byref(cfg), C.byref(h)) == pkg.capi.INVALID_ARGUMENT
    assert b"ABI" in L.opmhip_last_error(None)
