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
    assert L.opmhip_abi_version() == 11


def test_default_config_matches_flow_defaults(pkg):
    cfg = pkg.capi.Config()
    pkg.capi.lib().opmhip_default_config(C.byref(cfg))
    # linalg/FlowLinearSolverParameters.hpp:142-154
    assert (cfg.maxit, cfg.tolerance, cfg.ilu_relaxation) == (200, 1e-2, 0.9)
    # the ordering and the AMG smoother are the library's own choices - the configuration bench.py measures;
    # the reference's accelerator default (graph_coloring, bda/BdaBridge.cpp:72-73) stays available under its name
    assert cfg.reorder == pkg.capi.REORDER["auto"] and cfg.cpr_amg_ilu_levels == -1 and cfg.zero_diag_fix == 1
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


def test_ctypes_structs_match_the_header(pkg, tmp_path):
    """the Python binding's opmhip_config / opmhip_result mirror the C structs field by field: size and every offset, as a C compiler
    sees include/opmhip.h (a field added on one side only would shift everything behind it silently)"""
    import ctypes
    import os
    import subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg_fields = [f[0] for f in pkg.capi.Config._fields_]
    res_fields = [f[0] for f in pkg.capi.Result._fields_]
    wells_fields = [f[0] for f in pkg.capi.Wells._fields_]      # incl. the multisegment leg (num_ms_wells, ms_apply, ms_user; ABI 8) and `distributed` (ABI 9)
    src = tmp_path / "layout.c"
    lines = ['#include <stddef.h>', '#include <stdio.h>', '#include "opmhip.h"', 'int main(void) {',
             '  printf("config %zu\\n", sizeof(opmhip_config));', '  printf("result %zu\\n", sizeof(opmhip_result));']
    for f in cfg_fields:
        lines.append('  printf("config.%s %%zu\\n", offsetof(opmhip_config, %s));' % (f, f))
    for f in res_fields:
        lines.append('  printf("result.%s %%zu\\n", offsetof(opmhip_result, %s));' % (f, f))
    lines.append('  printf("wells %zu\\n", sizeof(opmhip_wells));')
    for f in wells_fields:
        lines.append('  printf("wells.%s %%zu\\n", offsetof(opmhip_wells, %s));' % (f, f))
    lines += ['  return 0;', '}']
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    out = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    assert int(out["config"]) == ctypes.sizeof(pkg.capi.Config) and int(out["result"]) == ctypes.sizeof(pkg.capi.Result)
    for f in cfg_fields:
        assert int(out["config." + f]) == getattr(pkg.capi.Config, f).offset, f
    for f in res_fields:
        assert int(out["result." + f]) == getattr(pkg.capi.Result, f).offset, f
    assert int(out["wells"]) == ctypes.sizeof(pkg.capi.Wells)
    for f in wells_fields:
        assert int(out["wells." + f]) == getattr(pkg.capi.Wells, f).offset, f
