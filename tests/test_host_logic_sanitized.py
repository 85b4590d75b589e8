"""The library's host-side index logic (csrc/reorder.cpp: orderings, L/U split, tiles, launch schedules, stencil tables; csrc/fluid_tables.cpp:
table blobs) built with g++ under AddressSanitizer + UBSan + libstdc++'s container assertions and run over grids from 1 to 729 000 rows,
decomposed subdomains with ghost columns, irregular patterns with rows of up to 20 blocks, every ordering and chain lengths from 1 to 64
(tests/san/host_logic_san.cpp, which also checks the invariants every ordering must keep).  No GPU: the harness serves the three HIP
runtime calls of reorder.cpp from the host heap.  The GPU sanitizers are not available on the pool; this is the CPU build the brief asks
to run them on."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "opm-autodiff_amd", "csrc")


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_reorder_and_fluid_tables_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "host_logic_san")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           "-D_GLIBCXX_ASSERTIONS", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           os.path.join(ROOT, "tests", "san", "host_logic_san.cpp"), os.path.join(CSRC, "reorder.cpp"), os.path.join(CSRC, "fluid_tables.cpp"), "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("OPMHIP_TUNING", None)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert "all checks passed" in out and "FAILED" not in out
    assert "runtime error" not in out and "AddressSanitizer" not in out and "LeakSanitizer" not in out, out[-4000:]
    assert out.count("\nok  ") > 150   # every case of the list ran
