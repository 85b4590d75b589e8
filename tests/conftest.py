import importlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure(path, makedir):
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", makedir], stdout=subprocess.DEVNULL)
    return path


@pytest.fixture(scope="session")
def pkg():
    """The product package (directory name has a hyphen, so importlib)."""
    return importlib.import_module("opm-autodiff_amd")


@pytest.fixture(scope="session")
def orc():
    """ctypes handle on the CPU oracle (test infrastructure)."""
    import oracle_bind
    alt = os.environ.get("ORACLE_LIB")   # another build of the same sources, e.g. tools/sanitize.sh's (ASan + UBSan + libstdc++ assertions)
    if alt:
        return oracle_bind.Oracle(alt)
    _ensure(os.path.join(ROOT, "oracle", "liboracle.so"), os.path.join(ROOT, "oracle"))
    return oracle_bind.Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))


@pytest.fixture(scope="session")
def golden():
    return GOLDEN
