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


# ---- the 10^6-cell case of BASELINE.json configs[1], built ONCE per session ----------------------------------------------------------
# tests/test_gpu_fullsize.py and tests/test_gpu_fullsize_oracle.py used to build the same case four times and assemble it on the oracle
# three times (two parametrised fixtures, the CPR test, the property tests): a quarter of those modules' time on a slow box.
@pytest.fixture(scope="session")
def case100(pkg):
    case = pkg.decks.cartesian_case(100, 100, 100, state="mixed", heterogeneous=False)
    src = pkg.decks.five_spot_source(case, rate_sm3_per_day=pkg.decks.BENCH_RATE_SM3_PER_DAY)
    return dict(case=case, src=src)


class _Oracle100:
    """ONE oracle model of that case, handed out in the state 'assembled at (dt = 1 day, Newton iteration 0)': Jacobian jo, residual ro.
    A test that moves it on (update, a second assembly) calls reset() when it is done - one assembly instead of a new model."""
    DT = 86400.0

    def __init__(self, orc, c):
        import oracle_bind
        self.case, self.src = c["case"], c["src"]
        self.o = oracle_bind.OracleModel(orc, self.case)
        self.reset()

    def reset(self):
        self.o.set_state(self.case["pv"], self.case["meaning"])
        self.o.set_source(self.src)
        self.jo, self.ro = self.o.assemble(self.DT, 0)


@pytest.fixture(scope="session")
def oracle100(orc, case100):
    return _Oracle100(orc, case100)
