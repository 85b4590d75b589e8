"""The device's fluid-system and saturation functions (opmhip_fluid_probe) against the oracle's, bit for bit, and
against the reference: the equilibration of tests/equil_liveoil.DATA run on top of the DEVICE functions reproduces the
numbers tests/test_equil.cc:656-732 expects (1e-4 percent)."""
import numpy as np
import pytest

import oracle_bind
from test_equil import check, check_wet_gas, load_case, wet_gas_setup

pytestmark = pytest.mark.gpu


def test_fluid_probe_bitwise_vs_oracle(pkg, orc):
    rng = np.random.default_rng(5)
    for fl in (pkg.fluid.spe1_fluid()[0], load_case()[1]):
        dev = pkg.capi.HipFluid(fl)
        ora = oracle_bind.OracleFluid(orc, fl)
        n = 4000
        p = rng.uniform(20e5, 600e5, n)
        rs_sat = ora.probe(p)[:, 3]
        rs = rs_sat * rng.uniform(0.0, 1.3, n)            # undersaturated and (clipped to the saturated curve) beyond
        sw, sg = rng.uniform(-0.05, 1.05, n), rng.uniform(-0.05, 1.05, n)
        a, b = dev.probe(p, rs, sw, sg), ora.probe(p, rs, sw, sg)
        assert np.array_equal(a, b)


def test_liveoil_deck_with_the_device_functions(pkg):
    d, fl, centre, span, limits, rho = load_case()
    props = pkg.capi.HipFluid(fl)
    r = pkg.equil.equilibrate(props, rho, d["equil"], centre, span, limits, grav=d["gravity"])
    check(d, r)


def test_fluid_probe_argument_errors(pkg):
    fl = pkg.fluid.spe1_fluid()[0]
    dev = pkg.capi.HipFluid(fl)
    with pytest.raises(pkg.capi.OpmHipError) as e:
        dev.probe([1e7], pvt_region=3)
    assert e.value.code == pkg.capi.INVALID_ARGUMENT
    s = pkg.capi.HipSolver()
    import ctypes as C
    assert pkg.capi.lib().opmhip_fluid_probe(s._h, 0, 0, 0, None, None, None, None, None) == pkg.capi.NOT_READY


@pytest.mark.parametrize("name", ["livegas", "rsvd_rvvd", "pbvd_pdvd"])
def test_wet_gas_decks_with_the_device_functions(pkg, name):
    """DeckWithLiveGas / DeckWithRSVDAndRVVD / DeckWithPBVDAndPDVD (tests/test_equil.cc:734-1004) through opmhip_fluid_probe and
    opmhip_gas_probe: pins the DEVICE's wet-gas tables (RvSat, 1/B_g(p, Rv) saturated and undersaturated) the same way"""
    d, r = wet_gas_setup(name, lambda fl: pkg.capi.HipFluid(fl))
    check_wet_gas(d, r)


def test_gas_probe_bitwise_vs_oracle(pkg, orc):
    from helpers import wetgas_fluid
    rng = np.random.default_rng(9)
    fl = wetgas_fluid(pkg)
    dev, ora = pkg.capi.HipFluid(fl), oracle_bind.OracleFluid(orc, fl)
    n = 4000
    p = rng.uniform(20e5, 650e5, n)
    rv = ora.probe_gas(p)[:, 2] * rng.uniform(0.0, 1.3, n)
    assert np.array_equal(dev.probe_gas(p, rv), ora.probe_gas(p, rv))
    dry = pkg.fluid.spe1_fluid()[0]
    a = pkg.capi.HipFluid(dry).probe_gas(p[:50], 0.0)
    assert np.array_equal(a[:, 0], oracle_bind.OracleFluid(orc, dry).probe(p[:50])[:, 1]) and not a[:, 2].any()
