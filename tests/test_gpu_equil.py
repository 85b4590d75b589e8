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


def test_all_dead_deck_and_capillary_inversion_with_the_device_functions(pkg):
    """tests/test_equil.cc DeckAllDead (:477-502) and CapillaryInversion (:504-554) on top of the DEVICE's water / gas
    densities and capillary pressures"""
    import json, os
    from test_equil import GOLDEN, DeadOilProps, dead_fluid
    with open(os.path.join(GOLDEN, "equil.json")) as f:
        all_ = json.load(f)
    d = all_["alldead"]
    dz = np.array(d["grid"]["dz"])
    top = d["grid"]["tops"] + np.concatenate([[0.0], np.cumsum(dz)[:-1]])
    limits = dict(Swl=d["swof"][0][0], Swu=d["swof"][-1][0], Sgl=d["sgof"][0][0], Sgu=d["sgof"][-1][0])
    rho = (d["density"]["oil"], d["density"]["water"], d["density"]["gas"])
    props = DeadOilProps(pkg.capi.HipFluid(dead_fluid(d)), d["pvdo"])
    r = pkg.equil.equilibrate(props, rho, d["equil"], top + 0.5 * dz, (float(top[0]), float(top[-1] + dz[-1])), limits, grav=d["gravity"],
                              rs_func=lambda z, p, sat_gas=0.0: 0.0)
    e = d["expected"]
    np.testing.assert_allclose([r["pw"][0], r["pw"][-1], r["po"][-1]], [e["pw_first"], e["pw_last"], e["po_last"]], rtol=e["reltol_percent"] / 100.0)
    d, e = all_["capillary"], all_["capillary_inversion"]
    props = pkg.capi.HipFluid(dead_fluid(d))
    pcow = lambda sw: float(props.probe(1e5, sw=sw)[0, pkg.equil.PCOW])
    pcgo = lambda sg: float(props.probe(1e5, sg=sg)[0, pkg.equil.PCGO])
    swl, swu, sgl, sgu = d["swof"][0][0], d["swof"][-1][0], d["sgof"][0][0], d["sgof"][-1][0]
    rel = e["reltol_percent"] / 100.0
    np.testing.assert_allclose([pkg.equil.sat_from_pc(pcow, swl, swu, pc, increasing=False) for pc in e["oil_water"]["pc"]], e["oil_water"]["s"], rtol=rel, atol=1e-12)
    np.testing.assert_allclose([pkg.equil.sat_from_pc(pcgo, sgl, sgu, pc, increasing=True) for pc in e["gas_oil"]["pc"]], e["gas_oil"]["s"], rtol=rel, atol=1e-12)
    np.testing.assert_allclose([pkg.equil.sat_from_sum_of_pcs(pcow, pcgo, swl, swu, pc) for pc in e["gas_water"]["pc"]], e["gas_water"]["s"], rtol=rel, atol=1e-12)
