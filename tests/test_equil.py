"""Equilibration (opm-autodiff_amd/equil.py) against the numbers the reference's tests/test_equil.cc expects for
tests/equil_liveoil.DATA (fixture tests/golden/equil.json, made by tests/golden/make_equil_fixtures.py).

The equilibration evaluates water / gas / live-oil densities, RsSat and both capillary-pressure curves through a
`probe` - here the CPU oracle's functions - so these expectations (1e-4 PERCENT in the reference) pin those oracle
functions: ConstantCompressibilityWaterPvt, DryGasPvt, LiveOilPvt (saturated and undersaturated branch, master-table
extension) and the SWOF / SGOF capillary pressures.  The GPU twin of this test (tests/test_gpu_equil.py) pins the
device's functions the same way."""
import importlib
import json
import os

import numpy as np
import pytest

import oracle_bind

pkg = importlib.import_module("opm-autodiff_amd")
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case():
    with open(os.path.join(GOLDEN, "equil.json")) as f:
        d = json.load(f)["liveoil"]
    w = d["pvtw"]
    fl = pkg.fluid.Fluid([dict(pvtw=[w["p_ref"], w["bw_ref"], w["cw"], w["mu_ref"], w["cv"]],
                               density=[d["density"]["oil"], d["density"]["water"], d["density"]["gas"]], pvdg=d["pvdg"], pvto=d["pvto"])],
                         [dict(swof=d["swof"], sgof=d["sgof"])], rock_pref=d["rock"]["p_ref"], rock_cr=d["rock"]["cr"])
    dz = np.array(d["grid"]["dz"])
    top = d["grid"]["tops"] + np.concatenate([[0.0], np.cumsum(dz)[:-1]])
    centre = top + 0.5 * dz
    limits = dict(Swl=d["swof"][0][0], Swu=d["swof"][-1][0], Sgl=d["sgof"][0][0], Sgu=d["sgof"][-1][0])
    rho = (d["density"]["oil"], d["density"]["water"], d["density"]["gas"])
    return d, fl, centre, (float(top[0]), float(top[-1] + dz[-1])), limits, rho


def check(d, r):
    e = d["expected"]
    rel = e["reltol_percent"] / 100.0      # BOOST_CHECK_CLOSE takes percent
    np.testing.assert_allclose([r["pw"][0], r["pw"][-1], r["po"][0], r["po"][-1]], [e["pw_first"], e["pw_last"], e["po_first"], e["po_last"]],
                               rtol=rel)
    # saturations: BOOST_CHECK_CLOSE is relative, and exact where the expectation is 0
    for k in ("sw", "so", "sg"):
        exp = np.array(e[k])
        np.testing.assert_allclose(r[k], exp, rtol=rel, atol=1e-12)
    np.testing.assert_allclose(r["rs"], e["rs"], rtol=rel)


def test_liveoil_deck_with_the_oracle_functions(orc):
    d, fl, centre, span, limits, rho = load_case()
    props = oracle_bind.OracleFluid(orc, fl)
    r = pkg.equil.equilibrate(props, rho, d["equil"], centre, span, limits, grav=d["gravity"])
    check(d, r)
    # hydrostatic consistency of what came out: dp/dz between neighbouring water-zone cells is rho_w g
    i = len(centre) - 1
    b = props.probe(0.5 * (r["pw"][i] + r["pw"][i - 1]))[0, 0]
    assert abs((r["pw"][i] - r["pw"][i - 1]) / (centre[i] - centre[i - 1]) - b * rho[1] * d["gravity"]) < 1e-3


def test_rk4_ivp_matches_closed_form():
    # y' = -2 y, y(0) = 1 on [0, 1]: RK4 global error ~ h^4
    ivp = pkg.equil.RK4IVP(lambda x, y: -2.0 * y, (0.0, 1.0), 1.0, 100)
    for x in (0.0, 0.013, 0.5, 0.999):   # not the end point itself: there the reference's index clamp returns the last but one node
        assert abs(ivp(x) - np.exp(-2.0 * x)) < 1e-8
    # integration towards smaller x (the "Up" direction of the pressure tables)
    up = pkg.equil.RK4IVP(lambda x, y: 3.0, (10.0, 4.0), 7.0, 50)
    assert abs(up(6.0) - (7.0 + 3.0 * (6.0 - 10.0))) < 1e-12


class DeadOilProps:
    """PVDO decks: the oil formation-volume factor is a table of its own (DeadOilPvt: 1/B linear in p, extrapolated);
    water, gas and both capillary pressures still come from the wrapped probe"""

    def __init__(self, inner, pvdo):
        self.inner = inner
        t = np.asarray(pvdo, float)
        self.p, self.invb = t[:, 0], 1.0 / t[:, 1]

    def probe(self, p, rs=0.0, sw=0.0, sg=0.0, **kw):
        out = self.inner.probe(p, rs=0.0, sw=sw, sg=sg, **kw).copy()
        pp = np.atleast_1d(np.asarray(p, float))
        slope = (self.invb[-1] - self.invb[-2]) / (self.p[-1] - self.p[-2]) if len(self.p) > 1 else 0.0
        lo = (self.invb[1] - self.invb[0]) / (self.p[1] - self.p[0]) if len(self.p) > 1 else 0.0
        v = np.interp(pp, self.p, self.invb)
        v = np.where(pp > self.p[-1], self.invb[-1] + slope * (pp - self.p[-1]), v)
        v = np.where(pp < self.p[0], self.invb[0] + lo * (pp - self.p[0]), v)
        out[:, 2] = v
        return out


@pytest.mark.parametrize("name", ["capillary", "capillary_overlap"])
def test_dead_oil_decks(orc, name):
    """tests/test_equil.cc DeckWithCapillary (:556-594) and DeckWithCapillaryOverlap (:596-654): regular and overlapping
    transition zones; pins the water / gas densities and the capillary-pressure inversion once more"""
    with open(os.path.join(GOLDEN, "equil.json")) as f:
        d = json.load(f)[name]
    w = d["pvtw"]
    dummy_pvto = [dict(rs=0.0, p=[1e5, 2e5], bo=[1.0, 0.999], mu=[1e-3, 1e-3]), dict(rs=100.0, p=[200e5, 300e5], bo=[1.2, 1.19], mu=[1e-3, 1e-3])]
    fl = pkg.fluid.Fluid([dict(pvtw=[w["p_ref"], w["bw_ref"], w["cw"], w["mu_ref"], w["cv"]],
                               density=[d["density"]["oil"], d["density"]["water"], d["density"]["gas"]], pvdg=d["pvdg"], pvto=dummy_pvto)],
                         [dict(swof=d["swof"], sgof=d["sgof"])])
    dz = np.array(d["grid"]["dz"])
    top = d["grid"]["tops"] + np.concatenate([[0.0], np.cumsum(dz)[:-1]])
    centre = top + 0.5 * dz
    limits = dict(Swl=d["swof"][0][0], Swu=d["swof"][-1][0], Sgl=d["sgof"][0][0], Sgu=d["sgof"][-1][0])
    rho = (d["density"]["oil"], d["density"]["water"], d["density"]["gas"])
    props = DeadOilProps(oracle_bind.OracleFluid(orc, fl), d["pvdo"])
    r = pkg.equil.equilibrate(props, rho, d["equil"], centre, (float(top[0]), float(top[-1] + dz[-1])), limits, grav=d["gravity"],
                              rs_func=lambda z, p, sat_gas=0.0: 0.0)
    e = d["expected"]
    rel = e["reltol_percent"] / 100.0
    np.testing.assert_allclose([r["pw"][0], r["pw"][-1], r["po"][-1]], [e["pw_first"], e["pw_last"], e["po_last"]], rtol=rel)
    for k in ("sw", "so", "sg"):
        np.testing.assert_allclose(r[k], e[k], rtol=rel, atol=1e-12)


# ---- wet gas decks: DeckWithLiveGas, DeckWithRSVDAndRVVD, DeckWithPBVDAndPDVD (tests/test_equil.cc:734-1004) -----------------
def wet_gas_setup(name, make_props):
    """-> (fixture dict, result of equilibrate) for one of the three PVTG decks; make_props(fluid) -> probe object"""
    with open(os.path.join(GOLDEN, "equil.json")) as f:
        d = json.load(f)[name]
    w = d["pvtw"]
    dummy_pvto = [dict(rs=0.0, p=[1e5, 2e5], bo=[1.0, 0.999], mu=[1e-3, 1e-3]), dict(rs=100.0, p=[200e5, 300e5], bo=[1.2, 1.19], mu=[1e-3, 1e-3])]
    fl = pkg.fluid.Fluid([dict(pvtw=[w["p_ref"], w["bw_ref"], w["cw"], w["mu_ref"], w["cv"]],
                               density=[d["density"]["oil"], d["density"]["water"], d["density"]["gas"]],
                               pvto=d.get("pvto", dummy_pvto), pvtg=d["pvtg"])],
                         [dict(swof=d["swof"], sgof=d["sgof"])])
    dz = np.array(d["grid"]["dz"])
    top = d["grid"]["tops"] + np.concatenate([[0.0], np.cumsum(dz)[:-1]])
    centre = top + 0.5 * dz
    limits = dict(Swl=d["swof"][0][0], Swu=d["swof"][-1][0], Sgl=d["sgof"][0][0], Sgu=d["sgof"][-1][0])
    rho = (d["density"]["oil"], d["density"]["water"], d["density"]["gas"])
    props = make_props(fl)
    rec = d["equil"]
    if "pvdo" in d:      # dead oil under wet gas: no dissolved gas (Miscibility::NoMixing)
        inner = props
        props = DeadOilProps(inner, d["pvdo"])
        props.probe_gas = inner.probe_gas
        rs_func = lambda z, p, sat_gas=0.0: 0.0
    elif "rsvd" in d:
        t = np.array(d["rsvd"])
        rs_func = pkg.equil.RsVD(props, t[:, 0], t[:, 1])
    elif "pbvd" in d:
        t = np.array(d["pbvd"])
        rs_func = pkg.equil.PBVD(props, t[:, 0], t[:, 1])
    else:
        rs_func = None
    if "rvvd" in d:
        t = np.array(d["rvvd"])
        rv_func = pkg.equil.RvVD(props, t[:, 0], t[:, 1])
    elif "pdvd" in d:
        t = np.array(d["pdvd"])
        rv_func = pkg.equil.PDVD(props, t[:, 0], t[:, 1])
    else:   # Rv saturated at the contact: p_contact = datum pressure + pcgo at the contact (initstateequil.hh:1695-1698)
        rv_func = pkg.equil.RvSatAtContact(props, rec["pressure"] + rec["pcgo_goc"])
    r = pkg.equil.equilibrate(props, rho, rec, centre, (float(top[0]), float(top[-1] + dz[-1])), limits, grav=d["gravity"],
                              rs_func=rs_func, rv_func=rv_func)
    return d, r


# Round 2 met the saturations of the three cells next to the gas-oil contact to 1e-4 absolute only.  Cause (round 3): the
# reference's WetGasPvt interpolates its (p_g, Rv) tables along guide lines ("RightExtreme": parallel to the saturated line
# at Rv = RvSat, vertical at Rv = 0), not vertically; with that policy in oracle/fluid.hpp (Tab2D, guide = 2) and on the
# device (assemble.hip tab2wg) all 60 saturations of every deck are within the reference's own tolerance (5e-14 absolute on
# the two decks whose expectations carry 16 digits).  The four pressures of DeckWithRSVDAndRVVD / DeckWithPBVDAndPDVD are
# printed with 10 / 8 digits and now differ by 4.3 / 16 Pa (3e-7 / 1.1e-6 relative, tolerance 1e-6 / 5e-6): they predate the
# saturations.


def check_wet_gas(d, r):
    e = d["expected"]
    rel, srel = e["reltol_percent"] / 100.0, e["sat_reltol_percent"] / 100.0
    np.testing.assert_allclose([r["pw"][0], r["pw"][-1], r["po"][0], r["po"][-1]], [e["pw_first"], e["pw_last"], e["po_first"], e["po_last"]], rtol=rel)
    for k in ("sw", "so", "sg"):      # BOOST_CHECK_CLOSE: relative, exact where the expectation is 0
        np.testing.assert_allclose(np.asarray(r[k]), np.asarray(e[k]), rtol=srel, atol=1e-12)
    np.testing.assert_allclose(r["rv"], e["rv"], rtol=rel)
    if "rs" in e:
        np.testing.assert_allclose(r["rs"], e["rs"], rtol=rel)


@pytest.mark.parametrize("name", ["livegas", "rsvd_rvvd", "pbvd_pdvd"])
def test_wet_gas_decks_with_the_oracle_functions(orc, name):
    """The three PVTG decks of tests/test_equil.cc pin the oracle's WetGasPvt: RvSat(p), 1/B_g(p, Rv) (saturated and
    undersaturated branch) through the gas density, and the RSVD / RVVD / PBVD / PDVD depth rules, to the reference's own
    tolerances (1e-4 % for the RSVD/RVVD deck)."""
    d, r = wet_gas_setup(name, lambda fl: oracle_bind.OracleFluid(orc, fl))
    check_wet_gas(d, r)


# ---- the remaining cases of tests/test_equil.cc: pressure tables per region, all-dead deck, capillary inversion --------------
class ConstProps:
    """initDefaultFluidSystem of the reference test (:119-182): formation-volume factors 1, no mixing, no capillary pressure"""

    def probe(self, p, rs=0.0, sw=0.0, sg=0.0, **kw):
        out = np.zeros((np.size(p), 8))
        out[:, [pkg.equil.INVBW, pkg.equil.INVBG, pkg.equil.INVBO]] = 1.0
        return out


def default_fluid_case():
    with open(os.path.join(GOLDEN, "equil.json")) as f:
        d = json.load(f)["default_fluid"]
    g = d["grid"]
    nx, nz = g["nx"], g["nz"]
    k = np.repeat(np.arange(nz), nx)                     # cell c = i + nx * k
    i = np.tile(np.arange(nx), nz)
    zmin = g["tops"] + k * g["d"]
    rho = (d["density"]["oil"], d["density"]["water"], d["density"]["gas"])
    return d, i, k, zmin, zmin + g["d"], zmin + 0.5 * g["d"], rho


def test_phase_pressure_of_the_default_fluid():
    """tests/test_equil.cc PhasePressure (:218-264)"""
    d, i, k, zmin, zmax, centre, rho = default_fluid_case()
    c = d["phase_pressure"]
    wat, oil, gas = pkg.equil.phase_pressure_tables(ConstProps(), rho, c["record"], (zmin.min(), zmax.max()), grav=d["gravity"])
    e, rel = c["expected"], c["reltol_percent"] / 100.0
    np.testing.assert_allclose([wat(centre[0]), wat(centre[-1]), oil(centre[0]), oil(centre[-1])],
                               [e["pw_first"], e["pw_last"], e["po_first"], e["po_last"]], rtol=rel)


@pytest.mark.parametrize("mapping", ["cell_subset", "reg_mapping"])
def test_equilibration_regions(mapping):
    """tests/test_equil.cc CellSubset (:266-369: coarse 2 x 1 x 2 blocks) and RegMapping (:371-475: EQLNUM): four regions, two
    EQUIL records; both describe the same partition of the 10 x 1 x 10 grid"""
    d, i, k, zmin, zmax, centre, rho = default_fluid_case()
    c = d["regions"]
    if mapping == "cell_subset":
        eqlnum = (i // 5) + 2 * (k // 5)          # ix = ic + cdim[0] * (jc + cdim[1] * kc), cdim = (2, 1, 2)
    else:
        eqlnum = np.where(k < 5, np.where(i < 5, 0, 1), np.where(i < 5, 2, 3))
    limits = dict(Swl=0.0, Swu=1.0, Sgl=0.0, Sgu=1.0)
    r = pkg.equil.equilibrate_regions(eqlnum, c["records"], ConstProps(), rho, centre, zmin, zmax, limits, grav=d["gravity"],
                                      rs_funcs=[lambda z, p, sat_gas=0.0: 0.0] * 4)
    e, rel = c["expected"], c["reltol_percent"] / 100.0
    # the reference reads the pressure tables at the cell centres (ptable.water / ptable.oil): po and the water pressure
    # before the end-point corrections of the saturation step, which only touch cells where a phase is absent
    wat0 = pkg.equil.phase_pressure_tables(ConstProps(), rho, c["records"][0], (0.0, 10.0), grav=d["gravity"])[0]
    wat3 = pkg.equil.phase_pressure_tables(ConstProps(), rho, c["records"][3], (0.0, 10.0), grav=d["gravity"])[0]
    np.testing.assert_allclose([wat0(centre[0]), wat3(centre[-1])], [e["pw_first"], e["pw_last"]], rtol=rel)
    np.testing.assert_allclose([r["po"][0]], [e["po_first"]], rtol=rel)
    # last cell: below the contact of its region the oil pressure follows the water pressure (no oil there); the table value
    oil3 = pkg.equil.phase_pressure_tables(ConstProps(), rho, c["records"][3], (0.0, 10.0), grav=d["gravity"])[1]
    np.testing.assert_allclose(oil3(centre[-1]), e["po_last"], rtol=rel)
    # the region loop: every cell belongs to exactly one region and was equilibrated with that region's record
    assert np.all(r["sw"] + r["so"] + r["sg"] == 1.0)
    for reg in range(4):
        cells = np.nonzero(eqlnum == reg)[0]
        one = pkg.equil.equilibrate(ConstProps(), rho, c["records"][reg], centre[cells], (zmin[cells].min(), zmax[cells].max()), limits,
                                    grav=d["gravity"], rs_func=lambda z, p, sat_gas=0.0: 0.0)
        for key in ("pw", "po", "pg", "sw", "sg"):
            np.testing.assert_array_equal(r[key][cells], one[key])


def dead_fluid(d):
    w = d["pvtw"]
    dummy_pvto = [dict(rs=0.0, p=[1e5, 2e5], bo=[1.0, 0.999], mu=[1e-3, 1e-3]), dict(rs=100.0, p=[200e5, 300e5], bo=[1.2, 1.19], mu=[1e-3, 1e-3])]
    return pkg.fluid.Fluid([dict(pvtw=[w["p_ref"], w["bw_ref"], w["cw"], w["mu_ref"], w["cv"]],
                                 density=[d["density"]["oil"], d["density"]["water"], d["density"]["gas"]], pvdg=d["pvdg"], pvto=dummy_pvto)],
                           [dict(swof=d["swof"], sgof=d["sgof"])])


def test_all_dead_deck(orc):
    """tests/test_equil.cc DeckAllDead (:477-502): dead oil, dry gas, datum in the water zone; the reference's own tolerance 0.1 %"""
    with open(os.path.join(GOLDEN, "equil.json")) as f:
        d = json.load(f)["alldead"]
    dz = np.array(d["grid"]["dz"])
    top = d["grid"]["tops"] + np.concatenate([[0.0], np.cumsum(dz)[:-1]])
    limits = dict(Swl=d["swof"][0][0], Swu=d["swof"][-1][0], Sgl=d["sgof"][0][0], Sgu=d["sgof"][-1][0])
    rho = (d["density"]["oil"], d["density"]["water"], d["density"]["gas"])
    props = DeadOilProps(oracle_bind.OracleFluid(orc, dead_fluid(d)), d["pvdo"])
    r = pkg.equil.equilibrate(props, rho, d["equil"], top + 0.5 * dz, (float(top[0]), float(top[-1] + dz[-1])), limits, grav=d["gravity"],
                              rs_func=lambda z, p, sat_gas=0.0: 0.0)
    e = d["expected"]
    np.testing.assert_allclose([r["pw"][0], r["pw"][-1], r["po"][-1]], [e["pw_first"], e["pw_last"], e["po_last"]], rtol=e["reltol_percent"] / 100.0)


def test_capillary_inversion(orc):
    """tests/test_equil.cc CapillaryInversion (:504-554): satFromPc (oil-water, gas-oil) and satFromSumOfPcs on the saturation
    tables of equil_capillary.DATA, the capillary pressures evaluated by the oracle's saturation functions"""
    with open(os.path.join(GOLDEN, "equil.json")) as f:
        all_ = json.load(f)
    d, e = all_["capillary"], all_["capillary_inversion"]
    props = oracle_bind.OracleFluid(orc, dead_fluid(d))
    pcow = lambda sw: float(props.probe(1e5, sw=sw)[0, pkg.equil.PCOW])
    pcgo = lambda sg: float(props.probe(1e5, sg=sg)[0, pkg.equil.PCGO])
    swl, swu, sgl, sgu = d["swof"][0][0], d["swof"][-1][0], d["sgof"][0][0], d["sgof"][-1][0]
    rel = e["reltol_percent"] / 100.0
    got = [pkg.equil.sat_from_pc(pcow, swl, swu, pc, increasing=False) for pc in e["oil_water"]["pc"]]
    np.testing.assert_allclose(got, e["oil_water"]["s"], rtol=rel, atol=1e-12)
    got = [pkg.equil.sat_from_pc(pcgo, sgl, sgu, pc, increasing=True) for pc in e["gas_oil"]["pc"]]
    np.testing.assert_allclose(got, e["gas_oil"]["s"], rtol=rel, atol=1e-12)
    got = [pkg.equil.sat_from_sum_of_pcs(pcow, pcgo, swl, swu, pc) for pc in e["gas_water"]["pc"]]
    np.testing.assert_allclose(got, e["gas_water"]["s"], rtol=rel, atol=1e-12)


def test_swatinit_deck(orc):
    """tests/test_equil.cc DeckWithSwatinit (:1006-1146, compiled out in the reference, its numbers kept): without SWATINIT the
    saturations of the plain capillary equilibrium; with it the imposed water saturations (clipped to Swl, and Swu where
    p_o < p_w) and per-cell rescaled oil-water curves that return p_o - p_w at those saturations.  Tolerance 0.1 % as there."""
    with open(os.path.join(GOLDEN, "equil.json")) as f:
        d = json.load(f)["swatinit"]
    dz = np.array(d["grid"]["dz"])
    top = d["grid"]["tops"] + np.concatenate([[0.0], np.cumsum(dz)[:-1]])
    limits = dict(Swl=d["swof"][0][0], Swu=d["swof"][-1][0], Sgl=d["sgof"][0][0], Sgu=d["sgof"][-1][0])
    rho = (d["density"]["oil"], d["density"]["water"], d["density"]["gas"])
    props = DeadOilProps(oracle_bind.OracleFluid(orc, dead_fluid(d)), d["pvdo"])
    args = (props, rho, d["equil"], top + 0.5 * dz, (float(top[0]), float(top[-1] + dz[-1])), limits)
    kw = dict(grav=d["gravity"], rs_func=lambda z, p, sat_gas=0.0: 0.0)
    e = d["expected"]
    rel = e["reltol_percent"] / 100.0
    plain = pkg.equil.equilibrate(*args, **kw)
    assert "pcw_scale" not in plain
    for k in ("sw", "so", "sg"):
        np.testing.assert_allclose(plain[k], e["without"][k], rtol=rel, atol=1e-12)
    r = pkg.equil.equilibrate(*args, swatinit=d["swatinit"], **kw)
    for k in ("sw", "so", "sg"):
        np.testing.assert_allclose(r[k], e["with_swatinit"][k], rtol=rel, atol=1e-12)
    # the capillary pressure of the rescaled curve at the computed saturation: pc_scaled of the reference's test
    pcow = np.array([r["pcw_scale"][c] * float(props.probe(1e5, sw=r["sw"][c])[0, pkg.equil.PCOW]) for c in range(len(dz))])
    np.testing.assert_allclose(pcow[:12], e["pcow_scaled"], rtol=rel)
    # cells where p_o < p_w keep the table's curve (and its value at Swu), like pc_scaled_truth = pc_original there
    np.testing.assert_array_equal(r["pcw_scale"][12:], 1.0)
    np.testing.assert_allclose(pcow[12:], d["swof"][-1][3], rtol=1e-12)
    # the gas-oil side is untouched by SWATINIT
    np.testing.assert_array_equal(r["sg"], plain["sg"])
    # and the rescaled state is an equilibrium: p_o - p_w equals the cell's own capillary pressure wherever water is mobile
    np.testing.assert_allclose((r["po"] - r["pw"])[:12], pcow[:12], rtol=1e-12)


def test_equil_item_9_horizontal_subdivision(orc):
    """EQUIL item 9 = -N (equilibrateHorizontal, initstateequil.hh:2027-2070; no deck of the reference uses it, so properties):
    with sharp contacts a cell that straddles a contact gets the thickness-weighted mix of the two zones, cells inside one zone
    are what the centre-point method gives, and with a capillary transition zone the slice average closes in on its limit"""
    rho = (700.0, 1000.0, 1.0)
    limits = dict(Swl=0.1, Swu=1.0, Sgl=0.0, Sgu=0.9)
    top = np.array([0.0, 2.0, 4.0, 6.0, 8.0])
    span = np.stack([top, top + 2.0], axis=1)
    centre = top + 1.0
    rec = dict(datum=0.0, pressure=1e7, zwoc=5.3, pcow_woc=0.0, zgoc=0.0, pcgo_goc=0.0, accuracy=0)
    kw = dict(grav=10.0, rs_func=lambda z, p, sat_gas=0.0: 0.0)
    r0 = pkg.equil.equilibrate(ConstProps(), rho, rec, centre, (0.0, 10.0), limits, **kw)
    assert list(r0["sw"]) == [0.1, 0.1, 0.1, 1.0, 1.0]            # cell 2 = [4, 6] has its centre above the contact
    r5 = pkg.equil.equilibrate(ConstProps(), rho, dict(rec, accuracy=-5), centre, (0.0, 10.0), limits, cell_zspan=span, **kw)
    np.testing.assert_allclose(r5["sw"], [0.1, 0.1, 0.6 * 0.1 + 0.4 * 1.0, 1.0, 1.0], rtol=1e-14)     # slices at 5.3 ... 5.9 are water
    np.testing.assert_allclose(r5["so"] + r5["sw"] + r5["sg"], 1.0, rtol=1e-14)
    for k in ("pw", "po", "pg"):
        np.testing.assert_allclose(r5[k][[0, 1, 3, 4]], r0[k][[0, 1, 3, 4]], rtol=1e-12)
    with pytest.raises(ValueError):
        pkg.equil.equilibrate(ConstProps(), rho, dict(rec, accuracy=-5), centre, (0.0, 10.0), limits, **kw)       # no cell_zspan
    with pytest.raises(ValueError):
        pkg.equil.equilibrate(ConstProps(), rho, dict(rec, accuracy=2), centre, (0.0, 10.0), limits, cell_zspan=span, **kw)
    with pytest.raises(ValueError):
        pkg.equil.equilibrate(ConstProps(), rho, dict(rec, accuracy=-1), centre, (0.0, 10.0), limits, cell_zspan=span[:, ::-1], **kw)
    # capillary transition zone (fixture 'capillary'): convergence of the slice average
    with open(os.path.join(GOLDEN, "equil.json")) as f:
        d = json.load(f)["capillary"]
    dz = np.array(d["grid"]["dz"])
    tp = d["grid"]["tops"] + np.concatenate([[0.0], np.cumsum(dz)[:-1]])
    lim = dict(Swl=d["swof"][0][0], Swu=d["swof"][-1][0], Sgl=d["sgof"][0][0], Sgu=d["sgof"][-1][0])
    rh = (d["density"]["oil"], d["density"]["water"], d["density"]["gas"])
    props = DeadOilProps(oracle_bind.OracleFluid(orc, dead_fluid(d)), d["pvdo"])
    sw = {}
    for n in (1, 4, 32):
        sw[n] = pkg.equil.equilibrate(props, rh, dict(d["equil"], accuracy=-n), tp + 0.5 * dz, (float(tp[0]), float(tp[-1] + dz[-1])), lim,
                                      grav=d["gravity"], rs_func=lambda z, p, sat_gas=0.0: 0.0, cell_zspan=np.stack([tp, tp + dz], axis=1))["sw"]
    e1, e4 = np.abs(sw[1] - sw[32]).max(), np.abs(sw[4] - sw[32]).max()
    assert 0.0 < e4 < 0.3 * e1 and e1 < 0.05       # the profile has kinks (two-row tables): no clean order, but it closes in
