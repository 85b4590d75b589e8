"""Saturation end-point scaling (ENDSCALE / SCALECRS / SWL ... SOGCR / KRW, KRO, KRG / KRWR ... / PCW, PCG) in the oracle
(oracle/fluid.hpp: eps_*, SatFunc::*Eps).  opm-material's EclEpsTwoPhaseLaw and opm-common's end-point extraction are not in
the reference tree, so this restatement is UNVERIFIED against upstream; what CAN be held: the defining properties of the
scaling (every curve reaches its scaled end points and values there), the identity when scaled = unscaled, equality with the
PCW-only path that tests/test_equil.cc's pc_scaled_truth pins, and the AD Jacobian against finite differences.  CPU only."""
import importlib

import numpy as np
import pytest

import oracle_bind
from test_oracle_assembly import _fd_check

EPS = None


def idx(name):
    pkg = importlib.import_module("opm-autodiff_amd")
    return pkg.capi.EPS_FIELDS.index(name)


def corey_fluid(pkg, pc_scaling=True):
    """SPE1's PVT with saturation tables whose end points are easy to read: Swl 0.15, Swcr 0.2, Sowcr 0.25, Sgcr 0.05, Sogcr 0.1"""
    fl = pkg.fluid.spe1_fluid()[0]
    sw = np.array([0.15, 0.2, 0.3, 0.45, 0.6, 0.75, 0.85, 1.0])
    krw = np.where(sw <= 0.2, 0.0, ((sw - 0.2) / 0.8) ** 2 * 0.7)
    krow = np.where(sw >= 0.75, 0.0, ((0.75 - sw) / 0.6) ** 2 * 0.9)
    pcow = 0.5e5 * ((1.0 - sw) / 0.85) ** 2 + 0.01e5
    swof = [[float(a), float(b), float(c), float(d)] for a, b, c, d in zip(sw, krw, krow, pcow)]
    sg = np.array([0.0, 0.05, 0.15, 0.3, 0.5, 0.7, 0.75, 0.85])
    krg = (np.maximum(sg - 0.05, 0.0) / 0.8) ** 1.5 * 0.8
    krog = np.where(sg >= 0.75, 0.0, ((0.75 - sg) / 0.75) ** 2 * 0.9)
    pcgo = 0.3e5 * (sg / 0.85) ** 2
    sgof = [[float(a), float(b), float(c), float(d)] for a, b, c, d in zip(sg, krg, krog, pcgo)]
    return pkg.fluid.Fluid(fl.pvt, [dict(swof=swof, sgof=sgof)], rock_pref=fl.rock_pref, rock_cr=fl.rock_cr, pc_scaling=pc_scaling)


def test_table_end_points(pkg, orc):
    fl = corey_fluid(pkg)
    u = oracle_bind.sat_end_points(orc, fl)
    want = dict(swl=0.15, swcr=0.2, swu=1.0, sowcr=0.25, sgl=0.0, sgcr=0.05, sgu=0.85, sogcr=0.1,
                max_pcow=0.5e5 + 0.01e5, max_pcgo=0.3e5, max_krw=0.7, max_krow=0.9, max_krg=0.8, max_krog=0.9)
    for k, v in want.items():
        assert abs(u[idx(k)] - v) < 1e-12 * max(1.0, abs(v)), (k, u[idx(k)], v)
    # values at the displacing phase's critical saturation: krw at Sw = 1 - Sowcr, krow at Sw = Swcr, krog at Sg = Sgcr, krg at So = Sogcr
    swof, sgof = np.array(fl.sat[0]["swof"]), np.array(fl.sat[0]["sgof"])
    assert abs(u[idx("krwr")] - np.interp(0.75, swof[:, 0], swof[:, 1])) < 1e-14
    assert abs(u[idx("krorw")] - np.interp(0.2, swof[:, 0], swof[:, 2])) < 1e-14
    assert abs(u[idx("krorg")] - np.interp(0.05, sgof[:, 0], sgof[:, 2])) < 1e-14
    assert abs(u[idx("krgr")] - np.interp(1.0 - 0.15 - 0.1, sgof[:, 0], sgof[:, 1])) < 1e-14


def scaled_points(u, rng):
    """a random but consistent set of scaled end points around the table's"""
    s = u.copy()
    s[idx("swl")] = u[idx("swl")] + rng.uniform(-0.05, 0.08)
    s[idx("swcr")] = s[idx("swl")] + rng.uniform(0.0, 0.1)
    s[idx("swu")] = 1.0 - rng.uniform(0.0, 0.05)
    s[idx("sowcr")] = rng.uniform(0.1, 0.3)
    s[idx("sgl")] = 0.0
    s[idx("sgcr")] = rng.uniform(0.0, 0.1)
    s[idx("sgu")] = 1.0 - s[idx("swl")] - rng.uniform(0.0, 0.05)
    s[idx("sogcr")] = rng.uniform(0.05, 0.25)
    for k in ("max_pcow", "max_pcgo", "max_krw", "max_krow", "max_krg", "max_krog"):
        s[idx(k)] = u[idx(k)] * rng.uniform(0.5, 1.3)
    s[idx("max_krog")] = s[idx("max_krow")]    # KRO is ONE keyword: the oil curve's maximum in both two-phase systems
    s[idx("krwr")] = s[idx("max_krw")] * rng.uniform(0.2, 0.7)
    s[idx("krorw")] = s[idx("max_krow")] * rng.uniform(0.3, 0.9)
    s[idx("krgr")] = s[idx("max_krg")] * rng.uniform(0.3, 0.9)
    s[idx("krorg")] = s[idx("max_krog")] * rng.uniform(0.3, 0.9)
    return s


@pytest.mark.parametrize("three", [0, 1])
@pytest.mark.parametrize("vert", [0, 1, 2])
def test_curves_reach_their_scaled_end_points(pkg, orc, three, vert):
    fl = corey_fluid(pkg)
    u = oracle_bind.sat_end_points(orc, fl)
    rng = np.random.default_rng(10 * three + vert)
    es = dict(sat_scaling=1, three_point_kr=three, krw=vert, kro=vert, krg=vert, pcw=1, pcg=1)
    for trial in range(20):
        s = scaled_points(u, rng)
        g = lambda k: s[idx(k)]
        probe = lambda sw, sg: oracle_bind.sat_probe_eps(orc, fl, es, s, [sw], [sg])[0]   # krw, kro, krg, pcow, pcgo
        vmax = lambda k: g(k) if vert else u[idx(k)]
        # water: immobile up to Swcr, its maximum at Swu
        assert abs(probe(g("swcr"), 0.0)[0]) < 1e-13 and probe(g("swl"), 0.0)[0] == 0.0
        assert abs(probe(g("swu"), 0.0)[0] - vmax("max_krw")) < 1e-13
        # gas: immobile up to Sgcr, its maximum at Sgu (with connate water)
        assert abs(probe(g("swl"), g("sgcr"))[2]) < 1e-13
        assert abs(probe(g("swl"), g("sgu"))[2] - vmax("max_krg")) < 1e-13
        # oil: its maximum at connate water without gas; immobile at Sowcr (no gas) and at Sogcr (connate water)
        assert abs(probe(g("swl"), 0.0)[1] - vmax("max_krow")) < 1e-12 * 1.0 + 1e-13
        assert abs(probe(1.0 - g("sowcr"), 0.0)[1]) < 1e-13
        assert abs(probe(g("swl"), 1.0 - g("swl") - g("sogcr"))[1]) < 1e-13
        # capillary pressures: their scaled maxima at the connate water / maximum gas saturation
        assert abs(probe(g("swl"), 0.0)[3] - g("max_pcow")) < 1e-9
        assert abs(probe(g("swl"), g("sgu"))[4] - g("max_pcgo")) < 1e-9
        if three and vert == 2:   # the third point: krw at Sw = 1 - Sowcr - Sgl takes KRWR, krg at So = Sogcr takes KRGR
            assert abs(probe(1.0 - g("sowcr") - g("sgl"), 0.0)[0] - g("krwr")) < 1e-13
            assert abs(probe(g("swl"), 1.0 - g("swl") - g("sogcr"))[2] - g("krgr")) < 1e-13
        # monotone where the tables are (three-point VERTICAL scaling presumes the three-point saturation scaling that puts
        # the table's critical point under the scaled one: without it the curve jumps at that saturation, by construction)
        sw = np.linspace(g("swl"), g("swu"), 41)
        kr = oracle_bind.sat_probe_eps(orc, fl, es, s, sw, np.zeros_like(sw))
        assert np.all(np.diff(kr[:, 3]) <= 1e-9)
        if three or vert < 2:
            assert np.all(np.diff(kr[:, 0]) >= -1e-13)


def test_identity_when_scaled_equals_unscaled(pkg, orc):
    fl = corey_fluid(pkg)
    u = oracle_bind.sat_end_points(orc, fl)
    rng = np.random.default_rng(2)
    sw = rng.uniform(0.1, 1.0, 400)
    sg = rng.uniform(0.0, 1.0, 400) * (1.0 - sw)
    plain = oracle_bind.OracleFluid(orc, fl).probe(np.full(len(sw), 1e7), sw=sw, sg=sg)
    for three in (0, 1):
        for vert in (0, 1, 2):
            es = dict(sat_scaling=1, three_point_kr=three, krw=vert, kro=vert, krg=vert, pcw=1, pcg=1)
            got = oracle_bind.sat_probe_eps(orc, fl, es, u, sw, sg)
            np.testing.assert_allclose(got[:, 3], plain[:, 4], rtol=1e-12, atol=1e-9)    # pcow
            np.testing.assert_allclose(got[:, 4], plain[:, 5], rtol=1e-12, atol=1e-9)    # pcgo
    # relative permeabilities: through a model (the probe of the unscaled functions does not report them)
    case = pkg.decks.cartesian_case(5, 4, 6, state="mixed", fluid=fl, heterogeneous=True)
    o = oracle_bind.OracleModel(orc, case)
    o.set_state(case["pv"], case["meaning"])
    a = o.iq()
    o.set_endpoint_scaling(dict(sat_scaling=1, three_point_kr=1, krw=2, kro=2, krg=2, pcw=1, pcg=1))   # every array absent: the tables' own points
    b = o.iq()
    np.testing.assert_allclose(b[:, :, 0], a[:, :, 0], rtol=2e-12, atol=1e-9)
    o.set_endpoint_scaling(None)
    assert np.array_equal(o.iq(), a)


def test_pcw_only_scaling_equals_the_pinned_pcw_path(pkg, orc):
    """PCW handed in as an end point (pcw = 1, nothing else scaled) must be the very curve opmhip_set_pcw / SWATINIT use - the
    one tests/test_equil.cc's pc_scaled_truth pins (tests/test_equil.py::test_swatinit_deck)"""
    fl = corey_fluid(pkg)
    case = pkg.decks.cartesian_case(5, 4, 6, state="mixed", fluid=fl, heterogeneous=True)
    rng = np.random.default_rng(4)
    pcw = fl.sat[0]["swof"][0][3] * rng.uniform(0.3, 2.5, case["Nb"])
    a = oracle_bind.OracleModel(orc, dict(case, pcw=pcw))
    b = oracle_bind.OracleModel(orc, dict(case, endscale=dict(pcw=1, max_pcow=pcw)))
    for m in (a, b):
        m.set_state(case["pv"], case["meaning"])
    assert np.array_equal(a.iq(), b.iq())
    ja, ra = a.assemble(86400.0, 0)
    jb, rb = b.assemble(86400.0, 0)
    assert np.array_equal(ja, jb) and np.array_equal(ra, rb)


def endscale_case(pkg, orc, nx=5, ny=4, nz=4, three=1, vert=2, seed=3, fluid=None):
    fl = fluid or corey_fluid(pkg)
    case = pkg.decks.cartesian_case(nx, ny, nz, state="mixed", fluid=fl, heterogeneous=True)
    u = oracle_bind.sat_end_points(orc, fl)
    rng = np.random.default_rng(seed)
    pts = np.array([scaled_points(u, rng) for _ in range(case["Nb"])])
    es = dict(sat_scaling=1, three_point_kr=three, krw=vert, kro=vert, krg=vert, pcw=1, pcg=1)
    for f, name in enumerate(pkg.capi.EPS_FIELDS):
        es[name] = np.ascontiguousarray(pts[:, f])
    case["endscale"] = es
    return case


@pytest.mark.parametrize("three,vert", [(0, 1), (1, 2)])
def test_jacobian_with_scaled_end_points_matches_finite_differences(pkg, orc, three, vert):
    case = endscale_case(pkg, orc, three=three, vert=vert)
    m = oracle_bind.OracleModel(orc, case)
    m.set_state(case["pv"], case["meaning"])
    # the end points matter: the linearisation differs from the unscaled one
    j1, r1 = m.assemble(86400.0, 0)
    plain = dict(case); plain.pop("endscale")
    p = oracle_bind.OracleModel(orc, plain)
    p.set_state(case["pv"], case["meaning"])
    j0, r0 = p.assemble(86400.0, 0)
    assert not np.array_equal(j0, j1)
    _fd_check(case, m, tol=5e-5)


def endscale_column(pkg, nz=30):
    """a 60 m column across both contacts with per-cell scaled end points and capillary-pressure maxima"""
    fl = corey_fluid(pkg)
    dz = 2.0
    centre = 2000.0 + dz * (np.arange(nz) + 0.5)
    rec = dict(datum=2025.0, pressure=250e5, zwoc=2045.0, pcow_woc=0.0, zgoc=2025.0, pcgo_goc=0.0, accuracy=0)
    rho = tuple(fl.pvt[0]["density"])
    rng = np.random.default_rng(6)
    es = dict(sat_scaling=1, pcw=1, pcg=1,
              swl=0.15 + rng.uniform(-0.04, 0.06, nz), swu=np.full(nz, 1.0), sgl=np.zeros(nz), sgu=np.zeros(nz),
              max_pcow=0.51e5 * rng.uniform(0.6, 1.5, nz), max_pcgo=0.3e5 * rng.uniform(0.6, 1.5, nz))
    es["sgu"] = 1.0 - es["swl"]
    es["swcr"] = es["swl"] + 0.05
    limits = dict(Swl=0.15, Swu=1.0, Sgl=0.0, Sgu=0.85)
    return fl, centre, rec, rho, limits, es, (2000.0, 2000.0 + dz * nz)


def test_equilibration_with_scaled_end_points(pkg, orc):
    """equil.equilibrate(endscale=...): every cell inverts ITS scaled capillary-pressure curves between ITS end points
    (satFromPc with the cell's scaled drainage end points, ebos/equil/equilibrationhelpers.hh:730-960).  No reference numbers
    exist for an ENDSCALE deck (the tree's equil decks have none): held by properties - defaults reproduce the unscaled run,
    the result is in capillary equilibrium with the cell's own curves, saturations respect the cell's own end points."""
    fl, centre, rec, rho, limits, es, span = endscale_column(pkg)
    props = oracle_bind.OracleFluid(orc, fl)
    plain = pkg.equil.equilibrate(props, rho, rec, centre, span, limits)
    same = pkg.equil.equilibrate(props, rho, rec, centre, span, limits, endscale=dict(sat_scaling=1, pcw=1, pcg=1))   # no arrays: the tables' own points
    for k in ("pw", "po", "pg", "sw", "so", "sg", "rs"):
        np.testing.assert_allclose(same[k], plain[k], rtol=1e-9, atol=1e-9)
    r = pkg.equil.equilibrate(props, rho, rec, centre, span, limits, endscale=es)
    sw, sg, so = r["sw"], r["sg"], r["so"]
    assert np.all(sw >= es["swl"] - 1e-12) and np.all(sw <= 1.0 + 1e-12) and np.all(sg >= -1e-12) and np.all(sg <= es["sgu"] + 1e-12)
    np.testing.assert_allclose(sw + sg + so, 1.0, atol=1e-12)
    assert not np.allclose(sw, plain["sw"], atol=1e-3)                  # the end points matter
    top, bottom = centre < rec["zgoc"] - 15.0, centre > rec["zwoc"] + 8.0
    assert np.allclose(sw[bottom], 1.0) and np.allclose(sg[top], es["sgu"][top]) and np.allclose(sw[top], es["swl"][top])
    # capillary equilibrium with the cell's OWN curves wherever a saturation is strictly inside its interval
    for c in range(len(centre)):
        e = {k: (float(v[c]) if np.ndim(v) else v) for k, v in es.items()}
        pc = props.sat_probe(sw[c], sg[c], e)[0]
        if es["swl"][c] + 1e-6 < sw[c] < 1.0 - 1e-6 and so[c] > 1e-6:
            assert abs(pc[3] - (r["po"][c] - r["pw"][c])) < 1.0          # Pa
        if 1e-6 < sg[c] < es["sgu"][c] - 1e-6 and so[c] > 1e-6:
            assert abs(pc[4] - (r["pg"][c] - r["po"][c])) < 1.0
