"""Per-cell end point of the oil-water capillary pressure (PCW, or what SWATINIT made of it during equilibration) on the
device: opmhip_fluid.pc_scaling + opmhip_set_pcw against the CPU oracle, bit for bit, and the physical point of it - the
state equil.equilibrate(swatinit=...) hands over stays at rest once the device has the rescaled curves."""
import numpy as np
import pytest

import oracle_bind
from helpers import wetgas_fluid

pytestmark = pytest.mark.gpu


def capillary_fluid(pkg, wet=False):
    """the SPE1 fluid with an oil-water capillary pressure curve of its own (SPE1's SWOF has none): 0.4 bar at the connate
    water saturation falling to 0.02 bar at Sw = 1"""
    fl = wetgas_fluid(pkg) if wet else pkg.fluid.spe1_fluid()[0]
    swof = [list(r) for r in fl.sat[0]["swof"]]
    s0, s1 = swof[0][0], swof[-1][0]
    for r in swof:
        x = (r[0] - s0) / (s1 - s0)
        r[3] = 0.4e5 * (1.0 - x) ** 2 + 0.02e5
    kw = dict(rock_pref=fl.rock_pref, rock_cr=fl.rock_cr, pc_scaling=True)
    return pkg.fluid.Fluid(fl.pvt, [dict(swof=swof, sgof=fl.sat[0]["sgof"])], **kw)


def scaled_case(pkg, nx, ny, nz, wet=False, seed=7):
    fl = capillary_fluid(pkg, wet)
    case = pkg.decks.cartesian_case(nx, ny, nz, state="mixed", fluid=fl, heterogeneous=True)
    rng = np.random.default_rng(seed)
    table_max = fl.sat[0]["swof"][0][3]
    pcw = table_max * rng.uniform(0.3, 3.0, case["Nb"])
    pcw[::5] = table_max                  # the branch "scaled == unscaled: factor 1"
    pcw[3::11] = 0.0                      # a cell without capillary pressure
    case["pcw"] = np.ascontiguousarray(pcw)
    if wet:   # the third primary-variable meaning on top: the pressure variable is then the gas pressure
        m = case["meaning"].copy()
        pv = case["pv"].reshape(-1, 3).copy()
        top = case["depth"] < np.quantile(case["depth"], 0.25)
        m[top] = 2
        pv[top, 2] = 1e-5
        case["meaning"], case["pv"] = m, np.ascontiguousarray(pv.reshape(-1))
    return case, table_max


def both(pkg, orc, case, reorder="line_coloring"):
    m = pkg.capi.HipModel(case, reorder=reorder)
    o = oracle_bind.OracleModel(orc, case)
    m.set_state(case["pv"], case["meaning"])
    o.set_state(case["pv"], case["meaning"])
    return m, o


@pytest.mark.parametrize("wet", [False, True])
def test_scaled_capillary_pressure_bitwise(pkg, orc, wet):
    case, table_max = scaled_case(pkg, 7, 6, 9, wet=wet)
    m, o = both(pkg, orc, case)
    a, b = m.iq(), o.iq()
    assert a.shape[1] == 19 and np.array_equal(a, b)
    # what the scaling does: p_o - p_w = factor x table value, with the derivative along
    plain = dict(case); plain.pop("pcw")
    u = pkg.capi.HipModel(plain, reorder="line_coloring")
    u.set_state(case["pv"], case["meaning"])
    q = u.iq()
    alpha = np.where(case["pcw"] == table_max, 1.0, case["pcw"] / table_max)
    pcow, pcow0 = a[:, 4, 0] - a[:, 3, 0], q[:, 4, 0] - q[:, 3, 0]
    np.testing.assert_allclose(pcow, alpha * pcow0, rtol=1e-12, atol=1e-6)
    same = case["pcw"] == table_max
    assert np.array_equal(a[same], q[same]) and not np.array_equal(a[~same], q[~same])
    assert np.all(pcow[case["pcw"] == 0.0] == 0.0)
    # only the water phase notices: its pressure and, through it, 1/B_w, mobility and density
    other = np.ones(19, bool); other[[3, 6, 9, 12]] = False
    assert np.array_equal(a[:, other], q[:, other])
    # set_pcw(None) brings the tables' own curves back
    m.set_pcw(None)
    assert np.array_equal(m.iq(), q)
    m.set_pcw(case["pcw"])
    assert np.array_equal(m.iq(), a)


@pytest.mark.parametrize("reorder", ["level_scheduling", "graph_coloring", "line_coloring"])
def test_jacobian_residual_and_newton_update_bitwise(pkg, orc, reorder):
    case, _ = scaled_case(pkg, 9, 7, 9, wet=(reorder == "graph_coloring"))
    src = pkg.decks.five_spot_source(case, rate_sm3_per_day=60.0)
    m, o = both(pkg, orc, case, reorder=reorder)
    for h in (m, o):
        h.set_source(src)
    dt = 86400.0
    jm, rm = m.assemble(dt, 0)
    jo, ro = o.assemble(dt, 0)
    assert np.array_equal(rm, ro) and np.array_equal(jm, jo)
    rng = np.random.default_rng(3)
    Nb, mean = case["Nb"], case["meaning"]
    dx = np.zeros((Nb, 3))
    dx[:, 0] = rng.uniform(-0.01, 0.01, Nb)
    dx[:, 1] = rng.uniform(-2e5, 2e5, Nb)
    dx[:, 2] = np.where(mean == 0, rng.uniform(-0.01, 0.01, Nb), np.where(mean == 1, rng.uniform(-1.0, 1.0, Nb), rng.uniform(-1e-6, 1e-6, Nb)))
    assert m.update(dx.reshape(-1)) == o.update(dx.reshape(-1))
    pm, mm = m.get_state()
    po, mo = o.get_state()
    assert np.array_equal(pm, po) and np.array_equal(mm, mo)
    jm, rm = m.assemble(dt, 1)
    jo, ro = o.assemble(dt, 1)
    assert np.array_equal(rm, ro) and np.array_equal(jm, jo)
    assert np.array_equal(m.convergence(dt)[3:6], o.convergence(dt)[3:6])


def test_set_pcw_is_refused_where_it_cannot_act(pkg):
    case = pkg.decks.cartesian_case(4, 4, 4, state="mixed")     # SPE1 fluid, no pc_scaling
    m = pkg.capi.HipModel(case)
    with pytest.raises(pkg.capi.OpmHipError) as e:
        m.set_pcw(np.full(case["Nb"], 1e4))
    assert "pc_scaling" in str(e.value)
    m.set_pcw(None)                                              # nothing to remove: fine
    case2, _ = scaled_case(pkg, 4, 4, 4)
    bad = case2["pcw"].copy(); bad[5] = np.nan
    m2 = pkg.capi.HipModel({k: v for k, v in case2.items() if k != "pcw"})
    with pytest.raises(pkg.capi.OpmHipError):
        m2.set_pcw(bad)


def test_swatinit_state_is_at_rest_with_the_rescaled_curves(pkg, orc):
    """equil.equilibrate(swatinit=...) with the DEVICE's property functions on a water-oil column: with the tables' own
    capillary pressure the imposed water saturations are out of equilibrium and the first assembly sees water flowing; with
    the per-cell end points (opmhip_set_pcw) the same state is (to discretisation error) at rest."""
    fl = capillary_fluid(pkg)
    nz, dz, top = 24, 4.0, 2500.0
    case = pkg.decks.cartesian_case(1, 1, nz, dz=dz, top=top, state="undersaturated", perturb=False, fluid=fl)
    depth = case["depth"]
    swof = fl.sat[0]["swof"]
    limits = dict(Swl=swof[0][0], Swu=swof[-1][0], Sgl=fl.sat[0]["sgof"][0][0], Sgu=fl.sat[0]["sgof"][-1][0])
    d = fl.pvt[0]["density"]
    zwoc = top + 0.75 * nz * dz
    rec = dict(datum=top, pressure=300e5, zwoc=zwoc, pcow_woc=0.05e5, zgoc=top, pcgo_goc=0.0, accuracy=0)
    props = pkg.capi.HipFluid(fl)
    rs0 = 0.5 * pkg.decks.rs_sat(fl, 300e5)
    swat = np.where(depth < zwoc - 20.0, 0.3, np.where(depth < zwoc, 0.6, 1.0))
    r = pkg.equil.equilibrate(props, (d[0], d[1], d[2]), rec, depth, (top, top + nz * dz), limits,
                              rs_func=lambda z, p, sat_gas=0.0: rs0, swatinit=swat)
    assert np.all(r["sg"] == 0.0) and np.ptp(r["pcw_scale"]) > 0.1
    pv = np.stack([r["sw"], r["po"], np.full(nz, rs0)], axis=1).reshape(-1)
    case["pv"], case["meaning"] = np.ascontiguousarray(pv), np.full(nz, 1, np.uint8)
    pcw = r["pcw_scale"] * swof[0][3]
    dt = 86400.0

    def water_imbalance(c):
        m, o = both(pkg, orc, c)
        jm, rm = m.assemble(dt, 0)
        jo, ro = o.assemble(dt, 0)
        assert np.array_equal(rm, ro) and np.array_equal(jm, jo)
        return np.abs(rm.reshape(-1, 3)[:, 1]).max()         # the water equation: storage term is zero, what is left is flux

    rest = water_imbalance(dict(case, pcw=np.ascontiguousarray(pcw)))
    moving = water_imbalance(case)
    assert moving > 0.0 and rest < 0.02 * moving
