"""Saturation end-point scaling on the device (opmhip_set_endpoint_scaling: ENDSCALE / SCALECRS / SWL ... SOGCR / KRW, KRO, KRG /
KRWR ... / PCW, PCG per cell; call site in the reference: ebos/eclproblem.hh:1490-1498) against the CPU oracle, bit for bit:
the tables' own end points, intensive quantities, Jacobian and residual, the Newton update with its primary-variable
switches, with dry and with wet gas.  The oracle side of it is examined in tests/test_oracle_endscale.py."""
import numpy as np
import pytest

import oracle_bind
from helpers import wetgas_fluid
from test_oracle_endscale import corey_fluid, endscale_case

pytestmark = pytest.mark.gpu
REORDERS = ["level_scheduling", "graph_coloring", "line_coloring"]


def wet_corey_fluid(pkg):
    dry, wet = corey_fluid(pkg), wetgas_fluid(pkg)
    return pkg.fluid.Fluid(wet.pvt, dry.sat, rock_pref=wet.rock_pref, rock_cr=wet.rock_cr, pc_scaling=True)


def both(pkg, orc, case, reorder="line_coloring"):
    m = pkg.capi.HipModel(case, reorder=reorder)
    o = oracle_bind.OracleModel(orc, case)
    m.set_state(case["pv"], case["meaning"])
    o.set_state(case["pv"], case["meaning"])
    return m, o


def test_table_end_points_bitwise(pkg, orc):
    for fl in (corey_fluid(pkg), pkg.fluid.spe1_fluid()[0]):
        dev = pkg.capi.HipFluid(fl).sat_end_points(0)
        ora = oracle_bind.sat_end_points(orc, fl)
        assert np.array_equal(np.array([dev[k] for k in pkg.capi.EPS_FIELDS]), ora)


@pytest.mark.parametrize("three,vert", [(0, 0), (0, 1), (1, 1), (1, 2)])
def test_intensive_quantities_bitwise(pkg, orc, three, vert):
    case = endscale_case(pkg, orc, 7, 6, 9, three=three, vert=vert)
    m, o = both(pkg, orc, case)
    a, b = m.iq(), o.iq()
    assert a.shape[1] == 19 and np.array_equal(a, b)
    # the scaling is felt: mobilities and phase pressures differ from the unscaled ones, and come back when it is withdrawn
    plain = dict(case); plain.pop("endscale")
    u = pkg.capi.HipModel(plain, reorder="line_coloring")
    u.set_state(case["pv"], case["meaning"])
    q = u.iq()
    assert not np.array_equal(a[:, 9:12], q[:, 9:12]) and not np.array_equal(a[:, 3], q[:, 3])
    m.set_endpoint_scaling(None)
    assert np.array_equal(m.iq(), q)
    m.set_endpoint_scaling(case["endscale"])
    assert np.array_equal(m.iq(), a)


@pytest.mark.parametrize("reorder", REORDERS)
@pytest.mark.parametrize("wet", [False, True])
def test_jacobian_residual_and_update_bitwise(pkg, orc, reorder, wet):
    fl = wet_corey_fluid(pkg) if wet else None
    case = endscale_case(pkg, orc, 8, 7, 6, three=1, vert=2, fluid=fl)
    if wet:   # the third primary-variable meaning on top: its switches evaluate the (scaled) gas-oil capillary pressure
        mng, pv = case["meaning"].copy(), case["pv"].reshape(-1, 3).copy()
        top = case["depth"] < np.quantile(case["depth"], 0.25)
        mng[top] = 2
        pv[top, 2] = 1e-5
        case["meaning"], case["pv"] = mng, np.ascontiguousarray(pv.reshape(-1))
    src = pkg.decks.five_spot_source(case, rate_sm3_per_day=30.0)
    m, o = both(pkg, orc, case, reorder)
    for q in (m, o):
        q.set_source(src)
    dt = 2 * 86400.0
    for it in range(3):
        jm, rm = m.assemble(dt, it)
        jo, ro = o.assemble(dt, it)
        assert np.array_equal(jm, jo) and np.array_equal(rm, ro), it
        # a large update, the same on both sides: chops and meaning switches are exercised
        dx = np.random.default_rng(it).standard_normal(3 * case["Nb"]) * np.tile([0.15, 3e5, 0.15], case["Nb"])
        m.update(dx, 1.0)
        o.update(dx)
        pm, mm = m.get_state()
        po, mo = o.get_state()
        assert np.array_equal(mm, mo) and np.array_equal(pm, po), it
    assert len(set(mm.tolist())) >= 2


def test_argument_errors(pkg):
    fl = corey_fluid(pkg)
    case = pkg.decks.cartesian_case(4, 4, 3, state="mixed", fluid=fl)
    m = pkg.capi.HipModel(case)
    bad = dict(sat_scaling=1, swl=np.full(case["Nb"], 0.9), swu=np.full(case["Nb"], 0.5))     # an empty saturation interval
    with pytest.raises(pkg.capi.OpmHipError) as e:
        m.set_endpoint_scaling(bad)
    assert e.value.code == pkg.capi.INVALID_ARGUMENT
    with pytest.raises(pkg.capi.OpmHipError):
        m.set_endpoint_scaling(dict(krw=3))
    m.set_endpoint_scaling(dict(pcw=1, max_pcow=np.full(case["Nb"], 0.3e5)))
    with pytest.raises(pkg.capi.OpmHipError):      # PCW comes in through the end points now
        m.set_pcw(np.full(case["Nb"], 0.3e5))
    plain = pkg.decks.cartesian_case(4, 4, 3, state="mixed")    # a fluid without pc_scaling: no per-cell end points
    with pytest.raises(pkg.capi.OpmHipError):
        pkg.capi.HipModel(plain).set_endpoint_scaling(dict(sat_scaling=1))


def test_scaled_saturation_functions_probe_bitwise(pkg, orc):
    """opmhip_sat_probe: krw, kro, krg, pcow, pcgo with one set of scaled end points at random saturations - the device's
    functions against the oracle's, bit for bit, for every combination of the scaling options; unscaled too"""
    from test_oracle_endscale import scaled_points
    fl = corey_fluid(pkg)
    dev, ora = pkg.capi.HipFluid(fl), oracle_bind.OracleFluid(orc, fl)
    u = oracle_bind.sat_end_points(orc, fl)
    rng = np.random.default_rng(31)
    sw = rng.uniform(0.0, 1.05, 3000)
    sg = rng.uniform(-0.02, 1.0, 3000) * (1.05 - np.minimum(sw, 1.0))
    assert np.array_equal(dev.sat_probe(sw, sg), ora.sat_probe(sw, sg))
    for three in (0, 1):
        for vert in (0, 1, 2):
            pts = scaled_points(u, rng)
            es = dict(sat_scaling=1, three_point_kr=three, krw=vert, kro=vert, krg=vert, pcw=1, pcg=1)
            es.update({k: float(pts[f]) for f, k in enumerate(pkg.capi.EPS_FIELDS)})
            a, b = dev.sat_probe(sw, sg, es), ora.sat_probe(sw, sg, es)
            assert np.array_equal(a, b), (three, vert)
    es = dict(pcw=1, max_pcow=0.2e5)                                   # vertical scaling of one curve only
    assert np.array_equal(dev.sat_probe(sw, sg, es), ora.sat_probe(sw, sg, es))


def test_equilibration_with_scaled_end_points_on_the_device_functions(pkg, orc):
    """equil.equilibrate(endscale=...) on top of the DEVICE's scaled functions gives the oracle-based result bit for bit (same
    probes, same bisection), and the state it hands over is at rest: with the cell's end points on the device the initial
    residual of a water-oil column is far smaller than without them"""
    from test_oracle_endscale import endscale_column
    fl, centre, rec, rho, limits, es, span = endscale_column(pkg)
    rd = pkg.equil.equilibrate(pkg.capi.HipFluid(fl), rho, rec, centre, span, limits, endscale=es)
    ro = pkg.equil.equilibrate(oracle_bind.OracleFluid(orc, fl), rho, rec, centre, span, limits, endscale=es)
    for k in ("pw", "po", "pg", "sw", "so", "sg", "rs"):
        assert np.array_equal(rd[k], ro[k]), k
