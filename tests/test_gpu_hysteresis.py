"""Relative-permeability hysteresis on the device (opmhip_set_hysteresis: SATOPTS HYSTER / EHYSTR item 2 in {0, 1} / IMBNUM;
updateHysteresis_ inside opmhip_begin_time_step, ebos/eclproblem.hh:1060, 2603-2626) against the CPU oracle, bit for bit:
turning points and shifts, intensive quantities, Jacobian, residual and the Newton update over time steps that take the
saturations through drainage -> imbibition -> drainage; with wet gas; with end-point scaling of the drainage AND of the
imbibition curves; restart of the turning points.  The law itself is examined in tests/test_oracle_hysteresis.py."""
import numpy as np
import pytest

import helpers
import oracle_bind
from test_oracle_endscale import corey_fluid, scaled_points

pytestmark = pytest.mark.gpu


def both(pkg, orc, case, reorder="line_coloring"):
    m = pkg.capi.HipModel(case, reorder=reorder)
    o = oracle_bind.OracleModel(orc, case)
    for q in (m, o):
        q.set_state(case["pv"], case["meaning"])
    return m, o


def same_hyst(m, o):
    return all(np.array_equal(a, b) for a, b in zip(m.hysteresis(), o.hysteresis()))


def same_state(m, o):
    pm, mm = m.get_state()
    po, mo = o.get_state()
    return np.array_equal(mm, mo) and np.array_equal(pm, po)


def run_reversal(pkg, orc, m, o, case, steps=6, its=3):
    """time steps under a source pair whose sign flips half way: gas and water advance, then retreat - scanning curves on the way
    back; every time step begins with begin_time_step on both sides"""
    src = pkg.decks.five_spot_source(case, rate_sm3_per_day=40.0)
    moved = 0
    scanning = 0
    for step in range(steps):
        s = src if step < steps // 2 else -src
        dt = (1.0 + step) * 86400.0
        for q in (m, o):
            q.set_source(s)
            q.begin_time_step(dt)
        assert same_hyst(m, o), step
        assert np.array_equal(m.iq(), o.iq()), step
        h = m.hysteresis()
        so = m.iq()[:, 1, 0]
        scanning += int(np.sum(1.0 - so > h[0]))
        for it in range(its):
            jm, rm = m.assemble(dt, it)
            jo, ro = o.assemble(dt, it)
            assert np.array_equal(jm, jo) and np.array_equal(rm, ro), (step, it)
            x, res = o.solve(tol=1e-6, maxit=200, w=0.9, mode="post_scale", reorder="none")
            m.update(x, 1.0)
            o.update(x)
            assert same_state(m, o), (step, it)
        for q in (m, o):
            q.end_time_step(dt)
        if step > 0:
            moved += int(not np.array_equal(h[0], prev[0])) + int(not np.array_equal(h[2], prev[2]))
        prev = [a.copy() for a in h]
    return moved, scanning


@pytest.mark.parametrize("kr_model", [0, 1])
@pytest.mark.parametrize("wetgas", [False, True])
def test_time_steps_with_reversal_bitwise(pkg, orc, kr_model, wetgas):
    case = helpers.hysteresis_case(pkg, 8, 7, 6, wetgas=wetgas, heterogeneous=True)
    m, o = both(pkg, orc, case)
    plain = m.iq().copy()
    for q in (m, o):
        q.set_hysteresis(kr_model, case["imbnum"])
    assert same_hyst(m, o) and np.all(m.hysteresis()[0] == 2.0)
    assert np.array_equal(m.iq(), o.iq())
    if kr_model == 0:
        assert np.array_equal(m.iq(), plain)          # nothing seen yet: every curve is the drainage curve
    moved, scanning = run_reversal(pkg, orc, m, o, case)
    assert moved > 0 and scanning > 0                 # turning points moved over the steps, and cells sat on scanning curves
    # the keyword goes out of force: the plain functions are back on both sides
    for q in (m, o):
        q.set_hysteresis(None)
    assert np.array_equal(m.iq(), o.iq())


@pytest.mark.parametrize("three,vert,imb_points", [(0, 1, False), (1, 2, True)])
def test_with_end_point_scaling_bitwise(pkg, orc, three, vert, imb_points):
    """ENDSCALE on the drainage curves, and (imb_points) ISWL ... on the imbibition curves: two saturation regions of Corey
    shape, the imbibition one trapping more gas and oil"""
    base = corey_fluid(pkg)
    d = base.sat[0]
    swof, sgof = np.array(d["swof"]), np.array(d["sgof"])
    swi, sgi = swof.copy(), sgof.copy()
    sgi[:, 1] = (np.maximum(sgof[:, 0] - 0.15, 0.0) / 0.7) ** 1.5 * 0.8          # gas immobile below 0.15
    swi[:, 2] = np.where(swof[:, 0] >= 0.7, 0.0, ((0.7 - swof[:, 0]) / 0.55) ** 2 * 0.9)   # oil immobile from Sw = 0.7 on
    swi[:, 1] = swof[:, 1] * 0.85
    sgi[:, 2] = sgof[:, 2] * 0.9
    fl = pkg.fluid.Fluid(base.pvt, [d, dict(swof=swi.tolist(), sgof=sgi.tolist())], rock_pref=base.rock_pref, rock_cr=base.rock_cr, pc_scaling=True)
    case = pkg.decks.cartesian_case(7, 6, 6, state="mixed", fluid=fl, heterogeneous=True)
    Nb = case["Nb"]
    case["satnum"] = np.zeros(Nb, np.int32)
    rng = np.random.default_rng(11)
    uD, uI = oracle_bind.sat_end_points(orc, fl, 0), oracle_bind.sat_end_points(orc, fl, 1)
    es = dict(sat_scaling=1, three_point_kr=three, krw=vert, kro=vert, krg=vert, pcw=1, pcg=1)
    ptsD = np.array([scaled_points(uD, rng) for _ in range(Nb)])
    for f, name in enumerate(pkg.capi.EPS_FIELDS):
        es[name] = np.ascontiguousarray(ptsD[:, f])
    case["endscale"] = es
    imb = None
    if imb_points:
        ptsI = np.array([scaled_points(uI, rng) for _ in range(Nb)])
        imb = {name: np.ascontiguousarray(ptsI[:, f]) for f, name in enumerate(pkg.capi.EPS_FIELDS)}
    imbnum = np.ones(Nb, np.int32)
    m, o = both(pkg, orc, case)
    for q in (m, o):
        q.set_hysteresis(1, imbnum, imb)
    assert np.array_equal(m.iq(), o.iq())
    moved, scanning = run_reversal(pkg, orc, m, o, case, steps=4, its=2)
    assert scanning > 0


def test_restart_of_the_turning_points(pkg, orc):
    case = helpers.hysteresis_case(pkg, 6, 5, 4, heterogeneous=True)
    m, o = both(pkg, orc, case)
    for q in (m, o):
        q.set_hysteresis(0, case["imbnum"])
    rng = np.random.default_rng(2)
    ow, go = rng.uniform(0.2, 0.9, case["Nb"]), rng.uniform(0.5, 1.0, case["Nb"])
    for q in (m, o):
        q.set_hysteresis_params(ow, go)
    h = m.hysteresis()
    assert np.array_equal(h[0], ow) and np.array_equal(h[2], go) and same_hyst(m, o)
    assert np.any(h[1] != 0.0) and np.any(h[3] != 0.0)
    assert np.array_equal(m.iq(), o.iq())
    # a time step later: turning points only fall
    for q in (m, o):
        q.begin_time_step(86400.0)
    h2 = m.hysteresis()
    assert np.all(h2[0] <= ow) and np.all(h2[2] <= go) and same_hyst(m, o)


def test_argument_errors(pkg):
    plain = pkg.decks.cartesian_case(4, 4, 3, state="mixed")          # base record: no hysteresis
    m = pkg.capi.HipModel(plain)
    with pytest.raises(pkg.capi.OpmHipError) as e:
        m.set_hysteresis(0, np.zeros(plain["Nb"], np.int32))
    assert e.value.code == pkg.capi.INVALID_ARGUMENT
    case = helpers.hysteresis_case(pkg, 4, 4, 3)
    m = pkg.capi.HipModel(case)
    with pytest.raises(pkg.capi.OpmHipError):      # Killough's models are refused, as in the reference
        m.set_hysteresis(2, case["imbnum"])
    with pytest.raises(pkg.capi.OpmHipError):      # region out of range
        m.set_hysteresis(0, np.full(case["Nb"], 5, np.int32))
    with pytest.raises(pkg.capi.OpmHipError):      # imbibition end points without end-point scaling in force
        m.set_hysteresis(0, case["imbnum"], dict(swl=np.full(case["Nb"], 0.1)))
    with pytest.raises(pkg.capi.OpmHipError):      # not in force
        m.hysteresis()
    m.set_hysteresis(0, case["imbnum"])
    m.set_hysteresis(None)
