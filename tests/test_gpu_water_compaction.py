"""Water-induced rock compaction on the device (opmhip_set_water_compaction: ROCKCOMP with ROCK2D / ROCK2DTR / ROCKWNOD) against
the CPU oracle, bit for bit: the 2-D multiplier tables over (effective oil pressure, SwMax - Sw_initial) in the porosity and
in the transmissibility multiplier of the upstream cell (ebos/eclproblem.hh:1962-1967, 2001-2005), the tracker
maxWaterSaturation_ (:2144-2169, :2289-2290) driven by opmhip_begin_time_step - with the reference's statement :2150 that
hands cell 1 the stored maximum of cell 0 - together with overburden pressure and rock regions."""
import numpy as np
import pytest

import helpers
import oracle_bind

pytestmark = pytest.mark.gpu

TABLES = helpers.ROCK2D_2


def make(pkg, orc, with_trans=True, n=(7, 6, 8)):
    case = helpers.wetgas_case(pkg, *n, heterogeneous=True)
    case["rocknum"] = (np.arange(case["Nb"]) % 2).astype(np.int32)
    case["overburden"] = 20e5 + 50.0 * (case["depth"] - case["depth"].min())
    m = pkg.capi.HipModel(case, reorder="line_coloring")
    o = oracle_bind.OracleModel(orc, case)
    for q in (m, o):
        q.set_state(case["pv"], case["meaning"])
    tabs = TABLES if with_trans else [{k: v for k, v in t.items() if k != "trans_mult"} for t in TABLES]
    return case, m, o, tabs


def same_state(m, o):
    pm, mm = m.get_state()
    po, mo = o.get_state()
    return np.array_equal(mm, mo) and np.array_equal(pm, po)


@pytest.mark.parametrize("with_trans", [True, False])
def test_water_compaction_through_time_steps_bitwise(pkg, orc, with_trans):
    case, m, o, tabs = make(pkg, orc, with_trans)
    src = pkg.decks.five_spot_source(case, rate_sm3_per_day=40.0)
    plain = m.iq().copy()
    for q in (m, o):
        q.set_source(src)
        q.set_water_compaction(tabs)
    sw0 = plain[:, 0, 0]
    assert np.array_equal(m.max_water_saturation(), o.max_water_saturation()) and np.array_equal(m.max_water_saturation(), sw0)
    a = m.iq()
    assert np.array_equal(a, o.iq())
    nf = a.shape[1]
    PORO, TMULT = nf - 1, nf - 2     # extended record: ..., Rv, tmult, poro
    assert not np.array_equal(a[:, PORO, 0], plain[:, PORO, 0])              # the multiplier at (p_eff, 0) is in
    assert np.array_equal(a[:, TMULT, 0] != 1.0, np.full(case["Nb"], with_trans)) or not with_trans
    if not with_trans:
        assert np.all(a[:, TMULT, 0] == 1.0)
    # at SwMax = Sw (tracker == current value) max(Sw, tracker) picks the constant: no Sw derivative yet (Opm::max)
    assert np.all(a[:, PORO, 1] == 0.0)
    grew = 0
    for step, dt in enumerate([0.5 * 86400.0, 86400.0, 2 * 86400.0]):
        for q in (m, o):
            q.begin_time_step(dt)
        assert np.array_equal(m.max_water_saturation(), o.max_water_saturation())
        assert np.array_equal(m.iq(), o.iq())
        for it in range(3):
            jm, rm = m.assemble(dt, it)
            jo, ro = o.assemble(dt, it)
            assert np.array_equal(jm, jo) and np.array_equal(rm, ro), (step, it)
            x, res = o.solve(tol=1e-6, maxit=200, w=0.9, mode="post_scale", reorder="none")
            m.update(x, 1.0)
            o.update(x)
            assert same_state(m, o)
            assert np.array_equal(m.iq(), o.iq())
        q_iq = m.iq()
        grew += int(np.any(q_iq[:, PORO, 1] != 0.0))        # cells whose Sw passed the stored maximum carry the Sw derivative
        for q in (m, o):
            q.end_time_step(dt)
    assert grew > 0
    t = m.max_water_saturation()
    assert np.any(t > sw0)                                    # water was injected: the maximum moved
    # off again: the plain intensive quantities of the present state on both sides
    for q in (m, o):
        q.set_water_compaction(None)
    assert np.array_equal(m.iq(), o.iq()) and np.all(m.max_water_saturation() == 0.0)


def test_the_tracker_keeps_the_reference_statement_for_cell_one(pkg, orc):
    """eclproblem.hh:2150: before the loop over the cells, entry 1 of the per-cell vector is overwritten with entry 0"""
    case, m, o, tabs = make(pkg, orc, n=(5, 4, 4))
    pv = case["pv"].reshape(-1, 3).copy()
    pv[0, 0] += 0.2                              # cell 0 starts with the larger water saturation
    for q in (m, o):
        q.set_state(pv.reshape(-1), case["meaning"])
        q.set_water_compaction(tabs)
    t0 = m.max_water_saturation()
    assert t0[0] > t0[1]
    for q in (m, o):
        q.begin_time_step(86400.0)
    t1 = m.max_water_saturation()
    assert np.array_equal(t1, o.max_water_saturation())
    assert t1[1] == t0[0] and np.array_equal(t1[2:], t0[2:]) and t1[0] == t0[0]
    assert np.array_equal(m.iq(), o.iq())
    # in every ordering of the unknowns the natural cells 0 and 1 are meant
    for reorder in ("graph_coloring", "level_scheduling", "graph_coloring_greedy"):
        m2 = pkg.capi.HipModel(case, reorder=reorder)
        m2.set_state(pv.reshape(-1), case["meaning"])
        m2.set_water_compaction(tabs)
        m2.begin_time_step(86400.0)
        assert np.array_equal(m2.max_water_saturation(), t1), reorder
        assert np.array_equal(m2.iq(), m.iq()), reorder


def test_argument_errors(pkg):
    case = pkg.decks.cartesian_case(4, 4, 3, state="mixed")
    m = pkg.capi.HipModel(case)
    with pytest.raises(pkg.capi.OpmHipError):      # the tracker starts from the initial solution
        m.set_water_compaction(TABLES)
    m.set_state(case["pv"], case["meaning"])
    with pytest.raises(pkg.capi.OpmHipError):      # base record: no transmissibility-multiplier field
        m.set_water_compaction(TABLES)
    wet = helpers.wetgas_case(pkg, 4, 4, 4, rocktab=helpers.ROCKTAB_2)
    m = pkg.capi.HipModel(wet)
    m.set_state(wet["pv"], wet["meaning"])
    with pytest.raises(pkg.capi.OpmHipError):      # ROCKTAB and ROCK2D exclude each other
        m.set_water_compaction(TABLES)
    wet = helpers.wetgas_case(pkg, 4, 4, 4)
    wet["rocknum"] = (np.arange(wet["Nb"]) % 3).astype(np.int32)
    m = pkg.capi.HipModel(wet)
    m.set_state(wet["pv"], wet["meaning"])
    with pytest.raises(pkg.capi.OpmHipError):      # a cell points at table 2, two tables given
        m.set_water_compaction(TABLES)
    bad = [dict(TABLES[0], sw=[0.0, 0.3, 0.1, 0.6])]
    wet["rocknum"] = np.zeros(wet["Nb"], np.int32)
    m = pkg.capi.HipModel(wet)
    m.set_state(wet["pv"], wet["meaning"])
    with pytest.raises(pkg.capi.OpmHipError):
        m.set_water_compaction(bad)
    m.set_water_compaction(TABLES[:1])
    assert np.array_equal(m.max_water_saturation(), m.iq()[:, 0, 0])
