"""Relative-permeability hysteresis in the oracle (oracle/fluid.hpp HystCell / SatFunc::relativePermeabilitiesHyst, restated from
opm-material's EclHysteresisTwoPhaseLaw - absent from the reference tree, UNVERIFIED; the tree only fixes what is supported:
Carlson, EHYSTR item 2 in {0, 1}, opm/simulators/utils/PartiallySupportedFlowKeywords.cpp:299-302 - and when the state is
updated: EclProblem::beginTimeStep, ebos/eclproblem.hh:1060, 2603-2626).  Properties the model must have, whatever its details:
no effect before the first time step; drainage curve while a system's wetting saturation keeps falling; a scanning curve that
leaves the drainage curve AT the turning point and traps the non-wetting phase; the turning points are extrema over the time
steps; AD Jacobian = finite differences on scanning curves.  CPU only."""
import numpy as np
import pytest

import helpers
import oracle_bind

KRW, KRO, KRG = 9, 10, 11     # mobility fields of the 19-field record; F_S = 0..2, viscosities divide the relative permeabilities


def model(pkg, orc, shape=(4, 3, 3), **kw):
    case = helpers.hysteresis_case(pkg, *shape, **kw)
    m = oracle_bind.OracleModel(orc, case)
    m.set_state(case["pv"], case["meaning"])
    return case, m


def uniform_state(case, sw, sg, p=250e5):
    pv = np.empty((case["Nb"], 3))
    pv[:, 0], pv[:, 1], pv[:, 2] = sw, p, sg
    return pv.reshape(-1), np.zeros(case["Nb"], np.uint8)


@pytest.mark.parametrize("kr_model", [0, 1])
def test_no_effect_before_the_first_time_step(pkg, orc, kr_model):
    case, m = model(pkg, orc)
    plain = m.iq().copy()
    m.set_hysteresis(kr_model, case["imbnum"])
    h = m.hysteresis()
    assert all(np.all(a == v) for a, v in zip(h, (2.0, 0.0, 2.0, 0.0)))
    if kr_model == 0:
        assert np.array_equal(m.iq(), plain)          # every curve is the drainage curve: the same bits
    else:                                             # wetting phases on their imbibition curves from the start: water and oil differ
        q = m.iq()
        assert np.array_equal(q[:, KRG], plain[:, KRG]) and not np.array_equal(q[:, KRW], plain[:, KRW])
    m.set_hysteresis(None)
    assert np.array_equal(m.iq(), plain)


def test_turning_points_are_extrema_over_the_time_steps(pkg, orc):
    case, m = model(pkg, orc)
    m.set_hysteresis(0, case["imbnum"])
    seen_ow, seen_go = np.full(case["Nb"], 2.0), np.full(case["Nb"], 2.0)
    rng = np.random.default_rng(3)
    for step in range(6):
        sw = rng.uniform(0.15, 0.5, case["Nb"])
        sg = rng.uniform(0.0, 0.4, case["Nb"])
        pv = np.stack([sw, np.full(case["Nb"], 250e5), sg], axis=1).reshape(-1)
        m.set_state(pv, np.zeros(case["Nb"], np.uint8))
        m.begin_time_step(86400.0)
        so = 1.0 - sw - sg
        seen_ow = np.minimum(seen_ow, 1.0 - so)
        seen_go = np.minimum(seen_go, 1.0 - sg)
        h = m.hysteresis()
        assert np.array_equal(h[0], seen_ow) and np.array_equal(h[2], seen_go)
    # restart: the same turning points handed in give the same shifts
    h = [a.copy() for a in m.hysteresis()]
    m.set_hysteresis(0, case["imbnum"])
    m.set_hysteresis_params(h[0], h[2])
    assert all(np.array_equal(a, b) for a, b in zip(m.hysteresis(), h))


def test_gas_drainage_then_imbibition_traps_gas(pkg, orc):
    """gas invades (Sg 0 -> 0.4: drainage), then retreats: krg follows the drainage curve on the way in, a scanning curve that
    starts ON the drainage curve at the turning point on the way out, and vanishes while the drainage curve still flows"""
    case, m = model(pkg, orc, shape=(2, 2, 2))
    m.set_hysteresis(0, case["imbnum"])
    ref = oracle_bind.OracleModel(orc, case)            # the same case without hysteresis: the drainage curves
    sw = 0.2
    def krg_pair(sg):
        for q in (m, ref):
            q.set_state(*uniform_state(case, sw, sg))
        return m.iq()[0, KRG, 0], ref.iq()[0, KRG, 0]
    for sg in (0.1, 0.2, 0.3, 0.4):                      # drainage: every step begins at a new extremum
        m.set_state(*uniform_state(case, sw, sg))
        m.begin_time_step(86400.0)
        a, b = krg_pair(sg)
        assert a == b
        a, b = krg_pair(sg + 0.03)                       # further drainage inside the step: still the drainage curve
        assert a == b
    turn = 1.0 - 0.4
    assert np.all(m.hysteresis()[2] == turn)
    # (the gas-oil system is UPDATED with 1 - Sg but EVALUATED at 1 - Swco - Sg - EclDefaultMaterial's own inconsistency, kept: the
    #  scanning curve begins once the gas has retreated by Swco from the turning point, and it begins ON the drainage curve)
    swco = 0.12
    a, b = krg_pair(0.4 - swco)
    assert a == b
    a, b = krg_pair(0.4 - swco - 1e-9)
    assert abs(a - b) < 1e-6 * b                         # continuous where the scanning curve takes over (mobilities: kr / viscosity)
    last = krg_pair(0.39)[1]
    trapped = None
    for sg in np.linspace(0.39, 0.0, 40):                # imbibition inside ONE time step and over several: same curve
        a, b = krg_pair(sg)
        assert 0.0 <= a <= last + 1e-15 and a <= b + 1e-15   # monotone, and below the drainage curve
        last = a
        if trapped is None and a == 0.0:
            trapped = sg
    assert trapped is not None and trapped > 0.025       # the drainage curve flows down to Sg = 0.02
    _, b = krg_pair(trapped)
    assert b > 0.0
    m.begin_time_step(86400.0)                           # a time step that begins during imbibition moves no turning point
    assert np.all(m.hysteresis()[2] == turn)
    # second drainage: back up the SAME scanning curve to the turning point, beyond it the drainage curve again
    m.set_state(*uniform_state(case, sw, 0.45))
    m.begin_time_step(86400.0)
    assert np.all(m.hysteresis()[2] == 1.0 - 0.45)
    a, b = krg_pair(0.45)
    assert a == b


@pytest.mark.parametrize("kr_model", [0, 1])
def test_oil_is_trapped_behind_a_water_flood(pkg, orc, kr_model):
    """oil-water system: oil is the non-wetting phase.  Water retreats to Sw = 0.2 (drainage), then floods: kro on the scanning
    curve reaches zero at a lower water saturation than the drainage curve does; model 1 puts water on its imbibition curve"""
    case, m = model(pkg, orc, shape=(2, 2, 2))
    m.set_hysteresis(kr_model, case["imbnum"])
    ref = oracle_bind.OracleModel(orc, case)
    m.set_state(*uniform_state(case, 0.2, 0.0))
    m.begin_time_step(86400.0)
    gone_h = gone_d = None
    for sw in np.linspace(0.2, 0.95, 76):
        for q in (m, ref):
            q.set_state(*uniform_state(case, sw, 0.0))
        a, b = m.iq()[0, KRO, 0], ref.iq()[0, KRO, 0]
        w, wd = m.iq()[0, KRW, 0], ref.iq()[0, KRW, 0]
        assert (w == wd) if kr_model == 0 else (w <= wd)
        if gone_h is None and a == 0.0: gone_h = sw
        if gone_d is None and b == 0.0: gone_d = sw
    assert gone_h is not None and gone_d is not None and gone_h < gone_d


@pytest.mark.parametrize("kr_model,wetgas", [(0, False), (1, False), (1, True)])
def test_jacobian_on_scanning_curves_matches_finite_differences(pkg, orc, kr_model, wetgas):
    case, m = model(pkg, orc, shape=(4, 3, 3), heterogeneous=True, wetgas=wetgas)
    m.set_hysteresis(kr_model, case["imbnum"])
    dt = 86400.0
    rng = np.random.default_rng(1)
    pv0 = case["pv"].reshape(-1, 3).copy()
    sg = case["meaning"] == 0
    # a first step at high gas / low water saturations sets the turning points, the state then moves back: scanning curves
    pv = pv0.copy()
    pv[sg, 2] = np.minimum(pv[sg, 2] + 0.15, 0.6)
    pv[:, 0] = np.maximum(pv[:, 0] - 0.05, 0.13)
    m.set_state(pv.reshape(-1), case["meaning"])
    m.begin_time_step(dt)
    m.assemble(dt, 0)
    pv = pv0.copy()
    pv[:, 0] += rng.uniform(0.0, 0.02, len(pv))
    pv[:, 1] *= 1 + rng.uniform(-0.003, 0.003, len(pv))
    m.set_state(pv.reshape(-1), case["meaning"])
    h = m.hysteresis()
    sw_ow = 1.0 - m.iq()[:, 1, 0]      # 1 - So
    assert np.mean(sw_ow > h[0]) > 0.5          # most cells are on their oil-water scanning curve
    jac, r0 = m.assemble(dt, 1)
    Nb, rp, ci = case["Nb"], case["rowptr"], case["col"]
    J = np.zeros((Nb * 3, Nb * 3))
    for i in range(Nb):
        for k in range(rp[i], rp[i + 1]):
            J[3 * i:3 * i + 3, 3 * ci[k]:3 * ci[k] + 3] = jac[9 * k:9 * k + 9].reshape(3, 3)
    scale = np.array([1e-7, 1.0, 1e-7])
    n = Nb * 3
    Jfd = np.zeros((n, n))
    for c in range(n):
        hh = scale[c % 3] * (1.0 if (c % 3 != 2 or case["meaning"][c // 3] == 0) else (1e2 if case["meaning"][c // 3] == 1 else 1e-4))
        xp = pv.reshape(-1).copy(); xp[c] += hh
        xm = pv.reshape(-1).copy(); xm[c] -= hh
        m.set_state(xp, case["meaning"]); _, rp_ = m.assemble(dt, 1)
        m.set_state(xm, case["meaning"]); _, rm_ = m.assemble(dt, 1)
        Jfd[:, c] = (rp_ - rm_) / (2 * hh)
    cs = np.maximum(np.abs(J).max(axis=0), 1e-300)
    err = np.abs(J - Jfd) / cs[None, :]
    assert err.max() < 5e-5, (err.max(), np.unravel_index(err.argmax(), err.shape))
