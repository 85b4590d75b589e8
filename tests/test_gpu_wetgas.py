"""GPU parity of the extended fluid model - wet gas (PVTG: Rv, vaporised oil in storage / flux / gas density, the third
primary-variable meaning Sw_pg_Rv, DRVDT cap) and rock compaction tables (ROCKTAB: pore-volume and transmissibility
multipliers, overburden pressure) - through the C-ABI against the CPU oracle, bit for bit.  The device uses its
19-field intensive-quantity record for these fluids (opmhip_iq_fields); every other deck keeps the 17-field one."""
import numpy as np
import pytest

import oracle_bind
import helpers
from helpers import ROCKTAB_2, rv_sat, wetgas_case

pytestmark = pytest.mark.gpu
REORDERS = ["level_scheduling", "graph_coloring", "line_coloring"]


def both(pkg, orc, case, reorder="graph_coloring_greedy"):
    m = pkg.capi.HipModel(case, reorder=reorder)
    o = oracle_bind.OracleModel(orc, case)
    m.set_state(case["pv"], case["meaning"])
    o.set_state(case["pv"], case["meaning"])
    return m, o


@pytest.mark.parametrize("rocktab", [None, ROCKTAB_2])
def test_intensive_quantities_bitwise(pkg, orc, rocktab):
    case = wetgas_case(pkg, 7, 6, 9, rocktab=rocktab, heterogeneous=True)
    assert set(case["meaning"]) == {0, 1, 2}
    m, o = both(pkg, orc, case)
    a, b = m.iq(), o.iq()
    assert a.shape[1] == 19 and np.array_equal(a, b)
    assert np.all(a[:, 17, 0] == 1.0) if rocktab is None else a[:, 17, 0].std() > 0


@pytest.mark.parametrize("reorder", REORDERS)
@pytest.mark.parametrize("rocktab", [None, ROCKTAB_2])
def test_jacobian_and_residual_bitwise(pkg, orc, reorder, rocktab):
    case = wetgas_case(pkg, 9, 7, 9, rocktab=rocktab, heterogeneous=True)
    src = pkg.decks.five_spot_source(case, rate_sm3_per_day=60.0)
    m, o = both(pkg, orc, case, reorder=reorder)
    for h in (m, o):
        h.set_source(src)
    dt = 86400.0
    j0, r0 = m.assemble(dt, 0)
    jo, ro = o.assemble(dt, 0)
    assert np.array_equal(r0, ro) and np.array_equal(j0, jo)
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
    j1, r1 = m.assemble(dt, 1)
    jo1, ro1 = o.assemble(dt, 1)
    assert np.array_equal(r1, ro1) and np.array_equal(j1, jo1)
    assert np.array_equal(m.convergence(dt)[3:6], o.convergence(dt)[3:6])


def test_switching_between_all_three_meanings(pkg, orc):
    case = wetgas_case(pkg, 5, 4, 8, perturb=False)
    m, o = both(pkg, orc, case)
    mean = case["meaning"]
    pv = case["pv"].reshape(-1, 3)
    three = np.flatnonzero(mean == 0)
    dx = np.zeros_like(pv)
    dx[three, 2] = -0.2           # Sg grows by the chop limit per update until the oil saturation turns negative
    for step in range(4):
        assert m.update(dx.reshape(-1)) == o.update(dx.reshape(-1))
        pm, mm = m.get_state()
        po, mo = o.get_state()
        assert np.array_equal(mm, mo) and np.array_equal(pm, po) and np.array_equal(m.iq(), o.iq())
    assert np.all(mm[three] == 2)                                              # (Sw, po, Sg) -> (Sw, pg, Rv)
    p1 = pm.reshape(-1, 3)
    dx[:] = 0.0
    dx[three, 2] = -2.0 * p1[three, 2]                                         # Rv far above saturation: oil re-appears
    und = np.flatnonzero(mean == 1)
    dx[und, 2] = -50.0                                                         # Rs above RsSat: gas appears
    dx[mean == 2, 0] = -0.9                                                    # water floods the gas cap: Sw >= 1
    for step in range(5):
        assert m.update(dx.reshape(-1)) == o.update(dx.reshape(-1))
        pm, mm = m.get_state()
        po, mo = o.get_state()
        assert np.array_equal(mm, mo) and np.array_equal(pm, po) and np.array_equal(m.iq(), o.iq())
        dx[three, 2] = 0.0
    assert np.all(mm[three] == 0) and np.all(mm[und] == 0) and np.all(mm[mean == 2] == 0)
    j, r = m.assemble(86400.0, 0)
    jo, ro = o.assemble(86400.0, 0)
    assert np.array_equal(r, ro) and np.array_equal(j, jo)


def test_drvdt_cap_and_state_validation(pkg, orc):
    case = wetgas_case(pkg, 5, 5, 6, heterogeneous=True)
    pv = case["pv"].reshape(-1, 3)
    case["rvmax"] = np.full(case["Nb"], 0.75 * np.median(rv_sat(case["fluid"], pv[:, 1])))   # caps the saturated cells and part of the gas cap
    m, o = both(pkg, orc, case)
    a, b = m.iq(), o.iq()
    assert np.array_equal(a, b) and np.isclose(a[:, 16, 0].max(), case["rvmax"][0]) and (a[:, 16, 0] < case["rvmax"][0]).any()
    j, r = m.assemble(3 * 86400.0, 0)
    jo, ro = o.assemble(3 * 86400.0, 0)
    assert np.array_equal(r, ro) and np.array_equal(j, jo)
    # a dry-gas context refuses the third meaning and the extras
    dry = pkg.decks.cartesian_case(4, 4, 3, state="mixed")
    md = pkg.capi.HipModel(dry)
    bad = dry["meaning"].copy()
    bad[0] = 2
    with pytest.raises(pkg.capi.OpmHipError) as e:
        md.set_state(dry["pv"], bad)
    assert e.value.code == pkg.capi.INVALID_ARGUMENT
    with pytest.raises(pkg.capi.OpmHipError):
        md.set_problem_extras(rvmax=np.ones(dry["Nb"]))
    md.set_state(dry["pv"], dry["meaning"])
    assert md.iq().shape[1] == 17


def test_newton_steps_with_wet_gas_match_oracle(pkg, orc):
    """whole Newton iterations (assemble, ILU0-BiCGStab in the device's ordering, update) on both sides"""
    case = wetgas_case(pkg, 8, 8, 9, rocktab=ROCKTAB_2, heterogeneous=True)
    src = pkg.decks.five_spot_source(case, rate_sm3_per_day=80.0)
    m, o = both(pkg, orc, case, reorder="line_coloring")
    for h in (m, o):
        h.set_source(src)
    dt = 3 * 86400.0
    for it in range(4):
        m.assemble(dt, it, fetch=False)
        o.assemble(dt, it)
        res = m.solve_jacobian_system()
        xo, reso = o.solve_in_order(*m.ordering()[:2], tol=1e-2, maxit=200, w=0.9)
        assert res.converged and res.it == reso.it
        m.update(None, 1.0)
        o.update(xo)
        pm, mm = m.get_state()
        po, mo = o.get_state()
        assert np.array_equal(mm, mo)
        np.testing.assert_allclose(pm, po, rtol=1e-7, atol=1e-12)


def test_second_set_fluid_is_refused_once_the_static_data_is_set(pkg, orc):
    """set_static sizes the intensive-quantity cache for the fluid's record layout (17 or 19 fields) and checks the region
    arrays against its table counts: a PVTG fluid handed in afterwards would make the 19-field kernels write past the
    17-field buffers.  The call is refused and the context keeps working with the fluid it has."""
    import ctypes as C
    case = pkg.decks.cartesian_case(6, 5, 4, state="mixed", heterogeneous=True)
    m = pkg.capi.HipModel(case, reorder="line_coloring")
    m.set_state(case["pv"], case["meaning"])
    j0, r0 = m.assemble(86400.0, 0)
    wet_fluid = helpers.wetgas_fluid(pkg)   # owns the arrays the descriptor points at
    wet = wet_fluid.desc()
    L = pkg.capi.lib()
    assert L.opmhip_set_fluid(m._h, C.addressof(wet)) == pkg.capi.INVALID_ARGUMENT
    assert b"set_static" in L.opmhip_last_error(m._h)
    assert L.opmhip_iq_fields(m._h) == 17
    j1, r1 = m.assemble(86400.0, 0)
    assert np.array_equal(j0, j1) and np.array_equal(r0, r1)
    # before set_static the fluid may still be replaced (the old table blobs are given back)
    f = pkg.capi.HipFluid(pkg.fluid.spe1_fluid()[0])
    assert L.opmhip_set_fluid(f._h, C.addressof(wet)) == pkg.capi.SUCCESS
    p = np.array([150e5, 200e5])
    assert np.array_equal(f.probe_gas(p, 0.0), pkg.capi.HipFluid(helpers.wetgas_fluid(pkg)).probe_gas(p, 0.0))
