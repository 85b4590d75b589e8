"""GPU parity of the assembly half of the hot path, through the C-ABI, against the CPU oracle on the same seeded
inputs: intensive quantities, Jacobian + residual (bit for bit), convergence norms, Newton update / switching, and
whole Newton iterations.  Tolerances: 0 (bitwise) for per-cell and per-face arithmetic; 1e-12 relative for global
reductions whose summation order differs (convergence sums, Krylov dot products)."""
import numpy as np
import pytest

import oracle_bind

pytestmark = pytest.mark.gpu
REORDERS = ["level_scheduling", "graph_coloring", "graph_coloring_greedy", "line_coloring"]


def both(pkg, orc, case, reorder="graph_coloring_greedy", **kw):
    m = pkg.capi.HipModel(case, reorder=reorder, **kw)
    o = oracle_bind.OracleModel(orc, case)
    m.set_state(case["pv"], case["meaning"])
    o.set_state(case["pv"], case["meaning"])
    return m, o


@pytest.mark.parametrize("state", ["undersaturated", "saturated", "mixed"])
def test_intensive_quantities_bitwise(pkg, orc, state):
    case = pkg.decks.cartesian_case(9, 7, 5, state=state, heterogeneous=True)
    m, o = both(pkg, orc, case)
    assert np.array_equal(m.iq(), o.iq())


@pytest.mark.parametrize("reorder", REORDERS)
@pytest.mark.parametrize("state", ["undersaturated", "saturated", "mixed"])
def test_jacobian_and_residual_bitwise(pkg, orc, reorder, state):
    case = pkg.decks.cartesian_case(11, 9, 6, state=state, heterogeneous=True)
    m, o = both(pkg, orc, case, reorder=reorder)
    dt = 86400.0
    j0, r0 = m.assemble(dt, 0)
    jo, ro = o.assemble(dt, 0)
    assert np.array_equal(r0, ro) and np.array_equal(j0, jo)
    # second iteration from a moved state: storage term active, cache from iteration 0
    rng = np.random.default_rng(3)
    dx = np.zeros((case["Nb"], 3))
    dx[:, 0] = rng.uniform(-0.01, 0.01, case["Nb"])
    dx[:, 1] = rng.uniform(-2e5, 2e5, case["Nb"])
    dx[:, 2] = np.where(case["meaning"] == 0, rng.uniform(-0.01, 0.01, case["Nb"]), rng.uniform(-1.0, 1.0, case["Nb"]))
    assert m.update(dx.reshape(-1)) == o.update(dx.reshape(-1))
    pm, mm = m.get_state()
    po, mo = o.get_state()
    assert np.array_equal(pm, po) and np.array_equal(mm, mo)
    j1, r1 = m.assemble(dt, 1)
    jo1, ro1 = o.assemble(dt, 1)
    assert np.array_equal(r1, ro1) and np.array_equal(j1, jo1)
    assert np.abs(r1).max() > 0 and not np.array_equal(j1, j0)


@pytest.mark.parametrize("reorder", REORDERS)
@pytest.mark.parametrize("state", ["undersaturated", "saturated", "mixed"])
def test_degenerate_upwind_ties_bitwise(pkg, orc, reorder, state):
    """Unperturbed, homogeneous state: every horizontal face has pressure difference exactly 0 and equal volumes, i.e.
    the tie-break branch of ebos/eclfluxmodule.hh:287-321 (lower GLOBAL index is upstream) is taken on 2/3 of the faces
    at Newton iteration 0.  Note that with dp == 0 the threshold test `|dp| > thpres` (:327-337) fails for every
    thpres >= 0, so the face contributes neither flux nor derivative whichever side is called upstream: the choice is
    not observable in J or r; this test pins exactly that, in every ordering."""
    case = pkg.decks.cartesian_case(8, 7, 6, state=state, heterogeneous=False, perturb=False)
    # the state really is degenerate: equal pressures and saturations inside every layer
    pv = case["pv"].reshape(6, 56, 3)
    assert np.all(pv == pv[:, :1, :]) and np.all(case["volume"] == case["volume"][0])
    m, o = both(pkg, orc, case, reorder=reorder)
    for dt, it in ((86400.0, 0), (10 * 86400.0, 1)):
        j, r = m.assemble(dt, it)
        jo, ro = o.assemble(dt, it)
        assert np.array_equal(r, ro) and np.array_equal(j, jo)
    # horizontal couplings carry no derivative at all in this state (zero blocks), vertical ones do
    fd = np.abs(case["face_dir"])
    jb = j.reshape(-1, 9)
    assert not jb[(fd == 1) | (fd == 2)].any() and jb[fd == 3].any()
    # with threshold pressures above the vertical pressure steps every face is closed
    case2 = dict(case)
    case2["thpres"] = np.where(fd > 0, 1e7, 0.0)
    m2, o2 = both(pkg, orc, case2, reorder=reorder)
    j2, r2 = m2.assemble(86400.0, 0)
    jo2, ro2 = o2.assemble(86400.0, 0)
    assert np.array_equal(r2, ro2) and np.array_equal(j2, jo2) and not j2.reshape(-1, 9)[fd > 0].any()


def test_spe1_grid_bitwise(pkg, orc):
    """BASELINE config 0: the SPE1 grid and fluid (300 cells); 'bit-for-bit residual' against the CPU restatement."""
    case = pkg.decks.spe1_case()
    m, o = both(pkg, orc, case, reorder="level_scheduling")
    j, r = m.assemble(86400.0, 0)
    jo, ro = o.assemble(86400.0, 0)
    assert np.array_equal(r, ro) and np.array_equal(j, jo)


def test_source_terms(pkg, orc):
    case = pkg.decks.cartesian_case(8, 8, 3, state="undersaturated")
    src = pkg.decks.five_spot_source(case)
    dsrc = np.zeros((case["Nb"], 9))
    dsrc[0, 4] = -1e-12  # a well-like rate derivative w.r.t. pressure on one cell
    m, o = both(pkg, orc, case)
    m.set_source(src, dsrc.reshape(-1))
    o.set_source(src, dsrc.reshape(-1))
    j, r = m.assemble(3600.0, 0)
    jo, ro = o.assemble(3600.0, 0)
    assert np.array_equal(r, ro) and np.array_equal(j, jo)
    assert r.reshape(-1, 3)[0, 1] < 0  # injected water shows up with the sign "residual -= source"


@pytest.mark.parametrize("state", ["mixed"])
def test_convergence_norms(pkg, orc, state):
    case = pkg.decks.cartesian_case(12, 10, 7, state=state, heterogeneous=True)
    m, o = both(pkg, orc, case)
    dt = 86400.0
    m.assemble(dt, 0, fetch=False)
    _, r = o.assemble(dt, 0)
    cm, co = m.convergence(dt, 1e-2), o.convergence(dt, 1e-2)
    # maxima are exact; sums differ only in summation order: bound the difference by 1e-13 of the sum of magnitudes
    # (at iteration 0 the residual sums cancel to rounding noise - mass conservation - so a relative test on the
    # sum itself would compare noise with noise)
    mag = np.abs(r.reshape(-1, 3)).sum(axis=0)
    assert np.array_equal(cm[3:6], co[3:6])
    np.testing.assert_allclose(cm[6:10], co[6:10], rtol=1e-13)
    assert np.all(np.abs(cm[0:3] - co[0:3]) <= 1e-13 * mag)
    np.testing.assert_allclose(cm[10], co[10], rtol=1e-12)
    np.testing.assert_allclose(cm[11:14], co[11:14], rtol=1e-13)
    assert np.all(np.abs(cm[14:17] - co[14:17]) <= 1e-13 * co[6:9] * mag * dt / co[9])
    # a state with a real imbalance: sums are O(1) of the magnitudes and must agree to 1e-10
    src = pkg.decks.five_spot_source(case, rate_sm3_per_day=500.0)
    m.set_source(src)
    o.set_source(src)
    m.assemble(dt, 0, fetch=False)
    o.assemble(dt, 0)
    cm, co = m.convergence(dt, 1e-2), o.convergence(dt, 1e-2)
    np.testing.assert_allclose(cm[[0, 1, 2, 14, 15, 16]], co[[0, 1, 2, 14, 15, 16]], rtol=1e-10)
    assert np.array_equal(cm[3:6], co[3:6]) and cm[10] > 0


def test_update_switching_matches_oracle(pkg, orc):
    case = pkg.decks.cartesian_case(6, 5, 4, state="mixed", perturb=False)
    m, o = both(pkg, orc, case)
    Nb = case["Nb"]
    dx = np.zeros((Nb, 3))
    dx[:, 1] = 0.5 * case["pv"].reshape(-1, 3)[:, 1]
    dx[:, 0] = 0.5
    dx[case["meaning"] == 0, 2] = 0.3
    for step in range(3):
        nm, no = m.update(dx.reshape(-1)), o.update(dx.reshape(-1))
        pm, mm = m.get_state()
        po, mo = o.get_state()
        assert nm == no and np.array_equal(mm, mo) and np.array_equal(pm, po)
        assert np.array_equal(m.iq(), o.iq())
        dx[:, 0] = -0.05
        dx[:, 1] *= 0.1
        dx[:, 2] = np.where(mm == 0, 0.2, 0.0)
    assert (mm != case["meaning"]).any()


@pytest.mark.parametrize("reorder", ["level_scheduling", "graph_coloring_greedy", "line_coloring"])
def test_newton_iterations_match_oracle(pkg, orc, reorder):
    """A full time step driven like BlackoilModelEbos::nonlinearIteration on both sides: same Newton and linear
    iteration counts, final pressures/saturations within 1e-7 relative (FP tolerance stated by the task: results
    within the project's own ECL-compare band, compareECLFiles.cmake:198-199, is far looser)."""
    case = pkg.decks.cartesian_case(10, 10, 6, state="mixed", heterogeneous=True)
    src = pkg.decks.five_spot_source(case, rate_sm3_per_day=200.0)
    m, o = both(pkg, orc, case, reorder=reorder)
    m.set_source(src)
    o.set_source(src)
    dt = 5 * 86400.0
    lin_m, lin_o = [], []
    for it in range(12):
        m.assemble(dt, it, fetch=False)
        o.assemble(dt, it)
        cm, co = m.convergence(dt), o.convergence(dt)
        np.testing.assert_allclose(cm[11:14], co[11:14], rtol=1e-4, atol=1e-8)  # CNV tolerance is 1e-2; Krylov rounding is amplified near convergence
        conv = np.all(co[11:14] < 1e-2) and np.all(co[14:17] < 1e-6) and it > 0
        if conv:
            break
        rm = m.solve_jacobian_system()
        xo, ro = o.solve_in_order(*m.ordering()[:2])
        assert rm.converged and ro.converged
        lin_m.append(rm.it)
        lin_o.append(ro.it)
        m.update(None, 1.0)
        o.update(xo)
    assert it < 11, "Newton did not converge"
    assert lin_m == lin_o
    pm, mm = m.get_state()
    po, mo = o.get_state()
    assert np.array_equal(mm, mo)
    # stated FP tolerance on the converged state: 1e-7 relative (pressures) / 1e-9 absolute (saturations); the
    # reference project itself accepts rel 1e-5 / abs 2e-2 between runs (compareECLFiles.cmake:198-199)
    np.testing.assert_allclose(pm, po, rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("reorder", ["line_coloring", "graph_coloring"])
def test_zero_diagonal_fix_of_the_device_assembled_jacobian(pkg, orc, reorder):
    """bda/BdaBridge.cpp:125-161 on the device-resident path: a cell without pore volume and without open faces assembles an
    all-zero diagonal block; the factorisation kernel puts 1e-15 on its diagonal as it stages the row - in the factors and
    in the matrix the operator reads - and the solve is the oracle's (same fix, same ordering)"""
    case = pkg.decks.cartesian_case(6, 5, 4, state="mixed", heterogeneous=True)
    cut = 37
    rp, ci = np.asarray(case["rowptr"]), np.asarray(case["col"])
    case["poro"] = np.asarray(case["poro"], float).copy()
    case["poro"][cut] = 0.0
    tr = np.asarray(case["trans"], float).copy()
    tr[rp[cut]:rp[cut + 1]] = 0.0
    tr[ci == cut] = 0.0
    case["trans"] = tr
    src = pkg.decks.five_spot_source(case, rate_sm3_per_day=50.0)
    src.reshape(-1, 3)[cut] = 0.0
    m, o = both(pkg, orc, case, reorder=reorder)
    m.set_source(src)
    o.set_source(src)
    dt = 86400.0
    jm, rm = m.assemble(dt, 0)
    jo, ro = o.assemble(dt, 0)
    assert np.array_equal(jm, jo) and np.array_equal(rm, ro)
    kd = [k for k in range(rp[cut], rp[cut + 1]) if ci[k] == cut][0]
    assert np.all(jm.reshape(-1, 9)[kd] == 0.0) and np.all(rm.reshape(-1, 3)[cut] == 0.0)
    sm = m.solve_jacobian_system()
    xo, so = o.solve_in_order(*m.ordering()[:2])
    assert sm.converged and so.converged and sm.it == so.it
    x = m.get_result()
    assert np.all(np.isfinite(x)) and np.all(x.reshape(-1, 3)[cut] == 0.0)
    np.testing.assert_allclose(x, xo, rtol=1e-8, atol=1e-12 * np.abs(xo).max())
    e = np.zeros(3 * case["Nb"])
    e[3 * cut:3 * cut + 3] = 1.0
    assert np.array_equal(m.spmv(e).reshape(-1, 3)[cut], np.full(3, 1e-15))


def test_time_levels_roll_back_bitwise(pkg, orc):
    """advance_time_level / update_failed (FvBaseDiscretization::advanceTimeLevel / updateFailed): after a rolled-back
    time step the device holds exactly the state it had when the step started - primary variables, meanings,
    intensive quantities and therefore the linearisation, bit for bit."""
    case = pkg.decks.cartesian_case(10, 9, 6, state="mixed", heterogeneous=True)
    src = pkg.decks.five_spot_source(case, rate_sm3_per_day=300.0)
    m = pkg.capi.HipModel(case, reorder="line_coloring")
    m.set_state(case["pv"], case["meaning"])
    m.set_source(src)
    with pytest.raises(RuntimeError):
        m.update_failed()            # nothing to roll back to yet
    dt = 20 * 86400.0
    m.advance_time_level()
    j0, r0 = m.assemble(dt, 0)
    iq0 = m.iq()
    for it in range(3):              # a few Newton updates move the state and switch cells
        if it:
            m.assemble(dt, it, fetch=False)
        assert m.solve_jacobian_system().converged
        m.update(None, 1.0)
    pv1, mean1 = m.get_state()
    assert not np.array_equal(pv1, case["pv"])
    m.update_failed()
    pv2, mean2 = m.get_state()
    assert np.array_equal(pv2, case["pv"]) and np.array_equal(mean2, case["meaning"])
    assert np.array_equal(m.iq(), iq0)
    j2, r2 = m.assemble(dt / 3, 0)
    o = oracle_bind.OracleModel(orc, case)
    o.set_state(case["pv"], case["meaning"])
    o.set_source(src)
    jo, ro = o.assemble(dt / 3, 0)
    assert np.array_equal(j2, jo) and np.array_equal(r2, ro)
    j3, r3 = m.assemble(dt, 0)
    assert np.array_equal(j3, j0) and np.array_equal(r3, r0)


def test_adaptive_time_stepping_matches_oracle_history(pkg, orc):
    """The sub-step control over the device model takes the same decisions as over the CPU restatement (same accepted
    and chopped time steps, same Newton counts) on a case that needs a chop."""
    case = pkg.decks.cartesian_case(8, 8, 8, state="mixed", heterogeneous=False)
    src = pkg.decks.five_spot_source(case, rate_sm3_per_day=60.0)
    m = pkg.capi.HipModel(case, reorder="level_scheduling", tolerance=1e-2, maxit=200, ilu_relaxation=0.9)
    m.set_state(case["pv"], case["meaning"])
    m.set_source(src)
    o = oracle_bind.OracleModel(orc, case)
    o.set_state(case["pv"], case["meaning"])
    o.set_source(src)
    par = dict(initial_dt=30 * 86400.0, max_dt=60 * 86400.0)
    sm = pkg.newton.AdaptiveTimeStepping(pkg.newton.BlackoilModelHip(m), pkg.newton.TimeSteppingParameters(**par))
    so = pkg.newton.AdaptiveTimeStepping(pkg.newton.BlackoilModelHip(oracle_bind.OracleAsHipModel(o)), pkg.newton.TimeSteppingParameters(**par))
    for _ in range(40):
        sm.next_newton_iteration()
        so.next_newton_iteration()
    assert sm.history == so.history
    assert sm.timesteps_done >= 2 and sm.timesteps_failed >= 1
    # the chopped step is a diverging Newton sequence: rounding differences of the Krylov dot products (reduction order)
    # are amplified there, so the linear iteration totals agree closely but not exactly
    assert abs(sm.report.total_linear_iterations - so.report.total_linear_iterations) <= 0.05 * so.report.total_linear_iterations


def two_region_fluid(pkg):
    import copy
    fl, d = pkg.fluid.spe1_fluid()
    p0, s0 = fl.pvt[0], fl.sat[0]
    p1 = copy.deepcopy(p0)
    p1["density"] = [d["density"]["oil"] * 1.07, d["density"]["water"] * 1.02, d["density"]["gas"] * 0.9]
    p1["pvtw"] = [p0["pvtw"][0], p0["pvtw"][1] * 1.01, p0["pvtw"][2] * 1.3, p0["pvtw"][3] * 0.8, p0["pvtw"][4]]
    p1["pvdg"] = [[r[0], r[1] * 1.04, r[2] * 1.15] for r in p0["pvdg"]]
    for n in p1["pvto"]:
        n["bo"] = [b * 1.03 for b in n["bo"]]
        n["mu"] = [m * 1.25 for m in n["mu"]]
        n["rs"] = n["rs"] * 0.93
    s1 = dict(swof=[[r[0], 0.8 * r[1], 0.9 * r[2], 2.0 * r[3]] for r in s0["swof"]],
              sgof=[[r[0], 0.7 * r[1], 0.85 * r[2], 1.5 * r[3]] for r in s0["sgof"]])
    return pkg.fluid.Fluid([p0, p1], [s0, s1], rock_pref=fl.rock_pref, rock_cr=fl.rock_cr)


@pytest.mark.parametrize("reorder", ["graph_coloring", "line_coloring"])
def test_regions_threshold_pressures_rs_cap_and_source_derivatives(pkg, orc, reorder):
    """The inputs the plain cases leave at their defaults, all at once: two PVTNUM and two SATNUM regions, THPRES on a
    third of the faces (eclfluxmodule.hh:269-290), a DRSDT-style cap on Rs below RsSat in some cells
    (eclproblem.hh:1711-1732) and well-like source derivatives.  Bit for bit through assembly, update and switching."""
    case = pkg.decks.cartesian_case(9, 8, 7, state="mixed", heterogeneous=True, fluid=two_region_fluid(pkg))
    Nb, rp, ci = case["Nb"], case["rowptr"], case["col"]
    rng = np.random.default_rng(77)
    case["pvtnum"] = rng.integers(0, 2, Nb).astype(np.int32)
    case["satnum"] = ((np.arange(Nb) // 9) % 2).astype(np.int32)
    row = np.repeat(np.arange(Nb), np.diff(rp))
    lo, hi = np.minimum(row, ci), np.maximum(row, ci)
    _, inv = np.unique(lo.astype(np.int64) * Nb + hi, return_inverse=True)
    th_face = np.where(rng.random(inv.max() + 1) < 0.33, rng.uniform(0.1e5, 3e5, inv.max() + 1), 0.0)
    thpres = th_face[inv]
    thpres[row == ci] = 0.0
    case["thpres"] = np.ascontiguousarray(thpres)
    p = case["pv"].reshape(-1, 3)[:, 1]
    case["rsmax"] = np.where(rng.random(Nb) < 0.3, 0.6 * pkg.decks.rs_sat(case["fluid"], p), 1e30)
    src = 1e-6 * rng.standard_normal((Nb, 3)) * (rng.random((Nb, 1)) < 0.05)
    dsrc = 1e-13 * rng.standard_normal((Nb, 9)) * (rng.random((Nb, 1)) < 0.05)
    m, o = both(pkg, orc, case, reorder=reorder)
    for q in (m, o):
        q.set_source(np.ascontiguousarray(src.reshape(-1)), np.ascontiguousarray(dsrc.reshape(-1)))
    assert np.array_equal(m.iq(), o.iq())
    dt = 3 * 86400.0
    for it in range(3):
        jm, rm = m.assemble(dt, it)
        jo, ro = o.assemble(dt, it)
        assert np.array_equal(rm, ro) and np.array_equal(jm, jo)
        np.testing.assert_allclose(m.convergence(dt)[11:17], o.convergence(dt)[11:17], rtol=1e-9, atol=1e-14)
        # the same update on both sides (so that the states stay bit-identical): a scaled random direction
        dx = (rng.standard_normal((Nb, 3)) * np.array([0.05, 2e5, 0.05])).reshape(-1)
        nm = m.update(dx, 1.0)
        no = o.update(dx)
        assert nm == no
        pm, mm = m.get_state()
        po, mo = o.get_state()
        assert np.array_equal(mm, mo) and np.array_equal(pm, po)
        assert np.array_equal(m.iq(), o.iq())
    assert len(np.unique(mm)) == 2     # both meanings present after the switches


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_states_outside_the_tables(pkg, orc, seed):
    """Whatever a diverging Newton iterate can throw at the property functions: saturations outside [0, 1], pressures
    far off the tables (extrapolation), Rs from zero to twice the saturated value, water-filled cells.  Intensive
    quantities, Jacobian and residual stay bit-identical (NaNs, should any appear, in the same places)."""
    case = pkg.decks.cartesian_case(7, 6, 5, state="mixed", heterogeneous=True)
    Nb = case["Nb"]
    rng = np.random.default_rng(100 + seed)
    pv = np.zeros((Nb, 3))
    pv[:, 0] = rng.uniform(-0.1, 1.2, Nb)
    pv[:, 1] = np.exp(rng.uniform(np.log(5e5), np.log(9e7), Nb))
    meaning = rng.integers(0, 2, Nb).astype(np.uint8)
    rs_hi = 2.0 * pkg.decks.rs_sat(case["fluid"], pv[:, 1])
    pv[:, 2] = np.where(meaning == pkg.decks.SW_PO_SG, rng.uniform(-0.1, 1.1, Nb), rng.uniform(0.0, 1.0, Nb) * np.maximum(rs_hi, 1.0))
    pv[rng.random(Nb) < 0.05, 0] = 1.0
    case["pv"], case["meaning"] = np.ascontiguousarray(pv.reshape(-1)), meaning
    m, o = both(pkg, orc, case, reorder="line_coloring")
    assert np.array_equal(m.iq(), o.iq(), equal_nan=True)
    jm, rm = m.assemble(86400.0, 0)
    jo, ro = o.assemble(86400.0, 0)
    assert np.array_equal(rm, ro, equal_nan=True) and np.array_equal(jm, jo, equal_nan=True)
    dx = (rng.standard_normal((Nb, 3)) * np.array([0.5, 5e6, 0.5])).reshape(-1)   # large: every chop and switch rule fires
    assert m.update(dx, 1.0) == o.update(dx)
    pm, mm = m.get_state()
    po, mo = o.get_state()
    assert np.array_equal(mm, mo) and np.array_equal(pm, po, equal_nan=True)
    assert np.array_equal(m.iq(), o.iq(), equal_nan=True)


@pytest.mark.parametrize("reorder", ["graph_coloring_greedy", "line_coloring"])
def test_drift_compensation_matches_oracle(pkg, orc, reorder):
    """EclEnableDriftCompensation (default true, ebos/eclproblem.hh:496-498): after an accepted time step the residual of
    its last linearisation, times dt, is fed back through the source term of the next step (:1126-1135, :1847-1875).
    Device and oracle run two time steps of different length with a deliberately sloppy first step (one Newton update
    only, so the drift is far from zero): Jacobian and residual of the second step bit for bit, with the cap of the
    compensation hit in some cells and not in others; then the same with the switch off."""
    case = pkg.decks.cartesian_case(9, 8, 5, state="mixed", heterogeneous=True)
    src = pkg.decks.five_spot_source(case, rate_sm3_per_day=400.0)
    m, o = both(pkg, orc, case, reorder=reorder)
    for h in (m, o):
        h.set_source(src)
    dt1, dt2 = 5 * 86400.0, 2 * 86400.0
    # time step 1: assemble, one (oracle-computed) Newton update on both sides, final linearisation = "accepted"
    m.assemble(dt1, 0, fetch=False)
    o.assemble(dt1, 0)
    x, _ = o.solve(tol=1e-2)
    assert m.update(x) == o.update(x)
    j1, r1 = m.assemble(dt1, 1)
    jo1, ro1 = o.assemble(dt1, 1)
    assert np.array_equal(r1, ro1) and np.array_equal(j1, jo1)
    # a tight cap so that both branches of "totalDriftRate > maxCompensation" are taken
    tot = (np.abs(ro1.reshape(-1, 3)) * dt1 / (case["volume"] * case["poro"])[:, None]).sum(axis=1)
    cap = float(np.median(tot))
    for h in (m, o):
        h.set_drift_compensation(True, cap)
        h.end_time_step(dt1)
    assert np.array_equal(o.drift(), ro1 * dt1)
    # time step 2 (shorter): iteration 0 and 1
    j2, r2 = m.assemble(dt2, 0)
    jo2, ro2 = o.assemble(dt2, 0)
    assert np.array_equal(r2, ro2) and np.array_equal(j2, jo2)
    # the compensation is really in there: against an assembly without it the residual moves, the Jacobian does not
    o.set_drift_compensation(False)
    jn, rn = o.assemble(dt2, 0)
    assert np.array_equal(jn, jo2) and not np.array_equal(rn, ro2)
    frac = np.abs((rn - ro2).reshape(-1, 3)).sum(axis=1) * dt2 / (case["volume"] * case["poro"])
    assert (frac > 0.99 * cap).any() and (frac < 0.5 * cap).any() and frac.max() <= cap * (1 + 1e-12)
    # switch off on the device too (clears the drift): equals the oracle without compensation
    m.set_drift_compensation(False)
    j3, r3 = m.assemble(dt2, 0)
    assert np.array_equal(r3, rn) and np.array_equal(j3, jn)


def test_drift_survives_a_rolled_back_step(pkg, orc):
    """A failed time step never reaches endTimeStep: update_failed restores the primary variables, the drift of the last
    ACCEPTED step stays (AdaptiveTimeSteppingEbos.hpp:355-441 calls problem.endTimeStep only on success)."""
    case = pkg.decks.cartesian_case(7, 6, 4, state="undersaturated", heterogeneous=True)
    src = pkg.decks.five_spot_source(case, rate_sm3_per_day=300.0)
    m, o = both(pkg, orc, case)
    for h in (m, o):
        h.set_source(src)
    dt = 4 * 86400.0
    m.advance_time_level()
    m.assemble(dt, 0, fetch=False)
    o.assemble(dt, 0)
    x, _ = o.solve(tol=1e-2)
    m.update(x), o.update(x)
    m.assemble(dt, 1, fetch=False)
    o.assemble(dt, 1)
    m.end_time_step(dt), o.end_time_step(dt)
    state = o.get_state()
    # next step: a wild update, then the roll-back
    m.advance_time_level()
    m.assemble(dt, 0, fetch=False)
    m.update(np.full(3 * case["Nb"], 1e-3))
    m.assemble(dt, 1, fetch=False)
    m.update_failed()
    pm, mm = m.get_state()
    assert np.array_equal(pm, state[0]) and np.array_equal(mm, state[1])
    j, r = m.assemble(0.33 * dt, 0)
    jo, ro = o.assemble(0.33 * dt, 0)
    assert np.array_equal(r, ro) and np.array_equal(j, jo)


@pytest.mark.parametrize("reorder", ["graph_coloring", "line_coloring"])
def test_deck_side_inputs_through_the_hot_path(pkg, orc, reorder):
    """transmissibility.py + thpres.py -> set_pattern / set_static: a layered grid with an inactive cell, NTG, MULTZ, two
    NNCs (one on top of a grid face, one between cells the grid does not connect: a pattern no Cartesian stencil has) and
    threshold pressures between three equilibration regions, one of them defaulted from the DEVICE's initial intensive
    quantities (ebos/eclthresholdpressure.hh:96-163).  Jacobian and residual bit for bit against the oracle."""
    T, H = pkg.transmissibility, pkg.thpres
    nx, ny, nz = 6, 5, 7
    rng = np.random.default_rng(11)
    dz = np.repeat(rng.uniform(2.0, 6.0, nz), nx * ny)
    act = np.ones(nx * ny * nz, int); act[40] = 0
    g = T.cartesian_faces(nx, ny, nz, 25.0, 20.0, dz, 2400.0, actnum=act)
    n, F = g["n"], g["faces"]
    perm = (rng.uniform(20.0, 300.0, (n, 1)) * np.array([1.0, 1.0, 0.1])) * 9.869233e-16
    multz = np.where(rng.random(n) < 0.2, 0.05, 1.0)
    t = T.face_transmissibilities(F, g["centroid"], perm, ntg=rng.uniform(0.5, 1.0, n), mult={"Z+": multz})
    c1, c2, t, _ = T.apply_nnc(F["cell1"], F["cell2"], t, nnc=[(3, 4, 0.5 * t[0]), (2, n - 5, 2.0 * t.mean())])
    pat = T.connections_to_pattern(n, c1, c2, t, g["face_area"])
    base = pkg.decks.cartesian_cells(n, 1, 1, 25.0, 20.0, 4.0, 2400.0, 0.25, 100.0, False, "mixed", True, None, 5)
    case = dict(Nb=n, rowptr=pat["rowptr"], col=pat["col"], trans=pat["trans"], area=pat["area"], poro=base["poro"],
                volume=np.ascontiguousarray(g["volume"]), depth=np.ascontiguousarray(g["depth"]), fluid=base["fluid"], pv=base["pv"], meaning=base["meaning"])
    # first pass without thresholds: the initial intensive quantities the defaults are made of
    m0 = pkg.capi.HipModel(case, reorder=reorder)
    m0.set_state(case["pv"], case["meaning"])
    eql = (np.arange(n) * 3) // n
    d = H.default_threshold_pressures(eql, 3, c1, c2, t, np.concatenate([g["face_area"], np.ones(len(t) - len(g["face_area"]))]), m0.iq()[:n], g["depth"])
    assert d[0, 1] > 0.0 and d[1, 2] > 0.0
    mat = H.threshold_pressure_matrix(3, [(1, 2, None), (2, 3, 0.4e5)], eql, c1, c2, defaults=d)
    case["thpres"] = np.ascontiguousarray(H.per_entry(pat["rowptr"], pat["col"], eql, mat))
    assert np.count_nonzero(case["thpres"]) > 0
    m, o = both(pkg, orc, case, reorder=reorder)
    assert np.array_equal(m.iq(), o.iq())
    for it in range(2):
        jm, rm = m.assemble(86400.0, it)
        jo, ro = o.assemble(86400.0, it)
        assert np.array_equal(rm, ro) and np.array_equal(jm, jo)


@pytest.mark.parametrize("reorder", ["graph_coloring", "line_coloring"])
def test_faulted_corner_point_grid_through_the_hot_path(pkg, orc, reorder):
    """transmissibility.cornerpoint_faces -> set_pattern / set_static: a corner-point grid with sheared pillars, dipping layers,
    an inactive cell and a fault of 1.5 layers' throw - cells next to the fault have two lateral neighbours on that side, rows
    of up to 8 blocks, connections between different layers - then Jacobian, residual, a Newton update and a linear solve
    against the oracle."""
    T = pkg.transmissibility
    nx, ny, nz = 6, 4, 7
    coord, zcorn = T.cartesian_cornerpoint(nx, ny, nz, 25.0, 20.0, 4.0, top=2400.0, shear=(0.1, -0.05), fault_i=3, throw=6.0, dip=0.02)
    act = np.ones(nx * ny * nz, int); act[50] = 0
    g = T.cornerpoint_faces(nx, ny, nz, coord, zcorn, actnum=act)
    n, F = g["n"], g["faces"]
    lay = lambda c: g["cart"][c] // (nx * ny)
    assert np.any(lay(F["cell1"]) != lay(F["cell2"]) - (F["face1"] == T.ZP))           # connections across layers at the fault
    rng = np.random.default_rng(4)
    perm = (rng.uniform(20.0, 300.0, (n, 1)) * np.array([1.0, 1.0, 0.1])) * 9.869233e-16
    t = T.face_transmissibilities(F, g["centroid"], perm, ntg=rng.uniform(0.6, 1.0, n))
    assert np.all(t > 0.0)
    pat = T.connections_to_pattern(n, F["cell1"], F["cell2"], t, g["face_area"])
    assert np.diff(pat["rowptr"]).max() == 8
    base = pkg.decks.cartesian_cells(n, 1, 1, 25.0, 20.0, 4.0, 2400.0, 0.25, 100.0, False, "mixed", True, None, 5)
    case = dict(Nb=n, rowptr=pat["rowptr"], col=pat["col"], trans=pat["trans"], area=pat["area"], poro=base["poro"],
                volume=np.ascontiguousarray(g["volume"]), depth=np.ascontiguousarray(g["depth"]), fluid=base["fluid"], pv=base["pv"], meaning=base["meaning"])
    m, o = both(pkg, orc, case, reorder=reorder)
    dt = 86400.0
    for it in range(2):
        jm, rm = m.assemble(dt, it)
        jo, ro = o.assemble(dt, it)
        assert np.array_equal(rm, ro) and np.array_equal(jm, jo)
        if it == 0:
            dx = np.random.default_rng(8).uniform(-1.0, 1.0, (n, 3)) * np.array([0.01, 1e5, 0.01])
            assert m.update(dx.reshape(-1)) == o.update(dx.reshape(-1))
    res = m.solve_jacobian_system()
    xo, ro_ = o.solve_in_order(*m.ordering()[:2])
    assert res.converged and ro_.converged and res.it == ro_.it
    np.testing.assert_allclose(m.get_result(), xo, rtol=1e-9, atol=1e-12 * np.abs(xo).max())


@pytest.mark.parametrize("reorder", ["level_scheduling", "line_coloring"])
def test_well_model_traffic_is_per_cell(pkg, orc, reorder):
    """opmhip_get_iq_cells / opmhip_set_source_cells (what a well model moves per Newton iteration: its perforated cells' records in, their
    rates out - updatePerforationIntensiveQuantities and computeTotalRatesForDof visit the perforations, wells/BlackoilWellModel_impl.hpp:
    1606-1630, 496-512) against the whole-grid forms opmhip_get_iq / opmhip_set_source: the same records bit for bit, in the order asked for,
    repeats included; the same Jacobian and residual bit for bit, a cell named twice receiving the sum; n = 0 clears every source; the error
    returns"""
    case = pkg.decks.cartesian_case(12, 9, 7, state="mixed", heterogeneous=True)
    m, o = both(pkg, orc, case, reorder=reorder)
    Nb = case["Nb"]
    rng = np.random.default_rng(31)
    full = m.iq()
    cells = rng.choice(Nb, 40, replace=False).astype(np.int32)
    cells = np.concatenate([cells, cells[:5], [0, Nb - 1]]).astype(np.int32)           # repeats, both ends of the grid
    got = m.iq_cells(cells)
    assert got.shape == (len(cells),) + full.shape[1:] and np.array_equal(got, full[cells])
    assert m.iq_cells(np.zeros(0, np.int32)).shape[0] == 0
    # sources: 30 distinct cells, five of them named a second time
    c1 = rng.choice(Nb, 30, replace=False).astype(np.int32)
    named = np.concatenate([c1, c1[:5]]).astype(np.int32)
    src = rng.uniform(-1e-3, 1e-3, (len(named), 3))
    dsrc = rng.uniform(-1e-9, 1e-9, (len(named), 3, 3))
    dense, ddense = np.zeros((Nb, 3)), np.zeros((Nb, 3, 3))
    np.add.at(dense, named, src)
    np.add.at(ddense, named, dsrc)
    dt = 86400.0
    m.set_source(dense.reshape(-1), ddense.reshape(-1))
    j0, r0 = m.assemble(dt, 0)
    o.set_source(dense.reshape(-1), ddense.reshape(-1))
    jo, ro = o.assemble(dt, 0)
    assert np.array_equal(j0, jo) and np.array_equal(r0, ro)
    m.set_source(None, None)
    m.set_source_cells(named, src.reshape(-1), dsrc.reshape(-1))
    j1, r1 = m.assemble(dt, 0)
    assert np.array_equal(j1, j0) and np.array_equal(r1, r0)
    m.set_source_cells(c1[:3], src[:3].reshape(-1))                                      # no derivative handed in: zero; the other 27 cells cleared
    j2, r2 = m.assemble(dt, 0)
    d3, z = np.zeros((Nb, 3)), np.zeros((Nb, 3, 3))
    d3[c1[:3]] = src[:3]
    o.set_source(d3.reshape(-1), z.reshape(-1))
    jo2, ro2 = o.assemble(dt, 0)
    assert np.array_equal(j2, jo2) and np.array_equal(r2, ro2) and not np.array_equal(r2, r0)
    m.set_source_cells(np.zeros(0, np.int32), np.zeros(0))                               # n = 0: no sources at all
    j3, r3 = m.assemble(dt, 0)
    o.set_source(np.zeros(3 * Nb), z.reshape(-1))
    jo3, ro3 = o.assemble(dt, 0)
    assert np.array_equal(j3, jo3) and np.array_equal(r3, ro3)
    for bad in ([Nb], [-1]):
        with pytest.raises(pkg.capi.OpmHipError) as e:
            m.iq_cells(np.array(bad, np.int32))
        assert e.value.code == pkg.capi.INVALID_ARGUMENT
        with pytest.raises(pkg.capi.OpmHipError) as e:
            m.set_source_cells(np.array(bad, np.int32), np.zeros(3))
        assert e.value.code == pkg.capi.INVALID_ARGUMENT
    with pytest.raises(pkg.capi.OpmHipError):
        m.set_source_cells(c1[:1], np.array([np.nan, 0.0, 0.0]))
    j4, r4 = m.assemble(dt, 0)                                                           # a refused call left the sources as they were
    assert np.array_equal(j4, j3) and np.array_equal(r4, r3)
