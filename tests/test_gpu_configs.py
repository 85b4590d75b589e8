"""BASELINE.json's remaining parity configurations on the GPU (SURVEY.md §8d recipes; synthetic stand-ins, the decks
themselves are not in the reference tree):
  configs[2] "SPE9 (9000-cell, 25 wells, heterogeneous perm) - well-coupling + ILU0 correctness vs CPU":
             24 x 25 x 15 heterogeneous grid, 26 standard wells (B, C, D^-1 blocks) in the operator - synthetic blocks in the single iterations,
             the well model itself (wells.StandardWells: SPE9's completions, rate targets, BHP limits, a schedule event) over three report steps;
  configs[4] "Norne (faulted corner-point grid, ~44k active cells) - irregular connectivity stress test":
             44 431 rows with 4..12 blocks per row and 2 % long-range NNC couplings, rng(7) (linear algebra and assembly on a
             random graph), and a Norne-SHAPED corner-point grid - 46 x 112 x 22, dome, sheared pillars, three faults, a pinched
             layer, ~44 800 active cells - whose connections and transmissibilities come from transmissibility.py.
Bit-exact where the arithmetic is per row / per face, iteration counts equal, solutions to the stated tolerance."""
import numpy as np
import pytest

import oracle_bind
from helpers import norne_shaped_case, oracle_solve_in_order

pytestmark = pytest.mark.gpu
REORDERS = ["level_scheduling", "graph_coloring", "graph_coloring_greedy", "line_coloring"]


def spe9_wells(case, rng):
    """1 injector (layers 11-15) + 25 producers (layers 2-4) at distinct (i, j), as SPE9 completes them; B and C are the
    4 x 3 well-cell blocks, D^-1 a well-conditioned 4 x 4 (bda/WellContributions.cu:36-126 layout)."""
    nx, ny = case["nx"], case["ny"]
    ij = rng.choice(nx * ny, 26, replace=False)
    perf_cells, vp = [], [0]
    for w, c in enumerate(ij):
        layers = range(10, 15) if w == 0 else range(1, 4)
        perf_cells += [int(c) + nx * ny * k for k in layers]
        vp.append(len(perf_cells))
    n = len(perf_cells)
    D = np.empty((26, 4, 4))
    for w in range(26):
        M = 0.2 * rng.standard_normal((4, 4)) + np.diag(2.0 + rng.random(4))
        D[w] = np.linalg.inv(M)
    cells = np.array(perf_cells, np.int32)
    return dict(numWells=26, val_pointers=np.array(vp, np.int32), Ccols=cells, Bcols=cells.copy(),
                Cnnzs=np.ascontiguousarray(1e-9 * rng.standard_normal(n * 12)), Bnnzs=np.ascontiguousarray(1e-9 * rng.standard_normal(n * 12)),
                Dnnzs=np.ascontiguousarray(D.reshape(-1)))


@pytest.mark.parametrize("reorder", REORDERS)
def test_spe9_shaped_iteration_with_wells(pkg, orc, reorder):
    case = pkg.decks.cartesian_case(24, 25, 15, state="mixed", heterogeneous=True)
    W = spe9_wells(case, np.random.default_rng(9))
    # well source terms on the perforated cells (what computeTotalRatesForDof hands the assembly)
    src = np.zeros((case["Nb"], 3))
    src[W["Ccols"][:5], 1] = 2e-4
    src[W["Ccols"][5:], 0] = -1e-5
    src = np.ascontiguousarray(src.reshape(-1))
    m = pkg.capi.HipModel(case, reorder=reorder, tolerance=1e-2, maxit=200, ilu_relaxation=0.9)
    o = oracle_bind.OracleModel(orc, case)
    for q in (m, o):
        q.set_state(case["pv"], case["meaning"])
        q.set_source(src)
    dt = 5 * 86400.0
    for it in range(3):
        jm, rm = m.assemble(dt, it)
        jo, ro = o.assemble(dt, it)
        if it == 0:
            assert np.array_equal(jm, jo) and np.array_equal(rm, ro)       # same state: bit for bit
        else:  # the two Krylov solutions differ in the last digits (dot-product order), and so do the states
            np.testing.assert_allclose(rm, ro, rtol=1e-6, atol=1e-9 * np.abs(ro).max())
            np.testing.assert_allclose(jm, jo, rtol=1e-5, atol=1e-9 * np.abs(jo).max())
        res = m.solve_jacobian_system(wells=W)
        x = m.get_result()
        xo, reso = o.solve_in_order(*m.ordering()[:2], wells=W, tol=1e-2, maxit=200, w=0.9)
        assert res.converged and reso.converged and res.it == reso.it
        np.testing.assert_allclose(x, xo, rtol=1e-6, atol=1e-8 * np.abs(xo).max())
        # x solves (J - C^T D^-1 B) x = r to the tolerance
        r = rm - orc.wells_apply(W, x, orc.spmv(case["Nb"], case["rowptr"], case["col"], jm, x))
        assert np.linalg.norm(r) < 1e-2 * np.linalg.norm(rm) * (1 + 1e-9)
        m.update(None, 1.0)
        o.update(xo)
    pm, mm = m.get_state()
    po, mo = o.get_state()
    assert np.array_equal(mm, mo)
    np.testing.assert_allclose(pm, po, rtol=1e-7, atol=1e-9)


def norne_like_graph(Nb=44431, seed=7):
    """symmetric pattern: row lengths drawn from {4..12} (local couplings at small offsets), 2 % of the rows get one
    long-range NNC partner"""
    rng = np.random.default_rng(seed)
    nbrs = [set([i]) for i in range(Nb)]
    offs = np.array([1, 2, 3, 46, 47, 113, 114, 5150, 5151, 5200, 5300])
    want = rng.integers(4, 13, Nb)
    for i in range(Nb):
        for o in offs:
            if len(nbrs[i]) >= want[i]:
                break
            j = i + int(o)
            if j < Nb and len(nbrs[j]) < 12:
                nbrs[i].add(j)
                nbrs[j].add(i)
    for i in np.flatnonzero(rng.random(Nb) < 0.02):
        j = int(rng.integers(0, Nb))
        if j != i:
            nbrs[int(i)].add(j)
            nbrs[j].add(int(i))
    rowptr = np.zeros(Nb + 1, np.int32)
    cols = []
    for i in range(Nb):
        cols.extend(sorted(nbrs[i]))
        rowptr[i + 1] = len(cols)
    return Nb, rowptr, np.array(cols, np.int32)


@pytest.fixture(scope="module")
def norne():
    return norne_like_graph()


@pytest.mark.parametrize("reorder", REORDERS)
def test_norne_like_linear_algebra(pkg, orc, norne, reorder):
    Nb, rp, ci = norne
    rng = np.random.default_rng(21)
    nnzb = len(ci)
    val = rng.uniform(-1, 1, size=(nnzb, 3, 3)) * 0.2
    row = np.repeat(np.arange(Nb), np.diff(rp))
    dk = np.flatnonzero(ci == row)
    s = np.zeros((Nb, 3))
    np.add.at(s, row, np.abs(val).sum(axis=2))
    for e in range(3):
        val[dk, e, e] = 1.5 * (s[:, e] + 0.5)
    v = np.ascontiguousarray(val.reshape(-1))
    b = rng.standard_normal(Nb * 3)
    sol = pkg.capi.HipSolver(tolerance=1e-6, maxit=200, reorder=reorder)
    res = sol.solve_system(Nb, rp, ci, v.copy(), b)
    x = sol.get_result()
    to, fr, rpc = sol.ordering()
    rr, rc, rv = orc.reorder_matrix(Nb, rp, ci, v, to, fr)
    assert np.array_equal(sol.ilu0_factor(), orc.ilu0_factor(Nb, rr, rc, rv))          # factors: bit for bit
    y = rng.standard_normal(Nb * 3)
    yo = orc.spmv(Nb, rr, rc, rv, y.reshape(Nb, 3)[fr].reshape(-1)).reshape(Nb, 3)[to].reshape(-1)
    assert np.array_equal(sol.spmv(y), yo)                                               # operator in the device's order: bit for bit
    np.testing.assert_allclose(sol.spmv(y), orc.spmv(Nb, rp, ci, v, y), rtol=1e-12, atol=1e-12)
    # preconditioner application: bit for bit (a wrong M^-1 would still let BiCGStab converge, so test it by itself)
    d = rng.standard_normal(Nb * 3)
    vo = orc.ilu0_apply(Nb, rr, rc, orc.ilu0_factor(Nb, rr, rc, rv), d.reshape(Nb, 3)[fr].reshape(-1), w=0.9, mode="post_scale")
    assert np.array_equal(sol.ilu0_apply(d), vo.reshape(Nb, 3)[to].reshape(-1))
    xo, ro = oracle_solve_in_order(orc, Nb, rp, ci, v, b, to, fr, tol=1e-6, maxit=200, w=0.9)
    assert res.converged and ro.converged and res.it == ro.it
    np.testing.assert_allclose(x, xo, rtol=1e-8, atol=1e-12)


@pytest.mark.parametrize("reorder", ["graph_coloring", "line_coloring"])
def test_norne_like_assembly_bitwise(pkg, orc, norne, reorder):
    Nb, rp, ci = norne
    rng = np.random.default_rng(33)
    fl = pkg.fluid.spe1_fluid()[0]
    row = np.repeat(np.arange(Nb), np.diff(rp))
    # per-face data must be symmetric: draw per unordered pair
    lo, hi = np.minimum(row, ci), np.maximum(row, ci)
    key = lo.astype(np.int64) * Nb + hi
    uniq, inv = np.unique(key, return_inverse=True)
    t_face = np.exp(rng.normal(np.log(5e-13), 1.0, len(uniq)))
    a_face = rng.uniform(50.0, 400.0, len(uniq))
    trans, area = t_face[inv], a_face[inv]
    trans[row == ci] = 0.0
    area[row == ci] = 0.0
    depth = 2500.0 + 400.0 * np.sort(rng.random(Nb))
    volume = rng.uniform(500.0, 4000.0, Nb)
    poro = rng.uniform(0.1, 0.3, Nb)
    p = 250e5 + 7000.0 * (depth - 2500.0) * (1.0 + rng.uniform(-0.01, 0.01, Nb))
    meaning = np.where(depth < np.median(depth), pkg.decks.SW_PO_SG, pkg.decks.SW_PO_RS).astype(np.uint8)
    pv = np.zeros((Nb, 3))
    pv[:, 0] = 0.2 + rng.uniform(-0.02, 0.02, Nb)
    pv[:, 1] = p
    pv[:, 2] = np.where(meaning == pkg.decks.SW_PO_SG, 0.1 + rng.uniform(-0.02, 0.02, Nb), 0.8 * pkg.decks.rs_sat(fl, p))
    case = dict(Nb=Nb, rowptr=rp, col=ci, trans=np.ascontiguousarray(trans), area=np.ascontiguousarray(area), poro=poro,
                volume=volume, depth=np.ascontiguousarray(depth), fluid=fl, pv=np.ascontiguousarray(pv.reshape(-1)), meaning=meaning)
    m = pkg.capi.HipModel(case, reorder=reorder)
    o = oracle_bind.OracleModel(orc, case)
    m.set_state(case["pv"], case["meaning"])
    o.set_state(case["pv"], case["meaning"])
    assert np.array_equal(m.iq(), o.iq())
    for it, dt in ((0, 86400.0), (1, 86400.0)):
        jm, rm = m.assemble(dt, it)
        jo, ro = o.assemble(dt, it)
        if it == 0:
            assert np.array_equal(rm, ro) and np.array_equal(jm, jo)
        else:  # after one update from two Krylov solutions that agree to ~1e-9
            np.testing.assert_allclose(rm, ro, rtol=1e-6, atol=1e-9 * np.abs(ro).max())
            np.testing.assert_allclose(jm, jo, rtol=1e-5, atol=1e-9 * np.abs(jo).max())
        if it == 0:
            res = m.solve_jacobian_system()
            xo, reso = o.solve_in_order(*m.ordering()[:2], tol=1e-2, maxit=200, w=0.9)
            assert res.converged and res.it == reso.it
            m.update(None, 1.0)
            o.update(xo)
    np.testing.assert_allclose(m.convergence(86400.0)[11:17], o.convergence(86400.0)[11:17], rtol=1e-6, atol=1e-12)


@pytest.fixture(scope="module")
def norne_grid(pkg):
    return norne_shaped_case(pkg)


@pytest.mark.parametrize("reorder", ["graph_coloring", "line_coloring"])
def test_norne_shaped_corner_point_grid(pkg, orc, norne_grid, reorder):
    """configs[4] with geometry: the faulted, pinched, partly inactive corner-point grid of helpers.norne_shaped_grid through
    set_pattern / set_static, then what a Newton iteration does - intensive quantities, Jacobian and residual bit for bit,
    the ILU0 factors bit for bit, the solve with the same number of half iterations, update, second assembly."""
    case, g, dims = norne_grid
    n = case["Nb"]
    rl = np.diff(case["rowptr"])
    F = g["faces"]
    lay = g["cart"] // (dims[0] * dims[1])
    lat = F["face1"] < 4
    assert 44000 < n < 46000 and rl.min() >= 2 and rl.max() >= 11
    assert (lay[F["cell1"]][lat] != lay[F["cell2"]][lat]).sum() > 10000        # cells meeting other layers across the faults
    assert ((lay[F["cell2"]] - lay[F["cell1"]])[~lat] > 1).sum() > 2000        # connections across the pinched-out layer
    m = pkg.capi.HipModel(case, reorder=reorder)
    o = oracle_bind.OracleModel(orc, case)
    m.set_state(case["pv"], case["meaning"])
    o.set_state(case["pv"], case["meaning"])
    assert np.array_equal(m.iq(), o.iq())
    dt = 86400.0
    jm, rm = m.assemble(dt, 0)
    jo, ro = o.assemble(dt, 0)
    assert np.array_equal(rm, ro) and np.array_equal(jm, jo)
    res = m.solve_jacobian_system()
    xo, reso = o.solve_in_order(*m.ordering()[:2], tol=1e-2, maxit=200, w=0.9)
    assert res.converged and reso.converged and res.it == reso.it
    np.testing.assert_allclose(m.get_result(), xo, rtol=1e-8, atol=1e-12 * np.abs(xo).max())
    m.update(None, 1.0)
    o.update(xo)
    jm, rm = m.assemble(dt, 1)
    jo, ro = o.assemble(dt, 1)
    np.testing.assert_allclose(rm, ro, rtol=1e-6, atol=1e-9 * np.abs(ro).max())
    np.testing.assert_allclose(jm, jo, rtol=1e-5, atol=1e-9 * np.abs(jo).max())
    np.testing.assert_allclose(m.convergence(dt)[11:17], o.convergence(dt)[11:17], rtol=1e-6, atol=1e-12)


def test_spe9_shaped_well_residual_and_recovery(pkg, orc):
    """The two other touch points of the standard wells' Schur complement on device-resident vectors:
    r -= C^T D^-1 resWell before the solve (bit for bit) and xw = D^-1 (resWell - B x) after it."""
    case = pkg.decks.cartesian_case(24, 25, 15, state="mixed", heterogeneous=True)
    rng = np.random.default_rng(10)
    W = spe9_wells(case, rng)
    res_well = 1e-3 * rng.standard_normal(4 * W["numWells"])
    m = pkg.capi.HipModel(case, reorder="line_coloring", tolerance=1e-6, maxit=200)
    o = oracle_bind.OracleModel(orc, case)
    for q in (m, o):
        q.set_state(case["pv"], case["meaning"])
    dt = 5 * 86400.0
    jm, rm = m.assemble(dt, 0)
    jo, ro = o.assemble(dt, 0)
    assert np.array_equal(m.get_rhs(), ro)
    m.wells_apply_residual(W, res_well)
    r2 = orc.wells_apply_residual(W, res_well, ro)
    assert np.array_equal(m.get_rhs(), r2) and not np.array_equal(r2, ro)
    res = m.solve_jacobian_system(wells=W)
    x = m.get_result()
    assert res.converged
    resid = r2 - orc.wells_apply(W, x, orc.spmv(case["Nb"], case["rowptr"], case["col"], jo, x))
    assert np.linalg.norm(resid) < 1e-6 * np.linalg.norm(r2) * 1.001
    xw = m.wells_recover_solution(W, res_well)
    np.testing.assert_array_equal(xw, orc.wells_recover(W, res_well, x))  # same x in, same operation order: same bits


def test_norne_shaped_grid_with_the_features_the_norne_deck_uses(pkg, orc, norne_grid):
    """configs[4] closer to the deck: on the faulted corner-point grid a fluid with PVTG (vaporised oil), two saturation
    regions, per-cell scaled saturation end points (ENDSCALE with SCALECRS and vertical scaling, as the deck's SWL / SWCR / SGU /
    KRW ... arrays do), relative-permeability hysteresis (SATOPTS HYSTER / EHYSTR / IMBNUM with ISWL ... end points of the
    imbibition curves, as the deck has them), DRSDT with its time-step bookkeeping, irreversible rock compaction tables -
    everything at once through begin_time_step / assemble / solve / update / end_time_step for three time steps (sources
    reversed in the last: scanning curves), device against oracle bit for bit."""
    from test_oracle_endscale import corey_fluid, scaled_points
    import helpers
    base, g, dims = norne_grid
    n = base["Nb"]
    rng = np.random.default_rng(23)
    wet = helpers.wetgas_fluid(pkg, rocktab=helpers.ROCKTAB_2)
    sat0 = corey_fluid(pkg).sat[0]
    sat1 = dict(swof=[[r[0], 0.9 * r[1], r[2], 1.3 * r[3]] for r in sat0["swof"]], sgof=[[r[0], r[1], 0.95 * r[2], r[3]] for r in sat0["sgof"]])
    fl = pkg.fluid.Fluid(wet.pvt, [sat0, sat1], rock_pref=wet.rock_pref, rock_cr=wet.rock_cr, rocktab=helpers.ROCKTAB_2, pc_scaling=True)
    case = dict(base, fluid=fl)
    case["satnum"] = (rng.random(n) < 0.4).astype(np.int32)
    case["rocknum"] = (rng.random(n) < 0.5).astype(np.int32)
    # scaled end points around each cell's own table
    u = [oracle_bind.sat_end_points(orc, fl, s) for s in (0, 1)]
    pts = np.array([scaled_points(u[s], rng) for s in case["satnum"]])
    es = dict(sat_scaling=1, three_point_kr=1, krw=2, kro=2, krg=2, pcw=1, pcg=1)
    for f, name in enumerate(pkg.capi.EPS_FIELDS):
        es[name] = np.ascontiguousarray(pts[:, f])
    case["endscale"] = es
    # state: keep Sw above every cell's connate water, the gas cap gets the third meaning in places
    pv = case["pv"].reshape(-1, 3).copy()
    pv[:, 0] = es["swl"] + 0.05 + 0.1 * rng.random(n)
    mng = case["meaning"].copy()
    cap = np.flatnonzero(mng == 0)
    dry = cap[rng.random(len(cap)) < 0.3]
    mng[dry] = 2
    pv[dry, 2] = helpers.rv_sat(fl, pv[dry, 1]) * 0.7
    case["pv"], case["meaning"] = np.ascontiguousarray(pv.reshape(-1)), mng
    m = pkg.capi.HipModel(case, reorder="line_coloring")
    o = oracle_bind.OracleModel(orc, case)
    for q in (m, o):
        q.set_state(case["pv"], case["meaning"])
        q.set_composition_change_limits([3.0e-6], [0], [5.0e-12])
        q.set_irreversible_compaction(True)
    # hysteresis: every cell's imbibition curves are those of the OTHER saturation region, with end points of their own
    imbnum = (1 - case["satnum"]).astype(np.int32)
    ptsI = np.array([scaled_points(u[s], rng) for s in imbnum])
    imb = {name: np.ascontiguousarray(ptsI[:, f]) for f, name in enumerate(pkg.capi.EPS_FIELDS)}
    for q in (m, o):
        q.set_hysteresis(1, imbnum, imb)
    assert np.array_equal(m.iq(), o.iq())
    src = np.zeros((n, 3))
    wellcells = rng.choice(n, 12, replace=False)
    src[wellcells[:6], 1] = 2e-3       # water in
    src[wellcells[6:], 0] = -1.5e-3    # oil out
    src = src.reshape(-1)
    scanning = 0
    for step, dt in enumerate([0.5 * 86400.0, 86400.0, 86400.0]):
        for q in (m, o):
            q.set_source(src if step < 2 else -src)
            q.begin_time_step(dt)
        for a, b in zip(m.hysteresis(), o.hysteresis()):
            assert np.array_equal(a, b)
        scanning += int(np.sum(1.0 - m.iq()[:, 1, 0] > m.hysteresis()[0]))
        assert np.array_equal(m.iq(), o.iq())
        for it in range(2):
            jm, rm = m.assemble(dt, it)
            jo, ro = o.assemble(dt, it)
            assert np.array_equal(jm, jo) and np.array_equal(rm, ro), (step, it)
            res = m.solve_jacobian_system()
            assert res.converged
            x = m.get_result()
            m.update(x, 1.0)
            o.update(x)
            pm, mm = m.get_state()
            po, mo = o.get_state()
            assert np.array_equal(mm, mo) and np.array_equal(pm, po), (step, it)
        for q in (m, o):
            q.end_time_step(dt)
        for a, b in zip(m.trackers(), o.trackers()):
            assert np.array_equal(a, b)
    assert len(set(mm.tolist())) == 3 and scanning > 0


def test_spe1case1_report_steps(pkg, orc):
    """BASELINE.json configs[0] as the deck it is (python/test_data/SPE1CASE1/SPE1CASE1.DATA): the deck's own SOLUTION section - EQUIL
    8400 ft / 4800 psia, RSVD 1.27 Mscf/stb, equilibrated on the DEVICE's property functions by equil.equilibrate - and its SCHEDULE:
    DRSDT 0 (the Rs cap of eclproblem.hh:1711-1732, through opmhip_set_composition_change_limits), the gas injector (100 MMscf/day into
    (1, 1, 1)) and the oil producer (20 000 stb/day out of (10, 10, 3)) as standard wells - well equations assembled on the host
    (wells.StandardWells), eliminated by the device (wells_apply_residual, the operator form inside the solve, wells_recover_solution) -
    over the first three report steps (TSTEP 31 28 31 days) under Flow's adaptive time-step control.  Device against oracle running the
    SAME loop: equal sub-steps, Newton and linear iteration counts, final pressures / saturations / Rs to 1e-7, the well state, and the
    DRSDT cap in force.  (What this does NOT pin: the reference's own numbers for this deck - the tree holds no output of it.)"""
    fl = pkg.fluid.spe1_fluid()[0]
    case = pkg.decks.spe1_case()                         # EQUIL on capi.HipFluid: the device's functions
    case_o = pkg.decks.spe1_case(props=oracle_bind.OracleFluid(orc, fl))
    assert np.array_equal(case["pv"], case_o["pv"]) and np.array_equal(case["meaning"], case_o["meaning"])   # bit-identical property functions
    pv0 = case["pv"].reshape(-1, 3)
    psia = 6894.757293168361
    # the deck's datum: 4800 psia at 8400 ft = the centre depth of layer 3; undersaturated oil at the RSVD value, connate water
    np.testing.assert_allclose(pv0[200:, 1], 4800.0 * psia, rtol=1e-6)
    assert np.all(pv0[:, 0] == 0.12) and np.all(case["meaning"] == 1)
    np.testing.assert_allclose(pv0[:, 2], 1.27 * 178.10760667903526, rtol=1e-12)
    assert case["drsdt"] == [0.0] and len(case["schedule"]["tstep"]) == 12
    m = pkg.capi.HipModel(case, tolerance=1e-2, maxit=200, ilu_relaxation=0.9)
    om = oracle_bind.OracleModel(orc, case)
    runs = []
    for side, h in (("device", m), ("oracle", om)):
        h.set_state(case["pv"], case["meaning"])
        h.set_composition_change_limits(drsdt=case["drsdt"], drsdt_all_cells=case["drsdt_all_cells"])
        wells = pkg.decks.spe1_wells(case)
        if side == "oracle":
            hm = oracle_bind.OracleAsHipModel(om, tol=1e-2, maxit=200, w=0.9)
            hm.kw["order"] = m.ordering()[:2]            # the ILU0 in the ordering the device chose
        else:
            hm = h
        model = pkg.newton.BlackoilModelHip(hm, well_model=wells)
        ts = pkg.newton.AdaptiveTimeStepping(model, pkg.newton.TimeSteppingParameters(initial_dt=86400.0))
        steps = []
        for length in case["schedule"]["tstep"][:3]:
            reps = ts.advance_report_step(length)
            steps.append((len(reps), sum(r.total_linear_iterations for r in reps)))
        runs.append(dict(steps=steps, history=list(ts.history), time=ts.time, wells=wells.x.copy(), controls=[w.control[0] for w in wells.wells],
                         state=h.get_state(), iq=h.iq()))
    dev, ora = runs
    np.testing.assert_allclose([dev["time"], ora["time"]], sum(case["schedule"]["tstep"][:3]), rtol=1e-12)
    assert all(ok for _, _, ok in dev["history"]) and all(ok for _, _, ok in ora["history"])       # no sub-step was chopped on either side
    # How close can the two runs be?  They start bit-identical (state, well blocks, J, r: tools/spe1_debug.py) and differ only in the order
    # the scalar products of BiCGStab are summed in: 3e-15 in the first update.  A linear solve to 1e-2 on a Jacobian of condition 1e5 ... 1e6
    # turns that into 1e-10, and the deck is not smooth there - gas appears at Sg = 0, a node of SGOF, where kr's slope jumps - so that a
    # few Newton iterations later two solves stop half an iteration apart and the runs are two equally valid paths through the same time
    # steps, no longer one path computed twice.  DRSDT 0 with its option ALL makes that the rule: every undersaturated cell sits exactly ON
    # its cap (Rs = lastRs + 0), where min(Rs, cap) changes branch - and d Rs / d(primary variable) jumps between 1 and 0 - on the last bit.
    # Hence: the first Newton iterations in lock step (test below), the whole run to what two converged runs share - Newton iterations within
    # two per report step, linear iterations per Newton iteration within a third, the state to the reach of the Newton tolerances.
    (pd, md), (po, mo) = dev["state"], ora["state"]
    pd2, po2 = pd.reshape(-1, 3), po.reshape(-1, 3)
    # (the primary variable's MEANING is not compared: with every cell on its Rs cap, "undersaturated at Rs = cap" and "saturated with Sg = 0"
    #  are one physical state under two names, and which name a cell carries hangs on the last bit; saturations and Rs are compared instead)
    qd, qo = dev["iq"], ora["iq"]
    print("SPE1CASE1, 3 report steps: (Newton, linear) per step device %r oracle %r; sub-steps %r | %r; cells whose meaning differs %d; max relative "
          "pressure difference %.2e, |dS| %.2e, Rs %.2e, bhp %.2e" %
          (dev["steps"], ora["steps"], [round(h[0] / 86400.0, 2) for h in dev["history"]], [round(h[0] / 86400.0, 2) for h in ora["history"]], int((md != mo).sum()),
           np.abs(pd2[:, 1] / po2[:, 1] - 1).max(), np.abs(qd[:, 0:3, 0] - qo[:, 0:3, 0]).max(), np.abs(qd[:, 15, 0] / qo[:, 15, 0] - 1).max(),
           np.abs(dev["wells"][:, 3] / ora["wells"][:, 3] - 1).max()))
    for (nd, ld), (no, lo) in zip(dev["steps"], ora["steps"]):
        assert abs(nd - no) <= 2 and abs(ld / nd - lo / no) <= 0.35 * max(ld / nd, lo / no), (dev["steps"], ora["steps"])
    assert sum(n for n, _ in dev["steps"]) >= 20 and abs(len(dev["history"]) - len(ora["history"])) <= 1
    np.testing.assert_allclose(pd2[:, 1], po2[:, 1], rtol=2e-4)                        # pressures: 1 psi in 5000 - two converged runs, CNV 1e-2 / MB 1e-6 (measured: 1.1e-5)
    np.testing.assert_allclose(qd[:, 0:3, 0], qo[:, 0:3, 0], atol=1e-4)                # saturations of the three phases
    np.testing.assert_allclose(qd[:, 15, 0], qo[:, 15, 0], rtol=2e-4)                  # Rs
    np.testing.assert_allclose(dev["wells"][:, 3], ora["wells"][:, 3], rtol=2e-4)      # bottom-hole pressures
    assert dev["controls"] == ["rate", "rate"]                                       # both wells still on their rate targets
    qg_inj, bhp_inj = dev["wells"][0, 2], dev["wells"][0, 3]
    qo_prod, bhp_prod = dev["wells"][1, 0], dev["wells"][1, 3]
    np.testing.assert_allclose([qg_inj, -qo_prod], [case["schedule"]["wells"][0]["surface_rate"], case["schedule"]["wells"][1]["oil_rate"]], rtol=1e-9)
    assert 1000.0 * psia < bhp_prod < 4800.0 * psia < bhp_inj < 9014.0 * psia
    # the injected gas has formed a free-gas region around the injector; DRSDT 0: nowhere did Rs rise above its initial value
    pd = pd.reshape(-1, 3)
    assert qd[0, 2, 0] > 0.05 and (qd[:, 2, 0] > 1e-3).sum() >= 3                     # free gas at the injector and around it
    assert np.all(qd[:, 15, 0] <= 1.27 * 178.10760667903526 * (1 + 1e-12))           # Rs of every cell: DRSDT 0 / ALL holds it at or below its initial value


def test_spe9_shaped_schedule_with_standard_wells(pkg, orc):
    """BASELINE.json configs[2] with wells that are wells: SPE9's injector (five completions) and 25 producers (three each) as
    wells.StandardWells on the 24 x 25 x 15 log-normal stand-in grid with a gas cap, over the shape of SPE9's schedule - producers at 1500
    stb/day, cut to 100 stb/day at a report step (StandardWells.set_rate_target: the PRODUCTION_UPDATE event), back at 1500 - under Flow's
    time-step control.  Producers fall to their BHP limit and return, the injector meets its upper limit: 26 wells x 4 unknowns eliminated
    by the device in every linear solve (wells_apply_residual, the operator C^T D^-1 B inside BiCGStab, wells_recover_solution).  Device
    against the oracle running the SAME loop (tests/test_spe9_shaped_wells.py holds the CPU side: finite differences of the blocks, the
    three components' balance): the same sub-steps, Newton iterations and controls at the end of every report step, the states and the
    well unknowns to 1e-7.  (The SPE9 deck is not in the reference tree: no number of it is pinned here.)"""
    from test_spe9_shaped_wells import DAY, PRODUCER_BHP_LIMIT, SCHEDULE, run_schedule
    case = pkg.decks.cartesian_case(24, 25, 15, dx=91.44, dy=91.44, dz=6.0, heterogeneous=True, state="mixed")
    m = pkg.capi.HipModel(case, tolerance=1e-2, maxit=200, ilu_relaxation=0.9)
    om = oracle_bind.OracleModel(orc, case)
    runs = []
    for side, h in (("device", m), ("oracle", om)):
        h.set_state(case["pv"], case["meaning"])
        wells = pkg.decks.spe9_shaped_wells(case, producer_bhp_limit=PRODUCER_BHP_LIMIT)
        if side == "oracle":
            hm = oracle_bind.OracleAsHipModel(om, tol=1e-2, maxit=200, w=0.9)
            hm.kw["order"] = m.ordering()[:2]            # the ILU0 in the ordering the device chose
        else:
            hm = h
        ts, steps = run_schedule(pkg, hm, wells, SCHEDULE)
        runs.append(dict(steps=steps, history=list(ts.history), time=ts.time, wells=wells.x.copy(), state=h.get_state(), iq=h.iq()))
    dev, ora = runs
    np.testing.assert_allclose([dev["time"], ora["time"]], 30 * DAY, rtol=1e-12)
    assert all(ok for _, _, ok in dev["history"]) and all(ok for _, _, ok in ora["history"])
    (pd, md), (po, mo) = dev["state"], ora["state"]
    pd2, po2 = pd.reshape(-1, 3), po.reshape(-1, 3)
    qd, qo = dev["iq"], ora["iq"]
    rate_scale = np.abs(ora["wells"][:, :3]).max(axis=1, keepdims=True)
    print("SPE9-shaped schedule, 3 report steps: (Newton, linear, controls) device %r oracle %r; sub-steps %r | %r; cells whose meaning differs %d; max relative "
          "pressure difference %.2e, |dS| %.2e, Rs %.2e, bhp %.2e, rates / the well's largest %.2e" %
          (dev["steps"], ora["steps"], [round(h[0] / DAY, 2) for h in dev["history"]], [round(h[0] / DAY, 2) for h in ora["history"]], int((md != mo).sum()),
           np.abs(pd2[:, 1] / po2[:, 1] - 1).max(), np.abs(qd[:, 0:3, 0] - qo[:, 0:3, 0]).max(), np.abs(qd[:, 15, 0] / qo[:, 15, 0] - 1).max(),
           np.abs(dev["wells"][:, 3] / ora["wells"][:, 3] - 1).max(), (np.abs(dev["wells"][:, :3] - ora["wells"][:, :3]) / rate_scale).max()))
    for (nd, ld, cd), (no, lo, co) in zip(dev["steps"], ora["steps"]):
        assert cd == co, (cd, co)                                                      # every well under the same control at the end of the report step
        # this run is smooth where SPE1 under DRSDT 0 is not (no cell on a switching threshold): the two sides stay one path - the same
        # Newton iterations, the linear iterations within a few (the scalar products' order moves a solve by half an iteration now and then)
        assert nd == no and abs(ld - lo) <= 5, (dev["steps"], ora["steps"])
    assert dev["steps"][0][2].count("B") >= 3 and dev["steps"][1][2] == "B" + "R" * 25 and dev["steps"][2][2][0] == "B"
    assert [h[0] for h in dev["history"]] == [h[0] for h in ora["history"]]                 # the same sub-steps
    # measured: pressures 3e-10, saturations 2e-9, Rs 1e-9, bottom-hole pressures 3e-10, rates 5e-9 of the well's largest
    np.testing.assert_allclose(pd2[:, 1], po2[:, 1], rtol=1e-7)
    np.testing.assert_allclose(qd[:, 0:3, 0], qo[:, 0:3, 0], atol=1e-7)
    np.testing.assert_allclose(qd[:, 15, 0], qo[:, 15, 0], rtol=1e-7)
    np.testing.assert_allclose(dev["wells"][:, 3], ora["wells"][:, 3], rtol=1e-7)
    assert (np.abs(dev["wells"][:, :3] - ora["wells"][:, :3]) / rate_scale).max() <= 1e-6
    assert np.array_equal(md, mo)                                                     # (no cell sits on a switching threshold here, unlike SPE1 under DRSDT 0)


def test_spe1case1_first_time_step_in_lock_step(pkg, orc):
    """the deck's first sub-step (1 day) Newton iteration by Newton iteration on both sides: intensive quantities, well blocks, Jacobian and
    residual of iteration 0 and the residual after the wells' elimination bit for bit, its linear solve on the oracle's half iteration with
    the oracle's reduction and solution (1e-9: the scalar products' order); the next iteration on the same half iteration, later ones within
    one; the converged state of the step to 1e-4 - the lock-step part of what test_spe1case1_report_steps cannot ask of three months"""
    from helpers import oracle_solve_in_order
    case = pkg.decks.spe1_case()
    m = pkg.capi.HipModel(case, tolerance=1e-2, maxit=200, ilu_relaxation=0.9)
    om = oracle_bind.OracleModel(orc, case)
    for h in (m, om):
        h.set_state(case["pv"], case["meaning"])
        h.set_composition_change_limits(drsdt=case["drsdt"], drsdt_all_cells=case["drsdt_all_cells"])
        h.begin_time_step(86400.0)
    to, fr, _ = m.ordering()
    wd, wo = pkg.decks.spe1_wells(case), pkg.decks.spe1_wells(case)
    nm = pkg.newton.BlackoilModelHip(m)
    dt = 86400.0
    for it in range(12):
        iqd, iqo = m.iq(), om.iq()
        if it == 0:
            assert np.array_equal(iqd, iqo)
            wd.solve_well_equations(iqd)
            wo.solve_well_equations(iqo)
        ad, ao = wd.assemble(iqd, case["Nb"]), wo.assemble(iqo, case["Nb"])
        m.set_source(ad["source"], ad["dsource"])
        om.set_source(ao["source"], ao["dsource"])
        jd, rd = m.assemble(dt, it)
        jo, ro = om.assemble(dt, it)
        if it == 0:
            assert all(np.array_equal(ad["wells"][k], ao["wells"][k]) for k in ("Cnnzs", "Bnnzs", "Dnnzs")) and np.array_equal(ad["res_well"], ao["res_well"])
            assert np.array_equal(jd, jo) and np.array_equal(rd, ro)
        conv, _ = nm.get_convergence(dt, it)
        if conv and it > 1 and wd.converged(ad["res_well"]):
            break
        m.wells_apply_residual(ad["wells"], ad["res_well"])
        ro2 = orc.wells_apply_residual(ao["wells"], ao["res_well"], ro)
        if it == 0:
            assert np.array_equal(m.get_rhs(), ro2)          # r -= C^T D^-1 r_w: the same bits
        res = m.solve_jacobian_system(wells=ad["wells"])
        xo, reso = oracle_solve_in_order(orc, case["Nb"], case["rowptr"], case["col"], jo, ro2, to, fr, wells=ao["wells"], tol=1e-2, maxit=200, w=0.9)
        # (iteration 0: identical systems, the scalar products' order alone; from then on two neighbouring systems - see above)
        assert res.converged and reso.converged and abs(res.it - reso.it) <= (0.0 if it < 2 else 1.0), (it, res.it, reso.it)
        if it == 0:
            assert abs(res.reduction - reso.reduction) <= 1e-9 * reso.reduction
            np.testing.assert_allclose(m.get_result(), xo, rtol=1e-9, atol=1e-12 * np.abs(xo).max())
        wd.update(m.wells_recover_solution(ad["wells"], ad["res_well"]))
        wo.update(orc.wells_recover(ao["wells"], ao["res_well"], xo))
        m.update(None, 1.0)
        om.update(xo)
    assert 2 <= it <= 10
    (pd, md), (po, mo) = m.get_state(), om.get_state()
    print("first sub-step: %d Newton iterations; max relative pressure difference %.2e, bhp %.2e" %
          (it, np.abs(pd.reshape(-1, 3)[:, 1] / po.reshape(-1, 3)[:, 1] - 1).max(), np.abs(wd.x[:, 3] / wo.x[:, 3] - 1).max()))
    qd, qo = m.iq(), om.iq()      # (saturations and Rs instead of the primary variables' meanings: see test_spe1case1_report_steps)
    np.testing.assert_allclose(pd.reshape(-1, 3)[:, 1], po.reshape(-1, 3)[:, 1], rtol=1e-5)
    np.testing.assert_allclose(qd[:, 0:3, 0], qo[:, 0:3, 0], atol=1e-6)
    np.testing.assert_allclose(qd[:, 15, 0], qo[:, 15, 0], rtol=1e-5)
    np.testing.assert_allclose(wd.x[:, 3], wo.x[:, 3], rtol=1e-5)
