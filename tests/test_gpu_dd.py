"""Domain-decomposed device path on ONE GPU: P contexts in one process (one host thread each) joined by libopmhip's
loopback communicator, i.e. the same set_pattern_dd / set_halo / halo-exchange / all-reduce code the RCCL path runs,
with device copies in place of ncclSend/ncclRecv.  Oracle: the CPU restatement on the GLOBAL grid, whose ILU0 drops the
couplings between subdomains (block-Jacobi, the reference's ghost_last_bilu0_decomposition,
opm/simulators/linalg/ParallelOverlappingILU0.hpp:439-494)."""
import threading
import uuid

import numpy as np
import pytest

import oracle_bind

pytestmark = pytest.mark.gpu


def run_ranks(world, fn):
    """fn(rank) in one thread per rank; returns the list of results, re-raises the first exception"""
    out, err = [None] * world, [None] * world

    def body(r):
        try:
            out[r] = fn(r)
        except BaseException as e:  # noqa: BLE001
            err[r] = e

    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    for e in err:
        if e is not None:
            raise e
    return out


def global_and_parts(pkg, n, world, **kw):
    px, py, pz = pkg.ras.block_layout(world)
    g = pkg.decks.cartesian_case(px * n, py * n, pz * n, **kw)
    owner = pkg.ras.cartesian_owner(px * n, py * n, pz * n, px, py, pz)
    parts = [pkg.ras.cartesian_subdomain_case(n, world, r, **kw) for r in range(world)]
    return g, owner, parts


@pytest.mark.parametrize("world,n", [(2, 6), (4, 6), (8, 6), (2, 28)])   # 28^3 cells per rank: > 512 tiles, the one-launch local reduction
def test_dd_assembly_bitwise_and_solve(pkg, orc, world, n):
    g, owner, parts = global_and_parts(pkg, n, world, state="mixed", heterogeneous=True)
    src = pkg.decks.five_spot_source(g, rate_sm3_per_day=30.0)
    o = oracle_bind.OracleModel(orc, g)
    o.set_state(g["pv"], g["meaning"])
    o.set_source(src)
    dt = 86400.0
    jo, ro = o.assemble(dt, 0)
    co = o.convergence(dt)
    xo, reso = orc.solve(g["Nb"], g["rowptr"], g["col"], jo, ro, tol=1e-2, maxit=200, w=0.9, owner=owner)
    group = "t" + uuid.uuid4().hex

    def rank_fn(r):
        c = parts[r]
        m = pkg.capi.HipModel(c, comm=("loopback", world, r, group), reorder="level_scheduling")
        m.set_state(c["pv"], c["meaning"])
        m.set_source(np.ascontiguousarray(src.reshape(-1, 3)[c["gids"]].reshape(-1)))
        j, res = m.assemble(dt, 0)
        conv = m.convergence(dt)
        sol = m.solve_jacobian_system()
        x = m.get_result()
        return j, res, conv, sol.it, sol.converged, x

    outs = run_ranks(world, rank_fn)
    for r, (j, res, conv, it, ok, x) in enumerate(outs):
        c = parts[r]
        gi = c["gids"][:c["Nb"]]
        # residual and Jacobian of the owned rows: the global oracle's, bit for bit
        assert np.array_equal(res.reshape(-1, 3)[:c["Nb"]], ro.reshape(-1, 3)[gi])
        assert np.array_equal(j.reshape(-1, 9), jo.reshape(-1, 9)[c["halo"]["entry_global"]])
        # convergence: global quantities, identical on every rank
        np.testing.assert_allclose(conv[3:6], co[3:6], rtol=0, atol=0)
        np.testing.assert_allclose(conv[6:10], co[6:10], rtol=1e-13)
        assert ok and it == reso.it
        np.testing.assert_allclose(x.reshape(-1, 3)[:c["Nb"]], xo.reshape(-1, 3)[gi], rtol=1e-8, atol=1e-12 * np.abs(xo).max())
    assert all(np.array_equal(outs[0][2], o_[2]) for o_ in outs)  # every rank holds the same reduced numbers


@pytest.mark.parametrize("world", [2, 4, 8])
def test_dd_degenerate_upwind_ties_bitwise(pkg, orc, world):
    """Unperturbed homogeneous state cut into 2/4/8 subdomains: pressure difference exactly 0 and equal volumes on every
    horizontal face, including the faces between subdomains, where the tie-break of ebos/eclfluxmodule.hh:303-314 must
    look at GLOBAL ids (local ids put every ghost after every owned cell).  Owned rows: the global oracle's bit for bit."""
    n = 5
    g, owner, parts = global_and_parts(pkg, n, world, state="mixed", heterogeneous=False, perturb=False)
    o = oracle_bind.OracleModel(orc, g)
    o.set_state(g["pv"], g["meaning"])
    dt = 86400.0
    jo, ro = o.assemble(dt, 0)
    group = "u" + uuid.uuid4().hex

    def rank_fn(r):
        c = parts[r]
        m = pkg.capi.HipModel(c, comm=("loopback", world, r, group), reorder="line_coloring")
        m.set_state(c["pv"], c["meaning"])
        return m.assemble(dt, 0)

    for r, (j, res) in enumerate(run_ranks(world, rank_fn)):
        c = parts[r]
        assert np.array_equal(res.reshape(-1, 3)[:c["Nb"]], ro.reshape(-1, 3)[c["gids"][:c["Nb"]]])
        assert np.array_equal(j.reshape(-1, 9), jo.reshape(-1, 9)[c["halo"]["entry_global"]])


@pytest.mark.parametrize("reorder", ["line_coloring"])
def test_dd_newton_step_matches_block_jacobi_oracle(pkg, orc, reorder):
    world, n = 2, 8
    g, owner, parts = global_and_parts(pkg, n, world, state="mixed", heterogeneous=False)
    src = pkg.decks.five_spot_source(g, rate_sm3_per_day=40.0)
    dt = 2 * 86400.0
    group = "n" + uuid.uuid4().hex

    def rank_fn(r):
        c = parts[r]
        m = pkg.capi.HipModel(c, comm=("loopback", world, r, group), reorder=reorder)
        m.set_state(c["pv"], c["meaning"])
        m.set_source(np.ascontiguousarray(src.reshape(-1, 3)[c["gids"]].reshape(-1)))
        drv = pkg.newton.BlackoilModelHip(m)
        rep = drv.step(dt)
        pv, mean = m.get_state()
        return rep.total_newton_iterations, rep.total_linear_iterations, pv, mean

    outs = run_ranks(world, rank_fn)
    assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1]
    # the decomposed run converges to the same state as the single-domain CPU restatement (Newton tolerance level)
    o = oracle_bind.OracleModel(orc, g)
    o.set_state(g["pv"], g["meaning"])
    o.set_source(src)
    for it in range(20):
        o.assemble(dt, it)
        c = o.convergence(dt)
        if it > 1 and np.all(c[11:14] < 1e-2) and np.all(c[14:17] < 1e-6):
            break
        x, res = o.solve(tol=1e-2)
        o.update(x)
    po, mo = o.get_state()
    for r in range(world):
        c = parts[r]
        gi = c["gids"]
        pv = outs[r][2].reshape(-1, 3)
        assert np.array_equal(outs[r][3][:c["Nb"]], mo[gi[:c["Nb"]]])
        np.testing.assert_allclose(pv[:c["Nb"], 1], po.reshape(-1, 3)[gi[:c["Nb"]], 1], rtol=1e-4)       # pressure
        np.testing.assert_allclose(pv[:c["Nb"], 0], po.reshape(-1, 3)[gi[:c["Nb"]], 0], atol=2e-3)       # Sw
        # ghost copies equal their owners' values
        other = outs[1 - r][2].reshape(-1, 3)
        og = parts[1 - r]["gids"][:parts[1 - r]["Nb"]]
        lut = {int(gg): k for k, gg in enumerate(og)}
        for k in range(c["Nb"], c["Nloc"]):
            assert np.array_equal(pv[k], other[lut[int(gi[k])]])


def test_rccl_communicator_single_rank(pkg):
    """RCCL itself (dlopen, ncclCommInitRank, ncclAllReduce on the context's stream) with a one-rank communicator: the
    part of the multi-GPU path that can be exercised on a single GPU."""
    import ctypes as C
    uid = pkg.capi.comm_unique_id()
    assert len(uid) == 128
    s = pkg.capi.HipSolver()
    L = pkg.capi.lib()
    L.opmhip_comm_selftest.argtypes = [C.c_void_p, C.c_void_p]
    s._check(L.opmhip_comm_init_rccl(s._h, 1, 0, uid))
    out = (C.c_double * 2)()
    s._check(L.opmhip_comm_selftest(s._h, out))
    assert (out[0], out[1]) == (1.0, 2.0)
    # what RCCL itself says about the communicator (bench.py prints this for N > 1: ncclCommCount must equal --gpus)
    info = s.comm_info()
    assert info == {"nranks": 1, "rank": 0, "device": 0, "kind": "rccl"}
    assert s.comm_selftest() == (1.0, 2.0)
    assert pkg.capi.HipSolver().comm_info()["kind"] == "none"


def irregular_global_case(pkg, Nb=3000, seed=41):
    """an unstructured three-phase case: random sparse symmetric connectivity, random face and cell data"""
    rng = np.random.default_rng(seed)
    nbrs = [set([i]) for i in range(Nb)]
    for i in range(Nb):
        for o in (1, 13, 170):
            if i + o < Nb and rng.random() < 0.8:
                nbrs[i].add(i + o)
                nbrs[i + o].add(i)
    for i in np.flatnonzero(rng.random(Nb) < 0.05):
        j = int(rng.integers(0, Nb))
        if j != i:
            nbrs[int(i)].add(j)
            nbrs[j].add(int(i))
    rp = np.zeros(Nb + 1, np.int32)
    cols = []
    for i in range(Nb):
        cols.extend(sorted(nbrs[i]))
        rp[i + 1] = len(cols)
    ci = np.array(cols, np.int32)
    row = np.repeat(np.arange(Nb), np.diff(rp))
    lo, hi = np.minimum(row, ci), np.maximum(row, ci)
    _, inv = np.unique(lo.astype(np.int64) * Nb + hi, return_inverse=True)
    trans = np.exp(rng.normal(np.log(5e-13), 1.0, inv.max() + 1))[inv]
    area = rng.uniform(50.0, 400.0, inv.max() + 1)[inv]
    trans[row == ci] = 0.0
    area[row == ci] = 0.0
    fl = pkg.fluid.spe1_fluid()[0]
    depth = 2500.0 + 300.0 * np.sort(rng.random(Nb))
    p = 250e5 + 7000.0 * (depth - 2500.0) * (1.0 + rng.uniform(-0.01, 0.01, Nb))
    meaning = np.where(depth < np.median(depth), pkg.decks.SW_PO_SG, pkg.decks.SW_PO_RS).astype(np.uint8)
    pv = np.zeros((Nb, 3))
    pv[:, 0] = 0.2 + rng.uniform(-0.02, 0.02, Nb)
    pv[:, 1] = p
    pv[:, 2] = np.where(meaning == pkg.decks.SW_PO_SG, 0.1 + rng.uniform(-0.02, 0.02, Nb), 0.8 * pkg.decks.rs_sat(fl, p))
    return dict(Nb=Nb, rowptr=rp, col=ci, trans=np.ascontiguousarray(trans), area=np.ascontiguousarray(area), poro=rng.uniform(0.1, 0.3, Nb),
                volume=rng.uniform(500.0, 4000.0, Nb), depth=np.ascontiguousarray(depth), fluid=fl, pv=np.ascontiguousarray(pv.reshape(-1)),
                meaning=meaning)


@pytest.mark.parametrize("reorder", ["graph_coloring", "line_coloring"])
def test_dd_irregular_graph_three_ranks(pkg, orc, reorder):
    """Three subdomains (not a power of two) of an unstructured graph with scattered ownership: ragged halos, every rank
    a neighbour of every other.  Owned rows of the assembly bit for bit against the single-domain oracle, block-Jacobi
    solve with the oracle's iteration count."""
    world = 3
    g = irregular_global_case(pkg)
    Nb = g["Nb"]
    rng = np.random.default_rng(8)
    owner = ((np.arange(Nb) // 97 + rng.integers(0, 2, Nb) * (rng.random(Nb) < 0.03)) % world).astype(np.int32)
    o = oracle_bind.OracleModel(orc, g)
    o.set_state(g["pv"], g["meaning"])
    dt = 86400.0
    jo, ro = o.assemble(dt, 0)
    group = "i" + uuid.uuid4().hex
    parts = []
    for r in range(world):
        lp = pkg.ras.local_problem(g["rowptr"], g["col"], owner, r)
        cells = lp["cells"]
        parts.append(dict(Nb=lp["Nown"], Nghost=lp["Nghost"], Nloc=lp["Nown"] + lp["Nghost"], rowptr=lp["rows"], col=lp["cols"],
                          trans=np.ascontiguousarray(g["trans"][lp["entry"]]), area=np.ascontiguousarray(g["area"][lp["entry"]]),
                          poro=np.ascontiguousarray(g["poro"][cells]), volume=np.ascontiguousarray(g["volume"][cells]),
                          depth=np.ascontiguousarray(g["depth"][cells]), fluid=g["fluid"],
                          pv=np.ascontiguousarray(g["pv"].reshape(-1, 3)[cells].reshape(-1)), meaning=np.ascontiguousarray(g["meaning"][cells]),
                          gids=lp["gids"], halo=lp, global_cells=Nb))

    def rank_fn(r):
        c = parts[r]
        m = pkg.capi.HipModel(c, comm=("loopback", world, r, group), reorder=reorder)
        m.set_state(c["pv"], c["meaning"])
        j, res = m.assemble(dt, 0)
        sol = m.solve_jacobian_system()
        return j, res, sol.it, sol.converged, m.get_result()

    outs = run_ranks(world, rank_fn)
    for r, (j, res, it, ok, x) in enumerate(outs):
        c = parts[r]
        gi = c["gids"][:c["Nb"]]
        assert np.array_equal(res.reshape(-1, 3)[:c["Nb"]], ro.reshape(-1, 3)[gi])
        assert np.array_equal(j.reshape(-1, 9), jo.reshape(-1, 9)[c["halo"]["entry"]])
        assert ok
    # all ranks agree on the iteration count, and together they hold a solution of the global system
    assert len({o_[2] for o_ in outs}) == 1
    x_glob = np.zeros((Nb, 3))
    for r, (_, _, _, _, x) in enumerate(outs):
        c = parts[r]
        x_glob[c["gids"][:c["Nb"]]] = x.reshape(-1, 3)[:c["Nb"]]
    A_x = orc.spmv(Nb, g["rowptr"], g["col"], jo, x_glob.reshape(-1))
    assert np.linalg.norm(ro - A_x) < 1e-2 * np.linalg.norm(ro) * (1 + 1e-9)


def test_dd_time_step_roll_back_restores_owned_and_ghost_cells(pkg):
    """update_failed in a decomposed run: owned and ghost primary variables, meanings and intensive quantities of every
    rank are those of the start of the time step again, with no communication (the saved time level holds the ghosts)."""
    world, n = 2, 7
    parts = [pkg.ras.cartesian_subdomain_case(n, world, r, state="mixed", heterogeneous=True, rate_scale=60.0) for r in range(world)]
    group = "r" + uuid.uuid4().hex
    dt = 20 * 86400.0

    def rank_fn(r):
        c = parts[r]
        m = pkg.capi.HipModel(c, comm=("loopback", world, r, group), reorder="line_coloring")
        m.set_state(c["pv"], c["meaning"])
        m.set_source(c["source"])
        m.advance_time_level()
        iq0 = m.iq()
        j0, r0 = m.assemble(dt, 0)
        for it in range(3):
            if it:
                m.assemble(dt, it, fetch=False)
            assert m.solve_jacobian_system().converged
            m.update(None, 1.0)
        pv1, _ = m.get_state()
        m.update_failed()
        pv2, mean2 = m.get_state()
        j2, r2 = m.assemble(dt, 0)
        return (not np.array_equal(pv1, c["pv"]), np.array_equal(pv2, c["pv"]), np.array_equal(mean2, c["meaning"]),
                np.array_equal(m.iq(), iq0), np.array_equal(j2, j0) and np.array_equal(r2, r0))

    for flags in run_ranks(world, rank_fn):
        assert all(flags), flags


@pytest.mark.parametrize("world,prec,ilu", [(2, "cpr_quasiimpes", 0), (4, "cpr", 0), (2, "cpr_quasiimpes", 2)])
def test_dd_cpr_one_hierarchy_per_subdomain(pkg, orc, world, prec, ilu):
    """CPR in a decomposed run with opmhip_config.cpr_gather_rows < 0 (nothing between the subdomains): every rank builds the pressure hierarchy of its own subdomain (owned rows and columns; the
    couplings to ghost cells left out, as the block ILU0 leaves them out) - the preconditioner application of every rank is
    the oracle's CPR of that subdomain's matrix bit for bit, and the solve takes the half-iteration count of the oracle's
    BiCGStab on the global system with one CPR per subdomain"""
    n = 8
    g, owner, parts = global_and_parts(pkg, n, world, state="mixed", heterogeneous=True)
    src = pkg.decks.five_spot_source(g, rate_sm3_per_day=30.0)
    o = oracle_bind.OracleModel(orc, g)
    o.set_state(g["pv"], g["meaning"])
    o.set_source(src)
    dt = 5 * 86400.0
    jo, ro = o.assemble(dt, 0)
    wts = o.true_impes_weights(dt) if prec == "cpr" else None
    group = "p" + uuid.uuid4().hex
    rng = np.random.default_rng(7)
    probe = rng.standard_normal(3 * g["Nb"])

    def rank_fn(r):
        c = parts[r]
        Nb = c["Nb"]
        m = pkg.capi.HipModel(c, comm=("loopback", world, r, group), reorder="line_coloring", preconditioner=prec, tolerance=1e-2, cpr_gather_rows=-1, cpr_amg_ilu_levels=ilu)
        m.set_state(c["pv"], c["meaning"])
        m.set_source(np.ascontiguousarray(src.reshape(-1, 3)[c["gids"]].reshape(-1)))
        j, res = m.assemble(dt, 0)
        sol = m.solve_jacobian_system()
        x = m.get_result()
        d = np.ascontiguousarray(probe.reshape(-1, 3)[c["gids"][:Nb]].reshape(-1))
        v = m.cpr_apply(d)
        return j, sol.it, sol.converged, x, d, v, m.ordering(), m.cpr_levels()[0], m.cpr_weights()

    outs = run_ranks(world, rank_fn)
    # the oracle's solve: the global system with every subdomain's rows in the ordering its rank's block ILU0 uses
    frg = np.concatenate([np.asarray(parts[r]["gids"][:parts[r]["Nb"]])[outs[r][6][1]] for r in range(world)]).astype(np.int32)
    tog = np.empty_like(frg)
    tog[frg] = np.arange(len(frg), dtype=np.int32)
    gr, gc, gv = orc.reorder_matrix(g["Nb"], g["rowptr"], g["col"], jo, tog, frg)
    xo, reso, lev = orc.cpr_solve_blocks(g["Nb"], gr, gc, gv, np.ascontiguousarray(ro.reshape(-1, 3)[frg].reshape(-1)), np.asarray(owner)[frg],
                                         weights=None if wts is None else wts[frg], natural=frg, tol=1e-2)
    xo = xo.reshape(-1, 3)[tog].reshape(-1)
    assert reso.converged and lev.min() >= 2
    for r, (j, it, ok, x, d, v, (to, fr, _), levels, w) in enumerate(outs):
        c = parts[r]
        Nb, rp, ci = c["Nb"], np.asarray(c["rowptr"]), np.asarray(c["col"])
        gi = c["gids"][:Nb]
        assert np.all(np.diff(gi) > 0)         # local numbering = the global order restricted to the subdomain
        # the subdomain's own matrix: owned rows, owned columns
        keep = ci < Nb
        rowof = np.repeat(np.arange(Nb), np.diff(rp))
        lrp = np.concatenate([[0], np.cumsum(np.bincount(rowof[keep], minlength=Nb))]).astype(np.int32)
        lci = np.ascontiguousarray(ci[keep], np.int32)
        lv = np.ascontiguousarray(j.reshape(-1, 9)[keep].reshape(-1))
        rr, rc, rv = orc.reorder_matrix(Nb, lrp, lci, lv, to, fr)
        cpr = oracle_bind.OracleCpr(orc)
        cpr.set_natural_ids(fr)
        if ilu:
            cpr.set_ilu_smoother(ilu, 1)   # the AMG's finest levels smooth with ILU0 (ghost columns play no part in it)
        if wts is not None:
            assert np.array_equal(w.reshape(-1, 3), wts[gi])
            cpr.set_weights(np.ascontiguousarray(wts[gi][fr].reshape(-1)))
        cpr.update(Nb, rr, rc, rv)
        vo = cpr.apply(np.ascontiguousarray(d.reshape(Nb, 3)[fr].reshape(-1))).reshape(Nb, 3)[to].reshape(-1)
        assert np.array_equal(v[:3 * Nb], vo), r
        assert levels == [int(q) for q in cpr.levels()[0]]
        assert ok and (ilu or abs(it - reso.it) <= 1.0)   # (the global oracle solve runs the Jacobi-smoothed hierarchies)
    # the two solutions: both meet the tolerance on the global system.  (They are not compared entry by entry: with the
    # couplings between the subdomains missing from the pressure hierarchy the first BiCGStab steps overshoot - the residual
    # grows fourfold before it falls - and the different summation orders of the scalar products show in the iterates.)
    xd = np.zeros((g["Nb"], 3))
    for r in range(world):
        xd[parts[r]["gids"][:parts[r]["Nb"]]] = outs[r][3].reshape(-1, 3)[:parts[r]["Nb"]]
    rn = np.linalg.norm(ro)
    assert np.linalg.norm(orc.spmv(g["Nb"], g["rowptr"], g["col"], jo, xd.reshape(-1)) - ro) < 1e-2 * rn
    assert np.linalg.norm(orc.spmv(g["Nb"], g["rowptr"], g["col"], jo, xo) - ro) < 1e-2 * rn


def test_dd_relative_change_is_summed_over_the_ranks(pkg):
    """BlackoilModelEbos::relativeChange ends in gridView.comm().sum (flow/BlackoilModelEbos.hpp:501-502): every rank of a
    decomposed run gets the same number, owned cells only, and it is the single-domain number up to the order of the sums"""
    world, n = 4, 6
    g, owner, parts = global_and_parts(pkg, n, world, state="mixed", heterogeneous=True)
    rng = np.random.default_rng(5)
    dpv = rng.normal(size=(g["Nb"], 3)) * np.array([0.01, 2e5, 0.005])     # a made-up change of the solution, natural order of the global grid
    single = pkg.capi.HipModel(g, reorder="line_coloring")
    single.set_state(g["pv"], g["meaning"])
    single.advance_time_level()
    single.update(dpv.reshape(-1), 1.0)             # a Newton update (chopped and switched cell by cell, the same on every side)
    want = single.relative_change()
    assert want > 0.0
    group = "c" + uuid.uuid4().hex

    def rank_fn(r):
        c = parts[r]
        m = pkg.capi.HipModel(c, comm=("loopback", world, r, group), reorder="line_coloring")
        m.set_state(c["pv"], c["meaning"])
        m.advance_time_level()
        gid = np.asarray(c["gids"])
        m.update(dpv[gid].reshape(-1), 1.0)
        return m.relative_change()

    got = run_ranks(world, rank_fn)
    assert len(set(got)) == 1                       # one all-reduced number
    assert got[0] == pytest.approx(want, rel=1e-13)


def test_dd_wet_gas_and_rock_tables_bitwise(pkg, orc):
    """The extended record layout (wet gas: Rv, third primary-variable meaning; ROCKTAB multipliers with overburden) in a
    decomposed run: ghost cells carry the 19-field records, faces towards them use Rv and the transmissibility multiplier
    of a cell another rank owns.  Owned rows of the Jacobian, residual, the update with its switches: the global oracle's,
    bit for bit, over two Newton iterations."""
    import helpers
    world, n = 4, 5
    px, py, pz = pkg.ras.block_layout(world)
    g = helpers.wetgas_case(pkg, px * n, py * n, pz * n, rocktab=helpers.ROCKTAB_2, heterogeneous=True)
    parts = []
    for r in range(world):
        c = pkg.ras.cartesian_subdomain_case(n, world, r, state="mixed", heterogeneous=True, fluid=g["fluid"])
        gid = c["gids"]
        c["pv"] = np.ascontiguousarray(g["pv"].reshape(-1, 3)[gid].reshape(-1))
        c["meaning"] = np.ascontiguousarray(g["meaning"][gid])
        c["rocknum"] = np.ascontiguousarray(g["rocknum"][gid])
        c["overburden"] = np.ascontiguousarray(g["overburden"][gid])
        parts.append(c)
    o = oracle_bind.OracleModel(orc, g)
    o.set_state(g["pv"], g["meaning"])
    dt = 2 * 86400.0
    rng = np.random.default_rng(9)
    dxs = [(rng.standard_normal((g["Nb"], 3)) * np.array([0.04, 2e5, 0.04])).reshape(-1) for _ in range(2)]
    ref = []
    for it in range(2):
        jo, ro = o.assemble(dt, it)
        o.update(dxs[it])
        po, mo = o.get_state()
        ref.append((jo, ro, po.copy(), mo.copy()))
    group = "t" + uuid.uuid4().hex

    def rank_fn(r):
        c = parts[r]
        m = pkg.capi.HipModel(c, comm=("loopback", world, r, group), reorder="line_coloring")
        m.set_state(c["pv"], c["meaning"])
        out = []
        for it in range(2):
            j, res = m.assemble(dt, it)
            m.update(np.ascontiguousarray(dxs[it].reshape(-1, 3)[c["gids"][:c["Nb"]]].reshape(-1)), 1.0)
            p, mm = m.get_state()
            out.append((j, res, p.copy(), mm.copy()))
        return out

    outs = run_ranks(world, rank_fn)
    assert len(np.unique(ref[-1][3])) == 3          # all three meanings live
    for r, per_it in enumerate(outs):
        c = parts[r]
        gi = c["gids"][:c["Nb"]]
        for it, (j, res, p, mm) in enumerate(per_it):
            jo, ro, po, mo = ref[it]
            assert np.array_equal(res.reshape(-1, 3)[:c["Nb"]], ro.reshape(-1, 3)[gi])
            assert np.array_equal(j.reshape(-1, 9), jo.reshape(-1, 9)[c["halo"]["entry_global"]])
            assert np.array_equal(mm[:c["Nb"]], mo[gi]) and np.array_equal(p.reshape(-1, 3)[:c["Nb"]], po.reshape(-1, 3)[gi])


def periodic_self_coupled_system(nx, ny, nz, seed=3):
    """A box of nx x ny x nz cells that is PERIODIC in x, written as a one-rank decomposition: the neighbour across the x
    faces is the rank itself, so cell (0, j, k) couples to a ghost image of (nx - 1, j, k) and vice versa.  -> local pattern
    (owned rows, ghost columns Nb ...), halo lists with the single neighbour 0, random diagonally dominant blocks, and the
    same operator as a scipy matrix over the owned cells."""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    Nb = nx * ny * nz
    cid = lambda i, j, k: i + nx * (j + ny * k)
    jj, kk = np.meshgrid(np.arange(ny), np.arange(nz), indexing="ij")
    left, right = cid(0, jj.ravel(), kk.ravel()), cid(nx - 1, jj.ravel(), kk.ravel())
    # ghosts: first the images of the right column (neighbours of the left cells), then the images of the left column
    ghost_of = {}
    for g, cell in enumerate(np.concatenate([right, left])):
        ghost_of[("R" if g < len(right) else "L", int(cell))] = Nb + g
    rows, cols_l, cols_g = [0], [], []
    for c in range(Nb):
        i, j, k = c % nx, (c // nx) % ny, c // (nx * ny)
        loc, glob = [c], [c]
        for d, (a, b, e) in enumerate([(-1, 0, 0), (1, 0, 0), (0, -1, 0), (0, 1, 0), (0, 0, -1), (0, 0, 1)]):
            ii, j2, k2 = i + a, j + b, k + e
            if not (0 <= j2 < ny and 0 <= k2 < nz):
                continue
            if ii < 0:
                loc.append(ghost_of[("R", int(cid(nx - 1, j, k)))]); glob.append(int(cid(nx - 1, j, k)))
            elif ii >= nx:
                loc.append(ghost_of[("L", int(cid(0, j, k)))]); glob.append(int(cid(0, j, k)))
            else:
                loc.append(int(cid(ii, j2, k2))); glob.append(int(cid(ii, j2, k2)))
        o = np.argsort(loc)
        cols_l += [loc[q] for q in o]
        cols_g += [glob[q] for q in o]
        rows.append(len(cols_l))
    rows, cols_l, cols_g = np.array(rows, np.int32), np.array(cols_l, np.int32), np.array(cols_g, np.int64)
    vals = rng.standard_normal((len(cols_l), 3, 3)) * 0.3
    rowid = np.repeat(np.arange(Nb), np.diff(rows))
    vals[cols_l == rowid] += 6.0 * np.eye(3)
    A = sp.bsr_matrix(sp.coo_matrix((vals.transpose(0, 1, 2).reshape(-1),
                                     ((3 * rowid[:, None, None] + np.arange(3)[None, :, None] + 0 * np.arange(3)[None, None, :]).reshape(-1),
                                      (3 * cols_g[:, None, None] + 0 * np.arange(3)[None, :, None] + np.arange(3)[None, None, :]).reshape(-1))),
                                    shape=(3 * Nb, 3 * Nb)).tocsr())
    halo = dict(neigh=np.array([0], np.int32), send_ptr=np.array([0, 2 * ny * nz], np.int32),
                send_cells=np.concatenate([right, left]).astype(np.int32), recv_ptr=np.array([0, 2 * ny * nz], np.int32))
    return Nb, 2 * ny * nz, rows, cols_l, vals.reshape(-1), halo, A


@pytest.mark.parametrize("transport", ["rccl", "loopback"])
@pytest.mark.parametrize("reorder,pipe", [("line_coloring", 64), ("graph_coloring", 0), ("level_scheduling", -1)])
def test_halo_exchange_beside_the_interior_product_one_rank_periodic(pkg, transport, reorder, pipe):
    """The operator application of decomposed runs - ev_x -> interior tiles | pack, ncclSend / ncclRecv on the halo stream ->
    boundary tiles - with REAL RCCL point-to-point calls on ONE GPU: the rank is its own neighbour (a box periodic in x).
    y = A x must equal the periodic operator formed by scipy (every row the same sum whichever launch multiplied it), the
    interior / boundary split must be a partition of the tiles, and BiCGStab with the block-Jacobi ILU0 must solve the
    periodic system; the loopback transport runs the same code with device copies."""
    import ctypes as C
    nx, ny, nz = 12, 10, 9
    Nb, Ng, rows, cols, vals, halo, A = periodic_self_coupled_system(nx, ny, nz)
    L = pkg.capi.lib()
    s = pkg.capi.HipSolver(reorder=reorder, tolerance=1e-9, maxit=100, spmv_pipe_wgs=pipe)
    pkg.capi.HipFluid  # binds the assembly-side entry points (set_pattern_dd, comm_*, set_halo)
    if not getattr(L, "_asm_bound", False):
        pkg.capi._bind_assembly(L)
        L._asm_bound = True
    s._check(L.opmhip_set_pattern_dd(s._h, Nb, Ng, len(cols), rows.ctypes.data_as(C.c_void_p), cols.ctypes.data_as(C.c_void_p)))
    s.Nb, s.nnzb = Nb, len(cols)
    if transport == "rccl":
        s._check(L.opmhip_comm_init_rccl(s._h, 1, 0, pkg.capi.comm_unique_id()))
    else:
        s._check(L.opmhip_comm_init_loopback(s._h, 1, 0, ("self" + uuid.uuid4().hex).encode()))
    keep = [halo[k] for k in ("neigh", "send_ptr", "send_cells", "recv_ptr")]
    s._check(L.opmhip_set_halo(s._h, Nb, 1, *[k.ctypes.data_as(C.c_void_p) for k in keep]))
    rng = np.random.default_rng(8)
    b = rng.standard_normal(3 * Nb)
    s.upload_system(vals, b)
    for _ in range(3):      # repeated: the send buffer and the two events are reused
        x = rng.standard_normal(3 * Nb)
        y = s.spmv(x)
        ref = A @ x
        assert np.max(np.abs(y - ref)) <= 1e-13 * np.max(np.abs(ref))
    res = s.solve_system(Nb, None, None, None, None)
    assert res.converged
    xs = s.get_result()
    assert np.linalg.norm(b - A @ xs) <= 1.01e-9 * np.linalg.norm(b)
    # and the same rows give the same bits when nothing is overlapped: a context without the halo stream cannot form the
    # periodic product, but it can form the owned-columns part - compare through linearity instead: A (x1 + x2) = A x1 + A x2
    x1, x2 = rng.standard_normal(3 * Nb), rng.standard_normal(3 * Nb)
    assert np.allclose(s.spmv(x1 + x2), s.spmv(x1) + s.spmv(x2), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("world,prec,n,rows", [(2, "cpr_quasiimpes", 8, 100), (4, "cpr", 8, 40), (8, "cpr_quasiimpes", 6, 1000), (2, "cpr_quasiimpes", 12, 0)])
def test_dd_cpr_pressure_stage_across_the_ranks(pkg, orc, world, prec, n, rows):
    """opmhip_config.cpr_gather_rows: the pressure stage of a decomposed CPR spans the ranks - level 0 smooths with the whole system's
    pressure operator (ghost entries exchanged), its residual goes down to each rank's first level of at most `rows` rows, those levels
    are joined (the couplings between the subdomains included), gathered and cycled on by every rank, and the post-smoothing residual is
    the whole system's too: the coarse part of the reference's parallel AMG (OwningTwoLevelPreconditioner.hpp).  Every rank's
    application = the oracle's (orc_cpr_solve_blocks with gather_rows) bit for bit; the solves stop on the same half iteration (+-1:
    the scalar products are summed in different orders) and need fewer iterations than one hierarchy per subdomain.  rows = 1000: above
    the subdomain size, level 0 itself is joined; rows = 0: the default."""
    g, owner, parts = global_and_parts(pkg, n, world, state="mixed", heterogeneous=True)
    src = pkg.decks.five_spot_source(g, rate_sm3_per_day=30.0)
    o = oracle_bind.OracleModel(orc, g)
    o.set_state(g["pv"], g["meaning"])
    o.set_source(src)
    dt = 5 * 86400.0
    jo, ro = o.assemble(dt, 0)
    wts = o.true_impes_weights(dt) if prec == "cpr" else None
    rng = np.random.default_rng(7)
    probe = rng.standard_normal(3 * g["Nb"])

    def run(gather_rows):
        group = "p" + uuid.uuid4().hex

        def rank_fn(r):
            c = parts[r]
            Nb = c["Nb"]
            m = pkg.capi.HipModel(c, comm=("loopback", world, r, group), reorder="line_coloring", preconditioner=prec, tolerance=1e-4, cpr_gather_rows=gather_rows, cpr_amg_ilu_levels=0)
            m.set_state(c["pv"], c["meaning"])
            m.set_source(np.ascontiguousarray(src.reshape(-1, 3)[c["gids"]].reshape(-1)))
            m.assemble(dt, 0)
            sol = m.solve_jacobian_system()
            x = m.get_result()
            d = np.ascontiguousarray(probe.reshape(-1, 3)[c["gids"][:Nb]].reshape(-1))
            v = m.cpr_apply(d)
            return sol.it, sol.converged, x, v, m.ordering(), m.cpr_levels()[0]
        return run_ranks(world, rank_fn)

    outs = run(rows)
    frg = np.concatenate([np.asarray(parts[r]["gids"][:parts[r]["Nb"]])[outs[r][4][1]] for r in range(world)]).astype(np.int32)
    tog = np.empty_like(frg)
    tog[frg] = np.arange(len(frg), dtype=np.int32)
    gr, gc, gv = orc.reorder_matrix(g["Nb"], g["rowptr"], g["col"], jo, tog, frg)
    xo, reso, lev, glev, pv = orc.cpr_solve_blocks(g["Nb"], gr, gc, gv, np.ascontiguousarray(ro.reshape(-1, 3)[frg].reshape(-1)), np.asarray(owner)[frg],
                                                   weights=None if wts is None else wts[frg], natural=frg, tol=1e-4, gather_rows=rows,
                                                   probe=np.ascontiguousarray(probe.reshape(-1, 3)[frg].reshape(-1)))
    pv = pv.reshape(-1, 3)[tog]
    assert reso.converged
    for r, (it, ok, x, v, (to, fr, _), levels) in enumerate(outs):
        c = parts[r]
        Nb = c["Nb"]
        assert levels[-len(glev):] == glev and len(levels) == int(lev[r]) + len(glev), (levels, lev, glev)
        assert np.array_equal(v[:3 * Nb].reshape(-1, 3), pv[c["gids"][:Nb]]), r
        assert ok and abs(it - reso.it) <= 1.0, (it, reso.it)
    xd = np.zeros((g["Nb"], 3))
    for r in range(world):
        xd[parts[r]["gids"][:parts[r]["Nb"]]] = outs[r][2].reshape(-1, 3)[:parts[r]["Nb"]]
    assert np.linalg.norm(orc.spmv(g["Nb"], g["rowptr"], g["col"], jo, xd.reshape(-1)) - ro) < 1e-4 * np.linalg.norm(ro)
    alone = run(-1)   # one hierarchy per subdomain, nothing between them
    assert alone[0][1] and outs[0][0] <= alone[0][0], (outs[0][0], alone[0][0])


@pytest.mark.parametrize("rows", [60, 5000])
def test_dd_cpr_pressure_stage_on_an_irregular_graph(pkg, orc, rows):
    """the pressure stage that spans the ranks on three subdomains of an unstructured graph with scattered ownership - every rank a
    neighbour of every other, rows with several couplings into one aggregate of another rank, rows of 2 to 9 blocks: every rank's
    application of the preconditioner = the oracle's bit for bit, same half-iteration count (+-1)"""
    world = 3
    g = irregular_global_case(pkg)
    Nb = g["Nb"]
    rng = np.random.default_rng(8)
    owner = ((np.arange(Nb) // 97 + rng.integers(0, 2, Nb) * (rng.random(Nb) < 0.03)) % world).astype(np.int32)
    o = oracle_bind.OracleModel(orc, g)
    o.set_state(g["pv"], g["meaning"])
    dt = 5 * 86400.0
    jo, ro = o.assemble(dt, 0)
    group = "q" + uuid.uuid4().hex
    parts = []
    for r in range(world):
        lp = pkg.ras.local_problem(g["rowptr"], g["col"], owner, r)
        cells = lp["cells"]
        parts.append(dict(Nb=lp["Nown"], Nghost=lp["Nghost"], Nloc=lp["Nown"] + lp["Nghost"], rowptr=lp["rows"], col=lp["cols"],
                          trans=np.ascontiguousarray(g["trans"][lp["entry"]]), area=np.ascontiguousarray(g["area"][lp["entry"]]),
                          poro=np.ascontiguousarray(g["poro"][cells]), volume=np.ascontiguousarray(g["volume"][cells]),
                          depth=np.ascontiguousarray(g["depth"][cells]), fluid=g["fluid"],
                          pv=np.ascontiguousarray(g["pv"].reshape(-1, 3)[cells].reshape(-1)), meaning=np.ascontiguousarray(g["meaning"][cells]),
                          gids=lp["gids"], halo=lp, global_cells=Nb))
    probe = np.random.default_rng(11).standard_normal(3 * Nb)

    def rank_fn(r):
        c = parts[r]
        m = pkg.capi.HipModel(c, comm=("loopback", world, r, group), reorder="line_coloring", preconditioner="cpr_quasiimpes", tolerance=1e-4, cpr_gather_rows=rows, cpr_amg_ilu_levels=0)
        m.set_state(c["pv"], c["meaning"])
        m.assemble(dt, 0)
        sol = m.solve_jacobian_system()
        d = np.ascontiguousarray(probe.reshape(-1, 3)[c["gids"][:c["Nb"]]].reshape(-1))
        return sol.it, sol.converged, m.cpr_apply(d), m.ordering(), m.cpr_levels()[0]

    outs = run_ranks(world, rank_fn)
    frg = np.concatenate([np.asarray(parts[r]["gids"][:parts[r]["Nb"]])[outs[r][3][1]] for r in range(world)]).astype(np.int32)
    tog = np.empty_like(frg)
    tog[frg] = np.arange(len(frg), dtype=np.int32)
    gr, gc, gv = orc.reorder_matrix(Nb, g["rowptr"], g["col"], jo, tog, frg)
    xo, reso, lev, glev, pv = orc.cpr_solve_blocks(Nb, gr, gc, gv, np.ascontiguousarray(ro.reshape(-1, 3)[frg].reshape(-1)), owner[frg], natural=frg, tol=1e-4,
                                                   gather_rows=rows, probe=np.ascontiguousarray(probe.reshape(-1, 3)[frg].reshape(-1)))
    pv = pv.reshape(-1, 3)[tog]
    assert reso.converged
    for r, (it, ok, v, _, levels) in enumerate(outs):
        c = parts[r]
        assert levels[-len(glev):] == glev
        assert np.array_equal(v[:3 * c["Nb"]].reshape(-1, 3), pv[c["gids"][:c["Nb"]]]), r
        assert ok and abs(it - reso.it) <= 1.0, (it, reso.it)


def test_dd_cpr_pressure_stage_rebuilt_with_the_structure(pkg, orc):
    """--cpr-reuse-setup=0 with the pressure stage that spans the ranks: every solve builds the ranks' hierarchies AND the joined level
    anew, all ranks together - after a second Newton iteration every rank's application is the oracle's built from the second matrix"""
    world, n, rows = 2, 8, 60
    g, owner, parts = global_and_parts(pkg, n, world, state="mixed", heterogeneous=True)
    src = pkg.decks.five_spot_source(g, rate_sm3_per_day=30.0)
    dt = 5 * 86400.0
    o = oracle_bind.OracleModel(orc, g)
    o.set_state(g["pv"], g["meaning"])
    o.set_source(src)
    probe = np.random.default_rng(9).standard_normal(3 * g["Nb"])
    group = "s" + uuid.uuid4().hex

    def rank_fn(r):
        c = parts[r]
        Nb = c["Nb"]
        m = pkg.capi.HipModel(c, comm=("loopback", world, r, group), reorder="line_coloring", preconditioner="cpr_quasiimpes", tolerance=1e-4, cpr_gather_rows=rows, cpr_reuse_setup=0, cpr_amg_ilu_levels=0)
        m.set_state(c["pv"], c["meaning"])
        m.set_source(np.ascontiguousarray(src.reshape(-1, 3)[c["gids"]].reshape(-1)))
        its = []
        for it in range(2):
            m.assemble(dt, it, fetch=False)
            sol = m.solve_jacobian_system()
            its.append((sol.it, sol.converged))
            if it == 0:
                m.update(None, 1.0)
        d = np.ascontiguousarray(probe.reshape(-1, 3)[c["gids"][:Nb]].reshape(-1))
        v = m.cpr_apply(d)
        j2, _ = m.assemble(dt, 1)   # the second Jacobian once more, fetched: what the oracle builds its structure from
        return its, v, m.ordering(), m.cpr_levels()[0], j2

    outs = run_ranks(world, rank_fn)
    assert all(ok for its, *_ in outs for _, ok in its)
    assert outs[0][0] == outs[1][0]                                   # the ranks agree on the iteration history
    # the global second Jacobian from the ranks' owned rows, then the oracle's preconditioner built from it
    j2g = np.zeros((len(g["col"]), 9))
    for r in range(world):
        j2g[parts[r]["halo"]["entry_global"]] = outs[r][4].reshape(-1, 9)
    j2g = np.ascontiguousarray(j2g.reshape(-1))
    frg = np.concatenate([np.asarray(parts[r]["gids"][:parts[r]["Nb"]])[outs[r][2][1]] for r in range(world)]).astype(np.int32)
    tog = np.empty_like(frg)
    tog[frg] = np.arange(len(frg), dtype=np.int32)
    gr, gc, gv = orc.reorder_matrix(g["Nb"], g["rowptr"], g["col"], j2g, tog, frg)
    b = np.random.default_rng(1).standard_normal(3 * g["Nb"])
    _, _, lev, glev, pv = orc.cpr_solve_blocks(g["Nb"], gr, gc, gv, b, np.asarray(owner)[frg], natural=frg, tol=1e-2, gather_rows=rows,
                                               probe=np.ascontiguousarray(probe.reshape(-1, 3)[frg].reshape(-1)))
    pv = pv.reshape(-1, 3)[tog]
    for r in range(world):
        c = parts[r]
        assert outs[r][3][-len(glev):] == glev
        assert np.array_equal(outs[r][1][:3 * c["Nb"]].reshape(-1, 3), pv[c["gids"][:c["Nb"]]]), r


@pytest.mark.parametrize("prec", ["ilu0", "cpr_quasiimpes"])
def test_dd_a_rank_without_neighbours(pkg, orc, prec):
    """three ranks, the third owns a part of the reservoir that touches no other (two grids in one system): it has no halo, its
    peers have - every exchange, all-reduce and (CPR) all-gather still counts it in.  The ranks agree on the iteration history and
    hold a solution of the whole system."""
    a = pkg.decks.cartesian_case(8, 6, 4, state="mixed", heterogeneous=True)
    b = pkg.decks.cartesian_case(5, 4, 3, state="mixed", heterogeneous=True)
    na = a["Nb"]
    g = dict(Nb=na + b["Nb"], fluid=a["fluid"],
             rowptr=np.concatenate([a["rowptr"], a["rowptr"][-1] + b["rowptr"][1:]]).astype(np.int32),
             col=np.concatenate([a["col"], b["col"] + na]).astype(np.int32))
    for k in ("trans", "area"):
        g[k] = np.concatenate([a[k], b[k]])
    for k in ("poro", "volume", "depth", "meaning"):
        g[k] = np.concatenate([a[k], b[k]])
    g["pv"] = np.concatenate([a["pv"], b["pv"]])
    owner = np.concatenate([(np.arange(na) >= na // 2).astype(np.int32), np.full(b["Nb"], 2, np.int32)])
    world = 3
    o = oracle_bind.OracleModel(orc, g)
    o.set_state(g["pv"], g["meaning"])
    dt = 5 * 86400.0
    jo, ro = o.assemble(dt, 0)
    group = "b" + uuid.uuid4().hex
    parts = []
    for r in range(world):
        lp = pkg.ras.local_problem(g["rowptr"], g["col"], owner, r)
        cells = lp["cells"]
        parts.append(dict(Nb=lp["Nown"], Nghost=lp["Nghost"], Nloc=lp["Nown"] + lp["Nghost"], rowptr=lp["rows"], col=lp["cols"],
                          trans=np.ascontiguousarray(g["trans"][lp["entry"]]), area=np.ascontiguousarray(g["area"][lp["entry"]]),
                          poro=np.ascontiguousarray(g["poro"][cells]), volume=np.ascontiguousarray(g["volume"][cells]),
                          depth=np.ascontiguousarray(g["depth"][cells]), fluid=g["fluid"],
                          pv=np.ascontiguousarray(g["pv"].reshape(-1, 3)[cells].reshape(-1)), meaning=np.ascontiguousarray(g["meaning"][cells]),
                          gids=lp["gids"], halo=lp, global_cells=g["Nb"]))
    assert parts[2]["Nghost"] == 0 and parts[0]["Nghost"] > 0

    def rank_fn(r):
        c = parts[r]
        m = pkg.capi.HipModel(c, comm=("loopback", world, r, group), reorder="line_coloring", preconditioner=prec, tolerance=1e-6, cpr_gather_rows=40, cpr_amg_ilu_levels=0)
        m.set_state(c["pv"], c["meaning"])
        m.assemble(dt, 0, fetch=False)
        sol = m.solve_jacobian_system()
        return sol.it, sol.converged, m.get_result()

    outs = run_ranks(world, rank_fn)
    assert all(o_ is not None for o_ in outs), "a rank never came back: the exchange lost count of the rank without neighbours"
    assert len({o_[0] for o_ in outs}) == 1 and all(o_[1] for o_ in outs)
    x = np.zeros((g["Nb"], 3))
    for r in range(world):
        c = parts[r]
        x[c["gids"][:c["Nb"]]] = outs[r][2].reshape(-1, 3)[:c["Nb"]]
    assert np.linalg.norm(orc.spmv(g["Nb"], g["rowptr"], g["col"], jo, x.reshape(-1)) - ro) < 1e-6 * np.linalg.norm(ro) * 1.001


@pytest.mark.parametrize("prec", ["ilu0", "cpr_quasiimpes"])
def test_dd_communication_scopes(pkg, prec):
    """The profiler's communication spans of a decomposed solve (opmhip_profile_get classes 9-11): one halo span per product - two per
    BiCGStab iteration, pack -> exchange -> ghosts in place, stamped on the halo stream it runs on (copyOwnerToAll in front of the operator,
    ParallelOverlappingILU0.hpp:897) - one all-reduce span per global reduction (local sums -> all-reduce), and with a CPR that spans the
    ranks three more halo spans and one gather span (the joined level's all-gather + its cycle) per application.  Counted against the
    solve's own iteration count; the times are positive and the halo spans do not exceed the wall time of the solve."""
    world, n = 2, 16
    group = "scopes" + uuid.uuid4().hex
    parts = [pkg.ras.cartesian_subdomain_case(n, world, r, state="mixed", heterogeneous=True) for r in range(world)]

    def rank_fn(r):
        import time
        c = parts[r]
        m = pkg.capi.HipModel(c, comm=("loopback", world, r, group), preconditioner=prec, tolerance=1e-6, cpr_amg_ilu_levels=0)
        m.set_state(c["pv"], c["meaning"])
        m.set_source(c["source"])
        m.assemble(86400.0, 0, fetch=False)
        m.profile_enable(True)
        t0 = time.perf_counter()
        sol = m.solve_jacobian_system()
        wall_ms = 1e3 * (time.perf_counter() - t0)
        return float(sol.it), bool(sol.converged), m.profile(), wall_ms

    outs = run_ranks(world, rank_fn)
    for it, ok, prof, wall_ms in outs:
        assert ok
        halves = int(round(2 * it))                       # products = preconditioner applications = half iterations run
        # a product = the interior tiles' launch (class "spmv"; none on a subdomain this thin) + the boundary tiles' behind the exchange
        products = prof["spmv_boundary"][0] or prof["spmv"][0]
        assert products == halves and prof["spmv"][0] in (0, halves)
        gathers = prof.get("cpr_gather", (0, 0.0))[0]
        if prec == "ilu0":
            assert gathers == 0 and prof["halo"][0] == products
        else:
            # (+ 1: the first solve's set-up of the joined level sends every ghost cell's aggregate number through the same exchange)
            assert gathers == halves and prof["halo"][0] == products + 3 * gathers + 1 and prof["cpr_gather"][1] > 0.0
        # one reduction behind the initial residual, two per half iteration (the scalar product, the norm)
        assert prof["allreduce"][0] == 1 + 2 * halves
        assert 0.0 < prof["halo"][1] < wall_ms and 0.0 < prof["allreduce"][1] < wall_ms


def wells_on_the_cut_grid(g, owner, rng, world, scale=3e-5):
    """Standard wells on the global grid: one horizontal well per pair of neighbouring subdomains crossing the cut between them (a few
    perforations on either side), and one vertical well inside every subdomain.  B, C: 4 x 3 blocks per perforation, D^-1 per well.
    scale: size of the entries of B and C - 3e-5 makes C^T D^-1 B x some 1e-4 of the residual on the test grids, so that a solve to 1e-6
    feels the wells (the callers assert that it does); larger random blocks cost BiCGStab its convergence (the ILU0 does not see them)."""
    nx, ny, nz = g["nx"], g["ny"], g["nz"]
    cell = lambda i, j, k: i + nx * (j + ny * k)   # noqa: E731
    lists = []
    j0, k0 = ny // 4, nz // 2
    lists.append([cell(i, j0, k0) for i in range(nx // 2 - 3, nx // 2 + 3)])              # along x, across the first cut
    if world >= 4:
        lists.append([cell(nx // 4, j, k0 + 1) for j in range(ny // 2 - 2, ny // 2 + 4)])  # along y, across the second cut
        lists.append([cell(nx // 2 - 1 + a, ny // 2 - 1 + b, nz - 2) for a, b in ((0, 0), (1, 0), (1, 1), (0, 1))])   # around the corner: four owners
    for r in range(world):   # a vertical well in the middle of every subdomain
        mine = np.flatnonzero(owner == r)
        c = int(mine[len(mine) // 2])
        i, j = c % nx, (c // nx) % ny
        col = [cell(i, j, k) for k in range(nz) if owner[cell(i, j, k)] == r][:4]
        lists.append(col)
    vp = np.cumsum([0] + [len(q) for q in lists]).astype(np.int32)
    cells = np.array([c for q in lists for c in q], np.int32)
    nw, n = len(lists), len(cells)
    D = np.empty((nw, 4, 4))
    for w in range(nw):
        D[w] = np.linalg.inv(0.2 * rng.standard_normal((4, 4)) + np.diag(2.0 + rng.random(4)))
    return dict(numWells=nw, val_pointers=vp, Ccols=cells, Bcols=cells.copy(), Cnnzs=np.ascontiguousarray(scale * rng.standard_normal(n * 12)),
                Bnnzs=np.ascontiguousarray(scale * rng.standard_normal(n * 12)), Dnnzs=np.ascontiguousarray(D.reshape(-1)))


def wells_of_rank(W, part, shared):
    """What rank `part` hands over.  shared: every well of the list with the perforations in cells the rank owns (possibly none);
    else only the wells that lie in the rank's subdomain as a whole."""
    local = {int(gid): k for k, gid in enumerate(part["gids"][:part["Nb"]])}
    vp, cells, keep, wells = [0], [], [], []
    for w in range(W["numWells"]):
        perfs = range(W["val_pointers"][w], W["val_pointers"][w + 1])
        mine = [p for p in perfs if int(W["Ccols"][p]) in local]
        if not shared and len(mine) != len(perfs):
            continue
        wells.append(w)
        keep += mine
        cells += [local[int(W["Ccols"][p])] for p in mine]
        vp.append(len(cells))
    keep = np.array(keep, np.int64)
    blk = lambda a, m: np.ascontiguousarray(a.reshape(-1, m)[keep].reshape(-1)) if len(keep) else np.zeros(0)   # noqa: E731
    return dict(numWells=len(wells), val_pointers=np.array(vp, np.int32), Ccols=np.array(cells, np.int32), Bcols=np.array(cells, np.int32),
                Cnnzs=blk(W["Cnnzs"], 12), Bnnzs=blk(W["Bnnzs"], 12), Dnnzs=np.ascontiguousarray(W["Dnnzs"].reshape(-1, 16)[wells].reshape(-1)),
                distributed=int(shared)), wells


@pytest.mark.parametrize("world,shared", [(2, True), (4, True), (8, True), (4, False)])
def test_dd_standard_wells(pkg, orc, world, shared):
    """Standard wells in a decomposed run.  shared: wells whose perforations lie in two or three subdomains (Flow with
    --allow-distributed-wells=true: ParallelStandardWellB sums B x over the ranks, wells/WellHelpers.hpp:68-123) - every rank hands over
    the whole list with its own perforations, opmhip_wells.distributed = 1.  Not shared: the default of the reference's partitioner, every
    well inside one subdomain, every rank hands over its own wells and nothing is exchanged for them.  Oracle: the global system with
    the global wells and the subdomains' block-Jacobi ILU0; r -= C^T D^-1 resWell bit for bit, the solve to the iteration count, the
    recovered well solutions identical on every rank."""
    n = 8
    g, owner, parts = global_and_parts(pkg, n, world, state="mixed", heterogeneous=True)
    rng = np.random.default_rng(100 + world)
    Wall = wells_on_the_cut_grid(g, owner, rng, world)
    if not shared:   # the global list without the wells that cross a cut
        inside = [w for w in range(Wall["numWells"]) if len(set(owner[Wall["Ccols"][Wall["val_pointers"][w]:Wall["val_pointers"][w + 1]]])) == 1]
        keep = np.concatenate([np.arange(Wall["val_pointers"][w], Wall["val_pointers"][w + 1]) for w in inside])
        vp = np.cumsum([0] + [Wall["val_pointers"][w + 1] - Wall["val_pointers"][w] for w in inside]).astype(np.int32)
        Wall = dict(numWells=len(inside), val_pointers=vp, Ccols=Wall["Ccols"][keep].copy(), Bcols=Wall["Bcols"][keep].copy(),
                    Cnnzs=np.ascontiguousarray(Wall["Cnnzs"].reshape(-1, 12)[keep].reshape(-1)), Bnnzs=np.ascontiguousarray(Wall["Bnnzs"].reshape(-1, 12)[keep].reshape(-1)),
                    Dnnzs=np.ascontiguousarray(Wall["Dnnzs"].reshape(-1, 16)[inside].reshape(-1)))
        assert len(inside) == world
    else:
        crossing = [w for w in range(Wall["numWells"]) if len(set(owner[Wall["Ccols"][Wall["val_pointers"][w]:Wall["val_pointers"][w + 1]]])) > 1]
        assert len(crossing) >= (1 if world == 2 else 3)
    res_well = 1e-3 * rng.standard_normal(4 * Wall["numWells"])
    src = pkg.decks.five_spot_source(g, rate_sm3_per_day=30.0)
    o = oracle_bind.OracleModel(orc, g)
    o.set_state(g["pv"], g["meaning"])
    o.set_source(src)
    dt = 86400.0
    jo, ro = o.assemble(dt, 0)
    r2 = orc.wells_apply_residual(Wall, res_well, ro)
    xo, reso = orc.solve(g["Nb"], g["rowptr"], g["col"], jo, r2, tol=1e-6, maxit=200, w=0.9, wells=Wall, owner=owner)
    xwo = orc.wells_recover(Wall, res_well, xo)
    assert reso.converged
    # the wells are felt: the oracle's solution leaves ten tolerances of residual when the operator forgets them
    assert np.linalg.norm(r2 - orc.spmv(g["Nb"], g["rowptr"], g["col"], jo, xo)) > 10 * 1e-6 * np.linalg.norm(r2)
    group = "w" + uuid.uuid4().hex

    def rank_fn(r):
        c = parts[r]
        Wr, ids = wells_of_rank(Wall, c, shared)
        m = pkg.capi.HipModel(c, comm=("loopback", world, r, group), reorder="level_scheduling", tolerance=1e-6, maxit=200)   # the oracle's ILU0 runs in the natural order
        m.set_state(c["pv"], c["meaning"])
        m.set_source(np.ascontiguousarray(src.reshape(-1, 3)[c["gids"]].reshape(-1)))
        m.assemble(dt, 0)
        rw = np.ascontiguousarray(res_well.reshape(-1, 4)[ids].reshape(-1))
        m.wells_apply_residual(Wr, rw)
        rhs = m.get_rhs()
        sol = m.solve_jacobian_system(wells=Wr)
        x = m.get_result()
        xw = m.wells_recover_solution(Wr, rw) if Wr["numWells"] else np.zeros(0)
        return rhs, sol.it, sol.converged, x, xw, ids

    outs = run_ranks(world, rank_fn)
    xg = np.zeros((g["Nb"], 3))   # the ranks' solutions put together
    for r, (rhs, it, ok, x, xw, ids) in enumerate(outs):
        c = parts[r]
        gi = c["gids"][:c["Nb"]]
        assert np.array_equal(rhs.reshape(-1, 3)[:c["Nb"]], r2.reshape(-1, 3)[gi])
        assert ok and it == reso.it
        xg[gi] = x.reshape(-1, 3)[:c["Nb"]]
    xg = xg.reshape(-1)
    # it solves the GLOBAL system with the GLOBAL wells to the tolerance (the wells are worth 40 to 100 tolerances here, see above) ...
    resid = r2 - orc.wells_apply(Wall, xg, orc.spmv(g["Nb"], g["rowptr"], g["col"], jo, xg))
    assert np.linalg.norm(resid) < 1e-6 * np.linalg.norm(r2) * 1.001
    # ... and is the oracle's solution as far as two solves that stop inside the same tolerance agree (the wells cost conditioning)
    np.testing.assert_allclose(xg, xo, rtol=2e-2, atol=1e-4 * np.abs(xo).max())
    # the recovered well solutions: D^-1 (resWell - B x) for the x the ranks hold, B x summed over them
    xw_expect = orc.wells_recover(Wall, res_well, xg).reshape(-1, 4)
    for r, (rhs, it, ok, x, xw, ids) in enumerate(outs):
        np.testing.assert_allclose(xw, xw_expect[ids].reshape(-1), rtol=1e-9, atol=1e-12 * np.abs(xw_expect).max())
        if shared:
            assert np.array_equal(xw, outs[0][4])   # one sum over the ranks: the same bits everywhere
    assert not np.array_equal(r2, ro)


@pytest.mark.parametrize("short", ["one_fewer", "none"])
def test_dd_shared_wells_need_the_same_list_on_every_rank(pkg, short):
    """distributed = 1 with lists of different lengths (one well fewer on rank 1; no well at all on rank 1): INVALID_ARGUMENT on every
    rank, nobody waits inside the reduction"""
    world, n = 2, 6
    g, owner, parts = global_and_parts(pkg, n, world, state="mixed", heterogeneous=False)
    Wall = wells_on_the_cut_grid(g, owner, np.random.default_rng(5), world)
    group = "v" + uuid.uuid4().hex

    def rank_fn(r):
        c = parts[r]
        Wr, _ = wells_of_rank(Wall, c, True)
        if r == 1 and short == "one_fewer":
            Wr = dict(Wr, numWells=Wr["numWells"] - 1, val_pointers=Wr["val_pointers"][:-1].copy())
        elif r == 1:
            Wr = dict(numWells=0, distributed=1)
        m = pkg.capi.HipModel(c, comm=("loopback", world, r, group))
        m.set_state(c["pv"], c["meaning"])
        m.assemble(86400.0, 0)
        with pytest.raises(pkg.capi.OpmHipError) as e:
            m.solve_jacobian_system(wells=Wr)
        return str(e.value)

    msgs = run_ranks(world, rank_fn)
    assert all("same wells" in s for s in msgs), msgs


def test_dd_an_empty_shared_list_is_agreed_on(pkg):
    """distributed = 1 and no well on any rank: the ranks agree that the list is empty and every entry point returns"""
    world, n = 2, 6
    g, owner, parts = global_and_parts(pkg, n, world, state="mixed", heterogeneous=False)
    group = "e" + uuid.uuid4().hex

    def rank_fn(r):
        c = parts[r]
        m = pkg.capi.HipModel(c, comm=("loopback", world, r, group))
        m.set_state(c["pv"], c["meaning"])
        m.assemble(86400.0, 0)
        W = dict(numWells=0, distributed=1)
        m.wells_apply_residual(W, np.zeros(0))
        sol = m.solve_jacobian_system(wells=W)
        return sol.converged

    assert all(run_ranks(world, rank_fn))


@pytest.mark.parametrize("prec", ["cpr", "cpr_quasiimpes"])
def test_dd_shared_wells_under_cpr(pkg, orc, prec):
    """Wells shared between subdomains with the CPR preconditioner of a decomposed run (the preconditioner does not see the wells, the
    Krylov operator does: WellModelMatrixAdapter, linalg/WellOperators.hpp:127-138): the solution the ranks return, put together,
    satisfies the GLOBAL system with the GLOBAL wells to the tolerance asked for, and every rank reports the same iteration count."""
    world, n = 4, 8
    g, owner, parts = global_and_parts(pkg, n, world, state="mixed", heterogeneous=True)
    rng = np.random.default_rng(77)
    Wall = wells_on_the_cut_grid(g, owner, rng, world)
    src = pkg.decks.five_spot_source(g, rate_sm3_per_day=30.0)
    o = oracle_bind.OracleModel(orc, g)
    o.set_state(g["pv"], g["meaning"])
    o.set_source(src)
    dt = 86400.0
    jo, ro = o.assemble(dt, 0)
    group = "c" + uuid.uuid4().hex
    tol = 1e-6

    def rank_fn(r):
        c = parts[r]
        Wr, _ = wells_of_rank(Wall, c, True)
        m = pkg.capi.HipModel(c, comm=("loopback", world, r, group), preconditioner=prec, tolerance=tol, maxit=200)
        m.set_state(c["pv"], c["meaning"])
        m.set_source(np.ascontiguousarray(src.reshape(-1, 3)[c["gids"]].reshape(-1)))
        m.assemble(dt, 0)
        sol = m.solve_jacobian_system(wells=Wr)
        return sol.it, sol.converged, m.get_result()

    outs = run_ranks(world, rank_fn)
    x = np.zeros((g["Nb"], 3))
    for r, (it, ok, xr) in enumerate(outs):
        assert ok and it == outs[0][0]
        x[parts[r]["gids"][:parts[r]["Nb"]]] = xr.reshape(-1, 3)[:parts[r]["Nb"]]
    x = x.reshape(-1)
    resid = ro - orc.wells_apply(Wall, x, orc.spmv(g["Nb"], g["rowptr"], g["col"], jo, x))
    assert np.linalg.norm(resid) < tol * np.linalg.norm(ro) * 1.001
    # the wells are felt at this tolerance: the same x leaves a much larger residual when the operator forgets them
    assert np.linalg.norm(ro - orc.spmv(g["Nb"], g["rowptr"], g["col"], jo, x)) > 10 * np.linalg.norm(resid)
