"""The benchmarked kernels at the benchmarked size, bit for bit against the oracle.

BASELINE.json configs[1] (100 x 100 x 100, 10^6 cells, 6.94 * 10^6 blocks) in the configuration bench.py runs - no ordering
flag: opmhip_default_config's `auto`, which must resolve to the line colouring with chains of 10 here - and once in the
reference's accelerator default (`graph_coloring`, Jones-Plassmann rounds, bda/Reorder.cpp:59-172).  What is timed there is what is
compared here: k_assemble over 111 112 tiles, k_spmv_pipe_st with ~16 tiles per workgroup, k_ilu_factor with its symbolic
elimination lists on 62 500 chains per colour, the stencil-form chain sweeps of M^-1, BiCGStab's half-iteration bookkeeping.

The oracle (oracle/, one thread) needs about 0.7 s for an assembly, 0.2 s for a factorisation and 0.2 s for a BiCGStab iteration at
this size (bench.py's cpu_baseline.single_thread), so the whole module costs it well under a minute.  Order of operations the oracle
follows: linalg/ParallelOverlappingILU0.hpp:439-494 (factorisation), :848-903 (application), bda/cusparseSolverBackend.cu:60-184
(BiCGStab's half steps)."""
import numpy as np
import pytest

from helpers import oracle_solve_in_order

pytestmark = pytest.mark.gpu
N = 100
DT = 86400.0       # bench.py's first time step


@pytest.fixture(scope="module", params=[None, "graph_coloring"], ids=["library_default", "graph_coloring"])
def big(request, pkg, orc, case100, oracle100):
    case, src = case100["case"], case100["src"]      # the session's one 100^3 case and its one oracle model (tests/conftest.py)
    assert oracle100.DT == DT
    kw = {} if request.param is None else {"reorder": request.param}
    m = pkg.capi.HipModel(case, tolerance=1e-2, maxit=200, ilu_relaxation=0.9, **kw)     # bench.py's solver settings
    m.set_state(case["pv"], case["meaning"])
    m.set_source(src)
    jac, res = m.assemble(DT, 0)
    o, jo, ro = oracle100.o, oracle100.jo, oracle100.ro
    to, fr, rpc = m.ordering()
    rr, rc, rv = orc.reorder_matrix(case["Nb"], case["rowptr"], case["col"], jo, to, fr)
    return dict(case=case, m=m, o=o, jac=jac, res=res, jo=jo, ro=ro, to=to, fr=fr, rpc=rpc, rr=rr, rc=rc, rv=rv, param=request.param, oracle100=oracle100)


def test_the_default_is_the_measured_configuration(big):
    """opmhip_default_config: reorder = auto -> line colouring, chains of 10, two colours at 10^6 structured rows (what bench.py
    reports in config.ilu_ordering without being told); an explicit graph_coloring stays the reference's Jones-Plassmann"""
    info = big["m"].ordering_info()
    if big["param"] is None:
        assert info["ilu_ordering"] == "line_coloring" and info["chain_length"] == 10 and info["colors"] == 2
    else:
        assert info["ilu_ordering"] == "graph_coloring" and info["chain_length"] == 0 and info["colors"] >= 2
    assert info["colors"] == len(big["rpc"]) and int(big["rpc"].sum()) == big["case"]["Nb"]
    # ... and, round 6, the half-product form of ILU0-BiCGStab where the ordering is the line colouring (a 7-point grid has no triangles:
    # U == upper(A)); the reference's Jones-Plassmann colouring keeps the plain form (its sweeps do not emit the row sums)
    form = big["m"].product_form()
    assert form["u_is_upper_a"] and form["half_product"] == (big["param"] is None)
    if big["param"] is None:
        assert form["rest_blocks"] == (len(big["case"]["col"]) + big["case"]["Nb"]) // 2     # lower entries + the diagonal


def test_jacobian_and_residual_bitwise(big):
    assert np.array_equal(big["res"], big["ro"])
    assert np.array_equal(big["jac"], big["jo"])


def test_spmv_bitwise(big, orc):
    m, case, to, fr = big["m"], big["case"], big["to"], big["fr"]
    Nb = case["Nb"]
    x = np.random.default_rng(41).standard_normal(3 * Nb)
    y = m.spmv(x)
    yo = orc.spmv(Nb, big["rr"], big["rc"], big["rv"], np.ascontiguousarray(x.reshape(Nb, 3)[fr].reshape(-1))).reshape(Nb, 3)[to].reshape(-1)
    assert np.array_equal(y, yo)


def test_ilu0_factors_and_application_bitwise(big, orc):
    m, case, to, fr = big["m"], big["case"], big["to"], big["fr"]
    Nb = case["Nb"]
    lu = m.ilu0_factor()
    lu_o = orc.ilu0_factor(Nb, big["rr"], big["rc"], big["rv"])
    assert np.array_equal(lu, lu_o)
    d = np.random.default_rng(42).standard_normal(3 * Nb)
    z = m.ilu0_apply(d)
    zo = orc.ilu0_apply(Nb, big["rr"], big["rc"], lu_o, np.ascontiguousarray(d.reshape(Nb, 3)[fr].reshape(-1)), w=0.9, mode="post_scale")
    assert np.array_equal(z, zo.reshape(Nb, 3)[to].reshape(-1))


def test_preconditioned_product_bitwise(big, orc):
    """the pair BiCGStab runs per half iteration - M^-1 d, then the product - in the form in force: with the library's default at this size
    the backward sweeps store their row sums and the product streams the matrix without its U part (k_spmv_pipe_st<.., UADD> over tiles of
    up to 64 rows, the rest stream written by k_ilu_factor), in the order oracle/linalg.hpp: ilu0_apply_u / spmv_rest state"""
    m, case, to, fr = big["m"], big["case"], big["to"], big["fr"]
    Nb = case["Nb"]
    hp = big["param"] is None
    m.ilu0_factor(want_factors=False)
    lu_o = orc.ilu0_factor(Nb, big["rr"], big["rc"], big["rv"])
    d = np.random.default_rng(43).standard_normal(3 * Nb)
    t, z = m.preconditioned_product(d)
    to_, zo = orc.preconditioned_product(Nb, big["rr"], big["rc"], big["rv"], lu_o, np.ascontiguousarray(d.reshape(Nb, 3)[fr].reshape(-1)), w=0.9,
                                         mode="post_scale", half_product=hp)
    assert np.array_equal(z, zo.reshape(Nb, 3)[to].reshape(-1))
    assert np.array_equal(t, to_.reshape(Nb, 3)[to].reshape(-1))


def test_solve_stops_on_the_oracles_half_iteration_and_the_next_iteration_follows(big, orc, request):
    """solveJacobianSystem on the assembled system: the oracle's half iteration, its reduction and its x (identical preconditioner and
    product bits; the scalar products are summed in another order); then updateSolution on both sides and a second assembly -
    Jacobian and residual of Newton iteration 1, storage term and switched cells included, again bit for bit"""
    m, o, case, to, fr = big["m"], big["o"], big["case"], big["to"], big["fr"]
    Nb = case["Nb"]
    request.addfinalizer(big["oracle100"].reset)   # this test moves the session's oracle model on: back to (initial state, assembled at iteration 0) afterwards, pass or fail
    m.assemble(DT, 0, fetch=False)            # the tests before this one may have left other factors / vectors behind
    sol = m.solve_jacobian_system()
    x = m.get_result()
    xo, so = oracle_solve_in_order(orc, Nb, case["rowptr"], case["col"], big["jo"], big["ro"], to, fr, tol=1e-2, maxit=200, w=0.9,
                                   half_product=big["param"] is None)
    assert sol.converged and so.converged and sol.it == so.it and sol.iterations == so.iterations
    assert abs(sol.reduction - so.reduction) <= 1e-8 * so.reduction
    np.testing.assert_allclose(x, xo, rtol=1e-8, atol=1e-11 * np.abs(xo).max())
    # the same update on both sides (the oracle's x: the two states stay comparable bit for bit)
    m.update(xo, 1.0)
    o.update(xo)
    pm, mm = m.get_state()
    po, mo = o.get_state()
    assert np.array_equal(mm, mo) and np.array_equal(pm, po)
    j1, r1 = m.assemble(DT, 1)
    j1o, r1o = o.assemble(DT, 1)
    assert np.array_equal(r1, r1o) and np.array_equal(j1, j1o)
    # and the convergence norms of that iteration (getReservoirConvergence): same cells, same maxima
    cm, co = m.convergence(DT, 1e-2), o.convergence(DT, 1e-2)
    assert np.array_equal(cm[3:6], co[3:6])                       # the maxima (CNV numerators) are exact
    np.testing.assert_allclose(cm[6:10], co[6:10], rtol=1e-12)    # sums over 10^6 cells: order of summation only
    np.testing.assert_allclose(cm[[0, 1, 2, 14, 15, 16]], co[[0, 1, 2, 14, 15, 16]], rtol=1e-9)


def test_cpr_at_the_benchmarked_size(pkg, orc, case100, oracle100):
    """bench.py's fastest configuration at its own size: `cpr` (= cpr_trueimpes, setupPropertyTree.cpp:62-76) on the 100^3 case with the
    library's defaults - line colouring, level 0 of the pressure AMG smoothed by ILU0 (asserted).  True-IMPES weights, the hierarchy's
    level sizes and one application of the preconditioner equal the oracle's bit for bit (the eight-level hierarchy over 10^6 rows with
    its lane-group levels and the Jacobi sweeps of the coarsest one: what the bench times, not a 9 000-row stand-in), and the solve -
    whose factorisation writes the pressure system as it stages the rows (k_ilu_factor's rider) - stops on the oracle's half iteration."""
    import oracle_bind
    case, src = case100["case"], case100["src"]
    Nb = case["Nb"]
    m = pkg.capi.HipModel(case, tolerance=1e-2, maxit=200, ilu_relaxation=0.9, preconditioner="cpr")
    m.set_state(case["pv"], case["meaning"])
    m.set_source(src)
    m.assemble(DT, 0, fetch=False)
    o, jo, ro = oracle100.o, oracle100.jo, oracle100.ro     # the session's oracle model, assembled at (DT, iteration 0)
    sol = m.solve_jacobian_system()
    info = m.ordering_info()
    assert info["ilu_ordering"] == "line_coloring" and info["chain_length"] == 10 and info["cpr_amg_ilu_levels"] == 1
    wo = o.true_impes_weights(DT)
    assert np.array_equal(m.cpr_weights(), wo)
    to, fr, _ = m.ordering()
    rr, rc, rv = orc.reorder_matrix(Nb, case["rowptr"], case["col"], jo, to, fr)
    cpr = oracle_bind.OracleCpr(orc)
    cpr.set_natural_ids(fr)
    cpr.set_weights(wo[fr])
    cpr.set_ilu_smoother(1, 1)
    cpr.update(Nb, rr, rc, rv)
    levels = m.cpr_levels()[0]
    assert levels == [int(x) for x in cpr.levels()[0]] and levels[0] == Nb and len(levels) >= 6
    d = np.random.default_rng(43).standard_normal(3 * Nb)
    vo = cpr.apply(np.ascontiguousarray(d.reshape(Nb, 3)[fr].reshape(-1))).reshape(Nb, 3)[to].reshape(-1)
    assert np.array_equal(m.cpr_apply(d), vo)
    xo, so = cpr.solve(Nb, rr, rc, rv, np.ascontiguousarray(ro.reshape(Nb, 3)[fr].reshape(-1)), tol=1e-2)
    assert sol.converged and so.converged and sol.it == so.it
    assert abs(sol.reduction - so.reduction) <= 1e-6 * so.reduction


@pytest.mark.parametrize("reorder", [None, "level_scheduling"], ids=["library_default", "level_scheduling"])
def test_two_subdomains_of_full_size_against_the_global_oracle(pkg, orc, reorder):
    """configs[3]'s building block with the oracle AT size: a 200 x 100 x 100 grid cut into two subdomains of 10^6 cells (+ 10^4 ghost cells
    each), two contexts on this GPU joined by the loopback communicator - set_pattern_dd, halo exchange beside the interior tiles, local
    reductions + all-reduce: the code path of the 8-GPU run with device copies in RCCL's place.  The oracle assembles the GLOBAL grid and
    solves it with the reference's block-Jacobi ILU0 (ghost_last_bilu0_decomposition: couplings between subdomains dropped,
    linalg/ParallelOverlappingILU0.hpp:439-494).  Every rank's owned rows of J and r equal the global oracle's bit for bit, the convergence
    sums are the global ones on both ranks; with level scheduling (the device's factors are then the oracle's natural-order ones) the solve
    stops on the oracle's half iteration with its x; with the library's default ordering (line colouring per subdomain) it meets the
    stopping rule on the global residual."""
    import threading
    import uuid
    import oracle_bind
    world = 2
    px, py, pz = pkg.ras.block_layout(world)
    g = pkg.decks.cartesian_case(px * N, py * N, pz * N, state="mixed", heterogeneous=False)
    owner = pkg.ras.cartesian_owner(px * N, py * N, pz * N, px, py, pz)
    parts = [pkg.ras.cartesian_subdomain_case(N, world, r, state="mixed", heterogeneous=False) for r in range(world)]
    src = pkg.decks.five_spot_source(g, rate_sm3_per_day=pkg.decks.BENCH_RATE_SM3_PER_DAY)
    o = oracle_bind.OracleModel(orc, g)
    o.set_state(g["pv"], g["meaning"])
    o.set_source(src)
    jo, ro = o.assemble(DT, 0)
    co = o.convergence(DT, 1e-2)
    group = "full2" + uuid.uuid4().hex
    out, err = [None] * world, [None] * world

    def body(r):
        try:
            c = parts[r]
            kw = {} if reorder is None else {"reorder": reorder}
            m = pkg.capi.HipModel(c, comm=("loopback", world, r, group), tolerance=1e-2, maxit=200, ilu_relaxation=0.9, **kw)
            m.set_state(c["pv"], c["meaning"])
            m.set_source(np.ascontiguousarray(src.reshape(-1, 3)[c["gids"]].reshape(-1)))
            j, res = m.assemble(DT, 0)
            conv = m.convergence(DT, 1e-2)
            sol = m.solve_jacobian_system()
            out[r] = (j, res, conv, float(sol.it), bool(sol.converged), float(sol.reduction), m.get_result(), m.ordering_info())
        except BaseException as e:  # noqa: BLE001
            err[r] = e

    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=900)
    for e in err:
        if e is not None:
            raise e
    xd = np.zeros((g["Nb"], 3))
    for r, (j, res, conv, it, ok, red, x, info) in enumerate(out):
        c = parts[r]
        Nb = c["Nb"]
        gi = c["gids"][:Nb]
        assert Nb == N ** 3 and c["Nghost"] == N * N
        assert np.array_equal(res.reshape(-1, 3)[:Nb], ro.reshape(-1, 3)[gi])
        assert np.array_equal(j.reshape(-1, 9), jo.reshape(-1, 9)[c["halo"]["entry_global"]])
        assert np.array_equal(conv[3:6], co[3:6])
        np.testing.assert_allclose(conv[6:10], co[6:10], rtol=1e-12)
        assert ok and red <= 1e-2
        xd[gi] = x.reshape(-1, 3)[:Nb]
        if reorder is None:
            assert info["ilu_ordering"] == "line_coloring" and info["chain_length"] == 10
    assert out[0][3] == out[1][3] and out[0][5] == out[1][5] and np.array_equal(out[0][2], out[1][2])   # the global scalars: the same bits on both ranks
    rg = ro - orc.spmv(g["Nb"], g["rowptr"], g["col"], jo, xd.reshape(-1))
    assert np.linalg.norm(rg) <= 1e-2 * np.linalg.norm(ro) * (1 + 1e-6)
    if reorder == "level_scheduling":
        xo, so = orc.solve(g["Nb"], g["rowptr"], g["col"], jo, ro, tol=1e-2, maxit=200, w=0.9, owner=owner)
        assert so.converged and out[0][3] == so.it
        np.testing.assert_allclose(xd.reshape(-1), xo, rtol=1e-7, atol=1e-11 * np.abs(xo).max())
