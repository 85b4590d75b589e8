"""The half-product form of ILU0-BiCGStab (opmhip_config.half_product; csrc/solver.hip: chain_sweep<.., UA>, k_spmv_pipe_st<.., UADD>,
k_ilu_factor's rest stream) through the C-ABI against the oracle's statement of the same order (oracle/linalg.hpp: ilu0_apply_u,
spmv_rest; held by the reference's data in tests/test_oracle_half_product.py).

On a pattern without triangles U == upper(A) bit for bit (linalg/ParallelOverlappingILU0.hpp:466-481 modifies A_ik only where (i,j), (j,k)
and (i,k) exist), so the backward sweep's row sums are the upper part of the product that follows it (bda/cusparseSolverBackend.cu:
103-118 runs the sweep and then the whole product); the device stores them and streams the matrix without its U part.  Checked here, on
systems small enough for the oracle: the pair (M^-1 d, A M^-1 d) bit for bit - line-coloured orderings of several chain lengths, one tile
per workgroup and the pipelined form with ragged last steps, both relaxation modes -, whole solves (the oracle's half iteration, x to the
rounding of the scalar products' order), standard wells in operator form on top, matrices whose values change between solves, and that a
pattern WITH triangles keeps the plain form.  tests/test_gpu_fullsize_oracle.py repeats the bit-for-bit part at 10^6 cells, where the
form is the library's default."""
import numpy as np
import pytest

from helpers import laplace_block_system, oracle_solve_in_order, random_block_system

pytestmark = pytest.mark.gpu


def _hp_solver(pkg, **kw):
    kw.setdefault("reorder", "line_coloring")
    kw.setdefault("half_product", 1)
    return pkg.capi.HipSolver(**kw)


@pytest.mark.parametrize("shape,chain,wgs", [((6, 5, 4), 0, 0), ((13, 11, 7), 4, 0), ((17, 9, 5), 3, 8), ((28, 35, 14), 8, 24),
                                             ((28, 35, 14), 10, 8), ((40, 30, 20), 10, 200), ((9, 1, 1), 0, 0), ((1, 1, 30), 10, 0)])
@pytest.mark.parametrize("mode,w", [("post_scale", 0.9), ("in_sweep", 0.9), ("post_scale", 1.0)])
def test_preconditioned_product_bit_exact(pkg, orc, shape, chain, wgs, mode, w):
    Nb, rp, ci, v = laplace_block_system(*shape, seed=31)
    s = _hp_solver(pkg, chain_length=chain, spmv_pipe_wgs=wgs, ilu_relaxation=w, relax_mode=mode)
    s.set_pattern(Nb, rp, ci)
    form = s.product_form()
    assert form["u_is_upper_a"] and form["half_product"] and 0 < form["rest_blocks"] < len(ci)
    s.upload_system(v)
    lu = s.ilu0_factor()
    to, fr, _ = s.ordering()
    rr, rc, rv = orc.reorder_matrix(Nb, rp, ci, v, to, fr)
    lu_o = orc.ilu0_factor(Nb, rr, rc, rv)
    assert np.array_equal(lu, lu_o)       # the factorisation that also writes the rest stream leaves the same factors
    rng = np.random.default_rng(32)
    for _ in range(2):
        d = rng.standard_normal(3 * Nb)
        t, z = s.preconditioned_product(d)
        to_, zo = orc.preconditioned_product(Nb, rr, rc, rv, lu_o, np.ascontiguousarray(d.reshape(Nb, 3)[fr].reshape(-1)), w=w, mode=mode, half_product=True)
        assert np.array_equal(z, zo.reshape(Nb, 3)[to].reshape(-1))
        assert np.array_equal(t, to_.reshape(Nb, 3)[to].reshape(-1))
        # and it is the product: the plain form's bits up to the order of a row's additions
        tp, _ = orc.preconditioned_product(Nb, rr, rc, rv, lu_o, np.ascontiguousarray(d.reshape(Nb, 3)[fr].reshape(-1)), w=w, mode=mode)
        scale = orc.spmv(Nb, rr, rc, np.abs(rv), np.abs(zo))
        assert np.all(np.abs(to_ - tp) <= 4e-15 * scale)
    # the plain entry points are untouched by the form: the whole product and M^-1 alone, bit for bit as before
    x = rng.standard_normal(3 * Nb)
    assert np.array_equal(s.spmv(x), orc.spmv(Nb, rr, rc, rv, np.ascontiguousarray(x.reshape(Nb, 3)[fr].reshape(-1))).reshape(Nb, 3)[to].reshape(-1))
    zo = orc.ilu0_apply(Nb, rr, rc, lu_o, np.ascontiguousarray(x.reshape(Nb, 3)[fr].reshape(-1)), w=w, mode=mode)
    assert np.array_equal(s.ilu0_apply(x), zo.reshape(Nb, 3)[to].reshape(-1))


@pytest.mark.parametrize("shape,chain,wgs", [((24, 20, 12), 8, 0), ((28, 35, 14), 10, 24), ((20, 16, 10), 4, 8)])
def test_solves_stop_on_the_oracles_half_iteration(pkg, orc, shape, chain, wgs):
    """whole solves over a ladder of tolerances (both kinds of exit, first and second half): the oracle's half iteration in the same form,
    x to the rounding the scalar products' summation order leaves; a second solve with other values on the same context (the rest stream
    is rewritten by every factorisation)"""
    Nb, rp, ci, v = laplace_block_system(*shape, seed=4)
    b = np.random.default_rng(9).standard_normal(Nb * 3)
    kinds = set()
    for tol in (0.2, 0.05, 1e-2, 2e-3, 1e-4, 1e-6, 1e-8):
        s = _hp_solver(pkg, tolerance=tol, maxit=200, chain_length=chain, spmv_pipe_wgs=wgs)
        res = s.solve_system(Nb, rp, ci, v.copy(), b)
        assert s.product_form()["half_product"]
        x = s.get_result()
        to, fr, _ = s.ordering()
        xo, ro = oracle_solve_in_order(orc, Nb, rp, ci, v, b, to, fr, tol=tol, maxit=200, w=0.9, half_product=True)
        assert res.converged and ro.converged and res.it == ro.it and res.iterations == ro.iterations
        np.testing.assert_allclose(x, xo, rtol=1e-8, atol=1e-11 * np.abs(xo).max())
        assert abs(res.reduction - ro.reduction) <= 1e-8 * ro.reduction
        r = b - orc.spmv(Nb, rp, ci, v, x)
        assert np.linalg.norm(r) < tol * np.linalg.norm(b) * (1 + 1e-9)      # the TRUE residual: the recurrence's norm is honest
        kinds.add(res.it % 1.0)
        v2 = v * 1.01
        res2 = s.solve_system(Nb, None, None, v2, b)
        xo2, ro2 = oracle_solve_in_order(orc, Nb, rp, ci, v2, b, to, fr, tol=tol, maxit=200, w=0.9, half_product=True)
        assert res2.it == ro2.it
        np.testing.assert_allclose(s.get_result(), xo2, rtol=1e-8, atol=1e-11 * np.abs(xo2).max())
    assert kinds == {0.0, 0.5}


def test_standard_wells_on_top_of_the_form(pkg, orc):
    """y -= C^T D^-1 B x after the product (bda/WellContributions.cu:36-126) with the half-product form underneath: the well operator sees
    the same x, the scalar products move to k_dots as with the plain form"""
    Nb, rp, ci, v = laplace_block_system(16, 14, 9, seed=14)
    rng = np.random.default_rng(15)
    nw, perf = 3, 4
    cells = rng.choice(Nb, size=nw * perf, replace=False).astype(np.int32)
    W = dict(numWells=nw, val_pointers=np.arange(0, nw * perf + 1, perf, dtype=np.int32), Ccols=cells.copy(), Bcols=cells.copy(),
             Cnnzs=rng.uniform(-0.05, 0.05, nw * perf * 12), Bnnzs=rng.uniform(-0.05, 0.05, nw * perf * 12),
             Dnnzs=np.concatenate([(np.eye(4) + rng.uniform(-0.1, 0.1, (4, 4))).reshape(-1) for _ in range(nw)]))
    b = rng.standard_normal(3 * Nb)
    s = _hp_solver(pkg, tolerance=1e-6, maxit=200, chain_length=4)
    res = s.solve_system(Nb, rp, ci, v.copy(), b, wells=W)
    to, fr, _ = s.ordering()
    xo, ro = oracle_solve_in_order(orc, Nb, rp, ci, v, b, to, fr, wells=W, tol=1e-6, maxit=200, w=0.9, half_product=True)
    assert s.product_form()["half_product"] and res.converged and res.it == ro.it
    np.testing.assert_allclose(s.get_result(), xo, rtol=1e-8, atol=1e-11 * np.abs(xo).max())


def test_assembled_jacobians_and_newton_steps(pkg, orc):
    """the device-resident Newton iteration with the form forced on at a size the oracle handles: assembly -> solve (device Jacobian, the
    zero-diagonal fix applied as the rows are staged) -> update, against the oracle running the same form: iteration counts, final state"""
    import oracle_bind
    for state in ("mixed", "saturated"):
        case = pkg.decks.cartesian_case(12, 10, 8, state=state, heterogeneous=True)
        src = pkg.decks.five_spot_source(case, rate_sm3_per_day=80.0)
        m = pkg.capi.HipModel(case, reorder="line_coloring", chain_length=4, half_product=1)
        o = oracle_bind.OracleModel(orc, case)
        for h in (m, o):
            h.set_state(case["pv"], case["meaning"])
            h.set_source(src)
        assert m.product_form()["half_product"]
        dt = 5 * 86400.0
        for it in range(3):
            j, r = m.assemble(dt, it)
            jo, ro = o.assemble(dt, it)
            assert np.array_equal(j, jo) and np.array_equal(r, ro)
            res = m.solve_jacobian_system()
            xo, reso = o.solve_in_order(*m.ordering()[:2], tol=1e-2, maxit=200, w=0.9, half_product=True)
            assert res.converged and res.it == reso.it
            np.testing.assert_allclose(m.get_result(), xo, rtol=1e-7, atol=1e-10 * np.abs(xo).max())
            m.update(xo, 1.0)
            o.update(xo)
            pm, mm = m.get_state()
            po, mo = o.get_state()
            assert np.array_equal(mm, mo) and np.array_equal(pm, po)


def test_patterns_with_triangles_keep_the_plain_form(pkg, orc):
    """a random graph (triangles: elimination steps reach right of the diagonal) and a grid with well cliques in its pattern: the property
    fails, the library says so and runs the plain form - its solves are the plain oracle's"""
    Nb, rp, ci, v = random_block_system(400, "random", seed=3, extra=3)
    b = np.random.default_rng(5).standard_normal(3 * Nb)
    for reorder in ("line_coloring", "graph_coloring_greedy"):
        s = pkg.capi.HipSolver(reorder=reorder, half_product=1, tolerance=1e-8)
        res = s.solve_system(Nb, rp, ci, v.copy(), b)
        form = s.product_form()
        assert not form["u_is_upper_a"] and not form["half_product"]
        xo, ro = oracle_solve_in_order(orc, Nb, rp, ci, v, b, *s.ordering()[:2], tol=1e-8, maxit=200, w=0.9)
        assert res.it == ro.it
        np.testing.assert_allclose(s.get_result(), xo, rtol=1e-8, atol=1e-12)
    # a colouring without chains has the property but not the sweeps that emit the row sums: plain form
    Nb, rp, ci, v = laplace_block_system(10, 9, 8, seed=2)
    s = pkg.capi.HipSolver(reorder="graph_coloring_greedy", half_product=1)
    s.set_pattern(Nb, rp, ci)
    form = s.product_form()
    assert form["u_is_upper_a"] and not form["half_product"]
    # asked never to: plain form on a pattern that would allow it
    s = pkg.capi.HipSolver(reorder="line_coloring", half_product=-1)
    s.set_pattern(Nb, rp, ci)
    assert not s.product_form()["half_product"]
    # the library's choice (0) on a system this small: plain form (the choice follows the pipelined kernels' size)
    s = pkg.capi.HipSolver(reorder="line_coloring")
    s.set_pattern(Nb, rp, ci)
    assert not s.product_form()["half_product"]


def test_cpr_ignores_the_form(pkg, orc):
    """with a CPR preconditioner the product follows the two-level application, not a backward sweep: the switch is ignored"""
    Nb, rp, ci, v = laplace_block_system(12, 10, 8, seed=6)
    s = pkg.capi.HipSolver(reorder="line_coloring", half_product=1, preconditioner="cpr_quasiimpes")
    s.set_pattern(Nb, rp, ci)
    assert not s.product_form()["half_product"]


@pytest.mark.parametrize("world,n,wgs", [(2, 20, 8), (4, 16, 8), (8, 8, 8), (2, 28, 0)])
def test_subdomains_interior_tiles_take_the_form(pkg, orc, world, n, wgs):
    """Decomposed runs (loopback communicator, the code path of the RCCL run): the interior tiles of a subdomain - no ghost column in any of
    their rows - take the half-product form, the boundary tiles keep the whole product, the halo exchange runs beside the interior launch as
    before.  Every row's sum is the plain form's up to the order of its additions: against the same decomposition with the form switched
    off the preconditioned product agrees to rounding on every rank, M^-1 d is identical, the solves stop on the same half
    iteration with the same solution (1e-8); the global oracle's block-Jacobi solve (natural order: other iterates) agrees to the tolerance"""
    import uuid
    import oracle_bind
    from test_gpu_dd import global_and_parts, run_ranks
    g, owner, parts = global_and_parts(pkg, n, world, state="mixed", heterogeneous=True)
    src = pkg.decks.five_spot_source(g, rate_sm3_per_day=30.0)
    o = oracle_bind.OracleModel(orc, g)
    o.set_state(g["pv"], g["meaning"])
    o.set_source(src)
    dt = 86400.0
    jo, ro = o.assemble(dt, 0)
    xo, reso = orc.solve(g["Nb"], g["rowptr"], g["col"], jo, ro, tol=1e-4, maxit=200, w=0.9, owner=owner)
    rng = np.random.default_rng(77)
    dglob = rng.standard_normal(3 * g["Nb"])
    out = {}
    for hp in (1, -1):
        group = "h" + uuid.uuid4().hex

        def rank_fn(r, hp=hp, group=group):
            c = parts[r]
            m = pkg.capi.HipModel(c, comm=("loopback", world, r, group), reorder="line_coloring", chain_length=4, spmv_pipe_wgs=wgs, tolerance=1e-4, half_product=hp)
            m.set_state(c["pv"], c["meaning"])
            m.set_source(np.ascontiguousarray(src.reshape(-1, 3)[c["gids"]].reshape(-1)))
            m.assemble(dt, 0, fetch=False)
            form = m.product_form()
            m.ilu0_factor(want_factors=False)
            d = np.ascontiguousarray(dglob.reshape(-1, 3)[c["gids"][:c["Nb"]]].reshape(-1))
            t, z = m.preconditioned_product(d)       # collective: the halo exchange of M^-1 d in front of the product
            sol = m.solve_jacobian_system()
            return form, t, z, sol.it, sol.converged, m.get_result()
        out[hp] = run_ranks(world, rank_fn)
    for r in range(world):
        c = parts[r]
        gi = c["gids"][:c["Nb"]]
        (f1, t1, z1, it1, ok1, x1), (f0, t0, z0, it0, ok0, x0) = out[1][r], out[-1][r]
        # a subdomain so small that every tile of 32 chains touches a cut has no interior tile: the form is then off, whatever was asked (8 x 8^3)
        assert f1["u_is_upper_a"] and not f0["half_product"] and f1["half_product"] == (f1["rest_positions"] > 0)
        if world == 2:
            assert f1["half_product"], f1     # one cut: tiles away from it exist (with cuts in x AND y every tile of 32 chains of these small boxes meets one)
        if f1["half_product"]:
            # what the product streams: interior rows without their U part, boundary rows whole - between the rest of every row and the whole matrix
            nloc, rest_all = len(c["col"]), sum(1 for i in range(c["Nb"]) for k in range(c["rowptr"][i], c["rowptr"][i + 1]) if c["col"][k] >= c["Nb"])
            assert (nloc - rest_all + c["Nb"]) // 2 + rest_all < f1["rest_blocks"] < nloc
        assert np.array_equal(z1, z0)                                    # the sweeps' statements are untouched
        assert np.all(np.abs(t1 - t0) <= 1e-12 * np.abs(t0).max())       # the same products, another order of a row's additions
        assert np.array_equal(t1, t0) == (not f1["half_product"])        # (interior rows DID take another order where the form is on)
        assert ok1 and ok0 and it1 == it0
        np.testing.assert_allclose(x1, x0, rtol=1e-8, atol=1e-11 * np.abs(x0).max())
    # the ranks' solutions put together solve the GLOBAL system to the tolerance (the global oracle's own iterate - block-Jacobi ILU0 in the
    # natural order - is another one of the same quality)
    xg = np.zeros_like(xo)
    for r in range(world):
        c = parts[r]
        xg.reshape(-1, 3)[c["gids"][:c["Nb"]]] = out[1][r][5].reshape(-1, 3)[:c["Nb"]]
    rg = ro - orc.spmv(g["Nb"], g["rowptr"], g["col"], jo, xg)
    assert np.linalg.norm(rg) <= 1e-4 * np.linalg.norm(ro) * (1 + 1e-6) and reso.converged
