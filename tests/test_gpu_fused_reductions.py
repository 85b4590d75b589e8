"""BiCGStab with ONE reduction per half iteration (opmhip_config.fused_reductions; csrc/solver.hip: k_spmv_pipe_st<3, ..>, k_dots3, k_finalize3,
finalize_scalars3) through the C-ABI against the oracle's statement of the same arithmetic (oracle/linalg.hpp: bicgstab_fused_reductions;
its own checks: tests/test_oracle_half_product.py::test_fused_reductions_recurrence).

The reference runs four reductions per iteration (bda/cusparseSolverBackend.cu:92, 120-127, 151-161); this form derives |r| and rho from
three scalar products with the product's result per half iteration.  Off by default: it is meant for runs over several GPUs, where every
reduction is an all-reduce.  Checked here: the oracle's half iteration and solution on every path the three sums can take - riding in the
pipelined stencil kernel (plain and half-product form), formed by k_dots3 behind the product (small systems, orderings without the
stencil form, wells) -, the reported reduction being the TRUE residual's, the refusal of tolerances the recurred norm cannot resolve,
and two / four subdomains over the loopback communicator (one all-reduce of three doubles per half iteration)."""
import threading
import uuid

import numpy as np
import pytest

from helpers import laplace_block_system, oracle_solve_in_order

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("reorder,chain,wgs,hp", [("line_coloring", 8, 24, -1), ("line_coloring", 8, 24, 1), ("line_coloring", 4, 0, 1), ("graph_coloring_greedy", 0, 0, -1),
                                                  ("level_scheduling", 0, 0, -1), ("graph_coloring", 0, 8, -1)])
def test_solves_stop_on_the_oracles_half_iteration(pkg, orc, reorder, chain, wgs, hp):
    Nb, rp, ci, v = laplace_block_system(28, 35, 14, seed=12) if wgs else laplace_block_system(20, 16, 10, seed=6)
    b = np.random.default_rng(7).standard_normal(Nb * 3)
    kinds = set()
    for tol in (0.2, 0.05, 1e-2, 2e-3, 1e-4, 1e-6):
        s = pkg.capi.HipSolver(tolerance=tol, maxit=200, reorder=reorder, chain_length=chain, spmv_pipe_wgs=wgs, half_product=hp, fused_reductions=1)
        res = s.solve_system(Nb, rp, ci, v.copy(), b)
        assert s.product_form()["half_product"] == (hp > 0)        # (hp = -1: the plain form; 0 would let the library choose)
        x = s.get_result()
        to, fr, _ = s.ordering()
        xo, ro = oracle_solve_in_order(orc, Nb, rp, ci, v, b, to, fr, tol=tol, maxit=200, w=0.9, half_product=hp > 0, fused_reductions=True)
        assert res.converged and ro.converged and res.it == ro.it, (tol, res.it, ro.it)
        np.testing.assert_allclose(x, xo, rtol=1e-7, atol=1e-10 * np.abs(xo).max())
        # the reduction that is reported is the iterate's own residual, not the recurred one
        true = np.linalg.norm(b - orc.spmv(Nb, rp, ci, v, x)) / np.linalg.norm(b)
        assert abs(res.reduction - true) <= 1e-6 * true and true < 2.0 * tol
        # ... and the default recurrence stops within half an iteration of it
        s0 = pkg.capi.HipSolver(tolerance=tol, maxit=200, reorder=reorder, chain_length=chain, spmv_pipe_wgs=wgs, half_product=hp)
        r0 = s0.solve_system(Nb, rp, ci, v.copy(), b)
        assert abs(r0.it - res.it) <= 0.5
        kinds.add(res.it % 1.0)
    assert kinds == {0.0, 0.5}


def test_wells_and_a_second_solve(pkg, orc):
    Nb, rp, ci, v = laplace_block_system(16, 14, 9, seed=14)
    rng = np.random.default_rng(15)
    nw, perf = 3, 4
    cells = rng.choice(Nb, size=nw * perf, replace=False).astype(np.int32)
    W = dict(numWells=nw, val_pointers=np.arange(0, nw * perf + 1, perf, dtype=np.int32), Ccols=cells.copy(), Bcols=cells.copy(),
             Cnnzs=rng.uniform(-0.05, 0.05, nw * perf * 12), Bnnzs=rng.uniform(-0.05, 0.05, nw * perf * 12),
             Dnnzs=np.concatenate([(np.eye(4) + rng.uniform(-0.1, 0.1, (4, 4))).reshape(-1) for _ in range(nw)]))
    b = rng.standard_normal(3 * Nb)
    for hp in (-1, 1):
        s = pkg.capi.HipSolver(tolerance=1e-6, maxit=200, reorder="line_coloring", chain_length=4, half_product=hp, fused_reductions=1)
        res = s.solve_system(Nb, rp, ci, v.copy(), b, wells=W)
        to, fr, _ = s.ordering()
        xo, ro = oracle_solve_in_order(orc, Nb, rp, ci, v, b, to, fr, wells=W, tol=1e-6, maxit=200, w=0.9, half_product=hp > 0, fused_reductions=True)
        assert res.converged and res.it == ro.it
        np.testing.assert_allclose(s.get_result(), xo, rtol=1e-7, atol=1e-10 * np.abs(xo).max())
        v2 = v * 1.02
        res2 = s.solve_system(Nb, None, None, v2, b)         # the scalars of the last solve must not leak into the next
        xo2, ro2 = oracle_solve_in_order(orc, Nb, rp, ci, v2, b, to, fr, tol=1e-6, maxit=200, w=0.9, half_product=hp > 0, fused_reductions=True)
        assert res2.it == ro2.it
        np.testing.assert_allclose(s.get_result(), xo2, rtol=1e-7, atol=1e-10 * np.abs(xo2).max())


def test_tolerances_below_the_recurred_norms_floor_are_refused(pkg):
    """|r|^2 = r.r - 2 a v.r + a^2 v.v carries an absolute error of eps |r_0|^2: a tolerance under sqrt(eps) would never be met"""
    with pytest.raises(pkg.capi.OpmHipError) as e:
        pkg.capi.HipSolver(tolerance=1e-8, fused_reductions=1)
    assert e.value.code == pkg.capi.INVALID_ARGUMENT
    pkg.capi.HipSolver(tolerance=1e-8)           # the default recurrence takes it


@pytest.mark.parametrize("world", [2, 4])
def test_decomposed_run_with_one_all_reduce_per_half_iteration(pkg, orc, world):
    """2 / 4 subdomains over the loopback communicator: the three local sums travel through ONE all-reduce per half iteration (the profiler's
    all-reduce spans say so: 1 + halves + 1 per solve instead of 1 + 2 x halves), the ranks stop on the same half iteration as the oracle's
    global solve with block-Jacobi ILU0 in the same recurrence, within half an iteration of the default recurrence"""
    import oracle_bind
    from test_gpu_dd import global_and_parts, run_ranks
    n = 8
    g, owner, parts = global_and_parts(pkg, n, world, state="mixed", heterogeneous=True)
    src = pkg.decks.five_spot_source(g, rate_sm3_per_day=30.0)
    o = oracle_bind.OracleModel(orc, g)
    o.set_state(g["pv"], g["meaning"])
    o.set_source(src)
    dt = 86400.0
    jo, ro = o.assemble(dt, 0)
    xo, reso = orc.solve(g["Nb"], g["rowptr"], g["col"], jo, ro, tol=1e-4, maxit=200, w=0.9, owner=owner, fused_reductions=True)
    xp, resp = orc.solve(g["Nb"], g["rowptr"], g["col"], jo, ro, tol=1e-4, maxit=200, w=0.9, owner=owner)
    out = {}
    for fused in (1, 0):
        group = "f" + uuid.uuid4().hex

        def rank_fn(r, fused=fused, group=group):
            c = parts[r]
            m = pkg.capi.HipModel(c, comm=("loopback", world, r, group), reorder="level_scheduling", tolerance=1e-4, fused_reductions=fused)
            m.set_state(c["pv"], c["meaning"])
            m.set_source(np.ascontiguousarray(src.reshape(-1, 3)[c["gids"]].reshape(-1)))
            m.assemble(dt, 0, fetch=False)
            m.profile_enable(True)
            sol = m.solve_jacobian_system()
            spans = m.profile()["allreduce"][0]
            return sol.it, sol.converged, sol.reduction, m.get_result(), spans
        out[fused] = run_ranks(world, rank_fn)
    for r, (it, ok, red, x, spans) in enumerate(out[1]):
        c = parts[r]
        gi = c["gids"][:c["Nb"]]
        assert ok and it == reso.it and abs(it - resp.it) <= 0.5
        np.testing.assert_allclose(x.reshape(-1, 3)[:c["Nb"]], xo.reshape(-1, 3)[gi], rtol=1e-6, atol=1e-10 * np.abs(xo).max())
        assert abs(red - reso.reduction) <= 1e-5 * reso.reduction
        halves = int(round(2 * it))
        assert spans == 1 + halves + 1, (spans, halves)            # b.b, one per half iteration, the true norm at the end
        assert out[0][r][4] == 1 + 2 * int(round(2 * out[0][r][0]))   # the default recurrence: two per half iteration


def test_decomposed_run_in_the_benchs_configuration(pkg, orc):
    """what `bench.py --gpus N` runs for N > 1: line colouring, the pipelined kernels, the half-product form on the interior tiles AND one
    all-reduce per half iteration - two subdomains over the loopback communicator.  Against the same decomposition with the plain product
    (same recurrence): the same half iteration, the same solution to rounding; against the default recurrence: within half an iteration;
    1 + halves + 1 all-reduces per solve; the reported reduction is the true residual's"""
    import oracle_bind
    from test_gpu_dd import global_and_parts, run_ranks
    n, world = 20, 2      # (large enough for tiles away from the cut: the interior part of the product exists)
    g, owner, parts = global_and_parts(pkg, n, world, state="mixed", heterogeneous=True)
    src = pkg.decks.five_spot_source(g, rate_sm3_per_day=30.0)
    dt = 86400.0
    out = {}
    for hp, fused in ((1, 1), (-1, 1), (1, 0)):
        group = "b" + uuid.uuid4().hex

        def rank_fn(r, hp=hp, fused=fused, group=group):
            c = parts[r]
            m = pkg.capi.HipModel(c, comm=("loopback", world, r, group), reorder="line_coloring", chain_length=4, spmv_pipe_wgs=8, tolerance=1e-4,
                                  half_product=hp, fused_reductions=fused)
            m.set_state(c["pv"], c["meaning"])
            m.set_source(np.ascontiguousarray(src.reshape(-1, 3)[c["gids"]].reshape(-1)))
            m.assemble(dt, 0, fetch=False)
            form = m.product_form()
            m.profile_enable(True)
            sol = m.solve_jacobian_system()
            spans = m.profile()["allreduce"][0]
            return form, sol.it, sol.converged, sol.reduction, m.get_result(), spans
        out[(hp, fused)] = run_ranks(world, rank_fn)
    # the global system and the iterate's own residual
    o = oracle_bind.OracleModel(orc, g)
    o.set_state(g["pv"], g["meaning"])
    o.set_source(src)
    jo, ro = o.assemble(dt, 0)
    xg = np.zeros((g["Nb"], 3))
    for r in range(world):
        c = parts[r]
        xg[c["gids"][:c["Nb"]]] = out[(1, 1)][r][4].reshape(-1, 3)[:c["Nb"]]
    true = np.linalg.norm(ro - orc.spmv(g["Nb"], g["rowptr"], g["col"], jo, xg.reshape(-1))) / np.linalg.norm(ro)
    for r in range(world):
        f11, it11, ok11, red11, x11, sp11 = out[(1, 1)][r]
        f01, it01, ok01, red01, x01, sp01 = out[(-1, 1)][r]
        f10, it10, ok10, red10, x10, sp10 = out[(1, 0)][r]
        assert f11["half_product"] and f10["half_product"] and not f01["half_product"]
        assert ok11 and ok01 and ok10 and it11 == it01 and abs(it11 - it10) <= 0.5, (it11, it01, it10)
        nb = parts[r]["Nb"]
        np.testing.assert_allclose(x11.reshape(-1, 3)[:nb], x01.reshape(-1, 3)[:nb], rtol=1e-7, atol=1e-10 * np.abs(x01).max())
        assert sp11 == 1 + int(round(2 * it11)) + 1 and sp10 == 1 + 2 * int(round(2 * it10))
        assert abs(red11 - true) <= 1e-6 * true and true < 2e-4
