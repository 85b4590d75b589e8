"""GPU parity of the linear-solve half of the hot path, through the C-ABI (libopmhip.so), against the CPU oracle
and the reference's own fixtures.  Reads like tests/test_cusparseSolver.cpp: load matr33/rhs3, solve_system,
get_result, compare."""
import json
import os

import numpy as np
import pytest

from helpers import laplace_block_system, oracle_solve_in_order, random_block_system

pytestmark = pytest.mark.gpu

REORDERS = ["level_scheduling", "graph_coloring", "graph_coloring_greedy", "line_coloring"]


def _load(pkg, golden, mat, rhs):
    Nb, rp, ci, v, bs = pkg.mmio.read_block_matrix(os.path.join(golden, "linalg", mat))
    b = pkg.mmio.read_block_vector(os.path.join(golden, "linalg", rhs))
    return Nb, rp, ci, v, b


@pytest.mark.parametrize("reorder", REORDERS)
def test_matr33_like_test_cusparseSolver(pkg, orc, golden, reorder):
    """tests/test_cusparseSolver.cpp:49-131 harness: tol 0.5, maxit 20 from options_flexiblesolver.json."""
    Nb, rp, ci, v, b = _load(pkg, golden, "matr33.txt", "rhs3.txt")
    s = pkg.capi.HipSolver(verbosity=0, maxit=20, tolerance=0.5, ilu_relaxation=1.0, reorder=reorder)
    res = s.solve_system(Nb, rp, ci, v.copy(), b)
    x = s.get_result()
    assert res.converged
    xo, ro = oracle_solve_in_order(orc, Nb, rp, ci, v, b, *s.ordering()[:2], tol=0.5, maxit=20, w=1.0)
    assert res.it == ro.it
    np.testing.assert_allclose(x, xo, rtol=1e-9)
    # exact solution pinned by tests/test_flexiblesolver.cpp:114-116 (ILU0 is exact on this block-tridiagonal matrix)
    with open(os.path.join(golden, "linalg", "expected.json")) as f:
        e = json.load(f)["exact_noprec_tol1e-12_maxit200"]
    if reorder == "level_scheduling":
        np.testing.assert_allclose(x, e["x"], rtol=2e-5)


@pytest.mark.parametrize("reorder", REORDERS)
@pytest.mark.parametrize("shape", [(6, 5, 4), (17, 9, 5), (40, 30, 20)])
def test_spmv_bit_exact(pkg, orc, reorder, shape):
    Nb, rp, ci, v = laplace_block_system(*shape, seed=11)
    x = np.random.default_rng(5).standard_normal(Nb * 3)
    s = pkg.capi.HipSolver(reorder=reorder)
    s.set_pattern(Nb, rp, ci)
    s.upload_system(v)
    y = s.spmv(x)
    # same matrix in the device's internal order through the oracle: identical operation order -> identical bits
    to, fr, rpc = s.ordering()
    rr, rc, rv = orc.reorder_matrix(Nb, rp, ci, v, to, fr)
    xi = x.reshape(Nb, 3)[fr].reshape(-1)
    yo = orc.spmv(Nb, rr, rc, rv, xi).reshape(Nb, 3)[to].reshape(-1)
    assert np.array_equal(y, yo)
    np.testing.assert_allclose(y, orc.spmv(Nb, rp, ci, v, x), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("reorder", ["graph_coloring", "line_coloring"])
@pytest.mark.parametrize("wgs", [8, 24, 200, -1])
def test_pipelined_spmv_bit_exact(pkg, orc, reorder, wgs):
    """The pipelined SpMV (every workgroup walks through several tiles with the loads of the next three tiles in flight) on
    grids small enough for the oracle: sized for 8 / 24 / 200 resident workgroups a 13 720-row system takes 54 / 18 / 3
    pipeline steps per workgroup (ragged last step included); -1 = the one-tile-per-workgroup kernel.  Same bits as the
    oracle, and the fused partial dots (BiCGStab's scalars) give the same solve."""
    Nb, rp, ci, v = laplace_block_system(28, 35, 14, seed=12)
    x = np.random.default_rng(6).standard_normal(Nb * 3)
    s = pkg.capi.HipSolver(reorder=reorder, spmv_pipe_wgs=wgs, tolerance=1e-8)
    s.set_pattern(Nb, rp, ci)
    s.upload_system(v)
    to, fr, rpc = s.ordering()
    rr, rc, rv = orc.reorder_matrix(Nb, rp, ci, v, to, fr)
    yo = orc.spmv(Nb, rr, rc, rv, x.reshape(Nb, 3)[fr].reshape(-1)).reshape(Nb, 3)[to].reshape(-1)
    assert np.array_equal(s.spmv(x), yo)
    b = np.random.default_rng(7).standard_normal(Nb * 3)
    res = s.solve_system(Nb, rp, ci, v.copy(), b)
    xo, reso = oracle_solve_in_order(orc, Nb, rp, ci, v, b, to, fr, tol=1e-8, maxit=200, w=0.9)
    assert res.converged and res.it == reso.it
    np.testing.assert_allclose(s.get_result(), xo, rtol=1e-7, atol=1e-10)


@pytest.mark.parametrize("reorder", REORDERS)
@pytest.mark.parametrize("mode,w", [("post_scale", 0.9), ("in_sweep", 0.9), ("post_scale", 1.0)])
def test_ilu0_factor_and_apply_bit_exact(pkg, orc, reorder, mode, w):
    Nb, rp, ci, v = laplace_block_system(13, 11, 7, seed=21)
    s = pkg.capi.HipSolver(reorder=reorder, ilu_relaxation=w, relax_mode=mode)
    s.set_pattern(Nb, rp, ci)
    s.upload_system(v)
    lu = s.ilu0_factor()
    to, fr, rpc = s.ordering()
    rr, rc, rv = orc.reorder_matrix(Nb, rp, ci, v, to, fr)
    lu_o = orc.ilu0_factor(Nb, rr, rc, rv)
    assert np.array_equal(lu, lu_o)
    d = np.random.default_rng(8).standard_normal(Nb * 3)
    z = s.ilu0_apply(d)
    zo = orc.ilu0_apply(Nb, rr, rc, lu_o, d.reshape(Nb, 3)[fr].reshape(-1), w=w, mode=mode)
    assert np.array_equal(z, zo.reshape(Nb, 3)[to].reshape(-1))


def test_level_scheduling_equals_cpu_natural_order_ilu(pkg, orc):
    """The level-scheduled device factorisation is the CPU's natural-order ILU0 (same factors, bit for bit on a
    Cartesian pattern where levels keep each row's lower entries in natural order)."""
    Nb, rp, ci, v = laplace_block_system(9, 8, 7, seed=2)
    s = pkg.capi.HipSolver(reorder="level_scheduling", ilu_relaxation=0.9)
    s.set_pattern(Nb, rp, ci)
    s.upload_system(v)
    s.ilu0_factor(want_factors=False)
    d = np.random.default_rng(3).standard_normal(Nb * 3)
    z = s.ilu0_apply(d)
    lu = orc.ilu0_factor(Nb, rp, ci, v)
    zo = orc.ilu0_apply(Nb, rp, ci, lu, d, w=0.9, mode="post_scale")
    assert np.array_equal(z, zo)


@pytest.mark.parametrize("reorder", REORDERS)
def test_solve_matches_oracle(pkg, orc, reorder):
    Nb, rp, ci, v = laplace_block_system(24, 20, 12, seed=4)
    b = np.random.default_rng(9).standard_normal(Nb * 3)
    s = pkg.capi.HipSolver(tolerance=1e-2, maxit=200, reorder=reorder)
    res = s.solve_system(Nb, rp, ci, v.copy(), b)
    x = s.get_result()
    to, fr, rpc = s.ordering()
    xo, ro = oracle_solve_in_order(orc, Nb, rp, ci, v, b, to, fr, tol=1e-2, maxit=200, w=0.9)
    assert res.converged and ro.converged
    assert res.it == ro.it and res.iterations == ro.iterations
    # identical preconditioner and SpMV bits; only the dot products are summed in a different order
    np.testing.assert_allclose(x, xo, rtol=1e-9, atol=1e-12)
    assert abs(res.reduction - ro.reduction) <= 1e-9 * ro.reduction
    # and it is a solution
    r = b - orc.spmv(Nb, rp, ci, v, x)
    assert np.linalg.norm(r) < 1e-2 * np.linalg.norm(b) * (1 + 1e-9)
    # second solve on the same context (pattern reuse, "initialized == true" path) with new values
    v2 = v * 1.01
    res2 = s.solve_system(Nb, None, None, v2, b)
    x2 = s.get_result()
    xo2, _ = oracle_solve_in_order(orc, Nb, rp, ci, v2, b, to, fr, tol=1e-2, maxit=200, w=0.9)
    np.testing.assert_allclose(x2, xo2, rtol=1e-9, atol=1e-12)


def test_solves_that_stop_on_either_half_iteration(pkg, orc):
    """The first half's update of x (x += alpha y, bda/cusparseSolverBackend.cu:110) is carried out by the second half's
    kernel, or by k_bicg_xhalf when the stopping rule is met right after a first half: over a ladder of tolerances both
    kinds of exit occur and every solution is the oracle's."""
    Nb, rp, ci, v = laplace_block_system(20, 16, 10, seed=6)
    b = np.random.default_rng(10).standard_normal(Nb * 3)
    kinds = set()
    for tol in (0.5, 0.2, 0.1, 0.05, 0.02, 1e-2, 5e-3, 2e-3, 1e-3, 1e-4, 1e-5, 1e-6):
        s = pkg.capi.HipSolver(tolerance=tol, maxit=200, reorder="line_coloring")
        res = s.solve_system(Nb, rp, ci, v.copy(), b)
        x = s.get_result()
        xo, ro = oracle_solve_in_order(orc, Nb, rp, ci, v, b, *s.ordering()[:2], tol=tol, maxit=200, w=0.9)
        assert res.converged and res.it == ro.it
        np.testing.assert_allclose(x, xo, rtol=1e-9, atol=1e-12)
        r = b - orc.spmv(Nb, rp, ci, v, x)
        assert np.linalg.norm(r) < tol * np.linalg.norm(b) * (1 + 1e-9)
        kinds.add(res.it % 1.0)
    assert kinds == {0.0, 0.5}


def test_irregular_rows_and_long_rows(pkg, orc):
    Nb, rp, ci, v = random_block_system(700, pattern="random", seed=6, extra=5)
    b = np.random.default_rng(1).standard_normal(Nb * 3)
    for reorder in REORDERS:
        s = pkg.capi.HipSolver(tolerance=1e-6, maxit=200, reorder=reorder)
        res = s.solve_system(Nb, rp, ci, v.copy(), b)
        x = s.get_result()
        to, fr, rpc = s.ordering()
        rr, rc, rv = orc.reorder_matrix(Nb, rp, ci, v, to, fr)
        lu = s.ilu0_factor()
        luo = orc.ilu0_factor(Nb, rr, rc, rv)
        assert np.array_equal(lu, luo)
        # M^-1 by itself (a wrong preconditioner would still let the Krylov loop converge)
        d = np.random.default_rng(2).standard_normal(Nb * 3)
        vo = orc.ilu0_apply(Nb, rr, rc, luo, d.reshape(Nb, 3)[fr].reshape(-1), w=0.9, mode="post_scale")
        assert np.array_equal(s.ilu0_apply(d), vo.reshape(Nb, 3)[to].reshape(-1))
        assert res.converged
        r = b - orc.spmv(Nb, rp, ci, v, x)
        assert np.linalg.norm(r) < 1e-6 * np.linalg.norm(b) * 1.001


@pytest.mark.parametrize("reorder", ["graph_coloring", "line_coloring"])
def test_dense_random_rows(pkg, orc, reorder):
    """rows of up to ~40 blocks (far beyond the 6 / 8 blocks the sweeps and the product prefetch per row, and beyond what one
    step of a chain-tile was sized for): the long-row loops, the columns inside a chain-tile that are not the lane's own
    previous row, and the one-tile product kernel - factors, M^-1 and A x bit for bit, then a solve"""
    Nb, rp, ci, v = random_block_system(900, pattern="random", seed=21, extra=18)
    assert np.diff(rp).max() > 24
    s = pkg.capi.HipSolver(tolerance=1e-8, maxit=300, reorder=reorder)
    b = np.random.default_rng(3).standard_normal(Nb * 3)
    res = s.solve_system(Nb, rp, ci, v.copy(), b)
    x = s.get_result()
    to, fr, rpc = s.ordering()
    rr, rc, rv = orc.reorder_matrix(Nb, rp, ci, v, to, fr)
    luo = orc.ilu0_factor(Nb, rr, rc, rv)
    assert np.array_equal(s.ilu0_factor(), luo)
    rng = np.random.default_rng(4)
    for mode, w in (("post_scale", 0.9), ("post_scale", 1.0)):
        d = rng.standard_normal(Nb * 3)
        s2 = pkg.capi.HipSolver(reorder=reorder, ilu_relaxation=w)
        s2.set_pattern(Nb, rp, ci); s2.upload_system(v)
        assert np.array_equal(s2.ordering()[0], to)
        s2.ilu0_factor()
        vo = orc.ilu0_apply(Nb, rr, rc, luo, d.reshape(Nb, 3)[fr].reshape(-1), w=w, mode=mode)
        assert np.array_equal(s2.ilu0_apply(d), vo.reshape(Nb, 3)[to].reshape(-1))
    y = rng.standard_normal(Nb * 3)
    yo = orc.spmv(Nb, rr, rc, rv, y.reshape(Nb, 3)[fr].reshape(-1)).reshape(Nb, 3)[to].reshape(-1)
    assert np.array_equal(s.spmv(y), yo)
    assert res.converged
    r = b - orc.spmv(Nb, rp, ci, v, x)
    assert np.linalg.norm(r) < 1e-8 * np.linalg.norm(b) * 1.001


@pytest.mark.parametrize("reorder", REORDERS)
def test_block_diagonal_and_nearly_empty_factors(pkg, orc, reorder):
    """No off-diagonal blocks at all (L and U are empty arrays), and a chain of three rows in a sea of isolated ones: the
    prefetch stages of the pipelined kernels read "some valid entry" for idle slots and empty steps - here there is next to
    none, and the base address of the last, empty step is the end of the array"""
    rng = np.random.default_rng(8)
    for Nb, links in ((150, []), (130, [(40, 41), (41, 42)])):
        nb = [set([i]) for i in range(Nb)]
        for a, b in links:
            nb[a].add(b); nb[b].add(a)
        rp = np.zeros(Nb + 1, np.int32)
        cols = []
        for i in range(Nb):
            cols.extend(sorted(nb[i])); rp[i + 1] = len(cols)
        ci = np.array(cols, np.int32)
        val = rng.uniform(-0.1, 0.1, (len(ci), 3, 3))
        row = np.repeat(np.arange(Nb), np.diff(rp))
        val[ci == row] += 2.0 * np.eye(3)
        v = np.ascontiguousarray(val.reshape(-1))
        b = rng.standard_normal(Nb * 3)
        s = pkg.capi.HipSolver(tolerance=1e-10, maxit=50, reorder=reorder)
        res = s.solve_system(Nb, rp, ci, v.copy(), b)
        x = s.get_result()
        assert res.converged
        np.testing.assert_allclose(orc.spmv(Nb, rp, ci, v, x), b, rtol=1e-8, atol=1e-10)
        to, fr, rpc = s.ordering()
        rr, rc, rv = orc.reorder_matrix(Nb, rp, ci, v, to, fr)
        luo = orc.ilu0_factor(Nb, rr, rc, rv)
        assert np.array_equal(s.ilu0_factor(), luo)
        d = rng.standard_normal(Nb * 3)
        vo = orc.ilu0_apply(Nb, rr, rc, luo, d.reshape(Nb, 3)[fr].reshape(-1), w=0.9, mode="post_scale")
        assert np.array_equal(s.ilu0_apply(d), vo.reshape(Nb, 3)[to].reshape(-1))
        y = rng.standard_normal(Nb * 3)
        yo = orc.spmv(Nb, rr, rc, rv, y.reshape(Nb, 3)[fr].reshape(-1)).reshape(Nb, 3)[to].reshape(-1)
        assert np.array_equal(s.spmv(y), yo)


def test_wells_operator(pkg, orc):
    rng = np.random.default_rng(11)
    Nb, rp, ci, v = laplace_block_system(12, 10, 6, seed=13)
    perfs = [5, 1, 9]
    vp = np.concatenate([[0], np.cumsum(perfs)]).astype(np.int32)
    nperf = int(vp[-1])
    cols = rng.choice(Nb, nperf, replace=False).astype(np.int32)
    W = dict(numWells=3, val_pointers=vp, Ccols=cols, Bcols=cols.copy(), Cnnzs=0.05 * rng.standard_normal(nperf * 12),
             Bnnzs=0.05 * rng.standard_normal(nperf * 12), Dnnzs=0.5 * rng.standard_normal(3 * 16))
    b = rng.standard_normal(Nb * 3)
    s = pkg.capi.HipSolver(tolerance=1e-8, maxit=200, reorder="graph_coloring_greedy")
    res = s.solve_system(Nb, rp, ci, v.copy(), b, wells=W)
    x = s.get_result()
    xo, ro = orc.solve(Nb, rp, ci, v, b, tol=1e-8, maxit=200, w=0.9, reorder="graph_coloring_greedy", wells=W)
    assert res.converged and res.it == ro.it
    np.testing.assert_allclose(x, xo, rtol=1e-8, atol=1e-12)
    # x solves (A - C^T D^-1 B) x = b
    r = b - orc.wells_apply(W, x, orc.spmv(Nb, rp, ci, v, x))
    assert np.linalg.norm(r) < 1e-8 * np.linalg.norm(b) * 1.001


def test_wells_with_more_completions_than_a_wavefront_has_lanes(pkg, orc):
    """B x of a well is formed 64 completions at a time (well_Bx in csrc/solver.hip: the products in parallel, their sum in the CPU loop's
    order): wells of 150, 64, 65 and 1 completions - the solve stops where the oracle's does with its solution, and x_w = D^-1 (r_w - B x),
    whose subtractions run through the same code, equals the oracle's bit for bit on the same x"""
    rng = np.random.default_rng(21)
    Nb, rp, ci, v = laplace_block_system(16, 12, 10, seed=23)
    perfs = [150, 64, 65, 1]
    vp = np.concatenate([[0], np.cumsum(perfs)]).astype(np.int32)
    nperf = int(vp[-1])
    cols = rng.choice(Nb, nperf, replace=False).astype(np.int32)
    Dm = [np.linalg.inv(0.2 * rng.standard_normal((4, 4)) + np.diag(2.0 + rng.random(4))) for _ in perfs]
    W = dict(numWells=len(perfs), val_pointers=vp, Ccols=cols, Bcols=cols.copy(), Cnnzs=0.02 * rng.standard_normal(nperf * 12),
             Bnnzs=0.02 * rng.standard_normal(nperf * 12), Dnnzs=np.ascontiguousarray(np.stack(Dm).reshape(-1)))
    b = rng.standard_normal(Nb * 3)
    rw = rng.standard_normal(4 * len(perfs))
    for reorder in ("graph_coloring_greedy", "line_coloring"):
        s = pkg.capi.HipSolver(tolerance=1e-8, maxit=200, reorder=reorder)
        res = s.solve_system(Nb, rp, ci, v.copy(), b, wells=W)
        x = s.get_result()
        to, fr, _ = s.ordering()
        xo, ro = oracle_solve_in_order(orc, Nb, rp, ci, v, b, to, fr, wells=W, tol=1e-8, maxit=200, w=0.9)
        assert res.converged and res.it == ro.it
        np.testing.assert_allclose(x, xo, rtol=1e-8, atol=1e-12)
        r = b - orc.wells_apply(W, x, orc.spmv(Nb, rp, ci, v, x))
        assert np.linalg.norm(r) < 1e-8 * np.linalg.norm(b) * 1.001
        np.testing.assert_array_equal(s.wells_recover_solution(W, rw), orc.wells_recover(W, rw, x))


def test_a_well_list_that_arrives_again(pkg, orc):
    """the device keeps a record of the well list it holds and copies an array only when it differs (a Newton iteration hands the same list
    over three times: residual, solve, well solution): the same list again, the same list with other values of ONE array, a list of another
    shape and the first list once more - every solve and every well solution equals the oracle's on that list"""
    rng = np.random.default_rng(41)
    Nb, rp, ci, v = laplace_block_system(14, 11, 8, seed=43)
    b = rng.standard_normal(Nb * 3)
    rw_all = rng.standard_normal(4 * 5)

    def make(perfs, seed):
        r = np.random.default_rng(seed)
        vp = np.concatenate([[0], np.cumsum(perfs)]).astype(np.int32)
        n = int(vp[-1])
        cols = r.choice(Nb, n, replace=False).astype(np.int32)
        Dm = [np.linalg.inv(0.2 * r.standard_normal((4, 4)) + np.diag(2.0 + r.random(4))) for _ in perfs]
        return dict(numWells=len(perfs), val_pointers=vp, Ccols=cols, Bcols=cols.copy(), Cnnzs=0.03 * r.standard_normal(n * 12),
                    Bnnzs=0.03 * r.standard_normal(n * 12), Dnnzs=np.ascontiguousarray(np.stack(Dm).reshape(-1)))
    W1 = make([4, 2, 7], 1)
    W1d = dict(W1, Dnnzs=W1["Dnnzs"] * 1.25)                       # only D differs
    W1c = dict(W1, Cnnzs=W1["Cnnzs"] * -0.5)                       # only C differs
    W2 = make([3, 9, 1, 2, 5], 2)                                  # other wells, other cells, more of both
    s = pkg.capi.HipSolver(tolerance=1e-9, maxit=300, reorder="line_coloring")
    first = True
    for W in (W1, W1, W1d, W1c, W2, W1, W2):
        res = s.solve_system(Nb, rp if first else None, ci if first else None, v.copy(), b, wells=W)
        first = False
        x = s.get_result()
        to, fr, _ = s.ordering()
        xo, ro = oracle_solve_in_order(orc, Nb, rp, ci, v, b, to, fr, wells=W, tol=1e-9, maxit=300, w=0.9)
        assert res.converged and res.it == ro.it
        np.testing.assert_allclose(x, xo, rtol=1e-8, atol=1e-12)
        rw = rw_all[:4 * W["numWells"]]
        for _ in range(2):                                           # the same list twice more, as a Newton iteration does
            np.testing.assert_array_equal(s.wells_recover_solution(W, rw), orc.wells_recover(W, rw, x))


@pytest.mark.parametrize("reorder", ["level_scheduling", "graph_coloring", "line_coloring"])
def test_wells_operator_in_every_ordering(pkg, orc, reorder):
    """The well operator inside a solve under the other orderings (the perforated cells are renamed at upload, opmhip_wells' indices are
    natural ones): wells large enough to be felt - forgetting them leaves a residual a thousand tolerances large - device against the
    oracle's solve in the device's ordering, to the iteration count."""
    import helpers
    rng = np.random.default_rng(12)
    Nb, rp, ci, v = laplace_block_system(12, 10, 6, seed=13)
    perfs = [7, 1, 4, 12]
    vp = np.concatenate([[0], np.cumsum(perfs)]).astype(np.int32)
    nperf = int(vp[-1])
    cols = rng.choice(Nb, nperf, replace=False).astype(np.int32)
    W = dict(numWells=4, val_pointers=vp, Ccols=cols, Bcols=cols.copy(), Cnnzs=0.05 * rng.standard_normal(nperf * 12),
             Bnnzs=0.05 * rng.standard_normal(nperf * 12), Dnnzs=0.5 * rng.standard_normal(4 * 16))
    b = rng.standard_normal(Nb * 3)
    s = pkg.capi.HipSolver(tolerance=1e-8, maxit=200, reorder=reorder)
    res = s.solve_system(Nb, rp, ci, v.copy(), b, wells=W)
    x = s.get_result()
    to, fr, _ = s.ordering()
    xo, ro = helpers.oracle_solve_in_order(orc, Nb, rp, ci, v, b, to, fr, wells=W, tol=1e-8, maxit=200, w=0.9)
    assert res.converged and ro.converged and res.it == ro.it
    np.testing.assert_allclose(x, xo, rtol=1e-8, atol=1e-12)
    ax = orc.spmv(Nb, rp, ci, v, x)
    assert np.linalg.norm(b - orc.wells_apply(W, x, ax)) < 1e-8 * np.linalg.norm(b) * 1.001
    assert np.linalg.norm(b - ax) > 1e3 * 1e-8 * np.linalg.norm(b)


def test_zero_diagonal_fix_and_nonconvergence(pkg, orc):
    Nb, rp, ci, v = laplace_block_system(6, 6, 4, seed=14)
    b = np.ones(Nb * 3)
    dk = [k for i in range(Nb) for k in range(rp[i], rp[i + 1]) if ci[k] == i]
    v0 = v.copy()
    v0[dk[3] * 9 + 4] = 0.0
    s = pkg.capi.HipSolver(tolerance=1e-2, maxit=200, reorder="level_scheduling")
    res = s.solve_system(Nb, rp, ci, v0.copy(), b)
    xo, ro = orc.solve(Nb, rp, ci, v0, b, tol=1e-2, maxit=200, w=0.9, zero_diag_fix=True)
    assert res.converged == ro.converged and res.it == ro.it
    # maxit exhaustion is reported, not raised (ISTLSolverEbos falls back to Dune in that case)
    s2 = pkg.capi.HipSolver(tolerance=1e-14, maxit=2, reorder="graph_coloring")
    res2 = s2.solve_system(Nb, rp, ci, v.copy(), b)
    assert not res2.converged and res2.iterations == 2 and res2.it == 2.5


def test_argument_errors(pkg):
    Nb, rp, ci, v = laplace_block_system(4, 4, 2, seed=1)
    s = pkg.capi.HipSolver()
    with pytest.raises(pkg.capi.OpmHipError) as e:
        s.get_result()
    assert e.value.code == pkg.capi.NOT_READY
    bad = ci.copy()
    bad[1], bad[2] = bad[2], bad[1]
    with pytest.raises(pkg.capi.OpmHipError) as e:
        s.set_pattern(Nb, rp, bad)
    assert e.value.code == pkg.capi.INVALID_ARGUMENT
    res = pkg.capi.Result()
    import ctypes as C
    assert pkg.capi.lib().opmhip_solve_system(s._h, 8, 16, 2, None, None, None, None, None, C.byref(res)) == pkg.capi.INVALID_ARGUMENT
    # the well helpers and get_rhs need a system / a solution on the device first
    W = dict(numWells=1, val_pointers=np.array([0, 1], np.int32), Ccols=np.array([0], np.int32), Bcols=np.array([0], np.int32),
             Cnnzs=np.zeros(12), Bnnzs=np.zeros(12), Dnnzs=np.eye(4).reshape(-1))
    s2 = pkg.capi.HipSolver()
    s2.set_pattern(Nb, rp, ci)
    for call in (lambda: s2.get_rhs(), lambda: s2.wells_apply_residual(W, np.zeros(4)), lambda: s2.add_well_contributions(W),
                 lambda: s2.wells_recover_solution(W, np.zeros(4))):
        with pytest.raises(pkg.capi.OpmHipError) as e:
            call()
        assert e.value.code == pkg.capi.NOT_READY
    s2.upload_system(v, np.ones(3 * Nb))
    s2.wells_apply_residual(W, np.zeros(4))      # fine now; a zero well residual changes nothing
    assert np.array_equal(s2.get_rhs(), np.ones(3 * Nb))
    bad_w = dict(W, Ccols=np.array([Nb], np.int32))
    with pytest.raises(pkg.capi.OpmHipError) as e:
        s2.wells_apply_residual(bad_w, np.zeros(4))
    assert e.value.code == pkg.capi.INVALID_ARGUMENT


@pytest.mark.parametrize("reorder", ["graph_coloring", "line_coloring"])
@pytest.mark.parametrize("repeat_in_well", [False, True])
def test_add_well_contributions_to_matrix(pkg, orc, reorder, repeat_in_well):
    """--matrix-add-well-contributions mode (StandardWell::addWellContributions): A -= C^T D^-1 B written into the
    device-resident matrix whose pattern carries the well cliques; bit for bit, and the solve with the modified
    matrix equals the solve with the operator form of the same wells up to the Krylov tolerance."""
    rng = np.random.default_rng(17)
    Nb, rp, ci, v = laplace_block_system(10, 9, 6, seed=19)
    perfs = [4, 2, 6]
    vp = np.concatenate([[0], np.cumsum(perfs)]).astype(np.int32)
    nperf = int(vp[-1])
    cells = rng.choice(Nb, nperf, replace=False).astype(np.int32)
    cells[6:12] = cells[6:12][np.argsort(cells[6:12])]
    cells[5] = cells[8]            # wells 1 and 2 share a cell: their contributions to block (cell, cell) are ordered
    if repeat_in_well:             # well 0 perforates one cell twice: four of its pairs add into the same block, in order
        cells[2] = cells[0]
    W = dict(numWells=3, val_pointers=vp, Ccols=cells, Bcols=cells.copy(), Cnnzs=0.05 * rng.standard_normal(nperf * 12),
             Bnnzs=0.05 * rng.standard_normal(nperf * 12), Dnnzs=0.5 * rng.standard_normal(3 * 16))
    # pattern + well cliques (zero blocks)
    nb = [set(ci[rp[i]:rp[i + 1]]) for i in range(Nb)]
    for w in range(3):
        cs = cells[vp[w]:vp[w + 1]]
        for a in cs:
            nb[a].update(int(x) for x in cs)
    rp2 = np.zeros(Nb + 1, np.int32)
    ci2, v2 = [], []
    old = {(i, int(ci[k])): k for i in range(Nb) for k in range(rp[i], rp[i + 1])}
    for i in range(Nb):
        for j in sorted(nb[i]):
            ci2.append(j)
            k = old.get((i, int(j)))
            v2.append(v[9 * k:9 * k + 9] if k is not None else np.zeros(9))
        rp2[i + 1] = len(ci2)
    ci2, v2 = np.array(ci2, np.int32), np.ascontiguousarray(np.concatenate(v2))
    s = pkg.capi.HipSolver(tolerance=1e-10, maxit=200, reorder=reorder)
    s.set_pattern(Nb, rp2, ci2)
    s.upload_system(v2)
    s.add_well_contributions(W)
    rc, vo = orc.wells_add_to_matrix(Nb, rp2, ci2, v2, W)
    assert rc == 0 and not np.array_equal(vo, v2)
    to, fr, rpc = s.ordering()
    rr, rc_, rv = orc.reorder_matrix(Nb, rp2, ci2, vo, to, fr)
    for seed in range(3):
        x = np.random.default_rng(seed).standard_normal(Nb * 3)
        yo = orc.spmv(Nb, rr, rc_, rv, x.reshape(Nb, 3)[fr].reshape(-1)).reshape(Nb, 3)[to].reshape(-1)
        assert np.array_equal(s.spmv(x), yo)
    # the same physics two ways: wells inside the matrix vs wells as an operator after every SpMV
    b = rng.standard_normal(Nb * 3)
    s2 = pkg.capi.HipSolver(tolerance=1e-10, maxit=200, reorder=reorder)
    r_op = s2.solve_system(Nb, rp2, ci2, v2.copy(), b, wells=W)
    x_op = s2.get_result()
    s3 = pkg.capi.HipSolver(tolerance=1e-10, maxit=200, reorder=reorder)
    r_mat = s3.solve_system(Nb, rp2, ci2, vo.copy(), b)
    assert r_op.converged and r_mat.converged
    np.testing.assert_allclose(s3.get_result(), x_op, rtol=1e-6, atol=1e-8)
    # a clique block missing from the pattern is an argument error and leaves the matrix alone
    s4 = pkg.capi.HipSolver(reorder=reorder)
    s4.set_pattern(Nb, rp, ci)
    s4.upload_system(v)
    with pytest.raises(pkg.capi.OpmHipError):
        s4.add_well_contributions(W)


def test_contexts_give_their_memory_back(pkg):
    """create / use / destroy in a loop: device memory returns to where it was (every allocation of a context is tracked
    and freed by opmhip_destroy, including the pinned read-back ring and the profiler's events)"""
    import gc
    Nb, rp, ci, v = laplace_block_system(20, 20, 10, seed=3)
    b = np.ones(3 * Nb)

    def once():
        s = pkg.capi.HipSolver(reorder="line_coloring")
        s.profile_enable(1)
        s.solve_system(Nb, rp, ci, v.copy(), b)
        s.profile()
        del s
        gc.collect()
    # free device memory straight from the HIP runtime the library itself uses (no second framework in the process)
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")

    def free_bytes():
        assert hip.hipDeviceSynchronize() == 0
        fr, tot = ctypes.c_size_t(0), ctypes.c_size_t(0)
        assert hip.hipMemGetInfo(ctypes.byref(fr), ctypes.byref(tot)) == 0
        return fr.value

    once()
    free0 = free_bytes()
    for _ in range(25):
        once()
    free1 = free_bytes()
    assert free0 - free1 < 8 * 1024 * 1024, (free0, free1)


def test_failure_statuses_of_the_backend_interface(pkg):
    """bda::SolverStatus as the reference's back-ends report it: a pattern the analysis cannot use, and a factorisation
    that meets a singular pivot (bda/BdaSolver.hpp:32-37; ISTLSolverEbos falls back to Dune on either)."""
    Nb, rp, ci, v = laplace_block_system(5, 4, 3, seed=2)
    b = np.ones(3 * Nb)
    # analysis: a row without its diagonal block ("diagonal entry missing", ParallelOverlappingILU0.hpp:484-485)
    keep = np.ones(len(ci), bool)
    d7 = [k for k in range(rp[7], rp[7 + 1]) if ci[k] == 7][0]
    keep[d7] = False
    rp2 = np.concatenate([[0], np.cumsum(np.add.reduceat(keep.astype(int), rp[:-1]))]).astype(np.int32)
    s = pkg.capi.HipSolver()
    with pytest.raises(pkg.capi.OpmHipError) as e:
        s.set_pattern(Nb, rp2, ci[keep])
    assert e.value.code == pkg.capi.ANALYSIS_FAILED
    # analysis: a row longer than one tile can stage
    n = 400
    nbr = [set([i]) for i in range(n)]
    for j in range(1, 300):
        nbr[0].add(j); nbr[j].add(0)
    rpl = np.zeros(n + 1, np.int32); cl = []
    for i in range(n):
        cl.extend(sorted(nbr[i])); rpl[i + 1] = len(cl)
    s = pkg.capi.HipSolver()
    res = pkg.capi.Result()
    import ctypes as C
    vl = np.ascontiguousarray(np.tile(np.eye(3).reshape(-1), len(cl)))
    bl = np.ones(3 * n)
    cla = np.array(cl, np.int32)
    rc = pkg.capi.lib().opmhip_solve_system(s._h, 3 * n, 9 * len(cl), 3, vl.ctypes.data_as(C.c_void_p), rpl.ctypes.data_as(C.c_void_p),
                                            cla.ctypes.data_as(C.c_void_p), bl.ctypes.data_as(C.c_void_p), None, C.byref(res))
    assert rc == pkg.capi.ANALYSIS_FAILED
    # factorisation: a singular diagonal block (zero_diag_fix only repairs exact zeros ON the diagonal of the block)
    v0 = v.copy()
    dk = [k for k in range(rp[0], rp[0 + 1]) if ci[k] == 0][0]    # row 0 has no lower neighbours: its pivot is the block itself
    v0[9 * dk:9 * dk + 9] = np.array([1.0, 2.0, 3.0, 2.0, 4.0, 6.0, 1.0, 1.0, 1.0])   # rank 2, determinant exactly 0
    s = pkg.capi.HipSolver(maxit=50, reorder="level_scheduling")
    with pytest.raises(pkg.capi.OpmHipError) as e:
        s.solve_system(Nb, rp, ci, v0, b)
    assert e.value.code == pkg.capi.CREATE_PRECONDITIONER_FAILED


def test_reorder_auto_picks_by_size_and_regularity(pkg):
    """OPMHIP_REORDER_AUTO: the greedy colouring on small or irregular patterns, the line colouring on a structured grid of at least
    30 000 rows in its natural order (chains of 4 / 8 / 10 by size) - the orderings of the explicit choices, entry for entry"""
    from helpers import cartesian_pattern
    def ordering(Nb, rp, ci, reorder, **kw):
        s = pkg.capi.HipSolver(reorder=reorder, **kw)
        s.set_pattern(Nb, rp, ci)
        return s.ordering()
    Nb, rp, ci = cartesian_pattern(12, 10, 9)
    ta, fa, ca = ordering(Nb, rp, ci, "auto")
    tg, fg, cg = ordering(Nb, rp, ci, "graph_coloring_greedy")
    assert np.array_equal(ta, tg) and np.array_equal(ca, cg)
    Nb, rp, ci = cartesian_pattern(60, 60, 50)          # 180 000 rows, seven column offsets: chains of 4
    ta, fa, ca = ordering(Nb, rp, ci, "auto")
    tl, fl, cl = ordering(Nb, rp, ci, "line_coloring", chain_length=4)
    assert np.array_equal(ta, tl) and np.array_equal(ca, cl) and len(ca) == 2
    ta, fa, ca = ordering(Nb, rp, ci, "auto", chain_length=7)   # a chain length that is given is kept
    tl, fl, cl = ordering(Nb, rp, ci, "line_coloring", chain_length=7)
    assert np.array_equal(ta, tl)
    # the same size with scattered extra couplings (more than 15 distinct column offsets): greedy
    rng = np.random.default_rng(1)
    extra = {}
    for a, b in zip(rng.integers(0, Nb, 400), rng.integers(0, Nb, 400)):
        if a != b:
            extra.setdefault(int(a), set()).add(int(b)); extra.setdefault(int(b), set()).add(int(a))
    rows = []
    for i in range(Nb):
        r = set(ci[rp[i]:rp[i + 1]].tolist()) | extra.get(i, set())
        rows.append(sorted(r))
    rp2 = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    ci2 = np.concatenate(rows).astype(np.int32)
    ta, fa, ca = ordering(Nb, rp2, ci2, "auto")
    tg, fg, cg = ordering(Nb, rp2, ci2, "graph_coloring_greedy")
    assert np.array_equal(ta, tg) and np.array_equal(ca, cg)


def test_solves_under_poisoned_allocations(tmp_path):
    """OPMHIP_POISON_ALLOC=1 (a debugging switch, tools/README.md): every floating-point device array starts as NaNs - a solve (ILU0 and
    CPR, the device-assembled Newton step included) that reads nothing it has not written gives the bits it gives without the switch"""
    import subprocess
    import sys
    code = r'''
import importlib, sys
import numpy as np
sys.path.insert(0, "tests")
from helpers import laplace_block_system
pkg = importlib.import_module("opm-autodiff_amd")
out = []
Nb, rp, ci, v = laplace_block_system(14, 12, 9, seed=3)
b = np.random.default_rng(4).standard_normal(3 * Nb)
for kw in (dict(reorder="line_coloring"), dict(reorder="graph_coloring_greedy"), dict(reorder="line_coloring", preconditioner="cpr_quasiimpes")):
    s = pkg.capi.HipSolver(tolerance=1e-8, maxit=200, **kw)
    res = s.solve_system(Nb, rp, ci, v.copy(), b)
    out.append(np.concatenate([[res.it, float(res.converged)], s.get_result()]))
case = pkg.decks.cartesian_case(9, 8, 6, state="mixed", heterogeneous=True)
m = pkg.capi.HipModel(case, tolerance=1e-6)
m.set_state(case["pv"], case["meaning"])
m.set_source(pkg.decks.five_spot_source(case, rate_sm3_per_day=20.0))
for it in range(2):
    m.assemble(86400.0, it, fetch=False)
    res = m.solve_jacobian_system()
    out.append(np.concatenate([[res.it, float(res.converged)], m.get_result()]))
    m.update(None, 1.0)
out.append(m.get_state()[0])
np.save(sys.argv[1], np.concatenate(out))
'''
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = []
    for tag, extra in (("plain", {}), ("poison", {"OPMHIP_TUNING": "1", "OPMHIP_POISON_ALLOC": "1"})):
        f = str(tmp_path / (tag + ".npy"))
        env = dict(os.environ, **extra)
        if not extra:
            env.pop("OPMHIP_POISON_ALLOC", None)
        r = subprocess.run([sys.executable, "-c", code, f], cwd=root, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        if extra:
            assert "OPMHIP_POISON_ALLOC=1 is in force" in r.stderr
        got.append(np.load(f))
    assert np.all(np.isfinite(got[0])) and np.array_equal(got[0], got[1])
