"""Shapes the structured grids never produce, for all four ILU orderings: random sparse graphs with degrees 0..24,
a star (one row coupled to many), a path, isolated rows, tiny colours.  For each: SpMV, ILU0 factors and M^-1 bit for bit
against the oracle in the device's ordering, and the solve's iteration count."""
import numpy as np
import pytest

import oracle_bind
from helpers import oracle_solve_in_order

pytestmark = pytest.mark.gpu
REORDERS = ["level_scheduling", "graph_coloring", "graph_coloring_greedy", "line_coloring"]


def graph(kind, n, rng):
    nb = [set([i]) for i in range(n)]

    def link(i, j):
        if i != j:
            nb[i].add(j)
            nb[j].add(i)
    if kind == "random":
        deg = rng.integers(0, 13, n)
        for i in range(n):
            for j in rng.integers(0, n, deg[i]):
                if len(nb[i]) < 25 and len(nb[int(j)]) < 25:
                    link(i, int(j))
    elif kind == "star":
        for j in range(1, min(n, 150)):
            link(0, j)
        for i in range(1, n - 1):
            link(i, i + 1)
    elif kind == "path":
        for i in range(n - 1):
            link(i, i + 1)
    elif kind == "isolated":
        for i in range(0, n - 1, 7):
            link(i, i + 1)
    elif kind == "banded":
        for i in range(n):
            for o in (1, 2, 3, 17, 40):
                if i + o < n:
                    link(i, i + o)
    rowptr = np.zeros(n + 1, np.int32)
    cols = []
    for i in range(n):
        cols.extend(sorted(nb[i]))
        rowptr[i + 1] = len(cols)
    col = np.array(cols, np.int32)
    nnzb = len(col)
    val = rng.uniform(-1, 1, size=(nnzb, 3, 3)) * 0.2
    row = np.repeat(np.arange(n), np.diff(rowptr))
    s = np.zeros((n, 3))
    np.add.at(s, row, np.abs(val).sum(axis=2))
    dk = np.flatnonzero(col == row)
    for e in range(3):
        val[dk, e, e] = 1.5 * (s[:, e] + 0.5)
    return rowptr, col, np.ascontiguousarray(val.reshape(-1))


CASES = [("random", 50, 1), ("random", 333, 2), ("random", 2500, 3), ("random", 4097, 4), ("star", 400, 5), ("path", 1000, 6),
         ("isolated", 300, 7), ("banded", 1500, 8), ("random", 1, 9), ("path", 2, 10), ("random", 65, 11), ("star", 33, 12)]


@pytest.mark.parametrize("reorder", REORDERS)
@pytest.mark.parametrize("kind,n,seed", CASES)
def test_random_graph(pkg, orc, kind, n, seed, reorder):
    rng = np.random.default_rng(seed)
    rp, ci, v = graph(kind, n, rng)
    b = rng.standard_normal(3 * n)
    for w, mode in ((0.9, "post_scale"), (1.0, "post_scale"), (0.9, "in_sweep")):
        sol = pkg.capi.HipSolver(tolerance=1e-8, maxit=200, reorder=reorder, ilu_relaxation=w, relax_mode=mode)
        res = sol.solve_system(n, rp, ci, v.copy(), b)
        x = sol.get_result()
        to, fr, rpc = sol.ordering()
        assert sorted(to) == list(range(n)) and np.array_equal(fr[to], np.arange(n)) and int(np.sum(rpc)) == n
        rr, rc, rv = orc.reorder_matrix(n, rp, ci, v, to, fr)
        luo = orc.ilu0_factor(n, rr, rc, rv)
        assert np.array_equal(sol.ilu0_factor(), luo)
        y = rng.standard_normal(3 * n)
        yo = orc.spmv(n, rr, rc, rv, y.reshape(n, 3)[fr].reshape(-1)).reshape(n, 3)[to].reshape(-1)
        assert np.array_equal(sol.spmv(y), yo)
        d = rng.standard_normal(3 * n)
        vo = orc.ilu0_apply(n, rr, rc, luo, d.reshape(n, 3)[fr].reshape(-1), w=w, mode=mode)
        assert np.array_equal(sol.ilu0_apply(d), vo.reshape(n, 3)[to].reshape(-1))
        xo, ro = oracle_solve_in_order(orc, n, rp, ci, v, b, to, fr, tol=1e-8, maxit=200, w=w, mode=mode)
        assert res.converged and ro.converged and res.it == ro.it
        np.testing.assert_allclose(x, xo, rtol=1e-8, atol=1e-12)


@pytest.mark.parametrize("chain_length", [1, 2, 3, 5, 13, 64, 128])
@pytest.mark.parametrize("shape", [(3, 3, 140), (13, 7, 11), (1, 1, 300)])
def test_line_colouring_chain_lengths(pkg, orc, shape, chain_length):
    """chains of every length a tall grid allows (single rows up to the 128-step limit of a chain tile), remainders at
    the column ends included: exact ILU0 of the permuted matrix, bit for bit"""
    from helpers import laplace_block_system
    n, rp, ci, v = laplace_block_system(*shape, seed=31)
    rng = np.random.default_rng(chain_length)
    b = rng.standard_normal(3 * n)
    sol = pkg.capi.HipSolver(tolerance=1e-8, maxit=200, reorder="line_coloring", chain_length=chain_length)
    res = sol.solve_system(n, rp, ci, v.copy(), b)
    to, fr, rpc = sol.ordering()
    rr, rc, rv = orc.reorder_matrix(n, rp, ci, v, to, fr)
    luo = orc.ilu0_factor(n, rr, rc, rv)
    assert np.array_equal(sol.ilu0_factor(), luo)
    d = rng.standard_normal(3 * n)
    vo = orc.ilu0_apply(n, rr, rc, luo, d.reshape(n, 3)[fr].reshape(-1), w=0.9, mode="post_scale")
    assert np.array_equal(sol.ilu0_apply(d), vo.reshape(n, 3)[to].reshape(-1))
    xo, ro = oracle_solve_in_order(orc, n, rp, ci, v, b, to, fr, tol=1e-8, maxit=200, w=0.9)
    assert res.converged and res.it == ro.it
    np.testing.assert_allclose(sol.get_result(), xo, rtol=1e-8, atol=1e-12)


def test_line_colouring_rejects_chains_beyond_the_step_limit(pkg):
    from helpers import laplace_block_system
    n, rp, ci, v = laplace_block_system(1, 1, 300, seed=1)
    sol = pkg.capi.HipSolver(reorder="line_coloring", chain_length=200)
    with pytest.raises(pkg.capi.OpmHipError) as e:
        sol.set_pattern(n, rp, ci)
    assert e.value.code == pkg.capi.ANALYSIS_FAILED


@pytest.mark.parametrize("reorder", ["graph_coloring", "line_coloring", "level_scheduling"])
@pytest.mark.parametrize("kind,n,seed", [("random", 333, 2), ("random", 2500, 3), ("banded", 1500, 8), ("path", 1000, 6), ("isolated", 300, 7), ("random", 1, 9), ("path", 2, 10)])
def test_cpr_with_ilu0_smoothed_amg_levels_on_random_graphs(pkg, orc, kind, n, seed, reorder):
    """opmhip_config.cpr_amg_ilu_levels on patterns with triangles, rows of 1 to 20 entries, isolated rows, hierarchies of one level: the
    general scalar factorisation (these levels are not `simple`), sweeps over many colours - the application of the preconditioner is the
    oracle's bit for bit"""
    rng = np.random.default_rng(seed)
    rp, ci, v = graph(kind, n, rng)
    s = pkg.capi.HipSolver(tolerance=1e-8, maxit=200, reorder=reorder, preconditioner="cpr_quasiimpes", cpr_amg_ilu_levels=2)
    s.set_pattern(n, rp, ci)
    s.upload_system(v)
    s.ilu0_factor(want_factors=False)
    to, fr, _ = s.ordering()
    rr, rc, rv = orc.reorder_matrix(n, rp, ci, v, to, fr)
    cpr = oracle_bind.OracleCpr(orc)
    cpr.set_natural_ids(fr)
    cpr.set_ilu_smoother(2, 1)
    cpr.update(n, rr, rc, rv)
    d = rng.standard_normal(3 * n)
    vo = cpr.apply(np.ascontiguousarray(d.reshape(n, 3)[fr].reshape(-1))).reshape(n, 3)[to].reshape(-1)
    assert np.array_equal(s.cpr_apply(d), vo)
    b = rng.standard_normal(3 * n)
    r = s.solve_system(n, rp, ci, v.copy(), b)
    assert r.converged
    assert np.linalg.norm(orc.spmv(n, rp, ci, v, s.get_result()) - b) <= 1e-7 * np.linalg.norm(b)
