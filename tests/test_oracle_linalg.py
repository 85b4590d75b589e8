"""Pins the CPU oracle's linear-solve restatement against the reference's own known answers
(SURVEY.md §8c): matr33/rhs3 -> three expected vectors, matr33rep/rhs3rep, the LUe == Ae identity of
tests/test_milu.cpp, and the 3x3 inverse.  CPU only."""
import json
import os

import numpy as np
import pytest

from helpers import laplace_block_system, random_block_system


def _load(pkg, golden, mat, rhs):
    Nb, rp, ci, v, bs = pkg.mmio.read_block_matrix(os.path.join(golden, "linalg", mat))
    b = pkg.mmio.read_block_vector(os.path.join(golden, "linalg", rhs))
    assert bs == 3
    return Nb, rp, ci, v, b


@pytest.fixture(scope="module")
def expected(golden):
    with open(os.path.join(golden, "linalg", "expected.json")) as f:
        return json.load(f)


def check_close(x, ref, percent):
    # BOOST_CHECK_CLOSE(a, b, tol%) : |a-b|/|a| <= tol% and |a-b|/|b| <= tol%
    x = np.asarray(x)
    ref = np.asarray(ref)
    d = np.abs(x - ref)
    assert np.all(d <= percent * 1e-2 * np.abs(ref) * (1 + 1e-12) + 0.0), (x, ref)
    assert np.all(d <= percent * 1e-2 * np.abs(x) * (1 + 1e-12) + 0.0), (x, ref)


def test_matr33_pattern(pkg, golden):
    Nb, rp, ci, v, b = _load(pkg, golden, "matr33.txt", "rhs3.txt")
    assert Nb == 3 and list(rp) == [0, 2, 5, 7] and list(ci) == [0, 1, 0, 1, 2, 1, 2]
    assert v.shape == (63,) and v[1] == 0.126774 and v[3] == 4.64465e-10  # row-major inside a block


# The printed expectations carry 5-6 significant digits; the reference compares with 1e-3 % = 1e-5 relative,
# which is at the rounding of the printed digits, so allow 2 units of the last printed digit on top.
def _cmp(x, e):
    ref = np.array(e["x"])
    for xi, ri in zip(x, ref):
        digits = len(("%r" % ri).split("e")[0].replace("-", "").replace(".", "").lstrip("0"))
        tol = max(e["check_close_percent"] * 1e-2, 2.0 * 10.0 ** (-(digits - 1)))
        assert abs(xi - ri) <= tol * abs(ri), (xi, ri, tol)


def _dense(Nb, rp, ci, v):
    A = np.zeros((Nb * 3, Nb * 3))
    for i in range(Nb):
        for k in range(rp[i], rp[i + 1]):
            A[3 * i:3 * i + 3, 3 * ci[k]:3 * ci[k] + 3] = v[9 * k:9 * k + 9].reshape(3, 3)
    return A


@pytest.mark.parametrize("key,w,mode", [("cusparse_ilu0_w1_tol0.5_maxit20", 1.0, "post_scale"),
                                        ("opencl_ilu0_w0.9_tol0.5_maxit20", 0.9, "in_sweep")])
def test_gpu_backend_vectors_vs_recurrence(pkg, orc, golden, expected, key, w, mode):
    """The two vectors hard-coded in tests/test_cusparseSolver.cpp:103-105 and tests/test_openclSolver.cpp:102-104.

    FINDING (documented in DESIGN.md "Oracle pins"): neither vector is an iterate the in-tree recurrence
    (bda/cusparseSolverBackend.cu:60-184) can stop on at tol 0.5: their TRUE residuals are 192x and 211x |b|,
    while matr33 is block tridiagonal, so block ILU0 is an exact LU, the first half iteration already lands on
    the exact solution (the one tests/test_flexiblesolver.cpp:114-116 pins) and the loop exits at it = 0.5.
    The reference skips both tests when no device is present, so its CI never checks them.  We therefore pin
    the oracle on the stopping rule itself plus the exact-solution vector, and record the mismatch here so
    that it is visible instead of silently dropped."""
    e = expected[key]
    Nb, rp, ci, v, b = _load(pkg, golden, e["matrix"], e["rhs"])
    A = _dense(Nb, rp, ci, v)
    ref = np.array(e["x"])
    assert np.linalg.norm(A @ ref - b) / np.linalg.norm(b) > 100.0  # not a tol-0.5 solution
    x, res = orc.solve(Nb, rp, ci, v, b, tol=e["tol"], maxit=e["maxit"], w=w, mode=mode)
    assert res.converged and res.it == 0.5
    assert np.linalg.norm(A @ x - b) < e["tol"] * np.linalg.norm(b)
    assert res.reduction < e["tol"]
    # what the two vectors DO share with the algorithm: the first block row of A x = b holds (row 0 of U is
    # row 0 of A), as it does for our result
    np.testing.assert_allclose((A @ ref)[:3], b[:3], rtol=1e-4)
    np.testing.assert_allclose((A @ x)[:3], b[:3], rtol=1e-6 if w == 1.0 else 0.2)


def test_level_scheduling_keeps_answer(pkg, orc, golden, expected):
    e = expected["opencl_ilu0_w0.9_tol0.5_maxit20"]
    Nb, rp, ci, v, b = _load(pkg, golden, e["matrix"], e["rhs"])
    x, _ = orc.solve(Nb, rp, ci, v, b, tol=e["tol"], maxit=e["maxit"], w=0.9, mode="in_sweep")
    x2, _ = orc.solve(Nb, rp, ci, v, b, tol=e["tol"], maxit=e["maxit"], w=0.9, mode="in_sweep",
                      reorder="level_scheduling")
    np.testing.assert_allclose(x2, x, rtol=1e-12)


def test_exact_solution_noprec(pkg, orc, golden, expected):
    e = expected["exact_noprec_tol1e-12_maxit200"]
    Nb, rp, ci, v, b = _load(pkg, golden, e["matrix"], e["rhs"])
    x, res = orc.solve_noprec(Nb, rp, ci, v, b, e["tol"], e["maxit"])
    assert res.converged
    _cmp(x, e)
    # and the ILU0 path at a tight tolerance lands on the same solution
    x2, res2 = orc.solve(Nb, rp, ci, v, b, tol=1e-12, maxit=200, w=0.9)
    assert res2.converged
    np.testing.assert_allclose(x2, x, rtol=1e-6)


def test_rep_operator(pkg, orc, golden, expected):
    e = expected["rep_operator_squared_noprec"]
    Nb, rp, ci, v, b = _load(pkg, golden, e["matrix"], e["rhs"])
    x, res = orc.solve_noprec(Nb, rp, ci, v, b, e["tol"], e["maxit"], repeat=2)
    assert res.converged
    check_close(x, e["x"], e["check_close_percent"])


# tests/test_milu.cpp:41-100 pins ILU through two properties on a Laplacian: L*U*e == A*e where the pattern is
# closed under elimination, and (LU)^-1 (LU e) == e.  The two tests below are those properties for 3x3 blocks.
def test_ilu_exact_for_block_tridiagonal(orc):
    Nb, rp, ci, v = random_block_system(40, pattern="tridiag", seed=3)
    lu = orc.ilu0_factor(Nb, rp, ci, v)
    e = np.ones(Nb * 3)
    Ae = orc.spmv(Nb, rp, ci, v, e)
    x = orc.ilu0_apply(Nb, rp, ci, lu, Ae, w=1.0)
    np.testing.assert_allclose(x, e, rtol=1e-9, atol=1e-9)


def test_ilu_apply_inverts_LU_on_grid(orc):
    # (LU)^-1 (L U e) == e for the 7-point pattern: build LUe from the factors directly
    Nb, rp, ci, v = laplace_block_system(5, 4, 3, seed=1)
    lu = orc.ilu0_factor(Nb, rp, ci, v)
    e = np.linspace(1.0, 2.0, Nb * 3)
    # U e  (U has diagonal D = inv(stored D^-1))
    blocks = lu.reshape(-1, 3, 3)
    ue = np.zeros((Nb, 3))
    ev = e.reshape(Nb, 3)
    for i in range(Nb):
        for k in range(rp[i], rp[i + 1]):
            j = ci[k]
            if j == i:
                ue[i] += np.linalg.solve(blocks[k], ev[j])
            elif j > i:
                ue[i] += blocks[k] @ ev[j]
    lue = ue.copy()
    for i in range(Nb):
        for k in range(rp[i], rp[i + 1]):
            j = ci[k]
            if j < i:
                lue[i] += blocks[k] @ ue[j]
    x = orc.ilu0_apply(Nb, rp, ci, lu, lue.reshape(-1), w=1.0)
    np.testing.assert_allclose(x, e, rtol=1e-10)
    # relaxed variants: post-scale is exactly w times that
    xw = orc.ilu0_apply(Nb, rp, ci, lu, lue.reshape(-1), w=0.9)
    np.testing.assert_allclose(xw, 0.9 * e, rtol=1e-10)


def test_reorderings_are_permutations_and_valid(orc):
    nx, ny, nz = 6, 5, 4
    Nb, rp, ci, v = laplace_block_system(nx, ny, nz, seed=2)
    for kind in ("level_scheduling", "graph_coloring", "graph_coloring_greedy"):
        to, fr, rpc = orc.reorder(Nb, rp, ci, kind)
        assert sorted(to) == list(range(Nb)) and np.all(fr[to] == np.arange(Nb))
        assert rpc.sum() == Nb
        rr, rc, rv = orc.reorder_matrix(Nb, rp, ci, v, to, fr)
        # rows inside one colour/level never reference each other below the diagonal
        start = np.concatenate([[0], np.cumsum(rpc)])
        color_of = np.repeat(np.arange(len(rpc)), rpc)
        for i in range(Nb):
            cols = rc[rr[i]:rr[i + 1]]
            assert np.all(np.diff(cols) > 0)
            lower = cols[cols < i]
            assert np.all(color_of[lower] < color_of[i])
            if kind != "level_scheduling":
                others = cols[cols != i]
                assert np.all(color_of[others] != color_of[i])
    to, fr, rpc = orc.reorder(Nb, rp, ci, "level_scheduling")
    assert len(rpc) == nx + ny + nz - 2  # i+j+k hyperplanes
    to, fr, rpc = orc.reorder(Nb, rp, ci, "graph_coloring_greedy")
    assert len(rpc) == 2  # red-black


def test_level_scheduling_matches_natural_order_bitwise(orc):
    Nb, rp, ci, v = laplace_block_system(6, 5, 4, seed=5)
    b = np.random.default_rng(0).standard_normal(Nb * 3)
    x0, r0 = orc.solve(Nb, rp, ci, v, b, tol=1e-8, maxit=100, w=0.9)
    x1, r1 = orc.solve(Nb, rp, ci, v, b, tol=1e-8, maxit=100, w=0.9, reorder="level_scheduling")
    assert r0.converged and r1.converged and r0.it == r1.it
    # same factors, but dot products run in permuted order -> equal to rounding, not bitwise
    np.testing.assert_allclose(x1, x0, rtol=1e-9, atol=1e-12)


def test_coloring_changes_preconditioner_but_not_solution(orc):
    Nb, rp, ci, v = laplace_block_system(8, 8, 6, seed=7)
    b = np.random.default_rng(1).standard_normal(Nb * 3)
    x0, r0 = orc.solve(Nb, rp, ci, v, b, tol=1e-10, maxit=200, w=0.9)
    for kind in ("graph_coloring", "graph_coloring_greedy"):
        x1, r1 = orc.solve(Nb, rp, ci, v, b, tol=1e-10, maxit=200, w=0.9, reorder=kind)
        assert r1.converged
        np.testing.assert_allclose(x1, x0, rtol=1e-6, atol=1e-9)


def test_block_jacobi_subdomains(orc):
    Nb, rp, ci, v = laplace_block_system(8, 8, 8, seed=9)
    b = np.random.default_rng(2).standard_normal(Nb * 3)
    x0, r0 = orc.solve(Nb, rp, ci, v, b, tol=1e-10, maxit=200)
    sub = np.linspace(0, Nb, 5).astype(np.int32)
    x1, r1 = orc.solve(Nb, rp, ci, v, b, tol=1e-10, maxit=200, sub_start=sub)
    assert r1.converged and r1.it >= r0.it
    np.testing.assert_allclose(x1, x0, rtol=1e-6, atol=1e-9)


def test_zero_diagonal_fix(orc):
    Nb, rp, ci, v = random_block_system(4, pattern="tridiag", seed=4)
    v = v.copy()
    dk = [k for i in range(Nb) for k in range(rp[i], rp[i + 1]) if ci[k] == i]
    v[dk[1] * 9 + 4] = 0.0
    n = orc.lib.orc_check_zero_diagonal(Nb, rp, ci, v)
    assert n == 1 and v[dk[1] * 9 + 4] == 1e-15


def test_wells_apply_matches_dense(orc):
    rng = np.random.default_rng(11)
    Nb = 30
    nw, perfs = 3, [4, 1, 6]
    vp = np.concatenate([[0], np.cumsum(perfs)]).astype(np.int32)
    nperf = int(vp[-1])
    cols = np.concatenate([rng.choice(Nb, p, replace=False) for p in perfs]).astype(np.int32)
    W = dict(numWells=nw, val_pointers=vp, Ccols=cols, Bcols=cols.copy(),
             Cnnzs=rng.standard_normal(nperf * 12), Bnnzs=rng.standard_normal(nperf * 12),
             Dnnzs=rng.standard_normal(nw * 16))
    x = rng.standard_normal(Nb * 3)
    y0 = rng.standard_normal(Nb * 3)
    y = orc.wells_apply(W, x, y0)
    ref = y0.copy().reshape(Nb, 3)
    for w in range(nw):
        B = W["Bnnzs"].reshape(-1, 4, 3)[vp[w]:vp[w + 1]]
        Cm = W["Cnnzs"].reshape(-1, 4, 3)[vp[w]:vp[w + 1]]
        D = W["Dnnzs"].reshape(-1, 4, 4)[w]
        c = cols[vp[w]:vp[w + 1]]
        z1 = sum(B[p] @ x.reshape(Nb, 3)[c[p]] for p in range(len(c)))
        z2 = D @ z1
        for p in range(len(c)):
            ref[c[p]] -= Cm[p].T @ z2
    np.testing.assert_allclose(y, ref.reshape(-1), rtol=1e-12, atol=1e-12)


def test_greedy_colouring_properties_of_test_graphcoloring(orc):
    """tests/test_graphcoloring.cpp:44-110 pins the reference's colouring through properties on a 10 x 10 five-point
    grid: two colours, checkerboard, and a renumbering that lists colour 0 first and keeps the natural order inside a
    colour (reorderVerticesPreserving).  The greedy colouring used for the red-black ILU0 ordering has all three."""
    N = 10
    rows, cols = [0], []
    for j in range(N):
        for i in range(N):
            idx = j * N + i
            nb = [idx]
            if i > 0: nb.append(idx - 1)
            if i < N - 1: nb.append(idx + 1)
            if j > 0: nb.append(idx - N)
            if j < N - 1: nb.append(idx + N)
            cols += sorted(nb)
            rows.append(len(cols))
    rp, ci = np.array(rows, np.int32), np.array(cols, np.int32)
    to, fr, rpc = orc.reorder(N * N, rp, ci, "graph_coloring_greedy")
    assert list(rpc) == [50, 50]                                   # noColors == 2, verticesPerColor
    color = (to >= 50).astype(int)
    first = color[0]
    for j in range(N):
        for i in range(N):
            assert color[j * N + i] == (first + i + j) % 2          # checkerboard
    assert sorted(to) == list(range(N * N))                        # checkAllIndices
    nxt = [0, 50]
    for v in range(N * N):                                         # colorIndex[colors[vertex]]++ == newOrder[vertex]
        assert to[v] == nxt[color[v]]
        nxt[color[v]] += 1
    assert np.array_equal(fr[to], np.arange(N * N))


def test_well_matrix_contribution_uses_the_transposed_product_of_test_multmatrixtransposed(orc):
    """tests/test_multmatrixtransposed.cpp:47-58 pins Detail::multMatrixTransposed (A^T B) on a 3 x 3 example; the matrix
    form of the well contributions is -C^T (D^-1 B) built from it (wells/StandardWell_impl.hpp:1688-1712).  With D = I
    and the example's A, B as the first three rows of C and B, the one block of a one-cell system must be -A^T B."""
    a3 = np.array([[1, 2, 3], [3, 4, 5], [6, 7, 8]], float)
    b3 = np.array([[3, 4, 5], [5, 6, 7], [7, 8, 9]], float)
    expect = np.array([[60, 70, 80], [75, 88, 101], [90, 106, 122]], float)
    assert np.array_equal(a3.T @ b3, expect)
    W = dict(numWells=1, val_pointers=np.array([0, 1], np.int32), Ccols=np.array([0], np.int32), Bcols=np.array([0], np.int32),
             Cnnzs=np.ascontiguousarray(np.vstack([a3, np.zeros((1, 3))]).reshape(-1)),
             Bnnzs=np.ascontiguousarray(np.vstack([b3, np.zeros((1, 3))]).reshape(-1)), Dnnzs=np.eye(4).reshape(-1).copy())
    rc, v = orc.wells_add_to_matrix(1, np.array([0, 1], np.int32), np.array([0], np.int32), np.zeros(9), W)
    assert rc == 0 and np.array_equal(v.reshape(3, 3), -expect)
