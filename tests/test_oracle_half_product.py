"""The oracle's statement of the half-product form (oracle/linalg.hpp: ilu0_apply_u, spmv_rest, is_upper_alias) - the order of
operations libopmhip's opmhip_config.half_product uses inside ILU0-BiCGStab - held by what the reference's own data can hold it with:

* matr33 / rhs3 and the exact solution tests/test_flexiblesolver.cpp:114-116 pins (a block-tridiagonal matrix has no triangles: the
  property the form rests on holds there);
* the identity behind tests/test_milu.cpp:41-100 read for this form: U == upper(A) bit for bit where no elimination step touches an
  entry right of the diagonal, L U e == A e;
* and against the plain form (ilu0_apply, spmv: linalg/ParallelOverlappingILU0.hpp:848-903 followed by the whole product, as
  bda/cusparseSolverBackend.cu:103-118 runs them): M^-1 d is the SAME bits - the sweep's statements are untouched -, the product differs
  by the order of a row's additions only, the solve stops on the same half iteration.
CPU only."""
import os

import numpy as np
import pytest

from helpers import laplace_block_system, random_block_system


def _load(pkg, golden, mat, rhs):
    Nb, rp, ci, v, bs = pkg.mmio.read_block_matrix(os.path.join(golden, "linalg", mat))
    b = pkg.mmio.read_block_vector(os.path.join(golden, "linalg", rhs))
    return Nb, rp, ci, v, b


def _upper_equal(Nb, rp, ci, a, lu):
    a, lu = a.reshape(-1, 9), lu.reshape(-1, 9)
    for i in range(Nb):
        for k in range(rp[i], rp[i + 1]):
            if ci[k] > i and not np.array_equal(a[k], lu[k]):
                return False
    return True


def test_matr33_exact_solution_in_the_half_product_form(pkg, orc, golden):
    """the reference's own system: block ILU0 of a block-tridiagonal matrix is an exact LU, the first half iteration lands on the solution
    test_flexiblesolver.cpp pins - in this form as in the plain one"""
    import json
    with open(os.path.join(golden, "linalg", "expected.json")) as f:
        e = json.load(f)["exact_noprec_tol1e-12_maxit200"]
    Nb, rp, ci, v, b = _load(pkg, golden, e["matrix"], e["rhs"])
    x, res = orc.solve(Nb, rp, ci, v, b, tol=1e-10, maxit=20, w=1.0, half_product=True)
    x0, res0 = orc.solve(Nb, rp, ci, v, b, tol=1e-10, maxit=20, w=1.0)
    assert res.converged and res.it == res0.it == 0.5
    ref = np.array(e["x"])
    for xi, ri in zip(x, ref):   # the printed expectations carry 5-6 digits (tests/test_oracle_linalg.py: _cmp)
        assert abs(xi - ri) <= 2e-4 * abs(ri), (xi, ri)
    np.testing.assert_allclose(x, x0, rtol=1e-12)
    # ... and with the relaxation the CPU path runs with (w = 0.9: no longer exact, a few iterations), both forms, the same count
    x2, r2 = orc.solve(Nb, rp, ci, v, b, tol=1e-12, maxit=200, w=0.9, half_product=True)
    x3, r3 = orc.solve(Nb, rp, ci, v, b, tol=1e-12, maxit=200, w=0.9)
    assert r2.converged and r2.it == r3.it
    np.testing.assert_allclose(x2, x0, rtol=1e-6)


@pytest.mark.parametrize("shape", [(7, 6, 5), (12, 10, 6)])
def test_u_is_upper_a_and_the_product_is_the_same_sum(orc, shape):
    """7-point grids: the factor's strict upper part equals the matrix's bit for bit; M^-1 d of the two forms is identical; the two
    products differ by rounding of a row's additions only (1e-15 of the row's magnitude); L U e == A e (test_milu.cpp's identity)"""
    Nb, rp, ci, v = laplace_block_system(*shape, seed=5)
    lu = orc.ilu0_factor(Nb, rp, ci, v)
    assert _upper_equal(Nb, rp, ci, v, lu)
    rng = np.random.default_rng(3)
    for mode in ("post_scale", "in_sweep"):
        d = rng.standard_normal(3 * Nb)
        t0, z0 = orc.preconditioned_product(Nb, rp, ci, v, lu, d, w=0.9, mode=mode)
        t1, z1 = orc.preconditioned_product(Nb, rp, ci, v, lu, d, w=0.9, mode=mode, half_product=True)
        assert np.array_equal(z0, z1)
        absrow = orc.spmv(Nb, rp, ci, np.abs(v), np.abs(z0))
        assert np.all(np.abs(t1 - t0) <= 4e-15 * absrow)
        assert not np.array_equal(t0, t1) or Nb < 10     # (it IS another order: some row rounds differently)
    # w = 1: M = L U exactly reproduces A on the pattern's diagonal ... the identity of test_milu.cpp: (L U) e == A e needs the full ILU
    # product; here its consequence for this form: with d = A e, M^-1 d == e to rounding only where ILU0 is exact (1-D chains) - so check
    # the chain case
    Nb, rp, ci, v = laplace_block_system(9, 1, 1, seed=7)
    lu = orc.ilu0_factor(Nb, rp, ci, v)
    e = np.ones(3 * Nb)
    d = orc.spmv(Nb, rp, ci, v, e)
    t1, z1 = orc.preconditioned_product(Nb, rp, ci, v, lu, d, w=1.0, half_product=True)
    np.testing.assert_allclose(z1, e, rtol=1e-12)
    np.testing.assert_allclose(t1, d, rtol=1e-12)


def test_solves_agree_on_assembled_jacobians(pkg, orc):
    """black-oil Jacobians (heterogeneous 10 x 9 x 8, both states): the two forms stop on the same half iteration and agree in x to the
    rounding a different order of additions leaves after a dozen iterations"""
    import oracle_bind
    for state in ("mixed", "saturated"):
        case = pkg.decks.cartesian_case(10, 9, 8, state=state, heterogeneous=True)
        o = oracle_bind.OracleModel(orc, case)
        o.set_state(case["pv"], case["meaning"])
        o.set_source(pkg.decks.five_spot_source(case, rate_sm3_per_day=50.0))
        jac, res = o.assemble(10 * 86400.0, 0)
        Nb = case["Nb"]
        for tol in (1e-2, 1e-8):
            x0, r0 = orc.solve(Nb, case["rowptr"], case["col"], jac, res, tol=tol, maxit=200, w=0.9)
            x1, r1 = orc.solve(Nb, case["rowptr"], case["col"], jac, res, tol=tol, maxit=200, w=0.9, half_product=True)
            assert r0.converged and r1.converged and r0.it == r1.it
            np.testing.assert_allclose(x1, x0, rtol=1e-7, atol=1e-10 * np.abs(x0).max())
            # the true residual of the half-product solution meets the tolerance the recurrence reported
            rr = res - orc.spmv(Nb, case["rowptr"], case["col"], jac, x1)
            assert np.linalg.norm(rr) <= 1.001 * max(r1.reduction, 1e-12) * np.linalg.norm(res) + 1e-9 * np.linalg.norm(res)


def test_patterns_with_triangles_are_refused(orc):
    """a random graph has triangles: elimination steps touch entries right of the diagonal, U != upper(A), the form does not apply and
    the oracle says so instead of computing something else"""
    Nb, rp, ci, v = random_block_system(60, "random", seed=11, extra=5)
    lu = orc.ilu0_factor(Nb, rp, ci, v)
    assert not _upper_equal(Nb, rp, ci, v, lu)
    rc = orc.lib.orc_preconditioned_product(Nb, rp, ci, v, lu, np.zeros(3 * Nb), 0.9, 0, 1, np.zeros(3 * Nb), np.zeros(3 * Nb))
    assert rc == -1000


def test_fused_reductions_recurrence(pkg, orc, golden):
    """BiCGStab with one reduction per half iteration (oracle/linalg.hpp: bicgstab_fused_reductions; opmhip_config.fused_reductions): on the
    reference's matr33 / rhs3 it lands on the pinned exact solution in the first half iteration like the reference's recurrence; on
    Jacobians it stops within half an iteration of it down to 1e-6, reports the TRUE residual norm of its iterate, and that iterate solves
    the system to the tolerance"""
    import json
    import oracle_bind
    with open(os.path.join(golden, "linalg", "expected.json")) as f:
        e = json.load(f)["exact_noprec_tol1e-12_maxit200"]
    Nb, rp, ci, v, b = _load(pkg, golden, e["matrix"], e["rhs"])
    x, res = orc.solve(Nb, rp, ci, v, b, tol=1e-6, maxit=20, w=1.0, fused_reductions=True)
    assert res.converged and res.it == 0.5
    for xi, ri in zip(x, np.array(e["x"])):
        assert abs(xi - ri) <= 2e-4 * abs(ri)
    case = pkg.decks.cartesian_case(10, 9, 8, state="mixed", heterogeneous=True)
    o = oracle_bind.OracleModel(orc, case)
    o.set_state(case["pv"], case["meaning"])
    o.set_source(pkg.decks.five_spot_source(case, rate_sm3_per_day=50.0))
    jac, r = o.assemble(10 * 86400.0, 0)
    Nb = case["Nb"]
    for tol in (1e-2, 1e-4, 1e-6):
        for hp in (False, True):
            x0, r0 = orc.solve(Nb, case["rowptr"], case["col"], jac, r, tol=tol, maxit=200, w=0.9, half_product=hp)
            x1, r1 = orc.solve(Nb, case["rowptr"], case["col"], jac, r, tol=tol, maxit=200, w=0.9, half_product=hp, fused_reductions=True)
            assert r0.converged and r1.converged and abs(r0.it - r1.it) <= 0.5
            true = np.linalg.norm(r - orc.spmv(Nb, case["rowptr"], case["col"], jac, x1)) / np.linalg.norm(r)
            assert abs(true - r1.reduction) <= 1e-6 * r1.reduction + 1e-12 and true < 2.0 * tol       # the reported reduction is the iterate's own
            if r0.it == r1.it:
                np.testing.assert_allclose(x1, x0, rtol=1e-6, atol=1e-9 * np.abs(x0).max())
