"""The oracle's CPR preconditioner (oracle/cpr.hpp): pinned where the reference's own test can pin it - CPR-BiCGStab on
tests/matr33.txt gives the exact solution tests/test_flexiblesolver.cpp:93-116 expects - plus the structural properties
of the restatement (quasi-IMPES weights, pressure system, AMG hierarchy) and its effect on a real Jacobian.  CPU only."""
import json
import os

import numpy as np

import oracle_bind
from test_oracle_linalg import _cmp, _load


def test_cpr_on_matr33_gives_the_flexiblesolver_vector(pkg, orc, golden):
    """options_flexiblesolver.json: bicgstab, tol 0.5, maxiter 20, preconditioner cpr (quasi-IMPES weights, ILU0 fine
    smoother, relaxation 1).  Block ILU0 is an exact LU of this block-tridiagonal matrix, so one post-smoothing step on the
    residual left by the pressure correction lands on the exact solution, whatever the coarse solver did."""
    Nb, rp, ci, v, b = _load(pkg, golden, "matr33.txt", "rhs3.txt")
    with open(os.path.join(golden, "linalg", "expected.json")) as f:
        e = json.load(f)["exact_noprec_tol1e-12_maxit200"]   # the same vector test_flexiblesolver.cpp:93-116 expects of the CPR run
    cpr = oracle_bind.OracleCpr(orc)
    x, res = cpr.solve(Nb, rp, ci, v, b, tol=0.5, maxit=20, zero_diag_fix=False)
    assert res.converged and res.it == 0.5
    _cmp(x, e)


def test_weights_and_pressure_system(pkg, orc):
    case = pkg.decks.cartesian_case(12, 10, 8, state="mixed", heterogeneous=True)
    o = oracle_bind.OracleModel(orc, case)
    o.set_state(case["pv"], case["meaning"])
    jac, res = o.assemble(5 * 86400.0, 0)
    Nb, rp, ci = case["Nb"], case["rowptr"], case["col"]
    cpr = oracle_bind.OracleCpr(orc)
    cpr.update(Nb, rp, ci, jac)
    w = cpr.weights(Nb)
    blk = jac.reshape(-1, 3, 3)
    row = np.repeat(np.arange(Nb), np.diff(rp))
    D = blk[ci == row]
    # getQuasiImpesWeights.hpp:46-85: D^T w = e_p up to the max-norm scaling
    e = np.einsum("nji,nj->ni", D, w)
    e /= e[:, 1:2]
    np.testing.assert_allclose(e, np.tile([0.0, 1.0, 0.0], (Nb, 1)), atol=1e-9)
    np.testing.assert_allclose(np.abs(w).max(axis=1), 1.0, rtol=1e-14)
    n, nnz = cpr.levels()
    assert n[0] == Nb and nnz[0] == len(ci) and n[-1] <= 128 and all(a > b for a, b in zip(n, n[1:]))
    agg = cpr.aggregates(0, Nb)
    cnt = np.bincount(agg)
    assert cnt.min() >= 1 and cnt.max() <= 4 and len(cnt) == n[1]
    # the preconditioner is a linear operator
    rng = np.random.default_rng(0)
    d1, d2 = rng.standard_normal(3 * Nb), rng.standard_normal(3 * Nb)
    np.testing.assert_allclose(cpr.apply(2.0 * d1 - d2), 2.0 * cpr.apply(d1) - cpr.apply(d2), rtol=1e-9, atol=1e-12 * np.abs(cpr.apply(d1)).max())


def test_cpr_needs_fewer_iterations_than_ilu0_on_a_stiff_step(pkg, orc):
    """against the ILU0 of the accelerator path's default ordering (graph colouring, bda/BdaBridge.cpp:72-73) - the
    natural-order ILU0 of a grid this small is too good a preconditioner to lose against"""
    case = pkg.decks.cartesian_case(30, 30, 24, state="mixed", heterogeneous=True)
    src = pkg.decks.five_spot_source(case, rate_sm3_per_day=40.0)
    o = oracle_bind.OracleModel(orc, case)
    o.set_state(case["pv"], case["meaning"])
    o.set_source(src)
    Nb, rp, ci = case["Nb"], case["rowptr"], case["col"]
    cpr = oracle_bind.OracleCpr(orc)
    dt = 10 * 86400.0
    its = []
    for it in range(3):
        jac, res = o.assemble(dt, it)
        x0, r0 = o.solve(tol=1e-2, reorder="graph_coloring")
        x1, r1 = cpr.solve(Nb, rp, ci, jac, res, tol=1e-2)
        assert r0.converged and r1.converged
        # both satisfy the stopping rule on the same system
        A = None
        its.append((r0.it, r1.it))
        o.update(x0)
    assert sum(c for _, c in its) < sum(i for i, _ in its), its


def test_true_impes_weights_solve_their_defining_system(orc):
    """getTrueImpesWeights (linalg/getQuasiImpesWeights.hpp:89-128): block^T (1000 w) = e_p with block = d storage / d x scaled
    by dt / V and the pressure column by 50e5 - checked against numpy on the storage derivatives of an assembled state"""
    import importlib
    pkg = importlib.import_module("opm-autodiff_amd")
    case = pkg.decks.cartesian_case(6, 5, 7, state="mixed", heterogeneous=True)
    o = oracle_bind.OracleModel(orc, case)
    o.set_state(case["pv"], case["meaning"])
    dt = 3 * 86400.0
    o.assemble(dt, 0)
    w = o.true_impes_weights(dt)
    iq = o.iq()                                          # (Nb, 17, 4): S 0-2, 1/B 6-8, Rs 15, porosity 16
    S, B, Rs, poro = iq[:, 0:3], iq[:, 6:9], iq[:, 15], iq[:, 16]
    def mul(a, b):                                       # AD product on (value, 3 derivatives)
        return np.concatenate([(a[:, :1] * b[:, :1]), a[:, 1:] * b[:, :1] + b[:, 1:] * a[:, :1]], axis=1)
    sv = [mul(mul(S[:, ph], B[:, ph]), poro) for ph in range(3)]       # water, oil, gas
    st = {0: sv[1], 1: sv[0], 2: sv[2] + mul(Rs, sv[1])}               # equations: oil, water, gas
    for c in range(0, case["Nb"], 17):
        block = np.array([[st[e][c, 1 + v] for v in range(3)] for e in range(3)]) / (case["volume"][c] / dt)
        block[:, 1] *= 50e5
        x = np.linalg.solve(block.T, np.array([0.0, 1.0, 0.0])) / 1000.0
        np.testing.assert_allclose(w[c], x, rtol=1e-9, atol=1e-300)


def test_reference_like_amg_hierarchy_and_iteration_counts(pkg, orc):
    """The restatement of the reference's pressure AMG (Dune::Amg-like aggregation: alpha 1/3, beta 1e-5, aggregates of 4-6,
    maxDistance 2; ILU0 smoothing; direct coarse solve; oracle/cpr.hpp DuneLikeAmg) as a yardstick for the product's own
    hierarchy: aggregates have the prescribed sizes, the preconditioner is a linear operator that reduces the residual, and
    CPR-BiCGStab needs about as many iterations with either hierarchy on a black-oil Jacobian."""
    case = pkg.decks.cartesian_case(20, 20, 16, state="mixed", heterogeneous=True)
    o = oracle_bind.OracleModel(orc, case)
    o.set_state(case["pv"], case["meaning"])
    o.set_source(pkg.decks.five_spot_source(case, rate_sm3_per_day=40.0))
    jac, res = o.assemble(5 * 86400.0, 0)
    Nb, rp, ci = case["Nb"], case["rowptr"], case["col"]
    its = {}
    for ref in (False, True):
        cpr = oracle_bind.OracleCpr(orc)
        cpr.use_reference_amg(ref)
        x, r = cpr.solve(Nb, rp, ci, jac, res, tol=1e-2, maxit=100)
        assert r.converged
        its[ref] = r.it
        if ref:
            n, nnz = cpr.reference_amg_levels()
            assert n[0] == Nb and n[-1] <= 1200 and len(n) >= 2
            rates = [n[l] / n[l + 1] for l in range(len(n) - 1)]
            assert all(3.0 <= q <= 6.5 for q in rates), (n, rates)      # aggregates of 4 to 6 vertices
            d1, d2 = np.random.default_rng(1).standard_normal((2, 3 * Nb))
            cpr.update(Nb, rp, ci, jac)
            assert np.allclose(cpr.apply(2.0 * d1 - 0.5 * d2), 2.0 * cpr.apply(d1) - 0.5 * cpr.apply(d2), rtol=1e-9, atol=1e-12 * np.abs(d1).max())
    print("CPR-BiCGStab iterations to 1e-2: product's AMG %.1f, reference-like AMG %.1f" % (its[False], its[True]))
    assert its[True] <= its[False] + 3 and its[False] <= 2 * its[True] + 3


def test_decomposed_cpr_with_a_joined_coarse_level(pkg, orc):
    """Decomposed CPR (orc_cpr_solve_blocks): with one hierarchy per subdomain and nothing between them the iteration count grows
    with the number of subdomains; with the hierarchies continued on the joined system from a small level on (gather_rows) it stays
    close to the single-domain count - the coarse part of the reference's parallel AMG (OwningTwoLevelPreconditioner.hpp).  Also: the
    joined level is what it claims to be - its rows sum like the Galerkin product of the global pressure matrix (constant vectors)."""
    n, world = 10, 8
    px, py, pz = pkg.ras.block_layout(world)
    g = pkg.decks.cartesian_case(px * n, py * n, pz * n, state="mixed", heterogeneous=True)
    owner = np.asarray(pkg.ras.cartesian_owner(px * n, py * n, pz * n, px, py, pz), np.int32)
    src = pkg.decks.five_spot_source(g, rate_sm3_per_day=60.0)
    o = oracle_bind.OracleModel(orc, g)
    o.set_state(g["pv"], g["meaning"])
    o.set_source(src)
    jac, res = o.assemble(20 * 86400.0, 0)
    Nb, rp, ci = g["Nb"], g["rowptr"], g["col"]
    cpr = oracle_bind.OracleCpr(orc)
    _, one = cpr.solve(Nb, rp, ci, jac, res, tol=1e-6)
    _, alone, lev = orc.cpr_solve_blocks(Nb, rp, ci, jac, res, owner, tol=1e-6)
    probe = np.random.default_rng(3).standard_normal(3 * Nb)
    its = {}
    for rows in (64, 300, 5000):   # joined from the third level / the second / level 0 itself (10^3 = 1000 cells per subdomain)
        x, r, lev, glev, pv = orc.cpr_solve_blocks(Nb, rp, ci, jac, res, owner, tol=1e-6, gather_rows=rows, probe=probe)
        assert r.converged and np.linalg.norm(orc.spmv(Nb, rp, ci, jac, x) - res) <= 2e-6 * np.linalg.norm(res)
        assert len(glev) >= 2 and glev[0] <= world * rows and np.isfinite(pv).all()
        its[rows] = r.it
    assert lev.max() == 1 and glev[0] == Nb        # gather_rows above the subdomain size: level 0 itself is joined
    assert one.converged and alone.converged
    # nothing between the subdomains costs iterations; the joined level takes (most of) that back
    assert alone.it >= 1.3 * one.it, (one.it, alone.it)
    for rows, it in its.items():
        assert it <= 1.3 * one.it + 1, (rows, it, one.it, alone.it)
