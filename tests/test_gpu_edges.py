"""Degenerate sizes through the whole path on the GPU: a single cell (no faces), a pair, one column, one layer - where
tiles hold one row, colours hold one tile and chains have one link."""
import numpy as np
import pytest

import oracle_bind

pytestmark = pytest.mark.gpu
REORDERS = ["level_scheduling", "graph_coloring", "graph_coloring_greedy", "line_coloring"]


@pytest.mark.parametrize("reorder", REORDERS)
@pytest.mark.parametrize("shape", [(1, 1, 1), (2, 1, 1), (1, 1, 7), (3, 2, 1), (1, 33, 1), (2, 2, 2)])
def test_tiny_grids(pkg, orc, shape, reorder):
    case = pkg.decks.cartesian_case(*shape, state="mixed", heterogeneous=True)
    src = np.zeros((case["Nb"], 3))
    src[0, 1] = 1e-6
    src[-1, 0] = -1e-6
    src = np.ascontiguousarray(src.reshape(-1))
    m = pkg.capi.HipModel(case, reorder=reorder, tolerance=1e-2, maxit=200, ilu_relaxation=0.9)
    o = oracle_bind.OracleModel(orc, case)
    for q in (m, o):
        q.set_state(case["pv"], case["meaning"])
        q.set_source(src)
    dt = 86400.0
    assert np.array_equal(m.iq(), o.iq())
    jm, rm = m.assemble(dt, 0)
    jo, ro = o.assemble(dt, 0)
    assert np.array_equal(jm, jo) and np.array_equal(rm, ro)
    # MB is |sum of residuals| * dt * B / pv: pure cancellation noise where the sum vanishes (tolerance there is 1e-6)
    np.testing.assert_allclose(m.convergence(dt)[11:17], o.convergence(dt)[11:17], rtol=1e-10, atol=1e-14)
    res = m.solve_jacobian_system()
    xo, reso = o.solve_in_order(*m.ordering()[:2], tol=1e-2, maxit=200, w=0.9)
    assert res.converged == bool(reso.converged) and res.it == reso.it
    np.testing.assert_allclose(m.get_result(), xo, rtol=1e-9, atol=1e-12 * max(1.0, np.abs(xo).max()))
    m.update(None, 1.0)
    o.update(xo)
    pm, mm = m.get_state()
    po, mo = o.get_state()
    assert np.array_equal(mm, mo)
    np.testing.assert_allclose(pm, po, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("prec", ["cpr_quasiimpes", "cpr"])
@pytest.mark.parametrize("shape", [(1, 1, 1), (2, 1, 1), (1, 1, 7), (3, 2, 1), (1, 33, 1), (5, 4, 3), (40, 2, 1)])
def test_tiny_grids_with_cpr(pkg, orc, shape, prec):
    """the CPR preconditioner where the pressure hierarchy has one level of a handful of rows (dense solve at once), rows past the end
    of a wavefront's group of 32 (the stencil form of level 0's columns), chains of one link: the solve converges to the oracle's
    ILU0-preconditioned solution within the tolerance of both"""
    case = pkg.decks.cartesian_case(*shape, state="mixed", heterogeneous=True)
    src = np.zeros((case["Nb"], 3))
    src[0, 1] = 1e-6
    src[-1, 0] = -1e-6
    src = np.ascontiguousarray(src.reshape(-1))
    m = pkg.capi.HipModel(case, reorder="line_coloring", tolerance=1e-8, maxit=200, preconditioner=prec, cpr_amg_ilu_levels=0)
    o = oracle_bind.OracleModel(orc, case)
    for q in (m, o):
        q.set_state(case["pv"], case["meaning"])
        q.set_source(src)
    dt = 86400.0
    jm, rm = m.assemble(dt, 0)
    jo, ro = o.assemble(dt, 0)
    assert np.array_equal(jm, jo) and np.array_equal(rm, ro)
    res = m.solve_jacobian_system()
    assert res.converged
    x = m.get_result()
    r = orc.spmv(case["Nb"], case["rowptr"], case["col"], jo, x) - ro
    assert np.linalg.norm(r) <= 1e-8 * np.linalg.norm(ro) * 1.001
    # the preconditioner application itself: the oracle's, bit for bit, in the device's ordering
    to, fr, _ = m.ordering()
    rr, rc, rv = orc.reorder_matrix(case["Nb"], case["rowptr"], case["col"], jo, to, fr)
    cpr = oracle_bind.OracleCpr(orc)
    cpr.set_natural_ids(fr)
    if prec == "cpr":
        cpr.set_weights(np.ascontiguousarray(o.true_impes_weights(dt)[fr].reshape(-1)))
    cpr.update(case["Nb"], rr, rc, rv)
    d = np.random.default_rng(3).standard_normal(3 * case["Nb"])
    vo = cpr.apply(np.ascontiguousarray(d.reshape(-1, 3)[fr].reshape(-1))).reshape(-1, 3)[to].reshape(-1)
    assert np.array_equal(m.cpr_apply(d), vo)


def test_call_order_and_refusals_of_round_5_entry_points(pkg):
    """opmhip_get_ordering_info before a pattern is set is a call-order violation (OPMHIP_NOT_READY), not garbage; a decomposed context
    refuses multisegment wells through the host callback (OPMHIP_INVALID_ARGUMENT, so that Flow falls back to Dune,
    ISTLSolverEbos.hpp:277-297) instead of applying a rank-local operator silently; a level-scheduled context reports its levels as colours"""
    import uuid
    s = pkg.capi.HipSolver()
    with pytest.raises(pkg.capi.OpmHipError) as e:
        s.ordering_info()
    assert e.value.code == pkg.capi.NOT_READY
    case = pkg.ras.cartesian_subdomain_case(6, 2, 0, state="mixed", heterogeneous=False)
    m = pkg.capi.HipModel(case, comm=("loopback", 2, 0, "edge" + uuid.uuid4().hex), reorder="level_scheduling")
    info = m.ordering_info()
    assert info["ilu_ordering"] == "level_scheduling" and info["chain_length"] == 0 and info["colors"] == len(m.ordering()[2]) and info["cpr_amg_ilu_levels"] == 0
    m.set_state(case["pv"], case["meaning"])
    m.set_source(case["source"])
    m.assemble(86400.0, 0, fetch=False)
    wells = dict(numWells=0, numMsWells=1, ms_apply=lambda x, y: None, N=3 * case["Nb"])
    with pytest.raises(pkg.capi.OpmHipError) as e:
        m.solve_jacobian_system(wells=wells)     # refused while the wells are taken over: before any collective of the solve
    assert e.value.code == pkg.capi.INVALID_ARGUMENT and "decomposed" in str(e.value)
