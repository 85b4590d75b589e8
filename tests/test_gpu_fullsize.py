"""BASELINE.json's full size (configs[1]: 100 x 100 x 100, 10^6 cells, 6.94 * 10^6 blocks) on the GPU, through
size-independent properties (this file calls no oracle; the bit-for-bit comparison with the oracle AT this size - Jacobian, residual,
factors, one product, one M^-1, the solve's half iteration - is tests/test_gpu_fullsize_oracle.py, about 10 s of oracle time per ordering):
  SpMV          linearity, and A * 1 against row sums formed independently with numpy;
  ILU0          M^-1 (L (U z)) = z with L, U taken from the factors the device reports (w = 1);
  BiCGStab      the returned x satisfies ||b - A x|| <= tol ||b|| with A x formed by scipy on the host;
  assembly      at iteration 0 the storage term vanishes, so the cell residuals sum to minus the sources (fluxes cancel
                pairwise); J is the derivative of R: R(x + eps d) - R(x) = eps J d + O(eps^2);
  update        a zero Newton update and a rolled-back time step leave the state bit-identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
N = 100
DT = 10 * 86400.0


@pytest.fixture(scope="module")
def full(pkg, case100):
    case, src = case100["case"], case100["src"]      # the session's one 100^3 case (tests/conftest.py)
    m = pkg.capi.HipModel(case, reorder="line_coloring", tolerance=1e-2, maxit=200, ilu_relaxation=1.0)
    m.set_state(case["pv"], case["meaning"])
    m.set_source(src)
    jac, res = m.assemble(DT, 0)
    return dict(case=case, src=src, m=m, jac=jac, res=res)


def host_matrix(case, jac):
    import scipy.sparse as sp
    Nb = case["Nb"]
    return sp.bsr_matrix((jac.reshape(-1, 3, 3), case["col"], case["rowptr"]), shape=(3 * Nb, 3 * Nb))


def test_spmv_linearity_and_row_sums(full):
    m, case, jac = full["m"], full["case"], full["jac"]
    Nb = case["Nb"]
    rng = np.random.default_rng(1)
    x, y = rng.standard_normal(3 * Nb), rng.standard_normal(3 * Nb)
    a, b = 0.75, -1.5
    Ax, Ay, Az = m.spmv(x), m.spmv(y), m.spmv(a * x + b * y)
    scale = np.abs(a * Ax) + np.abs(b * Ay) + 1e-300
    assert np.max(np.abs(Az - (a * Ax + b * Ay)) / scale) < 1e-12
    # A * 1 = row sums: per block row, the sum over its blocks of the block's row sums (numpy, float64, any order)
    rs = np.add.reduceat(jac.reshape(-1, 3, 3).sum(axis=2), case["rowptr"][:-1], axis=0).reshape(-1)
    mag = np.add.reduceat(np.abs(jac.reshape(-1, 3, 3)).sum(axis=2), case["rowptr"][:-1], axis=0).reshape(-1)
    assert np.max(np.abs(m.spmv(np.ones(3 * Nb)) - rs) / mag) < 1e-13


def test_ilu0_apply_inverts_its_own_factors(full):
    m, case = full["m"], full["case"]
    Nb = case["Nb"]
    lu = m.ilu0_factor().reshape(-1, 3, 3)          # strict lower = L, diagonal = D^-1, strict upper = U, device's order
    to, fr, rpc = m.ordering()
    # pattern of the reordered matrix: rebuilt on the host from the permutation (reorderBlockedMatrixByPattern)
    import scipy.sparse as sp
    rowp, col = case["rowptr"], case["col"]
    lens = np.diff(rowp)[fr]
    rrp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    # entries of reordered row p = entries of natural row fr[p], columns renamed by `to`, sorted
    start = np.repeat(rowp[fr], lens)
    within = np.arange(rrp[-1]) - np.repeat(rrp[:-1], lens)
    nat_entry = start + within
    rcol_unsorted = to[col[nat_entry]]
    rrow = np.repeat(np.arange(Nb), lens)
    order = np.lexsort((rcol_unsorted, rrow))
    rcol = rcol_unsorted[order]
    lower, upper, diag = rcol < rrow, rcol > rrow, rcol == rrow
    D = np.linalg.inv(lu[diag])                      # the factors hold D^-1
    z = np.random.default_rng(2).standard_normal((Nb, 3))
    U = sp.bsr_matrix((lu[upper], rcol[upper], np.concatenate([[0], np.cumsum(np.bincount(rrow[upper], minlength=Nb))])), shape=(3 * Nb, 3 * Nb))
    L = sp.bsr_matrix((lu[lower], rcol[lower], np.concatenate([[0], np.cumsum(np.bincount(rrow[lower], minlength=Nb))])), shape=(3 * Nb, 3 * Nb))
    w = np.einsum("nij,nj->ni", D, z) + (U @ z.reshape(-1)).reshape(Nb, 3)     # (D + U) z
    v = w + (L @ w.reshape(-1)).reshape(Nb, 3)                                     # (I + L) w
    # v lives in the device's order; the API speaks the natural order
    back = m.ilu0_apply(np.ascontiguousarray(v[to].reshape(-1))).reshape(Nb, 3)[fr]
    # the 3 x 3 diagonal blocks mix pressure and saturation columns (condition ~1e7): numpy's D = inv(D^-1) alone costs
    # that many digits, so the round trip is good to ~1e-8 relative; a wrong sweep would give O(1)
    assert np.max(np.abs(back - z)) / np.max(np.abs(z)) < 1e-6


def test_solve_residual_on_the_host(full):
    m, case, jac, res = full["m"], full["case"], full["jac"], full["res"]
    r = m.solve_jacobian_system()
    x = m.get_result()
    assert r.converged and 1 <= r.iterations <= 200
    A = host_matrix(case, jac)
    assert np.linalg.norm(res - A @ x) <= 1e-2 * np.linalg.norm(res) * (1 + 1e-6)
    assert abs(r.reduction - np.linalg.norm(res - A @ x) / np.linalg.norm(res)) < 1e-3 * r.reduction + 1e-12


def test_assembly_conserves_mass_and_matches_its_jacobian(full, pkg):
    m, case, src, jac, res = full["m"], full["case"], full["src"], full["jac"], full["res"]
    Nb = case["Nb"]
    R = res.reshape(Nb, 3)
    # iteration 0: storage term is zero, R_I = sum_faces flux - q_I, and fluxes cancel pairwise
    tot = R.sum(axis=0) + src.reshape(Nb, 3).sum(axis=0)
    assert np.all(np.abs(tot) < 1e-9 * np.abs(R).sum(axis=0))
    # directional derivative: perturb pressures and saturations a little (no cell changes its meaning)
    rng = np.random.default_rng(3)
    d = rng.standard_normal((Nb, 3))
    pv0 = case["pv"].reshape(Nb, 3)
    scale = np.array([1e-7, 10.0, 1e-7])             # dSw, dp [Pa], dSg / dRs-sized steps
    scale_rs = np.where(case["meaning"] == pkg.decks.SW_PO_RS, 1e-5, 1e-7)
    step = d * scale
    step[:, 2] = d[:, 2] * scale_rs
    m.set_state(np.ascontiguousarray((pv0 + step).reshape(-1)), case["meaning"])
    _, res1 = m.assemble(DT, 1)                       # iteration 1: keeps the storage cache of iteration 0
    _, mean1 = m.get_state()
    assert np.array_equal(mean1, case["meaning"])
    A = host_matrix(case, jac)
    lin = A @ step.reshape(-1)
    diff = res1 - res
    err = np.linalg.norm(diff - lin) / np.linalg.norm(lin)
    assert err < 2e-3, err
    # restore for the tests that follow
    m.set_state(case["pv"], case["meaning"])
    m.assemble(DT, 0, fetch=False)


def test_per_cell_well_calls_at_full_size(full):
    """opmhip_set_source_cells / opmhip_get_iq_cells on 10^6 cells: the five-spot's 200 cells named one by one give the residual and the
    Jacobian of the whole-grid call bit for bit (the staging of the per-cell calls is sized by the grid, their traffic by the cells
    named); the records of cells from all over the grid equal the whole array's rows"""
    m, case, src, jac, res = full["m"], full["case"], full["src"], full["jac"], full["res"]
    Nb = case["Nb"]
    s3 = src.reshape(Nb, 3)
    cells = np.flatnonzero(np.abs(s3).sum(axis=1) > 0).astype(np.int32)
    assert len(cells) == 200
    m.set_source_cells(cells, np.ascontiguousarray(s3[cells].reshape(-1)))
    j1, r1 = m.assemble(DT, 0)
    assert np.array_equal(r1, res) and np.array_equal(j1, jac)
    rng = np.random.default_rng(5)
    pick = np.concatenate([rng.choice(Nb, 500, replace=False), [0, Nb - 1]]).astype(np.int32)
    assert np.array_equal(m.iq_cells(pick), m.iq()[pick])
    m.set_source(src)                                  # as the tests that follow expect it
    m.assemble(DT, 0, fetch=False)


def test_zero_update_and_roll_back_are_idempotent(full):
    m, case = full["m"], full["case"]
    m.advance_time_level()
    assert m.update(np.zeros(3 * case["Nb"]), 1.0) == 0
    pv, mean = m.get_state()
    assert np.array_equal(pv, case["pv"]) and np.array_equal(mean, case["meaning"])
    m.assemble(DT, 0, fetch=False)
    assert m.solve_jacobian_system().converged
    m.update(None, 1.0)
    m.update_failed()
    pv, mean = m.get_state()
    assert np.array_equal(pv, case["pv"]) and np.array_equal(mean, case["meaning"])


@pytest.mark.parametrize("prec", ["ilu0", "cpr"])
def test_two_subdomains_of_full_size(pkg, prec):
    """configs[3]'s building block: two 10^6-cell subdomains side by side through the loopback communicator - the sizes at
    which the decomposed run takes the pipelined SpMV with its separate scalar products and the two-stage local reduction
    (1 953 partials per colour).  No oracle at this size: both ranks must see the same Newton / linear iteration history,
    every linear solve must converge, and the residual of the owned rows must satisfy the stopping rule.  With `cpr` every
    rank runs the CPR of its own subdomain (true-IMPES weights, a seven-level hierarchy over 10^6 cells) and needs fewer
    linear iterations than with ILU0."""
    import threading
    import uuid
    world, n = 2, N
    group = "full" + uuid.uuid4().hex
    out, err = [None] * world, [None] * world

    def body(r):
        try:
            case = pkg.ras.cartesian_subdomain_case(n, world, r, state="mixed", heterogeneous=False)
            m = pkg.capi.HipModel(case, comm=("loopback", world, r, group), reorder="line_coloring", tolerance=1e-2, maxit=200, ilu_relaxation=0.9, preconditioner=prec, cpr_amg_ilu_levels=0)
            m.set_state(case["pv"], case["meaning"])
            m.set_source(case["source"])
            hist = []
            for it in range(3):
                m.assemble(86400.0, it, fetch=False)
                conv = m.convergence(86400.0)
                sol = m.solve_jacobian_system()
                hist.append((float(sol.it), bool(sol.converged), float(sol.reduction), tuple(np.round(conv[11:17], 12))))
                m.update(None, 1.0)
            out[r] = hist
        except BaseException as e:  # noqa: BLE001
            err[r] = e

    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    for e in err:
        if e is not None:
            raise e
    assert out[0] == out[1]                       # global scalars: identical on both ranks, bit for bit
    for it, ok, red, _ in out[0]:
        assert ok and red <= 1e-2 and 0.5 <= it <= (60 if prec == "ilu0" else 20)


def test_eight_subdomains_of_full_size_configs3(pkg):
    """BASELINE configs[3] at size: the 200 x 200 x 200 grid (8 * 10^6 cells) cut into 2 x 2 x 2 subdomains of 10^6 cells,
    every subdomain a context of its own on this one GPU, connected by the loopback communicator - the code path of the
    8-GPU run (set_pattern_dd, halo pack / exchange, interior / boundary products, local reductions + all-reduce) with device
    copies in the place of RCCL.  Driven by bench.py's own time-step control for the first Newton iterations.  No oracle at
    this size: all eight ranks must report the same history bit for bit, every linear solve must meet the stopping rule,
    and the iteration counts must stay where the rehearsal of round 2 found them (22.8 linear iterations per Newton
    iteration over 25 Newton iterations; 17.45 for one 10^6-cell domain, 26.8 for the 8 * 10^6-cell grid as one domain)."""
    import os
    import sys
    import threading
    import uuid
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    world, n, steps = 8, N, 8
    group = "cfg3" + uuid.uuid4().hex
    out, err = [None] * world, [None] * world

    def body(r):
        try:
            case = pkg.ras.cartesian_subdomain_case(n, world, r, state="mixed", heterogeneous=False)
            assert case["Nb"] == n ** 3 and case["Nghost"] == 3 * n * n and case["global_cells"] == 8 * n ** 3
            m = pkg.capi.HipModel(case, comm=("loopback", world, r, group), reorder="line_coloring", tolerance=1e-2, maxit=200, ilu_relaxation=0.9)
            m.set_state(case["pv"], case["meaning"])
            m.set_source(case["source"])
            sim = bench.make_simulation(pkg, m)
            log = []
            for _ in range(steps):
                rep = sim.next_newton_iteration()
                log.append((sim.timesteps_done, sim.iteration, int(rep.total_linear_iterations)))
            out[r] = (log, sim.timesteps_failed)
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
    assert all(o == out[0] for o in out)            # the global scalars steer every rank the same way
    log, failed = out[0]
    assert failed == 0
    lin = sum(entry[2] for entry in log)
    assert 10 * steps <= lin <= 32 * steps, log     # 17 ... 26 linear iterations per Newton iteration expected
