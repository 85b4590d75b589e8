"""The C++ host mirror (opm-autodiff_amd/host/) on the GPU: bda::hipSolverBackend<3> driven exactly like the reference's
tests/test_cusparseSolver.cpp drives its backend, and Opm::BlackoilModelHip::step against the Python-driven loop."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "opm-autodiff_amd", "host")


def _exe(name):
    p = os.path.join(HOST, name)
    if not os.path.exists(p):
        subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    return p


@pytest.mark.parametrize("reorder", ["level_scheduling", "graph_coloring", "line_coloring"])
def test_hipSolverBackend_matr33(golden, reorder):
    out = subprocess.run([_exe("test_hipSolver"), os.path.join(golden, "linalg", "matr33.txt"), os.path.join(golden, "linalg", "rhs3.txt"),
                          "0.5", "20", reorder], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    assert lines[0].startswith("converged 1")
    x = np.array([float(v) for v in lines[1:]])
    with open(os.path.join(golden, "linalg", "expected.json")) as f:
        e = json.load(f)["exact_noprec_tol1e-12_maxit200"]
    # block ILU0 is an exact LU of this block-tridiagonal matrix in the natural (= level) order: the first half
    # iteration lands on the solution tests/test_flexiblesolver.cpp:114-116 pins
    if reorder == "level_scheduling":
        np.testing.assert_allclose(x, e["x"], rtol=2e-5)
    mm = __import__("importlib").import_module("opm-autodiff_amd").mmio
    Nb, rp, ci, v, _ = mm.read_block_matrix(os.path.join(golden, "linalg", "matr33.txt"))
    b = mm.read_block_vector(os.path.join(golden, "linalg", "rhs3.txt"))
    A = np.zeros((9, 9))
    for i in range(Nb):
        for k in range(rp[i], rp[i + 1]):
            A[3 * i:3 * i + 3, 3 * ci[k]:3 * ci[k] + 3] = v[9 * k:9 * k + 9].reshape(3, 3)
    assert np.linalg.norm(A @ x - b) < 0.5 * np.linalg.norm(b)


def test_hipSolverBackend_cpr_through_the_plugin_class(golden):
    """--linear-solver-configuration=cpr_quasiimpes through bda::hipSolverBackend<3> (both builds), the reference's CPR vector
    (tests/test_flexiblesolver.cpp:93-116 with options_flexiblesolver.json: tol 0.5 on matr33), solved twice - the second
    time behind recreateCprHierarchy(), the plug-in's --cpr-reuse-setup=1 path"""
    args = [os.path.join(golden, "linalg", "matr33.txt"), os.path.join(golden, "linalg", "rhs3.txt"), "0.5", "20", "level_scheduling", "-", "cpr_quasiimpes"]
    with open(os.path.join(golden, "linalg", "expected.json")) as f:
        e = json.load(f)["exact_noprec_tol1e-12_maxit200"]
    for exe in ("test_hipSolver", "test_hipSolver_opmhdr"):
        out = subprocess.run([_exe(exe)] + args, capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        lines = out.stdout.strip().splitlines()
        assert lines[0].startswith("converged 1")
        np.testing.assert_allclose(np.array([float(v) for v in lines[1:]]), e["x"], rtol=2e-5)
    bad = subprocess.run([_exe("test_hipSolver")] + args[:6] + ["amg"], capture_output=True, text=True)
    assert bad.returncode != 0 and "not a valid setting" in (bad.stderr + bad.stdout)


def test_hipSolverBackend_built_the_way_opm_simulators_builds_it(pkg, golden):
    """The OPMHIP_USE_OPM_HEADERS branch of hipSolverBackend.hpp - reference include paths, WellContributions reached through
    the getHostArrays accessor of INTEGRATION.md - gives the same bits as the stand-alone build, with a standard well, and
    both equal the C-ABI called from Python with the same well (host/Makefile: test_hipSolver_opmhdr)."""
    args = [os.path.join(golden, "linalg", "matr33.txt"), os.path.join(golden, "linalg", "rhs3.txt"), "1e-8", "50", "level_scheduling", "wells"]
    outs = []
    for exe in ("test_hipSolver", "test_hipSolver_opmhdr"):
        out = subprocess.run([_exe(exe)] + args, capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        outs.append(out.stdout)
    assert outs[0] == outs[1]
    x = np.array([float(v) for v in outs[0].strip().splitlines()[1:]])
    Nb, rp, ci, v, _ = pkg.mmio.read_block_matrix(args[0])
    b = pkg.mmio.read_block_vector(args[1])
    C = np.array([0.01 * (1 + (i * 7) % 5) for i in range(24)])
    B = np.array([0.02 * (1 + (i * 3) % 7) for i in range(24)])
    D = np.array([0.5 if i % 5 == 0 else 0.01 * (i % 3) for i in range(16)])
    wells = dict(numWells=1, val_pointers=[0, 2], Ccols=[1, Nb - 2], Bcols=[1, Nb - 2], Cnnzs=C, Dnnzs=D, Bnnzs=B)
    s = pkg.capi.HipSolver(tolerance=1e-8, maxit=50, reorder="level_scheduling", ilu_relaxation=1.0)
    res = s.solve_system(Nb, rp, ci, v.copy(), b, wells=wells)
    assert res.converged and np.array_equal(x, s.get_result())
    # and the well does something: the solution differs from the one without it
    out0 = subprocess.run([_exe("test_hipSolver")] + args[:-1], capture_output=True, text=True)
    x0 = np.array([float(t) for t in out0.stdout.strip().splitlines()[1:]])
    assert np.abs(x - x0).max() > 1e-6 * np.abs(x0).max()


def _ms_well(Nb):
    """the multisegment well of host/test_hipSolver.cpp (mswells / msonly) as dense matrices: B, C (8 x 3 Nb), D (8 x 8)"""
    Bv = np.array([0.03 * (1 + (i * 5) % 7) - 0.05 for i in range(36)]).reshape(3, 4, 3)
    Cv = np.array([0.02 * (1 + (i * 3) % 5) for i in range(36)]).reshape(3, 4, 3)
    cols, seg = [0, 2, Nb - 1], [0, 1, 1]
    B, Cm = np.zeros((8, 3 * Nb)), np.zeros((8, 3 * Nb))
    for blk in range(3):
        B[4 * seg[blk]:4 * seg[blk] + 4, 3 * cols[blk]:3 * cols[blk] + 3] += Bv[blk]
        Cm[4 * seg[blk]:4 * seg[blk] + 4, 3 * cols[blk]:3 * cols[blk] + 3] += Cv[blk]
    D = np.array([[2.0 + 0.1 * r if r == c else 0.05 * ((r * 3 + c * 5) % 4) - 0.04 for c in range(8)] for r in range(8)])
    return B, Cm, D


@pytest.mark.parametrize("mode", ["mswells", "msonly"])
def test_hipSolverBackend_with_a_multisegment_well(pkg, golden, mode):
    """A WellContributions that holds a multisegment well - getNumWells() counts it, the standard wells' C arrays do not hold it
    (bda/WellContributions.hpp:164-166).  The plug-in hands the library the standard-well count and a callback; the library applies the
    multisegment operator on the host after every product, as the reference's back-ends do (bda/WellContributions.cu:160-187).  Both builds
    of the plug-in give the same bits, the solution solves (A - sum C^T D^-1 B) x = b with the operators formed densely here, and the
    same system through the C ABI from Python with a numpy callback gives the same x."""
    args = [os.path.join(golden, "linalg", "matr33.txt"), os.path.join(golden, "linalg", "rhs3.txt"), "1e-10", "50", "level_scheduling", mode]
    outs = []
    for exe in ("test_hipSolver", "test_hipSolver_opmhdr"):
        out = subprocess.run([_exe(exe)] + args, capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        outs.append(out.stdout)
    assert outs[0] == outs[1] and outs[0].startswith("converged 1")
    x = np.array([float(v) for v in outs[0].strip().splitlines()[1:]])
    Nb, rp, ci, v, _ = pkg.mmio.read_block_matrix(args[0])
    b = pkg.mmio.read_block_vector(args[1])
    A = np.zeros((3 * Nb, 3 * Nb))
    for i in range(Nb):
        for k in range(rp[i], rp[i + 1]):
            A[3 * i:3 * i + 3, 3 * ci[k]:3 * ci[k] + 3] = v[9 * k:9 * k + 9].reshape(3, 3)
    B, Cm, D = _ms_well(Nb)
    Aeff = A - Cm.T @ np.linalg.solve(D, B)
    wells = dict(numWells=0)
    if mode == "mswells":   # plus the standard well of the `wells` mode (D^-1 arrives inverted: applied as it is)
        Cs = np.array([0.01 * (1 + (i * 7) % 5) for i in range(24)])
        Bs = np.array([0.02 * (1 + (i * 3) % 7) for i in range(24)])
        Ds = np.array([0.5 if i % 5 == 0 else 0.01 * (i % 3) for i in range(16)])
        Bd, Cd = np.zeros((4, 3 * Nb)), np.zeros((4, 3 * Nb))
        for blk, col in enumerate([1, Nb - 2]):
            Bd[:, 3 * col:3 * col + 3] += Bs[12 * blk:12 * blk + 12].reshape(4, 3)     # (+=: matr33 has three block rows, both perforations sit in cell 1)
            Cd[:, 3 * col:3 * col + 3] += Cs[12 * blk:12 * blk + 12].reshape(4, 3)
        Aeff = Aeff - Cd.T @ (Ds.reshape(4, 4) @ Bd)
        wells = dict(numWells=1, val_pointers=[0, 2], Ccols=[1, Nb - 2], Bcols=[1, Nb - 2], Cnnzs=Cs, Dnnzs=Ds, Bnnzs=Bs)
    assert np.linalg.norm(Aeff @ x - b) <= 1e-10 * np.linalg.norm(b) * (1 + 1e-6)
    # the well does something
    out0 = subprocess.run([_exe("test_hipSolver")] + args[:-1], capture_output=True, text=True)
    x0 = np.array([float(t) for t in out0.stdout.strip().splitlines()[1:]])
    assert np.abs(x - x0).max() > 1e-6 * np.abs(x0).max()
    # the same through the C ABI from Python: the callback sees natural-order host vectors although the ILU runs in its own order
    calls = []

    def ms_apply(hx, hy):
        calls.append(1)
        hy -= Cm.T @ np.linalg.solve(D, B @ hx)
    wells.update(numMsWells=1, ms_apply=ms_apply, N=3 * Nb)
    for reorder, prec in (("level_scheduling", "ilu0"), ("graph_coloring", "ilu0"), ("level_scheduling", "cpr_quasiimpes")):   # ... and behind a CPR-preconditioned solve
        s = pkg.capi.HipSolver(tolerance=1e-10, maxit=50, reorder=reorder, ilu_relaxation=1.0, preconditioner=prec)
        res = s.solve_system(Nb, rp, ci, v.copy(), b, wells=wells)
        assert res.converged and len(calls) >= 2
        xs = s.get_result()
        assert np.linalg.norm(Aeff @ xs - b) <= 1e-10 * np.linalg.norm(b) * (1 + 1e-6)
        if reorder == "level_scheduling" and prec == "ilu0":
            # two solves to 1e-10 with differently rounded well operators (numpy's dense product here, the stand-in's LU there)
            np.testing.assert_allclose(xs, x, rtol=1e-5, atol=1e-7 * np.abs(x).max())


def test_multisegment_wells_need_their_callback(pkg, golden):
    """num_ms_wells > 0 without ms_apply is refused (OPMHIP_INVALID_ARGUMENT), not silently dropped from the operator"""
    import ctypes as C
    Nb, rp, ci, v, _ = pkg.mmio.read_block_matrix(os.path.join(golden, "linalg", "matr33.txt"))
    b = pkg.mmio.read_block_vector(os.path.join(golden, "linalg", "rhs3.txt"))
    s = pkg.capi.HipSolver(tolerance=1e-2, maxit=20)
    s.set_pattern(Nb, rp, ci)
    w = pkg.capi.Wells(0)
    w.num_ms_wells = 1
    res = pkg.capi.Result()
    rc = pkg.capi.lib().opmhip_solve_system(s._h, 3 * Nb, 9 * len(ci), 3, v.ctypes.data_as(C.c_void_p), None, None, b.ctypes.data_as(C.c_void_p),
                                            C.byref(w), C.byref(res))
    assert rc == pkg.capi.INVALID_ARGUMENT and b"ms_apply" in pkg.capi.lib().opmhip_last_error(s._h)


def test_BlackoilModelHip_step_matches_python_loop(pkg, tmp_path):
    case = pkg.decks.cartesian_case(12, 12, 8, state="mixed", heterogeneous=False)
    src = pkg.decks.five_spot_source(case, rate_sm3_per_day=40.0)
    cf, so = str(tmp_path / "case.bin"), str(tmp_path / "state.bin")
    pkg.decks.write_case_binary(case, cf, source=src)
    dt, nsteps = 2 * 86400.0, 2
    out = subprocess.run([_exe("test_BlackoilModelHip"), cf, "line_coloring", repr(dt), str(nsteps), so], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr + out.stdout
    cpp = [tuple(int(t) for t in l.split()[3::2]) for l in out.stdout.splitlines() if l.startswith("step")]
    m = pkg.capi.HipModel(case, reorder="line_coloring")
    m.set_state(case["pv"], case["meaning"])
    m.set_source(src)
    drv = pkg.newton.BlackoilModelHip(m)
    py = []
    for s in range(nsteps):
        r = drv.step(dt)
        py.append((r.total_newton_iterations, r.total_linear_iterations))
    assert cpp == py
    raw = np.fromfile(so, dtype=np.uint8)
    pv = raw[:case["Nb"] * 24].view(np.float64)
    mean = raw[case["Nb"] * 24:]
    pm, mm = m.get_state()
    assert np.array_equal(mean, mm) and np.array_equal(pv, pm)  # same library, same call sequence: identical bits


def test_BlackoilModelHip_report_step_survives_a_chop(pkg, tmp_path):
    """advanceReportStep (C++ mirror of AdaptiveTimeSteppingEbos::step): a report step whose first sub-step is too long
    for the Newton method is rolled back on the device, chopped and completed; the same case without the sub-step
    control fails, as NonlinearSolverEbos::step does."""
    case = pkg.decks.cartesian_case(8, 8, 8, state="mixed", heterogeneous=False)
    src = pkg.decks.five_spot_source(case, rate_sm3_per_day=60.0)
    cf, so = str(tmp_path / "case.bin"), str(tmp_path / "state.bin")
    pkg.decks.write_case_binary(case, cf, source=src)
    args = [_exe("test_BlackoilModelHip"), cf, "level_scheduling", repr(30 * 86400.0), "1", so]
    plain = subprocess.run(args, capture_output=True, text=True)
    assert plain.returncode == 4 and "Failed to complete a time step" in plain.stderr
    out = subprocess.run(args, capture_output=True, text=True, env=dict(os.environ, OPMHIP_TEST_ADAPTIVE="1"))
    assert out.returncode == 0, out.stderr + out.stdout
    line = [l for l in out.stdout.splitlines() if l.startswith("report step 0")][0].split()
    assert int(line[4]) >= 1
    raw = np.fromfile(so, dtype=np.uint8)
    pv = raw[:case["Nb"] * 24].view(np.float64).reshape(-1, 3)
    assert np.all(np.isfinite(pv)) and np.all(pv[:, 1] > 1e7) and np.all(pv[:, 0] > 0.0) and np.all(pv[:, 0] < 1.0)


@pytest.mark.parametrize("mode,exe", [("host", "test_HipLinearizer"), ("device", "test_HipLinearizer"), ("device", "test_HipLinearizer_source_dofs")])
def test_HipLinearizer_drives_newton_iterations(pkg, tmp_path, mode, exe):
    """Opm::HipLinearizer<TypeTag> (host/HipLinearizer.hpp), compiled against the property-system include path and driven through
    the linearizer's public face - linearizeDomain(), jacobian(), residual(), solution(0) hand-off, invalidateAndUpdateIntensive
    Quantities (flow/BlackoilModelEbos.hpp:339-340, 424, 526-527, 552-562): every iteration's system and state are the bits the
    plain C-ABI sequence gives (the driver compares them itself, with host copies of J and r or with both left in HBM), and the
    final state equals the one this process reaches through the Python binding.  test_HipLinearizer_source_dofs: the same driver over a
    problem that names the dofs its sources sit in (sourceDofs(), Flow's perforated cells): their rates alone travel, through
    opmhip_set_source_cells"""
    case = pkg.decks.cartesian_case(10, 9, 7, state="mixed", heterogeneous=True)
    src = pkg.decks.five_spot_source(case, rate_sm3_per_day=40.0)
    cf, so = str(tmp_path / "case.bin"), str(tmp_path / "state.bin")
    pkg.decks.write_case_binary(case, cf, source=src)
    dt, its = 2 * 86400.0, 3
    out = subprocess.run([_exe(exe), cf, mode, repr(dt), str(its), so], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr + out.stdout
    lines = [l for l in out.stdout.splitlines() if l.startswith("iteration")]
    assert len(lines) == its and all("DIFFERS" not in l for l in lines) and out.stdout.strip().endswith("ok")
    m = pkg.capi.HipModel(case, reorder="line_coloring")
    m.set_state(case["pv"], case["meaning"])
    m.set_source(src)
    for it in range(its):
        m.assemble(dt, it, fetch=False)
        res = m.solve_jacobian_system()
        assert res.iterations == int(lines[it].split()[3])
        m.update(None, 1.0)
        pv, mg = m.get_state()
        pv = pv.reshape(-1, 3).copy()
        pv[::7, 0] += 1e-4                      # the driver's host-side change of solution(0)
        m.set_state(pv.reshape(-1), mg)
    raw = np.fromfile(so, dtype=np.uint8)
    pvc = raw[:case["Nb"] * 24].view(np.float64)
    pm, mm = m.get_state()
    assert np.array_equal(raw[case["Nb"] * 24:], mm) and np.array_equal(pvc, pm)


def test_registered_host_arrays_give_the_same_solve(pkg, orc):
    """opmhip_config.pin_host_arrays (what host/hipSolverBackend.hpp sets: Flow's value array, right-hand side and solution vector keep their
    addresses, bda/BdaBridge.cpp:199-232): the arrays are registered for DMA the first time they are seen and every later solve copies from
    / to them directly - same iterations, same solution bits as the plain copies, for new values in the same arrays and for a second set of
    arrays on the same context"""
    from helpers import laplace_block_system
    Nb, rp, ci, v = laplace_block_system(20, 16, 10, seed=6)
    b = np.random.default_rng(10).standard_normal(3 * Nb)
    ref = pkg.capi.HipSolver(tolerance=1e-6, reorder="line_coloring")
    r0 = ref.solve_system(Nb, rp, ci, v.copy(), b)
    x0 = ref.get_result()
    s = pkg.capi.HipSolver(tolerance=1e-6, reorder="line_coloring", pin_host_arrays=1)
    vals, rhs = v.copy(), b.copy()                 # these two keep their addresses over the solves, like Flow's
    for k in range(3):
        r = s.solve_system(Nb, rp if k == 0 else None, ci if k == 0 else None, vals, rhs)
        assert r.converged and r.it == r0.it
        assert np.array_equal(s.get_result(), x0)
    vals *= 1.01                                   # new values in the SAME (registered) array
    r1 = s.solve_system(Nb, None, None, vals, rhs)
    ref1 = ref.solve_system(Nb, None, None, v * 1.01, b)
    assert r1.it == ref1.it and np.array_equal(s.get_result(), ref.get_result())
    other = (v * 0.99).copy()                      # another array: registered in turn
    r2 = s.solve_system(Nb, None, None, other, rhs)
    ref2 = ref.solve_system(Nb, None, None, v * 0.99, b)
    assert r2.it == ref2.it and np.array_equal(s.get_result(), ref.get_result())
    with pytest.raises(pkg.capi.OpmHipError):
        pkg.capi.HipSolver(pin_host_arrays=2)
