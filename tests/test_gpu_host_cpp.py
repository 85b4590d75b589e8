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


@pytest.mark.parametrize("mode", ["host", "device"])
def test_HipLinearizer_drives_newton_iterations(pkg, tmp_path, mode):
    """Opm::HipLinearizer<TypeTag> (host/HipLinearizer.hpp), compiled against the property-system include path and driven through
    the linearizer's public face - linearizeDomain(), jacobian(), residual(), solution(0) hand-off, invalidateAndUpdateIntensive
    Quantities (flow/BlackoilModelEbos.hpp:339-340, 424, 526-527, 552-562): every iteration's system and state are the bits the
    plain C-ABI sequence gives (the driver compares them itself, with host copies of J and r or with both left in HBM), and the
    final state equals the one this process reaches through the Python binding"""
    case = pkg.decks.cartesian_case(10, 9, 7, state="mixed", heterogeneous=True)
    src = pkg.decks.five_spot_source(case, rate_sm3_per_day=40.0)
    cf, so = str(tmp_path / "case.bin"), str(tmp_path / "state.bin")
    pkg.decks.write_case_binary(case, cf, source=src)
    dt, its = 2 * 86400.0, 3
    out = subprocess.run([_exe("test_HipLinearizer"), cf, mode, repr(dt), str(its), so], capture_output=True, text=True)
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
