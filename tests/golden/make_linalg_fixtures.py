#!/usr/bin/env python3
"""Generates tests/golden/linalg/*: the reference's own known-answer data for the linear-solve path.

Run in the build container (needs /root/reference).  Output is DATA only:
  - matr33.txt, rhs3.txt, matr33rep.txt, rhs3rep.txt : MatrixMarket data files held by the reference's
    tests (tests/matr33.txt ... ), copied byte for byte;
  - expected.json : the expected solution vectors hard-coded in the reference's tests, with the file:line
    each one comes from and the solver settings that test uses.
"""
import json, os, shutil

REF = "/root/reference/tests"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "linalg")
os.makedirs(OUT, exist_ok=True)
for f in ("matr33.txt", "rhs3.txt", "matr33rep.txt", "rhs3rep.txt"):
    shutil.copyfile(os.path.join(REF, f), os.path.join(OUT, f))

expected = {
    "cusparse_ilu0_w1_tol0.5_maxit20": {
        "source": "tests/test_cusparseSolver.cpp:103-105 (options tests/options_flexiblesolver.json:2-3)",
        "matrix": "matr33.txt", "rhs": "rhs3.txt", "tol": 0.5, "maxit": 20,
        "ilu_relaxation": 1.0, "relax_mode": "post_scale", "reorder": "none",
        "check_close_percent": 1e-3,
        "x": [-0.0131626, -3.5826e-6, 1.138362e-9,
              -1.25425e-3, -1.4167e-4, -0.0029366,
              -4.54355e-4, 1.28682e-5, 4.7644e-6],
    },
    "opencl_ilu0_w0.9_tol0.5_maxit20": {
        "source": "tests/test_openclSolver.cpp:102-104 (reorder 'none' at :74; relaxation 0.9 hard-coded at "
                  "opm/simulators/linalg/bda/openclKernels.cpp:328)",
        "matrix": "matr33.txt", "rhs": "rhs3.txt", "tol": 0.5, "maxit": 20,
        "ilu_relaxation": 0.9, "relax_mode": "in_sweep", "reorder": "none",
        "check_close_percent": 1e-3,
        "x": [-1.30307e-2, -3.58263e-6, 1.13836e-9,
              -1.25425e-3, -1.4167e-4, -3.2213e-3,
              -4.5436e-4, 1.28682e-5, 4.7644e-6],
    },
    "exact_noprec_tol1e-12_maxit200": {
        "source": "tests/test_flexiblesolver.cpp:114-116 (options tests/options_flexiblesolver_simple.json)",
        "matrix": "matr33.txt", "rhs": "rhs3.txt", "tol": 1e-12, "maxit": 200,
        "preconditioner": "nothing",
        "check_close_percent": 1e-3,
        "x": [-1.62493, -1.76435e-06, 1.86991e-10,
              -458.542, 2.28308e-06, -2.45341e-07,
              -1.48005, -5.02264e-07, -1.049e-05],
    },
}
# matr33rep/rhs3rep: tests/test_preconditionerfactory.cpp:318-343 applies the operator A twice
# ("RepeatingOperator", :300-316) with preconditioner "nothing", tol 1e-12.
expected["rep_operator_squared_noprec"] = {
    "source": "tests/test_preconditionerfactory.cpp:318-343",
    "matrix": "matr33rep.txt", "rhs": "rhs3rep.txt", "tol": 1e-12, "maxit": 200,
    "operator": "A*A", "preconditioner": "nothing", "check_close_percent": 1e-3,
    "x": [0.285714285714286] * 3 + [-0.214285714285714] * 6,
}
with open(os.path.join(OUT, "expected.json"), "w") as f:
    json.dump(expected, f, indent=1)
print("wrote", OUT)
