#!/usr/bin/env python3
"""Parameters of the product's pressure AMG (Jacobi damping, prolongation damping, strength threshold, V- or W-cycle) against
CPR-BiCGStab iterations on steady-state systems of the 100^3 bench case: the case is advanced on the GPU to Newton iteration
--at, then --n systems are fetched and solved on the CPU by the oracle with every variant (hierarchy structure from the
run's FIRST Jacobian, as the device keeps it with --cpr-reuse-setup=3).  Prints one JSON line.
    python tools/cpr_param_sweep.py --n 3 --at 200"""
import argparse, ctypes, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, oracle_bind  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=100)
ap.add_argument("--at", type=int, default=200)
ap.add_argument("--n", type=int, default=3)
ap.add_argument("--omega", type=float, nargs="*", default=[0.6, 2.0 / 3.0, 0.75, 0.85])
ap.add_argument("--damp", type=float, nargs="*", default=[1.0, 1.3, 1.6, 1.9])
ap.add_argument("--beta", type=float, nargs="*", default=[0.25])
ap.add_argument("--wfrom", type=int, nargs="*", default=[-1, 2, 3])
a = ap.parse_args()
pkg = importlib.import_module("opm-autodiff_amd")
n = a.size
case = pkg.decks.cartesian_case(n, n, n, state="mixed", heterogeneous=False)
src = pkg.decks.five_spot_source(case, rate_sm3_per_day=pkg.decks.BENCH_RATE_SM3_PER_DAY * (n / 100.0) ** 2)
m = pkg.capi.HipModel(case, reorder="line_coloring", tolerance=1e-2, maxit=200, ilu_relaxation=0.9, preconditioner="cpr_quasiimpes")
m.set_state(case["pv"], case["meaning"])
m.set_source(src)
sim = bench.make_simulation(pkg, m)
first_jac, _ = m.assemble(sim.dt, 0, fetch=True)
for k in range(a.at):
    sim.next_newton_iteration()
orc = oracle_bind.Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
orc.lib.orc_cpr_set_wcycle_from.argtypes = [ctypes.c_void_p, ctypes.c_int]
Nb, rp, ci = case["Nb"], case["rowptr"], case["col"]
variants = {}
for om in a.omega:
    for dp in a.damp:
        for be in a.beta:
            for wf in a.wfrom:
                if wf >= 0 and not (abs(om - 2.0 / 3.0) < 1e-9 or dp == 1.6):
                    continue      # W-cycles only along the two lines through the product's point
                c = oracle_bind.OracleCpr(orc, omega=om, damp=dp, beta=be)
                orc.lib.orc_cpr_set_wcycle_from(c.h, wf)
                c.update(Nb, rp, ci, first_jac)
                variants["omega%.3f_damp%.2f_beta%.2f_w%d" % (om, dp, be, wf)] = c
out = {"size": n, "from_newton_iteration": a.at, "systems": [], "variants": list(variants)}
for k in range(a.n):
    dt, it = sim.dt, sim.iteration
    if it == 0:
        sim.next_newton_iteration()
        dt, it = sim.dt, sim.iteration
    jac, res = m.assemble(dt, it, fetch=True)
    rec = {"dt_days": dt / 86400.0, "newton_iteration_in_step": it}
    for name, c in variants.items():
        t0 = time.perf_counter()
        x, r = c.solve(Nb, rp, ci, jac, res, tol=1e-2, maxit=200)
        rec[name] = float(r.it) if r.converged else None
    rep = sim.next_newton_iteration()
    rec["device_product_cpr_iterations"] = int(rep.total_linear_iterations)
    out["systems"].append(rec)
    print(rec, file=sys.stderr, flush=True)
out["mean_iterations"] = {name: (float(np.mean([s[name] for s in out["systems"]])) if all(s[name] is not None for s in out["systems"]) else None) for name in variants}
print(json.dumps(out))
