#!/usr/bin/env python3
"""f3 yardstick: CPR-BiCGStab iterations per Newton iteration with the product's pressure AMG (pairwise matching + Jacobi,
csrc/cpr.hip = oracle CprAmg) against a restatement of the reference's (Dune::Amg-like aggregation + ILU0 smoothing + direct
coarse solve, oracle DuneLikeAmg), on the SAME Jacobians: the 100^3 bench case is advanced on the GPU with the product's CPR
to Newton iteration --at, then --n Newton iterations' systems are fetched and solved on the CPU by the oracle with either
hierarchy (quasi-IMPES weights, tolerance 1e-2, as in the bench).  Prints one JSON line.
    python tools/cpr_amg_compare.py --n 4 --at 200"""
import argparse, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, oracle_bind  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=100)
ap.add_argument("--at", type=int, default=200)
ap.add_argument("--n", type=int, default=4)
ap.add_argument("--first", action="store_true", help="also: hierarchies whose structure comes from the INITIAL Jacobian (as the device's does), with variants of the coarsest level")
ap.add_argument("--structure-at", type=int, nargs="*", default=[], help="also: the product's AMG with its structure built from the Jacobian of these Newton iterations")
ap.add_argument("--levels", type=int, nargs="*", default=[], help="also: the product's AMG cut off at this many levels")
a = ap.parse_args()
pkg = importlib.import_module("opm-autodiff_amd")
n = a.size
case = pkg.decks.cartesian_case(n, n, n, state="mixed", heterogeneous=False)
src = pkg.decks.five_spot_source(case, rate_sm3_per_day=pkg.decks.BENCH_RATE_SM3_PER_DAY * (n / 100.0) ** 2)
m = pkg.capi.HipModel(case, reorder="line_coloring", tolerance=1e-2, maxit=200, ilu_relaxation=0.9, preconditioner="cpr_quasiimpes")
m.set_state(case["pv"], case["meaning"])
m.set_source(src)
sim = bench.make_simulation(pkg, m)
first_jac = None
if a.first:
    first_jac, _ = m.assemble(sim.dt, 0, fetch=True)
early = {}
for k in range(a.at):
    if k in a.structure_at:     # the system the device is about to solve at Newton iteration k
        if sim.iteration == 0:
            sim.next_newton_iteration()
        early[k], _ = m.assemble(sim.dt, sim.iteration, fetch=True)
    sim.next_newton_iteration()
orc = oracle_bind.Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
Nb, rp, ci = case["Nb"], case["rowptr"], case["col"]
cprs = {}
import ctypes
orc.lib.orc_cpr_set_max_levels.argtypes = [ctypes.c_void_p, ctypes.c_int]
for name, ref, lev in [("product_amg", False, 0), ("reference_like_amg", True, 0), ("reference_aggregation_jacobi", 2, 0)] + [("product_amg_%d_levels" % k, False, k) for k in a.levels]:
    c = oracle_bind.OracleCpr(orc)
    c.use_reference_amg(ref)
    if lev:
        orc.lib.orc_cpr_set_max_levels(c.h, lev)
    cprs[name] = c
orc.lib.orc_cpr_set_coarse_sweeps.argtypes = [ctypes.c_void_p, ctypes.c_int]
orc.lib.orc_cpr_set_sweeps.argtypes = [ctypes.c_void_p, ctypes.c_int]
for name, kw in (("first_product_amg", {}), ("first_coarse10", dict(cs=10)), ("first_coarse20", dict(cs=20)), ("first_join_at_stall", dict(js=True))) if a.first else ():
    # the device keeps the hierarchy's STRUCTURE of its first matrix: build these from the initial Jacobian too
    c = oracle_bind.OracleCpr(orc)
    if kw.get("cs"):
        orc.lib.orc_cpr_set_coarse_sweeps(c.h, kw["cs"])
    if kw.get("js"):
        orc.lib.orc_cpr_set_sweeps(c.h, -1)
    c.update(Nb, rp, ci, first_jac)
    cprs[name] = c
for k, jk in early.items():
    c = oracle_bind.OracleCpr(orc)
    c.update(Nb, rp, ci, jk)
    cprs["structure_from_newton_%d" % k] = c
out = {"size": n, "from_newton_iteration": a.at, "systems": []}
for k in range(a.n):
    # the system of the next Newton iteration, exactly as the device solves it: assemble with a host copy, then let the device go on
    dt, it = sim.dt, sim.iteration
    if it == 0:
        sim.next_newton_iteration()      # starts a time step (advance_time_level, iteration 0): take the following iteration's system
        dt, it = sim.dt, sim.iteration
    jac, res = m.assemble(dt, it, fetch=True)
    rec = {"dt_days": dt / 86400.0, "newton_iteration_in_step": it}
    for name, c in cprs.items():
        t0 = time.perf_counter()
        x, r = c.solve(Nb, rp, ci, jac, res, tol=1e-2, maxit=200)
        rec[name] = {"iterations": float(r.it), "converged": bool(r.converged), "seconds": round(time.perf_counter() - t0, 2)}
    rep = sim.next_newton_iteration()
    rec["device_product_cpr_iterations"] = int(rep.total_linear_iterations)
    out["systems"].append(rec)
    print(rec, file=sys.stderr, flush=True)
out["levels_product_amg"] = [int(v) for v in cprs["product_amg"].levels()[0]]
out["levels_reference_like_amg"] = cprs["reference_like_amg"].reference_amg_levels()[0]
for name in cprs:
    out["mean_iterations_" + name] = float(np.mean([s[name]["iterations"] for s in out["systems"]]))
print(json.dumps(out))
