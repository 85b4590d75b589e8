#!/usr/bin/env python3
"""Where a Newton iteration with a host-side well model spends its time at the bench's size (two wells of 100 completions on the 100^3 case, as
tools/wells_at_scale.py): every call of the model and of wells.StandardWells wrapped with a wall-clock timer.    python tools/wells_breakdown.py"""
import importlib, os, sys, time, collections
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
sys.argv = ["x", "--steps", "0", "--warmup", "0"]
import numpy as np
import bench
pkg = importlib.import_module("opm-autodiff_amd")
n = 100
case = pkg.decks.cartesian_case(n, n, n, state="mixed", heterogeneous=False)
rate = pkg.decks.BENCH_RATE_SM3_PER_DAY * n * n / 1e4
W = pkg.wells
col = lambda i, j: [i + n * (j + n * k) for k in range(n)]
tw = lambda cells: [W.peaceman_factor(case["perm"][c], case["dx"], case["dy"], case["dz"], 0.1524) for c in cells]
ci, cp = col(0, 0), col(n - 1, n - 1)
q = rate / 86400.0
wells = W.StandardWells([W.Well("INJ", ci, tw(ci), case["depth"][ci[0]], False, ("rate", W.WATER, q), 1000e5, inj_phase="water"),
                         W.Well("PROD", cp, tw(cp), case["depth"][cp[0]], True, ("rate", W.OIL, q), 10e5)], case["depth"])
m = pkg.capi.HipModel(case, tolerance=1e-2, maxit=200, ilu_relaxation=0.9)
m.set_state(case["pv"], case["meaning"])
T = collections.defaultdict(float); Nn = collections.defaultdict(int)
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T[name] += time.perf_counter() - t0; Nn[name] += 1; return r
    setattr(obj, name, g)
for nm_ in ("iq_cells", "set_source_cells", "assemble", "wells_apply_residual", "solve_jacobian_system", "wells_recover_solution", "update", "convergence", "relative_change"):
    if hasattr(m, nm_): wrap(m, nm_)
for nm_ in ("assemble", "update_well_controls", "update", "converged", "records", "solve_well_equations", "calculate_explicit_quantities"):
    f = getattr(wells, nm_)
    def mk(f, key):
        def g(*a, **k):
            t0 = time.perf_counter(); r = f(*a, **k); T[key] += time.perf_counter() - t0; Nn[key] += 1; return r
        return g
    setattr(wells, nm_, mk(f, "wells." + nm_))
nmod = pkg.newton.BlackoilModelHip(m, well_model=wells)
sim = pkg.newton.AdaptiveTimeStepping(nmod, pkg.newton.TimeSteppingParameters(initial_dt=bench.DAY, max_dt=10 * bench.DAY))
for _ in range(5): sim.next_newton_iteration()
m.synchronize(); T.clear(); Nn.clear()
t0 = time.perf_counter()
for _ in range(20): sim.next_newton_iteration()
m.synchronize()
el = time.perf_counter() - t0
print("total %.2f ms per Newton iteration" % (1e3 * el / 20))
for k in sorted(T, key=lambda k: -T[k]): print("  %-34s %3d calls  %7.3f ms per Newton iteration" % (k, Nn[k], 1e3 * T[k] / 20))
print("  unaccounted %.3f ms (the wrapped calls nest: wells.records contains iq_cells)" % (1e3 * (el - sum(v for k, v in T.items() if k != "iq_cells")) / 20))
