#!/usr/bin/env python3
"""Prints CNV / MB / linear iterations of every Newton iteration of the bench workload (one GPU)."""
import argparse, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import numpy as np  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--reorder", default="line_coloring")
ap.add_argument("--rate", type=float, default=200.0)
ap.add_argument("--dt", type=float, default=10.0)
ap.add_argument("--noise", type=int, default=1)
ap.add_argument("--state", default="mixed")
ap.add_argument("--quiet", action="store_true")
a = ap.parse_args()
pkg = importlib.import_module("opm-autodiff_amd")
n = a.n
case = pkg.decks.cartesian_case(n, n, n, state=a.state, heterogeneous=False, perturb=bool(a.noise))
src = pkg.decks.five_spot_source(case, rate_sm3_per_day=a.rate * (n / 100.0) ** 2)
m = pkg.capi.HipModel(case, reorder=a.reorder, tolerance=1e-2, maxit=200, ilu_relaxation=0.9)
m.set_state(case["pv"], case["meaning"])
m.set_source(src)
sim = bench.make_simulation(pkg, m, report_step=a.dt * bench.DAY)
drv = sim.model
np.set_printoptions(precision=3, linewidth=200)
for k in range(a.steps):
    rep = sim.next_newton_iteration()
    if not a.quiet:
      print("t %.2f d  dt %.3f d  it %2d  lin %3d  relax %.1f | CNV %s | done %d chopped %d" % (sim.time / bench.DAY, sim.dt / bench.DAY, sim.iteration, rep.total_linear_iterations,
          drv.current_relaxation, np.array(drv.residual_norms_history[-1]), sim.timesteps_done, sim.timesteps_failed), flush=True)
print("time steps (days, newton its, accepted):", [(round(h[0] / bench.DAY, 3), h[1], h[2]) for h in sim.history])
