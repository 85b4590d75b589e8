#!/usr/bin/env python3
"""Assembly-side kernels alone on the bench workload (for rocprofv3 / PMC passes): k_iq_update via set_state, then
`reps` x (k_assemble + convergence), timed with the library's HIP-event scopes."""
import argparse, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--reorder", default="line_coloring")
a = ap.parse_args()
pkg = importlib.import_module("opm-autodiff_amd")
n = a.n
case = pkg.decks.cartesian_case(n, n, n, state="mixed", heterogeneous=False)
src = pkg.decks.five_spot_source(case, rate_sm3_per_day=pkg.decks.BENCH_RATE_SM3_PER_DAY * (n / 100.0) ** 2)
m = pkg.capi.HipModel(case, reorder=a.reorder)
m.set_state(case["pv"], case["meaning"])
m.set_source(src)
m.assemble(86400.0, 0, fetch=False)
m.profile_enable(1)
for i in range(a.reps):
    m.assemble(86400.0, 1, fetch=False)
    m.convergence(86400.0)
for k, (cnt, ms) in m.profile().items():
    if cnt:
        print("%-12s %4d launches  %.4f ms avg" % (k, cnt, ms / cnt))
