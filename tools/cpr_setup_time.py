#!/usr/bin/env python3
"""How long the one-time host-side set-up of the CPR hierarchy takes (development tool): first solve against second."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("opm-autodiff_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
case = pkg.decks.cartesian_case(n, n, n, state="mixed", heterogeneous=False)
m = pkg.capi.HipModel(case, reorder="line_coloring", preconditioner="cpr_quasiimpes")
m.set_state(case["pv"], case["meaning"])
m.assemble(86400.0, 0, fetch=False)
for k in range(3):
    t0 = time.perf_counter(); r = m.solve_jacobian_system(); t1 = time.perf_counter()
    print("solve %d: %.3f s (factor+setup %.3f, solve %.3f) it %.1f" % (k, t1 - t0, r.t_factor, r.t_solve, r.it), flush=True)
