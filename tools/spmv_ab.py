#!/usr/bin/env python3
"""SpMV forms back to back on the 100^3 bench matrix (development tool): plain, with one / two folded scalar products."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("opm-autodiff_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
case = pkg.decks.cartesian_case(n, n, n, state="mixed", heterogeneous=False)
m = pkg.capi.HipModel(case, reorder="line_coloring")
m.set_state(case["pv"], case["meaning"])
m.assemble(86400.0, 0, fetch=False)
m.solve_jacobian_system()
for rnd in range(3):
    print(" ".join("%s %.4f" % (k, m.time_kernel(k, 50)) for k in ("spmv", "spmv_dot1", "spmv_dot2", "stream_read", "ilu_apply")), flush=True)
