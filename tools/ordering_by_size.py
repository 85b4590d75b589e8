import importlib, os, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
pkg = importlib.import_module("opm-autodiff_amd")
for n in (16, 24, 32, 40, 50, 64, 80):
    case = pkg.decks.cartesian_case(n, n, n, state="mixed", heterogeneous=False)
    src = pkg.decks.five_spot_source(case, rate_sm3_per_day=pkg.decks.BENCH_RATE_SM3_PER_DAY * (n / 100.0) ** 2)
    out = []
    for reorder in ("line_coloring", "graph_coloring_greedy", "graph_coloring"):
        m = pkg.capi.HipModel(case, reorder=reorder, tolerance=1e-2, maxit=200, ilu_relaxation=0.9, chain_length=10)
        m.set_state(case["pv"], case["meaning"]); m.set_source(src)
        sim = bench.make_simulation(pkg, m)
        for _ in range(5): sim.next_newton_iteration()
        m.synchronize(); t0 = time.perf_counter(); lin = 0
        for _ in range(30): lin += sim.next_newton_iteration().total_linear_iterations
        m.synchronize(); el = time.perf_counter() - t0
        out.append("%s %.0f its/s (%.1f)" % (reorder, 30 / el, lin / 30))
    print("%3d^3 = %7d cells: %s" % (n, n ** 3, "   ".join(out)), flush=True)
