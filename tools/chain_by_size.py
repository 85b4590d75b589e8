import importlib, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
pkg = importlib.import_module("opm-autodiff_amd")
for n in [int(v) for v in os.environ.get("SIZES", "50 64 80").split()]:
    case = pkg.decks.cartesian_case(n, n, n, state="mixed", heterogeneous=False)
    src = pkg.decks.five_spot_source(case, rate_sm3_per_day=pkg.decks.BENCH_RATE_SM3_PER_DAY * (n / 100.0) ** 2)
    out = []
    for cl in [int(v) for v in os.environ.get("CHAINS", "4 5 8 10 16").split()]:
        if n % cl and cl not in (8, 16): pass
        m = pkg.capi.HipModel(case, reorder="line_coloring", tolerance=1e-2, maxit=200, ilu_relaxation=0.9, chain_length=cl)
        m.set_state(case["pv"], case["meaning"]); m.set_source(src)
        sim = bench.make_simulation(pkg, m)
        for _ in range(5): sim.next_newton_iteration()
        m.synchronize(); t0 = time.perf_counter(); lin = 0
        for _ in range(30): lin += sim.next_newton_iteration().total_linear_iterations
        m.synchronize(); el = time.perf_counter() - t0
        out.append("chain %d: %.0f its/s (%.1f)" % (cl, 30 / el, lin / 30))
    print("%3d^3: %s" % (n, "   ".join(out)), flush=True)
