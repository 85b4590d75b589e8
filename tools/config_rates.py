#!/usr/bin/env python3
"""Newton iterations/s of the device path AND of the CPU port beside it (the oracle through the same loop: bench.py's cpu_baseline) on
BASELINE.json's other configurations (the bench line is configs[1]):
configs[0] the SPE1CASE1 deck itself (EQUIL state, DRSDT 0, its two wells; and its 10 x 10 x 3 grid with rate sources), configs[2] an SPE9-shaped 24 x 25 x 15 grid (heterogeneous permeability; with rate sources, and with SPE9's 26 wells as wells.StandardWells), configs[4] the
Norne-shaped corner-point grid of the tests (46 x 112 x 22, 44 777 active cells, faults, pinch-outs: rows of 2 to 12 blocks) - each under
bench.py's own time-step control, with ILU0 and with CPR.  These sizes do not fill the chip: what they show is the launch-bound end of the
path (a Newton iteration is ~60 kernel launches with ILU0).    python tools/config_rates.py [--steps 40]"""
import argparse, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import helpers  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--threads", type=int, default=min(16, len(os.sched_getaffinity(0))), help="threads of the CPU-N column")
ap.add_argument("--cpr-amg-ilu-levels", type=int, default=-1, help="-1: the library picks (level 0 with ILU0 where the ordering has at most three colours)")
a = ap.parse_args()
pkg = importlib.import_module("opm-autodiff_amd")


def five_spot(case, rate):
    return pkg.decks.five_spot_source(case, rate_sm3_per_day=rate)


def scattered_sources(case, rate_sm3_per_day, n=12, seed=3):
    """injection of water in n cells, as much oil out of n others (no nx / ny to place a five-spot on)"""
    rng = np.random.default_rng(seed)
    cells = rng.choice(case["Nb"], 2 * n, replace=False)
    s = np.zeros((case["Nb"], 3))
    q = rate_sm3_per_day / 86400.0
    s[cells[:n], 1] = q        # (equation order of the library: oil, water, gas; see decks.five_spot_source)
    s[cells[n:], 0] = -q
    return np.ascontiguousarray(s.reshape(-1))


def cpu_rate(case, src, threads, wells=None, drsdt=None, budget_s=8.0, drsdt_all=None):
    """the oracle (the CPU port of the same algorithms: bench.py's cpu_baseline) through the same time-step control, `threads` OpenMP
    threads (> 1: block-Jacobi ILU0 over that many row ranges, what that many MPI ranks of Flow do on one host)"""
    import oracle_bind
    orc = oracle_bind.Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
    o = oracle_bind.OracleModel(orc, case)
    o.set_state(case["pv"], case["meaning"])
    if src is not None:
        o.set_source(src)
    if drsdt is not None:
        o.set_composition_change_limits(drsdt=drsdt, drsdt_all_cells=drsdt_all)
    hm = oracle_bind.OracleAsHipModel(o, tol=1e-2, maxit=200, w=0.9, mode="post_scale", reorder="none", threads=threads)
    nm = pkg.newton.BlackoilModelHip(hm, well_model=wells() if wells else None)
    sim = pkg.newton.AdaptiveTimeStepping(nm, pkg.newton.TimeSteppingParameters(initial_dt=bench.DAY, max_dt=10 * bench.DAY))
    n, t0 = 0, time.perf_counter()
    for _ in range(a.warmup):
        sim.next_newton_iteration()
    t0 = time.perf_counter()
    while n < a.steps and time.perf_counter() - t0 < budget_s:
        sim.next_newton_iteration()
        n += 1
    return n / (time.perf_counter() - t0)


def run(name, case, src, wells=None, drsdt=None, drsdt_all=None, **kw):
    out, rates = [], {}
    for prec in ("ilu0", "cpr"):
        m = pkg.capi.HipModel(case, tolerance=1e-2, maxit=200, ilu_relaxation=0.9, preconditioner=prec,
                              cpr_amg_ilu_levels=a.cpr_amg_ilu_levels, **kw)   # no reorder argument: the library's default (auto)
        m.set_state(case["pv"], case["meaning"])
        if src is not None:
            m.set_source(src)
        if drsdt is not None:
            m.set_composition_change_limits(drsdt=drsdt, drsdt_all_cells=drsdt_all)
        nm = pkg.newton.BlackoilModelHip(m, well_model=wells() if wells else None)
        sim = pkg.newton.AdaptiveTimeStepping(nm, pkg.newton.TimeSteppingParameters(initial_dt=bench.DAY, max_dt=10 * bench.DAY))
        for _ in range(a.warmup):
            sim.next_newton_iteration()
        m.synchronize()
        t0 = time.perf_counter()
        lin = 0
        for _ in range(a.steps):
            lin += sim.next_newton_iteration().total_linear_iterations
        m.synchronize()
        el = time.perf_counter() - t0
        rates[prec] = a.steps / el
        out.append("%s %6.0f its/s (%4.1f lin/Newton)" % (prec, a.steps / el, lin / a.steps))
    cpu1 = cpu_rate(case, src, 1, wells, drsdt, drsdt_all=drsdt_all)
    cpun = cpu_rate(case, src, a.threads, wells, drsdt, drsdt_all=drsdt_all) if not wells else float("nan")   # (the well path of the oracle's adapter solves single-threaded)
    best_gpu, best_cpu = max(rates.values()), np.nanmax([cpu1, cpun])
    print("%-62s %7d cells   GPU: %s   CPU-1 %7.1f   CPU-%d %7.1f   GPU / best CPU %5.1fx" %
          (name, case["Nb"], "   ".join(out), cpu1, a.threads, cpun, best_gpu / best_cpu), flush=True)
    table.append((name, case["Nb"], rates["ilu0"], rates["cpr"], cpu1, cpun))


table = []
print("host: %d CPUs in the affinity mask, CPU-N with %d threads" % (len(os.sched_getaffinity(0)), a.threads), flush=True)
s1 = pkg.decks.spe1_case()
run("configs[0]: SPE1CASE1 - its EQUIL state, DRSDT 0, its two wells", s1, None, wells=lambda: pkg.decks.spe1_wells(s1), drsdt=s1["drsdt"], drsdt_all=s1["drsdt_all_cells"])
c0 = pkg.decks.cartesian_case(10, 10, 3, state="mixed", heterogeneous=False)
run("           SPE1's grid, 10 x 10 x 3, rate sources", c0, five_spot(c0, 20.0))
for n3 in ((20, 20, 10), (32, 32, 32)):
    cc = pkg.decks.cartesian_case(*n3, state="mixed", heterogeneous=False)
    run("(ladder) %d x %d x %d homogeneous" % n3, cc, five_spot(cc, pkg.decks.BENCH_RATE_SM3_PER_DAY * n3[0] * n3[1] / 1e4))
c2 = pkg.decks.cartesian_case(24, 25, 15, state="mixed", heterogeneous=True)
run("configs[2]: SPE9-shaped, 24 x 25 x 15, heterogeneous", c2, five_spot(c2, 60.0))
c2w = pkg.decks.cartesian_case(24, 25, 15, dx=91.44, dy=91.44, dz=6.0, state="mixed", heterogeneous=True)
run("            the same grid at SPE9's cell size, its 26 wells as wells", c2w, None, wells=lambda: pkg.decks.spe9_shaped_wells(c2w, producer_bhp_limit=235e5))
c4, _, _ = helpers.norne_shaped_case(pkg)
run("configs[4]: Norne-shaped corner-point grid, 46 x 112 x 22", c4, scattered_sources(c4, 200.0))
c1 = pkg.decks.cartesian_case(50, 50, 50, state="mixed", heterogeneous=False)
run("(ladder) 50^3 homogeneous", c1, five_spot(c1, pkg.decks.BENCH_RATE_SM3_PER_DAY * 0.25))
print()
print("| configuration | cells | GPU ILU0 | GPU CPR | CPU, 1 thread | CPU, %d threads |" % a.threads)
print("|---|---|---|---|---|---|")
for name, nb, gi, gc, c1_, cn in table:
    print("| %s | %d | %.0f | %.0f | %.1f | %s |" % (name.strip(), nb, gi, gc, c1_, "-" if cn != cn else "%.1f" % cn))
