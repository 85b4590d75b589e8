#!/usr/bin/env python3
"""Newton iterations/s of the device path on the stand-ins of BASELINE.json's other configurations (the bench line is configs[1]):
configs[0] SPE1's 10 x 10 x 3 grid, configs[2] an SPE9-shaped 24 x 25 x 15 grid (heterogeneous permeability; rate sources in place of its wells), configs[4] the
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


def run(name, case, src, **kw):
    out = []
    for prec in ("ilu0", "cpr"):
        m = pkg.capi.HipModel(case, tolerance=1e-2, maxit=200, ilu_relaxation=0.9, preconditioner=prec,
                              cpr_amg_ilu_levels=a.cpr_amg_ilu_levels, **kw)   # no reorder argument: the library's default (auto)
        m.set_state(case["pv"], case["meaning"])
        m.set_source(src)
        sim = bench.make_simulation(pkg, m)
        for _ in range(a.warmup):
            sim.next_newton_iteration()
        m.synchronize()
        t0 = time.perf_counter()
        lin = 0
        for _ in range(a.steps):
            lin += sim.next_newton_iteration().total_linear_iterations
        m.synchronize()
        el = time.perf_counter() - t0
        out.append("%s: %.0f Newton its/s, %.1f linear iterations per Newton iteration" % (prec, a.steps / el, lin / a.steps))
    print("%-58s %7d cells   %s" % (name, case["Nb"], "   ".join(out)), flush=True)


c0 = pkg.decks.cartesian_case(10, 10, 3, state="mixed", heterogeneous=False)
run("configs[0]: SPE1's grid, 10 x 10 x 3", c0, five_spot(c0, 20.0))
c2 = pkg.decks.cartesian_case(24, 25, 15, state="mixed", heterogeneous=True)
run("configs[2]: SPE9-shaped, 24 x 25 x 15, heterogeneous", c2, five_spot(c2, 60.0))
c4, _, _ = helpers.norne_shaped_case(pkg)
run("configs[4]: Norne-shaped corner-point grid, 46 x 112 x 22", c4, scattered_sources(c4, 200.0))
c1 = pkg.decks.cartesian_case(50, 50, 50, state="mixed", heterogeneous=False)
run("(for scale) 50^3 homogeneous", c1, five_spot(c1, pkg.decks.BENCH_RATE_SM3_PER_DAY * 0.25))
