#!/usr/bin/env python3
"""What a host-side well model costs the device-resident Newton iteration at the bench's size: the 100^3 case with its five-spot as two
WELLS (wells.StandardWells: a water injector and an oil producer, one completion per layer - 100 each - on the bench's rates, the
producer with a BHP limit it never meets), under bench.py's time-step control, three ways inside one GPU session:
  sources    the bench itself: fixed-rate source terms, no well model
  per cell   the well model moving what its 200 perforated cells need (opmhip_get_iq_cells / opmhip_set_source_cells, ABI 11)
  whole grid the well model through the whole-grid calls (opmhip_get_iq: 544 B per cell back, opmhip_set_source: 96 B per cell over)
Prints Newton iterations/s and linear iterations per Newton iteration of each.    python tools/wells_at_scale.py [--n 100] [--steps 20]"""
import argparse, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=5)
a = ap.parse_args()
pkg = importlib.import_module("opm-autodiff_amd")
n = a.n
case = pkg.decks.cartesian_case(n, n, n, state="mixed", heterogeneous=False)
rate = pkg.decks.BENCH_RATE_SM3_PER_DAY * n * n / 1e4
src = pkg.decks.five_spot_source(case, rate_sm3_per_day=rate)


def two_wells():
    W = pkg.wells
    col = lambda i, j: [i + n * (j + n * k) for k in range(n)]
    tw = lambda cells: [W.peaceman_factor(case["perm"][c], case["dx"], case["dy"], case["dz"], 0.1524) for c in cells]
    ci, cp = col(0, 0), col(n - 1, n - 1)
    q = rate / 86400.0
    return W.StandardWells([W.Well("INJ", ci, tw(ci), case["depth"][ci[0]], False, ("rate", W.WATER, q), 1000e5, inj_phase="water"),
                            W.Well("PROD", cp, tw(cp), case["depth"][cp[0]], True, ("rate", W.OIL, q), 10e5)], case["depth"])


class WholeGrid:
    """the model without its per-cell calls: newton.BlackoilModelHip then asks for / hands over whole arrays"""

    def __init__(self, m):
        self._m = m

    def __getattr__(self, k):
        if k in ("iq_cells", "set_source_cells"):
            raise AttributeError(k)
        return getattr(self._m, k)


def run(name, with_wells, whole_grid=False):
    m = pkg.capi.HipModel(case, tolerance=1e-2, maxit=200, ilu_relaxation=0.9)
    m.set_state(case["pv"], case["meaning"])
    wells = None
    if with_wells:
        wells = two_wells()
    else:
        m.set_source(src)
    nm = pkg.newton.BlackoilModelHip(WholeGrid(m) if whole_grid else m, well_model=wells)
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
    extra = ""
    if wells is not None:
        extra = "   controls %s, q_inj %.1f m3/day, q_prod %.1f m3/day" % ("".join(w.control[0][0] for w in wells.wells), wells.x[0, 1] * 86400.0, -wells.x[1, 0] * 86400.0)
    print("%-44s %8.2f Newton its/s   %5.2f lin/Newton   %6.2f ms per Newton iteration%s" % (name, a.steps / el, lin / a.steps, 1e3 * el / a.steps, extra), flush=True)


print("%d^3 cells, %d Newton iterations after %d of warm-up, each variant in a context of its own" % (n, a.steps, a.warmup), flush=True)
for rep in range(2):
    run("sources (the bench, no well model)", False)
    run("two wells, 200 completions: per cell", True)
    run("two wells, 200 completions: whole grid", True, whole_grid=True)
