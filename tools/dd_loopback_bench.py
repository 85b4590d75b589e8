#!/usr/bin/env python3
"""Rehearsal of bench.py's multi-rank path on ONE GPU: `world` subdomains of n^3 cells each, one host thread per rank,
the loopback communicator instead of RCCL (same set_pattern_dd / halo / all-reduce code).  Reports Newton and linear
iteration counts per time step - the robustness check for the N = 2, 4, 8 runs only the driver can launch.
    python tools/dd_loopback_bench.py --world 8 --n 50 --steps 30"""
import argparse, importlib, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--n", type=int, default=50)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--reorder", default=None, help="default: the library's choice")
ap.add_argument("--preconditioner", default="ilu0", choices=["ilu0", "cpr", "cpr_trueimpes", "cpr_quasiimpes"])
ap.add_argument("--cpr-gather-rows", type=int, default=0, help="CPR: the pressure stage spans the ranks from each rank's first level of at most this many rows (0: the default, 100000; < 0: one hierarchy per subdomain)")
ap.add_argument("--fused-reductions", type=int, default=1, help="1 (default, as bench.py --gpus N > 1 with ILU0): one all-reduce per half iteration; 0: the reference's recurrence")
a = ap.parse_args()
pkg = importlib.import_module("opm-autodiff_amd")
group = "ddbench%d" % os.getpid()
out, err = [None] * a.world, [None] * a.world


def body(r):
    try:
        case = pkg.ras.cartesian_subdomain_case(a.n, a.world, r, state="mixed", heterogeneous=False)
        m = pkg.capi.HipModel(case, comm=("loopback", a.world, r, group), reorder=a.reorder, tolerance=1e-2, maxit=200, ilu_relaxation=0.9, preconditioner=a.preconditioner, cpr_gather_rows=a.cpr_gather_rows,
                              fused_reductions=a.fused_reductions if a.preconditioner == "ilu0" else 0)
        m.set_state(case["pv"], case["meaning"])
        m.set_source(case["source"])
        sim = bench.make_simulation(pkg, m)
        m.profile_enable(bench.PROFILE_EVERY)
        t0 = time.perf_counter()
        log = []
        for _ in range(a.steps):
            rep = sim.next_newton_iteration()
            log.append((sim.timesteps_done, sim.iteration, rep.total_linear_iterations))
        el = time.perf_counter() - t0
        out[r] = (el, log, sim.timesteps_done, sim.history, m.profile(), m.product_form())
    except Exception as e:  # noqa: BLE001
        err[r] = e
        raise


ts = [threading.Thread(target=body, args=(r,)) for r in range(a.world)]
[t.start() for t in ts]
[t.join() for t in ts]
if any(e is not None for e in err):
    raise SystemExit("rank failures: %r" % err)
el, log, done, hist, _, form = out[0]
print("product form on rank 0:", form, "(half-product form: its interior tiles; the boundary tiles run the whole product)")
print("world %d n %d %s: %d Newton iterations in %.2f s, %d time steps done, %d linear iterations" % (a.world, a.n, a.preconditioner, a.steps, el, done, sum(l[2] for l in log)))
print("(step, newton, linear its):", log)
print("time steps (days, newton its, accepted):", [(round(h[0] / bench.DAY, 3), h[1], h[2]) for h in hist])
assert all(o[1] == log for o in out), "ranks disagree on the iteration history"
# the communication spans (halo: pack -> exchange -> ghosts in; allreduce: local sums -> all-reduce; cpr_gather: the joined level's all-gather
# + cycle) of every bench.PROFILE_EVERY-th solve, per rank; on the loopback communicator the host drives the exchanges between two barriers,
# so these are rehearsal numbers for the plumbing, not xGMI times
lin = sum(l[2] for l in log)
print("linear iterations per Newton iteration: %.2f" % (lin / a.steps))
for k in ("halo", "allreduce", "cpr_gather", "spmv", "spmv_boundary", "ilu_apply", "cpr_amg", "vector"):
    per = [o[4].get(k, (0, 0.0)) for o in out]
    if not any(p[0] for p in per):
        continue
    avg = [p[1] / p[0] for p in per if p[0]]
    print("%-14s launches/rank %6d   avg ms max %.4f mean %.4f   total ms max %.2f mean %.2f" % (k, max(p[0] for p in per), max(avg), sum(avg) / len(avg),
                                                                                             max(p[1] for p in per), sum(p[1] for p in per) / len(per)))
