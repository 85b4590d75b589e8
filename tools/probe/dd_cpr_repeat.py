#!/usr/bin/env python3
"""tests/test_gpu_dd.py::test_dd_cpr_pressure_stage_across_the_ranks[8-cpr_quasiimpes-6-1000] failed ONCE in seven runs of the suite with
"non-finite residual norm" in the first decomposed solve: its first half (eight loopback ranks, level 0 joined) repeated in one process,
every rank's iteration count and the finiteness of what it returns printed per round.    python tools/probe/dd_cpr_repeat.py [rounds]"""
import importlib, os, sys, threading, uuid
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("opm-autodiff_amd")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 25
world, n, rows, prec = 8, 6, 1000, "cpr_quasiimpes"
px, py, pz = pkg.ras.block_layout(world)
g = pkg.decks.cartesian_case(px * n, py * n, pz * n, state="mixed", heterogeneous=True)
parts = [pkg.ras.cartesian_subdomain_case(n, world, r, state="mixed", heterogeneous=True) for r in range(world)]
src = pkg.decks.five_spot_source(g, rate_sm3_per_day=30.0)
dt = 5 * 86400.0
bad = 0
for rd in range(rounds):
    group = "p" + uuid.uuid4().hex
    out, err = [None] * world, [None] * world

    def body(r):
        try:
            c = parts[r]
            m = pkg.capi.HipModel(c, comm=("loopback", world, r, group), reorder="line_coloring", preconditioner=prec, tolerance=1e-4, cpr_gather_rows=rows, cpr_amg_ilu_levels=0)
            m.set_state(c["pv"], c["meaning"])
            m.set_source(np.ascontiguousarray(src.reshape(-1, 3)[c["gids"]].reshape(-1)))
            j, res = m.assemble(dt, 0)
            fin = (bool(np.isfinite(j).all()), bool(np.isfinite(res).all()))
            try:
                sol = m.solve_jacobian_system()
                out[r] = (sol.it, sol.converged, sol.reduction, fin, bool(np.isfinite(m.get_result()).all()))
            except Exception as e:  # noqa: BLE001
                # what does the rank hold?  (collective calls must still be made by everybody: all ranks fail together or this hangs - bounded by the joins)
                w = m.get_cpr_weights() if hasattr(m, "get_cpr_weights") else None
                out[r] = ("FAILED", str(e)[:80], fin, None if w is None else bool(np.isfinite(w).all()))
        except BaseException as e:  # noqa: BLE001
            err[r] = e
    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(timeout=60) for t in ts]
    failed = [r for r in range(world) if out[r] is None or out[r][0] == "FAILED" or err[r] is not None]
    if failed:
        bad += 1
        print("round %d: ranks %r failed: %r %r" % (rd, failed, [out[r] for r in failed][:3], [repr(e)[:100] for e in err if e is not None][:2]), flush=True)
    else:
        print("round %d: its %r" % (rd, sorted(set(o[0] for o in out))), flush=True)
print("%d of %d rounds failed" % (bad, rounds))
