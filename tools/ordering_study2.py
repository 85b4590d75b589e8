#!/usr/bin/env python3
"""ILU0-BiCGStab iteration counts (tol 1e-2, w = 0.9, the bench's settings) of candidate ILU orderings on STEADY-STATE
Jacobians of the bench case (review r03 item 1a).  Two phases, separate processes:

  --fetch  (GPU)  the 100^3 bench case is advanced on the device to Newton iteration --at, then --n systems (Jacobian and
                  residual in the natural order, exactly as the device is about to solve them) are written to --dir;
  --study  (CPU)  every ordering below is applied to every system (oracle: reorder_matrix, then an exact block ILU0 of the
                  permuted matrix + BiCGStab, single-threaded per solve, --workers solves side by side); prints one JSON line.

Orderings: natural; red-black; z-line chains of L rows in 2 colours (today's line colouring); BOXES bx x by x bz in 2 / 4 / 8
colours with the natural order kept inside a box (block multi-colour ordering: an exact ILU0 of the permuted matrix).
    python tools/ordering_study2.py --fetch --at 200 --n 3 --dir /tmp/ord && python tools/ordering_study2.py --study --dir /tmp/ord
"""
import argparse, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

ap = argparse.ArgumentParser()
ap.add_argument("--fetch", action="store_true")
ap.add_argument("--study", action="store_true")
ap.add_argument("--cpu", action="store_true", help="--fetch with the oracle instead of the device (small sizes)")
ap.add_argument("--size", type=int, default=100)
ap.add_argument("--at", type=int, nargs="*", default=[200])
ap.add_argument("--n", type=int, default=2)
ap.add_argument("--dir", default="/tmp/ord")
ap.add_argument("--workers", type=int, default=8)
ap.add_argument("--set", default="main", help="which list of orderings: main | wide | focus")
ap.add_argument("--out", default="")
a = ap.parse_args()
pkg = importlib.import_module("opm-autodiff_amd")
n = a.size
case = pkg.decks.cartesian_case(n, n, n, state="mixed", heterogeneous=False)
Nb, rp, ci = case["Nb"], case["rowptr"], case["col"]

if a.fetch:
    import bench
    src = pkg.decks.five_spot_source(case, rate_sm3_per_day=pkg.decks.BENCH_RATE_SM3_PER_DAY * (n / 100.0) ** 2)
    if a.cpu:
        import oracle_bind
        orc = oracle_bind.Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
        o = oracle_bind.OracleModel(orc, case)
        o.set_state(case["pv"], case["meaning"]); o.set_source(src)
        m = oracle_bind.OracleAsHipModel(o, tol=1e-2, maxit=200, w=0.9, mode="post_scale", reorder="none", threads=8)
    else:
        m = pkg.capi.HipModel(case, reorder="line_coloring", tolerance=1e-2, maxit=200, ilu_relaxation=0.9, chain_length=10)
        m.set_state(case["pv"], case["meaning"]); m.set_source(src)
    sim = bench.make_simulation(pkg, m)
    os.makedirs(a.dir, exist_ok=True)
    meta = []
    k = 0
    for at in sorted(a.at):
        while k < at:
            sim.next_newton_iteration(); k += 1
        for q in range(a.n):
            if sim.iteration == 0:
                sim.next_newton_iteration(); k += 1     # starts a time step: take the following iteration's system
            dt, it = sim.dt, sim.iteration
            if a.cpu:
                jac, res = o.assemble(dt, it)
            else:
                jac, res = m.assemble(dt, it, fetch=True)
            tag = "%d_%d" % (at, q)
            np.save(os.path.join(a.dir, "jac_%s.npy" % tag), jac); np.save(os.path.join(a.dir, "res_%s.npy" % tag), res)
            rep = sim.next_newton_iteration(); k += 1
            meta.append({"tag": tag, "newton": k - 1, "dt_days": dt / 86400.0, "it_in_step": it, "device_linear_iterations": int(rep.total_linear_iterations)})
            print(meta[-1], file=sys.stderr, flush=True)
    with open(os.path.join(a.dir, "meta.json"), "w") as f:
        json.dump({"size": n, "systems": meta}, f)
    sys.exit(0)

# ---------------------------------------------------------------------------------------------------------------------
import oracle_bind
orc = oracle_bind.Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
idx = np.arange(Nb); I = idx % n; J = (idx // n) % n; K = idx // (n * n)


def perm_from_keys(*keys):
    order = np.lexsort(tuple(reversed(keys)))  # first key most significant
    fr = order.astype(np.int32); to = np.empty(Nb, np.int32); to[fr] = np.arange(Nb, dtype=np.int32)
    return to, fr


def box_perm(bx, by, bz, ncolors):
    bi, bj, bk = I // bx, J // by, K // bz
    nbx, nby = (n + bx - 1) // bx, (n + by - 1) // by
    if ncolors == 2:
        color = (bi + bj + bk) % 2
    elif ncolors == 4:
        color = (bi % 2) + 2 * (bj % 2) if bz >= n else ((bi + bk) % 2) + 2 * ((bj + bk) % 2)
    else:
        color = (bi % 2) + 2 * (bj % 2) + 4 * (bk % 2)
    box = bi + nbx * (bj + nby * bk)
    return perm_from_keys(color, box, idx)


def orderings(which):
    ident = np.arange(Nb, dtype=np.int32)
    out = [("natural", lambda: (ident, ident)),
           ("red-black", lambda: perm_from_keys((I + J + K) % 2, idx)),
           ("z-chains 10, 2 colours (today)", lambda: perm_from_keys((I + J + K // 10) % 2, I + n * (J + n * (K // 10)), K)),
           ("z-chains 20, 2 colours", lambda: perm_from_keys((I + J + K // 20) % 2, I + n * (J + n * (K // 20)), K))]
    if which == "focus":
        for b in ((2, 2, 10), (4, 8, 10), (5, 5, 10), (8, 8, 10), (10, 10, 10), (10, 10, 20)):
            out.append(("box %dx%dx%d, 2 colours" % b, lambda b=b: box_perm(*b, 2)))
        return [o for o in out if not o[0].startswith(("red", "z-chains 20"))]
    boxes = [(2, 2, 10), (4, 4, 10), (4, 8, 10), (5, 5, 10), (5, 5, 20), (8, 8, 8), (10, 10, 10), (4, 4, 20), (2, 4, 10), (4, 4, 5), (2, 2, 20), (1, 2, 10), (1, 4, 10), (2, 16, 10), (1, 32, 10)]
    if which == "wide":
        boxes += [(8, 8, 10), (10, 10, 20), (20, 20, 20), (4, 4, 100), (10, 10, 100), (4, 8, 20), (8, 8, 4), (16, 16, 4), (25, 25, 25)]
    for b in boxes:
        out.append(("box %dx%dx%d, 2 colours" % b, lambda b=b: box_perm(*b, 2)))
    for b in ((4, 4, 10), (8, 8, 8), (10, 10, 10)):
        out.append(("box %dx%dx%d, 8 colours" % b, lambda b=b: box_perm(*b, 8)))
    return out


def one(job):
    name, tag = job
    jac = np.load(os.path.join(a.dir, "jac_%s.npy" % tag), mmap_mode="r")
    res = np.load(os.path.join(a.dir, "res_%s.npy" % tag))
    to, fr = dict(ORD)[name]()
    t0 = time.time()
    rr, rc, rv = orc.reorder_matrix(Nb, rp, ci, np.ascontiguousarray(jac), to, fr)
    b = res.reshape(Nb, 3)[fr].reshape(-1).copy()
    x, r = orc.solve(Nb, rr, rc, rv, b, tol=1e-2, maxit=400, w=0.9)
    return name, tag, float(r.it), bool(r.converged), round(time.time() - t0, 1)


ORD = orderings(a.set)
if a.study:
    import multiprocessing as mp
    with open(os.path.join(a.dir, "meta.json")) as f:
        meta = json.load(f)
    tags = [s["tag"] for s in meta["systems"]]
    jobs = [(name, tag) for name, _ in ORD for tag in tags]
    table = {}
    with mp.Pool(a.workers) as pool:
        for name, tag, it, conv, sec in pool.imap_unordered(one, jobs):
            table.setdefault(name, {})[tag] = it if conv else None
            print("%-36s %-8s it %5.1f conv %d (%.0fs)" % (name, tag, it, conv, sec), file=sys.stderr, flush=True)
    out = {"size": n, "systems": meta["systems"], "iterations": {k: table[k] for k, _ in ORD if k in table}}
    for k, v in out["iterations"].items():
        vals = [x for x in v.values() if x is not None]
        v["mean"] = float(np.mean(vals)) if vals else None
    line = json.dumps(out)
    if a.out:
        with open(a.out, "w") as f:
            f.write(line + "\n")
    print(line)
