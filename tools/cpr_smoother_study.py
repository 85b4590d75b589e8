#!/usr/bin/env python3
"""Review r03 item 2a: the product's pairwise pressure hierarchy with a scalar ILU0 smoother on its upper levels, against today's
damped Jacobi and against the restated reference hierarchy (DuneLikeAmg), on stored steady-state systems (tools/ordering_study2.py
--fetch writes them).  Oracle only; CPR-BiCGStab iterations to 1e-2, quasi-IMPES weights, each hierarchy built from the system it
solves.  `lc`: the system is first permuted into the device's line-coloured order (z-chains of 10, 2 colours), so that level 0's
ILU0 is the line-coloured one a device sweep would run (aggregation still in natural order).
    python tools/cpr_smoother_study.py --dir /tmp/ord --size 100 --workers 8"""
import argparse, ctypes, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_bind

ap = argparse.ArgumentParser()
ap.add_argument("--dir", default="/tmp/ord")
ap.add_argument("--size", type=int, default=100)
ap.add_argument("--workers", type=int, default=8)
ap.add_argument("--out", default="")
ap.add_argument("--max-systems", type=int, default=0)
ap.add_argument("--only", default="", help="comma-separated substrings: run only the variants whose name contains one of them")
a = ap.parse_args()
pkg = importlib.import_module("opm-autodiff_amd")
n = a.size
case = pkg.decks.cartesian_case(n, n, n, state="mixed", heterogeneous=False)
Nb, rp, ci = case["Nb"], case["rowptr"], case["col"]
orc = oracle_bind.Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
orc.lib.orc_cpr_set_ilu_smoother.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
orc.lib.orc_cpr_set_aggregation.argtypes = [ctypes.c_void_p, ctypes.c_int]
idx = np.arange(Nb); I = idx % n; J = (idx // n) % n; K = idx // (n * n)

# name -> (reference AMG? [3: the product's hierarchy with the reference's kind of AGGREGATION], ILU levels, colour from, line-coloured level 0?)
VARIANTS = {
    "product with aggregates of 4-6 (reference's kind), Jacobi": (3, 0, -1, False),
    "product with aggregates of 4-6 + ILU0 level 0 (natural order)": (3, 1, -1, False),
    "product with aggregates of 4-6 + ILU0 levels 0-1 (natural / multi-colour)": (3, 2, 1, False),
    "product: Jacobi V(1,1)": (False, 0, -1, False),
    "reference-like: aggregates 4-6, ILU0": (True, 0, -1, False),
    "product + ILU0 level 0 (natural order)": (False, 1, -1, False),
    "product + ILU0 levels 0-1 (natural / stored)": (False, 2, -1, False),
    "product + ILU0 levels 0-1 (natural / multi-colour)": (False, 2, 1, False),
    "product + ILU0 all levels (stored)": (False, 99, -1, False),
    "product + ILU0 all levels (multi-colour from 1)": (False, 99, 1, False),
    "lc: product Jacobi": (False, 0, -1, True),
    "lc: product + ILU0 level 0 (line-coloured)": (False, 1, -1, True),
    "lc: product + ILU0 levels 0-1 (line-coloured / multi-colour)": (False, 2, 1, True),
    "lc: product + ILU0 levels 0-2 (line-coloured / multi-colour)": (False, 3, 1, True),
    "lc: product + ILU0 all levels (line-coloured / multi-colour)": (False, 99, 1, True),
}


def one(job):
    name, tag = job
    ref, ilu, colfrom, lc = VARIANTS[name]
    jac = np.ascontiguousarray(np.load(os.path.join(a.dir, "jac_%s.npy" % tag)))
    res = np.load(os.path.join(a.dir, "res_%s.npy" % tag))
    c = oracle_bind.OracleCpr(orc)
    rr, rc, rv, b = rp, ci, jac, res
    if lc:
        order = np.lexsort((K, I + n * (J + n * (K // 10)), (I + J + K // 10) % 2))
        fr = order.astype(np.int32); to = np.empty(Nb, np.int32); to[fr] = np.arange(Nb, dtype=np.int32)
        rr, rc, rv = orc.reorder_matrix(Nb, rp, ci, jac, to, fr)
        b = res.reshape(Nb, 3)[fr].reshape(-1).copy()
        c.set_natural_ids(fr)
    c.use_reference_amg(ref if ref != 3 else False)
    orc.lib.orc_cpr_set_aggregation(c.h, 1 if ref == 3 else 0)
    orc.lib.orc_cpr_set_ilu_smoother(c.h, ilu, colfrom)
    t0 = time.time()
    x, r = c.solve(Nb, rr, rc, rv, b, tol=1e-2, maxit=200)
    return name, tag, float(r.it), bool(r.converged), round(time.time() - t0, 1)


if __name__ == "__main__":
    import multiprocessing as mp
    with open(os.path.join(a.dir, "meta.json")) as f:
        meta = json.load(f)
    tags = [s["tag"] for s in meta["systems"]]
    if a.max_systems:
        tags = tags[:a.max_systems]
    names = [nm for nm in VARIANTS if not a.only or any(t in nm for t in a.only.split(","))]
    jobs = [(name, tag) for name in names for tag in tags]
    table = {}
    with mp.Pool(a.workers) as pool:
        for name, tag, it, conv, sec in pool.imap_unordered(one, jobs):
            table.setdefault(name, {})[tag] = it if conv else None
            print("%-64s %-8s it %5.1f conv %d (%.0fs)" % (name, tag, it, conv, sec), file=sys.stderr, flush=True)
    out = {"size": n, "systems": meta["systems"], "cpr_bicgstab_iterations": {k: table[k] for k in VARIANTS if k in table}}
    for k, v in out["cpr_bicgstab_iterations"].items():
        vals = [x for x in v.values() if x is not None]
        v["mean"] = float(np.mean(vals)) if vals else None
    line = json.dumps(out)
    if a.out:
        with open(a.out, "w") as f:
            f.write(line + "\n")
    print(line)
