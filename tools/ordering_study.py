#!/usr/bin/env python3
"""Iteration counts of ILU0-BiCGStab (tol 1e-2) on a real Jacobian under different row orderings (CPU oracle)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("opm-autodiff_amd")
import oracle_bind
orc = oracle_bind.Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
case = pkg.decks.cartesian_case(n, n, n, state="mixed", heterogeneous=False)
src = pkg.decks.five_spot_source(case, rate_sm3_per_day=pkg.decks.BENCH_RATE_SM3_PER_DAY * (n / 100.0) ** 2)
m = oracle_bind.OracleModel(orc, case)
m.set_state(case["pv"], case["meaning"]); m.set_source(src)
dt = 10 * 86400.0
# take the Jacobian of the 2nd Newton iteration of a 10-day step (storage + flow both matter)
jac, res = m.assemble(dt, 0); x, r = m.solve(); m.update(x)
jac, res = m.assemble(dt, 1)
Nb, rp, ci = case["Nb"], case["rowptr"], case["col"]
idx = np.arange(Nb); i = idx % n; j = (idx // n) % n; k = idx // (n * n)

def perm_from_keys(*keys):
    order = np.lexsort(tuple(reversed(keys)))  # first key most significant
    fr = order.astype(np.int32); to = np.empty(Nb, np.int32); to[fr] = np.arange(Nb, dtype=np.int32)
    return to, fr

def run(name, to, fr):
    rr, rc, rv = orc.reorder_matrix(Nb, rp, ci, jac, to, fr)
    b = res.reshape(Nb, 3)[fr].reshape(-1)
    t0 = time.time(); x, r = orc.solve(Nb, rr, rc, rv, b, tol=1e-2, maxit=400, w=0.9); 
    print("%-34s it %5.1f conv %d  (%.1fs)" % (name, r.it, r.converged, time.time() - t0), flush=True)

ident = np.arange(Nb, dtype=np.int32)
run("natural", ident, ident)
run("red-black (i+j+k)", *perm_from_keys((i + j + k) % 2, idx))
for bs in (4,):
    bi, bj, bk = i // bs, j // bs, k // bs
    color = (bi % 2) + 2 * (bj % 2) + 4 * (bk % 2)
    brick = bi + (n // bs + 1) * (bj + (n // bs + 1) * bk)
    run("brick %d^3, 8 colours" % bs, *perm_from_keys(color, brick, idx))
for L in (2, 3, 4, 5, 8, 10):
    kseg = k // L
    color = (i + j + kseg) % 2
    seg = i + n * (j + n * kseg)
    run("z-line segments of %d, 2 colours" % L, *perm_from_keys(color, seg, k))
for L in ():
    # x-y tiles LxL full column? (plane blocks): colour by (i//L + j//L) parity, natural inside
    color = (i // L + j // L) % 2
    blk = (i // L) + (n // L + 1) * (j // L)
    run("xy-tile %dx%d columns, 2 colours" % (L, L), *perm_from_keys(color, blk, idx))

for L in (5, 10):
    iseg = i // L
    color = (iseg + j + k) % 2
    seg = iseg + n * (j + n * k)
    run("x-line segments of %d, 2 colours" % L, *perm_from_keys(color, seg, i))
