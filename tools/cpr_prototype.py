#!/usr/bin/env python3
"""Prototype of the CPR preconditioner planned for the device (numpy/scipy on the CPU, development tool): quasi-IMPES
weights, pressure matrix, pairwise-aggregation AMG with Jacobi smoothing and damped prolongation, ILU0 post-smoothing,
inside BiCGStab - iteration counts against plain ILU0 on Jacobians of the synthetic case."""
import importlib, os, sys, time
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle_bind
pkg = importlib.import_module("opm-autodiff_amd")
orc = oracle_bind.Oracle("oracle/liboracle.so")

def weights_quasiimpes(Nb, rp, ci, val):
    blk = val.reshape(-1, 3, 3)
    row = np.repeat(np.arange(Nb), np.diff(rp))
    D = blk[ci == row]                      # diagonal blocks
    e = np.zeros((Nb, 3)); e[:, 1] = 1.0
    w = np.linalg.solve(np.transpose(D, (0, 2, 1)), e[:, :, None])[:, :, 0]
    w /= np.abs(w).max(axis=1)[:, None]
    return w

def pressure_matrix(Nb, rp, ci, val, w):
    blk = val.reshape(-1, 3, 3)
    row = np.repeat(np.arange(Nb), np.diff(rp))
    a = (blk[:, :, 1] * w[row]).sum(axis=1)
    return sp.csr_matrix((a, ci, rp), shape=(Nb, Nb))

def pairwise(A, beta=0.25):
    """one pairwise matching pass: every node with its strongest (most negative) unmatched neighbour"""
    n = A.shape[0]
    rp, ci, v = A.indptr, A.indices, A.data
    agg = -np.ones(n, np.int64)
    na = 0
    for i in range(n):
        if agg[i] >= 0: continue
        best, bv = -1, 0.0
        mx = 0.0
        for k in range(rp[i], rp[i + 1]):
            if ci[k] != i: mx = max(mx, -v[k])
        for k in range(rp[i], rp[i + 1]):
            j = ci[k]
            if j == i or agg[j] >= 0: continue
            s = -v[k]
            if s > bv and s >= beta * mx: best, bv = j, s
        agg[i] = na
        if best >= 0: agg[best] = na
        na += 1
    return agg, na

def coarsen(A, passes=2):
    n = A.shape[0]
    agg = np.arange(n)
    Ac = A
    for _ in range(passes):
        a, na = pairwise(Ac)
        P = sp.csr_matrix((np.ones(len(a)), (np.arange(len(a)), a)), shape=(len(a), na))
        Ac = (P.T @ Ac @ P).tocsr()
        Ac.sort_indices()
        agg = a[agg]
    return agg, Ac

class AMG:
    def __init__(self, A, omega=0.67, damp=1.6, coarse=64, maxlevel=15, passes=2):
        self.lv = []
        t0 = time.time()
        while A.shape[0] > coarse and len(self.lv) < maxlevel:
            agg, Ac = coarsen(A, passes)
            self.lv.append((A, 1.0 / A.diagonal(), agg))
            if Ac.shape[0] >= 0.8 * A.shape[0]:
                A = Ac
                break
            A = Ac
        self.Ac = A.toarray()
        self.omega, self.damp = omega, damp
        print("AMG levels:", [l[0].shape[0] for l in self.lv] + [A.shape[0]], "nnz", [l[0].nnz for l in self.lv], "setup %.1fs" % (time.time() - t0))
    def vcycle(self, b, l=0):
        if l == len(self.lv):
            return np.linalg.solve(self.Ac, b)
        A, Dinv, agg = self.lv[l]
        x = self.omega * Dinv * b                       # pre-smoothing from x = 0
        r = b - A @ x
        rc = np.bincount(agg, weights=r, minlength=agg.max() + 1)
        xc = self.vcycle(rc, l + 1)
        x += self.damp * xc[agg]
        x += self.omega * Dinv * (b - A @ x)           # post-smoothing
        return x

def bicgstab(Aop, b, prec, tol=1e-2, maxit=200):
    x = np.zeros_like(b); r = b.copy(); rw = r.copy(); p = r.copy(); v = np.zeros_like(b)
    rho = rw @ r; norm0 = np.linalg.norm(r); it = 0.0
    alpha = omega = 1.0
    while it < maxit:
        if it > 0:
            beta = (rho / rhop) * (alpha / omega)
            p = (p - omega * v) * beta + r
        y = prec(p); v = Aop(y)
        alpha = rho / (rw @ v)
        x += alpha * y; r -= alpha * v
        it += 0.5
        if np.linalg.norm(r) < tol * norm0: break
        z = prec(r); t = Aop(z)
        omega = (t @ r) / (t @ t)
        x += omega * z; r -= omega * t
        it += 0.5
        if np.linalg.norm(r) < tol * norm0: break
        rhop = rho; rho = rw @ r
    return x, it

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
case = pkg.decks.cartesian_case(n, n, n, state="mixed", heterogeneous=len(sys.argv) > 2)
src = pkg.decks.five_spot_source(case, rate_sm3_per_day=pkg.decks.BENCH_RATE_SM3_PER_DAY * (n / 100.0) ** 2)
o = oracle_bind.OracleModel(orc, case); o.set_state(case["pv"], case["meaning"]); o.set_source(src)
Nb, rp, ci = case["Nb"], case["rowptr"], case["col"]
for dt_days, its in ((1, 2), (10, 3)):
    dt = dt_days * 86400.0
    for it in range(its):
        jac, res = o.assemble(dt, it)
        A = sp.bsr_matrix((jac.reshape(-1, 3, 3), ci, rp), shape=(3 * Nb, 3 * Nb)).tocsr()
        x0, r0 = o.solve(tol=1e-2)
        lu = orc.ilu0_factor(Nb, rp, ci, jac)
        w = weights_quasiimpes(Nb, rp, ci, jac)
        Ap = pressure_matrix(Nb, rp, ci, jac, w)
        amg = AMG(Ap)
        def cpr(d):
            rc = (d.reshape(Nb, 3) * w).sum(axis=1)
            xc = amg.vcycle(rc)
            v = np.zeros((Nb, 3)); v[:, 1] = xc
            v = v.reshape(-1)
            rr = d - A @ v
            return v + orc.ilu0_apply(Nb, rp, ci, lu, rr, w=1.0)
        x1, it1 = bicgstab(lambda y: A @ y, res, cpr)
        x2, it2 = bicgstab(lambda y: A @ y, res, lambda d: orc.ilu0_apply(Nb, rp, ci, lu, d, w=0.9))
        print("dt %2d d newton %d : ILU0 (oracle) %.1f  ILU0 (here) %.1f  CPR %.1f" % (dt_days, it, r0.it, it2, it1), flush=True)
        o.update(x0)
