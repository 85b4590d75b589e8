"""Synthetic block systems shared by the tests (no reference data involved)."""
import numpy as np


def cartesian_pattern(nx, ny, nz):
    """7-point block pattern, natural order i + nx*(j + ny*k), columns ascending."""
    Nb = nx * ny * nz
    idx = np.arange(Nb).reshape(nz, ny, nx)
    nbrs = [[] for _ in range(Nb)]
    def link(a, b):
        for p, q in zip(a.reshape(-1), b.reshape(-1)):
            nbrs[p].append(q)
            nbrs[q].append(p)
    link(idx[:, :, :-1], idx[:, :, 1:])
    link(idx[:, :-1, :], idx[:, 1:, :])
    link(idx[:-1, :, :], idx[1:, :, :])
    rowptr = np.zeros(Nb + 1, np.int32)
    cols = []
    for i in range(Nb):
        c = sorted(nbrs[i] + [i])
        cols.extend(c)
        rowptr[i + 1] = len(cols)
    return Nb, rowptr, np.array(cols, np.int32)


def laplace_block_system(nx, ny, nz, seed=0, dominance=1.5):
    """Block 7-point system with random dense 3x3 blocks, made block-diagonally dominant so that ILU0
    BiCGStab converges: a stand-in for a Jacobian when only the linear algebra is under test."""
    rng = np.random.default_rng(seed)
    Nb, rowptr, col = cartesian_pattern(nx, ny, nz)
    nnzb = int(rowptr[-1])
    val = rng.uniform(-1.0, 0.0, size=(nnzb, 3, 3)) * 0.3
    for i in range(Nb):
        ks = np.arange(rowptr[i], rowptr[i + 1])
        off = ks[col[ks] != i]
        dk = ks[col[ks] == i][0]
        s = np.abs(val[off]).sum(axis=(0, 2))  # row sums per block row
        val[dk] = rng.uniform(-0.1, 0.1, size=(3, 3))
        val[dk][np.arange(3), np.arange(3)] = dominance * (s + 0.5) + rng.uniform(0, 0.2, 3)
    return Nb, rowptr, col, np.ascontiguousarray(val.reshape(-1))


def random_block_system(Nb, pattern="tridiag", seed=0, extra=0):
    """Irregular patterns: 'tridiag' block-tridiagonal, 'random' adds `extra` symmetric random couplings
    per row (row lengths vary), always with a dominant diagonal."""
    rng = np.random.default_rng(seed)
    nb = [set([i]) for i in range(Nb)]
    for i in range(Nb - 1):
        nb[i].add(i + 1)
        nb[i + 1].add(i)
    if pattern == "random":
        for i in range(Nb):
            for j in rng.choice(Nb, size=rng.integers(0, extra + 1), replace=False):
                nb[i].add(int(j))
                nb[int(j)].add(i)
    rowptr = np.zeros(Nb + 1, np.int32)
    cols = []
    for i in range(Nb):
        cols.extend(sorted(nb[i]))
        rowptr[i + 1] = len(cols)
    col = np.array(cols, np.int32)
    nnzb = len(cols)
    val = rng.uniform(-1, 1, size=(nnzb, 3, 3)) * 0.2
    for i in range(Nb):
        ks = np.arange(rowptr[i], rowptr[i + 1])
        dk = ks[col[ks] == i][0]
        s = np.abs(val[ks]).sum(axis=(0, 2))
        val[dk][np.arange(3), np.arange(3)] = 1.5 * (s + 0.5)
    return Nb, rowptr, col, np.ascontiguousarray(val.reshape(-1))


def oracle_solve_in_order(orc, Nb, rp, ci, val, b, to, fr, wells=None, **kw):
    """ILU0-BiCGStab of the oracle on the system permuted by (toOrder, fromOrder) - the ordering the device reports
    through opmhip_get_ordering - i.e. natural-order block ILU0 of reorderBlockedMatrixByPattern's matrix
    (bda/Reorder.cpp:179-207), result mapped back to the natural order."""
    rr, rc, rv = orc.reorder_matrix(Nb, rp, ci, val, to, fr)
    rb = np.ascontiguousarray(b.reshape(Nb, 3)[fr].reshape(-1))
    W = None
    if wells:
        W = dict(wells)
        W["Ccols"] = np.ascontiguousarray(to[wells["Ccols"]], np.int32)
        W["Bcols"] = np.ascontiguousarray(to[wells["Bcols"]], np.int32)
    x, res = orc.solve(Nb, rr, rc, rv, rb, reorder="none", wells=W, **kw)
    return np.ascontiguousarray(x.reshape(Nb, 3)[to].reshape(-1)), res


# ---- wet gas / rock compaction test cases ---------------------------------------------------------------------------
def wetgas_fluid(pkg, rocktab=None, pvtg_region=1):
    """The SPE1 fluid (PVTO, PVTW, SWOF, SGOF, ROCK) with Norne's PVTG table in place of PVDG (tests/golden/norne_pvt.json,
    from the reference's tests/norne_pvt.data): oil may vaporise into the gas phase.  rocktab: optional list of ROCKTAB
    regions, rows (p, pore-volume multiplier, transmissibility multiplier)."""
    import json
    import os
    fl, d = pkg.fluid.spe1_fluid()
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "norne_pvt.json")) as f:
        pvtg = json.load(f)["pvtg"][pvtg_region]
    pvt = [dict(fl.pvt[0])]
    pvt[0]["pvtg"] = pvtg
    pvt[0].pop("pvdg", None)
    return pkg.fluid.Fluid(pvt, fl.sat, rock_pref=fl.rock_pref, rock_cr=fl.rock_cr, rocktab=rocktab)


def rv_sat(fluid, pg, region=0):
    nodes = fluid.pvt[region]["pvtg"]
    xp = np.array([n["pg"] for n in nodes])
    fp = np.array([n["rv"][0] for n in nodes])
    pg = np.asarray(pg, float)
    s = np.clip(np.searchsorted(xp, pg, side="right") - 1, 0, len(xp) - 2)
    return fp[s] + (fp[s + 1] - fp[s]) * (pg - xp[s]) / (xp[s + 1] - xp[s])


def wetgas_case(pkg, nx, ny, nz, rocktab=None, seed=5, **kw):
    """Cartesian case with all three primary-variable meanings: gas cap without oil (Sw, pg, Rv) on top, three-phase cells
    (Sw, po, Sg) in the middle, undersaturated oil (Sw, po, Rs) below."""
    fl = wetgas_fluid(pkg, rocktab)
    case = pkg.decks.cartesian_case(nx, ny, nz, state="mixed", fluid=fl, **kw)
    rng = np.random.default_rng(seed)
    pv = case["pv"].reshape(-1, 3).copy()
    m = case["meaning"].copy()
    d = case["depth"]
    top = d < np.quantile(d, 0.3)
    m[top] = 2                                    # Sw_pg_Rv
    pv[top, 2] = rv_sat(fl, pv[top, 1]) * rng.uniform(0.5, 0.95, top.sum())
    case["pv"], case["meaning"] = np.ascontiguousarray(pv.reshape(-1)), m
    if rocktab:
        case["rocknum"] = (np.arange(case["Nb"]) % len(rocktab)).astype(np.int32)
        case["overburden"] = 20e5 + 50.0 * (d - d.min())
    return case


ROCKTAB_2 = [[[100e5, 0.96, 0.90], [200e5, 0.985, 0.96], [300e5, 1.0, 1.0], [400e5, 1.012, 1.03]],
             [[50e5, 0.90, 0.80], [250e5, 0.99, 0.97], [450e5, 1.03, 1.08]]]
