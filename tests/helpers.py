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


# water-induced compaction (ROCK2D / ROCK2DTR over ROCKWNOD's saturation nodes), two rock regions
ROCK2D_2 = [
    {"pressure": [100e5, 200e5, 300e5, 400e5], "sw": [0.0, 0.1, 0.3, 0.6],
     "pv_mult": [[0.96, 0.95, 0.93, 0.90], [0.985, 0.975, 0.955, 0.93], [1.0, 0.99, 0.97, 0.945], [1.012, 1.0, 0.98, 0.955]],
     "trans_mult": [[0.90, 0.88, 0.84, 0.80], [0.96, 0.94, 0.90, 0.85], [1.0, 0.98, 0.94, 0.89], [1.03, 1.01, 0.97, 0.92]]},
    {"pressure": [50e5, 250e5, 450e5], "sw": [0.0, 0.2, 0.5],
     "pv_mult": [[0.90, 0.88, 0.85], [0.99, 0.97, 0.94], [1.03, 1.01, 0.98]],
     "trans_mult": [[0.80, 0.78, 0.74], [0.97, 0.95, 0.90], [1.08, 1.05, 1.0]]},
]


# ---- BASELINE configs[4]: a Norne-shaped faulted corner-point grid ------------------------------------------------------------
def norne_shaped_grid(pkg, seed=17):
    """46 x 112 x 22 cells (Norne's dimensions) as COORD / ZCORN: a dome with dipping flanks, slightly sheared pillars, three
    faults (throws of 1 to 4 layers, one of them dying out along its length), a layer pinched out over part of the field and
    an irregular active region of about 44 000 cells.  Synthetic - the deck itself is not in the reference tree - but with
    the features that make Norne's connectivity irregular: cells meeting several cells across a fault, missing neighbours,
    connections across pinched-out layers.  -> (geometry dict of transmissibility.cornerpoint_faces, dims, actnum)"""
    T = pkg.transmissibility
    nx, ny, nz = 46, 112, 22
    rng = np.random.default_rng(seed)
    dx, dy = 100.0, 100.0
    ip, jp = np.meshgrid(np.arange(nx + 1), np.arange(ny + 1))            # [jp, ip]
    xs, ys = ip * dx, jp * dy
    dome = 2500.0 + 120.0 * (((xs - 2300.0) / 2300.0) ** 2 + ((ys - 5600.0) / 5600.0) ** 2) + 0.01 * xs
    coord = np.zeros((ny + 1, nx + 1, 6))
    coord[..., 0], coord[..., 1], coord[..., 2] = xs - 6.0, ys + 4.0, 2300.0      # pillars lean by 12 m / 8 m over 600 m
    coord[..., 3], coord[..., 4], coord[..., 5] = xs + 6.0, ys - 4.0, 2900.0
    thick = rng.uniform(3.0, 9.0, nz)
    thick[9] = 0.02                                                      # the layer that pinches out
    ztop = np.concatenate([[0.0], np.cumsum(thick)])
    ci, cj = np.meshgrid(np.arange(nx), np.arange(ny))                    # [j, i] cell indices
    throw = np.zeros((ny, nx))
    throw += np.where(ci >= 15, 14.0, 0.0)                                # fault 1: the whole length
    throw += np.where((ci >= 30), -22.0 * np.clip((cj - 20) / 60.0, 0.0, 1.0), 0.0)   # fault 2: grows from nothing along j
    throw += np.where((cj >= 70) & (ci >= 8), 9.0, 0.0)                   # fault 3: across, ends inside the field
    z = np.zeros((nz, 2, ny, 2, nx, 2))
    for jj in range(2):
        for ii in range(2):
            surf = dome[jj:jj + ny, ii:ii + nx] + throw                   # depth of the top surface at this corner's pillar
            for kk in range(2):
                z[:, kk, :, jj, :, ii] = surf[None, :, :] + ztop[kk:kk + nz, None, None]
    act = np.ones((nz, ny, nx), bool)
    blob = ((ci - 22.0) / 21.0) ** 2 + ((cj - 55.0) / 54.0) ** 2 + 0.08 * np.sin(ci / 3.0) * np.cos(cj / 5.0) < 0.665
    act &= blob[None, :, :]
    act[:3] &= (((ci - 20.0) / 12.0) ** 2 + ((cj - 50.0) / 30.0) ** 2 < 1.0)[None, :, :]   # the top layers only over the crest
    act[9] = False                                                        # pinched out everywhere (0.02 m)
    act[15, 40:, :20] = False                                             # a thick inactive patch: a real barrier
    act &= rng.random((nz, ny, nx)) > 0.01                                # scattered inactive cells
    g = T.cornerpoint_faces(nx, ny, nz, coord.reshape(-1, 6), z.reshape(-1), actnum=act.reshape(-1), max_fault_throw=6, pinch=0.5)
    return g, (nx, ny, nz), act.reshape(-1)


def norne_shaped_case(pkg, seed=17):
    """the grid above with log-normal permeabilities, NTG, a MULTZ barrier, the SPE1 fluid and a gas cap over undersaturated
    oil in hydrostatic-like pressure: a case dict for capi.HipModel / oracle_bind.OracleModel"""
    T = pkg.transmissibility
    g, dims, act = norne_shaped_grid(pkg, seed)
    n, F = g["n"], g["faces"]
    rng = np.random.default_rng(seed + 1)
    perm = np.exp(rng.normal(np.log(200.0), 1.0, (n, 1))) * np.array([1.0, 1.0, 0.1]) * 9.869233e-16
    lay = g["cart"] // (dims[0] * dims[1])
    multz = np.where(lay == 14, 0.01, 1.0)
    t = T.face_transmissibilities(F, g["centroid"], perm, ntg=rng.uniform(0.5, 1.0, n), mult={"Z+": multz})
    pat = T.connections_to_pattern(n, F["cell1"], F["cell2"], t, g["face_area"])
    fl = pkg.fluid.spe1_fluid()[0]
    depth = g["depth"]
    p = 250e5 + 7000.0 * (depth - depth.min()) * (1.0 + rng.uniform(-0.005, 0.005, n))
    meaning = np.where(depth < np.quantile(depth, 0.3), pkg.decks.SW_PO_SG, pkg.decks.SW_PO_RS).astype(np.uint8)
    pv = np.zeros((n, 3))
    pv[:, 0] = 0.2 + rng.uniform(-0.02, 0.02, n)
    pv[:, 1] = p
    pv[:, 2] = np.where(meaning == pkg.decks.SW_PO_SG, 0.1 + rng.uniform(-0.02, 0.02, n), 0.8 * pkg.decks.rs_sat(fl, p))
    case = dict(Nb=n, rowptr=pat["rowptr"], col=pat["col"], trans=pat["trans"], area=pat["area"], poro=rng.uniform(0.15, 0.3, n),
                volume=np.ascontiguousarray(g["volume"]), depth=np.ascontiguousarray(depth), fluid=fl,
                pv=np.ascontiguousarray(pv.reshape(-1)), meaning=meaning)
    return case, g, dims


# ---- relative-permeability hysteresis: a drainage and an imbibition saturation region ------------------------------------------
def hysteresis_sat_regions(pkg):
    """[drainage, imbibition] saturation regions from the SPE1 tables: the imbibition curves trap the non-wetting phases - gas
    becomes immobile below Sg = 0.15 (drainage: 0.02), oil below So = 1 - 0.78 (drainage: 1 - 0.84) - and carry the wetting
    phases a little lower, so that EHYSTR's second item (0: wetting phases on the drainage curves, 1: on the imbibition curves)
    can be told apart; capillary pressures made non-zero so that they are seen to stay on the drainage curves"""
    fl, d = pkg.fluid.spe1_fluid()
    swof = np.array(d["swof"], float)
    sgof = np.array(d["sgof"], float)
    swof[:, 3] = 0.3e5 * (1.0 - (swof[:, 0] - swof[0, 0]) / (1.0 - swof[0, 0])) ** 2
    sgof[:, 3] = 0.2e5 * (sgof[:, 0] / sgof[-1, 0]) ** 2
    swi, sgi = swof.copy(), sgof.copy()
    # imbibition krg: the drainage curve squeezed into [0.15, max]: krg_i(Sg) = krg_d(Sgcr_d + (Sg - 0.15) (Sgmax - Sgcr_d) / (Sgmax - 0.15))
    sgmax, sgcr_d, sgcr_i = sgof[-1, 0], 0.02, 0.15
    sg_src = np.where(sgof[:, 0] <= sgcr_i, sgcr_d * sgof[:, 0] / sgcr_i, sgcr_d + (sgof[:, 0] - sgcr_i) * (sgmax - sgcr_d) / (sgmax - sgcr_i))
    sgi[:, 1] = np.interp(sg_src, sgof[:, 0], sgof[:, 1])
    sgi[:, 2] = sgof[:, 2] * 0.9                     # krog on the imbibition curve (wetting phase of the gas-oil system)
    # imbibition krow: oil immobile from Sw = 0.78 on (drainage: 0.84)
    sw_src = np.minimum(1.0, swof[0, 0] + (swof[:, 0] - swof[0, 0]) * (0.84 - swof[0, 0]) / (0.78 - swof[0, 0]))
    swi[:, 2] = np.interp(sw_src, swof[:, 0], swof[:, 2])
    swi[:, 1] = swof[:, 1] * 0.8                     # krw on the imbibition curve (wetting phase of the oil-water system)
    return [dict(swof=swof.tolist(), sgof=sgof.tolist()), dict(swof=swi.tolist(), sgof=sgi.tolist())]


def hysteresis_case(pkg, nx, ny, nz, wetgas=False, **kw):
    """Cartesian three-phase case whose cells have a drainage (SATNUM 1) and an imbibition (IMBNUM 2) saturation region; the fluid
    carries the extended record (pc_scaling) that opmhip_set_hysteresis needs"""
    base = wetgas_fluid(pkg) if wetgas else pkg.fluid.spe1_fluid()[0]
    fl = pkg.fluid.Fluid(base.pvt, hysteresis_sat_regions(pkg), rock_pref=base.rock_pref, rock_cr=base.rock_cr, pc_scaling=True)
    case = pkg.decks.cartesian_case(nx, ny, nz, state="mixed", fluid=fl, **kw)
    case["satnum"] = np.zeros(case["Nb"], np.int32)
    case["imbnum"] = np.ones(case["Nb"], np.int32)
    return case
