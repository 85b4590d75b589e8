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


# ---- the deck of the reference's tests/test_ecl_output.cc ----------------------------------------------------------------
def summary_deck_case(pkg):
    """tests/SUMMARY_DECK_NON_CONSTANT_POROSITY.DATA as a case (METRIC -> SI): 10 x 10 x 10 cells of 1 m^3, porosity 0.1 in the upper
    five layers and 0.2 below, PVTO with one saturated node at Rs = 0 and an undersaturated branch at Rs = 1 (1 / B_o grows linearly from
    0.1 at 1 bar to 1 at 10 bar), PVTW B_w = 1000 with no compressibility, PVDG, ROCK 3e-6 / bar at 14.7 bar, SWOF / SGOF of the deck;
    state from its SOLUTION section: p = 1 ... 10 bar layer by layer, S_w = 0.2, S_g = 0, Rs = 0 (undersaturated oil everywhere).
    FIPNUM: 400 cells region 1, 200 region 2, 400 region 3.  Returns (case, fipnum)."""
    import numpy as np
    bar, cP = 1e5, 1e-3
    swof = [[0.12, 0, 1, 0], [0.18, 4.64876033057851e-8, 1, 0], [0.24, 0.000000186, 0.997, 0], [0.3, 4.18388429752066e-7, 0.98, 0],
            [0.36, 7.43801652892562e-7, 0.7, 0], [0.42, 1.16219008264463e-6, 0.35, 0], [0.48, 1.67355371900826e-6, 0.2, 0],
            [0.54, 2.27789256198347e-6, 0.09, 0], [0.6, 2.97520661157025e-6, 0.021, 0], [0.66, 3.7654958677686e-6, 0.01, 0],
            [0.72, 4.64876033057851e-6, 0.001, 0], [0.78, 0.000005625, 0.0001, 0], [0.84, 6.69421487603306e-6, 0, 0],
            [0.91, 8.05914256198347e-6, 0, 0], [1, 0.00001, 0, 0]]
    sgof = [[0, 0, 1, 0], [0.001, 0, 1, 0], [0.02, 0, 0.997, 0], [0.05, 0.005, 0.980, 0], [0.12, 0.025, 0.700, 0], [0.2, 0.075, 0.350, 0],
            [0.25, 0.125, 0.200, 0], [0.3, 0.190, 0.090, 0], [0.4, 0.410, 0.021, 0], [0.45, 0.60, 0.010, 0], [0.5, 0.72, 0.001, 0],
            [0.6, 0.87, 0.0001, 0], [0.7, 0.94, 0.000, 0], [0.85, 0.98, 0.000, 0], [0.88, 0.984, 0.000, 0]]
    pvt = [dict(pvtw=[1 * bar, 1000.0, 0.0, 0.318 * cP, 0.0], density=[53.66, 64.49, 0.0533],
                pvdg=[[1 * bar, 100.0, 1 * cP], [10 * bar, 10.0, 1 * cP]],
                pvto=[dict(rs=0.0, p=[1 * bar], bo=[10.0], mu=[1 * cP]), dict(rs=1.0, p=[1 * bar, 10 * bar], bo=[10.001, 1.0], mu=[1 * cP, 1 * cP])])]
    fl = pkg.fluid.Fluid(pvt, [dict(swof=swof, sgof=sgof)], rock_pref=14.7 * bar, rock_cr=3e-6 / bar)
    with np.errstate(divide="ignore", invalid="ignore"):   # (the generator's default state interpolates RsSat(p) between the deck's two nodes at 1 bar; replaced below)
        case = pkg.decks.cartesian_case(10, 10, 10, dx=1.0, dy=1.0, dz=1.0, top=1.0, poro=0.1, perm_md=500.0, state="undersaturated", perturb=False, fluid=fl)
    Nb = case["Nb"]
    k = np.arange(Nb) // 100
    case["poro"] = np.ascontiguousarray(np.where(k < 5, 0.1, 0.2))
    pvars = np.zeros((Nb, 3))
    pvars[:, 0] = 0.2                    # Sw
    pvars[:, 1] = (k + 1.0) * bar        # p_o
    pvars[:, 2] = 0.0                    # Rs
    case["pv"] = np.ascontiguousarray(pvars.reshape(-1))
    case["meaning"] = np.full(Nb, pkg.decks.SW_PO_RS, np.uint8)
    fipnum = np.concatenate([np.full(400, 1), np.full(200, 2), np.full(400, 3)])
    return case, fipnum


def summary_deck_expectations():
    """the numbers tests/test_ecl_output.cc:192-225 expects of that deck at the first report step (METRIC: bar, sm^3) with its tolerances
    in per cent (BOOST_CHECK_CLOSE): field and region pressures weighted by hydrocarbon pore volume, fluids in place = sum b S pv"""
    return {"FPR": (((3 * 0.1 + 8 * 0.2) * 500 * (1 - 0.2)) / ((500 * 0.1 + 500 * 0.2) * (1 - 0.2)), 1e-3),
            "FOIP": ((0.3 * 0.1 + 0.8 * 0.2) * 500 * (1 - 0.2), 1e-1), "FGIP": (0.0, 1e-1), "FWIP": (1.0 / 1000 * (0.1 + 0.2) * 500 * 0.2, 1e-1),
            "RPR:1": ((2.5 * 0.1 * 400 * (1 - 0.2)) / (400 * 0.1 * (1 - 0.2)), 1e-3), "ROIP:1": (0.25 * 0.1 * 400 * (1 - 0.2), 1e-1),
            "RPR:2": (((5 * 0.1 * 100 + 6 * 0.2 * 100) * (1 - 0.2)) / ((100 * 0.1 + 100 * 0.2) * (1 - 0.2)), 1e-3),
            "ROIP:2": ((0.5 * 0.1 * 100 + 0.6 * 0.2 * 100) * (1 - 0.2), 1e-1)}


def summary_from_iq(iq, volume, fipnum):
    """FPR / F?IP / RPR / ROIP from intensive-quantity records (fields 0-2 saturations, 3-5 phase pressures, 6-8 inverse formation volume
    factors - water, oil, gas -, 15 Rs, 16 porosity incl. rock compressibility), as ebos/ecloutputblackoilmodule.hh forms them:
    in place = sum b S pv; pressures weighted by the hydrocarbon pore volume pv (1 - Sw).  METRIC units out (bar)."""
    import numpy as np
    sw, so, sg = iq[:, 0, 0], iq[:, 1, 0], iq[:, 2, 0]
    po = iq[:, 4, 0]
    bw, bo, bg = iq[:, 6, 0], iq[:, 7, 0], iq[:, 8, 0]
    pv = iq[:, 16, 0] * volume
    hcpv = pv * (1.0 - sw)
    out = {"FPR": float((po * hcpv).sum() / hcpv.sum()) / 1e5, "FOIP": float((bo * so * pv).sum()), "FWIP": float((bw * sw * pv).sum()),
           "FGIP": float((bg * sg * pv).sum() + (iq[:, 15, 0] * bo * so * pv).sum())}
    for r in (1, 2):
        m = fipnum == r
        out["RPR:%d" % r] = float((po[m] * hcpv[m]).sum() / hcpv[m].sum()) / 1e5
        out["ROIP:%d" % r] = float((bo[m] * so[m] * pv[m]).sum())
    return out
