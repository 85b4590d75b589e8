"""Synthetic black-oil cases in the form the hot path consumes (arrays, SI units): what Flow's deck-side setup
(EclProblem / EclTransmissibility / equilibration, all outside the hot path) would hand to the assembly.

Recipes follow SURVEY.md §8(d): Cartesian nx x ny x nz, DX = DY = 20 m, DZ = 5 m, tops at 2500 m, PORO 0.25,
PERM 100 mD (homogeneous) or log-normal exp(N(ln 100, 1)) with rng(12345); SPE1 fluid; hydrostatic initial state,
p = 250 bar at the top, Sw = 0.2 and either Sg = 0 with Rs = 0.8 RsSat(p) ("undersaturated") or Sg = 0.1
("saturated"), perturbed with rng(2024) by +-1 % on p and +-0.02 on the saturations.
"""
import numpy as np

from . import fluid as _fluid
from . import grid as _grid

MILLIDARCY = 9.869232667160130e-16
SW_PO_SG, SW_PO_RS = 0, 1  # PrimaryVariables::PrimaryVarsMeaning


def rs_sat(fl, p, region=0):
    """Saturated Rs(p) of the PVTO table (piecewise linear in the bubble-point pressures, linear extrapolation)."""
    nodes = fl.pvt[region]["pvto"]
    xp = np.array([n["p"][0] for n in nodes])
    fp = np.array([n["rs"] for n in nodes])
    p = np.asarray(p, float)
    s = np.clip(np.searchsorted(xp, p, side="right") - 1, 0, len(xp) - 2)
    return fp[s] + (fp[s + 1] - fp[s]) * (p - xp[s]) / (xp[s + 1] - xp[s])


def cartesian_cells(nx, ny, nz, dx=20.0, dy=20.0, dz=5.0, top=2500.0, poro=0.25, perm_md=100.0, heterogeneous=False,
                    state="undersaturated", perturb=True, fluid=None, seed_state=2024):
    """Per-cell arrays of the synthetic case (no connectivity): geometry, rock and the initial state."""
    fl = fluid if fluid is not None else _fluid.spe1_fluid()[0]
    Nb = nx * ny * nz
    if heterogeneous:
        perm = np.exp(np.random.default_rng(12345).normal(np.log(perm_md), 1.0, Nb)) * MILLIDARCY
    else:
        perm = np.full(Nb, perm_md * MILLIDARCY)
    k = np.arange(Nb) // (nx * ny)
    volume = np.full(Nb, dx * dy * dz)
    depth = top + (k + 0.5) * dz
    rng = np.random.default_rng(seed_state)
    p = 250e5 + 7000.0 * (depth - top)
    sw = np.full(Nb, 0.2)
    if perturb:
        p = p * (1.0 + rng.uniform(-0.01, 0.01, Nb))
        sw = sw + rng.uniform(-0.02, 0.02, Nb)
    pv = np.zeros((Nb, 3))
    pv[:, 0], pv[:, 1] = sw, p
    if state == "undersaturated":
        meaning = np.full(Nb, SW_PO_RS, np.uint8)
        pv[:, 2] = 0.8 * rs_sat(fl, p)
    elif state == "saturated":
        meaning = np.full(Nb, SW_PO_SG, np.uint8)
        pv[:, 2] = 0.1 + (rng.uniform(-0.02, 0.02, Nb) if perturb else 0.0)
    elif state == "mixed":  # both meanings in one grid: gas cap in the upper half
        meaning = np.where(depth < np.median(depth), SW_PO_SG, SW_PO_RS).astype(np.uint8)
        pv[:, 2] = np.where(meaning == SW_PO_SG, 0.1 + (rng.uniform(-0.02, 0.02, Nb) if perturb else 0.0),
                            0.8 * rs_sat(fl, p))
    else:
        raise ValueError(state)
    return dict(Nb=Nb, nx=nx, ny=ny, nz=nz, dx=dx, dy=dy, dz=dz, top=top, perm=perm, poro=np.full(Nb, float(poro)),
                volume=volume, depth=np.ascontiguousarray(depth), fluid=fl, pv=np.ascontiguousarray(pv.reshape(-1)), meaning=meaning)


def cartesian_case(nx, ny, nz, dx=20.0, dy=20.0, dz=5.0, top=2500.0, poro=0.25, perm_md=100.0, heterogeneous=False,
                   state="undersaturated", perturb=True, fluid=None, seed_state=2024):
    c = cartesian_cells(nx, ny, nz, dx, dy, dz, top, poro, perm_md, heterogeneous, state, perturb, fluid, seed_state)
    pat = _grid.cartesian_pattern(nx, ny, nz)
    _, _, area = _grid.cartesian_geometry(pat, dx, dy, dz, top)
    trans = _grid.tpfa_transmissibility(pat, c["perm"], c["perm"], c["perm"], dx, dy, dz)
    return dict(Nb=c["Nb"], nx=nx, ny=ny, nz=nz, rowptr=pat["rowptr"], col=pat["col"], face_dir=pat["face_dir"],
                trans=np.ascontiguousarray(trans), area=np.ascontiguousarray(area), poro=c["poro"],
                volume=np.ascontiguousarray(c["volume"]), depth=c["depth"], fluid=c["fluid"], pv=c["pv"], meaning=c["meaning"],
                perm=c["perm"], dx=dx, dy=dy, dz=dz)


def spe1_case(props=None, state="equil"):
    """The SPE1CASE1 deck (python/test_data/SPE1CASE1/SPE1CASE1.DATA; FIELD units converted to SI in data/spe1_fluid.json): its grid
    (10 x 10 x 3, layered DZ / PERM, :64-108), its PROPS (:109-250) and, state = "equil", its SOLUTION section (:252-290) - EQUIL 8400 ft /
    4800 psia, contacts outside the reservoir, RSVD 1.27 Mscf/stb - equilibrated by equil.equilibrate (initstateequil.hh's algorithm, held
    by the twelve decks of tests/test_equil.cc) on top of `props`: an object with probe(p, rs, sw, sg) - capi.HipFluid(fluid), the DEVICE's
    property functions (the default: a GPU is needed), or the oracle's in CPU tests.  The case carries the SCHEDULE section's DRSDT 0
    ("drsdt", "drsdt_all_cells": the Rs cap of eclproblem.hh:1711-1732 with the keyword's option ALL - every cell's Rs is held at its value
    of the last accepted step, :2053-2070) and, under "schedule", its wells and report steps (spe1_wells builds the well model).
    state = "rough" (no props needed): a constant oil gradient from the datum pressure, Sw = connate, Rs = the RSVD value - a plausible
    undersaturated state for tests that only need one."""
    fl, d = _fluid.spe1_fluid()
    g = d["grid"]
    nx, ny, nz = g["nx"], g["ny"], g["nz"]
    pat = _grid.cartesian_pattern(nx, ny, nz)
    Nb = pat["Nb"]
    k = np.arange(Nb) // (nx * ny)
    dzs = np.array(g["dz"])
    dz = dzs[k]
    ztop = g["tops"] + np.concatenate([[0.0], np.cumsum(dzs)[:-1]])[k]
    depth = ztop + 0.5 * dz
    volume = g["dx"] * g["dy"] * dz
    perm = np.array(g["perm"])[k]
    fd = np.abs(pat["face_dir"].astype(int))
    row = _grid.row_of_entries(pat["rowptr"])
    col = pat["col"]
    area = np.zeros(len(col))
    area[fd == 1] = (g["dy"] * dz[row])[fd == 1]
    area[fd == 2] = (g["dx"] * dz[row])[fd == 2]
    area[fd == 3] = g["dx"] * g["dy"]
    # half transmissibilities with per-cell DZ (ebos/ecltransmissibility.cc:928-944), harmonic combination (:352-356)
    T = np.zeros(len(col))
    for dnum, h_of, a_of in ((1, lambda c: np.full(len(c), g["dx"]), lambda c: g["dy"] * dz[c]),
                             (2, lambda c: np.full(len(c), g["dy"]), lambda c: g["dx"] * dz[c]),
                             (3, lambda c: dz[c], lambda c: np.full(len(c), g["dx"] * g["dy"]))):
        m = fd == dnum
        half = lambda c: perm[c] * a_of(c) * (h_of(c) / 2.0) / (h_of(c) / 2.0) ** 2
        t1, t2 = half(row[m]), half(col[m])
        T[m] = 1.0 / (1.0 / t1 + 1.0 / t2)
    e = d["equil"]
    pv = np.zeros((Nb, 3))
    meaning = np.full(Nb, SW_PO_RS, np.uint8)
    if state == "rough":
        rho_o = d["density"]["oil"] * 0.78  # rough reservoir oil gradient; the state only has to be plausible
        pv[:, 0] = d["swof"][0][0]
        pv[:, 1] = e["datum_pressure"] + rho_o * 9.80665 * (depth - e["datum_depth"])
        pv[:, 2] = d["rsvd"][0][1]
    else:
        from . import equil as _equil
        if props is None:
            from . import capi as _capi
            props = _capi.HipFluid(fl)           # the device's fluid and saturation functions (raises without a GPU: no CPU fallback)
        rec = dict(datum=e["datum_depth"], pressure=e["datum_pressure"], zwoc=e["woc"], pcow_woc=0.0, zgoc=e["goc"], pcgo_goc=0.0)
        rsvd = np.array(d["rsvd"])
        limits = dict(Swl=d["swof"][0][0], Swu=d["swof"][-1][0], Sgl=d["sgof"][0][0], Sgu=d["sgof"][-1][0])
        rho = (d["density"]["oil"], d["density"]["water"], d["density"]["gas"])
        # one equilibration per layer (cells of a layer share their centre depth), spread over the layer's cells
        zc = g["tops"] + np.concatenate([[0.0], np.cumsum(dzs)[:-1]]) + 0.5 * dzs
        r = _equil.equilibrate(props, rho, rec, zc, (g["tops"], g["tops"] + float(dzs.sum())), limits,
                               rs_func=_equil.RsVD(props, rsvd[:, 0], rsvd[:, 1]))
        sg = r["sg"][k]
        pv[:, 0] = r["sw"][k]
        pv[:, 1] = r["po"][k]
        pv[:, 2] = np.where(sg > 0.0, sg, r["rs"][k])
        meaning = np.where(sg > 0.0, SW_PO_SG, SW_PO_RS).astype(np.uint8)
    sched = d.get("schedule", {})
    return dict(Nb=Nb, nx=nx, ny=ny, nz=nz, rowptr=pat["rowptr"], col=col, face_dir=pat["face_dir"],
                trans=np.ascontiguousarray(T), area=np.ascontiguousarray(area), poro=np.full(Nb, g["poro"]),
                volume=np.ascontiguousarray(volume), depth=np.ascontiguousarray(depth), fluid=fl,
                pv=np.ascontiguousarray(pv.reshape(-1)), meaning=meaning, perm=perm, dx=g["dx"], dy=g["dy"], dz_cell=dz,
                drsdt=[sched["drsdt"]] if "drsdt" in sched else None,
                drsdt_all_cells=[1 if sched.get("drsdt_option", "ALL") == "ALL" else 0] if "drsdt" in sched else None, schedule=sched)


def spe1_wells(case):
    """The deck's two wells (SPE1CASE1.DATA:383-425) as a wells.StandardWells: INJ, gas at a surface rate of 100 MMscf/day into (1, 1, 1),
    BHP limit 9014 psia; PROD, 20 000 stb/day of oil out of (10, 10, 3), BHP limit 1000 psia; connection factors by Peaceman's formula from
    the cell's permeability and the wellbore diameter (COMPDAT item 9)"""
    from . import wells as _wells
    nx, ny = case["nx"], case["ny"]
    out = []
    for w in case["schedule"]["wells"]:
        cells = [w["i"] + nx * (w["j"] + ny * kk) for kk in range(w["k_upper"], w["k_lower"] + 1)]
        tw = [_wells.peaceman_factor(case["perm"][c], case["dx"], case["dy"], case["dz_cell"][c], w["diameter"]) for c in cells]
        if w["kind"] == "producer":
            ctl = ("rate", _wells.OIL, w["oil_rate"])
            out.append(_wells.Well(w["name"], cells, tw, w["ref_depth"], True, ctl, w["bhp_limit"]))
        else:
            comp = {"gas": _wells.GAS, "water": _wells.WATER, "oil": _wells.OIL}[w["injected"]]
            out.append(_wells.Well(w["name"], cells, tw, w["ref_depth"], False, ("rate", comp, w["surface_rate"]), w["bhp_limit"], inj_phase=w["injected"]))
    return _wells.StandardWells(out, case["depth"])


STB_PER_DAY = 0.158987294928 / 86400.0
PSIA = 6894.757293168361
# (i, j), one-based, of SPE9's 25 producers in the order of the public deck's COMPDAT (the deck is not in the reference tree: a stand-in
# that completes its wells where the comparative-solution project does)
SPE9_PRODUCERS_IJ = [(5, 1), (8, 2), (11, 3), (10, 4), (12, 5), (4, 6), (8, 7), (14, 8), (11, 9), (12, 10), (10, 11), (5, 12), (8, 13),
                     (11, 14), (13, 15), (15, 16), (11, 17), (12, 18), (5, 19), (8, 20), (11, 21), (15, 22), (12, 23), (10, 24), (17, 25)]


def spe9_shaped_wells(case, oil_rate_stb_day=1500.0, water_rate_stb_day=5000.0, producer_bhp_limit=1000.0 * PSIA, injector_bhp_limit=400e5,
                      diameter=0.1524):
    """BASELINE.json configs[2] with wells that are wells (wells.StandardWells) on a 24 x 25 x 15 cartesian_case: SPE9's water injector at
    (24, 25), completed in layers 11-15, on a surface-rate target with an upper BHP limit, and its 25 oil producers, completed in layers
    2-4, on an oil-rate target (1500 stb/day; the SPE9 schedule cuts it to 100 stb/day for a while) with a lower BHP limit of 1000 psia;
    reference depth = the centre of the topmost completion; connection factors by Peaceman's formula from the cell's own permeability,
    so that on the log-normal field some producers cannot hold their target and fall to the BHP limit.  A stand-in like the grid itself:
    the SPE9 deck is not in the reference tree, its fluid and its dipping layers are not reproduced."""
    from . import wells as _wells
    nx, ny, nz = case["nx"], case["ny"], case["nz"]
    if (nx, ny) != (24, 25) or nz < 15:
        raise ValueError("spe9_shaped_wells: a 24 x 25 x 15 grid is expected")
    def column(i, j, k0, k1):
        cells = [(i - 1) + nx * ((j - 1) + ny * k) for k in range(k0 - 1, k1)]
        tw = [_wells.peaceman_factor(case["perm"][c], case["dx"], case["dy"], case["dz"], diameter) for c in cells]
        return cells, tw
    cells, tw = column(24, 25, 11, 15)
    out = [_wells.Well("INJE1", cells, tw, case["depth"][cells[0]], False, ("rate", _wells.WATER, water_rate_stb_day * STB_PER_DAY), injector_bhp_limit,
                       inj_phase="water")]
    for n, (i, j) in enumerate(SPE9_PRODUCERS_IJ):
        cells, tw = column(i, j, 2, 4)
        out.append(_wells.Well("PRODU%d" % (n + 2), cells, tw, case["depth"][cells[0]], True, ("rate", _wells.OIL, oil_rate_stb_day * STB_PER_DAY), producer_bhp_limit))
    return _wells.StandardWells(out, case["depth"])


# Field rate of the five-spot pair on a 100 x 100 x 100 grid (scaled with the areal cell count elsewhere): 0.4 % of a well
# cell's pore volume per day.  Fixed-rate sources do not see the mobility of the cell they drain, so rates ten times
# higher desaturate the producer column within weeks and the Newton method then fails on 10-day steps.
BENCH_RATE_SM3_PER_DAY = 200.0


def five_spot_source(case, rate_sm3_per_day=50.0):
    """Fixed-rate source terms (surface m^3/s per cell, equations oil/water/gas): water injected in one corner
    column, the same surface volume of oil (+ its dissolved gas at the initial Rs) produced in the opposite one."""
    Nb, nx, ny, nz = case["Nb"], case["nx"], case["ny"], case["nz"]
    src = np.zeros((Nb, 3))
    q = rate_sm3_per_day / 86400.0 / nz
    for k in range(nz):
        inj = 0 + nx * (0 + ny * k)
        prod = (nx - 1) + nx * ((ny - 1) + ny * k)
        src[inj, 1] += q
        src[prod, 0] -= q
        rs = case["pv"][3 * prod + 2] if case["meaning"][prod] == SW_PO_RS else float(rs_sat(case["fluid"], case["pv"][3 * prod + 1]))
        src[prod, 2] -= q * rs
    return np.ascontiguousarray(src.reshape(-1))


def write_case_binary(case, path, source=None):
    """Case file for the C++ host driver (opm-autodiff_amd/host/test_BlackoilModelHip.cpp): named raw arrays."""
    import struct
    fl = case["fluid"]
    arrs = {k: case[k] for k in ("rowptr", "col", "trans", "area", "poro", "volume", "depth", "pv", "meaning")}
    arrs.update(fl.arrays)
    arrs["fluid_hdr"] = np.array([len(fl.pvt), len(fl.sat)], np.int32)
    arrs["rock"] = np.array([fl.rock_pref, fl.rock_cr], np.float64)
    if source is not None:
        arrs["source"] = np.asarray(source, np.float64)
    code = {np.dtype(np.int32): 0, np.dtype(np.float64): 1, np.dtype(np.uint8): 2}
    with open(path, "wb") as f:
        f.write(b"OPMHIPCASE1\0")
        for name, a in arrs.items():
            a = np.ascontiguousarray(a)
            if a.dtype == np.int64:
                a = a.astype(np.int32)
            nb = name.encode()
            f.write(struct.pack("<I", len(nb)) + nb + struct.pack("<BQ", code[a.dtype], a.size))
            f.write(a.tobytes())
