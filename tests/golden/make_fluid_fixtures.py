#!/usr/bin/env python3
"""Generates the fluid/rock DATA fixtures from data files the reference tree holds.  Run in the build container
(needs /root/reference).  Output is data only (tables converted to SI, and the expected numbers of the reference's
own PVT test), never program text:

  opm-autodiff_amd/data/spe1_fluid.json  <- python/test_data/SPE1CASE1/SPE1CASE1.DATA (GRID :64-108, PROPS :109-250,
                                            SOLUTION/EQUIL/RSVD :252-290, SCHEDULE :370-440: DRSDT, WELSPECS, COMPDAT,
                                            WCONPROD, WCONINJE, TSTEP), FIELD units -> SI
  tests/golden/norne_pvt.json            <- tests/norne_pvt.data (PVTO, DENSITY; METRIC -> SI) and the (Rs, p) ->
                                            (mu_o, 1/B_o) expectations of tests/test_norne_pvt.cpp:64-294

Unit factors are opm-common's (UnitSystem FIELD / METRIC; Units.hpp), restated here.
"""
import json
import os
import re

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# ---- units (SI per deck unit) ---------------------------------------------------------------------------------
POUND, INCH, FEET, GRAV = 0.45359237, 0.0254, 0.3048, 9.80665
PSIA = POUND * GRAV / INCH ** 2             # 6894.757293168361 Pa
STB = 0.158987294928                        # m^3
SCF = FEET ** 3
FIELD = dict(pressure=PSIA, length=FEET, perm=9.869232667160130e-16, viscosity=1e-3,
             density=POUND / FEET ** 3, rs=1000.0 * SCF / STB, gas_fvf=STB / (1000.0 * SCF), oil_fvf=1.0,
             water_fvf=1.0, compressibility=1.0 / PSIA)
METRIC = dict(pressure=1e5, length=1.0, perm=9.869232667160130e-16, viscosity=1e-3, density=1.0, rs=1.0,
              gas_fvf=1.0, oil_fvf=1.0, water_fvf=1.0, compressibility=1e-5)


def tokenize(path):
    """-> {KEYWORD: [records]} with a record = list of float tokens; N*v expanded; comments stripped."""
    kws = {}
    cur = None
    rec = []
    with open(path, errors="replace") as f:
        for line in f:
            line = line.split("--")[0].strip()
            if not line:
                continue
            m = re.fullmatch(r"[A-Z][A-Z0-9]{0,7}", line)
            if m and not rec:
                cur = line
                kws.setdefault(cur, [])
                continue
            if cur is None:
                continue
            while line:
                if "/" in line:
                    head, line = line.split("/", 1)
                    term = True
                else:
                    head, line, term = line, "", False
                for tok in head.split():
                    tok = tok.strip("'")
                    mm = re.fullmatch(r"(\d+)\*(.+)", tok)
                    md = re.fullmatch(r"(\d+)\*", tok)
                    try:
                        if md:      # N defaulted items
                            rec.extend([None] * int(md.group(1)))
                        elif mm:
                            rec.extend([float(mm.group(2))] * int(mm.group(1)))
                        else:
                            rec.append(float(tok))
                    except ValueError:
                        rec.append(tok)
                if term:
                    kws[cur].append(rec)
                    rec = []
                    line = ""  # anything after the slash on that line is a comment
    return kws


def table(rec, ncol):
    assert len(rec) % ncol == 0, (len(rec), ncol)
    return [rec[i:i + ncol] for i in range(0, len(rec), ncol)]


def parse_pvto(records, U):
    """records of one PVTO keyword -> list of regions; region = list of dict(rs, p[], bo[], mu[])."""
    regions, cur = [], []
    for r in records:
        if not r:
            regions.append(cur)
            cur = []
            continue
        rs, rest = r[0], r[1:]
        rows = table(rest, 3)
        cur.append(dict(rs=rs * U["rs"], p=[x[0] * U["pressure"] for x in rows], bo=[x[1] * U["oil_fvf"] for x in rows],
                        mu=[x[2] * U["viscosity"] for x in rows]))
    if cur:
        regions.append(cur)
    return regions


def parse_pvtg(records, U):
    """records of one PVTG keyword -> list of regions; region = list of dict(pg, rv[], bg[], mu[]) with the rows of a node
    in deck order (saturated first, Rv descending)."""
    regions, cur = [], []
    for r in records:
        if not r:
            regions.append(cur)
            cur = []
            continue
        pg, rest = r[0], r[1:]
        rows = table(rest, 3)
        cur.append(dict(pg=pg * U["pressure"], rv=[x[0] / U["rs"] for x in rows], bg=[x[1] * U["gas_fvf"] for x in rows],
                        mu=[x[2] * U["viscosity"] for x in rows]))
    if cur:
        regions.append(cur)
    return regions


def spe1():
    k = tokenize(os.path.join(REF, "python/test_data/SPE1CASE1/SPE1CASE1.DATA"))
    U = FIELD
    pvtw = k["PVTW"][0]
    rock = k["ROCK"][0]
    dens = k["DENSITY"][0]
    out = dict(
        source="python/test_data/SPE1CASE1/SPE1CASE1.DATA (FIELD units converted to SI)",
        pvtw=dict(p_ref=pvtw[0] * U["pressure"], bw_ref=pvtw[1], cw=pvtw[2] * U["compressibility"],
                  mu_ref=pvtw[3] * U["viscosity"], cv=pvtw[4] * U["compressibility"]),
        rock=dict(p_ref=rock[0] * U["pressure"], cr=rock[1] * U["compressibility"]),
        density=dict(oil=dens[0] * U["density"], water=dens[1] * U["density"], gas=dens[2] * U["density"]),
        swof=[[r[0], r[1], r[2], r[3] * U["pressure"]] for r in table(k["SWOF"][0], 4)],
        sgof=[[r[0], r[1], r[2], r[3] * U["pressure"]] for r in table(k["SGOF"][0], 4)],
        pvdg=[[r[0] * U["pressure"], r[1] * U["gas_fvf"], r[2] * U["viscosity"]] for r in table(k["PVDG"][0], 3)],
        pvto=parse_pvto(k["PVTO"], U)[0],
    )
    # grid of the deck: 10x10x3, layer-wise constants
    dz = k["DZ"][0]
    permx = k["PERMX"][0]
    out["grid"] = dict(nx=10, ny=10, nz=3, dx=k["DX"][0][0] * U["length"], dy=k["DY"][0][0] * U["length"],
                       dz=[dz[0] * U["length"], dz[100] * U["length"], dz[200] * U["length"]],
                       tops=k["TOPS"][0][0] * U["length"], poro=k["PORO"][0][0],
                       perm=[permx[0] * U["perm"], permx[100] * U["perm"], permx[200] * U["perm"]])
    eq = k["EQUIL"][0]
    out["equil"] = dict(datum_depth=eq[0] * U["length"], datum_pressure=eq[1] * U["pressure"], woc=eq[2] * U["length"],
                        goc=eq[4] * U["length"])
    out["rsvd"] = [[r[0] * U["length"], r[1] * U["rs"]] for r in table(k["RSVD"][0], 2)]
    # SCHEDULE (:370-...): DRSDT, the two wells with their completions and controls, the report steps.  FIELD: liquid rates stb/day,
    # gas rates Mscf/day, well diameter ft, TSTEP days; cell indices 1-based in the deck, 0-based here
    DAY = 86400.0
    ws = {r[0]: dict(name=r[0], i=int(r[2]) - 1, j=int(r[3]) - 1, ref_depth=r[4] * U["length"], preferred_phase=r[5].lower()) for r in k["WELSPECS"] if r}
    for r in k["COMPDAT"]:
        if r:
            ws[r[0]].update(k_upper=int(r[3]) - 1, k_lower=int(r[4]) - 1, diameter=r[8] * U["length"])
    for r in k["WCONPROD"]:
        if r:   # name, OPEN, ORAT, oil rate, 4 defaulted items, BHP limit
            assert r[2] == "ORAT", r
            ws[r[0]].update(kind="producer", control=r[2].lower(), oil_rate=r[3] * STB / DAY, bhp_limit=r[8] * U["pressure"])
    for r in k["WCONINJE"]:
        if r:   # name, GAS, OPEN, RATE, surface rate, defaulted, BHP limit
            assert r[1] == "GAS" and r[3] == "RATE", r
            ws[r[0]].update(kind="injector", injected=r[1].lower(), control=r[3].lower(), surface_rate=r[4] * 1000.0 * SCF / DAY, bhp_limit=r[6] * U["pressure"])
    # DRSDT item 2 (ALL | FREE: which cells the limit binds) is defaulted in the deck: ALL (the keyword's default in ECLIPSE and opm-common -
    # neither is in the tree; the deck's own comment says what it means: "GOR cannot rise and free gas does not dissolve in undersaturated oil")
    drsdt_rec = k["DRSDT"][0]
    out["schedule"] = dict(drsdt=drsdt_rec[0] * U["rs"] / DAY, drsdt_option=(drsdt_rec[1] if len(drsdt_rec) > 1 and drsdt_rec[1] else "ALL"), wells=[ws[n] for n in sorted(ws)], tstep=[d * DAY for d in k["TSTEP"][0]])
    out["units"] = {kk: vv for kk, vv in U.items()}
    path = os.path.join(ROOT, "opm-autodiff_amd", "data", "spe1_fluid.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


def cpp_array(txt, name, start):
    m = re.compile(r"std::vector<double>\s+" + name + r"\s*=\s*\{(.*?)\};", re.S).search(txt, start)
    vals = [float(t) for t in re.findall(r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?", m.group(1))]
    return vals, m.end()


def norne():
    k = tokenize(os.path.join(REF, "tests/norne_pvt.data"))
    U = METRIC
    regions = parse_pvto(k["PVTO"], U)
    assert len(regions) == 2, len(regions)
    with open(os.path.join(REF, "tests/test_norne_pvt.cpp")) as f:
        txt = f.read()
    exp = []
    pos = 0
    for region in (0, 1):
        pos = txt.index("verify_norne_oil_pvt_region%d" % (region + 1), pos)
        rs, p1 = cpp_array(txt, "rs", pos)
        P, p2 = cpp_array(txt, "P", p1)
        mu, p3 = cpp_array(txt, "mu_expected", p2)
        b, p4 = cpp_array(txt, "b_expected", p3)
        pos = p4
        assert len(P) == len(mu) == len(b) and len(rs) >= len(P), (len(rs), len(P), len(mu), len(b))
        exp.append(dict(region=region, rs=[x * U["rs"] for x in rs], p=[x * U["pressure"] for x in P], mu_expected=mu,
                        b_expected=b))
    out = dict(source="tests/norne_pvt.data (METRIC -> SI); expectations tests/test_norne_pvt.cpp:64-294",
               semantics="for i in range(len(p)): RsSat = saturatedGasDissolutionFactor(region, p[i]); if rs[i] >= RsSat "
                         "use the saturated viscosity / inverse FVF at p[i], else the undersaturated ones at (p[i], rs[i]); "
                         "BOOST_CHECK_CLOSE tolerance 1e-5 percent (tests/test_norne_pvt.cpp:118-133)",
               check_close_percent=1e-5,
               density=[dict(oil=r[0], water=r[1], gas=r[2]) for r in k["DENSITY"][:2]],
               pvto=regions, expected=exp,
               # wet gas tables of the same file (used as realistic PVTG input of the wet-gas parity tests; the reference
               # holds no expectations for them): Rv in Sm3/Sm3, Bg in rm3/Sm3
               pvtg=parse_pvtg(k["PVTG"], U))
    path = os.path.join(ROOT, "tests", "golden", "norne_pvt.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, [len(r) for r in regions], [len(e["p"]) for e in exp])


if __name__ == "__main__":
    spe1()
    norne()
