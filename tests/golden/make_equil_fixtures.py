#!/usr/bin/env python3
"""Generates tests/golden/equil.json: the inputs of tests/equil_liveoil.DATA (the reference's one equilibration deck
with live oil + dry gas + water, METRIC -> SI) and the numbers tests/test_equil.cc:656-732 (DeckWithLiveOil) expects
of Opm::EQUIL::DeckDependent::InitialStateComputer for it.  Run in the build container (needs /root/reference).
Data only - tables, the EQUIL record, the grid column, expected values and the test's tolerances (BOOST_CHECK_CLOSE
takes its tolerance in PERCENT)."""
import json
import os
import re

from make_fluid_fixtures import METRIC, REF, ROOT, parse_pvtg, parse_pvto, table, tokenize


def expected_liveoil():
    with open(os.path.join(REF, "tests/test_equil.cc")) as f:
        txt = f.read()
    a = txt.index("BOOST_AUTO_TEST_CASE(DeckWithLiveOil)")
    b = txt.index("BOOST_AUTO_TEST_CASE(DeckWithLiveGas)")
    body = txt[a:b]
    num = r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?"

    def vec(pattern):
        m = re.search(pattern + r"\s*=?\s*\{(.*?)\};", body, re.S)
        return [float(t) for t in re.findall(num, m.group(1))]

    def close(name, which):
        m = re.search(r"BOOST_CHECK_CLOSE\(pressures\[FluidSystem::" + name + r"PhaseIdx\]\[" + which + r"\s*\],\s*(" + num + r"),\s*reltol\);", body)
        return float(m.group(1))
    return dict(
        source="tests/test_equil.cc:656-732 (DeckWithLiveOil), the values tagged 'opm'",
        reltol_percent=1.0e-4,
        pw_first=close("water", "first"), pw_last=close("water", "last"), po_first=close("oil", "first"), po_last=close("oil", "last"),
        sw=vec(r"s_opm\[FluidSystem::waterPhaseIdx\]"), so=vec(r"s_opm\[FluidSystem::oilPhaseIdx\]"),
        sg=vec(r"s_opm\[FluidSystem::gasPhaseIdx\]"), rs=vec(r"const std::vector<double> rs_opm"))


def tokenize_sections(path):
    """tokenize() of a deck whose section keywords carry decoration ("GRID      ======")"""
    import tempfile
    with open(path, errors="replace") as f:
        lines = [re.sub(r"^([A-Z]+)\s+=+\s*$", r"\1", ln.rstrip("\n")) for ln in f]
    with tempfile.NamedTemporaryFile("w", suffix=".DATA", delete=False) as t:
        t.write("\n".join(lines) + "\n")
    try:
        return tokenize(t.name)
    finally:
        os.unlink(t.name)


def liveoil():
    k = tokenize_sections(os.path.join(REF, "tests/equil_liveoil.DATA"))
    U = METRIC
    pvtw, rock, dens, eq = k["PVTW"][0], k["ROCK"][0], k["DENSITY"][0], k["EQUIL"][0]
    dz = [v * U["length"] for v in k["DZV"][0]]
    return dict(
        source="tests/equil_liveoil.DATA (METRIC units converted to SI); EQUIL item 7 defaulted: Rs = RsSat at the contact; item 9 = 0: cell centres",
        gravity=9.80665,
        pvtw=dict(p_ref=pvtw[0] * U["pressure"], bw_ref=pvtw[1], cw=pvtw[2] * U["compressibility"], mu_ref=pvtw[3] * U["viscosity"],
                  cv=pvtw[4] * U["compressibility"]),
        rock=dict(p_ref=rock[0] * U["pressure"], cr=rock[1] * U["compressibility"]),
        density=dict(oil=dens[0] * U["density"], water=dens[1] * U["density"], gas=dens[2] * U["density"]),
        swof=[[r[0], r[1], r[2], r[3] * U["pressure"]] for r in table(k["SWOF"][0], 4)],
        sgof=[[r[0], r[1], r[2], r[3] * U["pressure"]] for r in table(k["SGOF"][0], 4)],
        pvdg=[[r[0] * U["pressure"], r[1] * U["gas_fvf"], r[2] * U["viscosity"]] for r in table(k["PVDG"][0], 3)],
        pvto=parse_pvto(k["PVTO"], U)[0],
        grid=dict(nz=len(dz), dz=dz, tops=k["TOPS"][0][0] * U["length"]),
        equil=dict(datum=eq[0] * U["length"], pressure=eq[1] * U["pressure"], zwoc=eq[2] * U["length"], pcow_woc=eq[3] * U["pressure"],
                   zgoc=eq[4] * U["length"], pcgo_goc=eq[5] * U["pressure"], accuracy=int(eq[8])),
        expected=expected_liveoil())


def dead_oil_case(deck, test_name, next_test, gravity, sat_names, lines):
    """A PVDO (dead oil) deck of the same family; the oil formation-volume factor travels as a table of its own"""
    k = tokenize_sections(os.path.join(REF, "tests", deck))
    U = METRIC
    pvtw, dens, eq = k["PVTW"][0], k["DENSITY"][0], k["EQUIL"][0]
    dz = [v * U["length"] for v in k["DZV"][0]]
    with open(os.path.join(REF, "tests/test_equil.cc")) as f:
        txt = f.read()
    body = txt[txt.index("BOOST_AUTO_TEST_CASE(%s)" % test_name):txt.index("BOOST_AUTO_TEST_CASE(%s)" % next_test)]
    num = r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?"
    vec = lambda name: [float(t) for t in re.findall(num, re.search(name + r"\s*=\s*\{(.*?)\};", body, re.S).group(1))]
    pr = {}
    for name, which, val in re.findall(r"BOOST_CHECK_CLOSE\(pressures\[FluidSystem::(\w+)PhaseIdx\]\s*\[(\w+)\s*\]\s*,\s*(" + num + r")\s*,\s*reltol\);", body):
        pr["p%s_%s" % (name[0], which)] = float(val)
    exp = dict(source="tests/test_equil.cc:%s (%s)" % (lines, test_name), reltol_percent=1.0e-4, **pr)
    for key, nm in zip(("sw", "so", "sg"), ("water", "oil", "gas")):
        exp[key] = vec(sat_names + r"\[FluidSystem::" + nm + r"PhaseIdx\]")
    rock = k.get("ROCK", [[1.0, 0.0]])[0]
    return dict(
        source="tests/%s (METRIC units converted to SI); dead oil (PVDO), no dissolved gas; EQUIL item 9 = 0" % deck,
        gravity=gravity,
        pvtw=dict(p_ref=pvtw[0] * U["pressure"], bw_ref=pvtw[1], cw=pvtw[2] * U["compressibility"], mu_ref=pvtw[3] * U["viscosity"],
                  cv=pvtw[4] * U["compressibility"]),
        rock=dict(p_ref=rock[0] * U["pressure"], cr=rock[1] * U["compressibility"]),
        density=dict(oil=dens[0] * U["density"], water=dens[1] * U["density"], gas=dens[2] * U["density"]),
        swof=[[r[0], r[1], r[2], r[3] * U["pressure"]] for r in table(k["SWOF"][0], 4)],
        sgof=[[r[0], r[1], r[2], r[3] * U["pressure"]] for r in table(k["SGOF"][0], 4)],
        pvdg=[[r[0] * U["pressure"], r[1] * U["gas_fvf"], r[2] * U["viscosity"]] for r in table(k["PVDG"][0], 3)],
        pvdo=[[r[0] * U["pressure"], r[1] * U["oil_fvf"], r[2] * U["viscosity"]] for r in table(k["PVDO"][0], 3)],
        grid=dict(nz=len(dz), dz=dz, tops=(k["TOPS"][0][0] if "TOPS" in k else k["DEPTHZ"][0][0]) * U["length"]),
        equil=dict(datum=eq[0] * U["length"], pressure=eq[1] * U["pressure"], zwoc=eq[2] * U["length"], pcow_woc=eq[3] * U["pressure"],
                   zgoc=eq[4] * U["length"], pcgo_goc=eq[5] * U["pressure"], accuracy=int(eq[8])),
        expected=exp)


def wet_gas_case(deck, test_name, next_test, lines):
    """equil_livegas / equil_rsvd_and_rvvd / equil_pbvd_and_pdvd: wet gas (PVTG, VAPOIL) over dead (PVDO) or live (PVTO) oil,
    with the optional depth tables RSVD / RVVD / PBVD / PDVD; expectations: the values tagged 'opm' and the test's own
    tolerances (reltol in percent; saturations of the live-gas deck are checked at 100 x reltol)"""
    k = tokenize_sections(os.path.join(REF, "tests", deck))
    U = METRIC
    pvtw, dens, eq = k["PVTW"][0], k["DENSITY"][0], k["EQUIL"][0]
    dz = [v * U["length"] for v in (k["DZV"][0] if "DZV" in k else k["DZ"][0])]
    with open(os.path.join(REF, "tests/test_equil.cc")) as f:
        txt = f.read()
    a = txt.index("BOOST_AUTO_TEST_CASE(%s)" % test_name)
    b = txt.index("BOOST_AUTO_TEST_CASE(%s)" % next_test) if next_test else len(txt)
    body = txt[a:b]
    num = r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?"
    vec = lambda name: [float(t) for t in re.findall(num, re.search(name + r"\s*=?\s*\{(.*?)\};", body, re.S).group(1))]
    reltol = float(re.search(r"const double reltol = (" + num + r");", body).group(1))
    pr = {}
    for name, which, val in re.findall(r"BOOST_CHECK_CLOSE\(pressures\[FluidSystem::(\w+)PhaseIdx\]\s*\[(\w+)\s*\]\s*,\s*(" + num + r")\s*,\s*reltol\);", body):
        pr["p%s_%s" % (name[0], which)] = float(val)
    exp = dict(source="tests/test_equil.cc:%s (%s), the values tagged 'opm'" % (lines, test_name), reltol_percent=reltol,
               sat_reltol_percent=(100.0 * reltol if "100.*reltol" in body else reltol), **pr)
    for key, nm in zip(("sw", "so", "sg"), ("water", "oil", "gas")):
        exp[key] = vec(r"s_opm\[FluidSystem::" + nm + r"PhaseIdx\]")
    if re.search(r"std::vector<double> rs_opm", body):
        exp["rs"] = vec(r"const std::vector<double> rs_opm")
    exp["rv"] = vec(r"const std::vector<double> rv_opm")
    rock = k.get("ROCK", [[1.0, 0.0]])[0]
    out = dict(
        source="tests/%s (METRIC units converted to SI); EQUIL item 9 = 0: cell centres" % deck,
        gravity=9.80665,
        pvtw=dict(p_ref=pvtw[0] * U["pressure"], bw_ref=pvtw[1], cw=pvtw[2] * U["compressibility"], mu_ref=pvtw[3] * U["viscosity"],
                  cv=pvtw[4] * U["compressibility"]),
        rock=dict(p_ref=rock[0] * U["pressure"], cr=rock[1] * U["compressibility"]),
        density=dict(oil=dens[0] * U["density"], water=dens[1] * U["density"], gas=dens[2] * U["density"]),
        swof=[[r[0], r[1], r[2], r[3] * U["pressure"]] for r in table(k["SWOF"][0], 4)],
        sgof=[[r[0], r[1], r[2], r[3] * U["pressure"]] for r in table(k["SGOF"][0], 4)],
        pvtg=parse_pvtg(k["PVTG"], U)[0],
        grid=dict(nz=len(dz), dz=dz, tops=k["TOPS"][0][0] * U["length"]),
        equil=dict(datum=eq[0] * U["length"], pressure=eq[1] * U["pressure"], zwoc=eq[2] * U["length"], pcow_woc=eq[3] * U["pressure"],
                   zgoc=eq[4] * U["length"], pcgo_goc=eq[5] * U["pressure"], accuracy=int(eq[8])),
        expected=exp)
    if "PVTO" in k:
        out["pvto"] = parse_pvto(k["PVTO"], U)[0]
    else:
        out["pvdo"] = [[r[0] * U["pressure"], r[1] * U["oil_fvf"], r[2] * U["viscosity"]] for r in table(k["PVDO"][0], 3)]
    for kw, key, scale in (("RSVD", "rsvd", U["rs"]), ("RVVD", "rvvd", 1.0 / U["rs"]), ("PBVD", "pbvd", U["pressure"]), ("PDVD", "pdvd", U["pressure"])):
        if kw in k:
            out[key] = [[r[0] * U["length"], r[1] * scale] for r in table(k[kw][0], 2)]
    return out


def all_dead_case():
    """equil_deadfluids.DATA / DeckAllDead (tests/test_equil.cc:477-502): three pressures, tolerance 0.1 %; no saturation vectors"""
    d = dead_oil_case_tables("equil_deadfluids.DATA", 10.0)
    with open(os.path.join(REF, "tests/test_equil.cc")) as f:
        txt = f.read()
    body = txt[txt.index("BOOST_AUTO_TEST_CASE(DeckAllDead)"):txt.index("BOOST_AUTO_TEST_CASE(CapillaryInversion)")]
    num = r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?"
    pr = {}
    for name, which, val in re.findall(r"BOOST_CHECK_CLOSE\(pressures\[FluidSystem::(\w+)PhaseIdx\]\s*\[(\w+)\s*\]\s*,\s*(" + num + r")\s*,\s*reltol\);", body):
        pr["p%s_%s" % (name[0], which)] = float(val)
    d["expected"] = dict(source="tests/test_equil.cc:477-502 (DeckAllDead)", reltol_percent=float(re.search(r"const double reltol = (" + num + r");", body).group(1)), **pr)
    return d


def dead_oil_case_tables(deck, gravity):
    """the property tables, grid and EQUIL record of a PVDO deck (no expectations)"""
    k = tokenize_sections(os.path.join(REF, "tests", deck))
    U = METRIC
    pvtw, dens, eq = k["PVTW"][0], k["DENSITY"][0], k["EQUIL"][0]
    dz = [v * U["length"] for v in k["DZV"][0]]
    rock = k.get("ROCK", [[1.0, 0.0]])[0]
    eq = list(eq) + [0] * (9 - len(eq))
    return dict(
        source="tests/%s (METRIC units converted to SI); dead oil (PVDO), no dissolved gas; EQUIL item 9 = 0" % deck,
        gravity=gravity,
        pvtw=dict(p_ref=pvtw[0] * U["pressure"], bw_ref=pvtw[1], cw=pvtw[2] * U["compressibility"], mu_ref=pvtw[3] * U["viscosity"],
                  cv=pvtw[4] * U["compressibility"]),
        rock=dict(p_ref=rock[0] * U["pressure"], cr=rock[1] * U["compressibility"]),
        density=dict(oil=dens[0] * U["density"], water=dens[1] * U["density"], gas=dens[2] * U["density"]),
        swof=[[r[0], r[1], r[2], r[3] * U["pressure"]] for r in table(k["SWOF"][0], 4)],
        sgof=[[r[0], r[1], r[2], r[3] * U["pressure"]] for r in table(k["SGOF"][0], 4)],
        pvdg=[[r[0] * U["pressure"], r[1] * U["gas_fvf"], r[2] * U["viscosity"]] for r in table(k["PVDG"][0], 3)],
        pvdo=[[r[0] * U["pressure"], r[1] * U["oil_fvf"], r[2] * U["viscosity"]] for r in table(k["PVDO"][0], 3)],
        grid=dict(nz=len(dz), dz=dz, tops=(k["TOPS"][0][0] if "TOPS" in k else k["DEPTHZ"][0][0]) * U["length"]),
        equil=dict(datum=eq[0] * U["length"], pressure=eq[1] * U["pressure"], zwoc=eq[2] * U["length"], pcow_woc=eq[3] * U["pressure"],
                   zgoc=eq[4] * U["length"], pcgo_goc=eq[5] * U["pressure"], accuracy=int(eq[8] or 0)))


def swatinit_case():
    """equil_capillary_swatinit.DATA / DeckWithSwatinit (tests/test_equil.cc:1006-1146; the case is compiled out there with
    `#if 0`, its numbers are still the reference's own): saturations without and with SWATINIT applied, and the oil-water
    capillary pressure the rescaled curves must give in the twelve cells above the contact; tolerance 0.1 % (reltol 1.0e-1)"""
    deck = "equil_capillary_swatinit.DATA"
    k = tokenize_sections(os.path.join(REF, "tests", deck))
    U = METRIC
    pvtw, dens, eq, rock = k["PVTW"][0], k["DENSITY"][0], k["EQUIL"][0], k["ROCK"][0]
    eq = [0.0 if v is None else v for v in list(eq) + [0] * (9 - len(eq))]
    dz = [v * U["length"] for v in k["DZ"][0]]
    with open(os.path.join(REF, "tests/test_equil.cc")) as f:
        txt = f.read()
    body = txt[txt.index("BOOST_AUTO_TEST_CASE(DeckWithSwatinit)"):]
    num = r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?"
    rows = lambda name: [[float(t) for t in re.findall(num, row)]
                         for row in re.findall(r"\{([^{}]*)\}", re.search(r"const std::vector<double> " + name + r"\[3\]\{(.*?)\};", body, re.S).group(1))]
    s_plain, s_swat = rows("s"), rows("swatinit")
    pcs = [float(v) for _, v in sorted((int(i), v) for i, v in re.findall(r"pc_scaled_truth\[3\*(\d+) \+ 0\] =\s*(" + num + r");", body))]
    assert len(s_plain) == 3 and len(s_swat) == 3 and len(pcs) == 12
    return dict(
        source="tests/%s (METRIC units converted to SI); dead oil (PVDO); EQUIL item 9 = 0" % deck,
        gravity=9.81,
        pvtw=dict(p_ref=pvtw[0] * U["pressure"], bw_ref=pvtw[1], cw=pvtw[2] * U["compressibility"], mu_ref=pvtw[3] * U["viscosity"],
                  cv=pvtw[4] * U["compressibility"]),
        rock=dict(p_ref=rock[0] * U["pressure"], cr=rock[1] * U["compressibility"]),
        density=dict(oil=dens[0] * U["density"], water=dens[1] * U["density"], gas=dens[2] * U["density"]),
        swof=[[r[0], r[1], r[2], r[3] * U["pressure"]] for r in table(k["SWOF"][0], 4)],
        sgof=[[r[0], r[1], r[2], r[3] * U["pressure"]] for r in table(k["SGOF"][0], 4)],
        pvdg=[[r[0] * U["pressure"], r[1] * U["gas_fvf"], r[2] * U["viscosity"]] for r in table(k["PVDG"][0], 3)],
        pvdo=[[r[0] * U["pressure"], r[1] * U["oil_fvf"], r[2] * U["viscosity"]] for r in table(k["PVDO"][0], 3)],
        grid=dict(nz=len(dz), dz=dz, tops=k["TOPS"][0][0] * U["length"]),
        swatinit=list(k["SWATINIT"][0]),
        equil=dict(datum=eq[0] * U["length"], pressure=eq[1] * U["pressure"], zwoc=eq[2] * U["length"], pcow_woc=eq[3] * U["pressure"],
                   zgoc=eq[4] * U["length"], pcgo_goc=eq[5] * U["pressure"], accuracy=int(eq[8] or 0)),
        expected=dict(source="tests/test_equil.cc:1006-1146 (DeckWithSwatinit, under #if 0 in the reference)", reltol_percent=1.0e-1,
                      phase_order="water, oil, gas",
                      without=dict(sw=s_plain[0], so=s_plain[1], sg=s_plain[2]),
                      with_swatinit=dict(sw=s_swat[0], so=s_swat[1], sg=s_swat[2]),
                      pcow_scaled=pcs))


def capillary_inversion():
    """CapillaryInversion (tests/test_equil.cc:504-554): (pc, saturation) vectors of satFromPc for oil-water and gas-oil and of
    satFromSumOfPcs, on the saturation tables of equil_capillary.DATA (fixture 'capillary')"""
    with open(os.path.join(REF, "tests/test_equil.cc")) as f:
        txt = f.read()
    body = txt[txt.index("BOOST_AUTO_TEST_CASE(CapillaryInversion)"):txt.index("BOOST_AUTO_TEST_CASE(DeckWithCapillary)")]
    num = r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?"
    pcs = [[float(t) for t in re.findall(num, m)] for m in re.findall(r"std::vector<double> pc = \{(.*?)\};", body, re.S)]
    ss = [[float(t) for t in re.findall(num, m)] for m in re.findall(r"std::vector<double> s = \{(.*?)\};", body, re.S)]
    assert len(pcs) == 3 and len(ss) == 3
    return dict(source="tests/test_equil.cc:504-554 (CapillaryInversion); tables: fixture 'capillary'", reltol_percent=1.0e-5,
                oil_water=dict(pc=pcs[0], s=ss[0], increasing=False), gas_oil=dict(pc=pcs[1], s=ss[1], increasing=True),
                gas_water=dict(pc=pcs[2], s=ss[2]))


def default_fluid_cases():
    """PhasePressure, CellSubset, RegMapping (tests/test_equil.cc:218-475): the test's own constant-density fluid
    (initDefaultFluidSystem :119-182: B = 1, 700 / 1000 / 1000 kg/m3, no mixing), 10 x 1 x 10 cells of 1 m (equil_base.DATA), g = 10"""
    rec = lambda datd, datp, zwoc, pcow, zgoc, pcgo: dict(datum=datd, pressure=datp, zwoc=zwoc, pcow_woc=pcow, zgoc=zgoc, pcgo_goc=pcgo, accuracy=0)
    two = [rec(0, 1e5, 2.5, -0.075e5, 0, 0), rec(5, 1.35e5, 7.5, -0.225e5, 5, 0)]
    return dict(source="tests/test_equil.cc:218-475; grid tests/equil_base.DATA (DIMENS 10 1 10, DX = DY = DZ = 1, TOPS 0)",
                gravity=10.0, density=dict(oil=700.0, water=1000.0, gas=1000.0), grid=dict(nx=10, ny=1, nz=10, d=1.0, tops=0.0),
                phase_pressure=dict(record=rec(0, 1e5, 5, 0, 0, 0), reltol_percent=1.0e-6,
                                    expected=dict(pw_first=90e3, pw_last=180e3, po_first=103.5e3, po_last=166.5e3)),
                regions=dict(records=[two[0], two[0], two[1], two[1]],
                             cell_subset="coarse blocks of 5 x 1 x 5 cells (:309-326)", reg_mapping="EQLNUM [0 1; 2 3] by layer and column (:425-444)",
                             reltol_percent=1.0e-6, expected=dict(pw_first=105e3, pw_last=195e3, po_first=103.5e3, po_last=166.5e3)))


if __name__ == "__main__":
    out = dict(liveoil=liveoil(),
               capillary=dead_oil_case("equil_capillary.DATA", "DeckWithCapillary", "DeckWithCapillaryOverlap", 10.0, r"\bs", "556-594"),
               capillary_overlap=dead_oil_case("equil_capillary_overlap.DATA", "DeckWithCapillaryOverlap", "DeckWithLiveOil", 9.80665,
                                               r"s_opm", "596-654"),
               livegas=wet_gas_case("equil_livegas.DATA", "DeckWithLiveGas", "DeckWithRSVDAndRVVD", "734-812"),
               rsvd_rvvd=wet_gas_case("equil_rsvd_and_rvvd.DATA", "DeckWithRSVDAndRVVD", "DeckWithPBVDAndPDVD", "814-912"),
               pbvd_pdvd=wet_gas_case("equil_pbvd_and_pdvd.DATA", "DeckWithPBVDAndPDVD", "DeckWithSwatinit", "914-1004"),
               alldead=all_dead_case(), swatinit=swatinit_case(), capillary_inversion=capillary_inversion(), default_fluid=default_fluid_cases())
    path = os.path.join(ROOT, "tests", "golden", "equil.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)
