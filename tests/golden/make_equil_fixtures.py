#!/usr/bin/env python3
"""Generates tests/golden/equil.json: the inputs of tests/equil_liveoil.DATA (the reference's one equilibration deck
with live oil + dry gas + water, METRIC -> SI) and the numbers tests/test_equil.cc:656-732 (DeckWithLiveOil) expects
of Opm::EQUIL::DeckDependent::InitialStateComputer for it.  Run in the build container (needs /root/reference).
Data only - tables, the EQUIL record, the grid column, expected values and the test's tolerances (BOOST_CHECK_CLOSE
takes its tolerance in PERCENT)."""
import json
import os
import re

from make_fluid_fixtures import METRIC, REF, ROOT, parse_pvto, table, tokenize


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


if __name__ == "__main__":
    out = dict(liveoil=liveoil())
    path = os.path.join(ROOT, "tests", "golden", "equil.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)
