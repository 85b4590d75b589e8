"""A reference-held number for a piece of the ASSEMBLY half: the reference's tests/test_ecl_output.cc:150-225 runs
SUMMARY_DECK_NON_CONSTANT_POROSITY.DATA through ebos' intensive quantities and expects the field / region pressures and the fluids in
place of the first report step - sums of b S pv over the cells, i.e. of the factors of the storage term (BlackOilLocalResidual::
computeStorage): the inverse formation volume factors of undersaturated oil at Rs = 0 (PVTO: one saturated node, master-table extension
of the Rs = 1 branch - 1 / B_o linear in p), of water (PVTW) and gas, the saturations of the initial state, the pore volume with the
rock compressibility.  The oracle's intensive quantities reproduce every one of them within the reference's own tolerance.  What this
pins is narrow - values, not derivatives; PVT and porosity, not relative permeabilities or fluxes - and is listed as such in DESIGN.md
section 2."""
import numpy as np

import helpers
import oracle_bind


def check(got, tol_scale=1.0):
    for key, (want, tol_percent) in helpers.summary_deck_expectations().items():
        if want == 0.0:
            assert abs(got[key]) < 1e-12, key
        else:
            assert abs(got[key] - want) <= tol_scale * tol_percent / 100.0 * abs(want), (key, got[key], want)


def test_summary_deck_fluids_in_place_and_pressures(pkg, orc):
    case, fipnum = helpers.summary_deck_case(pkg)
    o = oracle_bind.OracleModel(orc, case)
    o.set_state(case["pv"], case["meaning"])
    iq = o.iq()
    got = helpers.summary_from_iq(iq, case["volume"], fipnum)
    check(got)
    # the reference's tolerances are loose (0.1 % on the in-place figures); what the oracle actually leaves: b_o follows 0.1 p to 1e-4
    # (the 10.001 of the PVTO record), the pore volume the rock compressibility to 4e-5
    bo = iq[:, 7, 0]
    p_bar = case["pv"].reshape(-1, 3)[:, 1] / 1e5
    assert np.abs(bo / (0.1 * p_bar) - 1.0).max() < 2e-4
    assert np.abs(iq[:, 6, 0] - 1e-3).max() < 1e-15 and np.all(iq[:, 2, 0] == 0.0)
