"""BASELINE.json configs[0] as the deck it is - python/test_data/SPE1CASE1/SPE1CASE1.DATA - on the CPU side: the host logic the device run
of tests/test_gpu_configs.py::test_spe1case1_report_steps rests on (decks.spe1_case / spe1_wells, wells.StandardWells, the well hooks of
newton.BlackoilModelHip, AdaptiveTimeStepping.advance_report_step), driven over the oracle.

What is checked against the DECK (data the reference tree holds): the SOLUTION section's numbers - 4800 psia at the 8400 ft datum, connate
water, Rs = 1.27 Mscf/stb - and the SCHEDULE section's - DRSDT 0, the wells' cells, targets and limits, the report steps.  What is checked
by properties (no output of this deck is in the tree): the well blocks B, C, D against finite differences of the well equations, the
Schur complement against the coupled system solved densely, rates on target and surface-volume balance over a report step."""
import numpy as np
import pytest

import oracle_bind

PSIA, FT, STB, MSCF, DAY = 6894.757293168361, 0.3048, 0.158987294928, 28.316846592, 86400.0


@pytest.fixture(scope="module")
def spe1(pkg, orc):
    fl = pkg.fluid.spe1_fluid()[0]
    case = pkg.decks.spe1_case(props=oracle_bind.OracleFluid(orc, fl))
    return case


def test_solution_section(pkg, spe1):
    pv = spe1["pv"].reshape(-1, 3)
    np.testing.assert_allclose(spe1["depth"][200:], 8400.0 * FT, rtol=1e-12)           # the datum is the centre of layer 3
    np.testing.assert_allclose(pv[200:, 1], 4800.0 * PSIA, rtol=1e-6)                  # EQUIL item 2 at item 1
    assert np.all(pv[:, 0] == 0.12) and np.all(spe1["meaning"] == 1)                   # contacts outside the reservoir: connate water, no free gas
    np.testing.assert_allclose(pv[:, 2], 1.27 * MSCF / STB, rtol=1e-12)                # RSVD, below RsSat(p): undersaturated
    grad = (pv[200, 1] - pv[0, 1]) / (spe1["depth"][200] - spe1["depth"][0])           # oil column: rho_o(p, Rs) g, about 0.3 psi/ft
    assert 0.25 < grad / PSIA * FT < 0.36
    rough = pkg.decks.spe1_case(state="rough")                                        # the fabricated state of earlier rounds is still there, by name
    assert np.abs(rough["pv"].reshape(-1, 3)[:, 1] - pv[:, 1]).max() < 20.0 * PSIA


def test_schedule_section(pkg, spe1):
    s = spe1["schedule"]
    assert spe1["drsdt"] == [0.0] and spe1["drsdt_all_cells"] == [1] and [round(t / DAY) for t in s["tstep"]] == [31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31]
    inj, prod = s["wells"]
    assert (inj["name"], inj["i"], inj["j"], inj["k_upper"], inj["k_lower"], inj["injected"]) == ("INJ", 0, 0, 0, 0, "gas")
    assert (prod["name"], prod["i"], prod["j"], prod["k_upper"], prod["k_lower"], prod["control"]) == ("PROD", 9, 9, 2, 2, "orat")
    np.testing.assert_allclose([inj["surface_rate"], prod["oil_rate"]], [100000.0 * MSCF / DAY, 20000.0 * STB / DAY], rtol=1e-12)
    np.testing.assert_allclose([inj["bhp_limit"], prod["bhp_limit"], inj["diameter"]], [9014.0 * PSIA, 1000.0 * PSIA, 0.5 * FT], rtol=1e-12)
    w = pkg.decks.spe1_wells(spe1)
    assert list(w.cells) == [0, 299] and list(w.vp) == [0, 1, 2]
    # Peaceman: 2 pi K h / ln(r0 / rw), r0 = 0.14 sqrt(dx^2 + dy^2); 500 mD x 20 ft and 200 mD x 50 ft give the same K h
    kh = 500.0 * 9.869232667160130e-16 * 20.0 * FT
    np.testing.assert_allclose([x.tw[0] for x in w.wells], 2 * np.pi * kh / np.log(0.14 * np.sqrt(2.0) * 1000.0 * FT / (0.25 * FT)), rtol=1e-12)


def _model(pkg, orc, case, drsdt=True):
    om = oracle_bind.OracleModel(orc, case)
    om.set_state(case["pv"], case["meaning"])
    if drsdt:
        om.set_composition_change_limits(drsdt=case["drsdt"], drsdt_all_cells=case["drsdt_all_cells"])
    return om


def test_well_blocks_against_finite_differences(pkg, orc, spe1):
    """B = d r_w / d(cell variables), D = d r_w / d(well unknowns), C^T = d r_cell / d(well unknowns), dsource = d(connection rates) /
    d(cell variables): each against central differences of the quantities wells.StandardWells itself evaluates, at a state with free gas
    at the injector and a producer that draws all three phases"""
    case = dict(spe1)
    pv = spe1["pv"].reshape(-1, 3).copy()
    meaning = spe1["meaning"].copy()
    pv[0] = [0.23, 5200.0 * PSIA, 0.27]              # (between the nodes of SWOF / SGOF: central differences must not straddle a kink)
    meaning[0] = 0                                   # free gas in the injector's cell
    pv[299] = [0.33, 4300.0 * PSIA, 0.14]             # (Sw + Sg and Sg + Sw - Swco, where the three-phase model reads its tables, are off the nodes too)
    meaning[299] = 0
    case["pv"], case["meaning"] = np.ascontiguousarray(pv.reshape(-1)), meaning
    # without the DRSDT bookkeeping: set_state re-initialises lastRs from the state it is given, so a perturbed pressure would move the
    # Rs cap along with it and the differences would see d RsSat / dp where the derivative record (rightly) sees a constant cap
    om = _model(pkg, orc, case, drsdt=False)
    wells = pkg.decks.spe1_wells(case)
    iq = om.iq()
    wells.solve_well_equations(iq)
    wells.x[:, :3] *= 1.03                            # off the well equations' solution: non-zero residuals
    wells.x[:, 3] += [2e5, -3e5]
    a = wells.assemble(iq, case["Nb"])
    W = a["wells"]
    B, C = W["Bnnzs"].reshape(2, 4, 3), W["Cnnzs"].reshape(2, 4, 3)
    D = np.stack([np.linalg.inv(W["Dnnzs"].reshape(2, 4, 4)[k]) for k in range(2)])
    rw0 = a["res_well"].reshape(2, 4)

    def residuals(pvx, xw):
        om.set_state(np.ascontiguousarray(pvx.reshape(-1)), meaning)
        wells.x = xw.copy()
        r = wells.assemble(om.iq(), case["Nb"])
        return r["res_well"].reshape(2, 4).copy(), r["source"].reshape(-1, 3).copy()
    x0 = wells.x.copy()
    for k, cell in enumerate((0, 299)):
        for v, h in enumerate((1e-6, 50.0, 1e-6 if meaning[cell] == 0 else 1e-4)):      # Sw, p, Sg
            pp, pm = pv.copy(), pv.copy()
            pp[cell, v] += h
            pm[cell, v] -= h
            (rp, sp), (rm, sm) = residuals(pp, x0), residuals(pm, x0)
            np.testing.assert_allclose(B[k][:, v], (rp[k] - rm[k]) / (2 * h), rtol=2e-5, atol=1e-9 * np.abs(B[k]).max())
            np.testing.assert_allclose(a["dsource"].reshape(-1, 3, 3)[cell][:, v], (sp[cell] - sm[cell]) / (2 * h), rtol=2e-5,
                                       atol=1e-9 * np.abs(a["dsource"]).max())
        for u, h in enumerate((1e-7, 1e-7, 1e-5, 100.0)):                                   # q_o, q_w, q_g, bhp
            xp, xm = x0.copy(), x0.copy()
            xp[k, u] += h
            xm[k, u] -= h
            (rp, sp), (rm, sm) = residuals(pv, xp), residuals(pv, xm)
            np.testing.assert_allclose(D[k][:, u], (rp[k] - rm[k]) / (2 * h), rtol=2e-5, atol=1e-9 * np.abs(D[k]).max())
            # C^T[e, u] = d r_cell[e] / d x_w[u] = - d source[e] / d x_w[u]
            np.testing.assert_allclose(C[k][u, :], -(sp[cell] - sm[cell]) / (2 * h), rtol=2e-5, atol=1e-9 * max(np.abs(C[k]).max(), 1e-30))
    assert np.abs(rw0).max() > 0.0


def test_schur_complement_equals_the_coupled_system(pkg, orc, spe1):
    """the elimination the device performs - r -= C^T D^-1 r_w, (A - C^T D^-1 B) x = r, x_w = D^-1 (r_w - B x) - against the coupled
    system [[A, C^T], [B, D]] solved densely (numpy), on the SPE1 Jacobian with the deck's wells"""
    om = _model(pkg, orc, spe1)
    wells = pkg.decks.spe1_wells(spe1)
    iq = om.iq()
    wells.solve_well_equations(iq)
    wells.x[:, 3] += [5e5, -5e5]
    a = wells.assemble(iq, spe1["Nb"])
    om.set_source(a["source"], a["dsource"])
    jac, res = om.assemble(DAY, 0)
    Nb, rp, ci = spe1["Nb"], spe1["rowptr"], spe1["col"]
    A = np.zeros((3 * Nb, 3 * Nb))
    for i in range(Nb):
        for k in range(rp[i], rp[i + 1]):
            A[3 * i:3 * i + 3, 3 * ci[k]:3 * ci[k] + 3] = jac[9 * k:9 * k + 9].reshape(3, 3)
    W = a["wells"]
    n = 3 * Nb
    K = np.zeros((n + 8, n + 8))
    K[:n, :n] = A
    for w, cell in enumerate((0, 299)):
        Bw, Cw = W["Bnnzs"].reshape(2, 4, 3)[w], W["Cnnzs"].reshape(2, 4, 3)[w]
        K[n + 4 * w:n + 4 * w + 4, 3 * cell:3 * cell + 3] = Bw
        K[3 * cell:3 * cell + 3, n + 4 * w:n + 4 * w + 4] = Cw.T
        K[n + 4 * w:n + 4 * w + 4, n + 4 * w:n + 4 * w + 4] = np.linalg.inv(W["Dnnzs"].reshape(2, 4, 4)[w])
    full = np.linalg.solve(K, np.concatenate([res, a["res_well"]]))
    r2 = orc.wells_apply_residual(W, a["res_well"], res)
    x, sr = orc.solve(Nb, rp, ci, jac, r2, tol=1e-13, maxit=400, w=0.9, wells=W)
    assert sr.converged
    np.testing.assert_allclose(x, full[:n], rtol=1e-6, atol=1e-9 * np.abs(full[:n]).max())
    np.testing.assert_allclose(orc.wells_recover(W, a["res_well"], x), full[n:], rtol=1e-6, atol=1e-9 * np.abs(full[n:]).max())


def _surface_volumes(iq, case):
    """oil and gas in place in surface volumes: sum over the cells of pore volume x (b_o S_o, b_g S_g + Rs b_o S_o)"""
    pvol = case["volume"] * iq[:, 16, 0]
    so, sg, bo, bg, rs = iq[:, 1, 0], iq[:, 2, 0], iq[:, 7, 0], iq[:, 8, 0], iq[:, 15, 0]
    return float((pvol * bo * so).sum()), float((pvol * (bg * sg + rs * bo * so)).sum())


def test_first_report_step_on_the_oracle(pkg, orc, spe1):
    """TSTEP 31 days under Flow's time-step control with the deck's wells: every sub-step converges, both wells end on their rate targets
    between their BHP limits, the surface volumes in place change by what the wells moved (oil: -20 000 stb/day; gas: +100 MMscf/day minus
    the producer's gas), free gas appears at the injector and - DRSDT 0 - Rs rises nowhere"""
    om = _model(pkg, orc, spe1)
    oil0, gas0 = _surface_volumes(om.iq(), spe1)
    wells = pkg.decks.spe1_wells(spe1)
    hm = oracle_bind.OracleAsHipModel(om, tol=1e-2, maxit=200, w=0.9)
    model = pkg.newton.BlackoilModelHip(hm, well_model=wells)
    ts = pkg.newton.AdaptiveTimeStepping(model, pkg.newton.TimeSteppingParameters(initial_dt=DAY))
    moved = np.zeros(3)          # surface volumes the wells put into the reservoir, summed sub-step by sub-step (implicit Euler: the converged
                                 # rates of a sub-step times its length are exactly what its discrete equations conserve)

    def on_accept(dt):
        moved[:] += (wells.x[0, :3] + wells.x[1, :3]) * dt
    ts.on_accept = on_accept
    reps = ts.advance_report_step(spe1["schedule"]["tstep"][0])
    np.testing.assert_allclose(ts.time, 31 * DAY, rtol=1e-12)
    assert all(ok for _, _, ok in ts.history) and len(ts.history) >= 4 and len(reps) >= 15
    inj, prod = wells.x
    s = spe1["schedule"]["wells"]
    np.testing.assert_allclose([inj[2], -prod[0]], [s[0]["surface_rate"], s[1]["oil_rate"]], rtol=1e-9)
    assert [w.control[0] for w in wells.wells] == ["rate", "rate"]
    assert 1000.0 * PSIA < prod[3] < 4800.0 * PSIA < inj[3] < 9014.0 * PSIA
    iq = om.iq()
    oil1, gas1 = _surface_volumes(iq, spe1)
    # oil: the rate is the target throughout; the balance closes to what the Newton method's own stopping rule leaves per sub-step - MB <= 1e-6
    # of the pore volume (8.4e7 m^3: 84 m^3), seven sub-steps - against 98 572 m^3 produced
    mb_slack = 1e-6 * float((spe1["volume"] * spe1["poro"]).sum()) * len(ts.history)
    assert abs((oil0 - oil1) - s[1]["oil_rate"] * 31 * DAY) <= 2.0 * mb_slack
    np.testing.assert_allclose(moved[0], -s[1]["oil_rate"] * 31 * DAY, rtol=1e-9)
    # gas: injected minus produced, the producer's gas rate integrated over the sub-steps (its cell drops below the bubble point within
    # the month - 4014.7 psia, PVTO - and free gas joins the dissolved gas in its stream)
    rs_prod = 1.27 * MSCF / STB
    assert abs((gas1 - gas0) - moved[2]) <= 2.0 * mb_slack * rs_prod       # (the gas equation's balance, in gas volumes)
    # (the producer's gas-oil ratio: its cell's Rs - below the initial 1.27 Mscf/stb once the pressure is under the bubble point - plus free gas)
    assert 0.9 * rs_prod < -prod[2] / -prod[0] < 1.5 * rs_prod and abs(moved[1]) < 1e-6 * abs(moved[0])     # no mobile water
    mm = om.get_state()[1]
    assert mm[0] == 0 and (mm == 0).sum() >= 3                           # free gas around the injector (and, by now, at the producer)
    assert np.all(iq[:, 15, 0] <= rs_prod * (1 + 1e-12))                  # DRSDT 0: Rs capped at its initial value everywhere
