"""Independent evidence for the oracle's assembly restatement (parity is UNPINNED inside the reference tree, see
oracle/blackoil.hpp): finite-difference check of the AD Jacobian, mass conservation of the TPFA fluxes, storage
bookkeeping, convergence norms and the Newton update rules.  CPU only."""
import numpy as np
import pytest

import oracle_bind

STATES = ["undersaturated", "saturated", "mixed"]


def make(pkg, orc, shape=(5, 4, 3), state="mixed", **kw):
    case = pkg.decks.cartesian_case(*shape, state=state, **kw)
    m = oracle_bind.OracleModel(orc, case)
    m.set_state(case["pv"], case["meaning"])
    return case, m


def dense(case, jac):
    Nb, rp, ci = case["Nb"], case["rowptr"], case["col"]
    A = np.zeros((Nb * 3, Nb * 3))
    for i in range(Nb):
        for k in range(rp[i], rp[i + 1]):
            A[3 * i:3 * i + 3, 3 * ci[k]:3 * ci[k] + 3] = jac[9 * k:9 * k + 9].reshape(3, 3)
    return A


@pytest.mark.parametrize("state", STATES)
def test_jacobian_matches_finite_differences(pkg, orc, state):
    case, m = make(pkg, orc, state=state, heterogeneous=True)
    dt = 86400.0
    m.assemble(dt, 0)  # fills the storage cache from the initial state
    # move to a different state so that the storage term is non-trivial
    rng = np.random.default_rng(0)
    pv = case["pv"].reshape(-1, 3).copy()
    pv[:, 0] += rng.uniform(-0.01, 0.01, len(pv))
    pv[:, 1] *= 1 + rng.uniform(-0.003, 0.003, len(pv))
    sg = case["meaning"] == 0
    pv[sg, 2] += rng.uniform(-0.01, 0.01, sg.sum())
    pv[~sg, 2] *= 1 + rng.uniform(-0.01, 0.01, (~sg).sum())
    m.set_state(pv.reshape(-1), case["meaning"])
    jac, r0 = m.assemble(dt, 1)
    J = dense(case, jac)
    scale = np.array([1e-7, 1.0, 1e-7])
    n = case["Nb"] * 3
    Jfd = np.zeros((n, n))
    for c in range(n):
        h = scale[c % 3] * (1.0 if (c % 3 != 2 or case["meaning"][c // 3] == 0) else 1e2)
        xp = pv.reshape(-1).copy(); xp[c] += h
        xm = pv.reshape(-1).copy(); xm[c] -= h
        m.set_state(xp, case["meaning"]); _, rp_ = m.assemble(dt, 1)
        m.set_state(xm, case["meaning"]); _, rm_ = m.assemble(dt, 1)
        Jfd[:, c] = (rp_ - rm_) / (2 * h)
    # compare column-scaled (variables have very different magnitudes)
    cs = np.maximum(np.abs(J).max(axis=0), 1e-300)
    err = np.abs(J - Jfd) / cs[None, :]
    assert err.max() < 2e-5, (err.max(), np.unravel_index(err.argmax(), err.shape))
    # pattern: nothing outside the stencil
    assert np.all((np.abs(Jfd) / cs[None, :] < 1e-6) | (np.abs(J) > 0))


@pytest.mark.parametrize("state", STATES)
def test_fluxes_conserve_mass(pkg, orc, state):
    case, m = make(pkg, orc, shape=(6, 5, 4), state=state, heterogeneous=True)
    jac, r = m.assemble(86400.0, 0)  # iteration 0: storage term is exactly zero, residual = sum of face fluxes
    r = r.reshape(-1, 3)
    assert np.abs(r).max() > 0
    for e in range(3):
        assert abs(r[:, e].sum()) <= 1e-12 * np.abs(r[:, e]).sum()


def test_storage_term_and_cache(pkg, orc):
    case, m = make(pkg, orc, shape=(3, 3, 2), state="saturated", perturb=False)
    # no flow: uniform depth-consistent state is not exactly hydrostatic, so cut the transmissibilities instead
    case["trans"][:] = 0.0
    m = oracle_bind.OracleModel(orc, case)
    m.set_state(case["pv"], case["meaning"])
    dt = 1000.0
    _, r0 = m.assemble(dt, 0)
    assert np.all(r0 == 0.0)
    iq0 = m.iq()
    pv = case["pv"].reshape(-1, 3).copy()
    pv[:, 1] *= 1.01
    m.set_state(pv.reshape(-1), case["meaning"])
    _, r1 = m.assemble(dt, 1)
    iq1 = m.iq()

    def storage(iq):  # [oil, water, gas] = phi * (So bo, Sw bw, Sg bg + Rs So bo)
        S, b, rs, phi = iq[:, 0:3, 0], iq[:, 6:9, 0], iq[:, 15, 0], iq[:, 16, 0]
        return np.stack([phi * S[:, 1] * b[:, 1], phi * S[:, 0] * b[:, 0], phi * (S[:, 2] * b[:, 2] + rs * S[:, 1] * b[:, 1])], axis=1)

    want = (storage(iq1) - storage(iq0)) * (case["volume"][:, None] / dt)
    np.testing.assert_allclose(r1.reshape(-1, 3), want, rtol=1e-9, atol=1e-18)


def test_intensive_quantities_sanity(pkg, orc):
    case, m = make(pkg, orc, shape=(4, 3, 2), state="mixed")
    iq = m.iq()
    S, p, b, mob, rho, rs, phi = iq[:, 0:3], iq[:, 3:6], iq[:, 6:9], iq[:, 9:12], iq[:, 12:15], iq[:, 15], iq[:, 16]
    np.testing.assert_allclose(S[:, :, 0].sum(axis=1), 1.0, rtol=1e-14)
    assert np.all(S[:, 0, 1] == 1.0) and np.all(p[:, 1, 2] == 1.0)  # dSw/dx0, dpo/dx1
    sat = case["meaning"] == 0
    fl = case["fluid"]
    np.testing.assert_allclose(rs[sat, 0], pkg.decks.rs_sat(fl, p[sat, 1, 0]), rtol=1e-12)
    assert np.all(rs[sat, 2] > 0) and np.all(rs[~sat, 3] == 1.0)  # saturated: dRs/dp ; undersaturated: dRs/dx2
    assert np.all(b[:, :, 0] > 0) and np.all(mob[:, :, 0] >= 0) and np.all(rho[:, :, 0] > 0)
    assert np.all(rho[:, 2, 0] < rho[:, 1, 0]) and np.all(rho[:, 1, 0] < rho[:, 0, 0])
    assert np.all(phi[:, 0] > 0.25) and np.all(phi[:, 2] > 0)  # rock compressibility: dphi/dp > 0


def test_convergence_norms(pkg, orc):
    case, m = make(pkg, orc, shape=(5, 4, 3), state="mixed", heterogeneous=True)
    dt = 86400.0
    _, r = m.assemble(dt, 0)
    c = m.convergence(dt, 1e-2)
    r = r.reshape(-1, 3)
    pv = case["poro"] * case["volume"]
    iq = m.iq()
    Bavg = np.array([(1.0 / iq[:, 6 + ph, 0]).mean() for ph in (1, 0, 2)])  # equations oil, water, gas
    np.testing.assert_allclose(c[6:9], Bavg, rtol=1e-13)
    np.testing.assert_allclose(c[0:3], r.sum(axis=0), rtol=1e-9, atol=1e-25)
    np.testing.assert_allclose(c[3:6], (np.abs(r) / pv[:, None]).max(axis=0), rtol=1e-13)
    np.testing.assert_allclose(c[9], pv.sum(), rtol=1e-13)
    np.testing.assert_allclose(c[11:14], Bavg * dt * c[3:6], rtol=1e-13)
    np.testing.assert_allclose(c[14:17], np.abs(Bavg * c[0:3]) * dt / pv.sum(), rtol=1e-12, atol=1e-30)


def test_update_chopping_and_switching(pkg, orc):
    case, m = make(pkg, orc, shape=(3, 2, 2), state="mixed", perturb=False)
    Nb = case["Nb"]
    pv0 = case["pv"].reshape(-1, 3).copy()
    mean0 = case["meaning"].copy()
    dx = np.zeros((Nb, 3))
    dx[:, 1] = 0.5 * pv0[:, 1]          # asks for -50 % pressure: chopped to 30 %
    dx[:, 0] = 0.5                      # asks for dSw = -0.5: saturations limited to 0.2
    sat = mean0 == 0
    dx[sat, 2] = 0.3
    m.update(dx.reshape(-1))
    pv1, mean1 = m.get_state()
    pv1 = pv1.reshape(-1, 3)
    np.testing.assert_allclose(pv1[:, 1], 0.7 * pv0[:, 1], rtol=1e-14)
    # saturated cells: deltas (Sw 0.5, Sg 0.3, So -0.8) -> alpha = 0.2/0.8
    np.testing.assert_allclose(pv1[sat, 0], pv0[sat, 0] - 0.5 * 0.25, rtol=1e-13)
    # Sg went 0.1 - 0.075 = 0.025 > 0: no switch
    assert np.all(mean1[sat] == 0)
    np.testing.assert_allclose(pv1[sat, 2], pv0[sat, 2] - 0.3 * 0.25, rtol=1e-12)
    # undersaturated cells: alpha = 0.2/0.5 ; Rs untouched (delta 0); pressure drop makes RsSat(p) < Rs -> gas appears
    np.testing.assert_allclose(pv1[~sat, 0], pv0[~sat, 0] - 0.2, rtol=1e-13)
    rs_sat_new = pkg.decks.rs_sat(case["fluid"], pv1[~sat, 1])
    sw = pv0[~sat, 2] > rs_sat_new
    assert sw.any()
    assert np.all(mean1[~sat][sw] == 0) and np.all(pv1[~sat][sw, 2] == 0.0)
    # now drive Sg negative in a saturated cell: switches to Rs = RsSat(p)
    dx2 = np.zeros((Nb, 3))
    dx2[sat, 2] = 0.1
    m.update(dx2.reshape(-1))
    pv2, mean2 = m.get_state()
    pv2 = pv2.reshape(-1, 3)
    gone = sat & (pv1[:, 2] - 0.1 < 0)
    assert gone.any() and np.all(mean2[gone] == 1)
    np.testing.assert_allclose(pv2[gone, 2], pkg.decks.rs_sat(case["fluid"], pv2[gone, 1]), rtol=1e-12)


def test_spe1_case_assembles(pkg, orc):
    # (the deck's EQUIL state on the oracle's property functions: without arguments spe1_case asks the device's, and there is no GPU here)
    case = pkg.decks.spe1_case(props=oracle_bind.OracleFluid(orc, pkg.fluid.spe1_fluid()[0]))
    assert case["Nb"] == 300 and len(case["col"]) == 1780  # faces 740 -> nnzb = 300 + 2*740
    m = oracle_bind.OracleModel(orc, case)
    m.set_state(case["pv"], case["meaning"])
    jac, r = m.assemble(86400.0, 0)
    assert np.all(np.isfinite(jac)) and np.all(np.isfinite(r))
    x, res = m.solve(tol=1e-2)
    assert res.converged


# ---- wet gas (PVTG / Rv / third primary-variable meaning) and rock compaction tables -------------------------------------
def _fd_check(case, m, dt=86400.0, tol=2e-5):
    m.assemble(dt, 0)
    rng = np.random.default_rng(0)
    pv = case["pv"].reshape(-1, 3).copy()
    mean = case["meaning"]
    pv[:, 0] += rng.uniform(-0.01, 0.01, len(pv))
    pv[:, 1] *= 1 + rng.uniform(-0.003, 0.003, len(pv))
    sg = mean == 0
    pv[sg, 2] += rng.uniform(-0.01, 0.01, sg.sum())
    pv[~sg, 2] *= 1 + rng.uniform(-0.01, 0.01, (~sg).sum())
    m.set_state(pv.reshape(-1), mean)
    jac, r0 = m.assemble(dt, 1)
    J = dense(case, jac)
    n = case["Nb"] * 3
    Jfd = np.zeros((n, n))
    for c in range(n):
        k, mg = c % 3, mean[c // 3]
        h = [1e-7, 1.0, 1e-7][k]
        if k == 2 and mg == 1:
            h = 1e-5            # Rs ~ 1e2
        elif k == 2 and mg == 2:
            h = 1e-11           # Rv ~ 1e-4
        xp = pv.reshape(-1).copy(); xp[c] += h
        xm = pv.reshape(-1).copy(); xm[c] -= h
        m.set_state(xp, mean); _, rp_ = m.assemble(dt, 1)
        m.set_state(xm, mean); _, rm_ = m.assemble(dt, 1)
        Jfd[:, c] = (rp_ - rm_) / (2 * h)
    cs = np.maximum(np.abs(J).max(axis=0), 1e-300)
    err = np.abs(J - Jfd) / cs[None, :]
    assert err.max() < tol, (err.max(), np.unravel_index(err.argmax(), err.shape))


def test_wetgas_jacobian_matches_finite_differences(pkg, orc):
    from helpers import wetgas_case
    case = wetgas_case(pkg, 4, 3, 6, heterogeneous=True)
    assert set(case["meaning"]) == {0, 1, 2}
    m = oracle_bind.OracleModel(orc, case)
    m.set_state(case["pv"], case["meaning"])
    _fd_check(case, m, tol=5e-5)


def test_wetgas_fluxes_conserve_mass_and_carry_vaporised_oil(pkg, orc):
    from helpers import wetgas_case
    case = wetgas_case(pkg, 5, 4, 6, heterogeneous=True)
    m = oracle_bind.OracleModel(orc, case)
    m.set_state(case["pv"], case["meaning"])
    _, r = m.assemble(86400.0, 0)
    r = r.reshape(-1, 3)
    for e in range(3):
        assert abs(r[:, e].sum()) <= 1e-12 * np.abs(r[:, e]).sum()
    iq = m.iq()
    assert iq.shape[1] == 19
    top = case["meaning"] == 2
    assert np.all(iq[top, 1, 0] == 0.0) and np.all(iq[top, 16, 0] > 0)       # no oil phase, Rv > 0
    assert np.abs(r[top, 0]).max() > 0                                        # yet the oil component moves (with the gas)
    # gas density carries the vaporised oil: rho_g = b_g (rho_g,ref + Rv rho_o,ref)
    rr = case["fluid"].pvt[0]["density"]
    np.testing.assert_allclose(iq[:, 14, 0], iq[:, 8, 0] * rr[2] + iq[:, 8, 0] * iq[:, 16, 0] * rr[0], rtol=1e-14)


def test_wetgas_meaning_switches(pkg, orc):
    """(Sw, po, Sg) -> (Sw, pg, Rv) when the oil saturation turns negative, and back when Rv exceeds the saturated value"""
    from helpers import wetgas_case, rv_sat
    case = wetgas_case(pkg, 3, 3, 6, perturb=False)
    m = oracle_bind.OracleModel(orc, case)
    m.set_state(case["pv"], case["meaning"])
    mean = case["meaning"]
    pv = case["pv"].reshape(-1, 3)
    dx = np.zeros_like(pv)
    three = np.flatnonzero(mean == 0)
    dx[three, 2] = -0.2           # Sg += 0.2 per update (the chop limit): So = 0.7 - ... turns negative after four updates
    for _ in range(4):
        m.update(dx.reshape(-1))
    p1, m1 = m.get_state()
    assert np.all(m1[three] == 2) and np.all(m1[mean == 2] == 2)
    p1 = p1.reshape(-1, 3)
    np.testing.assert_allclose(p1[three, 2], rv_sat(case["fluid"], p1[three, 1]), rtol=1e-12)   # starts saturated
    # now push Rv above saturation: oil re-appears, Sg = 1 - Sw
    dx[:] = 0.0
    dx[three, 2] = -2.0 * p1[three, 2]
    m.update(dx.reshape(-1))
    p2, m2 = m.get_state()
    p2 = p2.reshape(-1, 3)
    assert np.all(m2[three] == 0)
    np.testing.assert_allclose(p2[three, 2], 1.0 - p2[three, 0], rtol=0, atol=0)


def test_rocktab_jacobian_matches_finite_differences(pkg, orc):
    from helpers import wetgas_case, ROCKTAB_2
    case = wetgas_case(pkg, 4, 3, 5, rocktab=ROCKTAB_2, heterogeneous=True)
    m = oracle_bind.OracleModel(orc, case)
    m.set_state(case["pv"], case["meaning"])
    iq = m.iq()
    assert iq[:, 17, 0].min() < 0.99 and iq[:, 17, 0].max() > 0.9 and np.abs(iq[:, 17, 2]).max() > 0   # tmult active, depends on p
    _fd_check(case, m, tol=5e-5)


def test_vappars_jacobian_matches_finite_differences(pkg, orc):
    """VAPPARS (maximum oil saturation below which the saturated Rs / Rv shrink with (S_o / S_o,max)^vap): the AD derivative
    of the power against central differences, with oil saturations below the tracked maximum"""
    import helpers
    case = helpers.wetgas_case(pkg, 4, 4, 5, heterogeneous=True)
    m = oracle_bind.OracleModel(orc, case)
    m.set_state(case["pv"], case["meaning"])
    m.set_vappars(0.5, 0.8)
    pv = case["pv"].reshape(-1, 3).copy()
    pv[case["meaning"] == 0, 2] += 0.06
    pv[:, 0] += 0.02
    shifted = dict(case, pv=np.ascontiguousarray(pv.reshape(-1)))
    m.set_state(shifted["pv"], case["meaning"])
    a = m.iq()
    m.set_vappars(0.0, 0.0, enable=False)
    assert not np.array_equal(a[:, 15, 0], m.iq()[:, 15, 0])
    m.set_state(case["pv"], case["meaning"])
    m.set_vappars(0.5, 0.8)
    _fd_check(shifted, m, tol=5e-5)


def test_water_compaction_jacobian_matches_finite_differences(pkg, orc):
    """water-induced compaction (ROCK2D / ROCK2DTR): the pore-volume and transmissibility multipliers over (effective oil
    pressure, SwMax - Sw_initial) - AD derivatives in p_o and, where S_w is above the tracked maximum, in S_w - against
    central differences"""
    import helpers
    case = helpers.wetgas_case(pkg, 4, 4, 5, heterogeneous=True)
    case["rocknum"] = (np.arange(case["Nb"]) % 2).astype(np.int32)
    m = oracle_bind.OracleModel(orc, case)
    m.set_state(case["pv"], case["meaning"])
    m.set_water_compaction(helpers.ROCK2D_2)
    pv = case["pv"].reshape(-1, 3).copy()
    up = np.arange(case["Nb"]) % 3 != 0
    pv[up, 0] += 0.05                 # above the tracked maximum: SwMax = Sw, with its derivative
    pv[~up, 0] -= 0.02                # below: SwMax is the constant
    shifted = dict(case, pv=np.ascontiguousarray(pv.reshape(-1)))
    m.set_state(shifted["pv"], case["meaning"])
    iq = m.iq()
    nf = iq.shape[1]
    assert np.all(iq[up, nf - 1, 1] != 0.0) and np.all(iq[~up, nf - 1, 1] == 0.0)      # d poro / d Sw
    assert np.abs(iq[:, nf - 2, 2]).max() > 0                                           # d tmult / d p
    _fd_check(shifted, m, tol=5e-5)
