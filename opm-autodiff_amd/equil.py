"""Hydrostatic equilibration of a black-oil model (EQUIL keyword): phase pressures, saturations and Rs per cell.

Deck-side input generation for the hot path, a restatement of the reference's
Opm::EQUIL::DeckDependent::InitialStateComputer for one equilibration region with live oil, dry gas and water:

  ebos/equil/initstateequil.hh  :79-147   RK4IVP (fixed-step RK4 + Hermite evaluation)
                                :150-285  PhasePressODE::Water / Oil / Gas  (dp/dz = rho(z, p) g)
                                :596-723  PressureTable::equil_WOG / equil_GOW / equil_OWG (which phase starts at the datum)
                                :880-893, 1150-1330 PhaseSaturations (capillary-pressure inversion, overlapping transition zones,
                                          pressure corrections at the saturation end points)
                                :1985-2012 equilibrateCellCentres (EQUIL item 9 = 0)
  ebos/equil/equilibrationhelpers.hh :186-254, 469-537 RsVD, RsSatAtContact ; :730-960 satFromPc, satFromSumOfPcs, satFromDepth

The fluid and saturation functions are NOT restated here: `props.probe(p, rs, sw, sg)` evaluates them - on the device
through capi.HipFluid (opmhip_fluid_probe), or, in tests, through the CPU oracle - so that the numbers pinned by the
reference's tests/test_equil.cc pin those very functions.
"""
import numpy as np

NSAMPLE = 2000  # PressureTable's default number of RK4 intervals per direction (initstateequil.hh:302)
INVBW, INVBG, INVBO, RSSAT, PCOW, PCGO = range(6)


class RK4IVP:
    """y' = f(x, y), y(span[0]) = y0 on N equal steps; evaluation by the cubic Hermite of the reference (:120-136)"""

    def __init__(self, f, span, y0, N):
        self.N, self.span = N, span
        h = self.h = (span[1] - span[0]) / N
        y, fv = [y0], [f(span[0], y0)]
        for i in range(N):
            x = span[0] + i * h
            yi, k1 = y[-1], fv[i]
            k2 = f(x + h / 2, yi + h / 2 * k1)
            k3 = f(x + h / 2, yi + h / 2 * k2)
            k4 = f(x + h, yi + h * k3)
            y.append(yi + h / 6 * (k1 + 2 * (k2 + k3) + k4))
            fv.append(f(x + h, y[-1]))
        self.y, self.f = y, fv

    def __call__(self, x):
        h = self.h
        i = int((x - self.span[0]) / h)
        t = (x - (self.span[0] + i * h)) / h
        i = min(max(i, 0), self.N - 1)
        y0, y1, f0, f1 = self.y[i], self.y[i + 1], self.f[i], self.f[i + 1]
        u = (1 - 2 * t) * (y1 - y0)
        u += h * ((t - 1) * f0 + t * f1)
        u *= t * (t - 1)
        u += (1 - t) * y0 + t * y1
        return u


class PressureFunction:
    def __init__(self, ode, depth0, p0, span, nsample=NSAMPLE):
        self.depth0, self.p0 = depth0, p0
        self.up = RK4IVP(ode, (depth0, span[0]), p0, nsample)
        self.down = RK4IVP(ode, (depth0, span[1]), p0, nsample)

    def __call__(self, depth):
        if depth < self.depth0:
            return self.up(depth)
        if depth > self.depth0:
            return self.down(depth)
        return self.p0


class RsSatAtContact:
    """min(RsSat(p), RsSat(p at the gas-oil contact)); the saturated value where free gas is present (:469-537)"""

    def __init__(self, props, p_contact):
        self.props = props
        self.rs_contact = self.sat_rs(p_contact)

    def sat_rs(self, p):
        return float(self.props.probe(p)[0, RSSAT])

    def __call__(self, depth, p, sat_gas=0.0):
        if sat_gas > 0.0:
            return self.sat_rs(p)
        return min(self.sat_rs(p), self.rs_contact)


class RsVD:
    """Rs from a depth table (RSVD), capped by RsSat(p) (:186-254)"""

    def __init__(self, props, depth, rs):
        self.props, self.d, self.rs = props, np.asarray(depth, float), np.asarray(rs, float)

    def __call__(self, depth, p, sat_gas=0.0):
        sat = float(self.props.probe(p)[0, RSSAT])
        if sat_gas > 0.0:
            return sat
        if self.d[0] > depth:
            return float(self.rs[0])
        if self.d[-1] < depth:
            return float(self.rs[-1])
        return min(sat, float(np.interp(depth, self.d, self.rs)))


class PBVD:
    """Rs = RsSat(min(bubble-point pressure at this depth, cell pressure)) (PBVD, :256-322)"""

    def __init__(self, props, depth, pbub):
        self.props, self.d, self.pb = props, np.asarray(depth, float), np.asarray(pbub, float)

    def __call__(self, depth, p, sat_gas=0.0):
        press = p
        if sat_gas <= 0.0:
            press = float(self.pb[0]) if self.d[0] > depth else float(self.pb[-1]) if self.d[-1] < depth else float(np.interp(depth, self.d, self.pb))
        return float(self.props.probe(min(press, p))[0, RSSAT])


def _rv_sat(props, p):
    return float(props.probe_gas(p)[0, 2])


class RvSatAtContact:
    """min(RvSat(p), RvSat(p at the gas-oil contact)); the saturated value where oil is present (:540-607)"""

    def __init__(self, props, p_contact):
        self.props = props
        self.rv_contact = _rv_sat(props, p_contact)

    def __call__(self, depth, p, sat_oil=0.0):
        if sat_oil > 0.0:
            return _rv_sat(self.props, p)
        return min(_rv_sat(self.props, p), self.rv_contact)


class RvVD:
    """Rv from a depth table (RVVD), capped by RvSat(p) (:393-468)"""

    def __init__(self, props, depth, rv):
        self.props, self.d, self.rv = props, np.asarray(depth, float), np.asarray(rv, float)

    def __call__(self, depth, p, sat_oil=0.0):
        if abs(sat_oil) > 1e-16:
            return _rv_sat(self.props, p)
        if self.d[0] > depth:
            return float(self.rv[0])
        if self.d[-1] < depth:
            return float(self.rv[-1])
        return min(_rv_sat(self.props, p), float(np.interp(depth, self.d, self.rv)))


class PDVD:
    """Rv = RvSat(min(dew-point pressure at this depth, cell pressure)) (PDVD, :324-391)"""

    def __init__(self, props, depth, pdew):
        self.props, self.d, self.pd = props, np.asarray(depth, float), np.asarray(pdew, float)

    def __call__(self, depth, p, sat_oil=0.0):
        press = p
        if sat_oil <= 0.0:
            press = float(self.pd[0]) if self.d[0] > depth else float(self.pd[-1]) if self.d[-1] < depth else float(np.interp(depth, self.d, self.pd))
        return _rv_sat(self.props, min(press, p))


def phase_pressure_tables(props, rho_ref, rec, z_span, grav=9.80665, rs_func=None, rv_func=None, nsample=NSAMPLE):
    """PressureTable::equilibrate (initstateequil.hh:596-723): the three phase pressures as functions of depth over
    z_span widened to the contacts; which phase starts at the datum follows from the datum's zone.
    rs_func(depth, p_o) / rv_func(depth, p_g): dissolved gas / vaporised oil along the column (None: none).
    -> (water, oil, gas) callables"""
    rho_o, rho_w, rho_g = rho_ref

    def f_water(z, p):
        return float(props.probe(p)[0, INVBW]) * rho_w * grav

    def f_oil(z, p):
        rs = rs_func(z, p) if rs_func is not None else 0.0
        b = float(props.probe(p, rs=rs)[0, INVBO])     # the probe switches to the saturated curve where rs >= RsSat(p)
        return (b * rho_o + rs * b * rho_g) * grav

    def f_gas(z, p):
        if rv_func is None:
            return float(props.probe(p)[0, INVBG]) * rho_g * grav
        # PhasePressODE::Gas (initstateequil.hh:240-285): vaporised oil adds to the gas density
        rv = rv_func(z, p)
        b = float(props.probe_gas(p, rv)[0, 0])       # saturated curve where rv >= RvSat(p)
        return (b * rho_g + rv * b * rho_o) * grav

    span = (min(z_span[0], rec["zgoc"], rec["zwoc"]), max(z_span[1], rec["zgoc"], rec["zwoc"]))
    mk = lambda ode, z0, p0: PressureFunction(ode, z0, p0, span, nsample)
    if rec["datum"] > rec["zwoc"]:      # datum in the water zone
        wat = mk(f_water, rec["datum"], rec["pressure"])
        oil = mk(f_oil, rec["zwoc"], wat(rec["zwoc"]) + rec["pcow_woc"])
        gas = mk(f_gas, rec["zgoc"], oil(rec["zgoc"]) + rec["pcgo_goc"])
    elif rec["datum"] < rec["zgoc"]:    # datum in the gas zone
        gas = mk(f_gas, rec["datum"], rec["pressure"])
        oil = mk(f_oil, rec["zgoc"], gas(rec["zgoc"]) - rec["pcgo_goc"])
        wat = mk(f_water, rec["zwoc"], oil(rec["zwoc"]) - rec["pcow_woc"])
    else:                               # datum in the oil zone
        oil = mk(f_oil, rec["datum"], rec["pressure"])
        wat = mk(f_water, rec["zwoc"], oil(rec["zwoc"]) - rec["pcow_woc"])
        gas = mk(f_gas, rec["zgoc"], oil(rec["zgoc"]) + rec["pcgo_goc"])
    return wat, oil, gas


def _root(fun, s0, s1):
    """zero of a monotone function with fun(s0) > 0 > fun(s1) (RegulaFalsiBisection to 1e-10 in the reference)"""
    f0, f1 = fun(s0), fun(s1)
    if f0 <= 0.0:
        return s0
    if f1 >= 0.0:
        return s1
    a, b = s0, s1
    for _ in range(200):
        m = 0.5 * (a + b)
        if fun(m) > 0.0:
            a = m
        else:
            b = m
        if abs(b - a) < 1e-13:
            break
    return 0.5 * (a + b)


def sat_from_pc(pc_of_s, smin, smax, target, increasing):
    """satFromPc (equilibrationhelpers.hh:730-830): the saturation at which the capillary pressure equals `target`,
    clamped to [smin, smax]; `increasing`: pc grows with the saturation (gas-oil) or falls (oil-water)"""
    s0, s1 = (smax, smin) if increasing else (smin, smax)
    return _root(lambda s: pc_of_s(s) - target, s0, s1)


def sat_from_sum_of_pcs(pcow_of_sw, pcgo_of_sg, swl, swu, target):
    """satFromSumOfPcs (equilibrationhelpers.hh:850-930): water saturation of a gas-water contact,
    pcow(sw) + pcgo(1 - sw) = target"""
    return _root(lambda s: pcow_of_sw(s) + pcgo_of_sg(1.0 - s) - target, swl, swu)


def equilibrate_regions(eqlnum, records, props, rho_ref, cell_depth, cell_zmin, cell_zmax, sat_limits, grav=9.80665,
                        rs_funcs=None, rv_funcs=None, nsample=NSAMPLE, swatinit=None, cell_zspan=None):
    """InitialStateComputer::calcPressSatRsRv (initstateequil.hh:1882-1940): every equilibration region (EQLNUM, 0-based)
    with its own EQUIL record over the vertical extent of ITS cells.  props / rho_ref / sat_limits / rs_funcs / rv_funcs: one
    per region (list) or one for all.  -> the dict of `equilibrate`, arrays over all cells (zeros where no region applies)"""
    eqlnum = np.asarray(eqlnum, int)
    n = len(eqlnum)
    cell_depth, cell_zmin, cell_zmax = (np.asarray(a, float) for a in (cell_depth, cell_zmin, cell_zmax))
    per = lambda x, r: x[r] if isinstance(x, list) else x   # a list: one per region
    out = {k: np.zeros(n) for k in ("pw", "po", "pg", "sw", "so", "sg", "rs", "rv")}
    if swatinit is not None:
        swatinit = np.asarray(swatinit, float)
        out["pcw_scale"] = np.ones(n)
    for r, rec in enumerate(records):
        cells = np.nonzero(eqlnum == r)[0]
        if len(cells) == 0:
            continue
        if rec.get("accuracy", 0) > 0:
            raise ValueError("EQUIL record %d: positive item 9 is not supported (neither is it by the reference)" % (r + 1))
        zspan = np.stack([cell_zmin[cells], cell_zmax[cells]], axis=1) if cell_zspan is None else np.asarray(cell_zspan, float)[cells]
        span = (float(cell_zmin[cells].min()), float(cell_zmax[cells].max()))
        rs_f = rs_funcs[r] if rs_funcs is not None else None
        rv_f = rv_funcs[r] if rv_funcs is not None else None
        res = equilibrate(per(props, r), per(rho_ref, r), rec, cell_depth[cells], span, per(sat_limits, r), grav=grav,
                          rs_func=rs_f, nsample=nsample, rv_func=rv_f, swatinit=None if swatinit is None else swatinit[cells],
                          cell_zspan=zspan)
        for k in out:
            out[k][cells] = res[k]
    return out


def equilibrate(props, rho_ref, rec, cell_depth, z_span, sat_limits, grav=9.80665, rs_func=None, nsample=NSAMPLE, rv_func=None,
                swatinit=None, cell_zspan=None, endscale=None):
    """props.probe(p, rs, sw, sg) -> (n, 8) (capi.HipFluid layout); rho_ref = (oil, water, gas) surface densities;
    rec = dict(datum, pressure, zwoc, pcow_woc, zgoc, pcgo_goc); cell_depth[]: centre depths; z_span = (top, bottom) of the
    region's cells; sat_limits = dict(Swl, Swu, Sgl, Sgu) (unscaled end points of the saturation tables).
    rv_func (wet gas, VAPOIL): Rv(depth, p_g, sat_oil) - RvSatAtContact / RvVD / PDVD; props then also needs
    probe_gas(p, rv) -> (n, 3): 1/B_g(p, rv), mu_g, RvSat(p).
    swatinit (SWATINIT, per cell): the water saturation is imposed instead of derived, and the cell's oil-water capillary
    pressure curve is rescaled so that the imposed saturation is in equilibrium with the phase pressures
    (deriveWaterSat initstateequil.hh:1186-1214 -> applySwatInit :1330-1343 -> EclMaterialLawManager::applySwatinit of
    opm-material, absent here, restated from its published form: pcow < 0 -> Swu; else Sw = max(Sw, Swl) and, where
    |pcow(Sw)| > 1 Pa, maxPcow *= pcow / pcow(Sw)).  The result then carries "pcw_scale": the factor on the cell's pcow
    curve (scaled maxPcow / table maxPcow), to be handed to the device as PCW = pcw_scale * pcow(Swl) (opmhip_set_pcw).
    rec["accuracy"] (EQUIL item 9): 0 = cell centres; -N = the average over 2N horizontal slices of each cell between its
    mean top and mean bottom depth (cell_zspan (n, 2), cellZSpan :1495-1512), Rs / Rv then from the averaged state at the
    centre depth; positive values are refused, as the reference does.
    endscale (ENDSCALE family): the dict capi.HipModel.set_endpoint_scaling takes - flags and per-cell arrays of scaled end
    points (over THESE cells).  Every cell then inverts ITS capillary pressure curves (props.sat_probe(sw, sg, end points of the
    cell): the scaled EclEpsTwoPhaseLaw the device evaluates) between ITS end points, as the reference does through the
    material-law manager (satFromPc with oilWaterScaledEpsInfoDrainage(cell), equilibrationhelpers.hh:730-960;
    accountForScaledSaturations, initstateequil.hh:1257-1330).
    -> dict(pw, po, pg, sw, so, sg, rs, rv) arrays over the cells."""
    if rs_func is None:
        if rec["zgoc"] != rec["datum"]:
            raise ValueError("without an RSVD table the datum depth must be at the gas-oil contact")
        rs_func = RsSatAtContact(props, rec["pressure"])
    wat, oil, gas = phase_pressure_tables(props, rho_ref, rec, z_span, grav, rs_func, rv_func, nsample)

    Swl, Swu, Sgl, Sgu = (sat_limits[k] for k in ("Swl", "Swu", "Sgl", "Sgu"))
    pcow_table = lambda sw: float(props.probe(1e5, sw=sw)[0, PCOW])
    pcgo = lambda sg: float(props.probe(1e5, sg=sg)[0, PCGO])
    es_flags, es_arrays = {}, {}
    if endscale is not None:
        for k, v in endscale.items():
            if v is None:
                continue
            if np.ndim(v) == 0:
                es_flags[k] = v
            else:
                es_arrays[k] = np.asarray(v, float)
                if es_arrays[k].shape != (len(cell_depth),):
                    raise ValueError("endscale[%r]: one value per cell expected" % k)

    root = _root

    n = len(cell_depth)
    out = {k: np.zeros(n) for k in ("pw", "po", "pg", "sw", "so", "sg", "rs", "rv")}
    if swatinit is not None:
        swatinit = np.asarray(swatinit, float)
        if swatinit.shape != (n,):
            raise ValueError("swatinit: one value per cell expected")
        out["pcw_scale"] = np.ones(n)
    limits0 = (Swl, Swu, Sgl, Sgu)
    const0 = (abs(pcow_table(Swl) - pcow_table(Swu)) < np.finfo(float).eps, abs(pcgo(Sgl) - pcgo(Sgu)) < np.finfo(float).eps)
    const_pcow, const_pcgo = const0
    acc = int(rec.get("accuracy", 0) or 0)
    if acc > 0:
        raise ValueError("EQUIL item 9 > 0 is not supported (neither is it by the reference, initstateequil.hh:1902-1908)")
    if acc < 0:
        if cell_zspan is None:
            raise ValueError("EQUIL item 9 < 0 (horizontal subdivision) needs cell_zspan: (top, bottom) depth of every cell")
        if swatinit is not None:
            raise ValueError("SWATINIT together with EQUIL item 9 < 0 is not supported here")
        cell_zspan = np.asarray(cell_zspan, float).reshape(n, 2)
    for c, zc in enumerate(cell_depth):
        scale = [1.0]                                   # this cell's factor on the pcow curve (SWATINIT)
        if endscale is not None:    # this cell's curves and end points (EclEpsTwoPhaseLaw with the cell's scaled points)
            es_c = dict(es_flags, **{k: float(a[c]) for k, a in es_arrays.items()})
            Swl, Swu, Sgl, Sgu = (es_c.get(k, d) for k, d in zip(("swl", "swu", "sgl", "sgu"), limits0))
            pcow_table = lambda sw, es_c=es_c: float(props.sat_probe(sw, 0.0, es_c)[0, 3])
            pcgo = lambda sg, es_c=es_c, swl=Swl: float(props.sat_probe(swl, sg, es_c)[0, 4])
            const_pcow = abs(pcow_table(Swl) - pcow_table(Swu)) < np.finfo(float).eps
            const_pcgo = abs(pcgo(Sgl) - pcgo(Sgu)) < np.finfo(float).eps
        pcow = lambda sw: scale[0] * pcow_table(sw)

        def apply_swatinit(pc, sw_in):
            if pc < 0.0:
                return Swu
            sw_c = max(sw_in, Swl)
            at_sw = pcow(sw_c)
            if abs(at_sw) > 1.0:                        # Pascal: no division by a vanishing capillary pressure
                scale[0] *= pc / at_sw
            return sw_c

        def at_depth(z):
            """deriveSaturations + correctedPhasePressures at one depth (initstateequil.hh:1100-1330)"""
            po, pg, pw = oil(z), gas(z), wat(z)
            # water: dPcow/dSw <= 0 ; gas: dPcgo/dSg >= 0
            if const_pcow:
                sw = Swl if z < rec["zwoc"] else Swu
            elif swatinit is not None:
                sw = apply_swatinit(po - pw, swatinit[c])
            else:
                sw = sat_from_pc(pcow, Swl, Swu, po - pw, increasing=False)
            if const_pcgo:
                sg = Sgu if z < rec["zgoc"] else Sgl
            else:
                sg = sat_from_pc(pcgo, Sgl, Sgu, pg - po, increasing=True)
            if sg + sw > 1.0:   # overlapping transition zones: gas-water contact, sw from the sum of both capillary pressures
                pcgw = pg - pw
                if swatinit is not None:   # the curve is rescaled once more, for a vanishing oil phase (:1229-1235)
                    sw = apply_swatinit(pcgw, sw)
                sw = root(lambda s: pcow(s) + pcgo(1.0 - s) - pcgw, Swl, Swu)
                sg = 1.0 - sw
                po = pg - pcgo(sg)
            so = 1.0 - sw - sg
            # pressure corrections at the saturation end points (accountForScaledSaturations)
            thr = 1.0e-6
            if sw + thr > Swu:
                po = pw + pcow(Swu)
            elif sg + thr > Sgu:
                po = pg - pcgo(Sgu)
            if sg - thr < Sgl:
                pg = po + pcgo(Sgl)
            if sw - thr < Swl:
                pw = po - pcow(Swl)
            return np.array([pw, po, pg, sw, so, sg])

        if acc == 0:     # centre-point method (equilibrateCellCentres :1994-2024)
            pw, po, pg, sw, so, sg = at_depth(zc)
        else:            # horizontal subdivision (equilibrateHorizontal :2027-2070): 2 |N| slices of equal thickness between the
            # cell's mean top and mean bottom depth (subdivisionCentrePoints :1441-1455), weights = thickness
            top, bot = cell_zspan[c]
            if top > bot:
                raise ValueError("negative thickness (inverted top / bottom faces) in cell %d" % c)
            nint = 2 * (-acc)
            h = (bot - top) / nint
            tot = np.zeros(6)
            totfrac, end = 0.0, top
            for q in range(nint):
                start, end = end, top + (q + 1) * h
                tot = tot + at_depth((start + end) / 2) * h
                totfrac += h
            pw, po, pg, sw, so, sg = (tot / totfrac) if totfrac > 0.0 else at_depth(zc)
        out["pw"][c], out["po"][c], out["pg"][c] = pw, po, pg
        out["sw"][c], out["so"][c], out["sg"][c] = sw, so, sg
        out["rs"][c] = rs_func(zc, po, sg)
        out["rv"][c] = rv_func(zc, pg, so) if rv_func is not None else 0.0
        if swatinit is not None:
            out["pcw_scale"][c] = scale[0]
    return out
