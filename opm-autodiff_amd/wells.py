"""Host-side standard wells for the device-resident Newton iteration: the well equations a deck's SCHEDULE section asks for, assembled on
the host (as in the reference: only the reservoir part of the linearisation and the Schur-complement operator are accelerated) and handed
to the device as the blocks bda::WellContributions carries - B, C (4 x 3 per perforation), D^-1 (4 x 4 per well).

What of the reference this restates, minimally (vertical wells, rate or BHP control, no crossflow, no well storage term):
  wells/StandardWell_impl.hpp: computePerfRate (:195-420; producing perforations: phase rate = -Tw mob drawdown, surface rate through
      1/B, dissolved gas with the oil; injecting perforations: total mobility, the injected phase's 1/B), assembleWellEqWithoutIteration
      (:516-640: mass-balance equation per component = surface rate - sum of connection rates, one control equation), apply (:1254-1296),
      recoverSolutionWell (:1298-1311), updateWellState (full Newton update of the well unknowns);
  wells/BlackoilWellModel_impl.hpp: assemble / computeTotalRatesForDof (:496-512: the connection rates as source terms of the perforated
      cells), updateWellControls (:rate control <-> BHP limit), getWellConvergence.
Well unknowns per well (dim_wells = 4, bda/WellContributions.cpp:215-225): the surface rates of oil, water and gas INTO the reservoir
(production negative) and the bottom-hole pressure - a linear re-parametrisation of the reference's (WQTotal, WFrac, GFrac, BHP).

Coupled system per Newton iteration (x = reservoir update, x_w = well update; the reference's sign convention: new = old - update):
    [ A   C^T ] [ x   ]   [ r   ]        A : reservoir Jacobian with -d(connection rates)/d(cell variables) on the perforated cells' diagonal
    [ B   D   ] [ x_w ] = [ r_w ]            blocks (opmhip_set_source's dsource), C^T = d r_cell / d x_w, B = d r_w / d x_cell, D = d r_w / d x_w
The device solves (A - C^T D^-1 B) x = r - C^T D^-1 r_w (opmhip_wells_apply_residual, opmhip_solve_system with the wells in operator form)
and returns x_w = D^-1 (r_w - B x) (opmhip_wells_recover_solution).

The model object (capi.HipModel, or the oracle behind the same method names) supplies iq(): the cached intensive quantities of the cells.
"""
import numpy as np

OIL, WATER, GAS = 0, 1, 2          # equation / component order of the blocks (csrc/assemble.hip: EQ_OIL, EQ_WATER, EQ_GAS)
PH_W, PH_O, PH_G = 0, 1, 2         # phase order of the intensive-quantity record (opmhip_get_iq)
F_S, F_P, F_B, F_MOB, F_RHO, F_RS = 0, 3, 6, 9, 12, 15
GRAVITY = 9.80665


def peaceman_factor(perm, dx, dy, dz, diameter, skin=0.0):
    """connection transmissibility factor of a vertical well in an isotropic cell (opm-common's Connection / Peaceman:
    2 pi K h / (ln(r0 / rw) + S), r0 = 0.28 sqrt(dx^2 + dy^2) / 2)"""
    r0 = 0.28 * np.sqrt(dx * dx + dy * dy) / 2.0
    return 2.0 * np.pi * perm * dz / (np.log(r0 / (0.5 * diameter)) + skin)


class Well:
    """name; cells: perforated cells (natural order); tw: connection transmissibility factors; ref_depth; producer or injector of `phase`;
    control: ("rate", component, target > 0 surface m^3/s) or ("bhp", pascal); bhp_limit: lower (producer) / upper (injector) limit"""

    def __init__(self, name, cells, tw, ref_depth, producer, control, bhp_limit, inj_phase=None):
        self.name, self.cells, self.tw = name, np.asarray(cells, np.int32), np.asarray(tw, float)
        self.ref_depth, self.producer, self.inj_phase = float(ref_depth), bool(producer), inj_phase
        self.control, self.bhp_limit = control, float(bhp_limit)
        self.rate_control = control       # the deck's rate target, kept for switching back from the BHP limit


class StandardWells:
    def __init__(self, wells, cell_depth):
        self.wells = list(wells)
        self.nw = len(self.wells)
        self.vp = np.concatenate([[0], np.cumsum([len(w.cells) for w in self.wells])]).astype(np.int32)
        self.cells = np.concatenate([w.cells for w in self.wells]).astype(np.int32)
        self.depth = np.asarray(cell_depth, float)
        self.x = np.zeros((self.nw, 4))            # q_oil, q_water, q_gas (into the reservoir), bhp
        self.initialised = False

    # ---- connection rates of one well with their derivatives: (nperf, 3 components, 1 + 3 cell variables + bhp) ----------------------
    def _perf_rates(self, w, iq, bhp):
        out = np.zeros((len(w.cells), 3, 5))
        for j, (c, tw) in enumerate(zip(w.cells, w.tw)):
            q = iq[c]
            # 5-vectors: value, d/dSw, d/dp, d/dX of the cell, d/dbhp
            ad = lambda f: np.concatenate([q[f], [0.0]])
            cst = lambda v: np.array([v, 0.0, 0.0, 0.0, 0.0])
            mul = lambda a, b: np.concatenate([[a[0] * b[0]], a[0] * b[1:] + b[0] * a[1:]])
            p = ad(F_P + PH_O)
            rho_avg = q[F_RHO + PH_O][0]                      # hydrostatic head between the reference depth and the perforation (frozen density)
            head = rho_avg * GRAVITY * (self.depth[c] - w.ref_depth)
            dd = p - cst(bhp + head)
            dd[4] = -1.0                                       # d(drawdown)/d(bhp)
            b = [ad(F_B + ph) for ph in range(3)]
            mob = [ad(F_MOB + ph) for ph in range(3)]
            rs = ad(F_RS)
            if dd[0] > 0.0 and w.producer:
                vol = [-tw * mul(mob[ph], dd) for ph in range(3)]          # reservoir volumes per second, out of the cell
                surf = [mul(b[ph], vol[ph]) for ph in range(3)]
                out[j, OIL] = surf[PH_O]
                out[j, WATER] = surf[PH_W]
                out[j, GAS] = surf[PH_G] + mul(rs, surf[PH_O])
            elif dd[0] < 0.0 and not w.producer:
                tot = mob[0] + mob[1] + mob[2]
                ph = {"gas": PH_G, "water": PH_W, "oil": PH_O}[w.inj_phase]
                comp = {"gas": GAS, "water": WATER, "oil": OIL}[w.inj_phase]
                out[j, comp] = mul(b[ph], -tw * mul(tot, dd))
            # else: a perforation that would flow against the well's kind is closed (no crossflow)
        return out

    def _control_row(self, w, x):
        """(residual, d/d(q_o, q_w, q_g, bhp)) of the control equation"""
        kind = w.control[0]
        if kind == "bhp":
            return x[3] - w.control[1], np.array([0.0, 0.0, 0.0, 1.0])
        comp, target = w.control[1], w.control[2]
        sign = -1.0 if w.producer else 1.0
        g = np.zeros(4)
        g[comp] = 1.0
        return x[comp] - sign * target, g

    def update_well_controls(self):
        """BlackoilWellModel::updateWellControls, for the two controls a well here has: a rate target whose BHP leaves its limit goes under
        BHP control; under BHP control it returns to the rate target once the rate exceeds it"""
        for w, x in zip(self.wells, self.x):
            sign = -1.0 if w.producer else 1.0
            if w.control[0] == "rate":
                if (w.producer and x[3] < w.bhp_limit) or (not w.producer and x[3] > w.bhp_limit):
                    w.control = ("bhp", w.bhp_limit)
                    x[3] = w.bhp_limit
            else:
                comp, target = w.rate_control[1], w.rate_control[2]
                if sign * x[comp] > target:
                    w.control = w.rate_control

    def _assemble_well(self, k, iq):
        """residual r_w (4), D (4 x 4), per perforation B (4 x 3: d r_w / d cell variables), C (4 x 3: C^T = d r_cell / d x_w),
        source (3) and dsource (3 x 3)"""
        w, x = self.wells[k], self.x[k]
        pr = self._perf_rates(w, iq, x[3])
        r = np.zeros(4)
        D = np.zeros((4, 4))
        np_ = len(w.cells)
        B, C = np.zeros((np_, 4, 3)), np.zeros((np_, 4, 3))
        src, dsrc = np.zeros((np_, 3)), np.zeros((np_, 3, 3))
        for c in range(3):
            r[c] = x[c] - pr[:, c, 0].sum()
            D[c, c] = 1.0
            D[c, 3] = -pr[:, c, 4].sum()
        r[3], D[3] = self._control_row(w, x)
        for j in range(np_):
            for c in range(3):
                B[j, c, :] = -pr[j, c, 1:4]               # d r_w[c] / d (Sw, p, X) of the perforated cell
                C[j, 3, c] = -pr[j, c, 4]                 # d r_cell[c] / d bhp = - d(connection rate) / d bhp
                src[j, c] = pr[j, c, 0]
                dsrc[j, c, :] = pr[j, c, 1:4]
        return r, D, B, C, src, dsrc

    def solve_well_equations(self, iq, iterations=20):
        """the well equations alone at a frozen reservoir state (StandardWell::solveWellEqUntilConverged / prepareTimeStep): Newton on the
        4 unknowns of every well"""
        for k, w in enumerate(self.wells):
            if not self.initialised:
                c0 = w.cells[0]
                self.x[k, 3] = iq[c0][F_P + PH_O][0] + (-1e5 if w.producer else 1e5)
            for _ in range(iterations):
                r, D, *_ = self._assemble_well(k, iq)
                dx = np.linalg.solve(D, r)
                self.x[k] -= dx
                if np.abs(dx[:3]).max() <= 1e-12 * max(1e-6, np.abs(self.x[k, :3]).max()) and abs(dx[3]) <= 1e-3:
                    break
        self.initialised = True

    def assemble(self, iq, ncells):
        """-> dict(wells for the C-ABI, res_well, source, dsource): BlackoilWellModel::assemble at the present reservoir and well state"""
        nperf = len(self.cells)
        Bn, Cn = np.zeros((nperf, 4, 3)), np.zeros((nperf, 4, 3))
        Dinv = np.zeros((self.nw, 4, 4))
        rw = np.zeros((self.nw, 4))
        source, dsource = np.zeros((ncells, 3)), np.zeros((ncells, 3, 3))
        for k, w in enumerate(self.wells):
            r, D, B, C, src, dsrc = self._assemble_well(k, iq)
            p0, p1 = self.vp[k], self.vp[k + 1]
            Bn[p0:p1], Cn[p0:p1] = B, C
            Dinv[k] = np.linalg.inv(D)
            rw[k] = r
            for j, c in enumerate(w.cells):
                source[c] += src[j]
                dsource[c] += dsrc[j]
        W = dict(numWells=self.nw, val_pointers=self.vp, Ccols=self.cells, Bcols=self.cells.copy(),
                 Cnnzs=np.ascontiguousarray(Cn.reshape(-1)), Bnnzs=np.ascontiguousarray(Bn.reshape(-1)), Dnnzs=np.ascontiguousarray(Dinv.reshape(-1)))
        return dict(wells=W, res_well=np.ascontiguousarray(rw.reshape(-1)), source=np.ascontiguousarray(source.reshape(-1)),
                    dsource=np.ascontiguousarray(dsource.reshape(-1)))

    def update(self, xw, relax=1.0):
        """updateWellState: the well unknowns follow their Newton update (x_w = D^-1 (r_w - B x) from the device)"""
        self.x -= relax * np.asarray(xw, float).reshape(self.nw, 4)

    def converged(self, res_well, tol_rate=1e-7, tol_bhp=1.0):
        """getWellConvergence: component equations relative to the largest rate of the well, control equation in its own unit"""
        rw = np.asarray(res_well, float).reshape(self.nw, 4)
        for k, w in enumerate(self.wells):
            scale = max(np.abs(self.x[k, :3]).max(), 1e-9)
            if np.abs(rw[k, :3]).max() > tol_rate * scale:
                return False
            ctl = abs(rw[k, 3])
            if ctl > (tol_bhp if w.control[0] == "bhp" else tol_rate * scale):
                return False
        return True

    def state(self):
        return self.x.copy(), [w.control for w in self.wells]

    def set_state(self, st):
        self.x = st[0].copy()
        for w, c in zip(self.wells, st[1]):
            w.control = c
