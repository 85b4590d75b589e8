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


class CellRecords:
    """the intensive-quantity records of some cells (model.iq_cells(cells)), addressed by cell id like the full array model.iq() returns"""

    def __init__(self, cells, records):
        self.row = {int(c): i for i, c in enumerate(cells)}
        self.rec = records

    def rows(self, cells):
        return self.rec[[self.row[int(c)] for c in cells]]


def _rows(iq, cells):
    return iq.rows(cells) if isinstance(iq, CellRecords) else np.asarray(iq)[np.asarray(cells, int)]


class StandardWells:
    """All wells at once: every step below is one pass of array arithmetic over the perforations (nperf x 5: value, d/dSw, d/dp, d/dX of the
    perforated cell, d/dbhp), per-well sums in the order of the perforations."""

    def __init__(self, wells, cell_depth):
        self.wells = list(wells)
        self.nw = len(self.wells)
        self.vp = np.concatenate([[0], np.cumsum([len(w.cells) for w in self.wells])]).astype(np.int32)
        self.cells = np.concatenate([w.cells for w in self.wells]).astype(np.int32)
        self.tw = np.concatenate([w.tw for w in self.wells])
        self.well_of_perf = np.repeat(np.arange(self.nw), np.diff(self.vp))
        # the cells the model is asked about / told about: every perforated cell once
        self.ucells, self.perf_row = np.unique(self.cells, return_inverse=True)
        self.ucells = self.ucells.astype(np.int32)
        self.depth = np.asarray(cell_depth, float)
        self.ref_depth_of_perf = np.array([w.ref_depth for w in self.wells])[self.well_of_perf]
        self.x = np.zeros((self.nw, 4))            # q_oil, q_water, q_gas (into the reservoir), bhp
        self.head = None                           # per perforation: pressure in the well bore there - bhp (calculate_explicit_quantities)
        self.initialised = False

    def records(self, model):
        """the perforated cells' intensive quantities from the model - the perforated cells only (updatePerforationIntensiveQuantities,
        wells/BlackoilWellModel_impl.hpp:1606-1630); a model without iq_cells hands over its whole array"""
        if hasattr(model, "iq_cells"):
            return CellRecords(self.ucells, model.iq_cells(self.ucells))
        return model.iq()

    def calculate_explicit_quantities(self, iq):
        """The pressure differences between the reference depth and the completions, once per time step from the state it starts with and
        constant through its Newton iterations (BlackoilWellModel::assemble, iteration 0: calculateExplicitQuantities ->
        StandardWell::computeWellConnectionPressures, wells/BlackoilWellModel_impl.hpp:824-827, wells/StandardWell_impl.hpp:1198-1245).
        Minimal form: the column between the reference depth and a completion weighs what the oil of the completion's cell weighs (the
        reference averages the well-bore mixture's phase densities segment by segment, StandardWellGeneric::computeConnectionPressureDelta)."""
        q = _rows(iq, self.cells)
        self.head = q[:, F_RHO + PH_O, 0] * GRAVITY * (self.depth[self.cells] - self.ref_depth_of_perf)

    # ---- connection rates of every perforation with their derivatives: (nperf, 3 components, 1 + 3 cell variables + bhp) ----------------
    def _perf_rates(self, iq, bhp):
        """bhp: per well"""
        if self.head is None:
            self.calculate_explicit_quantities(iq)
        q = _rows(iq, self.cells)
        n = len(self.cells)
        ad = lambda f: np.concatenate([q[:, f, :], np.zeros((n, 1))], axis=1)     # value, d/dSw, d/dp, d/dX of the cell, d/dbhp

        def mul(a, b):
            out = np.empty_like(a)
            out[:, 0] = a[:, 0] * b[:, 0]
            out[:, 1:] = a[:, :1] * b[:, 1:] + b[:, :1] * a[:, 1:]
            return out
        dd = ad(F_P + PH_O)
        dd[:, 0] -= np.asarray(bhp)[self.well_of_perf] + self.head      # the head between the reference depth and the completion: explicit, see above
        dd[:, 4] = -1.0                                                 # d(drawdown)/d(bhp)
        b = [ad(F_B + ph) for ph in range(3)]
        mob = [ad(F_MOB + ph) for ph in range(3)]
        rs = ad(F_RS)
        tw = self.tw[:, None]
        out = np.zeros((n, 3, 5))
        producer = np.array([w.producer for w in self.wells])[self.well_of_perf]
        # producing perforations: phase rate = -Tw mob drawdown (reservoir volumes, out of the cell), surface volumes through 1/B, dissolved gas
        # with the oil
        flows = producer & (dd[:, 0] > 0.0)
        if flows.any():
            surf = [mul(b[ph], -tw * mul(mob[ph], dd)) for ph in range(3)]
            out[flows, OIL] = surf[PH_O][flows]
            out[flows, WATER] = surf[PH_W][flows]
            out[flows, GAS] = (surf[PH_G] + mul(rs, surf[PH_O]))[flows]
        # injecting perforations: total mobility, the injected phase's 1/B
        inj = ~producer & (dd[:, 0] < 0.0)
        if inj.any():
            tot = mob[0] + mob[1] + mob[2]
            vol = -tw * mul(tot, dd)
            for name, ph, comp in (("gas", PH_G, GAS), ("water", PH_W, WATER), ("oil", PH_O, OIL)):
                sel = inj & np.array([w.inj_phase == name for w in self.wells])[self.well_of_perf]
                if sel.any():
                    out[sel, comp] = mul(b[ph], vol)[sel]
        # a perforation that would flow against the well's kind is closed (no crossflow)
        return out

    def _control_rows(self):
        """(residual, d/d(q_o, q_w, q_g, bhp)) of every well's control equation"""
        r, g = np.zeros(self.nw), np.zeros((self.nw, 4))
        for k, (w, x) in enumerate(zip(self.wells, self.x)):
            if w.control[0] == "bhp":
                r[k], g[k, 3] = x[3] - w.control[1], 1.0
            else:
                comp, target = w.control[1], w.control[2]
                r[k], g[k, comp] = x[comp] - (-1.0 if w.producer else 1.0) * target, 1.0
        return r, g

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

    def set_rate_target(self, k, target):
        """a WCONPROD / WCONINJE record at a report step (ScheduleEvents::PRODUCTION_UPDATE / INJECTION_UPDATE, wells/WellState.hpp:57): the
        well is under the deck's control mode again with the new target and its rate unknown starts there (updateWellStateWithTarget from
        prepareTimeStep, wells/BlackoilWellModel_impl.hpp:1426-1431); update_well_controls sends it back to its BHP limit if it cannot hold it"""
        w = self.wells[k]
        w.rate_control = ("rate", w.rate_control[1], float(target))
        w.control = w.rate_control
        old = self.x[k, w.rate_control[1]]
        new = (-1.0 if w.producer else 1.0) * float(target)
        if old != 0.0:
            self.x[k, :3] *= new / old           # the other components keep their ratio to the controlled one
        else:
            self.x[k, w.rate_control[1]] = new

    def _assemble_wells(self, iq):
        """residuals r_w (nw x 4), D (nw x 4 x 4), per perforation B (4 x 3: d r_w / d cell variables), C (4 x 3: C^T = d r_cell / d x_w),
        source (3) and dsource (3 x 3)"""
        pr = self._perf_rates(iq, self.x[:, 3])
        nperf = len(self.cells)
        r = np.zeros((self.nw, 4))
        D = np.zeros((self.nw, 4, 4))
        seg = self.vp[:-1]
        r[:, :3] = self.x[:, :3] - np.add.reduceat(pr[:, :, 0], seg, axis=0)          # surface rate - sum of the connection rates
        D[:, [0, 1, 2], [0, 1, 2]] = 1.0
        D[:, :3, 3] = -np.add.reduceat(pr[:, :, 4], seg, axis=0)
        r[:, 3], D[:, 3, :] = self._control_rows()
        # a well none of whose completions flows (its bottom-hole pressure on the wrong side of every completion's pressure) has no rate that
        # answers to its bottom-hole pressure: under a rate target its equations would be singular.  It keeps its bottom-hole pressure for
        # this iteration and its rates go to zero (the reference takes such a well out of operation: checkWellOperability,
        # wells/BlackoilWellModel_impl.hpp:1421-1423)
        for k in np.flatnonzero(np.all(D[:, :3, 3] == 0.0, axis=1) & (D[:, 3, 3] == 0.0)):
            r[k, 3], D[k, 3, :] = 0.0, (0.0, 0.0, 0.0, 1.0)
        B, C = np.zeros((nperf, 4, 3)), np.zeros((nperf, 4, 3))
        B[:, :3, :] = -pr[:, :, 1:4]                 # d r_w[c] / d (Sw, p, X) of the perforated cell
        C[:, 3, :] = -pr[:, :, 4]                    # d r_cell[c] / d bhp = - d(connection rate) / d bhp
        return r, D, B, C, pr[:, :, 0], pr[:, :, 1:4]

    def solve_well_equations(self, iq, iterations=20):
        """the well equations alone at a frozen reservoir state (StandardWell::solveWellEqUntilConverged / prepareTimeStep): Newton on the
        4 unknowns of every well"""
        if not self.initialised:
            q = _rows(iq, [w.cells[0] for w in self.wells])
            self.x[:, 3] = q[:, F_P + PH_O, 0] + np.where([w.producer for w in self.wells], -1e5, 1e5)
        active = np.ones(self.nw, bool)
        for _ in range(iterations):
            r, D, *_ = self._assemble_wells(iq)
            dx = np.linalg.solve(D, r[:, :, None])[:, :, 0]
            self.x[active] -= dx[active]
            small = (np.abs(dx[:, :3]).max(axis=1) <= 1e-12 * np.maximum(1e-6, np.abs(self.x[:, :3]).max(axis=1))) & (np.abs(dx[:, 3]) <= 1e-3)
            active &= ~small
            if not active.any():
                break
        self.initialised = True

    def assemble(self, iq, ncells=None):
        """-> dict(wells for the C-ABI, res_well, cells / source_cells / dsource_cells: the connection rates per perforated cell, each named once):
        BlackoilWellModel::assemble at the present reservoir and well state.  ncells: also `source` / `dsource` as arrays over the whole grid
        (opmhip_set_source's form)."""
        rw, D, Bn, Cn, src, dsrc = self._assemble_wells(iq)
        Dinv = np.linalg.inv(D)
        nu = len(self.ucells)
        source_cells, dsource_cells = np.zeros((nu, 3)), np.zeros((nu, 3, 3))
        np.add.at(source_cells, self.perf_row, src)
        np.add.at(dsource_cells, self.perf_row, dsrc)
        W = dict(numWells=self.nw, val_pointers=self.vp, Ccols=self.cells, Bcols=self.cells.copy(),
                 Cnnzs=np.ascontiguousarray(Cn.reshape(-1)), Bnnzs=np.ascontiguousarray(Bn.reshape(-1)), Dnnzs=np.ascontiguousarray(Dinv.reshape(-1)))
        out = dict(wells=W, res_well=np.ascontiguousarray(rw.reshape(-1)), cells=self.ucells, source_cells=np.ascontiguousarray(source_cells.reshape(-1)),
                   dsource_cells=np.ascontiguousarray(dsource_cells.reshape(-1)))
        if ncells is not None:
            source, dsource = np.zeros((ncells, 3)), np.zeros((ncells, 3, 3))
            source[self.ucells] = source_cells
            dsource[self.ucells] = dsource_cells
            out["source"], out["dsource"] = np.ascontiguousarray(source.reshape(-1)), np.ascontiguousarray(dsource.reshape(-1))
        return out

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
