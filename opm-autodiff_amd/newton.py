"""Host-side mirror of the reference's Newton loop for the device-resident model, used by bench.py and the tests.

Names, control flow, counters and defaults follow the reference:
  BlackoilModelEbos::nonlinearIteration  opm/simulators/flow/BlackoilModelEbos.hpp:274-392
  BlackoilModelEbos::getReservoirConvergence                                        :767-904
  NonlinearSolverEbos::step / detectOscillations / stabilizeNonlinearUpdate
                                         opm/simulators/flow/NonlinearSolverEbos.hpp:180-236, 278-353
  SimulatorReportSingle counters         opm/simulators/timestepping/SimulatorReport.hpp:29-49
Defaults: BlackoilModelParametersEbos.hpp (ToleranceMb 1e-6, ToleranceCnv 1e-2, ToleranceCnvRelaxed 1,
RelaxedMaxPvFraction 0.03, MaxStrictIter 0, MaxResidualAllowed 1e7), NonlinearSolverEbos.hpp:64-76 (max 20 / min 1
iterations, relaxation "dampen", NewtonMaxRelax 0.5, increment 0.1, relaxRelTol 0.2).

The same loop exists in C++ for Flow-side use: opm-autodiff_amd/host/BlackoilModelHip.hpp.
"""
import time
from dataclasses import dataclass, field


class NumericalIssue(RuntimeError):
    pass


class TooManyIterations(RuntimeError):
    pass


@dataclass
class SimulatorReportSingle:
    assemble_time: float = 0.0
    linear_solve_setup_time: float = 0.0
    linear_solve_time: float = 0.0
    update_time: float = 0.0
    total_linearizations: int = 0
    total_newton_iterations: int = 0
    total_linear_iterations: int = 0
    converged: bool = False

    def __iadd__(self, o):
        for k in ("assemble_time", "linear_solve_setup_time", "linear_solve_time", "update_time", "total_linearizations",
                  "total_newton_iterations", "total_linear_iterations"):
            setattr(self, k, getattr(self, k) + getattr(o, k))
        return self

    def solver_time(self):
        return self.assemble_time + self.linear_solve_setup_time + self.linear_solve_time + self.update_time


@dataclass
class ModelParameters:
    tolerance_mb: float = 1e-6
    tolerance_cnv: float = 1e-2
    tolerance_cnv_relaxed: float = 1.0
    relaxed_max_pv_fraction: float = 0.03
    max_strict_iter: int = 0
    max_residual_allowed: float = 1e7
    use_update_stabilization: bool = True
    newton_max_iter: int = 20
    newton_min_iter: int = 1
    relax_max: float = 0.5
    relax_increment: float = 0.1
    relax_rel_tol: float = 0.2


class BlackoilModelHip:
    """model = capi.HipModel (device context) ; one instance drives one grid on one GPU."""

    def __init__(self, model, param=None, well_model=None):
        """well_model (wells.StandardWells or None): the host-side well equations of BlackoilWellModel - assembled in front of the
        reservoir's linearisation (BlackoilModelEbos::assembleReservoir -> wellModel().assemble, flow/BlackoilModelEbos.hpp:418-428),
        eliminated from the linear system by the device (wells/StandardWell_impl.hpp:1254-1311) and updated with the reservoir"""
        self.m = model
        self.param = param or ModelParameters()
        self.wells = well_model
        self.residual_norms_history = []
        self.current_relaxation = 1.0
        self.last_linear_iterations = 0
        self._well_saved = None

    # -- BlackoilModelEbos::getReservoirConvergence ------------------------------------------------------------
    def get_convergence(self, dt, iteration):
        p = self.param
        c = self.m.convergence(dt, p.tolerance_cnv)
        pv_sum, cnv_err = c[9], c[10]
        cnv_frac = cnv_err / pv_sum
        use_relaxed = cnv_frac < p.relaxed_max_pv_fraction and iteration >= p.max_strict_iter
        tol_cnv = p.tolerance_cnv_relaxed if use_relaxed else p.tolerance_cnv
        CNV, MB = c[11:14], c[14:17]
        converged = True
        for res, tol in list(zip(MB, [p.tolerance_mb] * 3)) + list(zip(CNV, [tol_cnv] * 3)):
            if res != res:
                raise NumericalIssue("NaN residual found!")
            if res > p.max_residual_allowed:
                raise NumericalIssue("Too large residual found!")
            if res < 0.0 or res > tol:
                converged = False
        return converged, list(CNV)

    # -- NonlinearSolverEbos::detectOscillations ---------------------------------------------------------------
    def detect_oscillations(self, it):
        h = self.residual_norms_history
        if it < 2:
            return False
        F0, F1, F2 = h[it], h[it - 1], h[it - 2]
        osc = 0
        for p in range(3):
            d1 = abs((F0[p] - F2[p]) / F0[p])
            d2 = abs((F0[p] - F1[p]) / F0[p])
            osc += (d1 < self.param.relax_rel_tol) and (self.param.relax_rel_tol < d2)
        return osc > 1

    # -- BlackoilModelEbos::nonlinearIteration ------------------------------------------------------------------
    def nonlinear_iteration(self, iteration, dt):
        rep = SimulatorReportSingle()
        if iteration == 0:
            self.residual_norms_history = []
            self.current_relaxation = 1.0
        rep.total_linearizations = 1
        t0 = time.perf_counter()
        wa = None
        if self.wells is not None:
            # wellModel().beginIteration / assemble (wells/BlackoilWellModel_impl.hpp:148-171, 1033-1101): controls, well equations at the
            # present reservoir state, their connection rates as the perforated cells' source terms (computeTotalRatesForDof :496-512)
            # (only the perforated cells' records come back from the device and only their rates go there: updatePerforationIntensiveQuantities
            #  :1606-1630 - opmhip_get_iq_cells / opmhip_set_source_cells; a model without the two calls is asked for / handed whole arrays)
            iq = self.wells.records(self.m)
            if iteration == 0:
                self.wells.calculate_explicit_quantities(iq)  # the completions' pressure differences, constant through the time step (:824-827)
                self.wells.solve_well_equations(iq)          # prepareTimeStep: the wells alone against the frozen reservoir
            self.wells.update_well_controls()
            if hasattr(self.m, "set_source_cells"):
                wa = self.wells.assemble(iq)
                self.m.set_source_cells(wa["cells"], wa["source_cells"], wa["dsource_cells"])
            else:
                wa = self.wells.assemble(iq, iq.shape[0])
                self.m.set_source(wa["source"], wa["dsource"])
        self.m.assemble(dt, iteration, fetch=False)          # assembleReservoir -> linearizeDomain (asynchronous)
        conv, norms = self.get_convergence(dt, iteration)    # synchronises: reads the reduced scalars back
        if wa is not None:
            conv = conv and self.wells.converged(wa["res_well"])   # getWellConvergence (flow/BlackoilModelEbos.hpp:906-912)
        t1 = time.perf_counter()
        # The reference books assembly under assemble_time and the convergence check under update_time
        # (BlackoilModelEbos.hpp:296-297, 311-323).  Both are enqueued back to back here and only the convergence
        # read-back synchronises, so the sum is measured and split by the device-side event times when profiled;
        # wall-clock-wise everything up to the read-back goes to assemble_time.
        rep.assemble_time += t1 - t0
        # "the step is not considered converged until at least minIter iterations is done" (:309-311)
        rep.converged = conv and iteration > self.param.newton_min_iter
        self.residual_norms_history.append(norms)
        if not rep.converged:
            rep.total_newton_iterations = 1
            if wa is not None:
                self.m.wells_apply_residual(wa["wells"], wa["res_well"])        # wellModel().apply(r): r -= C^T D^-1 r_w (:523-527)
                res = self.m.solve_jacobian_system(wells=wa["wells"])           # the operator A - C^T D^-1 B (WellModelMatrixAdapter)
            else:
                res = self.m.solve_jacobian_system()         # solveJacobianSystem: ILU0 setup + BiCGStab
            rep.linear_solve_setup_time += res.t_factor
            rep.linear_solve_time += res.t_solve + res.t_copy
            rep.total_linear_iterations += res.iterations
            self.last_linear_iterations = res.iterations
            if not res.converged:
                # ISTLSolverEbos::checkConvergence (linalg/ISTLSolverEbos.hpp:334-345)
                raise NumericalIssue("Convergence failure for linear solver.")
            t2 = time.perf_counter()
            if self.param.use_update_stabilization and self.detect_oscillations(iteration):
                self.current_relaxation = max(self.current_relaxation - self.param.relax_increment, self.param.relax_max)
            if wa is not None:   # recoverWellSolutionAndUpdateWellState (:1033-1042): x_w = D^-1 (r_w - B x), same relaxation
                self.wells.update(self.m.wells_recover_solution(wa["wells"], wa["res_well"]), self.current_relaxation)
            self.m.update(None, self.current_relaxation)      # stabilizeNonlinearUpdate (dampen) + updateSolution
            rep.update_time += time.perf_counter() - t2
        return rep

    # -- NonlinearSolverEbos::step --------------------------------------------------------------------------------
    def step(self, dt):
        report = SimulatorReportSingle()
        iteration = 0
        converged = False
        while True:
            it_rep = self.nonlinear_iteration(iteration, dt)
            report += it_rep
            converged = it_rep.converged
            iteration += 1
            if not ((not converged and iteration <= self.param.newton_max_iter) or iteration <= self.param.newton_min_iter):
                break
        if not converged:
            raise TooManyIterations("Failed to complete a time step within %d iterations." % self.param.newton_max_iter)
        report.converged = True
        return report

    # -- FvBaseDiscretization::advanceTimeLevel / updateFailed ---------------------------------------------------
    def advance_time_level(self):
        self.m.advance_time_level()
        if self.wells is not None:
            self._well_saved = self.wells.state()      # the well state of the last accepted step (WellState copy of a failed step's restart)

    def update_failed(self):
        self.m.update_failed()
        if self.wells is not None and self._well_saved is not None:
            self.wells.set_state(self._well_saved)

    def relative_change(self):
        """BlackoilModelEbos::relativeChange (flow/BlackoilModelEbos.hpp:431-510); None where the model object has none"""
        return self.m.relative_change() if hasattr(self.m, "relative_change") else None

    # -- EclProblem::beginTimeStep (ebos/eclproblem.hh:1042-1075): DRSDT / DRVDT caps of a step of size dt, minimum pressure --
    def begin_time_step(self, dt):
        if hasattr(self.m, "begin_time_step"):
            self.m.begin_time_step(dt)

    # -- EclProblem::endTimeStep (ebos/eclproblem.hh:1101-1135): the drift of the accepted step ------------------
    def end_time_step(self, dt):
        self.m.end_time_step(dt)


@dataclass
class TimeSteppingParameters:
    """AdaptiveTimeSteppingEbos.hpp:112-200 defaults (FlowTimeSteppingParameters)"""
    restart_factor: float = 0.33          # SolverRestartFactor
    growth_factor: float = 2.0            # SolverGrowthFactor (first step after a chop)
    max_growth: float = 3.0               # SolverMaxGrowth
    max_restarts: int = 10                # SolverMaxRestarts
    target_newton_iterations: int = 8     # TimeStepControlTargetNewtonIterations
    decay_damping: float = 1.0            # TimeStepControlDecayDampingFactor
    growth_damping: float = 3.2           # TimeStepControlGrowthDampingFactor
    time_step_control: str = "pid+newtoniteration"   # TimeStepControl (AdaptiveTimeSteppingEbos.hpp:168-170); "newtoniteration": without the PID part
    time_step_control_tolerance: float = 1e-1        # TimeStepControlTolerance (:172-174)
    min_time_step_based_on_newton_iterations: float = 0.0   # MinTimeStepBasedOnNewtonIterations (:215-217)
    initial_dt: float = 86400.0           # InitialTimeStepInDays
    max_dt: float = 365.0 * 86400.0       # SolverMaxTimeStepInDays (bench.py passes its report-step length)


class PIDTimeStepControl:
    """PIDTimeStepControl::computeTimeStepSize (timestepping/TimeStepControl.cpp:117-161): the last three relative changes
    of the solution steer dt towards the tolerance; IEEE arithmetic as in C++ (a step without any change gives inf, which the
    caller's min() with the iteration-count estimate and the growth cap then replaces)"""

    def __init__(self, tol):
        self.tol = float(tol)
        self.errors = [self.tol] * 3

    def compute(self, dt, error):
        import numpy as np
        e = self.errors
        e[0], e[1] = e[1], e[2]
        e[2] = float(error)
        tol = self.tol
        if error > tol:
            return dt * tol / error
        kP, kI, kD = 0.075, 0.175, 0.01          # "values taking from turek time stepping paper"
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            e0, e1, e2 = (np.float64(v) for v in e)
            return float(dt * np.power(e1 / e2, kP) * np.power(tol / e2, kI) * np.power(e0 * e0 / e1 / e2, kD))


class AdaptiveTimeStepping:
    """The sub-stepping loop of AdaptiveTimeSteppingEbos::step (opm/simulators/timestepping/AdaptiveTimeSteppingEbos.hpp
    :283-520) over any model object with nonlinear_iteration(iteration, dt) -> report, advance_time_level(),
    update_failed(), end_time_step(dt) and param.newton_max_iter: a failed time step (TooManyIterations, NumericalIssue) is rolled back and
    retried with dt * restart_factor; an accepted one sets the next dt with the Newton-iteration-count rule of
    PIDAndIterationCountTimeStepControl::computeTimeStepSize (timestepping/TimeStepControl.cpp:188-208): the smaller of the
    PID estimate - from the model's relative_change(), BlackoilModelEbos::relativeChange on the device - and the
    Newton-iteration-count estimate; a model object without relative_change() gets the iteration-count estimate alone.
    Hands out Newton iterations one at a time so that a benchmark can count and time them."""

    def __init__(self, model, param=None):
        self.model = model
        self.p = param or TimeSteppingParameters()
        self.dt = self.p.initial_dt
        self.iteration = 0
        self.restarts = 0
        self.timesteps_done = 0
        self.timesteps_failed = 0
        self.time = 0.0
        self.report = SimulatorReportSingle()
        self.history = []          # (dt, newton iterations, accepted)
        self.pid = PIDTimeStepControl(self.p.time_step_control_tolerance) if self.p.time_step_control == "pid+newtoniteration" else None
        self.relative_changes = []  # what the PID control saw, per accepted time step
        self.t_end = None           # end of the report step under way (advance_report_step): sub-steps are cut to land on it
        self.on_accept = None       # callable(dt): after every accepted sub-step, with the model still in that step's converged state (Flow's
                                    # per-step reporting hooks in AdaptiveTimeSteppingEbos::step; here: tests that integrate well rates)

    def _next_dt(self, dt, iterations):
        p = self.p
        tgt = p.target_newton_iterations
        if iterations > tgt:
            est = max(dt / (1.0 + (iterations - tgt) / tgt * p.decay_damping), p.min_time_step_based_on_newton_iterations)
        else:
            est = dt * (1.0 + (tgt - iterations) / tgt * p.growth_damping)
        if self.pid is not None:
            err = self.model.relative_change() if hasattr(self.model, "relative_change") else None
            if err is not None:
                self.relative_changes.append(err)
                est_pid = self.pid.compute(dt, err)
                est = est if est < est_pid else est_pid      # std::min(dtEstimatePID, dtEstimateIter)
        est = min(est, p.max_growth * dt)                      # AdaptiveTimeSteppingEbos.hpp:404-406
        if self.restarts > 0:                                  # :408-411
            est = min(p.growth_factor * dt, est)
            self.restarts = 0
        return min(est, p.max_dt)

    def next_newton_iteration(self):
        """Runs nonlinear iterations until one of them actually solved a system; returns its report."""
        while True:
            if self.t_end is not None and self.iteration == 0:
                if self.time >= self.t_end - 1e-9 * max(1.0, abs(self.t_end)):
                    return None                                   # the report step is complete
                self.dt = min(self.dt, self.t_end - self.time)    # the sub-step timer never runs past the report step (AdaptiveSimulatorTimer)
            if self.iteration == 0 and self.restarts == 0:
                self.model.advance_time_level()
            if self.iteration == 0 and hasattr(self.model, "begin_time_step"):
                self.model.begin_time_step(self.dt)      # also in front of every retry of a chopped step
            failed = False
            try:
                rep = self.model.nonlinear_iteration(self.iteration, self.dt)
            except NumericalIssue:
                failed, rep = True, None
            if rep is not None:
                self.report += rep
                self.iteration += 1
                if rep.converged:
                    self.model.end_time_step(self.dt)   # problem.endTimeStep() of the accepted sub-step
                    if self.on_accept is not None:
                        self.on_accept(self.dt)
                    self.history.append((self.dt, self.iteration - 1, True))
                    self.time += self.dt
                    self.dt = self._next_dt(self.dt, self.iteration - 1)
                    self.timesteps_done += 1
                    self.iteration = 0
                    continue   # the converged check cost an assembly; it is inside the timed region like in Flow
                failed = self.iteration > self.model.param.newton_max_iter
            if failed:
                self.history.append((self.dt, self.iteration, False))
                self.timesteps_failed += 1
                self.restarts += 1
                if self.restarts > self.p.max_restarts:
                    raise TooManyIterations("time step chopped %d times in a row" % self.p.max_restarts)
                self.model.update_failed()
                self.dt *= self.p.restart_factor
                self.iteration = 0
                if rep is None:
                    continue
            if rep is not None and rep.total_newton_iterations:
                return rep

    def advance_report_step(self, length):
        """AdaptiveTimeSteppingEbos::step over one report step of the SCHEDULE section (TSTEP): sub-steps under the time-step control until
        its end, the last one cut to land on it (:283-520).  -> the reports of its Newton iterations"""
        self.t_end = self.time + float(length)
        reports = []
        try:
            while True:
                r = self.next_newton_iteration()
                if r is None:
                    return reports
                reports.append(r)
        finally:
            self.t_end = None
