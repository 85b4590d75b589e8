"""Host logic of the sub-step control (newton.AdaptiveTimeStepping) against a scripted model: no GPU, no oracle."""
import importlib

import pytest

newton = importlib.import_module("opm-autodiff_amd").newton
DAY = 86400.0


class Scripted:
    """Converges after `need(dt)` Newton iterations; never if need(dt) is None."""

    def __init__(self, need, max_iter=20):
        self.need = need
        self.param = newton.ModelParameters(newton_max_iter=max_iter)
        self.calls = []
        self.advanced = self.rolled_back = 0
        self.ended = []   # problem.endTimeStep(): accepted steps only

    def nonlinear_iteration(self, iteration, dt):
        self.calls.append((iteration, dt))
        rep = newton.SimulatorReportSingle()
        n = self.need(dt)
        if n is not None and iteration >= n:
            rep.converged = True
        else:
            rep.total_newton_iterations = 1
            rep.total_linear_iterations = 5
        return rep

    def advance_time_level(self):
        self.advanced += 1

    def update_failed(self):
        self.rolled_back += 1

    def end_time_step(self, dt):
        self.ended.append(dt)


def test_growth_follows_the_iteration_count_rule():
    mdl = Scripted(lambda dt: 3)
    ts = newton.AdaptiveTimeStepping(mdl, newton.TimeSteppingParameters(initial_dt=DAY, max_dt=10 * DAY))
    for _ in range(10):   # the 10th call sees step 3 converge and hands out the first iteration of step 4
        ts.next_newton_iteration()
    # 3 iterations against a target of 8: factor 1 + 5/8 * 3.2 = 3 = max growth; then the 10-day cap
    assert [h[0] / DAY for h in ts.history] == pytest.approx([1.0, 3.0, 9.0])
    assert ts.dt == 10 * DAY and ts.timesteps_done == 3 and mdl.advanced == 4 and mdl.rolled_back == 0
    assert [d / DAY for d in mdl.ended] == pytest.approx([1.0, 3.0, 9.0])


def test_decay_above_target():
    mdl = Scripted(lambda dt: 12)
    ts = newton.AdaptiveTimeStepping(mdl, newton.TimeSteppingParameters(initial_dt=8 * DAY, max_dt=10 * DAY))
    for _ in range(13):
        ts.next_newton_iteration()
    assert ts.timesteps_done == 1 and ts.dt == pytest.approx(8 * DAY / 1.5)


def test_chop_rolls_back_and_limits_the_next_growth():
    mdl = Scripted(lambda dt: None if dt > 2.5 * DAY else 2, max_iter=4)
    ts = newton.AdaptiveTimeStepping(mdl, newton.TimeSteppingParameters(initial_dt=10 * DAY, max_dt=10 * DAY))
    n = 0
    while ts.timesteps_done < 2:
        ts.next_newton_iteration()
        n += 1
    # 10 d fails after max_iter + 1 = 5 iterations, 3.3 d fails too, 1.089 d converges; next step at most x2 (restart)
    assert [round(h[0] / DAY, 4) for h in ts.history[:3]] == [10.0, 3.3, 1.089]
    assert [h[2] for h in ts.history[:3]] == [False, False, True]
    assert ts.history[3][0] == pytest.approx(2 * 1.089 * DAY)
    # the call that sees step 2 converge already hands out the first iteration of step 3
    assert mdl.rolled_back == 2 and mdl.advanced == 3 and ts.timesteps_failed == 2
    assert len(mdl.ended) == 2 and mdl.ended[0] == pytest.approx(1.089 * DAY)   # failed steps never reach endTimeStep
    assert n == 5 + 5 + 2 + 2 + 1


def test_gives_up_after_max_restarts():
    mdl = Scripted(lambda dt: None, max_iter=1)
    ts = newton.AdaptiveTimeStepping(mdl, newton.TimeSteppingParameters(initial_dt=DAY, max_restarts=3))
    with pytest.raises(newton.TooManyIterations):
        for _ in range(100):
            ts.next_newton_iteration()
    assert mdl.rolled_back == 3


def test_numerical_issue_is_a_failed_step():
    class Bad(Scripted):
        def nonlinear_iteration(self, iteration, dt):
            if dt > DAY:
                raise newton.NumericalIssue("NaN residual found!")
            return super().nonlinear_iteration(iteration, dt)
    mdl = Bad(lambda dt: 1)
    ts = newton.AdaptiveTimeStepping(mdl, newton.TimeSteppingParameters(initial_dt=2 * DAY))
    ts.next_newton_iteration()
    assert ts.history[0] == (2 * DAY, 0, False) and mdl.rolled_back == 1 and ts.dt == pytest.approx(0.66 * DAY)


class ScriptedWithChange(Scripted):
    """... and reports a scripted relative change of the solution after every accepted time step"""

    def __init__(self, need, changes):
        super().__init__(need)
        self.changes = list(changes)

    def relative_change(self):
        return self.changes.pop(0)


def test_pid_control_is_the_formula_of_the_reference():
    """PIDTimeStepControl::computeTimeStepSize (timestepping/TimeStepControl.cpp:127-161), by hand"""
    pid = newton.PIDTimeStepControl(1e-1)
    assert pid.errors == [1e-1] * 3
    # an error above the tolerance: dt * tol / error
    assert pid.compute(4.0, 0.4) == 4.0 * 1e-1 / 0.4
    assert pid.errors == [1e-1, 1e-1, 0.4]
    # below: dt (e1/e2)^kP (tol/e2)^kI (e0^2/e1/e2)^kD with the shifted history
    e0, e1, e2 = 1e-1, 0.4, 1e-3
    want = 2.0 * (e1 / e2) ** 0.075 * (1e-1 / e2) ** 0.175 * (e0 * e0 / e1 / e2) ** 0.01
    assert pid.compute(2.0, 1e-3) == pytest.approx(want, rel=1e-15)
    assert pid.errors == [e0, e1, e2]
    # a step without any change: IEEE arithmetic as in the reference's doubles - the estimate is infinite, the caller's min() decides
    assert pid.compute(1.0, 0.0) == float("inf")


def test_pid_and_newton_iteration_control_takes_the_smaller_estimate():
    """PIDAndIterationCountTimeStepControl (TimeStepControl.cpp:188-208) inside the sub-step loop, Flow's default control"""
    # three Newton iterations per step: the iteration-count estimate alone would grow by 3 (the cap) every time
    mdl = ScriptedWithChange(lambda dt: 3, [0.4, 1e-6, 1e-6])
    ts = newton.AdaptiveTimeStepping(mdl, newton.TimeSteppingParameters(initial_dt=4 * DAY, max_dt=100 * DAY))
    for _ in range(10):
        ts.next_newton_iteration()
    # step 1: error 0.4 > tol 0.1: the PID estimate 4 d * 0.1 / 0.4 = 1 d wins over 12 d; steps 2, 3: tiny errors, the PID
    # estimate is far beyond the growth cap of 3
    assert [h[0] / DAY for h in ts.history] == pytest.approx([4.0, 1.0, 3.0])
    assert ts.dt == pytest.approx(9.0 * DAY) and ts.relative_changes == [0.4, 1e-6, 1e-6]
    # the same model under the iteration-count control alone
    mdl = ScriptedWithChange(lambda dt: 3, [0.4] * 3)
    ts = newton.AdaptiveTimeStepping(mdl, newton.TimeSteppingParameters(initial_dt=4 * DAY, max_dt=100 * DAY, time_step_control="newtoniteration"))
    for _ in range(10):
        ts.next_newton_iteration()
    assert [h[0] / DAY for h in ts.history] == pytest.approx([4.0, 12.0, 36.0]) and ts.relative_changes == []
