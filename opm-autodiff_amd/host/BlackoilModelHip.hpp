// C++ host mirror of the reference's Newton iteration for the device-resident model: same method names, control flow,
// counters and defaults as
//   Opm::BlackoilModelEbos::nonlinearIteration / assembleReservoir / getReservoirConvergence / solveJacobianSystem /
//   updateSolution                       opm/simulators/flow/BlackoilModelEbos.hpp:274-392, 418-428, 767-904, 523-563
//   Opm::NonlinearSolverEbos::step / detectOscillations / stabilizeNonlinearUpdate ("dampen")
//                                        opm/simulators/flow/NonlinearSolverEbos.hpp:180-236, 278-353
//   Opm::SimulatorReportSingle           opm/simulators/timestepping/SimulatorReport.hpp:29-49
// Everything heavy happens behind the C-ABI (include/opmhip.h); this class owns no device memory.
#pragma once
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/opmhip.h"

namespace Opm {

struct NumericalIssue : std::runtime_error { using std::runtime_error::runtime_error; };
struct TooManyIterations : std::runtime_error { using std::runtime_error::runtime_error; };

struct SimulatorReportSingle {
    double assemble_time = 0.0, linear_solve_setup_time = 0.0, linear_solve_time = 0.0, update_time = 0.0;
    unsigned total_linearizations = 0, total_newton_iterations = 0, total_linear_iterations = 0;
    bool converged = false;
    void operator+=(const SimulatorReportSingle& o) {
        assemble_time += o.assemble_time; linear_solve_setup_time += o.linear_solve_setup_time;
        linear_solve_time += o.linear_solve_time; update_time += o.update_time;
        total_linearizations += o.total_linearizations; total_newton_iterations += o.total_newton_iterations;
        total_linear_iterations += o.total_linear_iterations;
    }
};

struct ModelParametersHip {  // BlackoilModelParametersEbos.hpp / NonlinearSolverEbos.hpp:64-76 defaults
    double tolerance_mb_ = 1e-6, tolerance_cnv_ = 1e-2, tolerance_cnv_relaxed_ = 1.0, relaxed_max_pv_fraction_ = 0.03;
    int max_strict_iter_ = 0;
    double max_residual_allowed_ = 1e7;
    bool use_update_stabilization_ = true;
    int newton_max_iter_ = 20, newton_min_iter_ = 1;
    double relax_max_ = 0.5, relax_increment_ = 0.1, relax_rel_tol_ = 0.2;
};

class BlackoilModelHip {
    opmhip_ctx* ctx_;
    ModelParametersHip param_;
    std::vector<std::array<double, 3>> residual_norms_history_;
    double current_relaxation_ = 1.0;
    int linear_iterations_last_solve_ = 0;
    int N_, nnz_;

    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    void check(int rc, const char* what) const {
        if (rc != OPMHIP_SUCCESS) throw std::logic_error(std::string(what) + ": " + opmhip_last_error(ctx_));
    }

public:
    /// ctx: a context on which set_pattern / set_fluid / set_static / set_state have been called
    BlackoilModelHip(opmhip_ctx* ctx, int Nb, int nnzb, const ModelParametersHip& p = ModelParametersHip())
        : ctx_(ctx), param_(p), N_(3 * Nb), nnz_(9 * nnzb) {}
    const ModelParametersHip& param() const { return param_; }
    int linearIterationsLastSolve() const { return linear_iterations_last_solve_; }

    SimulatorReportSingle assembleReservoir(double dt, int iterationIdx) {
        SimulatorReportSingle r;
        check(opmhip_assemble(ctx_, dt, iterationIdx, nullptr, nullptr), "assembleReservoir");
        return r;
    }

    /// getReservoirConvergence: returns converged?, fills the CNV norms used by the oscillation detector
    bool getConvergence(double dt, int iteration, std::array<double, 3>& residual_norms) {
        double c[17];
        check(opmhip_convergence(ctx_, dt, param_.tolerance_cnv_, c), "getReservoirConvergence");
        const double cnvErrorPvFraction = c[10] / c[9];
        const bool use_relaxed = cnvErrorPvFraction < param_.relaxed_max_pv_fraction_ && iteration >= param_.max_strict_iter_;
        const double tol_cnv = use_relaxed ? param_.tolerance_cnv_relaxed_ : param_.tolerance_cnv_;
        bool converged = true;
        for (int comp = 0; comp < 3; ++comp) {
            const double res[2] = {c[14 + comp], c[11 + comp]};
            const double tol[2] = {param_.tolerance_mb_, tol_cnv};
            for (int ii = 0; ii < 2; ++ii) {
                if (std::isnan(res[ii])) throw NumericalIssue("NaN residual found!");
                if (res[ii] > param_.max_residual_allowed_) throw NumericalIssue("Too large residual found!");
                if (res[ii] < 0.0 || res[ii] > tol[ii]) converged = false;
            }
            residual_norms[comp] = c[11 + comp];
        }
        return converged;
    }

    /// solveJacobianSystem: x = 0; prepare (ILU0 factorisation); solve (BiCGStab); the solution stays on the device
    void solveJacobianSystem(SimulatorReportSingle& report) {
        opmhip_result r;
        check(opmhip_solve_system(ctx_, N_, nnz_, 3, nullptr, nullptr, nullptr, nullptr, nullptr, &r), "solveJacobianSystem");
        report.linear_solve_setup_time += r.t_factor;
        report.linear_solve_time += r.t_solve + r.t_copy;
        report.total_linear_iterations += (unsigned)r.iterations;
        linear_iterations_last_solve_ = r.iterations;
        // ISTLSolverEbos::checkConvergence (linalg/ISTLSolverEbos.hpp:334-345)
        if (!r.converged) throw NumericalIssue("Convergence failure for linear solver.");
    }

    bool detectOscillations(int it) const {
        if (it < 2) return false;
        const auto &F0 = residual_norms_history_[it], &F1 = residual_norms_history_[it - 1], &F2 = residual_norms_history_[it - 2];
        int oscillatePhase = 0;
        for (int p = 0; p < 3; ++p) {
            const double d1 = std::abs((F0[p] - F2[p]) / F0[p]);
            const double d2 = std::abs((F0[p] - F1[p]) / F0[p]);
            oscillatePhase += (d1 < param_.relax_rel_tol_) && (param_.relax_rel_tol_ < d2);
        }
        return oscillatePhase > 1;
    }

    SimulatorReportSingle nonlinearIteration(int iteration, double dt) {
        SimulatorReportSingle report;
        if (iteration == 0) {
            residual_norms_history_.clear();
            current_relaxation_ = 1.0;
        }
        report.total_linearizations = 1;
        double t0 = now();
        report += assembleReservoir(dt, iteration);
        std::array<double, 3> residual_norms{};
        const bool conv = getConvergence(dt, iteration, residual_norms);  // synchronises
        report.assemble_time += now() - t0;
        report.converged = conv && iteration > param_.newton_min_iter_;
        residual_norms_history_.push_back(residual_norms);
        if (!report.converged) {
            report.total_newton_iterations = 1;
            solveJacobianSystem(report);
            t0 = now();
            if (param_.use_update_stabilization_ && detectOscillations(iteration))
                current_relaxation_ = std::max(current_relaxation_ - param_.relax_increment_, param_.relax_max_);
            int nsw = 0;
            check(opmhip_update(ctx_, nullptr, current_relaxation_, &nsw), "updateSolution");
            report.update_time += now() - t0;
        }
        return report;
    }

    /// NonlinearSolverEbos::step
    SimulatorReportSingle step(double dt) {
        SimulatorReportSingle report;
        int iteration = 0;
        bool converged = false;
        do {
            SimulatorReportSingle iterReport = nonlinearIteration(iteration, dt);
            report += iterReport;
            report.converged = iterReport.converged;
            converged = report.converged;
            iteration += 1;
        } while ((!converged && (iteration <= param_.newton_max_iter_)) || (iteration <= param_.newton_min_iter_));
        if (!converged)
            throw TooManyIterations("Solver convergence failure - Failed to complete a time step within " + std::to_string(param_.newton_max_iter_) + " iterations.");
        report.converged = true;
        return report;
    }

    /// FvBaseDiscretization::advanceTimeLevel / updateFailed: solution(1) <-> solution(0) on the device
    void advanceTimeLevel() { check(opmhip_advance_time_level(ctx_), "advanceTimeLevel"); }
    void updateFailed() { check(opmhip_update_failed(ctx_), "updateFailed"); }
    /// EclProblem::beginTimeStep, the per-cell part (ebos/eclproblem.hh:1042-1075): DRSDT / DRVDT caps of a step of size dt,
    /// minimum oil pressure of irreversible compaction (a no-op where neither is in force)
    void beginTimeStep(double dt) { check(opmhip_begin_time_step(ctx_, dt), "beginTimeStep"); }
    /// EclProblem::endTimeStep, the drift-compensation part (ebos/eclproblem.hh:1126-1135): after an ACCEPTED time step
    void endTimeStep(double dt) { check(opmhip_end_time_step(ctx_, dt), "endTimeStep"); }

    /// BlackoilModelEbos::relativeChange (flow/BlackoilModelEbos.hpp:431-510): the PID time-step control's error measure, on the device
    double relativeChange() const {
        double v = 0.0;
        check(opmhip_relative_change(ctx_, &v), "relativeChange");
        return v;
    }

    /// PIDTimeStepControl::computeTimeStepSize (timestepping/TimeStepControl.cpp:117-161)
    struct PIDTimeStepControl {
        double tol;
        double errors[3];
        explicit PIDTimeStepControl(double t = 1e-1) : tol(t), errors{t, t, t} {}
        double computeTimeStepSize(double dt, double error) {
            errors[0] = errors[1]; errors[1] = errors[2]; errors[2] = error;
            if (error > tol) return dt * tol / error;
            const double kP = 0.075, kI = 0.175, kD = 0.01;
            return dt * std::pow(errors[1] / errors[2], kP) * std::pow(tol / errors[2], kI) * std::pow(errors[0] * errors[0] / errors[1] / errors[2], kD);
        }
    };
    PIDTimeStepControl pid_;   // Flow's default --time-step-control=pid+newtoniteration, tolerance 1e-1

    /// The sub-step loop of AdaptiveTimeSteppingEbos::step (timestepping/AdaptiveTimeSteppingEbos.hpp:283-520) for one
    /// report step of length `length` starting with sub-step `dt`: a failed sub-step is rolled back and retried with
    /// dt * 0.33 (SolverRestartFactor, at most SolverMaxRestarts = 10 times in a row); an accepted one sets the next dt
    /// by PIDAndIterationCountTimeStepControl (TimeStepControl.cpp:169-208): the smaller of the PID estimate from
    /// relativeChange() and the Newton-iteration-count estimate (target 8, growth damping 3.2, decay damping 1),
    /// capped by SolverMaxGrowth = 3 and, right after a chop, by SolverGrowthFactor = 2.  Returns the suggested next dt.
    double advanceReportStep(double length, double dt, SimulatorReportSingle& report, int* chopped = nullptr) {
        const double restartFactor = 0.33, growthFactor = 2.0, maxGrowth = 3.0, growthDamping = 3.2, decayDamping = 1.0;
        const int maxRestarts = 10, target = 8;
        double t = 0.0;
        int restarts = 0;
        while (t < length * (1.0 - 1e-12)) {
            dt = std::min(dt, length - t);
            if (restarts == 0) advanceTimeLevel();
            beginTimeStep(dt);   // also in front of every retry of a chopped step
            int newtons = 0;
            bool ok = true;
            try {
                SimulatorReportSingle r = step(dt);
                report += r;
                newtons = r.total_newton_iterations;
            } catch (const TooManyIterations&) { ok = false; }
            catch (const NumericalIssue&) { ok = false; }
            if (!ok) {
                if (++restarts > maxRestarts) throw TooManyIterations("time step chopped " + std::to_string(maxRestarts) + " times in a row");
                if (chopped) ++*chopped;
                updateFailed();
                dt *= restartFactor;
                continue;
            }
            endTimeStep(dt);
            t += dt;
            const double estIter = newtons > target ? dt / (1.0 + double(newtons - target) / target * decayDamping)
                                                    : dt * (1.0 + double(target - newtons) / target * growthDamping);
            double est = std::min(pid_.computeTimeStepSize(dt, relativeChange()), estIter);
            est = std::min(est, maxGrowth * dt);
            if (restarts > 0) { est = std::min(growthFactor * dt, est); restarts = 0; }
            dt = est;
        }
        return dt;
    }
};

}  // namespace Opm
