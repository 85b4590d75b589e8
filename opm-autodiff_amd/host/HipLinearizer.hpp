// The assembly half of the drop-in: a Linearizer for Flow's TypeTag system whose linearizeDomain() runs on the GPU.
//
// Flow reaches the linearizer only through
//     model().linearizer().linearizeDomain()                      opm/simulators/flow/BlackoilModelEbos.hpp:424
//     model().linearizer().jacobian() / .residual()               :339-340 (wellModel().linearize), :526-527 (the solve), :636, :738
//                                                                 (convergence), ebos/eclproblem.hh:1128 (drift), ebos/eclnewtonmethod.hh:151
//     model().solution(0), invalidateAndUpdateIntensiveQuantities :552-562 (updateSolution)
// and it is chosen by the property system (pattern: flow/BlackoilModelEbos.hpp:129-136):
//     template<class TypeTag> struct Linearizer<TypeTag, TTag::EclFlowProblemHip> { using type = Opm::HipLinearizer<TypeTag>; };
// The class below has the public face of opm-models' FvBaseLinearizer (registerParameters, init, eraseMatrix, linearize,
// linearizeDomain, finalize, linearizeAuxiliaryEquations, jacobian, residual, constraintsMap - that class is not in the reference
// tree, its face is what the call sites above use) and does its work through the C-ABI (include/opmhip.h).  Two ways to run it:
//   host copies ON  (default): after linearizeDomain() jacobian() and residual() hold what FvBaseLinearizer would have put there
//                   (bit for bit what opmhip_assemble computes), so everything downstream - wellModel().linearize, any linear
//                   solver, the convergence check, the host-side Newton update - runs unchanged;
//   host copies OFF: Jacobian and residual stay in HBM; the solve is opmhip_solve_system(vals = NULL) on context() (the
//                   hipSolverBackend plug-in of the same context), the update opmhip_update through updateSolutionOnDevice().
// What the problem has to offer (EclProblem has all of it; the numbers are ebos/eclproblem.hh lines):
//   transmissibility(i, j) :1332-1340   thresholdPressure(i, j) :1409   porosity(i) :1430   dofCenterDepth(i) :1447
//   pvtRegionIndex(i) / satnumRegionIndex(i)   maxGasDissolutionFactor(timeIdx, i) :1711-1732   source(rate, i, timeIdx) :1823-1845
//   model().dofTotalVolume(i), model().numGridDof(), model().solution(0), model().newtonMethod().numIterations(),
//   simulator.timeStepSize(); and two accessors a maintainer adds: hipFluidTables() (the deck's PVT / saturation tables as
//   opmhip_fluid, from eclState's TableManager) and stencilNeighbors(i) (the ECFV stencil's neighbour list, what
//   FvBaseLinearizer::createMatrix_ walks through).
#pragma once
#include <opm/models/utils/propertysystem.hh>

#include <algorithm>
#include <map>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/opmhip.h"

namespace Opm {

/// does the problem name the dofs its source terms can sit in (`const std::vector<int>& sourceDofs() const`)?
template <class Problem, class = void> struct HasSourceDofs : std::false_type {};
template <class Problem> struct HasSourceDofs<Problem, std::void_t<decltype(std::declval<const Problem&>().sourceDofs())>> : std::true_type {};

template <class TypeTag>
class HipLinearizer {
    using Simulator = GetPropType<TypeTag, Properties::Simulator>;
    using SparseMatrixAdapter = GetPropType<TypeTag, Properties::SparseMatrixAdapter>;
    using GlobalEqVector = GetPropType<TypeTag, Properties::GlobalEqVector>;
    using RateVector = GetPropType<TypeTag, Properties::RateVector>;

public:
    struct Constraints {};   // Flow's black-oil model sets none (the map stays empty: ebos/eclnewtonmethod.hh:151 only iterates it)

    HipLinearizer() = default;
    HipLinearizer(const HipLinearizer&) = delete;
    ~HipLinearizer() { if (ctx_) opmhip_destroy(ctx_); }

    static void registerParameters() {}

    /// config: solver / ordering settings of the context (NULL: opmhip_default_config); hostCopies: see the header comment
    void init(Simulator& simulator, const opmhip_config* config = nullptr, bool hostCopies = true) {
        simulatorPtr_ = &simulator;
        hostCopies_ = hostCopies;
        opmhip_config cfg;
        if (config) cfg = *config; else opmhip_default_config(&cfg);
        if (opmhip_create(&cfg, &ctx_) != OPMHIP_SUCCESS) throw std::runtime_error(std::string("HipLinearizer: ") + opmhip_last_error(nullptr));
        createMatrix_();
        const auto& problem = simulator.problem();
        const auto& model = simulator.model();
        check_(opmhip_set_fluid(ctx_, &problem.hipFluidTables()), "opmhip_set_fluid");
        // per-connection and per-cell static data in the pattern's order (EclProblem serves them by (i, j) / by i)
        std::vector<double> trans(nnz_, 0.0), area(nnz_, 0.0), thpres(nnz_, 0.0), poro(N_), volume(N_), depth(N_), rsmax(N_);
        std::vector<int> pvtnum(N_), satnum(N_);
        bool anyThpres = false, anyRsMax = false;
        for (int i = 0; i < N_; ++i) {
            for (int k = rows_[i]; k < rows_[i + 1]; ++k) {
                const int j = cols_[k];
                if (j == i) continue;
                trans[k] = problem.transmissibility(i, j);
                area[k] = problem.faceArea(i, j);
                thpres[k] = problem.thresholdPressure(i, j);
                anyThpres |= thpres[k] != 0.0;
            }
            poro[i] = problem.porosity(i);
            volume[i] = model.dofTotalVolume(i);
            depth[i] = problem.dofCenterDepth(i);
            pvtnum[i] = problem.pvtRegionIndex(i);
            satnum[i] = problem.satnumRegionIndex(i);
            rsmax[i] = problem.maxGasDissolutionFactor(/*timeIdx=*/0, i);
            anyRsMax |= rsmax[i] < 1e100;
        }
        check_(opmhip_set_static(ctx_, trans.data(), area.data(), anyThpres ? thpres.data() : nullptr, poro.data(), volume.data(), depth.data(),
                                 pvtnum.data(), satnum.data(), anyRsMax ? rsmax.data() : nullptr), "opmhip_set_static");
        residual_.resize(N_);
        pv_.resize((size_t)N_ * 3);
        meaning_.resize(N_);
        source_.resize((size_t)N_ * 3);
        dsource_.resize((size_t)N_ * 9);
    }

    void eraseMatrix() { jacobian_.reset(); }

    void linearize() { linearizeDomain(); linearizeAuxiliaryEquations(); }

    /// the mass-balance equations of every cell: solution(0) and the source terms go to the device, k_iq_update + k_assemble run
    /// there; with host copies the results come back into jacobian() / residual()
    void linearizeDomain() {
        if (!ctx_) throw std::logic_error("HipLinearizer::linearizeDomain before init");
        if (!jacobian_) createJacobian_();
        auto& sim = *simulatorPtr_;
        if (!stateOnDevice_) solutionToDevice();
        // EclProblem::source (:1823-1845): the wells' total rates per dof and equation, with derivatives
        RateVector rate;
        auto record = [&](int i, size_t slot) {
            sim.problem().source(rate, i, /*timeIdx=*/0);
            for (int e = 0; e < 3; ++e) {
                source_[slot * 3 + e] = rate.value(e);
                for (int v = 0; v < 3; ++v) dsource_[slot * 9 + e * 3 + v] = rate.derivative(e, v);
            }
        };
        if constexpr (HasSourceDofs<std::decay_t<decltype(sim.problem())>>::value) {
            // a problem that names the dofs a source can sit in (Flow: the cells BlackoilWellModel::is_cell_perforated_ marks,
            // wells/BlackoilWellModel_impl.hpp:496-512, 1606-1630): only their rates are evaluated and cross PCIe
            const std::vector<int>& dofs = sim.problem().sourceDofs();
            for (size_t q = 0; q < dofs.size(); ++q) record(dofs[q], q);
            check_(opmhip_set_source_cells(ctx_, (int)dofs.size(), dofs.data(), source_.data(), dsource_.data()), "opmhip_set_source_cells");
        } else {
            for (int i = 0; i < N_; ++i) record(i, (size_t)i);
            check_(opmhip_set_source(ctx_, source_.data(), dsource_.data()), "opmhip_set_source");
        }
        const int iteration = sim.model().newtonMethod().numIterations();
        // the block values of a BCRSMatrix and the entries of a BlockVector are contiguous: the pointers BdaBridge hands its
        // back-ends (linalg/bda/BdaBridge.cpp:231-232) are the ones the assembled system is written through
        double* jac = hostCopies_ ? &(jacobian_->istlMatrix()[0][0][0][0]) : nullptr;
        double* res = hostCopies_ ? &(residual_[0][0]) : nullptr;
        check_(opmhip_assemble(ctx_, sim.timeStepSize(), iteration, jac, res), "opmhip_assemble");
        stateOnDevice_ = false;   // the host may change solution(0) before the next linearisation (the Newton update does)
    }

    void finalize() {}
    void linearizeAuxiliaryEquations() {}   // Flow's wells enter through wellModel().linearize / the well operator, not as auxiliary modules of this path

    const SparseMatrixAdapter& jacobian() const { return *jacobian_; }
    SparseMatrixAdapter& jacobian() { return *jacobian_; }
    const GlobalEqVector& residual() const { return residual_; }
    GlobalEqVector& residual() { return residual_; }
    const std::map<unsigned, Constraints>& constraintsMap() const { return constraintsMap_; }

    // ---- the device side of model().solution(0) / invalidateAndUpdateIntensiveQuantities (BlackoilModelEbos.hpp:552-562) ----
    /// solution(0) -> device, intensive quantities recomputed there (= opmhip_set_state)
    void solutionToDevice() {
        const auto& sol = simulatorPtr_->model().solution(/*timeIdx=*/0);
        for (int i = 0; i < N_; ++i) {
            for (int v = 0; v < 3; ++v) pv_[(size_t)i * 3 + v] = sol[i][v];
            meaning_[i] = (unsigned char)sol[i].primaryVarsMeaning();
        }
        check_(opmhip_set_state(ctx_, pv_.data(), meaning_.data()), "opmhip_set_state");
        stateOnDevice_ = true;
    }
    /// what the model's invalidateAndUpdateIntensiveQuantities(0) forwards to once solution(0) was changed on the host
    void invalidateAndUpdateIntensiveQuantities(unsigned /*timeIdx*/) { solutionToDevice(); }
    /// device -> solution(0)
    void solutionToHost() {
        check_(opmhip_get_state(ctx_, pv_.data(), meaning_.data()), "opmhip_get_state");
        auto& sol = simulatorPtr_->model().solution(/*timeIdx=*/0);
        for (int i = 0; i < N_; ++i) {
            for (int v = 0; v < 3; ++v) sol[i][v] = pv_[(size_t)i * 3 + v];
            sol[i].setPrimaryVarsMeaning(meaning_[i]);
        }
    }
    /// BlackoilModelEbos::updateSolution on the device (update_ with its chops and switches, then the intensive quantities) with the
    /// solution of the last opmhip_solve_system of this context; solution(0) of the host follows
    void updateSolutionOnDevice(double relaxation = 1.0) {
        check_(opmhip_update(ctx_, nullptr, relaxation, nullptr), "opmhip_update");
        solutionToHost();
        stateOnDevice_ = true;
    }

    opmhip_ctx* context() { return ctx_; }
    int numRows() const { return N_; }
    int numBlocks() const { return nnz_; }

private:
    // FvBaseLinearizer::createMatrix_: the sparsity pattern from the stencils (row i: i and its neighbours, ascending)
    void createMatrix_() {
        const auto& model = simulatorPtr_->model();
        N_ = (int)model.numGridDof();
        rows_.assign(N_ + 1, 0);
        cols_.clear();
        for (int i = 0; i < N_; ++i) {
            std::vector<int> row(model.stencilNeighbors(i).begin(), model.stencilNeighbors(i).end());
            row.push_back(i);
            std::sort(row.begin(), row.end());
            row.erase(std::unique(row.begin(), row.end()), row.end());
            cols_.insert(cols_.end(), row.begin(), row.end());
            rows_[i + 1] = (int)cols_.size();
        }
        nnz_ = (int)cols_.size();
        check_(opmhip_set_pattern(ctx_, N_, nnz_, rows_.data(), cols_.data()), "opmhip_set_pattern");
    }
    // FvBaseLinearizer::createMatrix_'s second half: the adapter allocated from the sparsity pattern (std::vector<std::set<unsigned>>)
    void createJacobian_() {
        std::vector<std::set<unsigned>> sparsityPattern(N_);
        for (int i = 0; i < N_; ++i) sparsityPattern[i].insert(cols_.begin() + rows_[i], cols_.begin() + rows_[i + 1]);
        jacobian_.reset(new SparseMatrixAdapter(*simulatorPtr_));
        jacobian_->reserve(sparsityPattern);
    }
    void check_(int rc, const char* what) const {
        if (rc != OPMHIP_SUCCESS) throw std::logic_error(std::string("HipLinearizer: ") + what + ": " + opmhip_last_error(ctx_));
    }

    Simulator* simulatorPtr_ = nullptr;
    opmhip_ctx* ctx_ = nullptr;
    bool hostCopies_ = true, stateOnDevice_ = false;
    int N_ = 0, nnz_ = 0;
    std::vector<int> rows_, cols_;
    std::unique_ptr<SparseMatrixAdapter> jacobian_;
    GlobalEqVector residual_;
    std::map<unsigned, Constraints> constraintsMap_;
    std::vector<double> pv_, source_, dsource_;
    std::vector<unsigned char> meaning_;
};

}  // namespace Opm
