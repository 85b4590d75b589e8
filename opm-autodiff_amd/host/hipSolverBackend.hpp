// bda::hipSolverBackend<3> : the BdaSolver<block_size> plugin a Flow build selects with --accelerator-mode=hip
// (one more branch in BdaBridge's ctor, opm/simulators/linalg/bda/BdaBridge.cpp:65-120).  Header-only shim over the
// C-ABI of libopmhip.so: no HIP types here, so it compiles with the host compiler of the reference.
//
// Behaviour kept from the existing backends (bda/cusparseSolverBackend.cu:480-499, bda/openclSolverBackend.cpp:783-805):
//   - first solve_system(): analysis of the sparsity pattern (rows/cols are constant afterwards);
//   - values and rhs are borrowed host memory, copied to the device on every call;
//   - a failed analysis / factorisation is reported through SolverStatus, never thrown; device errors are
//     std::logic_error like OPM_THROW(std::logic_error, ...) in bda/cuda_header.hpp:36-44;
//   - a non-converged solve returns SUCCESS with res.converged == false and ISTLSolverEbos falls back to Dune
//     (linalg/ISTLSolverEbos.hpp:277-297).
#pragma once
#ifdef OPMHIP_USE_OPM_HEADERS
#include <opm/simulators/linalg/bda/BdaResult.hpp>
#include <opm/simulators/linalg/bda/BdaSolver.hpp>
#include <opm/simulators/linalg/bda/WellContributions.hpp>
#else
#include "BdaCompat.hpp"
#endif
#include <stdexcept>
#include <string>

#include "../../include/opmhip.h"

namespace bda {

template <unsigned int block_size>
class hipSolverBackend : public BdaSolver<block_size> {
    typedef BdaSolver<block_size> Base;
    using Base::deviceID;
    using Base::maxit;
    using Base::tolerance;
    using Base::verbosity;
    opmhip_ctx* ctx = nullptr;
    opmhip_wells wells{};
    bool haveWells = false;

public:
    /// ilu_reorder as --opencl-ilu-reorder: "auto" (default: the library's measured choice - line colouring on large structured grids,
    /// the greedy colouring elsewhere) | "level_scheduling" | "graph_coloring" (the reference's Jones-Plassmann rounds, bda/Reorder.cpp:59-172)
    /// | "graph_coloring_greedy" | "line_coloring".  The BdaBridge branch of INTEGRATION.md hands over "auto" when the user left
    /// --opencl-ilu-reorder untouched and the user's string otherwise (bda/BdaBridge.cpp:72-73);
    /// linsolver as --linear-solver-configuration: "ilu0" | "cpr_quasiimpes" | "cpr" = "cpr_trueimpes" (setupPropertyTree.cpp:46-138).
    /// The true-IMPES variant needs the model's storage term: the caller hands the result of
    /// ISTLSolverEbos::getTrueImpesWeights (ISTLSolverEbos.hpp:466-475) to setCprWeights() before each solve.
    hipSolverBackend(int linear_solver_verbosity, int maxit_, double tolerance_, unsigned int deviceID_,
                     const std::string& ilu_reorder = "auto", double ilu_relaxation = 0.9,
                     const std::string& linsolver = "ilu0", int cpr_reuse_setup = 3, int cpr_amg_ilu_levels = -1, int cpr_gather_rows = 0)
        : Base(linear_solver_verbosity, maxit_, tolerance_, deviceID_) {
        static_assert(block_size == 3, "libopmhip handles 3x3 blocks (three-phase black-oil)");
        opmhip_config cfg;
        opmhip_default_config(&cfg);
        cfg.verbosity = verbosity;
        cfg.maxit = maxit;
        cfg.tolerance = tolerance;
        cfg.device_id = (int)deviceID;
        cfg.ilu_relaxation = ilu_relaxation;
        if (ilu_reorder == "level_scheduling") cfg.reorder = OPMHIP_REORDER_LEVEL_SCHEDULING;
        else if (ilu_reorder == "graph_coloring") cfg.reorder = OPMHIP_REORDER_GRAPH_COLORING;
        else if (ilu_reorder == "graph_coloring_greedy") cfg.reorder = OPMHIP_REORDER_GRAPH_COLORING_GREEDY;
        else if (ilu_reorder == "line_coloring") cfg.reorder = OPMHIP_REORDER_LINE_COLORING;
        else if (ilu_reorder == "auto" || ilu_reorder.empty()) cfg.reorder = OPMHIP_REORDER_AUTO;   // opmhip_default_config's
        else throw std::logic_error("Error invalid argument for --opencl-ilu-reorder, usage: '--opencl-ilu-reorder=[auto|level_scheduling|graph_coloring|graph_coloring_greedy|line_coloring]'");
        if (linsolver == "cpr_quasiimpes") cfg.preconditioner = OPMHIP_PRECOND_CPR_QUASIIMPES;
        else if (linsolver == "cpr" || linsolver == "cpr_trueimpes") cfg.preconditioner = OPMHIP_PRECOND_CPR_TRUEIMPES;
        else if (linsolver != "ilu0") throw std::invalid_argument(linsolver + " is not a valid setting for --linear-solver-configuration with --accelerator-mode=hip. Please use ilu0, cpr, cpr_trueimpes, or cpr_quasiimpes");
        // --cpr-reuse-setup (FlowLinearSolverParameters.hpp:212-214).  The library acts on 0 (every solve) and 2 (after a solve of
        // more than 10 iterations) by itself; 1 (first Newton iteration of a time step) is the caller's to signal: recreateCprHierarchy()
        cfg.cpr_reuse_setup = cpr_reuse_setup;
        // The reference's pressure AMG smooths with ILU0 on every level (PreconditionerFactory.hpp:126-151).  Here: on the
        // cpr_amg_ilu_levels finest levels; < 0 = the library's choice: level 0 where the block ILU0's ordering has at most three colours
        // (four sweep launches per application, +5 ... +8 % Newton iterations/s on the 100^3 case), Jacobi elsewhere (with the seven colours
        // of a greedy-coloured corner-point grid Jacobi is 1.5 x faster)
        cfg.cpr_amg_ilu_levels = cpr_amg_ilu_levels;
        cfg.cpr_gather_rows = cpr_gather_rows;   // parallel runs: the pressure stage spans the ranks (0: default size of the joined level; < 0: off)
        // Flow's matrix values, right-hand side and solution vector keep their addresses for the life of the run (bda/BdaBridge.cpp:199-232,
        // linalg/ISTLSolverEbos.hpp:216-219): registered for DMA the first time they are seen, every later copy runs at the link's rate
        cfg.pin_host_arrays = 1;
        const int rc = opmhip_create(&cfg, &ctx);
        if (rc != OPMHIP_SUCCESS) throw std::logic_error(std::string("hipSolverBackend: ") + opmhip_last_error(nullptr));
    }
    ~hipSolverBackend() override { opmhip_destroy(ctx); }
    /// weights of the CPR preconditioner (3 per block row), e.g. Amg::getTrueImpesWeights; nullptr: computed by the library
    void setCprWeights(const double* weights) {
        if (opmhip_set_cpr_weights(ctx, weights) != OPMHIP_SUCCESS) throw std::logic_error(std::string("hipSolverBackend: ") + opmhip_last_error(ctx));
    }
    /// ISTLSolverEbos::shouldCreateSolver said yes (ISTLSolverEbos.hpp:401-426): the next solve_system builds the hierarchy anew
    void recreateCprHierarchy() {
        if (opmhip_cpr_recreate(ctx) != OPMHIP_SUCCESS) throw std::logic_error(std::string("hipSolverBackend: ") + opmhip_last_error(ctx));
    }
    hipSolverBackend(const hipSolverBackend&) = delete;
    hipSolverBackend& operator=(const hipSolverBackend&) = delete;

    SolverStatus solve_system(int N_, int nnz_, int dim, double* vals, int* rows, int* cols, double* b,
                              WellContributions& wellContribs, BdaResult& res) override {
        // WellContributions::getNumWells() counts standard AND multisegment wells (bda/WellContributions.hpp:164-166); the C arrays
        // belong to the standard ones only.  Multisegment wells stay what they are in the reference - host objects with a sparse LU of D -
        // and are applied through the host round trip of its back-ends (bda/WellContributions.cu:160-187): the library calls back.
        opmhip_wells* wp = nullptr;
        if (wellContribs.getNumWells() > 0) {
            wells = opmhip_wells{};
            wells.num_wells = (int)wellContribs.getNumStdWells();
            if (wells.num_wells > 0) {
#ifndef OPMHIP_USE_OPM_HEADERS
                wells.val_pointers = wellContribs.valPointers.data();
                wells.Ccols = wellContribs.Ccols.data();
                wells.Bcols = wellContribs.Bcols.data();
                wells.Cnnzs = wellContribs.Cnnzs.data();
                wells.Dnnzs = wellContribs.Dnnzs.data();
                wells.Bnnzs = wellContribs.Bnnzs.data();
#else
                // with the reference's WellContributions the host-side arrays are reached through the accessors the
                // patch in INTEGRATION.md adds (the class keeps them private for the CUDA/OpenCL paths)
                wellContribs.getHostArrays(&wells.val_pointers, &wells.Ccols, &wells.Bcols, &wells.Cnnzs, &wells.Dnnzs, &wells.Bnnzs);
#endif
            }
            wells.num_ms_wells = (int)wellContribs.getNumMSWells();
            if (wells.num_ms_wells > 0) {
                wells.ms_user = &wellContribs;
                wells.ms_apply = [](void* user, const double* h_x, double* h_y) {
                    static_cast<WellContributions*>(user)->applyMSWellsHost(const_cast<double*>(h_x), h_y);   // apply() reads x (MultisegmentWellContribution.cpp:78-90)
                };
            }
            wp = &wells;
        }
        opmhip_result r;
        const int rc = opmhip_solve_system(ctx, N_, nnz_, dim, vals, rows, cols, b, wp, &r);
        res.iterations = r.iterations;
        res.reduction = r.reduction;
        res.converged = r.converged != 0;
        res.conv_rate = r.conv_rate;
        res.elapsed = r.elapsed;
        this->initialized = true;
        switch (rc) {
        case OPMHIP_SUCCESS: return SolverStatus::BDA_SOLVER_SUCCESS;
        case OPMHIP_ANALYSIS_FAILED: return SolverStatus::BDA_SOLVER_ANALYSIS_FAILED;
        case OPMHIP_CREATE_PRECONDITIONER_FAILED: return SolverStatus::BDA_SOLVER_CREATE_PRECONDITIONER_FAILED;
        case OPMHIP_DEVICE_ERROR:
        case OPMHIP_NO_DEVICE: throw std::logic_error(std::string("hipSolverBackend: ") + opmhip_last_error(ctx));
        default: return SolverStatus::BDA_SOLVER_UNKNOWN_ERROR;
        }
    }

    void get_result(double* x) override {
        if (opmhip_get_result(ctx, x) != OPMHIP_SUCCESS) throw std::logic_error(std::string("hipSolverBackend::get_result: ") + opmhip_last_error(ctx));
    }
    opmhip_ctx* context() { return ctx; }
};

}  // namespace bda
