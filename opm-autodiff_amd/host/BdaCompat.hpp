// Stand-ins for the few reference declarations the HIP backend shim touches, so that the shim can be compiled and
// tested outside an opm-simulators tree.  Inside opm-simulators these come from the reference's own headers
// (opm/simulators/linalg/bda/BdaResult.hpp:28-40, BdaSolver.hpp:32-92, WellContributions.hpp:60-214) and this file is
// not used: define OPMHIP_USE_OPM_HEADERS and include those instead.
#pragma once
#include <algorithm>
#include <cmath>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace bda {

class BdaResult {
public:
    int iterations = 0;
    double reduction = 0.0;
    bool converged = false;
    double conv_rate = 0.0;
    double elapsed = 0.0;
};

enum class SolverStatus { BDA_SOLVER_SUCCESS, BDA_SOLVER_ANALYSIS_FAILED, BDA_SOLVER_CREATE_PRECONDITIONER_FAILED, BDA_SOLVER_UNKNOWN_ERROR };

}  // namespace bda

namespace Opm {

// Stand-in for Opm::MultisegmentWellContribution (bda/MultisegmentWellContribution.hpp:43-130): B and C in blocked CSR with one
// pattern (blocks of dim_wells x dim, C's stored like B's), D in CSC over M = Mb * dim_wells scalar rows, apply(h_x, h_y) on host
// vectors.  The interface is the reference's; the inside is not: the reference factors D with UMFPack (not in this image), this
// stand-in with a dense LU with partial pivoting - test scaffolding, the product never sees it (it only calls apply through the
// plug-in's callback).
class MultisegmentWellContribution {
public:
    using UMFPackIndex = int;
    MultisegmentWellContribution(unsigned dim_, unsigned dim_wells_, unsigned Mb_, std::vector<double>& Bvalues, std::vector<unsigned>& BcolIndices,
                                 std::vector<unsigned>& BrowPointers, unsigned DnumBlocks, double* Dvalues, UMFPackIndex* DcolPointers,
                                 UMFPackIndex* DrowIndices, std::vector<double>& Cvalues)
        : dim(dim_), dim_wells(dim_wells_), M(Mb_ * dim_wells_), Mb(Mb_), Cvals(std::move(Cvalues)), Bvals(std::move(Bvalues)),
          Bcols(std::move(BcolIndices)), Brows(std::move(BrowPointers)), z1(M), z2(M), lu((size_t)M * M, 0.0), piv(M) {
        (void)DnumBlocks;
        for (unsigned c = 0; c < M; ++c)
            for (int k = DcolPointers[c]; k < DcolPointers[c + 1]; ++k) lu[(size_t)DrowIndices[k] * M + c] += Dvalues[k];
        for (unsigned k = 0; k < M; ++k) {
            unsigned p = k;
            for (unsigned i = k + 1; i < M; ++i)
                if (std::fabs(lu[(size_t)i * M + k]) > std::fabs(lu[(size_t)p * M + k])) p = i;
            if (lu[(size_t)p * M + k] == 0.0) throw std::logic_error("MultisegmentWellContribution: singular D");
            piv[k] = p;
            if (p != k)
                for (unsigned j = 0; j < M; ++j) std::swap(lu[(size_t)k * M + j], lu[(size_t)p * M + j]);
            for (unsigned i = k + 1; i < M; ++i) {
                const double f = lu[(size_t)i * M + k] /= lu[(size_t)k * M + k];
                for (unsigned j = k + 1; j < M; ++j) lu[(size_t)i * M + j] -= f * lu[(size_t)k * M + j];
            }
        }
    }
    // y -= C^T (D^-1 (B x)), statement for statement what bda/MultisegmentWellContribution.cpp:70-110 computes
    void apply(double* h_x, double* h_y) {
        std::fill(z1.begin(), z1.end(), 0.0);
        for (unsigned row = 0; row < Mb; ++row)
            for (unsigned blockID = Brows[row]; blockID < Brows[row + 1]; ++blockID) {
                const unsigned colIdx = getColIdx(Bcols[blockID]);
                for (unsigned j = 0; j < dim_wells; ++j) {
                    double temp = 0.0;
                    for (unsigned k = 0; k < dim; ++k) temp += Bvals[blockID * dim * dim_wells + j * dim + k] * h_x[colIdx * dim + k];
                    z1[row * dim_wells + j] += temp;
                }
            }
        z2 = z1;
        for (unsigned k = 0; k < M; ++k) {
            std::swap(z2[k], z2[piv[k]]);
            for (unsigned i = k + 1; i < M; ++i) z2[i] -= lu[(size_t)i * M + k] * z2[k];
        }
        for (unsigned k = M; k-- > 0;) {
            for (unsigned j = k + 1; j < M; ++j) z2[k] -= lu[(size_t)k * M + j] * z2[j];
            z2[k] /= lu[(size_t)k * M + k];
        }
        for (unsigned row = 0; row < Mb; ++row)
            for (unsigned blockID = Brows[row]; blockID < Brows[row + 1]; ++blockID) {
                const unsigned colIdx = getColIdx(Bcols[blockID]);
                for (unsigned j = 0; j < dim; ++j) {
                    double temp = 0.0;
                    for (unsigned k = 0; k < dim_wells; ++k) temp += Cvals[blockID * dim * dim_wells + j + k * dim] * z2[row * dim_wells + k];
                    h_y[colIdx * dim + j] -= temp;
                }
            }
    }
    void setReordering(int* toOrder_, bool reorder_) { toOrder = toOrder_; reorder = reorder_; }
private:
    unsigned getColIdx(unsigned idx) const { return reorder ? (unsigned)toOrder[idx] : idx; }
    unsigned dim, dim_wells, M, Mb;
    std::vector<double> Cvals, Bvals;
    std::vector<unsigned> Bcols, Brows;
    std::vector<double> z1, z2, lu;
    std::vector<unsigned> piv;
    int* toOrder = nullptr;
    bool reorder = false;
};

// The subset of Opm::WellContributions a backend reads: standard wells filled in the order C, D, B per well
// (wells/StandardWellEval.cpp:1206-1250) with dim = 3, dim_wells = 4 (WellContributions.cpp:215-225), and multisegment wells as a list of
// MultisegmentWellContribution objects (WellContributions.hpp:85-92, WellContributions.cpp:261-274); getNumWells() counts BOTH kinds, as the
// reference's does (WellContributions.hpp:164-166).
class WellContributions {
public:
    using UMFPackIndex = MultisegmentWellContribution::UMFPackIndex;
    enum class MatrixType { C, D, B };
    void setBlockSize(unsigned dim_, unsigned dimWells_) { dim = dim_; dimWells = dimWells_; }
    unsigned getNumWells() const { return numWells + (unsigned)multisegments.size(); }
    void addMultisegmentWellContribution(unsigned dim_, unsigned dim_wells_, unsigned Mb, std::vector<double>& Bvalues, std::vector<unsigned>& BcolIndices,
                                         std::vector<unsigned>& BrowPointers, unsigned DnumBlocks, double* Dvalues, UMFPackIndex* DcolPointers,
                                         UMFPackIndex* DrowIndices, std::vector<double>& Cvalues) {
        multisegments.emplace_back(new MultisegmentWellContribution(dim_, dim_wells_, Mb, Bvalues, BcolIndices, BrowPointers, DnumBlocks, Dvalues,
                                                                    DcolPointers, DrowIndices, Cvalues));
    }
    // the accessors INTEGRATION.md's patch adds to the reference class (num_std_wells and the list are private there)
    unsigned getNumStdWells() const { return numWells; }
    unsigned getNumMSWells() const { return (unsigned)multisegments.size(); }
    void applyMSWellsHost(double* h_x, double* h_y) {   // the loop of WellContributions.cu:175-178 / WellContributions.cpp:128-132, natural order
        for (auto& well : multisegments) {
            well->setReordering(nullptr, false);
            well->apply(h_x, h_y);
        }
    }
    void addNumBlocks(unsigned nb) { numBlocksPending.push_back(nb); }
    void alloc() {
        valPointers.assign(1, 0);
        for (unsigned nb : numBlocksPending) valPointers.push_back(valPointers.back() + (int)nb);
        numWells = (unsigned)numBlocksPending.size();
        const size_t np = (size_t)valPointers.back();
        Cnnzs.assign(np * dim * dimWells, 0.0); Bnnzs.assign(np * dim * dimWells, 0.0); Dnnzs.assign((size_t)numWells * dimWells * dimWells, 0.0);
        Ccols.assign(np, 0); Bcols.assign(np, 0);
        cursorC = cursorB = cursorD = 0;
    }
    void addMatrix(MatrixType type, const int* colIndices, const double* values, unsigned valSize) {
        const unsigned blk = dim * dimWells;
        switch (type) {
        case MatrixType::C:
            for (unsigned i = 0; i < valSize; ++i) Ccols[cursorC + i] = colIndices[i];
            std::copy(values, values + (size_t)valSize * blk, Cnnzs.begin() + (size_t)cursorC * blk);
            cursorC += valSize;
            break;
        case MatrixType::D:
            std::copy(values, values + (size_t)dimWells * dimWells, Dnnzs.begin() + (size_t)cursorD * dimWells * dimWells);
            cursorD += 1;
            break;
        case MatrixType::B:
            for (unsigned i = 0; i < valSize; ++i) Bcols[cursorB + i] = colIndices[i];
            std::copy(values, values + (size_t)valSize * blk, Bnnzs.begin() + (size_t)cursorB * blk);
            cursorB += valSize;
            break;
        }
    }
    // the accessor INTEGRATION.md's patch adds to the reference class (its host vectors are private there)
    void getHostArrays(const int** val_pointers, const int** Ccols_, const int** Bcols_, const double** Cnnzs_,
                       const double** Dnnzs_, const double** Bnnzs_) const {
        *val_pointers = valPointers.data(); *Ccols_ = Ccols.data(); *Bcols_ = Bcols.data();
        *Cnnzs_ = Cnnzs.data(); *Dnnzs_ = Dnnzs.data(); *Bnnzs_ = Bnnzs.data();
    }
    // raw views for a backend
    unsigned dim = 3, dimWells = 4, numWells = 0;
    std::vector<int> valPointers, Ccols, Bcols;
    std::vector<double> Cnnzs, Dnnzs, Bnnzs;
private:
    std::vector<unsigned> numBlocksPending;
    unsigned cursorC = 0, cursorB = 0, cursorD = 0;
    std::vector<std::unique_ptr<MultisegmentWellContribution>> multisegments;
};

}  // namespace Opm

namespace bda {
using Opm::WellContributions;

template <unsigned int block_size>
class BdaSolver {
protected:
    int verbosity = 0;
    int maxit = 200;
    double tolerance = 1e-2;
    int N = 0, Nb = 0, nnz = 0, nnzb = 0;
    unsigned int deviceID = 0;
    bool initialized = false;
public:
    BdaSolver(int linear_solver_verbosity, int max_it, double tolerance_, unsigned int deviceID_)
        : verbosity(linear_solver_verbosity), maxit(max_it), tolerance(tolerance_), deviceID(deviceID_) {}
    virtual ~BdaSolver() {}
    virtual SolverStatus solve_system(int N, int nnz, int dim, double* vals, int* rows, int* cols, double* b,
                                      WellContributions& wellContribs, BdaResult& res) = 0;
    virtual void get_result(double* x) = 0;
};
}  // namespace bda
