// Stand-ins for the few reference declarations the HIP backend shim touches, so that the shim can be compiled and
// tested outside an opm-simulators tree.  Inside opm-simulators these come from the reference's own headers
// (opm/simulators/linalg/bda/BdaResult.hpp:28-40, BdaSolver.hpp:32-92, WellContributions.hpp:60-214) and this file is
// not used: define OPMHIP_USE_OPM_HEADERS and include those instead.
#pragma once
#include <algorithm>
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

// The subset of Opm::WellContributions a backend reads: standard wells only, filled in the order C, D, B per well
// (wells/StandardWellEval.cpp:1206-1250) with dim = 3, dim_wells = 4 (WellContributions.cpp:215-225).
class WellContributions {
public:
    enum class MatrixType { C, D, B };
    void setBlockSize(unsigned dim_, unsigned dimWells_) { dim = dim_; dimWells = dimWells_; }
    unsigned getNumWells() const { return numWells; }
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
