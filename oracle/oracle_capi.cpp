// ORACLE — TEST INFRASTRUCTURE ONLY (see linalg.hpp header).  C entry points so that tests/,
// smoke() and bench.py's cpu_baseline leg can drive the CPU restatement through ctypes.
#include <chrono>
#include <cstdio>

#include "linalg.hpp"

using namespace orc;

namespace {
Bcrs wrap(int Nb, const int* rowptr, const int* col, const double* val) {
    Bcrs A;
    A.Nb = Nb;
    A.rowptr.assign(rowptr, rowptr + Nb + 1);
    const int nnzb = rowptr[Nb];
    A.col.assign(col, col + nnzb);
    if (val) A.val.assign(val, val + (size_t)nnzb * BB);
    else A.val.assign((size_t)nnzb * BB, 0.0);
    return A;
}
Wells wrap_wells(int numWells, const int* val_pointers, const int* Ccols, const int* Bcols,
                 const double* Cnnzs, const double* Dnnzs, const double* Bnnzs) {
    Wells W;
    W.numWells = numWells;
    if (numWells <= 0) return W;
    const int nperf = val_pointers[numWells];
    W.val_pointers.assign(val_pointers, val_pointers + numWells + 1);
    W.Ccols.assign(Ccols, Ccols + nperf);
    W.Bcols.assign(Bcols, Bcols + nperf);
    W.Cnnzs.assign(Cnnzs, Cnnzs + (size_t)nperf * 12);
    W.Bnnzs.assign(Bnnzs, Bnnzs + (size_t)nperf * 12);
    W.Dnnzs.assign(Dnnzs, Dnnzs + (size_t)numWells * 16);
    return W;
}
}  // namespace

extern "C" {

int orc_spmv(int Nb, const int* rowptr, const int* col, const double* val, const double* x, double* y) {
    Bcrs A = wrap(Nb, rowptr, col, val);
    spmv(A, x, y);
    return 0;
}

int orc_check_zero_diagonal(int Nb, const int* rowptr, const int* col, double* val) {
    Bcrs A = wrap(Nb, rowptr, col, val);
    int n = check_zero_diagonal(A);
    std::memcpy(val, A.val.data(), A.val.size() * sizeof(double));
    return n;
}

// natural-order block ILU0; lu_out gets L (strict lower), U (strict upper), D^-1 (diagonal)
int orc_ilu0_factor(int Nb, const int* rowptr, const int* col, const double* val, int interiorSize,
                    double* lu_out) {
    Bcrs A = wrap(Nb, rowptr, col, val);
    int rc = bilu0_decompose(A, interiorSize < 0 ? Nb : interiorSize);
    std::memcpy(lu_out, A.val.data(), A.val.size() * sizeof(double));
    return rc;
}

int orc_ilu0_apply(int Nb, const int* rowptr, const int* col, const double* lu, int interiorSize,
                   const double* d, double* v, double w, int mode) {
    Bcrs LU = wrap(Nb, rowptr, col, lu);
    std::vector<int> dg = diag_index(LU);
    ilu0_apply(LU, dg, interiorSize < 0 ? Nb : interiorSize, d, v, w, mode);
    return 0;
}

// kind: 0 none, 1 level scheduling, 2 graph colouring (Jones-Plassmann, deterministic weights),
//       3 graph colouring (greedy first fit)
// rowsPerColor must have room for Nb ints.  Returns the number of colours.
int orc_reorder(int Nb, const int* rowptr, const int* col, int kind, int* toOrder, int* fromOrder,
                int* rowsPerColor) {
    Bcrs A = wrap(Nb, rowptr, col, nullptr);
    Reordering R;
    if (kind == 1) R = level_schedule(A);
    else if (kind == 2) R = graph_color(A, 0);
    else if (kind == 3) R = graph_color(A, 1);
    else {
        R.toOrder.resize(Nb);
        std::iota(R.toOrder.begin(), R.toOrder.end(), 0);
        R.fromOrder = R.toOrder;
        R.rowsPerColor.assign(1, Nb);
    }
    std::memcpy(toOrder, R.toOrder.data(), Nb * sizeof(int));
    std::memcpy(fromOrder, R.fromOrder.data(), Nb * sizeof(int));
    std::memcpy(rowsPerColor, R.rowsPerColor.data(), R.rowsPerColor.size() * sizeof(int));
    return R.numColors();
}

int orc_reorder_matrix(int Nb, const int* rowptr, const int* col, const double* val, const int* toOrder,
                       const int* fromOrder, int* rrowptr, int* rcol, double* rval) {
    Bcrs A = wrap(Nb, rowptr, col, val);
    Reordering R;
    R.toOrder.assign(toOrder, toOrder + Nb);
    R.fromOrder.assign(fromOrder, fromOrder + Nb);
    Bcrs B = reorder_matrix(A, R);
    std::memcpy(rrowptr, B.rowptr.data(), (Nb + 1) * sizeof(int));
    std::memcpy(rcol, B.col.data(), B.col.size() * sizeof(int));
    if (rval) std::memcpy(rval, B.val.data(), B.val.size() * sizeof(double));
    return 0;
}

int orc_wells_apply(int numWells, const int* val_pointers, const int* Ccols, const int* Bcols,
                    const double* Cnnzs, const double* Dnnzs, const double* Bnnzs, const double* x, double* y) {
    Wells W = wrap_wells(numWells, val_pointers, Ccols, Bcols, Cnnzs, Dnnzs, Bnnzs);
    wells_apply(W, x, y);
    return 0;
}

struct orc_result {
    int iterations;
    int converged;
    double reduction;
    double conv_rate;
    double it;
    double t_factor;  // seconds in the ILU0 factorisation ("linear_solve_setup_time")
    double t_solve;   // seconds in the Krylov loop     ("linear_solve_time")
    int num_colors;
};

// ILU0-preconditioned BiCGStab on the system A x = b, x0 = 0.
//  relax_mode 0: M^-1 = w U^-1 L^-1 (Dune path, w = --ilu-relaxation, FlowLinearSolverParameters.hpp:147-149;
//                w = 1 is what the cusparse backend does, Appendix A of SURVEY.md)
//  relax_mode 1: relaxation inside the backward sweep (OpenCL backend)
//  reorder: as orc_reorder.  With reorder != 0 matrix and rhs are permuted, solved and the result permuted
//           back (bda/openclSolverBackend.cpp:699-716, 783-805).
//  nsub > 1: block-Jacobi ILU0 over nsub contiguous row ranges (couplings across ranges dropped in the
//            preconditioner only) - the CPU-N baseline of BASELINE.md §3, modelled on
//            ghost_last_bilu0_decomposition (ParallelOverlappingILU0.hpp:439-494).  sub_start[nsub+1].
int orc_solve(int Nb, const int* rowptr, const int* col, const double* val, const double* b, double* x,
              double tol, int maxit, double w, int relax_mode, int reorder, int zero_diag_fix, int numWells,
              const int* val_pointers, const int* Ccols, const int* Bcols, const double* Cnnzs,
              const double* Dnnzs, const double* Bnnzs, int nsub, const int* sub_start, orc_result* out) {
    using clk = std::chrono::steady_clock;
    Bcrs A = wrap(Nb, rowptr, col, val);
    if (zero_diag_fix) check_zero_diagonal(A);
    Wells W = wrap_wells(numWells, val_pointers, Ccols, Bcols, Cnnzs, Dnnzs, Bnnzs);
    const size_t n = (size_t)Nb * BS;
    std::vector<double> rb(b, b + n), rx(n);
    Reordering R;
    if (reorder != 0) {
        if (reorder == 1) R = level_schedule(A);
        else R = graph_color(A, reorder == 2 ? 0 : 1);
        A = reorder_matrix(A, R);
        reorder_vector(Nb, b, R.fromOrder, rb.data());
        for (auto& c : W.Ccols) c = R.toOrder[c];
        for (auto& c : W.Bcols) c = R.toOrder[c];
    }
    auto t0 = clk::now();
    Bcrs LU = A;
    int rc = 0;
    std::vector<int> dg;
    if (nsub > 1) {
        // drop couplings that leave a subdomain: they stay in the pattern with value 0 so that the
        // sweeps simply see zeros there (same effect as the ghost-last loops that never touch them)
        std::vector<int> owner(Nb);
        for (int s = 0; s < nsub; ++s)
            for (int i = sub_start[s]; i < sub_start[s + 1]; ++i) owner[i] = s;
        for (int i = 0; i < Nb; ++i)
            for (int k = LU.rowptr[i]; k < LU.rowptr[i + 1]; ++k)
                if (owner[LU.col[k]] != owner[i]) std::fill_n(&LU.val[(size_t)k * BB], BB, 0.0);
    }
    rc = bilu0_decompose(LU, Nb);
    dg = diag_index(LU);
    auto t1 = clk::now();
    if (rc != 0) return rc;
    std::vector<double> tmp(n);
    auto prec = [&](const double* d, double* v) { ilu0_apply(LU, dg, Nb, d, v, w, relax_mode); };
    auto op = [&](const double* xin, double* y) {
        spmv(A, xin, y);
        if (W.numWells > 0) wells_apply(W, xin, y);
    };
    SolveResult r = bicgstab(n, rb.data(), rx.data(), prec, op, tol, maxit);
    auto t2 = clk::now();
    if (reorder != 0) {
        for (int i = 0; i < Nb; ++i)
            for (int c = 0; c < BS; ++c) x[(size_t)i * BS + c] = rx[(size_t)R.toOrder[i] * BS + c];
    } else {
        std::memcpy(x, rx.data(), n * sizeof(double));
    }
    if (out) {
        out->iterations = r.iterations;
        out->converged = r.converged;
        out->reduction = r.reduction;
        out->conv_rate = r.conv_rate;
        out->it = r.it;
        out->t_factor = std::chrono::duration<double>(t1 - t0).count();
        out->t_solve = std::chrono::duration<double>(t2 - t1).count();
        out->num_colors = reorder ? R.numColors() : 0;
    }
    return 0;
}

// unpreconditioned BiCGStab ("nothing" preconditioner of tests/test_flexiblesolver.cpp:93-116)
int orc_solve_noprec(int Nb, const int* rowptr, const int* col, const double* val, const double* b, double* x,
                     double tol, int maxit, int repeat, orc_result* out) {
    // repeat = 2 gives the "RepeatingOperator" A*A of tests/test_preconditionerfactory.cpp:300-316
    Bcrs A = wrap(Nb, rowptr, col, val);
    const size_t n = (size_t)Nb * BS;
    std::vector<double> t1(n);
    auto prec = [&](const double* d, double* v) { std::memcpy(v, d, n * sizeof(double)); };
    auto op = [&](const double* xin, double* y) {
        spmv(A, xin, y);
        for (int k = 1; k < repeat; ++k) {
            std::memcpy(t1.data(), y, n * sizeof(double));
            spmv(A, t1.data(), y);
        }
    };
    SolveResult r = bicgstab(n, b, x, prec, op, tol, maxit);
    if (out) {
        out->iterations = r.iterations;
        out->converged = r.converged;
        out->reduction = r.reduction;
        out->conv_rate = r.conv_rate;
        out->it = r.it;
        out->t_factor = out->t_solve = 0;
        out->num_colors = 0;
    }
    return 0;
}

}  // extern "C"
