// ORACLE — TEST INFRASTRUCTURE ONLY (see linalg.hpp header).  C entry points so that tests/,
// smoke() and bench.py's cpu_baseline leg can drive the CPU restatement through ctypes.
#include <chrono>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <cstdio>

#include "linalg.hpp"
#include "cpr.hpp"

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
// vals is modified in place
int orc_wells_add_to_matrix(int Nb, const int* rowptr, const int* col, double* vals, int numWells, const int* val_pointers, const int* Ccols,
                            const int* Bcols, const double* Cnnzs, const double* Dnnzs, const double* Bnnzs) {
    Bcrs A = wrap(Nb, rowptr, col, vals);
    Wells W = wrap_wells(numWells, val_pointers, Ccols, Bcols, Cnnzs, Dnnzs, Bnnzs);
    const int rc = wells_add_to_matrix(W, A);
    std::memcpy(vals, A.val.data(), A.val.size() * sizeof(double));
    return rc;
}
int orc_wells_apply_residual(int numWells, const int* val_pointers, const int* Ccols, const int* Bcols, const double* Cnnzs,
                             const double* Dnnzs, const double* Bnnzs, const double* resWell, double* r) {
    Wells W = wrap_wells(numWells, val_pointers, Ccols, Bcols, Cnnzs, Dnnzs, Bnnzs);
    wells_apply_residual(W, resWell, r);
    return 0;
}
int orc_wells_recover(int numWells, const int* val_pointers, const int* Ccols, const int* Bcols, const double* Cnnzs,
                      const double* Dnnzs, const double* Bnnzs, const double* resWell, const double* x, double* xw) {
    Wells W = wrap_wells(numWells, val_pointers, Ccols, Bcols, Cnnzs, Dnnzs, Bnnzs);
    wells_recover(W, resWell, x, xw);
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
int orc_solve_hp(int Nb, const int* rowptr, const int* col, const double* val, const double* b, double* x,
                 double tol, int maxit, double w, int relax_mode, int reorder, int zero_diag_fix, int numWells,
                 const int* val_pointers, const int* Ccols, const int* Bcols, const double* Cnnzs,
                 const double* Dnnzs, const double* Bnnzs, int nsub, const int* sub_start, int half_product, orc_result* out);
int orc_solve(int Nb, const int* rowptr, const int* col, const double* val, const double* b, double* x,
              double tol, int maxit, double w, int relax_mode, int reorder, int zero_diag_fix, int numWells,
              const int* val_pointers, const int* Ccols, const int* Bcols, const double* Cnnzs,
              const double* Dnnzs, const double* Bnnzs, int nsub, const int* sub_start, orc_result* out) {
    return orc_solve_hp(Nb, rowptr, col, val, b, x, tol, maxit, w, relax_mode, reorder, zero_diag_fix, numWells, val_pointers, Ccols, Bcols, Cnnzs,
                        Dnnzs, Bnnzs, nsub, sub_start, 0, out);
}
// half_product != 0: the product after every ILU0 application is formed from the backward sweep's row sums (oracle/linalg.hpp:
// ilu0_apply_u / spmv_rest - the order libopmhip's opmhip_config.half_product states); returns -1000 if the (reordered) pattern does not
// have the property that rests on (is_upper_alias)
int orc_solve_hp(int Nb, const int* rowptr, const int* col, const double* val, const double* b, double* x,
                 double tol, int maxit, double w, int relax_mode, int reorder, int zero_diag_fix, int numWells,
                 const int* val_pointers, const int* Ccols, const int* Bcols, const double* Cnnzs,
                 const double* Dnnzs, const double* Bnnzs, int nsub, const int* sub_start, int half_product, orc_result* out) {
    using clk = std::chrono::steady_clock;
    const bool fused = (half_product & 2) != 0;
    half_product &= 1;
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
    std::vector<int> owner;
    if (nsub > 1 || nsub < 0) {
        // drop couplings that leave a subdomain: they stay in the pattern with value 0 so that the
        // sweeps simply see zeros there (same effect as the ghost-last loops that never touch them).
        // nsub > 1: contiguous row ranges sub_start[nsub+1]; nsub < 0: sub_start is an owner id per (natural) row.
        owner.resize(Nb);
        if (nsub < 0) {
            for (int i = 0; i < Nb; ++i) owner[reorder != 0 ? R.toOrder[i] : i] = sub_start[i];
        } else
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
    if (half_product && !is_upper_alias(A)) return -1000;
    std::vector<double> ubuf(half_product ? n : 0);
    auto prec = [&](const double* d, double* v) {
        if (half_product) ilu0_apply_u(LU, dg, Nb, d, v, w, relax_mode, ubuf.data());
        else ilu0_apply(LU, dg, Nb, d, v, w, relax_mode);
    };
    auto op = [&](const double* xin, double* y) {
        // (half_product: xin is the vector the last prec() call produced - BiCGStab applies the operator to nothing else)
        if (half_product) spmv_rest(A, xin, ubuf.data(), relax_mode == 0 ? w : 1.0, y, owner.empty() ? nullptr : owner.data());
        else spmv(A, xin, y);
        if (W.numWells > 0) wells_apply(W, xin, y);
    };
    // half_product bit 1 (value 2 or 3): the recurrence with one reduction per half iteration (bicgstab_fused_reductions)
    SolveResult r = fused ? bicgstab_fused_reductions(n, rb.data(), rx.data(), prec, op, tol, maxit) : bicgstab(n, rb.data(), rx.data(), prec, op, tol, maxit);
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

// t = A (M^-1 d), z = M^-1 d: one ILU0 application and the product behind it - plain (ilu0_apply, spmv) or in the half-product form
// (ilu0_apply_u, spmv_rest); lu = the factors of orc_ilu0_factor.  The pair of bda/cusparseSolverBackend.cu:103-118.
int orc_preconditioned_product(int Nb, const int* rowptr, const int* col, const double* val, const double* lu, const double* d, double w,
                               int relax_mode, int half_product, double* t, double* z) {
    Bcrs A = wrap(Nb, rowptr, col, val);
    Bcrs LU = wrap(Nb, rowptr, col, lu);
    const std::vector<int> dg = diag_index(LU);
    const size_t n = (size_t)Nb * BS;
    if (half_product) {
        if (!is_upper_alias(A)) return -1000;
        std::vector<double> u(n);
        ilu0_apply_u(LU, dg, Nb, d, z, w, relax_mode, u.data());
        spmv_rest(A, z, u.data(), relax_mode == 0 ? w : 1.0, t);
    } else {
        ilu0_apply(LU, dg, Nb, d, z, w, relax_mode);
        spmv(A, z, t);
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

// ---- CPR (oracle/cpr.hpp): handle keeps the AMG hierarchy's structure between solves -------------------------------------
struct orc_cpr { Cpr P; };
orc_cpr* orc_cpr_create(double omega, double damp, double beta) {
    orc_cpr* h = new orc_cpr();
    if (omega > 0.0) h->P.amg.omega = omega;
    if (damp > 0.0) h->P.amg.damp = damp;
    if (beta >= 0.0) h->P.amg.beta = beta;
    if (beta <= -2.0) h->P.amg.join = true;   // experiment switch
    return h;
}
void orc_cpr_destroy(orc_cpr* h) { delete h; }
// which pressure AMG the handle uses: 0 = the product's (pairwise matching + Jacobi), 1 = the restatement of the reference's
// (Dune::Amg-like aggregation + ILU0 smoothing + direct coarse solve; comparison only).  Before the first update / solve.
// the system handed to this preconditioner is a REORDERED one: nat[i] = natural id of its row i (n = 0: it is in natural order)
int orc_cpr_set_natural_ids(orc_cpr* h, int n, const int* nat) {
    CprAmg& G = h->P.amg;
    G.natOf.assign(nat, nat + n);
    G.atNat.assign(n, 0);
    for (int i = 0; i < n; ++i) G.atNat[nat[i]] = i;
    h->P.structured = false;
    return 0;
}
int orc_cpr_set_coarse_sweeps(orc_cpr* h, int n) { h->P.amg.coarseSweeps = n < 0 ? 0 : n; return 0; }
// the next update / solve builds the hierarchy's STRUCTURE anew from its matrix (what --cpr-reuse-setup 0 / 1 / 2 make the product do)
int orc_cpr_rebuild_structure(orc_cpr* h) { h->P.structured = false; return 0; }
int orc_cpr_set_max_levels(orc_cpr* h, int n) { h->P.amg.maxLevels = n < 1 ? 1 : n; h->P.structured = false; return 0; }
int orc_cpr_set_sweeps(orc_cpr* h, int nu) { if (nu < 0) { h->P.amg.joinAtStall = true; h->P.structured = false; return 0; } h->P.amg.nu = nu < 1 ? 1 : nu; return 0; }   // experiments: V(nu, nu)
// smoother of the product's hierarchy: scalar ILU0 (relaxation 1) on levels < levels, eliminating in greedy multi-colour order on levels >= colour_from
int orc_cpr_set_ilu_smoother(orc_cpr* h, int levels, int colour_from) { h->P.amg.iluLevels = levels < 0 ? 0 : levels; h->P.amg.iluColourFrom = colour_from < 0 ? (1 << 30) : colour_from; h->P.structured = false; return 0; }
int orc_cpr_set_aggregation(orc_cpr* h, int kind) { h->P.amg.duneAgg = kind == 1; h->P.structured = false; return 0; }   // experiment: 1 = the reference's kind of aggregation in the product's hierarchy
int orc_cpr_set_wcycle_from(orc_cpr* h, int l) { h->P.amg.wFrom = l < 0 ? (1 << 30) : l; return 0; }   // experiment
int orc_cpr_use_reference_amg(orc_cpr* h, int on) { h->P.useDune = on != 0; h->P.dune.jacobi = on == 2;   /* 2: experiment - its aggregation with damped Jacobi smoothing */ h->P.structured = false; return 0; }
// levels of that hierarchy
int orc_cpr_reference_amg_levels(orc_cpr* h, int* n, int* nnz, int cap) {
    const int L = (int)h->P.dune.lv.size();
    for (int l = 0; l < L && l < cap; ++l) { n[l] = h->P.dune.lv[l].A.n; nnz[l] = (int)h->P.dune.lv[l].A.col.size(); }
    return L;
}
// BiCGStab with the CPR preconditioner on the system AS GIVEN (callers hand over the matrix in the ordering they want the
// ILU0 smoother and the aggregation to see); x natural to that ordering
int orc_cpr_solve(orc_cpr* h, int Nb, const int* rowptr, const int* col, const double* val, const double* b, double* x,
                  double tol, int maxit, int zero_diag_fix, orc_result* out) {
    Bcrs A = wrap(Nb, rowptr, col, val);
    if (zero_diag_fix) check_zero_diagonal(A);
    const int rc = h->P.update(A);
    if (rc) return rc;
    const size_t n = (size_t)Nb * BS;
    auto prec = [&](const double* d, double* v) { h->P.apply(d, v); };
    auto op = [&](const double* xin, double* y) { spmv(A, xin, y); };
    SolveResult r = bicgstab(n, b, x, prec, op, tol, maxit);
    if (out) {
        out->iterations = r.iterations; out->converged = r.converged; out->reduction = r.reduction; out->conv_rate = r.conv_rate;
        out->it = r.it; out->t_factor = out->t_solve = 0; out->num_colors = (int)h->P.amg.lv.size();
    }
    return 0;
}
// Decomposed runs: BiCGStab on the GLOBAL system with one CPR per subdomain (owner id per row) as preconditioner - each over its
// owned rows and columns, the couplings between subdomains left out of the pressure hierarchy and of the block ILU0 alike, as
// the product's per-rank CPR does; local numbering = the global order restricted to the subdomain.  weights: NULL = quasi-IMPES
// per subdomain, else Nb x 3 (true-IMPES from the model).  levels (may be NULL): number of AMG levels of every subdomain.
// nat (may be NULL): the system is handed over in another ordering than the natural one (the product's ILU0 ordering inside every
// subdomain); nat[i] = natural id of row i - the finest level is then aggregated in natural visiting order, as the product does.
// gather_rows >= 0 (0: 100 000, the product's default): a pressure stage that spans the subdomains (the product: csrc/cpr.hip, cpr_gather_*; the reference: Dune's
// parallel AMG behind linalg/OwningTwoLevelPreconditioner.hpp).  Every subdomain coarsens on its own (aggregates never cross a
// boundary) down to its first level of at most that many rows; those levels are joined into ONE system - their own entries plus,
// between aggregates of different subdomains, the Galerkin sums of the fine couplings - which is coarsened further and cycled on as a
// whole.  One application:  x = omega D^-1 r_p;  r = r_p - A_p x with the pressure matrix of the WHOLE system;  r summed over the
// aggregates level by level down to the joined level;  one V-cycle there;  x' = x + (the result, per aggregate);  x'' = x' + omega
// D^-1 (r_p - A_p x');  v = (0, x'', 0) + ILU0_subdomain(d - A (0, x'', 0)), again with the whole system's operator.  The levels between
// level 0 and the joined one carry no smoothing (measured: it is level 0's and the block ILU0's that count, tools/cpr_gather_study.py).
// Row sums run over the subdomain's own columns first (ascending), then over the others by (owner, natural id) - the order of a rank's
// local numbering, ghost cells last (ras.py).  < 0: off.
// glevels (may be NULL, room for 32): rows of the joined hierarchy's levels, their number in *nglevels.  probe_d / probe_v (may be NULL):
// one application of the preconditioner to probe_d, for the device's parity check.
int orc_cpr_solve_blocks(int Nb, const int* rowptr, const int* col, const double* val, const double* b, double* x, const int* owner, int nown,
                         const double* weights, const int* nat, double tol, int maxit, int zero_diag_fix, int* levels, orc_result* out,
                         int gather_rows, int* glevels, int* nglevels, const double* probe_d, double* probe_v, int ilu_levels) {
    Bcrs A = wrap(Nb, rowptr, col, val);
    if (zero_diag_fix) check_zero_diagonal(A);
    const bool gather = gather_rows >= 0;
    std::vector<std::vector<int>> rows(nown);
    std::vector<int> local(Nb);
    for (int i = 0; i < Nb; ++i) { local[i] = (int)rows[owner[i]].size(); rows[owner[i]].push_back(i); }
    std::vector<Bcrs> sub(nown);
    std::vector<Cpr> prec(nown);
    for (int s = 0; s < nown; ++s) {
        Bcrs& S = sub[s];
        S.Nb = (int)rows[s].size();
        S.rowptr.assign(1, 0);
        for (int i : rows[s]) {
            for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k)
                if (owner[A.col[k]] == s) {
                    S.col.push_back(local[A.col[k]]);
                    S.val.insert(S.val.end(), &A.val[(size_t)k * BB], &A.val[(size_t)k * BB] + BB);
                }
            S.rowptr.push_back((int)S.col.size());
        }
        if (nat) {   // rank of every row's natural id inside the subdomain, and its inverse
            std::vector<int> order(S.Nb);
            for (int q = 0; q < S.Nb; ++q) order[q] = q;
            std::sort(order.begin(), order.end(), [&](int a, int c) { return nat[rows[s][a]] < nat[rows[s][c]]; });
            prec[s].amg.natOf.resize(S.Nb);
            prec[s].amg.atNat = order;
            for (int v = 0; v < S.Nb; ++v) prec[s].amg.natOf[order[v]] = v;
        }
        if (weights) {
            prec[s].w_given.resize((size_t)S.Nb * BS);
            for (int q = 0; q < S.Nb; ++q)
                for (int k = 0; k < BS; ++k) prec[s].w_given[(size_t)q * BS + k] = weights[(size_t)rows[s][q] * BS + k];
        }
        if (gather) { prec[s].amg.stopRows = gather_rows > 0 ? gather_rows : 100000; prec[s].amg.external = true; }
        // ilu_levels > 0 (experiment; the product smooths level 0 of the spanning stage with Jacobi - measured here: 13.0 -> 12.0 and 11.5 -> 10.5
        // iterations on 8 x 20^3 cells, not worth a sweep): the AMGs' finest levels smooth with ILU0 (CprAmg::iluLevels); with the stage that spans the subdomains: level 0
        // alone, each subdomain with the ILU0 of its own part of the pressure matrix (the residuals are the whole system's all the same)
        if (ilu_levels > 0) { prec[s].amg.iluLevels = gather ? 1 : ilu_levels; prec[s].amg.iluColourFrom = 1; prec[s].amg.iluAlways0 = gather; }
        const int rc = prec[s].update(S);
        if (rc) return rc;
        if (levels) levels[s] = (int)prec[s].amg.lv.size();
    }
    CprAmg G;
    std::vector<int> offs(nown + 1, 0), gcid(Nb, 0), korder, pdiag(Nb, -1);
    std::vector<double> ap, pdinv(Nb, 0.0);
    if (gather) {
        for (int s = 0; s < nown; ++s) offs[s + 1] = offs[s] + prec[s].amg.lv.back().A.n;
        for (int s = 0; s < nown; ++s) {   // row of the joined level every cell belongs to
            const CprAmg& M = prec[s].amg;
            for (int q = 0; q < sub[s].Nb; ++q) {
                int a = q;
                for (size_t l = 0; l + 1 < M.lv.size(); ++l) a = M.lv[l].agg[a];
                gcid[rows[s][q]] = offs[s] + a;
            }
        }
        // the pressure matrix of the whole system, every row with its subdomain's weights; korder: a row's entries in the order a rank
        // walks them (own columns, then the others by owner and natural id)
        ap.resize(A.col.size());
        korder.resize(A.col.size());
        for (int i = 0; i < Nb; ++i) {
            const int s = owner[i];
            const double* w = &prec[s].w[(size_t)local[i] * BS];
            std::vector<int> own, other;
            for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k) {
                double v = 0.0;
                for (int r = 0; r < BS; ++r) v += A.val[(size_t)k * BB + r * BS + CPR_PRESSURE_INDEX] * w[r];
                ap[k] = v;
                (owner[A.col[k]] == s ? own : other).push_back(k);
                if (A.col[k] == i) pdiag[i] = k;
            }
            std::sort(other.begin(), other.end(), [&](int ka, int kb) {
                const int ja = A.col[ka], jb = A.col[kb];
                if (owner[ja] != owner[jb]) return owner[ja] < owner[jb];
                return (nat ? nat[ja] : ja) < (nat ? nat[jb] : jb);
            });
            int o = A.rowptr[i];
            for (int k : own) korder[o++] = k;
            for (int k : other) korder[o++] = k;
            pdinv[i] = 1.0 / ap[pdiag[i]];
        }
        // couplings between subdomains, summed per pair of aggregates in ascending (row, column)
        std::vector<std::map<int, double>> cross(offs[nown]);
        for (int i = 0; i < Nb; ++i)
            for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k) {
                const int j = A.col[k];
                if (owner[j] == owner[i]) continue;
                auto it = cross[gcid[i]].find(gcid[j]);
                if (it == cross[gcid[i]].end()) cross[gcid[i]][gcid[j]] = 0.0 + ap[k]; else it->second += ap[k];
            }
        Csr J;
        J.n = offs[nown];
        J.rowptr.assign(1, 0);
        for (int s = 0; s < nown; ++s) {
            const Csr& LA = prec[s].amg.lv.back().A;
            for (int I = 0; I < LA.n; ++I) {
                std::map<int, double> row(cross[offs[s] + I]);
                for (int k = LA.rowptr[I]; k < LA.rowptr[I + 1]; ++k) row[offs[s] + LA.col[k]] = LA.val[k];
                for (auto& e : row) { J.col.push_back(e.first); J.val.push_back(e.second); }
                J.rowptr.push_back((int)J.col.size());
            }
        }
        G.omega = prec[0].amg.omega; G.damp = prec[0].amg.damp; G.beta = prec[0].amg.beta;
        G.setup_structure(J);
        if (nglevels) *nglevels = (int)G.lv.size();
        if (glevels) for (size_t l = 0; l < G.lv.size() && l < 32; ++l) glevels[l] = G.lv[l].A.n;
    } else if (nglevels) *nglevels = 0;
    const size_t n = (size_t)Nb * BS;
    // s = bi - sum over row i's entries, in the rank's order, of a_p x
    auto prow = [&](int i, double bi, const std::vector<double>& xv) {
        double sres = bi;
        for (int o = A.rowptr[i]; o < A.rowptr[i + 1]; ++o) sres -= ap[korder[o]] * xv[A.col[korder[o]]];
        return sres;
    };
    auto apply = [&](const double* d, double* v) {
        std::vector<std::vector<double>> dl(nown);
        for (int s = 0; s < nown; ++s) {
            const int m = sub[s].Nb;
            dl[s].resize((size_t)m * BS);
            for (int q = 0; q < m; ++q)
                for (int k = 0; k < BS; ++k) dl[s][(size_t)q * BS + k] = d[(size_t)rows[s][q] * BS + k];
        }
        if (!gather) {
            for (int s = 0; s < nown; ++s) {
                const int m = sub[s].Nb;
                std::vector<double> vl((size_t)m * BS);
                prec[s].apply(dl[s].data(), vl.data());
                for (int q = 0; q < m; ++q)
                    for (int k = 0; k < BS; ++k) v[(size_t)rows[s][q] * BS + k] = vl[(size_t)q * BS + k];
            }
            return;
        }
        const double om = prec[0].amg.omega;
        std::vector<double> rp(Nb), x0(Nb), r(Nb), x1(Nb), x2(Nb), bG(offs[nown], 0.0), xG(offs[nown], 0.0);
        for (int s = 0; s < nown; ++s) {
            std::vector<double> rc(sub[s].Nb);
            prec[s].restrict_fine(dl[s].data(), rc.data());
            for (int q = 0; q < sub[s].Nb; ++q) rp[rows[s][q]] = rc[q];
        }
        // level 0's smoother: damped Jacobi, or every subdomain's scalar ILU0 of its own part of the pressure matrix
        auto smooth0 = [&](const std::vector<double>& in, std::vector<double>& outv) {
            if (ilu_levels <= 0) { for (int i = 0; i < Nb; ++i) outv[i] = om * pdinv[i] * in[i]; return; }
            for (int s = 0; s < nown; ++s) {
                const int m = sub[s].Nb;
                std::vector<double> a(m), t(m);
                for (int q = 0; q < m; ++q) a[q] = in[rows[s][q]];
                prec[s].amg.smooth(prec[s].amg.lv[0], a.data(), t.data());
                for (int q = 0; q < m; ++q) outv[rows[s][q]] = t[q];
            }
        };
        smooth0(rp, x0);
        for (int i = 0; i < Nb; ++i) r[i] = prow(i, rp[i], x0);
        for (int s = 0; s < nown; ++s) {   // the residual summed over the aggregates, level by level, down to the joined level
            const CprAmg& M = prec[s].amg;
            std::vector<double> cur(sub[s].Nb);
            for (int q = 0; q < sub[s].Nb; ++q) cur[q] = r[rows[s][q]];
            for (size_t l = 0; l + 1 < M.lv.size(); ++l) {
                const AmgLevel& L = M.lv[l];
                std::vector<double> nx(L.nc);
                for (int I = 0; I < L.nc; ++I) {
                    double sum = 0.0;
                    for (int q = L.mptr[I]; q < L.mptr[I + 1]; ++q) sum += cur[L.midx[q]];
                    nx[I] = sum;
                }
                cur.swap(nx);
            }
            std::copy(cur.begin(), cur.end(), bG.begin() + offs[s]);
        }
        G.vcycle(bG.data(), xG.data());
        for (int i = 0; i < Nb; ++i) x1[i] = x0[i] + 1.0 * xG[gcid[i]];
        {
            std::vector<double> r1(Nb), t1(Nb);
            for (int i = 0; i < Nb; ++i) r1[i] = prow(i, rp[i], x1);
            smooth0(r1, t1);
            for (int i = 0; i < Nb; ++i) x2[i] = x1[i] + t1[i];
        }
        // v = (0, x'', 0) + ILU0_subdomain(d - A (0, x'', 0)) with the whole system's operator, rows in the rank's order
        std::vector<double> rr(n);
        for (int i = 0; i < Nb; ++i) {
            double y[BS] = {0.0, 0.0, 0.0};
            for (int o = A.rowptr[i]; o < A.rowptr[i + 1]; ++o) {
                const int k = korder[o];
                for (int cidx = 0; cidx < BS; ++cidx) y[cidx] += A.val[(size_t)k * BB + cidx * BS + CPR_PRESSURE_INDEX] * x2[A.col[k]];
            }
            for (int cidx = 0; cidx < BS; ++cidx) rr[(size_t)i * BS + cidx] = d[(size_t)i * BS + cidx] - y[cidx];
        }
        for (int s = 0; s < nown; ++s) {
            const int m = sub[s].Nb;
            std::vector<double> rl((size_t)m * BS), zl((size_t)m * BS);
            for (int q = 0; q < m; ++q)
                for (int k = 0; k < BS; ++k) rl[(size_t)q * BS + k] = rr[(size_t)rows[s][q] * BS + k];
            ilu0_apply(prec[s].LU, prec[s].dg, m, rl.data(), zl.data(), 1.0, 0);
            for (int q = 0; q < m; ++q)
                for (int k = 0; k < BS; ++k) v[(size_t)rows[s][q] * BS + k] = (k == CPR_PRESSURE_INDEX ? x2[rows[s][q]] : 0.0) + zl[(size_t)q * BS + k];
        }
    };
    if (probe_d && probe_v) apply(probe_d, probe_v);
    auto op = [&](const double* xin, double* y) { spmv(A, xin, y); };
    SolveResult r = bicgstab(n, b, x, apply, op, tol, maxit);
    if (out) {
        out->iterations = r.iterations; out->converged = r.converged; out->reduction = r.reduction; out->conv_rate = r.conv_rate;
        out->it = r.it; out->t_factor = out->t_solve = 0; out->num_colors = nown;
    }
    return 0;
}
// v = M_cpr^-1 d with the preconditioner of the last orc_cpr_solve / orc_cpr_update
int orc_cpr_update(orc_cpr* h, int Nb, const int* rowptr, const int* col, const double* val) {
    static thread_local Bcrs keep;   // Cpr::apply reads the matrix it was updated with
    keep = wrap(Nb, rowptr, col, val);
    return h->P.update(keep);
}
int orc_cpr_apply(orc_cpr* h, const double* d, double* v) { h->P.apply(d, v); return 0; }
// sizes of the AMG levels (unknowns, entries); returns the number of levels
int orc_cpr_levels(orc_cpr* h, int* n, int* nnz, int cap) {
    const int L = (int)h->P.amg.lv.size();
    for (int l = 0; l < L && l < cap; ++l) { n[l] = h->P.amg.lv[l].A.n; nnz[l] = (int)h->P.amg.lv[l].A.col.size(); }
    return L;
}
// weights from outside (true-IMPES: orc_bo_true_impes_weights); n == 0: back to quasi-IMPES
int orc_cpr_set_weights(orc_cpr* h, int n, const double* w) {
    if (n > 0) h->P.w_given.assign(w, w + n); else h->P.w_given.clear();
    return 0;
}
int orc_cpr_weights(orc_cpr* h, double* w) { std::memcpy(w, h->P.w.data(), h->P.w.size() * sizeof(double)); return 0; }
// level l: aggregate of every node (for the device parity test of the host-side aggregation)
int orc_cpr_aggregates(orc_cpr* h, int l, int* agg) {
    const AmgLevel& L = h->P.amg.lv[l];
    std::memcpy(agg, L.agg.data(), L.agg.size() * sizeof(int));
    return (int)L.agg.size();
}

}  // extern "C"

// =====================================================================================================
// black-oil assembly path
// =====================================================================================================
#include "blackoil.hpp"

namespace {
// flat deck-level fluid description; identical layout to `struct opmhip_fluid` of include/opmhip.h
struct orc_fluid_desc {
    int num_pvt, num_sat;
    const double *pvtw, *density;
    const int* pvdg_ptr; const double* pvdg;
    const int* pvto_node_ptr; const double* pvto_rs; const int* pvto_row_ptr; const double* pvto;
    const int* swof_ptr; const double* swof;
    const int* sgof_ptr; const double* sgof;
    double rock_pref, rock_cr;
    // wet gas (PVTG) and rock compaction tables (ROCKTAB); pointers may be NULL / counts 0
    const int* pvtg_node_ptr; const double* pvtg_pg; const int* pvtg_row_ptr; const double* pvtg;
    int num_rock; const int* rocktab_ptr; const double* rocktab;
    int pc_scaling;   // the product's switch for its extended record; the oracle scales whenever set_pcw gave it an array
};
FluidInput to_input(const orc_fluid_desc* d) {
    FluidInput in;
    for (int r = 0; r < d->num_pvt; ++r) {
        FluidInput::Pvt p;
        for (int i = 0; i < 5; ++i) p.pvtw[i] = d->pvtw[5 * r + i];
        for (int i = 0; i < 3; ++i) p.density[i] = d->density[3 * r + i];
        if (d->pvdg_ptr && d->pvdg) p.pvdg.assign(d->pvdg + 3 * d->pvdg_ptr[r], d->pvdg + 3 * d->pvdg_ptr[r + 1]);
        for (int n = d->pvto_node_ptr[r]; n < d->pvto_node_ptr[r + 1]; ++n) {
            PvtoNode node;
            node.rs = d->pvto_rs[n];
            for (int q = d->pvto_row_ptr[n]; q < d->pvto_row_ptr[n + 1]; ++q) {
                node.p.push_back(d->pvto[3 * q]); node.bo.push_back(d->pvto[3 * q + 1]); node.mu.push_back(d->pvto[3 * q + 2]);
            }
            p.pvto.push_back(node);
        }
        if (d->pvtg_node_ptr && d->pvtg_node_ptr[d->num_pvt] > 0)
            for (int n = d->pvtg_node_ptr[r]; n < d->pvtg_node_ptr[r + 1]; ++n) {
                PvtgNode node;
                node.pg = d->pvtg_pg[n];
                for (int q = d->pvtg_row_ptr[n]; q < d->pvtg_row_ptr[n + 1]; ++q) {
                    node.rv.push_back(d->pvtg[3 * q]); node.bg.push_back(d->pvtg[3 * q + 1]); node.mu.push_back(d->pvtg[3 * q + 2]);
                }
                p.pvtg.push_back(node);
            }
        in.pvt.push_back(p);
    }
    for (int t = 0; t < d->num_rock; ++t)
        in.rocktab.emplace_back(d->rocktab + 3 * d->rocktab_ptr[t], d->rocktab + 3 * d->rocktab_ptr[t + 1]);
    for (int s = 0; s < d->num_sat; ++s) {
        FluidInput::Sat t;
        t.swof.assign(d->swof + 4 * d->swof_ptr[s], d->swof + 4 * d->swof_ptr[s + 1]);
        t.sgof.assign(d->sgof + 4 * d->sgof_ptr[s], d->sgof + 4 * d->sgof_ptr[s + 1]);
        in.sat.push_back(t);
    }
    in.rock_pref = d->rock_pref;
    in.rock_cr = d->rock_cr;
    return in;
}
}  // namespace

extern "C" {

// point evaluation of the fluid-system / saturation functions, same output layout as opmhip_fluid_probe:
// out[8 i ..] = 1/B_w(p), 1/B_g(p), 1/B_o(p, rs) (saturated curve where rs >= RsSat(p)), RsSat(p), pcow(sw), pcgo(sg),
// mu_o(p, rs), mu_g(p)
int orc_fluid_probe(const orc_fluid_desc* d, int pr, int sr, int n, const double* p, const double* rs, const double* sw,
                    const double* sg, double* out) {
    Fluid F;
    F.init(to_input(d));
    const OilPvt& O = F.oil[pr];
    for (int i = 0; i < n; ++i) {
        double* o = out + (size_t)i * 8;
        o[0] = F.water[pr].invB(p[i]);
        o[1] = F.hasWetGas ? F.wetGas[pr].invBSat(p[i]) : F.gas[pr].invB(p[i]);
        o[3] = O.rsSat(p[i]);
        if (rs[i] >= o[3]) { o[2] = O.invBSat(p[i]); o[6] = O.viscositySat(p[i]); }
        else { o[2] = O.invB(p[i], rs[i]); o[6] = O.viscosity(p[i], rs[i]); }
        double pC[3];
        F.sat[sr].capillaryPressures(pC, sw[i], sg[i]);
        o[4] = -pC[0];
        o[5] = pC[2];
        o[7] = F.hasWetGas ? F.wetGas[pr].viscositySat(p[i]) : F.gas[pr].viscosity(p[i]);
    }
    return 0;
}

// wet-gas PVT probe: per point RvSat(p); rv >= RvSat -> saturated 1/B_g, mu_g at p ; else 1/B_g(p, rv), mu_g(p, rv)
int orc_gas_pvt_probe(const orc_fluid_desc* d, int region, int n, const double* rv, const double* p, double* mu,
                      double* invB, double* rvSat) {
    Fluid F;
    F.init(to_input(d));
    if (!F.hasWetGas) return -1;
    const WetGasPvt& G = F.wetGas[region];
    for (int i = 0; i < n; ++i) {
        rvSat[i] = G.rvSat(p[i]);
        if (rv[i] >= rvSat[i]) { mu[i] = G.viscositySat(p[i]); invB[i] = G.invBSat(p[i]); }
        else { mu[i] = G.viscosity(p[i], rv[i]); invB[i] = G.invB(p[i], rv[i]); }
    }
    return 0;
}

// oil PVT probe used to pin LiveOilPvt against tests/test_norne_pvt.cpp:
// for each point: RsSat(p); if rs >= RsSat -> saturated mu, 1/B at p ; else mu(p, rs), 1/B(p, rs)
int orc_oil_pvt_probe(const orc_fluid_desc* d, int region, int n, const double* rs, const double* p, double* mu,
                      double* invB, double* rsSat) {
    Fluid F;
    F.init(to_input(d));
    const OilPvt& O = F.oil[region];
    for (int i = 0; i < n; ++i) {
        rsSat[i] = O.rsSat(p[i]);
        if (rs[i] >= rsSat[i]) { mu[i] = O.viscositySat(p[i]); invB[i] = O.invBSat(p[i]); }
        else { mu[i] = O.viscosity(p[i], rs[i]); invB[i] = O.invB(p[i], rs[i]); }
    }
    return 0;
}

struct orc_model {
    Model M;
    // work copies of the multi-threaded solve (orc_bo_solve_mt): allocated and first-touched once, then reused - a fresh
    // 500 MB std::vector per Newton iteration was zero-filled by ONE thread and made the 16-thread "setup" slower than the
    // serial one
    Bcrs mtA, mtLU;
    std::vector<int> mtDiag;
};

orc_model* orc_bo_create(int Nb, const int* rowptr, const int* col, const double* trans, const double* area,
                         const double* thpres, const double* poro, const double* volume, const double* depth,
                         const int* pvtnum, const int* satnum, const double* rsMax, const orc_fluid_desc* fluid) {
    orc_model* h = new orc_model();
    Model& M = h->M;
    M.P.pat = wrap(Nb, rowptr, col, nullptr);
    M.P.pat.val.clear();
    const int nnzb = rowptr[Nb];
    M.P.trans.assign(trans, trans + nnzb);
    M.P.area.assign(area, area + nnzb);
    if (thpres) M.P.thpres.assign(thpres, thpres + nnzb);
    M.P.poro.assign(poro, poro + Nb);
    M.P.volume.assign(volume, volume + Nb);
    M.P.depth.assign(depth, depth + Nb);
    if (pvtnum) M.P.pvtnum.assign(pvtnum, pvtnum + Nb);
    if (satnum) M.P.satnum.assign(satnum, satnum + Nb);
    if (rsMax) M.P.rsMax.assign(rsMax, rsMax + Nb);
    M.P.fluid.init(to_input(fluid));
    M.init();
    return h;
}
void orc_bo_destroy(orc_model* h) { delete h; }

int orc_bo_set_state(orc_model* h, const double* pv, const unsigned char* meaning) {
    Model& M = h->M;
    M.pv.assign(pv, pv + M.pv.size());
    M.meaning.assign(meaning, meaning + M.meaning.size());
    std::fill(M.wasSwitched.begin(), M.wasSwitched.end(), 0);
    M.update_all_iq();
    return 0;
}
int orc_bo_get_state(orc_model* h, double* pv, unsigned char* meaning) {
    Model& M = h->M;
    std::memcpy(pv, M.pv.data(), M.pv.size() * sizeof(double));
    std::memcpy(meaning, M.meaning.data(), M.meaning.size());
    return 0;
}
int orc_bo_set_source(orc_model* h, const double* source, const double* dsource) {
    Model& M = h->M;
    if (source) M.source.assign(source, source + M.source.size());
    if (dsource) M.dsource.assign(dsource, dsource + M.dsource.size());
    return 0;
}
// per-cell intensive quantities, values and derivatives, in the product's export layout:
// fields [S_w S_o S_g | p_w p_o p_g | b_w b_o b_g | mob_w mob_o mob_g | rho_w rho_o rho_g | Rs | poro] = 17 fields x 4
int orc_bo_get_iq(orc_model* h, double* out) {
    Model& M = h->M;
    const int Nb = M.P.pat.Nb;
    for (int c = 0; c < Nb; ++c) {
        const IQT<Ev>& q = M.iqF[c];
        const Ev* f[17] = {&q.S[0], &q.S[1], &q.S[2], &q.p[0], &q.p[1], &q.p[2], &q.invB[0], &q.invB[1], &q.invB[2],
                           &q.mob[0], &q.mob[1], &q.mob[2], &q.rho[0], &q.rho[1], &q.rho[2], &q.Rs, &q.poro};
        for (int k = 0; k < 17; ++k) {
            double* o = &out[((size_t)c * 17 + k) * 4];
            o[0] = f[k]->v; o[1] = f[k]->d[0]; o[2] = f[k]->d[1]; o[3] = f[k]->d[2];
        }
    }
    return 0;
}
// extended layout (wet gas / ROCKTAB): 19 fields x 4 = the 16 of above up to Rs, then Rv, transmissibility multiplier, porosity
int orc_bo_get_iq_ext(orc_model* h, double* out) {
    Model& M = h->M;
    const int Nb = M.P.pat.Nb;
    for (int c = 0; c < Nb; ++c) {
        const IQT<Ev>& q = M.iqF[c];
        const Ev* f[19] = {&q.S[0], &q.S[1], &q.S[2], &q.p[0], &q.p[1], &q.p[2], &q.invB[0], &q.invB[1], &q.invB[2],
                           &q.mob[0], &q.mob[1], &q.mob[2], &q.rho[0], &q.rho[1], &q.rho[2], &q.Rs, &q.Rv, &q.tmult, &q.poro};
        for (int k = 0; k < 19; ++k) {
            double* o = &out[((size_t)c * 19 + k) * 4];
            o[0] = f[k]->v; o[1] = f[k]->d[0]; o[2] = f[k]->d[1]; o[3] = f[k]->d[2];
        }
    }
    return 0;
}
// per-cell extras of the problem: DRVDT cap (maxOilVaporizationFactor), rock-table index, overburden pressure; any may be NULL
int orc_bo_set_extras(orc_model* h, const double* rvMax, const int* rockNum, const double* overburden) {
    Model& M = h->M;
    const int Nb = M.P.pat.Nb;
    if (rvMax) M.P.rvMax.assign(rvMax, rvMax + Nb); else M.P.rvMax.clear();
    if (rockNum) M.P.rockNum.assign(rockNum, rockNum + Nb); else M.P.rockNum.clear();
    if (overburden) M.P.overburden.assign(overburden, overburden + Nb); else M.P.overburden.clear();
    M.update_all_iq();
    return 0;
}
// per-cell scaled maximum of the oil-water capillary pressure (PCW / SWATINIT); NULL = the tables' own
// saturation end-point scaling: cfg[7] = sat_scaling, three_point_kr, krw, kro, krg, pcw, pcg; eps: Nb x EPS_COUNT scaled end
// points (cell-major) or NULL = off
int orc_bo_set_endpoint_scaling(orc_model* h, const int* cfg, const double* eps) {
    Model& M = h->M;
    const int Nb = M.P.pat.Nb;
    M.P.eps.clear();
    if (eps) {
        M.P.eps.resize(Nb);
        for (int i = 0; i < Nb; ++i)
            for (int f = 0; f < EPS_COUNT; ++f) M.P.eps[i].v[f] = eps[(size_t)i * EPS_COUNT + f];
        EpsConfig& C = M.P.epsCfg;
        C.satScaling = cfg[0] != 0; C.threePointKr = cfg[1] != 0; C.krw = cfg[2]; C.kro = cfg[3]; C.krg = cfg[4]; C.pcw = cfg[5] != 0; C.pcg = cfg[6] != 0;
    }
    M.update_all_iq();
    return 0;
}
// the end points of a saturation region's own tables: out[EPS_COUNT]
int orc_sat_end_points(const orc_fluid_desc* fluid, int sat_region, double* out) {
    Fluid F;
    F.init(to_input(fluid));
    if (sat_region < 0 || sat_region >= (int)F.sat.size()) return -1;
    for (int f = 0; f < EPS_COUNT; ++f) out[f] = F.sat[sat_region].unscaled.v[f];
    return 0;
}
// point evaluation of the scaled saturation functions: out[6 i ..] = krw, kro, krg, pcow, pcgo, 0 at (sw, sg) of point i with
// the scaled end points eps[EPS_COUNT] (tests of the scaling's properties)
int orc_sat_probe_eps(const orc_fluid_desc* fluid, int sat_region, const int* cfg, const double* eps, int n, const double* sw, const double* sg, double* out) {
    Fluid F;
    F.init(to_input(fluid));
    if (sat_region < 0 || sat_region >= (int)F.sat.size()) return -1;
    EpsPoints P;
    for (int f = 0; f < EPS_COUNT; ++f) P.v[f] = eps[f];
    EpsConfig C;
    C.satScaling = cfg[0] != 0; C.threePointKr = cfg[1] != 0; C.krw = cfg[2]; C.kro = cfg[3]; C.krg = cfg[4]; C.pcw = cfg[5] != 0; C.pcg = cfg[6] != 0;
    for (int i = 0; i < n; ++i) {
        double kr[3], pc[3];
        F.sat[sat_region].relativePermeabilitiesEps(kr, sw[i], sg[i], P, C);
        F.sat[sat_region].capillaryPressuresEps(pc, sw[i], sg[i], P, C);
        out[6 * i] = kr[0]; out[6 * i + 1] = kr[1]; out[6 * i + 2] = kr[2]; out[6 * i + 3] = -pc[0]; out[6 * i + 4] = pc[2]; out[6 * i + 5] = 0.0;
    }
    return 0;
}

// DRSDT / DRVDT: rates per PVT region (1/s; negative = none; NULL = keyword absent), OILVAP option per region
int orc_bo_set_composition_change_limits(orc_model* h, int num_pvt, const double* drsdt, const int* drsdt_all, const double* drvdt) {
    Model& M = h->M;
    M.drsdt.clear(); M.drvdt.clear(); M.drsdtAll.clear(); M.lastRs.clear(); M.lastRv.clear();
    if (drsdt) { M.drsdt.assign(drsdt, drsdt + num_pvt); M.drsdtAll.assign(num_pvt, 0); if (drsdt_all) M.drsdtAll.assign(drsdt_all, drsdt_all + num_pvt); }
    if (drvdt) M.drvdt.assign(drvdt, drvdt + num_pvt);
    if (!drsdt) M.P.rsMax.clear();
    if (!drvdt) M.P.rvMax.clear();
    if (M.limits_active()) { M.update_composition_change_limits(); M.set_limits_for(0.0); }   // until the first begin_time_step: the caps of the state itself
    M.storageFrozen = false;
    M.update_all_iq();
    return 0;
}
// ROCKCOMP IRREVERS: minOilPressure_ = min(1e99, initial oil pressure) (eclgenericproblem.cc:165, eclproblem.hh:2293-2294)
int orc_bo_set_irreversible_compaction(orc_model* h, int enable) {
    Model& M = h->M;
    M.P.minOilPressure.clear();
    if (enable) {
        const int Nb = M.P.pat.Nb;
        M.P.minOilPressure.assign(Nb, 1e99);
        for (int c = 0; c < Nb; ++c) M.P.minOilPressure[c] = std::min(M.P.minOilPressure[c], M.iqV[c].p[OIL]);
    }
    M.update_all_iq();
    return 0;
}
// VAPPARS: vap1 (oil vaporisation propensity, on RvSat), vap2 (gas re-dissolution, on RsSat); enable = 0: keyword not in force.
// maxOilSaturation_ starts from the state now present (eclproblem.hh:2291-2292)
int orc_bo_set_vappars(orc_model* h, int enable, double vap1, double vap2) {
    Model& M = h->M;
    M.P.maxOilSaturation.clear();
    M.P.vapPar1 = vap1; M.P.vapPar2 = vap2;
    if (enable) {
        const int Nb = M.P.pat.Nb;
        M.P.maxOilSaturation.assign(Nb, 0.0);
        for (int c = 0; c < Nb; ++c) M.P.maxOilSaturation[c] = std::max(M.P.maxOilSaturation[c], M.iqV[c].S[OIL]);
    }
    M.update_all_iq();
    return 0;
}
// water-induced compaction: ntab tables (0: off) over (pressure, Sw delta); pv / tr row-major by pressure node; tr may be NULL.
// maxWaterSaturation_ and the initial saturation start from the state now present (eclproblem.hh:2289-2290, initialFluidStates_)
int orc_bo_set_water_compaction(orc_model* h, int ntab, const int* np, const int* ns, const double* p, const double* sw, const double* pv, const double* tr) {
    Model& M = h->M;
    M.P.rock2dPoro.clear(); M.P.rock2dTrans.clear(); M.P.maxWaterSaturation.clear(); M.P.initialSw.clear();
    size_t op = 0, os = 0, ov = 0;
    for (int t = 0; t < ntab; ++t) {
        Tab2D a, b;
        a.xs.assign(p + op, p + op + np[t]);
        for (int i = 0; i < np[t]; ++i) {
            a.ys.emplace_back(sw + os, sw + os + ns[t]);
            a.vs.emplace_back(pv + ov + (size_t)i * ns[t], pv + ov + (size_t)(i + 1) * ns[t]);
        }
        M.P.rock2dPoro.push_back(a);
        if (tr) {
            b.xs = a.xs; b.ys = a.ys;
            for (int i = 0; i < np[t]; ++i) b.vs.emplace_back(tr + ov + (size_t)i * ns[t], tr + ov + (size_t)(i + 1) * ns[t]);
            M.P.rock2dTrans.push_back(b);
        }
        op += np[t]; os += ns[t]; ov += (size_t)np[t] * ns[t];
    }
    if (ntab > 0) {
        const int Nb = M.P.pat.Nb;
        M.P.maxWaterSaturation.assign(Nb, 0.0);
        M.P.initialSw.resize(Nb);
        for (int c = 0; c < Nb; ++c) {
            M.P.initialSw[c] = M.iqV[c].S[WATER];
            M.P.maxWaterSaturation[c] = std::max(M.P.maxWaterSaturation[c], M.iqV[c].S[WATER]);
        }
    }
    M.update_all_iq();
    return 0;
}
// relative-permeability hysteresis: kr_model 0 | 1 (EHYSTR item 2; < 0: off), imbnum per cell (0-based), eps_imb: Nb x EPS_COUNT
// scaled end points of the imbibition curves (cell-major) or NULL.  The turning points start at 2.0 ("nothing seen"): the first
// begin_time_step sets them, as the reference's first beginTimeStep does.
int orc_bo_set_hysteresis(orc_model* h, int kr_model, const int* imbnum, const double* eps_imb) {
    Model& M = h->M;
    const int Nb = M.P.pat.Nb;
    M.P.hyst.clear(); M.P.imbnum.clear(); M.P.epsImb.clear();
    M.P.hystKrModel = kr_model;
    if (kr_model >= 0) {
        if (!imbnum) return -1;
        for (int c = 0; c < Nb; ++c)
            if (imbnum[c] < 0 || imbnum[c] >= (int)M.P.fluid.sat.size()) return -1;
        M.P.imbnum.assign(imbnum, imbnum + Nb);
        M.P.hyst.assign(Nb, HystCell());
        if (eps_imb) {
            M.P.epsImb.resize(Nb);
            for (int i = 0; i < Nb; ++i)
                for (int f = 0; f < EPS_COUNT; ++f) M.P.epsImb[i].v[f] = eps_imb[(size_t)i * EPS_COUNT + f];
        }
    }
    M.update_all_iq();
    return 0;
}
// the hysteresis state, per cell: krnSwMdc and deltaSwImbKrn of the oil-water and of the gas-oil system (4 arrays, any NULL)
int orc_bo_get_hysteresis(orc_model* h, double* sw_ow, double* delta_ow, double* sw_go, double* delta_go) {
    Model& M = h->M;
    if (M.P.hyst.empty()) return -1;
    for (int c = 0; c < M.P.pat.Nb; ++c) {
        const HystCell& q = M.P.hyst[c];
        if (sw_ow) sw_ow[c] = q.krnSwMdcOw;
        if (delta_ow) delta_ow[c] = q.deltaSwImbKrnOw;
        if (sw_go) sw_go[c] = q.krnSwMdcGo;
        if (delta_go) delta_go[c] = q.deltaSwImbKrnGo;
    }
    return 0;
}
// restart (initHysteresisParams, ebos/ecloutputblackoilmodule.hh:569-589 -> setOilWaterHysteresisParams / setGasOilHysteresisParams):
// the turning points handed in, the shifts recomputed from them
int orc_bo_set_hysteresis_params(orc_model* h, const double* sw_ow, const double* sw_go) {
    Model& M = h->M;
    if (M.P.hyst.empty()) return -1;
    const bool scaled = !M.P.eps.empty();
    for (int c = 0; c < M.P.pat.Nb; ++c) {
        HystCell q;   // from "nothing seen": update() takes any value below 2
        const int sr = M.P.satnum.empty() ? 0 : M.P.satnum[c];
        const SatFunc& imb = M.P.fluid.sat[M.P.imbnum[c]];
        EpsPoints own;
        const EpsPoints* scI = nullptr;
        if (scaled) { if (M.P.epsImb.empty()) { own = imb.unscaled; scI = &own; } else scI = &M.P.epsImb[c]; }
        M.P.fluid.sat[sr].hystSee(q, sw_ow[c], sw_go[c], imb, scaled ? &M.P.eps[c] : nullptr, scI, M.P.epsCfg);
        M.P.hyst[c] = q;
    }
    M.update_all_iq();
    return 0;
}
int orc_bo_get_max_water_saturation(orc_model* h, double* out) {
    Model& M = h->M;
    for (int c = 0; c < M.P.pat.Nb; ++c) out[c] = M.P.maxWaterSaturation.empty() ? 0.0 : M.P.maxWaterSaturation[c];
    return 0;
}
int orc_bo_get_max_oil_saturation(orc_model* h, double* out) {
    Model& M = h->M;
    for (int c = 0; c < M.P.pat.Nb; ++c) out[c] = M.P.maxOilSaturation.empty() ? 0.0 : M.P.maxOilSaturation[c];
    return 0;
}
// BlackoilModelEbos::relativeChange (flow/BlackoilModelEbos.hpp:431-510): the state now present against an old time level
// handed in, the cells in index order, every term added to the running sums as the reference adds it
int orc_bo_relative_change(orc_model* h, const double* pvOld, const uint8_t* meaningOld, double* out) {
    Model& M = h->M;
    const int Nb = M.P.pat.Nb;
    double resultDelta = 0.0, resultDenom = 0.0;
    for (int c = 0; c < Nb; ++c) {
        const double* a = &M.pv[(size_t)c * 3];
        const double* b = &pvOld[(size_t)c * 3];
        const double pressureNew = a[PV_P];
        double saturationsNew[3] = {0.0, 0.0, 0.0};
        double oilSaturationNew = 1.0;
        saturationsNew[WATER] = a[PV_SW];
        oilSaturationNew -= saturationsNew[WATER];
        if (M.meaning[c] == Sw_po_Sg) {
            saturationsNew[GAS] = a[PV_X];
            oilSaturationNew -= saturationsNew[GAS];
        }
        saturationsNew[OIL] = oilSaturationNew;
        const double pressureOld = b[PV_P];
        double saturationsOld[3] = {0.0, 0.0, 0.0};
        double oilSaturationOld = 1.0;
        const double tmp = pressureNew - pressureOld;
        resultDelta += tmp * tmp;
        resultDenom += pressureNew * pressureNew;
        saturationsOld[WATER] = b[PV_SW];
        oilSaturationOld -= saturationsOld[WATER];
        if (meaningOld[c] == Sw_po_Sg) {
            saturationsOld[GAS] = b[PV_X];
            oilSaturationOld -= saturationsOld[GAS];
        }
        saturationsOld[OIL] = oilSaturationOld;
        for (int ph = 0; ph < 3; ++ph) {
            const double tmpSat = saturationsNew[ph] - saturationsOld[ph];
            resultDelta += tmpSat * tmpSat;
            resultDenom += saturationsNew[ph] * saturationsNew[ph];
        }
    }
    *out = resultDenom > 0.0 ? resultDelta / resultDenom : 0.0;
    return 0;
}
int orc_bo_begin_time_step(orc_model* h, double dt) { h->M.begin_time_step(dt); return 0; }
// trackers, for tests: out[0..Nb) lastRs, [Nb..2Nb) lastRv, [2Nb..3Nb) minOilPressure (0 where not kept)
int orc_bo_get_trackers(orc_model* h, double* out) {
    Model& M = h->M;
    const int Nb = M.P.pat.Nb;
    for (int c = 0; c < Nb; ++c) {
        out[c] = M.lastRs.empty() ? 0.0 : M.lastRs[c];
        out[Nb + c] = M.lastRv.empty() ? 0.0 : M.lastRv[c];
        out[2 * Nb + c] = M.P.minOilPressure.empty() ? 0.0 : M.P.minOilPressure[c];
    }
    return 0;
}

int orc_bo_set_pcw(orc_model* h, const double* pcw) {
    Model& M = h->M;
    const int Nb = M.P.pat.Nb;
    if (pcw) M.P.pcw.assign(pcw, pcw + Nb); else M.P.pcw.clear();
    M.update_all_iq();
    return 0;
}
int orc_bo_assemble(orc_model* h, double dt, int iteration, double* jac, double* residual) {
    Model& M = h->M;
    M.assemble(dt, iteration);
    if (jac) std::memcpy(jac, M.J.val.data(), M.J.val.size() * sizeof(double));
    if (residual) std::memcpy(residual, M.residual.data(), M.residual.size() * sizeof(double));
    return 0;
}
// true-IMPES weights of every cell from the storage term's derivatives at the present state (getQuasiImpesWeights.hpp:89-128)
int orc_bo_true_impes_weights(orc_model* h, double dt, double* w) {
    Model& M = h->M;
    const int Nb = M.P.pat.Nb;
    for (int c = 0; c < Nb; ++c) {
        Ev st[3];
        compute_storage(M.iqF[c], st, M.P.fluid.hasWetGas);
        double dS[BS][BS];
        for (int e = 0; e < BS; ++e)
            for (int v = 0; v < BS; ++v) dS[e][v] = st[e].d[v];
        true_impes_weights_cell(dS, M.P.volume[c] / dt, &w[(size_t)c * BS]);
    }
    return 0;
}
// the Jacobian / residual of the last orc_bo_assemble, without assembling again
int orc_bo_assemble_fetch(orc_model* h, double* jac, double* residual) {
    Model& M = h->M;
    if (jac) std::memcpy(jac, M.J.val.data(), M.J.val.size() * sizeof(double));
    if (residual) std::memcpy(residual, M.residual.data(), M.residual.size() * sizeof(double));
    return 0;
}
// out[0..2] R_sum, [3..5] maxCoeff, [6..8] B_avg, [9] pvSum, [10] cnvErrorPv, [11..13] CNV, [14..16] MB
int orc_bo_convergence(orc_model* h, double dt, double tol_cnv, double* out) {
    Model::Convergence c = h->M.convergence(dt, tol_cnv);
    for (int e = 0; e < 3; ++e) { out[e] = c.R_sum[e]; out[3 + e] = c.maxCoeff[e]; out[6 + e] = c.B_avg[e]; out[11 + e] = c.CNV[e]; out[14 + e] = c.MB[e]; }
    out[9] = c.pvSum;
    out[10] = c.cnvErrorPv;
    return 0;
}
int orc_bo_update(orc_model* h, const double* dx) { return h->M.update(dx); }
// EclProblem::endTimeStep (drift part) after an accepted time step; enable < 0 leaves the switch as it is
int orc_bo_end_time_step(orc_model* h, double dt) { h->M.end_time_step(dt); return 0; }
int orc_bo_set_drift_compensation(orc_model* h, int enable, double max_compensation) {
    h->M.enableDriftCompensation = enable != 0;
    if (max_compensation > 0.0) h->M.maxCompensation = max_compensation;
    return 0;
}
int orc_bo_get_drift(orc_model* h, double* out) { std::memcpy(out, h->M.drift.data(), h->M.drift.size() * sizeof(double)); return 0; }
// threads for the OpenMP loops (assembly, IQ update, the *_mt solver below); returns the previous maximum
int orc_set_threads(int n) {
#ifdef _OPENMP
    const int old = omp_get_max_threads();
    if (n > 0) omp_set_num_threads(n);
    return old;
#else
    (void)n;
    return 1;
#endif
}
// solveJacobianSystem as `threads` MPI ranks of Flow would run it on one host: natural order, the rows cut into
// `threads` contiguous ranges, block-Jacobi ILU0 (one range per thread), threaded SpMV and scalar products.
// bench.py's multi-core CPU baseline only.
int orc_bo_solve_mt(orc_model* h, double* x, double tol, int maxit, double w, int relax_mode, int threads, orc_result* out) {
    using clk = std::chrono::steady_clock;
    Model& M = h->M;
    const int Nb = M.P.pat.Nb;
    const size_t n = (size_t)Nb * BS;
    if (threads < 1) threads = 1;
    std::vector<int> sub(threads + 1);
    for (int s = 0; s <= threads; ++s) sub[s] = (int)((long long)Nb * s / threads);
    if (h->mtA.Nb != M.J.Nb) {   // first call: pattern copies and the diagonal positions, once (the pattern never changes)
        for (Bcrs* D : {&h->mtA, &h->mtLU}) {
            D->Nb = M.J.Nb; D->rowptr = M.J.rowptr; D->col = M.J.col;
            D->val.resize(M.J.val.size());
        }
        h->mtDiag = diag_index(M.J);
    }
    auto t0 = clk::now();
    // values copied (threaded): the zero-diagonal fix and the factorisation work on copies, as in the 1-thread path
    auto copy_vals = [](const Bcrs& S, Bcrs& D) {
        const long long m = (long long)S.val.size();
#pragma omp parallel for schedule(static)
        for (long long i = 0; i < m; ++i) D.val[i] = S.val[i];
    };
    Bcrs& A = h->mtA;
    Bcrs& LU = h->mtLU;
    copy_vals(M.J, A);
    check_zero_diagonal(A);
    copy_vals(A, LU);
    const std::vector<int>& dg = h->mtDiag;
    const int rc = bilu0_decompose_bj(LU, sub, &dg);
    if (rc != 0) return rc;
    auto t1 = clk::now();
    auto prec = [&](const double* d, double* v) { ilu0_apply_bj(LU, dg, sub, d, v, w, relax_mode); };
    auto op = [&](const double* xin, double* y) { spmv_mt(A, xin, y); };
    SolveResult r = bicgstab_mt(n, M.residual.data(), x, prec, op, tol, maxit);
    auto t2 = clk::now();
    if (out) {
        out->iterations = r.iterations;
        out->converged = r.converged;
        out->reduction = r.reduction;
        out->conv_rate = r.conv_rate;
        out->it = r.it;
        out->t_factor = std::chrono::duration<double>(t1 - t0).count();
        out->t_solve = std::chrono::duration<double>(t2 - t1).count();
        out->num_colors = 0;
    }
    return 0;
}

// ILU0-BiCGStab on the model's own Jacobian and residual (solveJacobianSystem, BlackoilModelEbos.hpp:523-544)
int orc_bo_solve(orc_model* h, double* x, double tol, int maxit, double w, int relax_mode, int reorder, int nsub,
                 const int* sub_start, orc_result* out) {
    Model& M = h->M;
    return orc_solve(M.J.Nb, M.J.rowptr.data(), M.J.col.data(), M.J.val.data(), M.residual.data(), x, tol, maxit, w,
                     relax_mode, reorder, 1, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nsub, sub_start, out);
}

}  // extern "C"
