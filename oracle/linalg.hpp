// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or
// executed by the product path (opm-autodiff_amd/, libopmhip.so).  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, as the checker.
//
// CPU restatement (C++17, scalar, single thread) of the reference's block-sparse linear
// algebra for 3x3 double blocks.  Each routine cites the reference lines it follows
// (paths relative to /root/reference).  dune-istl / dune-common are NOT in the reference
// tree: where the arithmetic lives there, the operation order is restated
// from the in-tree twin of the same algorithm and tagged "dune order".
//
// Parity pin: tests/test_oracle_linalg.py checks this file against the reference's own
// known answers (tests/matr33.txt + rhs3.txt -> three expected vectors, matr33rep/rhs3rep,
// the LUe == Ae identity of tests/test_milu.cpp).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>

namespace orc {

constexpr int BS = 3;
constexpr int BB = 9;

// Block-CSR matrix: what BdaBridge hands to a backend (bda/BdaBridge.cpp:232):
// rowptr[Nb+1], col[nnzb] ascending in a row, val[nnzb*9] row-major blocks
// (bda/cusparseSolverBackend.cu:48, bda/openclKernels.cpp:195).
struct Bcrs {
    int Nb = 0;
    std::vector<int> rowptr, col;
    std::vector<double> val;
    int nnzb() const { return rowptr.empty() ? 0 : rowptr[Nb]; }
};

// ---- dense 3x3 block kernels, dune-common DenseMatrix operation order ----------------
// y -= A x   (DenseMatrix::mmv: for i, for j: y[i] -= A[i][j]*x[j])
inline void blk_mmv(const double* A, const double* x, double* y) {
    for (int i = 0; i < BS; ++i)
        for (int j = 0; j < BS; ++j) y[i] -= A[i * BS + j] * x[j];
}
// y += A x   (DenseMatrix::umv)
inline void blk_umv(const double* A, const double* x, double* y) {
    for (int i = 0; i < BS; ++i)
        for (int j = 0; j < BS; ++j) y[i] += A[i * BS + j] * x[j];
}
// y = A x    (DenseMatrix::mv: y[i] = 0 then accumulate)
inline void blk_mv(const double* A, const double* x, double* y) {
    for (int i = 0; i < BS; ++i) {
        y[i] = 0.0;
        for (int j = 0; j < BS; ++j) y[i] += A[i * BS + j] * x[j];
    }
}
// A <- A * M  (DenseMatrix::rightmultiply; used as L_ij = A_ij * A_jj^-1,
// linalg/ParallelOverlappingILU0.hpp:462)
inline void blk_rightmultiply(double* A, const double* M) {
    double C[BB];
    std::memcpy(C, A, sizeof C);
    for (int i = 0; i < BS; ++i)
        for (int j = 0; j < BS; ++j) {
            double s = 0.0;
            for (int k = 0; k < BS; ++k) s += C[i * BS + k] * M[k * BS + j];
            A[i * BS + j] = s;
        }
}
// B <- M * B  (DenseMatrix::leftmultiply; linalg/ParallelOverlappingILU0.hpp:471-472)
inline void blk_leftmultiply(double* B, const double* M) {
    double C[BB];
    std::memcpy(C, B, sizeof C);
    for (int i = 0; i < BS; ++i)
        for (int j = 0; j < BS; ++j) {
            double s = 0.0;
            for (int k = 0; k < BS; ++k) s += M[i * BS + k] * C[k * BS + j];
            B[i * BS + j] = s;
        }
}
// Closed-form 3x3 inverse: same expression tree as Opm::Detail::Inverter<3>
// (linalg/MatrixBlock.hpp:722-747), which the file says mirrors Dune::DenseMatrix.
inline void blk_invert(const double* m, double* inv) {
    const double m00m11 = m[0] * m[4], m00m12 = m[0] * m[5];
    const double m01m10 = m[1] * m[3], m02m10 = m[2] * m[3];
    const double m01m20 = m[1] * m[6], m02m20 = m[2] * m[6];
    const double det = (m00m11 * m[8] - m00m12 * m[7] - m01m10 * m[8] + m02m10 * m[7] +
                        m01m20 * m[5] - m02m20 * m[4]);
    const double rdet = 1.0 / det;
    inv[0] = (m[4] * m[8] - m[5] * m[7]) * rdet;
    inv[1] = -(m[1] * m[8] - m[2] * m[7]) * rdet;
    inv[2] = (m[1] * m[5] - m[2] * m[4]) * rdet;
    inv[3] = -(m[3] * m[8] - m[5] * m[6]) * rdet;
    inv[4] = (m[0] * m[8] - m02m20) * rdet;
    inv[5] = -(m00m12 - m02m10) * rdet;
    inv[6] = (m[3] * m[7] - m[4] * m[6]) * rdet;
    inv[7] = -(m[0] * m[7] - m01m20) * rdet;
    inv[8] = (m00m11 - m01m10) * rdet;
}

// ---- y = A x : dune-istl BCRSMatrix::mv order (row: y_i = 0; for each block umv) ------
// call site: Dune::MatrixAdapter / WellModelMatrixAdapter::apply, linalg/WellOperators.hpp:127-138
inline void spmv(const Bcrs& A, const double* x, double* y) {
    for (int i = 0; i < A.Nb; ++i) {
        double acc[BS] = {0.0, 0.0, 0.0};
        for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k)
            blk_umv(&A.val[(size_t)k * BB], &x[(size_t)A.col[k] * BS], acc);
        for (int r = 0; r < BS; ++r) y[(size_t)i * BS + r] = acc[r];
    }
}

inline double dot(const double* a, const double* b, size_t n) {
    double s = 0.0;
    for (size_t i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}
inline double norm2(const double* a, size_t n) { return std::sqrt(dot(a, a, n)); }

// checkZeroDiagonal: exact 0.0 on a diagonal block's diagonal -> 1e-15
// (bda/BdaBridge.cpp:125-161).  Returns the number of replaced zeros.
inline int check_zero_diagonal(Bcrs& A) {
    int n = 0;
    for (int i = 0; i < A.Nb; ++i)
        for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k)
            if (A.col[k] == i)
                for (int r = 0; r < BS; ++r) {
                    double& v = A.val[(size_t)k * BB + r * BS + r];
                    if (v == 0.0) { v = 1e-15; ++n; }
                }
    return n;
}

inline std::vector<int> diag_index(const Bcrs& A) {
    std::vector<int> d(A.Nb, -1);
    for (int i = 0; i < A.Nb; ++i)
        for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k)
            if (A.col[k] == i) d[i] = k;
    return d;
}

// ---- block ILU0, in place, natural row order ------------------------------------------
// Follows detail::ghost_last_bilu0_decomposition (linalg/ParallelOverlappingILU0.hpp:439-494),
// which for interiorSize == N is Dune::bilu0_decomposition (call site :1025).  On exit the
// strict lower part holds L (unit diagonal implied), the strict upper part U, and the diagonal
// block holds its INVERSE.  Rows >= interiorSize are left untouched (ghost rows).
// Returns 0, or -(i+1) if row i has no diagonal block ("diagonal entry missing", :484-485).
inline int bilu0_decompose(Bcrs& A, int interiorSize) {
    const std::vector<int> dg = diag_index(A);
    for (int i = 0; i < interiorSize; ++i) {
        const int endi = A.rowptr[i + 1];
        int ij = A.rowptr[i];
        for (; ij < endi && A.col[ij] < i; ++ij) {
            const int j = A.col[ij];
            const int jj = dg[j];
            if (jj < 0) return -(j + 1);
            double* Lij = &A.val[(size_t)ij * BB];
            blk_rightmultiply(Lij, &A.val[(size_t)jj * BB]);  // A_ij * A_jj^-1 (stored inverse)
            const int endj = A.rowptr[j + 1];
            int jk = jj + 1, ik = ij + 1;
            while (ik < endi && jk < endj) {
                if (A.col[ik] == A.col[jk]) {
                    double B[BB];
                    std::memcpy(B, &A.val[(size_t)jk * BB], sizeof B);
                    blk_leftmultiply(B, Lij);
                    double* Aik = &A.val[(size_t)ik * BB];
                    for (int e = 0; e < BB; ++e) Aik[e] -= B[e];
                    ++ik; ++jk;
                } else if (A.col[ik] < A.col[jk]) ++ik;
                else ++jk;
            }
        }
        if (ij >= endi || A.col[ij] != i) return -(i + 1);
        double inv[BB];
        blk_invert(&A.val[(size_t)ij * BB], inv);
        std::memcpy(&A.val[(size_t)ij * BB], inv, sizeof inv);
    }
    return 0;
}

// ---- ILU0 apply -------------------------------------------------------------------------
// mode 0 ("dune"):   v = w * U^-1 L^-1 d, scaling after both sweeps
//                    (linalg/ParallelOverlappingILU0.hpp:848-903; L sweep :867-879 with rhs -= L_ij v_j
//                    in ascending column order, U sweep :881-895 where the reference walks its reversed
//                    CRS, i.e. DESCENDING column order, then inv.mv; scale :899-901).
// mode 1 ("opencl"): relaxation folded into the backward sweep, x_i = w * D_i^-1 (y_i - sum U_ij x_j),
//                    with already-relaxed x_j on the right (bda/openclKernels.cpp:301-383, w=0.9 at :328).
//                    Column order ascending as in the kernel's block loop.
// LU is the output of bilu0_decompose.  d and v may not alias.
inline void ilu0_apply(const Bcrs& LU, const std::vector<int>& dg, int interiorSize, const double* d,
                       double* v, double w, int mode) {
    for (int i = 0; i < interiorSize; ++i) {
        double rhs[BS] = {d[(size_t)i * BS], d[(size_t)i * BS + 1], d[(size_t)i * BS + 2]};
        for (int k = LU.rowptr[i]; k < dg[i]; ++k)
            blk_mmv(&LU.val[(size_t)k * BB], &v[(size_t)LU.col[k] * BS], rhs);
        for (int r = 0; r < BS; ++r) v[(size_t)i * BS + r] = rhs[r];
    }
    for (int i = interiorSize - 1; i >= 0; --i) {
        double rhs[BS] = {v[(size_t)i * BS], v[(size_t)i * BS + 1], v[(size_t)i * BS + 2]};
        if (mode == 0) {
            for (int k = LU.rowptr[i + 1] - 1; k > dg[i]; --k)
                blk_mmv(&LU.val[(size_t)k * BB], &v[(size_t)LU.col[k] * BS], rhs);
        } else {
            for (int k = dg[i] + 1; k < LU.rowptr[i + 1]; ++k)
                blk_mmv(&LU.val[(size_t)k * BB], &v[(size_t)LU.col[k] * BS], rhs);
        }
        double out[BS];
        blk_mv(&LU.val[(size_t)dg[i] * BB], rhs, out);
        for (int r = 0; r < BS; ++r) v[(size_t)i * BS + r] = (mode == 1) ? w * out[r] : out[r];
    }
    if (mode == 0 && w != 1.0)
        for (size_t e = 0; e < (size_t)interiorSize * BS; ++e) v[e] *= w;
}

// ---- the product after an ILU0 application, formed from the backward sweep's row sums -----------------
// libopmhip's opmhip_config.half_product (include/opmhip.h): on a pattern without triangles no elimination step of bilu0_decompose
// touches an entry right of the diagonal (the update above needs (i,j), (j,k) and (i,k): k == i alone), so the strict upper part of LU
// equals that of A bit for bit and u_i = sum_{j>i} U_ij x_j - formed by the backward sweep anyway - is the upper part of (A x)_i.
// ilu0_apply_u is ilu0_apply (same statements, same order, the same v) that also leaves those row sums in u: the same rounded products
// U_ij x_j, added up from 0 in the sweep's column order (x = the sweep's own result: unscaled in mode 0, relaxed in mode 1).
// spmv_rest forms y_i = (sum over the entries NOT in the ILU's U part - columns <= i, and columns of another subdomain where `owner`
// is given - ascending, of A_ik x_k) + s u_i, s = the factor x carries beside the sweep's own result (w in mode 0, 1 in mode 1).
// This restates the order libopmhip's kernels use (csrc/solver.hip: chain_sweep<.., UA>, k_spmv_pipe_st<.., UADD>); the reference runs
// the sweep and then the whole product (bda/cusparseSolverBackend.cu:103-118).  is_upper_alias says whether the property holds.
inline void ilu0_apply_u(const Bcrs& LU, const std::vector<int>& dg, int interiorSize, const double* d, double* v, double w, int mode, double* u) {
    for (int i = 0; i < interiorSize; ++i) {
        double rhs[BS] = {d[(size_t)i * BS], d[(size_t)i * BS + 1], d[(size_t)i * BS + 2]};
        for (int k = LU.rowptr[i]; k < dg[i]; ++k)
            blk_mmv(&LU.val[(size_t)k * BB], &v[(size_t)LU.col[k] * BS], rhs);
        for (int r = 0; r < BS; ++r) v[(size_t)i * BS + r] = rhs[r];
    }
    auto step = [&](int k, double* rhs, double* us) {
        const double* A = &LU.val[(size_t)k * BB];
        const double* x = &v[(size_t)LU.col[k] * BS];
        for (int r = 0; r < BS; ++r)
            for (int c = 0; c < BS; ++c) {
                const double p = A[r * BS + c] * x[c];
                rhs[r] -= p;
                us[r] += p;
            }
    };
    for (int i = interiorSize - 1; i >= 0; --i) {
        double rhs[BS] = {v[(size_t)i * BS], v[(size_t)i * BS + 1], v[(size_t)i * BS + 2]};
        double us[BS] = {0.0, 0.0, 0.0};
        if (mode == 0) {
            for (int k = LU.rowptr[i + 1] - 1; k > dg[i]; --k) step(k, rhs, us);
        } else {
            for (int k = dg[i] + 1; k < LU.rowptr[i + 1]; ++k) step(k, rhs, us);
        }
        double out[BS];
        blk_mv(&LU.val[(size_t)dg[i] * BB], rhs, out);
        for (int r = 0; r < BS; ++r) {
            v[(size_t)i * BS + r] = (mode == 1) ? w * out[r] : out[r];
            u[(size_t)i * BS + r] = us[r];
        }
    }
    if (mode == 0 && w != 1.0)
        for (size_t e = 0; e < (size_t)interiorSize * BS; ++e) v[e] *= w;
}
inline void spmv_rest(const Bcrs& A, const double* x, const double* u, double s, double* y, const int* owner = nullptr) {
    for (int i = 0; i < A.Nb; ++i) {
        double acc[BS] = {0.0, 0.0, 0.0};
        for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k)
            if (A.col[k] <= i || (owner && owner[A.col[k]] != owner[i]))
                blk_umv(&A.val[(size_t)k * BB], &x[(size_t)A.col[k] * BS], acc);
        for (int r = 0; r < BS; ++r) y[(size_t)i * BS + r] = acc[r] + s * u[(size_t)i * BS + r];
    }
}
// does any elimination step of the block ILU0 touch an entry right of the diagonal?  (pattern only)
inline bool is_upper_alias(const Bcrs& A) {
    const std::vector<int> dg = diag_index(A);
    for (int i = 0; i < A.Nb; ++i)
        for (int ij = A.rowptr[i]; ij < A.rowptr[i + 1] && A.col[ij] < i; ++ij) {
            const int j = A.col[ij];
            int jk = dg[j] + 1, ik = ij + 1;
            while (ik < A.rowptr[i + 1] && jk < A.rowptr[j + 1]) {
                if (A.col[ik] == A.col[jk]) { if (A.col[ik] > i) return false; ++ik; ++jk; }
                else if (A.col[ik] < A.col[jk]) ++ik;
                else ++jk;
            }
        }
    return true;
}

// ---- reorderings (accelerator path) ----------------------------------------------------
struct Reordering {
    std::vector<int> toOrder, fromOrder, rowsPerColor;  // as bda/Reorder.cpp
    int numColors() const { return (int)rowsPerColor.size(); }
};

inline void csr_to_csc(const Bcrs& A, std::vector<int>& cptr, std::vector<int>& ridx) {
    // pattern transpose, bda/Reorder.cpp:333-366
    cptr.assign(A.Nb + 1, 0);
    ridx.resize(A.nnzb());
    for (int k = 0; k < A.nnzb(); ++k) cptr[A.col[k] + 1]++;
    std::partial_sum(cptr.begin(), cptr.end(), cptr.begin());
    std::vector<int> w(cptr.begin(), cptr.end() - 1);
    for (int i = 0; i < A.Nb; ++i)
        for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k) ridx[w[A.col[k]]++] = i;
}

// Level scheduling (bda/Reorder.cpp:266-318, Saad 11.6.3): level(i) = 1 + max level of the rows j<i
// that row i references; rows of one level are mutually independent and keep their natural
// relative order, so the reordered ILU0 equals the natural-order ILU0 up to the permutation.
inline Reordering level_schedule(const Bcrs& A) {
    Reordering R;
    std::vector<int> level(A.Nb, 0);
    int nlev = 0;
    for (int i = 0; i < A.Nb; ++i) {
        int l = 0;
        for (int k = A.rowptr[i]; k < A.rowptr[i + 1] && A.col[k] < i; ++k)
            l = std::max(l, level[A.col[k]] + 1);
        level[i] = l;
        nlev = std::max(nlev, l + 1);
    }
    R.rowsPerColor.assign(nlev, 0);
    for (int i = 0; i < A.Nb; ++i) R.rowsPerColor[level[i]]++;
    std::vector<int> start(nlev + 1, 0);
    for (int l = 0; l < nlev; ++l) start[l + 1] = start[l] + R.rowsPerColor[l];
    R.toOrder.resize(A.Nb);
    R.fromOrder.resize(A.Nb);
    for (int i = 0; i < A.Nb; ++i) {
        const int p = start[level[i]]++;
        R.toOrder[i] = p;
        R.fromOrder[p] = i;
    }
    return R;
}

// Deterministic stand-in for the reference's std::random_device-seeded weights
// (bda/Reorder.cpp:35-44): splitmix64 of the row index.
inline uint32_t jp_weight(uint32_t i) {
    uint64_t z = (uint64_t)i + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(z & 0x7fffffffu);
}

// Graph colouring.  kind 0: Jones-Plassmann rounds exactly as colorBlockedNodes
// (bda/Reorder.cpp:59-172; a node takes colour c if no neighbour has colour c and it holds the
// strict maximum weight among its uncoloured neighbours, looking at both row and column
// neighbours), with deterministic weights.  kind 1: greedy first-fit in natural order (2 colours =
// red-black on a Cartesian 7-point grid; the CPU side's Welsh-Powell gives the same there,
// tests/test_graphcoloring.cpp:29-95).  Rows keep natural order inside a colour (colorsToReordering,
// bda/Reorder.cpp:212-226).
inline Reordering graph_color(const Bcrs& A, int kind) {
    std::vector<int> cptr, ridx;
    csr_to_csc(A, cptr, ridx);
    std::vector<int> colors(A.Nb, -1);
    int ncol = 0;
    if (kind == 0) {
        std::vector<uint32_t> wgt(A.Nb);
        for (int i = 0; i < A.Nb; ++i) wgt[i] = jp_weight((uint32_t)i);
        int left = A.Nb;
        for (int c = 0; left > 0; ++c) {
            for (int i = 0; i < A.Nb; ++i) {
                if (colors[i] != -1) continue;
                bool isMax = true;
                auto scan = [&](const int* idx, int b, int e) {
                    for (int k = b; k < e && isMax; ++k) {
                        const int j = idx[k];
                        const int jc = colors[j];
                        if ((jc != -1 && jc != c) || j == i) continue;
                        if (jc == c) { isMax = false; break; }
                        // ties broken by index so that the loop always terminates
                        if (wgt[i] < wgt[j] || (wgt[i] == wgt[j] && i < j)) isMax = false;
                    }
                };
                scan(A.col.data(), A.rowptr[i], A.rowptr[i + 1]);
                scan(ridx.data(), cptr[i], cptr[i + 1]);
                if (isMax) { colors[i] = c; --left; }
            }
            ncol = c + 1;
        }
    } else {
        std::vector<char> used;
        for (int i = 0; i < A.Nb; ++i) {
            used.assign(ncol + 1, 0);
            for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k)
                if (colors[A.col[k]] >= 0) used[colors[A.col[k]]] = 1;
            for (int k = cptr[i]; k < cptr[i + 1]; ++k)
                if (colors[ridx[k]] >= 0) used[colors[ridx[k]]] = 1;
            int c = 0;
            while (used[c]) ++c;
            colors[i] = c;
            ncol = std::max(ncol, c + 1);
        }
    }
    Reordering R;
    R.rowsPerColor.assign(ncol, 0);
    R.toOrder.resize(A.Nb);
    R.fromOrder.resize(A.Nb);
    int p = 0;
    for (int c = 0; c < ncol; ++c)
        for (int i = 0; i < A.Nb; ++i)
            if (colors[i] == c) {
                R.rowsPerColor[c]++;
                R.toOrder[i] = p;
                R.fromOrder[p] = i;
                ++p;
            }
    return R;
}

// reorderBlockedMatrixByPattern (bda/Reorder.cpp:179-207): row p of the result is old row
// fromOrder[p], columns renamed by toOrder and each row re-sorted by column.
inline Bcrs reorder_matrix(const Bcrs& A, const Reordering& R) {
    Bcrs B;
    B.Nb = A.Nb;
    B.rowptr.assign(A.Nb + 1, 0);
    B.col.resize(A.nnzb());
    B.val.resize((size_t)A.nnzb() * BB);
    std::vector<std::pair<int, int>> tmp;
    for (int p = 0; p < A.Nb; ++p) {
        const int i = R.fromOrder[p];
        tmp.clear();
        for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k) tmp.emplace_back(R.toOrder[A.col[k]], k);
        std::sort(tmp.begin(), tmp.end());
        int o = B.rowptr[p];
        for (auto& t : tmp) {
            B.col[o] = t.first;
            std::memcpy(&B.val[(size_t)o * BB], &A.val[(size_t)t.second * BB], BB * sizeof(double));
            ++o;
        }
        B.rowptr[p + 1] = o;
    }
    return B;
}
inline void reorder_vector(int Nb, const double* v, const std::vector<int>& fromOrder, double* rv) {
    // reorderBlockedVectorByPattern, bda/Reorder.cpp:230-238
    for (int p = 0; p < Nb; ++p)
        for (int r = 0; r < BS; ++r) rv[(size_t)p * BS + r] = v[(size_t)fromOrder[p] * BS + r];
}

// ---- standard-well contributions: y -= C^T (D^-1 (B x)) --------------------------------
// bda/WellContributions.cu:36-126 (dim = 3, dim_wells = 4); host fill order C, D, B per well
// (wells/StandardWellEval.cpp:1206-1250).  Cnnzs/Bnnzs: per perforation a 4x3 block (row-major:
// [r*3+c], r < 4 well equations, c < 3 cell variables); Dnnzs per well a 4x4 row-major block that
// already holds D^-1; val_pointers[numWells+1] = perforation ranges.
struct Wells {
    int numWells = 0;
    std::vector<double> Cnnzs, Dnnzs, Bnnzs;
    std::vector<int> Ccols, Bcols, val_pointers;
};
inline void wells_apply(const Wells& W, const double* x, double* y) {
    for (int w = 0; w < W.numWells; ++w) {
        double z1[4] = {0, 0, 0, 0}, z2[4];
        for (int b = W.val_pointers[w]; b < W.val_pointers[w + 1]; ++b)
            for (int r = 0; r < 4; ++r)
                for (int c = 0; c < 3; ++c)
                    z1[r] += W.Bnnzs[(size_t)b * 12 + r * 3 + c] * x[(size_t)W.Bcols[b] * 3 + c];
        for (int r = 0; r < 4; ++r) {
            double t = 0.0;
            for (int c = 0; c < 4; ++c) t += W.Dnnzs[(size_t)w * 16 + r * 4 + c] * z1[c];
            z2[r] = t;
        }
        for (int b = W.val_pointers[w]; b < W.val_pointers[w + 1]; ++b)
            for (int c = 0; c < 3; ++c) {
                double t = 0.0;
                for (int j = 0; j < 4; ++j) t += W.Cnnzs[(size_t)b * 12 + j * 3 + c] * z2[j];
                y[(size_t)W.Ccols[b] * 3 + c] -= t;
            }
    }
}

// A -= C^T D^-1 B written into the matrix: StandardWell::addWellContributions, wells/StandardWell_impl.hpp:1688-1712,
// with Detail::multMatrix / negativeMultMatrixTransposed of linalg/MatrixBlock.hpp:496-570.  Returns -(w+1) if a block of
// well w is missing from the pattern.
inline int wells_add_to_matrix(const Wells& W, Bcrs& A) {
    for (int w = 0; w < W.numWells; ++w)
        for (int c = W.val_pointers[w]; c < W.val_pointers[w + 1]; ++c)
            for (int b = W.val_pointers[w]; b < W.val_pointers[w + 1]; ++b) {
                double tmp[4][3];
                for (int i = 0; i < 4; ++i)
                    for (int j = 0; j < 3; ++j) {
                        tmp[i][j] = 0.0;
                        for (int k = 0; k < 4; ++k) tmp[i][j] += W.Dnnzs[(size_t)w * 16 + i * 4 + k] * W.Bnnzs[(size_t)b * 12 + k * 3 + j];
                    }
                const int row = W.Ccols[c], col = W.Bcols[b];
                int e = -1;
                for (int k = A.rowptr[row]; k < A.rowptr[row + 1]; ++k)
                    if (A.col[k] == col) { e = k; break; }
                if (e < 0) return -(w + 1);
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j) {
                        double sum = 0.0;
                        for (int k = 0; k < 4; ++k) sum += W.Cnnzs[(size_t)c * 12 + k * 3 + i] * tmp[k][j];
                        A.val[(size_t)e * BB + i * 3 + j] += -sum;
                    }
            }
    return 0;
}
// r -= C^T (D^-1 resWell): StandardWell::apply(BVector& r), wells/StandardWell_impl.hpp:1283-1296
inline void wells_apply_residual(const Wells& W, const double* resWell, double* r) {
    for (int w = 0; w < W.numWells; ++w) {
        double z2[4];
        for (int i = 0; i < 4; ++i) {
            double t = 0.0;
            for (int c = 0; c < 4; ++c) t += W.Dnnzs[(size_t)w * 16 + i * 4 + c] * resWell[(size_t)w * 4 + c];
            z2[i] = t;
        }
        for (int b = W.val_pointers[w]; b < W.val_pointers[w + 1]; ++b)
            for (int c = 0; c < 3; ++c) {
                double t = 0.0;
                for (int j = 0; j < 4; ++j) t += W.Cnnzs[(size_t)b * 12 + j * 3 + c] * z2[j];
                r[(size_t)W.Ccols[b] * 3 + c] -= t;
            }
    }
}
// xw = D^-1 (resWell - B x): StandardWell::recoverSolutionWell, wells/StandardWell_impl.hpp:1298-1311
inline void wells_recover(const Wells& W, const double* resWell, const double* x, double* xw) {
    for (int w = 0; w < W.numWells; ++w) {
        double z1[4];
        for (int i = 0; i < 4; ++i) z1[i] = resWell[(size_t)w * 4 + i];
        for (int i = 0; i < 4; ++i)
            for (int b = W.val_pointers[w]; b < W.val_pointers[w + 1]; ++b)
                for (int c = 0; c < 3; ++c) z1[i] -= W.Bnnzs[(size_t)b * 12 + i * 3 + c] * x[(size_t)W.Bcols[b] * 3 + c];
        for (int i = 0; i < 4; ++i) {
            double t = 0.0;
            for (int c = 0; c < 4; ++c) t += W.Dnnzs[(size_t)w * 16 + i * 4 + c] * z1[c];
            xw[(size_t)w * 4 + i] = t;
        }
    }
}

// ---- right-preconditioned BiCGStab, half-iteration bookkeeping ---------------------------
// Recurrence and stopping rule of bda/cusparseSolverBackend.cu:60-184 (same as
// bda/openclSolverBackend.cpp:317-460; the CPU path's Dune::BiCGSTABSolver, call site
// linalg/FlexibleSolver_impl.hpp:151-157, runs the same recurrence - SURVEY App. B.8).
struct SolveResult {
    int iterations = 0;
    double reduction = 0.0;
    int converged = 0;
    double conv_rate = 0.0;
    float it = 0.f;  // raw half-iteration counter
};

template <class Prec, class Op>
SolveResult bicgstab(size_t n, const double* b, double* x, Prec&& prec, Op&& op, double tol, int maxit) {
    std::vector<double> r(b, b + n), rw(b, b + n), p(b, b + n), v(n, 0.0), s(n), t(n), pw(n);
    std::fill(x, x + n, 0.0);  // x0 = 0 (flow/BlackoilModelEbos.hpp:530)
    double rho = 1.0, rhop, alpha = 1.0, omega = 1.0, beta, tmp1, tmp2;
    double norm = norm2(r.data(), n);
    const double norm_0 = norm;
    float it;
    for (it = 0.5f; it < maxit; it += 0.5f) {
        rhop = rho;
        rho = dot(rw.data(), r.data(), n);
        if (it > 1) {
            beta = (rho / rhop) * (alpha / omega);
            // p = (p - omega v) * beta + r   (openclKernels.cpp:130-153 "custom")
            for (size_t i = 0; i < n; ++i) p[i] = (p[i] - omega * v[i]) * beta + r[i];
        }
        prec(p.data(), pw.data());
        op(pw.data(), v.data());
        tmp1 = dot(rw.data(), v.data(), n);
        alpha = rho / tmp1;
        for (size_t i = 0; i < n; ++i) r[i] -= alpha * v[i];
        for (size_t i = 0; i < n; ++i) x[i] += alpha * pw[i];
        norm = norm2(r.data(), n);
        if (norm < tol * norm_0) break;
        it += 0.5f;
        prec(r.data(), s.data());
        op(s.data(), t.data());
        tmp1 = dot(t.data(), r.data(), n);
        tmp2 = dot(t.data(), t.data(), n);
        omega = tmp1 / tmp2;
        for (size_t i = 0; i < n; ++i) x[i] += omega * s[i];
        for (size_t i = 0; i < n; ++i) r[i] -= omega * t[i];
        norm = norm2(r.data(), n);
        if (norm < tol * norm_0) break;
    }
    SolveResult res;
    res.it = it;
    res.iterations = (int)std::min(it, (float)maxit);
    res.reduction = norm / norm_0;
    res.conv_rate = std::pow(res.reduction, 1.0 / it);
    res.converged = (it != (maxit + 0.5f));
    return res;
}


// ---- the same recurrence with ONE reduction per half iteration (libopmhip's opmhip_config.fused_reductions) ---------------------------
// The reference (above) forms alpha, |r|, omega, |r| and rho from five scalar products in four reductions per iteration.  Here a half
// iteration has three scalar products, all with the product's result, and everything else follows from them:
//   first half,  v = A M^-1 p:  v.rw, v.v, v.r     alpha = rho / v.rw      |r - alpha v|^2 = r.r - 2 alpha v.r + alpha^2 v.v
//   second half, t = A M^-1 r:  t.r, t.t, t.rw     omega = t.r / t.t       |r - omega t|^2 = r.r - 2 omega t.r + omega^2 t.t
//                                                  rho'  = (rho - alpha v.rw) - omega t.rw
// r.r is carried along (a cancellation that comes out negative is clamped to 0), the stopping rule looks at these recurred norms, and the
// vector r is still updated explicitly: its own norm is formed once at the end and reported as the reduction (csrc/solver.hip:
// finalize_scalars3 states the same arithmetic; the scalar products themselves are summed in another order there).
template <class Prec, class Op>
SolveResult bicgstab_fused_reductions(size_t n, const double* b, double* x, Prec&& prec, Op&& op, double tol, int maxit) {
    std::vector<double> r(b, b + n), rw(b, b + n), p(b, b + n), v(n, 0.0), s(n), t(n), pw(n);
    std::fill(x, x + n, 0.0);
    double rr = dot(r.data(), r.data(), n);
    double rho = rr, rhoh = rr, alpha = 1.0, omega = 1.0, beta = 0.0;
    double norm = std::sqrt(rr);
    const double norm_0 = norm;
    auto clamp = [](double q) { return (q > 0.0) ? q : ((q != q) ? q : 0.0); };
    float it;
    for (it = 0.5f; it < maxit; it += 0.5f) {
        if (it > 1)
            for (size_t i = 0; i < n; ++i) p[i] = (p[i] - omega * v[i]) * beta + r[i];
        prec(p.data(), pw.data());
        op(pw.data(), v.data());
        {
            const double s0 = dot(v.data(), rw.data(), n), s1 = dot(v.data(), v.data(), n), s2 = dot(v.data(), r.data(), n);
            alpha = rho / s0;
            rr = clamp((rr - 2.0 * alpha * s2) + alpha * alpha * s1);
            rhoh = rho - alpha * s0;
            norm = std::sqrt(rr);
        }
        for (size_t i = 0; i < n; ++i) r[i] -= alpha * v[i];
        for (size_t i = 0; i < n; ++i) x[i] += alpha * pw[i];
        if (norm < tol * norm_0) break;
        it += 0.5f;
        prec(r.data(), s.data());
        op(s.data(), t.data());
        {
            const double s0 = dot(t.data(), r.data(), n), s1 = dot(t.data(), t.data(), n), s2 = dot(t.data(), rw.data(), n);
            omega = s0 / s1;
            rr = clamp((rr - 2.0 * omega * s0) + omega * omega * s1);
            const double rhop = rho;
            rho = rhoh - omega * s2;
            beta = (rho / rhop) * (alpha / omega);
            norm = std::sqrt(rr);
        }
        for (size_t i = 0; i < n; ++i) x[i] += omega * s[i];
        for (size_t i = 0; i < n; ++i) r[i] -= omega * t[i];
        if (norm < tol * norm_0) break;
    }
    const double truth = norm2(r.data(), n);
    SolveResult res;
    res.it = it;
    res.iterations = (int)std::min(it, (float)maxit);
    res.reduction = truth / norm_0;
    res.conv_rate = std::pow(res.reduction, 1.0 / it);
    res.converged = (it != (maxit + 0.5f)) && (truth < 2.0 * tol * norm_0);
    return res;
}

// =====================================================================================================
// Threaded variants for bench.py's multi-core CPU baseline ("CPU-N": what N MPI ranks of Flow do on one host):
// block-Jacobi ILU0 over nsub contiguous row ranges, one range per thread (ghost_last_bilu0_decomposition per rank,
// linalg/ParallelOverlappingILU0.hpp:439-494: couplings that leave the range are skipped), row-parallel SpMV,
// OpenMP reductions for the scalar products.  Same recurrence as bicgstab() above.  Baseline only: no parity claim
// rests on these (the reduction order depends on the thread count).
// =====================================================================================================
inline int bilu0_decompose_bj(Bcrs& A, const std::vector<int>& sub, const std::vector<int>* diag = nullptr) {
    const std::vector<int> dgOwn = diag ? std::vector<int>() : diag_index(A);
    const std::vector<int>& dg = diag ? *diag : dgOwn;
    const int nsub = (int)sub.size() - 1;
    int err = 0;
#pragma omp parallel for schedule(static, 1)
    for (int sd = 0; sd < nsub; ++sd) {
        const int s0 = sub[sd], s1 = sub[sd + 1];
        for (int i = s0; i < s1; ++i) {
            const int endi = A.rowptr[i + 1];
            int ij = A.rowptr[i];
            for (; ij < endi && A.col[ij] < i; ++ij) {
                const int j = A.col[ij];
                if (j < s0) continue;  // another rank's row: not part of this rank's ILU0
                const int jj = dg[j];
                double* Lij = &A.val[(size_t)ij * BB];
                blk_rightmultiply(Lij, &A.val[(size_t)jj * BB]);
                const int endj = A.rowptr[j + 1];
                int jk = jj + 1, ik = ij + 1;
                while (ik < endi && jk < endj) {
                    if (A.col[ik] == A.col[jk]) {
                        if (A.col[ik] < s1) {
                            double B[BB];
                            std::memcpy(B, &A.val[(size_t)jk * BB], sizeof B);
                            blk_leftmultiply(B, Lij);
                            double* Aik = &A.val[(size_t)ik * BB];
                            for (int e = 0; e < BB; ++e) Aik[e] -= B[e];
                        }
                        ++ik; ++jk;
                    } else if (A.col[ik] < A.col[jk]) ++ik;
                    else ++jk;
                }
            }
            if (ij >= endi || A.col[ij] != i) { err = -(i + 1); continue; }
            double inv[BB];
            blk_invert(&A.val[(size_t)ij * BB], inv);
            std::memcpy(&A.val[(size_t)ij * BB], inv, sizeof inv);
        }
    }
    return err;
}
inline void ilu0_apply_bj(const Bcrs& LU, const std::vector<int>& dg, const std::vector<int>& sub, const double* d,
                          double* v, double w, int mode) {
    const int nsub = (int)sub.size() - 1;
#pragma omp parallel for schedule(static, 1)
    for (int sd = 0; sd < nsub; ++sd) {
        const int s0 = sub[sd], s1 = sub[sd + 1];
        for (int i = s0; i < s1; ++i) {
            double rhs[BS] = {d[(size_t)i * BS], d[(size_t)i * BS + 1], d[(size_t)i * BS + 2]};
            for (int k = LU.rowptr[i]; k < dg[i]; ++k)
                if (LU.col[k] >= s0) blk_mmv(&LU.val[(size_t)k * BB], &v[(size_t)LU.col[k] * BS], rhs);
            for (int r = 0; r < BS; ++r) v[(size_t)i * BS + r] = rhs[r];
        }
        for (int i = s1 - 1; i >= s0; --i) {
            double rhs[BS] = {v[(size_t)i * BS], v[(size_t)i * BS + 1], v[(size_t)i * BS + 2]};
            if (mode == 0) {
                for (int k = LU.rowptr[i + 1] - 1; k > dg[i]; --k)
                    if (LU.col[k] < s1) blk_mmv(&LU.val[(size_t)k * BB], &v[(size_t)LU.col[k] * BS], rhs);
            } else {
                for (int k = dg[i] + 1; k < LU.rowptr[i + 1]; ++k)
                    if (LU.col[k] < s1) blk_mmv(&LU.val[(size_t)k * BB], &v[(size_t)LU.col[k] * BS], rhs);
            }
            double out[BS];
            blk_mv(&LU.val[(size_t)dg[i] * BB], rhs, out);
            for (int r = 0; r < BS; ++r) v[(size_t)i * BS + r] = (mode == 1) ? w * out[r] : out[r];
        }
        if (mode == 0 && w != 1.0)
            for (size_t e = (size_t)s0 * BS; e < (size_t)s1 * BS; ++e) v[e] *= w;
    }
}
inline void spmv_mt(const Bcrs& A, const double* x, double* y) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < A.Nb; ++i) {
        double acc[BS] = {0.0, 0.0, 0.0};
        for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k)
            blk_umv(&A.val[(size_t)k * BB], &x[(size_t)A.col[k] * BS], acc);
        for (int r = 0; r < BS; ++r) y[(size_t)i * BS + r] = acc[r];
    }
}
inline double dot_mt(const double* a, const double* b, size_t n) {
    double s = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : s)
    for (long long i = 0; i < (long long)n; ++i) s += a[i] * b[i];
    return s;
}
template <class Prec, class Op>
SolveResult bicgstab_mt(size_t n, const double* b, double* x, Prec&& prec, Op&& op, double tol, int maxit) {
    std::vector<double> r(b, b + n), rw(b, b + n), p(b, b + n), v(n, 0.0), s(n), t(n), pw(n);
    std::fill(x, x + n, 0.0);
    const long long N = (long long)n;
    double rho = 1.0, rhop, alpha = 1.0, omega = 1.0, beta, tmp1, tmp2;
    double norm = std::sqrt(dot_mt(r.data(), r.data(), n));
    const double norm_0 = norm;
    float it;
    for (it = 0.5f; it < maxit; it += 0.5f) {
        rhop = rho;
        rho = dot_mt(rw.data(), r.data(), n);
        if (it > 1) {
            beta = (rho / rhop) * (alpha / omega);
#pragma omp parallel for schedule(static)
            for (long long i = 0; i < N; ++i) p[i] = (p[i] - omega * v[i]) * beta + r[i];
        }
        prec(p.data(), pw.data());
        op(pw.data(), v.data());
        tmp1 = dot_mt(rw.data(), v.data(), n);
        alpha = rho / tmp1;
#pragma omp parallel for schedule(static)
        for (long long i = 0; i < N; ++i) { r[i] -= alpha * v[i]; x[i] += alpha * pw[i]; }
        norm = std::sqrt(dot_mt(r.data(), r.data(), n));
        if (norm < tol * norm_0) break;
        it += 0.5f;
        prec(r.data(), s.data());
        op(s.data(), t.data());
        tmp1 = dot_mt(t.data(), r.data(), n);
        tmp2 = dot_mt(t.data(), t.data(), n);
        omega = tmp1 / tmp2;
#pragma omp parallel for schedule(static)
        for (long long i = 0; i < N; ++i) { x[i] += omega * s[i]; r[i] -= omega * t[i]; }
        norm = std::sqrt(dot_mt(r.data(), r.data(), n));
        if (norm < tol * norm_0) break;
    }
    SolveResult res;
    res.it = it;
    res.iterations = (int)std::min(it, (float)maxit);
    res.reduction = norm / norm_0;
    res.conv_rate = std::pow(res.reduction, 1.0 / it);
    res.converged = (it != (maxit + 0.5f));
    return res;
}

}  // namespace orc
