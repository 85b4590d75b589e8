// CPR (constrained pressure residual) preconditioner on the device: --linear-solver-configuration=cpr_quasiimpes of the
// reference (opm/simulators/linalg/setupPropertyTree.cpp:94-138) behind the same BiCGStab driver as the ILU0 path.
//
//   M^-1 d:  r_p = sum_k w[k] d[k]  ->  x_p = one AMG V-cycle on A_p  ->  v = (0, x_p, 0)  ->  v += ILU0_{w=1}(d - A v)
//
// Followed line by line: the two-level structure (twolevelmethodcpr.hh:476-498, 0 pre- / 1 post-smoothing step), the
// quasi-IMPES weights (getQuasiImpesWeights.hpp:46-85), the pressure system and the transfers (PressureTransferPolicy.hpp:
// 92-160), the fine smoother (ILU0, relaxation 1), one V-cycle as coarse solve, prolongation damping 1.6.
// NOT the reference's: the AMG itself.  Dune::Amg (not in the reference tree) aggregates with a sequential front algorithm
// and smooths with ILU0, both level-scheduled sequential sweeps; here the hierarchy is one that needs no schedule per level:
// aggregates from two passes of pairwise matching per level (host, once per pattern, from the first pressure matrix; the
// Galerkin VALUES are recomputed on the device for every solve), damped-Jacobi smoothing, dense LU on the coarsest level.
// The CPU restatement of exactly this algorithm is oracle/cpr.hpp; the device reproduces its preconditioner application bit
// for bit in the same ordering (same summation orders, -ffp-contract=off).
#include <climits>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <numeric>
#include <atomic>
#include <map>
#include <memory>
#include <string>
#include <thread>

#include "internal.hpp"

namespace opmhip {

constexpr int CPR_P = 1;             // pressure index inside a block (BlackOilIndices::pressureSwitchIdx)
constexpr int CPR_COARSE_DIRECT = 128;
constexpr int CPR_MAX_LEVELS = 15;

// ---------------------------------------------------------------- host: hierarchy (mirrors oracle/cpr.hpp) -------------
namespace {
struct HCsr {
    int n = 0;
    std::vector<int> rowptr, col;
    std::vector<double> val;
};
// natOf / atNat (level 0 only, else NULL): natural id of every internal index and its inverse - the cells are visited in
// NATURAL order, ties to the lowest natural id: the aggregates of the natural-order matrix whatever the ILU ordering is
// (matching colour by colour pairs cells across the grid and stalls after five levels; oracle/cpr.hpp: same statements)
void pairwise(const HCsr& A, double beta, bool anySign, std::vector<int>& agg, int& na, const int* natOf = nullptr, const int* atNat = nullptr) {
    const int n = A.n;
    agg.assign(n, -1);
    na = 0;
    for (int v = 0; v < n; ++v) {
        const int i = atNat ? atNat[v] : v;
        if (agg[i] >= 0) continue;
        double mx = 0.0;
        for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k)
            if (A.col[k] != i) mx = std::max(mx, anySign ? std::fabs(A.val[k]) : -A.val[k]);
        int best = -1;
        double bv = 0.0;
        for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k) {
            const int j = A.col[k];
            if (j == i || agg[j] >= 0) continue;
            const double s = anySign ? std::fabs(A.val[k]) : -A.val[k];
            if (s >= beta * mx && (s > bv || (natOf && best >= 0 && s == bv && natOf[j] < natOf[best]))) { best = j; bv = s; }
        }
        agg[i] = na;
        if (best >= 0) agg[best] = na;
        ++na;
    }
}
// Galerkin product for piecewise-constant prolongation: coarse pattern (columns ascending), gather lists (fine entries of a
// coarse entry in ascending order), values.  Row by row over the members of each aggregate: the few (coarse column, fine
// entry) pairs of a coarse row are sorted on the spot - O(nnz log(row)) instead of a stable sort of all nnz keys (round 2: 1.5 s
// of host time for the hierarchy of a 10^6-cell grid, most of it here); same lists, same sums, same order.
void galerkin(const HCsr& A, const std::vector<int>& agg, int nc, HCsr& C, std::vector<int>& gptr, std::vector<int>& gidx) {
    const int nnz = (int)A.col.size();
    std::vector<int> mp(nc + 1, 0), mi(A.n);
    for (int i = 0; i < A.n; ++i) mp[agg[i] + 1]++;
    for (int I = 0; I < nc; ++I) mp[I + 1] += mp[I];
    {
        std::vector<int> w(mp.begin(), mp.end() - 1);
        for (int i = 0; i < A.n; ++i) mi[w[agg[i]]++] = i;   // members ascending
    }
    // the coarse rows are independent: slices of them are built by a few host threads, each into vectors of its own, and
    // joined in order - the same lists and sums whatever the number of threads
    const int T = std::max(1, std::min({(int)std::thread::hardware_concurrency(), 8, nc / 4096 + 1}));
    struct Part { std::vector<int> rowlen, col, glen, gidx; std::vector<double> val; };
    std::vector<Part> parts(T);
    auto work = [&](int t) {
        Part& Q = parts[t];
        const int I0 = (int)((long long)nc * t / T), I1 = (int)((long long)nc * (t + 1) / T);
        std::vector<std::pair<int, int>> pairs;   // (coarse column, fine entry) of the coarse row in hand
        Q.rowlen.reserve(I1 - I0);
        for (int I = I0; I < I1; ++I) {
            pairs.clear();
            for (int q = mp[I]; q < mp[I + 1]; ++q) {
                const int i = mi[q];
                for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k) pairs.emplace_back(agg[A.col[k]], k);
            }
            std::sort(pairs.begin(), pairs.end());
            int len = 0;
            for (size_t q = 0; q < pairs.size();) {
                const int cc = pairs[q].first;
                double sum = 0.0;
                size_t e = q;
                while (e < pairs.size() && pairs[e].first == cc) { sum += A.val[pairs[e].second]; Q.gidx.push_back(pairs[e].second); ++e; }
                Q.col.push_back(cc);
                Q.val.push_back(sum);
                Q.glen.push_back((int)(e - q));
                ++len;
                q = e;
            }
            Q.rowlen.push_back(len);
        }
    };
    if (T == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back(work, t);
        for (auto& x : th) x.join();
    }
    C.n = nc;
    C.rowptr.assign(nc + 1, 0);
    C.col.clear();
    C.val.clear();
    gptr.assign(1, 0);
    gidx.clear();
    gidx.reserve(nnz);
    int I = 0;
    for (const Part& Q : parts) {
        for (int len : Q.rowlen) { C.rowptr[I + 1] = C.rowptr[I] + len; ++I; }
        C.col.insert(C.col.end(), Q.col.begin(), Q.col.end());
        C.val.insert(C.val.end(), Q.val.begin(), Q.val.end());
        for (int gl : Q.glen) gptr.push_back(gptr.back() + gl);
        gidx.insert(gidx.end(), Q.gidx.begin(), Q.gidx.end());
    }
}
}  // namespace

// ---------------------------------------------------------------- kernels -----------------------------------------------
#define CPR_DONE_CHECK if (*done != 0.0) return;
constexpr int CPR_MAX_W = 96;   // longest row an ELL level may have
// closed-form 3x3 inverse, expression tree of Opm::Detail::Inverter<3> (linalg/MatrixBlock.hpp:722-747); same as solver.hip
__device__ __forceinline__ void inv3(const double* m, double* inv) {
    const double t4 = m[0] * m[4], t6 = m[0] * m[5], t8 = m[1] * m[3];
    const double t10 = m[2] * m[3], t12 = m[1] * m[6], t14 = m[2] * m[6];
    const double det = (t4 * m[8] - t6 * m[7] - t8 * m[8] + t10 * m[7] + t12 * m[5] - t14 * m[4]);
    const double t17 = 1.0 / det;
    inv[0] = (m[4] * m[8] - m[5] * m[7]) * t17;
    inv[1] = -(m[1] * m[8] - m[2] * m[7]) * t17;
    inv[2] = (m[1] * m[5] - m[2] * m[4]) * t17;
    inv[3] = -(m[3] * m[8] - m[5] * m[6]) * t17;
    inv[4] = (m[0] * m[8] - t14) * t17;
    inv[5] = -(t6 - t10) * t17;
    inv[6] = (m[3] * m[7] - m[4] * m[6]) * t17;
    inv[7] = -(m[0] * m[7] - t12) * t17;
    inv[8] = (t4 - t8) * t17;
}
// quasi-IMPES weights: w_i = D_ii^-T e_p / max|.| (getQuasiImpesWeights.hpp:46-85)
__global__ __launch_bounds__(256) void k_cpr_weights(int Nb, const int* __restrict__ diag, const double* __restrict__ A, double* __restrict__ w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Nb) return;
    const double* D = &A[(size_t)diag[i] * BB];
    double Dt[BB], inv[BB];
    for (int r = 0; r < BS; ++r)
        for (int c = 0; c < BS; ++c) Dt[r * BS + c] = D[c * BS + r];
    inv3(Dt, inv);
    double bw[BS], mx = 0.0;
    for (int r = 0; r < BS; ++r) { bw[r] = inv[r * BS + CPR_P]; mx = fmax(mx, fabs(bw[r])); }
    for (int r = 0; r < BS; ++r) w[(size_t)i * BS + r] = bw[r] / mx;
}
// Level matrices live in ELL form: entry j of row i at [j * n + i] (j < W = longest row of the level; padding: column i, value 0),
// so that the one-thread-per-row kernels read coalesced and with a uniform trip count.  Row sums run over j ascending = the
// CSR order of the oracle; a padding term subtracts 0 * x_i and leaves the sum's bits alone.
// pressure matrix: a_p[k] = sum_r A_k[r][p] w_row[r] (PressureTransferPolicy::calculateCoarseEntries, :116-139)
// pcol != NULL: the pressure COLUMN of every block (3 doubles) goes into an ELL image of its own, component-major
// [c][j * Nb + i]: what the post-smoothing residual d - A (0, x_p, 0) needs of the matrix (k_cpr_presid) - a third of its bytes
// col != NULL (decomposed runs): a coupling to a ghost cell (column >= Nb) is left out of the subdomain's pressure system
// - its slot holds 0, which the row sums treat like padding - as the block ILU0 leaves it out of its factors
__global__ __launch_bounds__(256) void k_cpr_pvals(int Nb, int W, const int* __restrict__ rowptr, const int* __restrict__ col, const double* __restrict__ A,
                                                   const double* __restrict__ w, double* __restrict__ ap, double* __restrict__ pcol) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Nb) return;
    const double w0 = w[(size_t)i * BS], w1 = w[(size_t)i * BS + 1], w2 = w[(size_t)i * BS + 2];
    const int kb = rowptr[i];
    for (int k = kb; k < rowptr[i + 1]; ++k) {
        const double* B = &A[(size_t)k * BB];
        const bool ghost = col && col[k] >= Nb;
        const double b0 = ghost ? 0.0 : B[0 * BS + CPR_P], b1 = ghost ? 0.0 : B[1 * BS + CPR_P], b2 = ghost ? 0.0 : B[2 * BS + CPR_P];
        double s = 0.0;
        s += b0 * w0; s += b1 * w1; s += b2 * w2;
        const size_t e = (size_t)(k - kb) * Nb + i;
        ap[e] = s;
        if (pcol) {
            const size_t plane = (size_t)W * Nb;
            pcol[e] = b0; pcol[plane + e] = b1; pcol[2 * plane + e] = b2;
        }
    }
}
// The ELL columns of level 0 in stencil form (the SpMV's encoding, k_spmv_pipe_st): in the ILU ordering the rows of an aligned group of 32
// share their column offsets, so a row needs one word of 4-bit indices (15 = padding: the row itself) into the group's table instead of W
// column indices - a quarter of what the one-thread-per-row kernels of level 0 read.  The table entries travel lane to lane: every lane of
// a wavefront must get here (rows past the end are clamped, their results not stored).
struct EllStencil {
    const unsigned* __restrict__ word;   // NULL: explicit columns
    const int* __restrict__ table;
};
struct EllCols {
    unsigned w;
    int tab, i, half;
    __device__ __forceinline__ EllCols(const EllStencil S, int i_) : i(i_), half((int)(threadIdx.x & 32u)) {
        w = S.word[i];
        tab = S.table[(size_t)(i >> 5) * 16 + (threadIdx.x & 15u)];
    }
    __device__ __forceinline__ int col(int j) const {
        const int nib = (int)((w >> (4 * j)) & 15u);
        const int off = __shfl(tab, half + nib, 64);
        return nib != 15 ? i + off : i;
    }
};
// r = d - A v for v = (0, x_p, 0): of every block only its pressure column meets a non-zero, so the row sums of
// BCRSMatrix::mv (y_i = 0, then block by block in ascending column order, each component adding its three products) reduce
// to the products with x_p - adding the +-0 products of the two other columns changes no bit of a sum that started at +0.
// One thread per row on the ELL image of the pressure columns (padding entries are 0 * x_i).  Replaces a full SpMV (579 MB)
// and the subtraction kernel behind it by one pass over 196 MB of matrix data.
template <bool ST>
__global__ __launch_bounds__(256) void k_cpr_presid(int n, int W, const int* __restrict__ ecol, const double* __restrict__ pcol, const double* __restrict__ d,
                                                    const double* __restrict__ xp, double* __restrict__ r, const double* __restrict__ done, const EllStencil S) {
    CPR_DONE_CHECK
    const int i0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (!ST && i0 >= n) return;
    const int i = i0 < n ? i0 : n - 1;
    const size_t plane = (size_t)W * n;
    double y0 = 0.0, y1 = 0.0, y2 = 0.0;
    if (ST) {
        const EllCols C(S, i);
#pragma unroll
        for (int j = 0; j < 8; ++j) {   // W <= 8 in this form (the same for every lane)
            if (j < W) {
                const size_t e = (size_t)j * n + i;
                const double x = xp[C.col(j)];
                y0 += pcol[e] * x; y1 += pcol[plane + e] * x; y2 += pcol[2 * plane + e] * x;
            }
        }
        if (i0 >= n) return;
    } else {
#pragma unroll 4
        for (int j = 0; j < W; ++j) {
            const size_t e = (size_t)j * n + i;
            const double x = xp[ecol[e]];
            y0 += pcol[e] * x; y1 += pcol[plane + e] * x; y2 += pcol[2 * plane + e] * x;
        }
    }
    r[(size_t)i * BS] = d[(size_t)i * BS] - y0;
    r[(size_t)i * BS + 1] = d[(size_t)i * BS + 1] - y1;
    r[(size_t)i * BS + 2] = d[(size_t)i * BS + 2] - y2;
}
// Galerkin values: coarse entry e (at ELL position cpos[e]) = sum of its fine entries (ELL positions gidx) in ascending order
__global__ __launch_bounds__(256) void k_cpr_galerkin(int nce, const int* __restrict__ gptr, const int* __restrict__ gidx, const int* __restrict__ cpos,
                                                      const double* __restrict__ fine, double* __restrict__ coarse) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nce) return;
    double s = 0.0;
    for (int q = gptr[e]; q < gptr[e + 1]; ++q) s += fine[gidx[q]];
    coarse[cpos[e]] = s;
}
__global__ __launch_bounds__(256) void k_cpr_dinv(int n, const int* __restrict__ diag, const double* __restrict__ val, double* __restrict__ dinv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dinv[i] = 1.0 / val[diag[i]];
}
// dense LU without pivoting of the coarsest level, in place in global memory, one workgroup (n <= CPR_COARSE_DIRECT)
__global__ __launch_bounds__(256) void k_cpr_dense_lu(int n, int W, int rm, const int* __restrict__ ecol, const int* __restrict__ rlen, const double* __restrict__ val,
                                                      double* lu) {
    for (int e = threadIdx.x; e < n * n; e += 256) lu[e] = 0.0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256)
        for (int j = 0; j < rlen[i]; ++j) {
            const size_t e = rm ? (size_t)i * W + j : (size_t)j * n + i;
            lu[(size_t)i * n + ecol[e]] = val[e];
        }
    __syncthreads();
    if (threadIdx.x == 0) lu[(size_t)n * n] = 0.0;   // flag behind the factors: 1 = a pivot vanished or is not finite (no pivoting here)
    for (int k = 0; k < n; ++k) {
        const double pv = lu[(size_t)k * n + k];
        if (threadIdx.x == 0 && (pv == 0.0 || !isfinite(pv))) lu[(size_t)n * n] = 1.0;
        const double piv = 1.0 / pv;
        for (int i = k + 1 + threadIdx.x; i < n; i += 256) lu[(size_t)i * n + k] = lu[(size_t)i * n + k] * piv;
        __syncthreads();
        const int m = n - k - 1;
        for (int e = threadIdx.x; e < m * m; e += 256) {
            const int i = k + 1 + e / m, j = k + 1 + e % m;
            lu[(size_t)i * n + j] -= lu[(size_t)i * n + k] * lu[(size_t)k * n + j];
        }
        if (threadIdx.x == 0) lu[(size_t)k * n + k] = piv;   // the diagonal keeps 1 / u_kk: the substitution multiplies (oracle/cpr.hpp: update_values)
        __syncthreads();
    }
}
// the same factorisation with the matrix in LDS (n x n doubles of dynamic shared memory: 128 KB at n = 128, which gfx950's 160 KB hold) and
// 1 024 threads on a 32 x 32 tiling of the trailing block: every entry's updates still come in the order k ascending, each one the same
// rounded product and subtraction - the same factors - without 2 n barriers over global memory (0.56 ms per solve on a 100-row level).
constexpr int CPR_LU_THREADS = 1024, CPR_LU_TX = 32;   // 0.12 ms at n = 123; 256 threads on 16 x 16 tiles: 0.21 ms (the work between two barriers, not the barriers, is what takes the time)
__global__ __launch_bounds__(CPR_LU_THREADS) void k_cpr_dense_lu_lds(int n, int W, int rm, const int* __restrict__ ecol, const int* __restrict__ rlen, const double* __restrict__ val,
                                                           double* __restrict__ lu) {
    extern __shared__ double slu[];
    const int tid = threadIdx.x, tx = tid & (CPR_LU_TX - 1), ty = tid / CPR_LU_TX;
    for (int e = tid; e < n * n; e += CPR_LU_THREADS) slu[e] = 0.0;
    __syncthreads();
    for (int i = tid; i < n; i += CPR_LU_THREADS)
        for (int j = 0; j < rlen[i]; ++j) {
            const size_t e = rm ? (size_t)i * W + j : (size_t)j * n + i;
            slu[i * n + ecol[e]] = val[e];
        }
    __syncthreads();
    bool bad = false;   // (thread 0's copy is the one that counts)
    for (int k = 0; k < n; ++k) {
        const double pv = slu[k * n + k];
        if (pv == 0.0 || !isfinite(pv)) bad = true;
        const double piv = 1.0 / pv;
        __syncthreads();   // every thread has read the pivot before column k is scaled (the pivot itself is not touched)
        for (int i = k + 1 + tid; i < n; i += CPR_LU_THREADS) slu[i * n + k] = slu[i * n + k] * piv;
        __syncthreads();
        for (int i = k + 1 + ty; i < n; i += CPR_LU_THREADS / CPR_LU_TX) {
            const double f = slu[i * n + k];
            for (int j = k + 1 + tx; j < n; j += CPR_LU_TX) slu[i * n + j] -= f * slu[k * n + j];
        }
        if (tid == 0) slu[k * n + k] = piv;   // the diagonal keeps 1 / u_kk (nobody reads it again in the factorisation)
        __syncthreads();
    }
    for (int e = tid; e < n * n; e += CPR_LU_THREADS) lu[e] = slu[e];
    if (tid == 0) lu[(size_t)n * n] = bad ? 1.0 : 0.0;   // flag behind the factors: 1 = a pivot vanished or is not finite (no pivoting here)
}
// x = U^-1 L^-1 b with the dense factors of the coarsest level (n <= CPR_COARSE_DIRECT).  Both substitutions go COLUMN by column, as the
// oracle's do (oracle/cpr.hpp: vcycle; the same terms in the same order, the same bits): forward (unit lower factor) x_j is final once the
// columns before it are applied, and the rows below take l_ij x_j off at once; backward x_j = s_j * (1 / u_jj) - the factorisation leaves the reciprocal on the diagonal - is final once the columns behind
// it are applied, and the rows above take u_ij x_j off at once - a row's terms in the order j = n - 1 ... i + 1.  ONE wavefront, two rows per
// lane IN REGISTERS: the 2 n steps are chained only through "x_j of its owner lane -> everybody" (v_readlane with a uniform lane: a few
// cycles, where a round trip through LDS is a hundred) and one multiply-subtract; the factors are staged into LDS once (rows padded to an
// odd length: the lanes' column reads then spread over the banks) and a step's two entries are read four steps ahead.
// History: until round 5 the iterate lived in global memory and one thread walked both triangles row by row through dependent loads of
// what it had just stored - 0.7 ms per application on a 100-row level; row-oriented in LDS 0.13 ms (a chain of n^2 / 2 subtractions:
// why the backward ORDER changed, on both sides); column-oriented through LDS 0.06 ms (2 n LDS round trips).
// (profiles/r05_config_rates.txt, r05_cpr_tail_ab.txt).
__device__ __forceinline__ double cpr_bcast_lane(double v, int src) {   // src: the same for all lanes
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
constexpr int CPR_DENSE_AHEAD = 4;
// mem4f != NULL: b[i] = sum of the finer level's residual over aggregate i (k_cpr_restrict's statement), formed here
constexpr int CPR_DENSE_THREADS = 512;   // all of them stage the factors; the first wavefront substitutes
__global__ __launch_bounds__(CPR_DENSE_THREADS) void k_cpr_dense_solve(int n, const double* __restrict__ lu, const double* __restrict__ b, double* __restrict__ x,
                                                        const int* __restrict__ mem4f, const double* __restrict__ rf, const double* __restrict__ done) {
    CPR_DONE_CHECK
    extern __shared__ double slu[];   // n rows of np doubles
    const int lane = threadIdx.x, np = n | 1;
    {   // thread t: column t mod 128 of the rows t / 128, t / 128 + 4, ... - eight loads in flight per thread, no division
        const int j = lane & 127;
#pragma unroll 8
        for (int i = lane >> 7; i < n; i += CPR_DENSE_THREADS / 128)
            if (j < n) slu[i * np + j] = lu[(size_t)i * n + j];
    }
    __syncthreads();
    if (lane >= 64) return;
    const int i0 = lane, i1 = lane + 64;
    auto rhs = [&](int i) {
        if (i >= n) return 0.0;
        if (!mem4f) return b[i];
        const int* m = &mem4f[4 * i];
        double s = 0.0;
        s += rf[m[0]];
        if (m[1] >= 0) s += rf[m[1]];
        if (m[2] >= 0) s += rf[m[2]];
        if (m[3] >= 0) s += rf[m[3]];
        return s;
    };
    double s0 = rhs(i0), s1 = rhs(i1);
    auto at = [&](int i, int j) { return (i < n && j >= 0 && j < n) ? slu[i * np + j] : 0.0; };
    {   // forward
        double a0[CPR_DENSE_AHEAD], a1[CPR_DENSE_AHEAD];
#pragma unroll
        for (int u = 0; u < CPR_DENSE_AHEAD; ++u) { a0[u] = at(i0, u); a1[u] = at(i1, u); }
        for (int jb = 0; jb + 1 < n; jb += CPR_DENSE_AHEAD) {
            double c0[CPR_DENSE_AHEAD], c1[CPR_DENSE_AHEAD];
#pragma unroll
            for (int u = 0; u < CPR_DENSE_AHEAD; ++u) { c0[u] = a0[u]; c1[u] = a1[u]; a0[u] = at(i0, jb + CPR_DENSE_AHEAD + u); a1[u] = at(i1, jb + CPR_DENSE_AHEAD + u); }
#pragma unroll
            for (int u = 0; u < CPR_DENSE_AHEAD; ++u) {
                const int j = jb + u;
                if (j + 1 >= n) break;
                const double xj = cpr_bcast_lane(j < 64 ? s0 : s1, j & 63);
                s0 -= (i0 > j ? c0[u] : 0.0) * xj;     // rows at or above the column take 0 * x_j = 0 off: their bits stay (at() gave 0 for rows >= n)
                s1 -= (i1 > j ? c1[u] : 0.0) * xj;
            }
        }
    }
    {   // backward
        double a0[CPR_DENSE_AHEAD], a1[CPR_DENSE_AHEAD];
#pragma unroll
        for (int u = 0; u < CPR_DENSE_AHEAD; ++u) { a0[u] = at(i0, n - 1 - u); a1[u] = at(i1, n - 1 - u); }
        for (int jb = n - 1; jb >= 0; jb -= CPR_DENSE_AHEAD) {
            double c0[CPR_DENSE_AHEAD], c1[CPR_DENSE_AHEAD];
#pragma unroll
            for (int u = 0; u < CPR_DENSE_AHEAD; ++u) { c0[u] = a0[u]; c1[u] = a1[u]; a0[u] = at(i0, jb - CPR_DENSE_AHEAD - u); a1[u] = at(i1, jb - CPR_DENSE_AHEAD - u); }
#pragma unroll
            for (int u = 0; u < CPR_DENSE_AHEAD; ++u) {
                const int j = jb - u;
                if (j < 0) break;
                s0 = s0 * (i0 == j ? c0[u] : 1.0);     // the lane that holds row j holds the diagonal entry 1 / u_jj; s * 1 = s for the others
                s1 = s1 * (i1 == j ? c1[u] : 1.0);
                const double xj = cpr_bcast_lane(j < 64 ? s0 : s1, j & 63);
                s0 -= (i0 < j ? c0[u] : 0.0) * xj;
                s1 -= (i1 < j ? c1[u] : 0.0) * xj;
            }
        }
    }
    if (i0 < n) x[i0] = s0;
    if (i1 < n) x[i1] = s1;
}
// r_p[i] = sum_k d_i[k] w_i[k]  (moveToCoarseLevel, :141-160); x0 != NULL: the level's pre-smoothing from x = 0 rides along,
// x0[i] = omega D^-1 r_p[i] - the statement of k_cpr_presmooth, one launch less
__global__ __launch_bounds__(256) void k_cpr_restrict_fine(int Nb, const double* __restrict__ d, const double* __restrict__ w, double* __restrict__ rc,
                                                           double omega, const double* __restrict__ dinv, double* __restrict__ x0,
                                                           const double* __restrict__ done) {
    CPR_DONE_CHECK
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Nb) return;
    double s = 0.0;
    for (int k = 0; k < BS; ++k) s += d[(size_t)i * BS + k] * w[(size_t)i * BS + k];
    rc[i] = s;
    if (x0) x0[i] = omega * dinv[i] * s;
}
// coarsest level without a direct solve: x = omega D^-1 b, then Jacobi sweeps x_out = x_in + omega D^-1 (b - A x_in)
__global__ __launch_bounds__(256) void k_cpr_presmooth(int n, double omega, const double* __restrict__ dinv, const double* __restrict__ b, double* __restrict__ x,
                                                       const double* __restrict__ done) {
    CPR_DONE_CHECK
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = omega * dinv[i] * b[i];
}
__global__ __launch_bounds__(256) void k_cpr_jacobi(int n, int W, double omega, const int* __restrict__ ecol, const double* __restrict__ val,
                                                    const double* __restrict__ dinv, const double* __restrict__ b, const double* __restrict__ xin,
                                                    double* __restrict__ xout, const double* __restrict__ done) {
    CPR_DONE_CHECK
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = b[i];
#pragma unroll 4
    for (int j = 0; j < W; ++j) s -= val[(size_t)j * n + i] * xin[ecol[(size_t)j * n + i]];
    xout[i] = xin[i] + omega * dinv[i] * s;
}
// r = b - A x, one thread per row (large levels; x = omega D^-1 b was stored by k_cpr_presmooth: the same bits the
// lane-group kernel below forms on the fly)
template <bool ST>
__global__ __launch_bounds__(256) void k_cpr_resid(int n, int W, const int* __restrict__ ecol, const double* __restrict__ val,
                                                   const double* __restrict__ b, const double* __restrict__ x, double* __restrict__ r, const double* __restrict__ done,
                                                   const EllStencil S) {
    CPR_DONE_CHECK
    const int i0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (!ST && i0 >= n) return;
    const int i = i0 < n ? i0 : n - 1;
    double s = b[i];
    if (ST) {
        const EllCols C(S, i);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < W) s -= val[(size_t)j * n + i] * x[C.col(j)];
        if (i0 >= n) return;
    } else {
#pragma unroll 4
        for (int j = 0; j < W; ++j) s -= val[(size_t)j * n + i] * x[ecol[(size_t)j * n + i]];
    }
    r[i] = s;
}
// Coarse levels (<= CPR_LPR_ROWS rows, rows of up to CPR_MAX_W entries): one thread per row walks W dependent
// (column -> vector entry) round trips - 12 us for 14 000 rows.  Here CPR_LPR lanes share a row (row-major storage): every
// lane fetches its entries and forms its products at once, then the group subtracts the products one after the other in
// the row's order - the same roundings as the one-thread loop (padding entries are 0 * x there too).
constexpr int CPR_LPR = 16, CPR_LPR_SLOTS = CPR_MAX_W / CPR_LPR, CPR_LPR_ROWS = 32768;
static_assert(CPR_COARSE_DIRECT <= 128, "k_cpr_dense_solve: two rows per lane of one wavefront, 128 columns per staging pass");
static_assert(CPR_MAX_W % CPR_LPR == 0 && 256 % CPR_LPR == 0, "lane groups tile a row and a workgroup");
__device__ __forceinline__ double cpr_group_subtract(double s, const double (&p)[CPR_LPR_SLOTS], int W) {
#pragma unroll
    for (int slot = 0; slot < CPR_LPR_SLOTS; ++slot) {
        if (slot * CPR_LPR >= W) break;    // W is the same for all lanes
#pragma unroll
        for (int l = 0; l < CPR_LPR; ++l) {
            const double q = __shfl(p[slot], l, CPR_LPR);
            if (slot * CPR_LPR + l < W) s -= q;
        }
    }
    return s;
}
// Right-hand side of a coarse row formed where it is needed: b[c] = sum of the finer level's residual over the aggregate's
// members - k_cpr_restrict's statement (s = 0, then the members in ascending order), so the same bits - from a four-int member
// record.  The lane-group kernels of a coarse level call it for the row and for every neighbour: the restriction launch
// between two levels (5 us for a few thousand rows) is gone.
__device__ __forceinline__ double cpr_restricted(const int4* __restrict__ mem4, const double* __restrict__ rf, int c) {
    const int4 m = mem4[c];
    double s = 0.0;
    s += rf[m.x];
    if (m.y >= 0) s += rf[m.y];
    if (m.z >= 0) s += rf[m.z];
    if (m.w >= 0) s += rf[m.w];
    return s;
}
// FUSED (coarsest level, first sweep): b and the iterate x = omega D^-1 b it starts from are formed on the fly (b is stored for the sweeps that follow)
template <bool FUSED>
__global__ __launch_bounds__(256) void k_cpr_jacobi_lpr(int n, int W, double omega, const int* __restrict__ ecol, const double* __restrict__ val,
                                                        const double* __restrict__ dinv, double* __restrict__ b, const double* __restrict__ xin,
                                                        double* __restrict__ xout, const int4* __restrict__ mem4f, const double* __restrict__ rf,
                                                        const double* __restrict__ done) {
    CPR_DONE_CHECK
    const int g = (blockIdx.x * blockDim.x + threadIdx.x) / CPR_LPR, l = threadIdx.x % CPR_LPR;
    const int i = g < n ? g : n - 1;
    double p[CPR_LPR_SLOTS];
#pragma unroll
    for (int slot = 0; slot < CPR_LPR_SLOTS; ++slot) {
        const int j = slot * CPR_LPR + l;
        p[slot] = 0.0;
        if (j < W) {
            const int c = ecol[(size_t)i * W + j];
            p[slot] = val[(size_t)i * W + j] * (FUSED ? omega * dinv[c] * cpr_restricted(mem4f, rf, c) : xin[c]);
        }
    }
    const double bi = FUSED ? cpr_restricted(mem4f, rf, i) : b[i];
    const double s = cpr_group_subtract(bi, p, W);
    if (l == 0 && g < n) {
        if (FUSED) b[i] = bi;
        xout[i] = (FUSED ? omega * dinv[i] * bi : xin[i]) + omega * dinv[i] * s;
    }
}
// FUSED: b comes from the finer level's residual (cpr_restricted) and is stored for the way up
template <bool FUSED>
__global__ __launch_bounds__(256) void k_cpr_down_lpr(int n, int W, double omega, const int* __restrict__ ecol, const double* __restrict__ val,
                                                      const double* __restrict__ dinv, double* __restrict__ b, double* __restrict__ x,
                                                      double* __restrict__ r, const int4* __restrict__ mem4f, const double* __restrict__ rf,
                                                      const double* __restrict__ done) {
    CPR_DONE_CHECK
    const int g = (blockIdx.x * blockDim.x + threadIdx.x) / CPR_LPR, l = threadIdx.x % CPR_LPR;
    const int i = g < n ? g : n - 1;
    double p[CPR_LPR_SLOTS];
#pragma unroll
    for (int slot = 0; slot < CPR_LPR_SLOTS; ++slot) {
        const int j = slot * CPR_LPR + l;
        p[slot] = 0.0;
        if (j < W) {
            const int c = ecol[(size_t)i * W + j];
            p[slot] = val[(size_t)i * W + j] * (omega * dinv[c] * (FUSED ? cpr_restricted(mem4f, rf, c) : b[c]));
        }
    }
    const double bi = FUSED ? cpr_restricted(mem4f, rf, i) : b[i];
    const double s = cpr_group_subtract(bi, p, W);
    if (l == 0 && g < n) {
        if (FUSED) b[i] = bi;
        x[i] = omega * dinv[i] * bi;
        r[i] = s;
    }
}
__global__ __launch_bounds__(256) void k_cpr_up_lpr(int n, int W, double omega, double damp, const int* __restrict__ ecol, const double* __restrict__ val,
                                                    const double* __restrict__ dinv, const int* __restrict__ agg, const double* __restrict__ xc,
                                                    const double* __restrict__ b, const double* __restrict__ x, double* __restrict__ xout,
                                                    const double* __restrict__ done) {
    CPR_DONE_CHECK
    const int g = (blockIdx.x * blockDim.x + threadIdx.x) / CPR_LPR, l = threadIdx.x % CPR_LPR;
    const int i = g < n ? g : n - 1;
    double p[CPR_LPR_SLOTS];
#pragma unroll
    for (int slot = 0; slot < CPR_LPR_SLOTS; ++slot) {
        const int j = slot * CPR_LPR + l;
        p[slot] = 0.0;
        if (j < W) {
            const int c = ecol[(size_t)i * W + j];
            const double xp = x[c] + damp * xc[agg[c]];
            p[slot] = val[(size_t)i * W + j] * xp;
        }
    }
    const double s = cpr_group_subtract(b[i], p, W);
    if (l == 0 && g < n) {
        const double xi = x[i] + damp * xc[agg[i]];
        xout[i] = xi + omega * dinv[i] * s;
    }
}
// xc != NULL: the coarse level's pre-smoothing from x = 0 rides along (xc[I] = omega D_c^-1 rc[I], k_cpr_presmooth's statement)
__global__ __launch_bounds__(256) void k_cpr_restrict(int nc, const int* __restrict__ mptr, const int* __restrict__ midx, const double* __restrict__ r,
                                                      double* __restrict__ rc, double omega, const double* __restrict__ dinvc, double* __restrict__ xc,
                                                      const double* __restrict__ done) {
    CPR_DONE_CHECK
    const int I = blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= nc) return;
    double s = 0.0;
    for (int q = mptr[I]; q < mptr[I + 1]; ++q) s += r[midx[q]];
    rc[I] = s;
    if (xc) xc[I] = omega * dinvc[I] * s;
}
// going up: damped piecewise-constant prolongation x' = x + damp xc[agg], then the residual of x' and the post-smoothing
// xout = x' + omega D^-1 (b - A x') (large levels: two passes, one gather per entry; small levels: k_cpr_up_lpr forms the
// x'_j of the neighbours on the fly - the same expression, hence the same bits)
__global__ __launch_bounds__(256) void k_cpr_prolong(int n, double damp, const int* __restrict__ agg, const double* __restrict__ xc,
                                                     const double* __restrict__ x, double* __restrict__ xp, const double* __restrict__ done) {
    CPR_DONE_CHECK
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) xp[i] = x[i] + damp * xc[agg[i]];
}
// vfine != NULL (level 0): the result goes straight into the block vector v = (0, x_p, 0) (moveToFineLevel: the pressure
// component only) - k_cpr_prolong_fine's statement, one launch and one pass over x_p less
template <bool ST>
__global__ __launch_bounds__(256) void k_cpr_post(int n, int W, double omega, const int* __restrict__ ecol, const double* __restrict__ val,
                                                  const double* __restrict__ dinv, const double* __restrict__ b, const double* __restrict__ xp,
                                                  double* __restrict__ xout, double* __restrict__ vfine, const double* __restrict__ done, const EllStencil S) {
    CPR_DONE_CHECK
    const int i0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (!ST && i0 >= n) return;
    const int i = i0 < n ? i0 : n - 1;
    double s = b[i];
    if (ST) {
        const EllCols C(S, i);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < W) s -= val[(size_t)j * n + i] * xp[C.col(j)];
        if (i0 >= n) return;
    } else {
#pragma unroll 4
        for (int j = 0; j < W; ++j) s -= val[(size_t)j * n + i] * xp[ecol[(size_t)j * n + i]];
    }
    const double xo = xp[i] + omega * dinv[i] * s;
    xout[i] = xo;
    if (vfine) {
        double* v = &vfine[(size_t)i * BS];
#pragma unroll
        for (int k = 0; k < BS; ++k) v[k] = (k == CPR_P) ? xo : 0.0;
    }
}
// v = (0, x_p, 0)  (moveToFineLevel: the pressure component only)
__global__ __launch_bounds__(256) void k_cpr_prolong_fine(int Nb, const double* __restrict__ xc, double* __restrict__ v, const double* __restrict__ done) {
    CPR_DONE_CHECK
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= Nb * BS) return;
    v[e] = (e % BS == CPR_P) ? xc[e / BS] : 0.0;
}
__global__ __launch_bounds__(256) void k_cpr_sub(int n, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ r,
                                                 const double* __restrict__ done) {
    CPR_DONE_CHECK
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) r[e] = a[e] - b[e];
}
__global__ __launch_bounds__(256) void k_cpr_add(int n, double* __restrict__ v, const double* __restrict__ z, const double* __restrict__ done) {
    CPR_DONE_CHECK
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) v[e] += z[e];
}


// ---------------------------------------------------------------- ILU0 smoothing of the finest AMG levels -----------------
// The reference's pressure AMG smooths with ILU0, relaxation 1 (PreconditionerFactory.hpp:126-151, setupPropertyTree.cpp:116-137);
// opmhip_config.cpr_amg_ilu_levels = L gives the L finest levels of this hierarchy that smoother in place of damped Jacobi
// (oracle/cpr.hpp: CprAmg::iluLevels, ilu_factor, smooth - the same statements in the same order).  Elimination order: level 0 in
// its stored order (whatever ordering the block ILU0 uses: its colours are this one's), the levels below in a greedy multi-colouring
// of their graphs, colour by colour.  One launch per colour; a thread walks one SEQUENCE of rows that depend on each other inside
// the colour (the chains of a line-coloured level 0; single rows elsewhere).
template <bool RM> __device__ __forceinline__ size_t cpr_at(int j, int i, int n, int W) { return RM ? (size_t)i * W + j : (size_t)j * n + i; }
// the rows of one colour: row i against the finished rows j of its lower slots, in ascending elimination position (IKJ); fval is
// read where another thread (an earlier launch) or this thread (an earlier step) wrote it: no __restrict__
template <bool RM>
__global__ __launch_bounds__(256) void k_cpr_ilu_factor(int nseq, int nsteps, const int* __restrict__ rowAt, int n, int W, int MW, int WL,
                                                        const int* __restrict__ ecol, const int* __restrict__ rlen, const int* __restrict__ diag,
                                                        const unsigned* __restrict__ mask, const unsigned char* __restrict__ lorder, double* fval) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nseq) return;
    for (int st = 0; st < nsteps; ++st) {
        const int i = rowAt[(size_t)st * nseq + t];
        if (i < 0) break;
        const int len = rlen[i];
        for (int q = 0; q < WL; ++q) {
            const int slot = lorder[(size_t)q * n + i];
            if (slot == 255) break;
            const size_t e = cpr_at<RM>(slot, i, n, W);
            const int j = ecol[e];
            const double l = fval[e] * fval[diag[j]];
            fval[e] = l;
            const int lenj = rlen[j];
            for (int sj = 0; sj < lenj; ++sj) {   // the U part of row j, in the row's order
                if (!((mask[(size_t)(MW + (sj >> 5)) * n + j] >> (sj & 31)) & 1u)) continue;
                const size_t ej = cpr_at<RM>(sj, j, n, W);
                const int cc = ecol[ej];
                size_t tgt = (size_t)-1;
                if (cc == i) tgt = (size_t)diag[i];
                else
                    for (int si = 0; si < len; ++si) {
                        const size_t ei = cpr_at<RM>(si, i, n, W);
                        if (ecol[ei] == cc) { tgt = ei; break; }
                    }
                if (tgt != (size_t)-1) fval[tgt] -= l * fval[ej];
            }
        }
        fval[diag[i]] = 1.0 / fval[diag[i]];
    }
}
// The factors as the sweeps read them: per row its lower entries (value, column) in the row's order, its upper entries likewise and
// 1 / U_ii, each kind in a compact image of its own ([q * n + i], column -1 = none) - a forward sweep of a colour whose rows have
// one lower entry reads one, not the level's whole row.  Written once per factorisation.
__global__ __launch_bounds__(256) void k_cpr_ilu_pack(int n, int W, int MW, int rm, int WLc, int WUc, const int* __restrict__ ecol, const int* __restrict__ rlen,
                                                      const int* __restrict__ diag, const unsigned* __restrict__ mask, const double* __restrict__ fval,
                                                      double* __restrict__ lv, int* __restrict__ lc, double* __restrict__ uv, int* __restrict__ uc, double* __restrict__ ud) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int len = rlen[i];
    int ql = 0, qu = 0;
    for (int j = 0; j < len; ++j) {
        const size_t e = rm ? (size_t)i * W + j : (size_t)j * n + i;
        if ((mask[(size_t)(j >> 5) * n + i] >> (j & 31)) & 1u) { lv[(size_t)ql * n + i] = fval[e]; lc[(size_t)ql * n + i] = ecol[e]; ++ql; }
        else if ((mask[(size_t)(MW + (j >> 5)) * n + i] >> (j & 31)) & 1u) { uv[(size_t)qu * n + i] = fval[e]; uc[(size_t)qu * n + i] = ecol[e]; ++qu; }
    }
    for (; ql < WLc; ++ql) lc[(size_t)ql * n + i] = -1;
    for (; qu < WUc; ++qu) uc[(size_t)qu * n + i] = -1;
    ud[i] = fval[diag[i]];
}
// The factorisation of a SIMPLE level (CprIluHost::simple), one colour: lv holds the lower entries a_ij on entry (k_cpr_ilu_pack of the
// matrix itself) and l_ij = a_ij / u_jj on exit, ud receives 1 / u_ii with u_ii = a_ii - sum_j l_ij a_ji, the lower entries in the row's
// order - the statements of k_cpr_ilu_factor where nothing but the diagonal is ever updated.  Written like k_cpr_ilu_sweep_fast: the
// steps of a sequence G at a time, every load independent of what the walk computes (1 / u_jj of the row before travels in a register).
template <int WQ, int G>
__global__ __launch_bounds__(64) void k_cpr_ilu_factor_simple(int nseq, int nsteps, const int* __restrict__ rowAt, int n, int wq, const double* __restrict__ val,
                                                              const int* __restrict__ diag, const int* __restrict__ lc, const int* __restrict__ tpos,
                                                              double* __restrict__ lv, const double* __restrict__ udg, double* __restrict__ ud) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nseq) return;
    int iprev = -1;
    double dprev = 0.0;
    for (int g0 = 0; g0 < nsteps; g0 += G) {
        int ri[G];
#pragma unroll
        for (int u = 0; u < G; ++u) ri[u] = (g0 + u < nsteps) ? rowAt[(size_t)(g0 + u) * nseq + t] : -1;
        double a[G][WQ], at[G][WQ], dj[G][WQ], dd[G];
        int cc[G][WQ];
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int ic = ri[u] < 0 ? 0 : ri[u];
#pragma unroll
            for (int q = 0; q < WQ; ++q) {
                cc[u][q] = (q < wq && ri[u] >= 0) ? lc[(size_t)q * n + ic] : -1;
                a[u][q] = q < wq ? lv[(size_t)q * n + ic] : 0.0;
                const int tp = q < wq ? tpos[(size_t)q * n + ic] : -1;
                at[u][q] = (tp >= 0 && cc[u][q] >= 0) ? val[tp] : 0.0;
            }
            dd[u] = val[diag[ic]];
        }
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int nb = u == 0 ? iprev : ri[u - 1];
#pragma unroll
            for (int q = 0; q < WQ; ++q) dj[u][q] = (cc[u][q] >= 0 && cc[u][q] != nb) ? udg[cc[u][q]] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < G; ++u) {
            if (ri[u] < 0) continue;
            double sdiag = dd[u];
#pragma unroll
            for (int q = 0; q < WQ; ++q)
                if (cc[u][q] >= 0) {
                    const double l = a[u][q] * (cc[u][q] == iprev ? dprev : dj[u][q]);
                    lv[(size_t)q * n + ri[u]] = l;
                    sdiag -= l * at[u][q];
                }
            const double inv = 1.0 / sdiag;
            ud[ri[u]] = inv;
            iprev = ri[u];
            dprev = inv;
        }
    }
}
// One colour of a sweep; a thread walks its sequence of rows.  BWD = false: v_i = d_i - sum over the lower entries, in the row's
// order; BWD = true: v_i = (v_i - sum over the upper entries) / U_ii, steps in reverse; out != NULL (the backward sweeps of a
// post-smoothing): out_i = add_i + v_i, the statement "x += t" of the cycle (vfine: and the block vector (0, out_i, 0)).
// FAST (cpr_ilu_schedule: every coupling inside the colour joins neighbours of a sequence): the value of the row before is handed
// on in a register, every value read from memory was written by an earlier launch - gathers (vg) and stores (v) touch different
// entries, so the loads of later steps need not wait for the stores of earlier ones.  Otherwise values of the thread's own earlier
// steps come back through memory (vg == v, no __restrict__).  WQ: entries per row of this colour and direction at most.
template <bool BWD, bool FAST, int WQ>
__device__ __forceinline__ void cpr_ilu_sweep_body(int t, int nseq, int nsteps, const int* __restrict__ rowAt, int n, int wq, const double* __restrict__ fv,
                                                   const int* __restrict__ fc, const double* __restrict__ ud, const double* __restrict__ d, const double* vg,
                                                   double* v, const double* __restrict__ add, double* __restrict__ out, double* __restrict__ vfine) {
    int iprev = -1;
    double vprev = 0.0;
#pragma unroll 2
    for (int s0 = 0; s0 < nsteps; ++s0) {
        const int st = BWD ? nsteps - 1 - s0 : s0;
        const int i = rowAt[(size_t)st * nseq + t];
        if (i < 0) { if (BWD) continue; else break; }
        double s = BWD ? v[i] : d[i];
        if constexpr (WQ == 0) {   // long rows (coarse levels): entry by entry
            for (int q = 0; q < wq; ++q) {
                const int c = fc[(size_t)q * n + i];
                if (c < 0) break;
                s -= fv[(size_t)q * n + i] * ((FAST && c == iprev) ? vprev : vg[c]);
            }
        } else {
            double f[WQ], x[WQ];
            int cc[WQ];
#pragma unroll
            for (int q = 0; q < WQ; ++q) {
                cc[q] = q < wq ? fc[(size_t)q * n + i] : -1;
                f[q] = q < wq ? fv[(size_t)q * n + i] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < WQ; ++q) x[q] = (cc[q] >= 0 && !(FAST && cc[q] == iprev)) ? vg[cc[q]] : 0.0;
#pragma unroll
            for (int q = 0; q < WQ; ++q)
                if (cc[q] >= 0) s -= f[q] * ((FAST && cc[q] == iprev) ? vprev : x[q]);
        }
        if (BWD) s = s * ud[i];
        v[i] = s;
        if (FAST) { iprev = i; vprev = s; }
        if (BWD && out) {
            const double xo = add[i] + s;
            out[i] = xo;
            if (vfine) {
                double* vf = &vfine[(size_t)i * BS];
#pragma unroll
                for (int k = 0; k < BS; ++k) vf[k] = (k == CPR_P) ? xo : 0.0;
            }
        }
    }
}
// FAST colours, written for the shape the launch has: a colour of a line-coloured level 0 offers one thread per chain - 50 000 threads for
// 10^6 rows, not one wavefront per SIMD - so a walk that waits for memory at every step takes ten round trips however little it moves.
// Nothing a step loads depends on the steps before it (the value handed on travels in a register), so the steps are taken G at a time:
// all row numbers, then all entries, then all gathered values, then the recurrence - three rounds of loads per group instead of three
// per step.  One wavefront per workgroup: the registers are there (one wavefront per SIMD at most anyway).
template <bool BWD, int WQ, int G>
__global__ __launch_bounds__(64) void k_cpr_ilu_sweep_fast(int nseq, int nsteps, const int* __restrict__ rowAt, int n, int wq, const double* __restrict__ fv,
                                                           const int* __restrict__ fc, const double* __restrict__ ud, const double* __restrict__ d,
                                                           const double* __restrict__ vg, double* __restrict__ v, const double* __restrict__ add,
                                                           double* __restrict__ out, double* __restrict__ vfine, const double* __restrict__ done) {
    CPR_DONE_CHECK
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nseq) return;
    int iprev = -1;
    double vprev = 0.0;
    for (int g0 = 0; g0 < nsteps; g0 += G) {
        // step of slot u of this group: forward g0 + u, backward nsteps - 1 - g0 - u (descending)
        int ri[G];
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int st = BWD ? nsteps - 1 - g0 - u : g0 + u;
            ri[u] = (st >= 0 && st < nsteps) ? rowAt[(size_t)st * nseq + t] : -1;
        }
        double f[G][WQ], x[G][WQ], dd[G], du[G], da[G];
        int cc[G][WQ];
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int ic = ri[u] < 0 ? 0 : ri[u];
#pragma unroll
            for (int q = 0; q < WQ; ++q) {
                cc[u][q] = (q < wq && ri[u] >= 0) ? fc[(size_t)q * n + ic] : -1;
                f[u][q] = q < wq ? fv[(size_t)q * n + ic] : 0.0;
            }
            dd[u] = BWD ? v[ic] : d[ic];
            du[u] = BWD ? ud[ic] : 1.0;
            da[u] = (BWD && out) ? add[ic] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int nb = u == 0 ? iprev : ri[u - 1];   // the row before in walking order
#pragma unroll
            for (int q = 0; q < WQ; ++q) x[u][q] = (cc[u][q] >= 0 && cc[u][q] != nb) ? vg[cc[u][q]] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < G; ++u) {
            if (ri[u] < 0) continue;
            double s = dd[u];
#pragma unroll
            for (int q = 0; q < WQ; ++q)
                if (cc[u][q] >= 0) s -= f[u][q] * (cc[u][q] == iprev ? vprev : x[u][q]);
            if (BWD) s = s * du[u];
            v[ri[u]] = s;
            iprev = ri[u];
            vprev = s;
            if (BWD && out) {
                const double xo = da[u] + s;
                out[ri[u]] = xo;
                if (vfine) {
                    double* vf = &vfine[(size_t)ri[u] * BS];
#pragma unroll
                    for (int k = 0; k < BS; ++k) vf[k] = (k == CPR_P) ? xo : 0.0;
                }
            }
        }
    }
}
template <bool BWD, int WQ>
__global__ __launch_bounds__(256) void k_cpr_ilu_sweep(int nseq, int nsteps, const int* __restrict__ rowAt, int n, int wq, const double* __restrict__ fv,
                                                       const int* __restrict__ fc, const double* __restrict__ ud, const double* __restrict__ d,
                                                       double* v, const double* __restrict__ add, double* __restrict__ out, double* __restrict__ vfine,
                                                       const double* __restrict__ done) {
    CPR_DONE_CHECK
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nseq) cpr_ilu_sweep_body<BWD, false, WQ>(t, nseq, nsteps, rowAt, n, wq, fv, fc, ud, d, v, v, add, out, vfine);
}


// ---------------------------------------------------------------- the level that is continued across the ranks ------------
// pressure values of the couplings to ghost cells (k_cpr_pvals leaves them out of the subdomain's own system): entry of the joined level's
// matrix between aggregates of two subdomains = the sum of these over the fine couplings between them (Galerkin, piecewise-constant
// prolongation); k_cpr_pvals's statement
__global__ __launch_bounds__(256) void k_cpr_ghost_pvals(int nq, const int* __restrict__ qrow, const int* __restrict__ qentry, const double* __restrict__ A,
                                                         const double* __restrict__ w, double* __restrict__ apg) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const int i = qrow[q];
    const double* B = &A[(size_t)qentry[q] * BB];
    double s = 0.0;
    s += B[0 * BS + CPR_P] * w[(size_t)i * BS]; s += B[1 * BS + CPR_P] * w[(size_t)i * BS + 1]; s += B[2 * BS + CPR_P] * w[(size_t)i * BS + 2];
    apg[q] = s;
}
// my entries of the joined level's matrix: the value of my last level's entry, or the sum over the fine couplings of a pair of aggregates
__global__ __launch_bounds__(256) void k_cpr_gather_vals(int nnzloc, const int* __restrict__ src, const int* __restrict__ xptr, const int* __restrict__ xidx,
                                                         const double* __restrict__ apg, const double* __restrict__ lastval, double* __restrict__ vsend) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnzloc) return;
    const int sidx = src[e];
    if (sidx >= 0) { vsend[e] = lastval[sidx]; return; }
    const int x = -1 - sidx;
    double s = 0.0;
    for (int t = xptr[x]; t < xptr[x + 1]; ++t) s += apg[xidx[t]];
    vsend[e] = s;
}
__global__ __launch_bounds__(256) void k_cpr_scatter_vals(int nnzG, const int* __restrict__ vunpad, const int* __restrict__ vpos, const double* __restrict__ vrecv,
                                                          double* __restrict__ val) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < nnzG) val[vpos[e]] = vrecv[vunpad[e]];
}
// the gathered right-hand sides, slice by slice, into the joined level's b; x0 != NULL: its pre-smoothed iterate rides along (k_cpr_presmooth's statement)
__global__ __launch_bounds__(256) void k_cpr_unpad(int NG, const int* __restrict__ unpad, const double* __restrict__ recv, double* __restrict__ b, double omega,
                                                   const double* __restrict__ dinv, double* __restrict__ x0, const double* __restrict__ done) {
    CPR_DONE_CHECK
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NG) return;
    const double s = recv[unpad[i]];
    b[i] = s;
    if (x0) x0[i] = omega * dinv[i] * s;
}

static inline dim3 g256(int n) { return dim3((n + 255) / 256); }

// ---------------------------------------------------------------- setup --------------------------------------------------
// Host-side description of the hierarchy (cpr_build_coarse_host): everything the device arrays of a level are uploaded from.
// The set-up is split in two so that its expensive half - matching, Galerkin lists, level images: pure host work on copies -
// can run on a thread of its own beside the solves (--cpr-reuse-setup=2 with opmhip_config.cpr_async_setup), and only the uploads
// touch the context.
// ILU0 smoothing schedule of one level, from its image: which slots of a row are lower / upper entries in the elimination order
// `pos`, the lower slots in ascending position, and per colour the sequences of rows a thread walks.  colour[i] ascending = the
// order of the launches; rows of one colour may depend on each other only along a sequence (a chain of level 0's line colouring).
struct CprIluHost {
    int ncol = 0, MW = 0, WL = 0, WU = 0;
    std::vector<int> nseq, nsteps, off, rowAt, wl, wu;   // wl / wu: lower / upper entries per row of a colour at most
    std::vector<char> fast;
    std::vector<unsigned> mask;
    std::vector<unsigned char> lorder;
    // simple: a level whose elimination steps touch nothing but the diagonal (no triangles in its graph: step (i, j) finds of row j's
    // upper entries only (j, i) in row i - a seven-point grid in any of the orderings here), whose couplings inside a colour join
    // neighbours of a sequence and whose rows hold their lower entries in elimination order: U keeps the matrix's own values,
    // l_ij = a_ij / u_jj, u_ii = a_ii - sum_j l_ij a_ji - a recurrence along the sequences with loads that depend on nothing it computes
    // (k_cpr_ilu_factor_simple).  tpos: per lower entry (slot order) the place of the transposed entry (j, i) in the level's image
    bool simple = false;
    std::vector<int> tpos;
    std::string error;
};
struct CprHostLevel {
    int n = 0, nnz = 0, nc = 0, W = 0;
    bool rm = false;
    std::vector<int> ecol, rlen, diag;                           // ELL image of the level's pattern
    std::vector<int> agg, mptr, midx, mem4, gptr, gidx, cpos;    // transfer to the next level (empty on the coarsest)
    CprIluHost ilu;                                              // ncol > 0: the level's ILU0 smoothing schedule
};
struct CprHostCoarse {
    CprHostLevel l0;                 // of level 0 only the transfer part (its image belongs to the pattern: cpr_setup_level0)
    std::vector<CprHostLevel> lv;    // levels 1 ..
    double tAgg = 0.0, tGal = 0.0, tImg = 0.0;
    std::string error;               // non-empty: the build failed
    HCsr lastA;                      // the last level's matrix and the place of its entries in that level's image
    std::vector<int> lastPos;
};
struct CprAsyncJob {
    std::thread th;
    std::atomic<int> ready{0};
    CprHostCoarse result;
    ~CprAsyncJob() { if (th.joinable()) th.join(); }
};
// ELL image of a level's pattern: columns (padding: the row itself), row lengths, position of the diagonal, position of every
// CSR entry
static bool ell_image(const HCsr& A, CprHostLevel& L, std::vector<int>& pos, bool rowMajor, int ncols = INT_MAX) {
    const int n = A.n;
    int W = 1;
    for (int i = 0; i < n; ++i) W = std::max(W, A.rowptr[i + 1] - A.rowptr[i]);
    L.n = n; L.nnz = (int)A.col.size(); L.W = W; L.rm = rowMajor;
    if (W > CPR_MAX_W) return false;
    // entry j of row i: [j * n + i] (one thread per row reads coalesced) or, row-major, [i * W + j] (a group of lanes per row does)
    auto at = [&](int j, int i) { return rowMajor ? (size_t)i * W + j : (size_t)j * n + i; };
    L.ecol.assign((size_t)W * n, 0); L.rlen.assign(n, 0); L.diag.assign(n, 0);
    pos.resize(A.col.size());
    for (int i = 0; i < n; ++i) {
        const int kb = A.rowptr[i], len = A.rowptr[i + 1] - kb;
        L.rlen[i] = len;
        for (int j = 0; j < W; ++j) L.ecol[at(j, i)] = (j < len && A.col[kb + j] < ncols) ? A.col[kb + j] : i;   // padding and ghost columns (value 0 for good): the row itself
        for (int j = 0; j < len; ++j) {
            pos[kb + j] = (int)at(j, i);
            if (A.col[kb + j] == i) L.diag[i] = (int)at(j, i);
        }
    }
    return true;
}

static void cpr_ilu_schedule(const CprHostLevel& L, const std::vector<int>& pos, const std::vector<int>& colour, int ncol, CprIluHost& S) {
    const int n = L.n, W = L.W;
    auto at = [&](int j, int i) { return L.rm ? (size_t)i * W + j : (size_t)j * n + i; };
    S.ncol = ncol;
    S.MW = (W + 31) / 32;
    S.mask.assign((size_t)2 * S.MW * n, 0u);
    std::vector<std::vector<std::pair<int, int>>> low(n);   // (position, slot) of every lower entry
    int WL = 0;
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < L.rlen[i]; ++j) {
            const int c = L.ecol[at(j, i)];
            if (c == i) continue;   // the diagonal, or a ghost column's slot (value 0 for good)
            if (pos[c] < pos[i]) { S.mask[(size_t)(j >> 5) * n + i] |= 1u << (j & 31); low[i].emplace_back(pos[c], j); }
            else S.mask[(size_t)(S.MW + (j >> 5)) * n + i] |= 1u << (j & 31);
        }
        std::sort(low[i].begin(), low[i].end());
        WL = std::max(WL, (int)low[i].size());
    }
    S.WL = std::max(WL, 1);
    if (W > 254) { S.error = "cpr: ILU0 smoothing of a level with rows of more than 254 entries"; return; }
    S.lorder.assign((size_t)S.WL * n, 255);
    for (int i = 0; i < n; ++i)
        for (size_t q = 0; q < low[i].size(); ++q) S.lorder[q * n + i] = (unsigned char)low[i][q].second;
    // sequences: rows of a colour in ascending position; a row with a lower entry of its own colour continues that row's sequence
    std::vector<int> byPos(n);
    for (int i = 0; i < n; ++i) byPos[pos[i]] = i;
    std::vector<int> seqOf(n, -1), idxIn(n, 0);
    std::vector<std::vector<std::vector<int>>> seqs(ncol);
    for (int p = 0; p < n; ++p) {
        const int i = byPos[p], cc = colour[i];
        int sq = -1;
        for (auto& e : low[i]) {
            const int j = L.ecol[at(e.second, i)];
            if (colour[j] != cc) {
                if (colour[j] > cc) { S.error = "cpr: ILU0 smoothing: the colours are not an elimination order"; return; }
                continue;
            }
            if (sq >= 0 && seqOf[j] != sq) { S.error = "cpr: ILU0 smoothing: a row depends on two sequences of its own colour"; return; }
            sq = seqOf[j];
        }
        if (sq < 0) { sq = (int)seqs[cc].size(); seqs[cc].emplace_back(); }
        seqOf[i] = sq;
        idxIn[i] = (int)seqs[cc][sq].size();
        seqs[cc][sq].push_back(i);
    }
    S.wl.assign(ncol, 0); S.wu.assign(ncol, 0);
    for (int i = 0; i < n; ++i) {
        int nu = 0;
        for (int w = 0; w < S.MW; ++w) nu += __builtin_popcount(S.mask[(size_t)(S.MW + w) * n + i]);
        S.wl[colour[i]] = std::max(S.wl[colour[i]], (int)low[i].size());
        S.wu[colour[i]] = std::max(S.wu[colour[i]], nu);
        S.WU = std::max(S.WU, nu);
    }
    S.WU = std::max(S.WU, 1);
    S.nseq.assign(ncol, 0); S.nsteps.assign(ncol, 0); S.off.assign(ncol + 1, 0); S.fast.assign(ncol, 1);
    for (int cc = 0; cc < ncol; ++cc) {
        int steps = 0;
        for (auto& q : seqs[cc]) steps = std::max(steps, (int)q.size());
        S.nseq[cc] = (int)seqs[cc].size();
        S.nsteps[cc] = steps;
        S.off[cc + 1] = S.off[cc] + steps * S.nseq[cc];
    }
    S.rowAt.assign(S.off[ncol], -1);
    for (int cc = 0; cc < ncol; ++cc)
        for (int t = 0; t < S.nseq[cc]; ++t)
            for (size_t st = 0; st < seqs[cc][t].size(); ++st) S.rowAt[S.off[cc] + st * S.nseq[cc] + t] = seqs[cc][t][st];
    // fast: every coupling inside the colour joins neighbours of a sequence (the sweeps may then hand the value on in a register)
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < L.rlen[i]; ++j) {
            const int c = L.ecol[at(j, i)];
            if (c == i || colour[c] != colour[i]) continue;
            if (seqOf[c] != seqOf[i] || std::abs(idxIn[c] - idxIn[i]) != 1) S.fast[colour[i]] = 0;
        }
    // simple?
    bool simple = !L.rm;
    for (int cc = 0; cc < ncol && simple; ++cc) simple = S.fast[cc] != 0;
    std::vector<int> tpos(simple ? (size_t)S.WL * n : 0, -1);
    for (int i = 0; i < n && simple; ++i) {
        int q = 0, lastPos = -1;
        for (int j = 0; j < L.rlen[i] && simple; ++j) {
            if (!((S.mask[(size_t)(j >> 5) * n + i] >> (j & 31)) & 1u)) continue;
            const int cj = L.ecol[at(j, i)];
            if (pos[cj] < lastPos) simple = false;   // the row's order is not the elimination order
            lastPos = pos[cj];
            // row cj's upper entries that row i holds too: must be (cj, i) alone
            int found = -1;
            for (int t = 0; t < L.rlen[cj] && simple; ++t) {
                if (!((S.mask[(size_t)(S.MW + (t >> 5)) * n + cj] >> (t & 31)) & 1u)) continue;
                const int ct = L.ecol[at(t, cj)];
                if (ct == i) { found = (int)at(t, cj); continue; }
                for (int u = 0; u < L.rlen[i]; ++u)
                    if (L.ecol[at(u, i)] == ct && (int)at(u, i) != L.diag[i] && ct != i) { simple = false; break; }
            }
            if (found < 0) simple = false;   // (an unsymmetric pattern)
            if (simple) tpos[(size_t)q * n + i] = found;
            ++q;
        }
    }
    S.simple = simple;
    if (simple) S.tpos = std::move(tpos);
}
// greedy multi-colouring of a level's graph in index order, colour-major elimination positions (oracle/cpr.hpp: ilu_factor, colour = true)
static int cpr_greedy_colours(const CprHostLevel& L, std::vector<int>& colour, std::vector<int>& pos) {
    const int n = L.n, W = L.W;
    auto at = [&](int j, int i) { return L.rm ? (size_t)i * W + j : (size_t)j * n + i; };
    colour.assign(n, -1);
    int nc = 0;
    std::vector<char> used;
    for (int i = 0; i < n; ++i) {
        used.assign(nc + 1, 0);
        for (int j = 0; j < L.rlen[i]; ++j) {
            const int c = L.ecol[at(j, i)];
            if (colour[c] >= 0) used[colour[c]] = 1;
        }
        int k = 0;
        while (used[k]) ++k;
        colour[i] = k;
        nc = std::max(nc, k + 1);
    }
    pos.assign(n, 0);
    int p = 0;
    for (int k = 0; k < nc; ++k)
        for (int i = 0; i < n; ++i)
            if (colour[i] == k) pos[i] = p++;
    return nc;
}
static int cpr_upload_ilu(opmhip_ctx* c, const CprIluHost& S, CprLevelDev& L) {
    if (!S.error.empty()) return fail(c, OPMHIP_ANALYSIS_FAILED, "%s", S.error.c_str());
    int rc;
    L.iluMW = S.MW; L.iluWL = S.WL; L.iluWU = S.WU;
    L.iluNseq = S.nseq; L.iluNsteps = S.nsteps; L.iluOff = S.off; L.iluFast = S.fast; L.iluWl = S.wl; L.iluWu = S.wu;
    if ((rc = dev_alloc(c, &L.d_ilv, (size_t)S.WL * L.n))) return rc;
    if ((rc = dev_alloc(c, &L.d_ilc, (size_t)S.WL * L.n))) return rc;
    if ((rc = dev_alloc(c, &L.d_iuv, (size_t)S.WU * L.n))) return rc;
    if ((rc = dev_alloc(c, &L.d_iuc, (size_t)S.WU * L.n))) return rc;
    if ((rc = dev_alloc(c, &L.d_iud, (size_t)L.n))) return rc;
    if ((rc = dev_upload(c, &L.d_imask, S.mask))) return rc;
    if ((rc = dev_upload(c, &L.d_lorder, S.lorder))) return rc;
    if ((rc = dev_upload(c, &L.d_rowAt, S.rowAt))) return rc;
    L.iluSimple = S.simple;
    if (S.simple) { if ((rc = dev_upload(c, &L.d_tpos, S.tpos))) return rc; }
    else if ((rc = dev_alloc(c, &L.d_fval, (size_t)L.W * L.n))) return rc;
    if ((rc = dev_alloc(c, &L.d_t, (size_t)L.n))) return rc;
    L.ilu = true;
    return OPMHIP_SUCCESS;
}
static void cpr_free_ilu(opmhip_ctx* c, CprLevelDev& L) {
    dev_free(c, &L.d_imask); dev_free(c, &L.d_lorder); dev_free(c, &L.d_rowAt); dev_free(c, &L.d_fval); dev_free(c, &L.d_t);
    dev_free(c, &L.d_ilv); dev_free(c, &L.d_ilc); dev_free(c, &L.d_iuv); dev_free(c, &L.d_iuc); dev_free(c, &L.d_iud); dev_free(c, &L.d_tpos);
    L.ilu = false;
}
// nvec: entries of the level's vectors (level 0 of a rank whose pressure stage spans the ranks: ghost cells included)
static int upload_ell(opmhip_ctx* c, const CprHostLevel& H, CprLevelDev& L, int nvec = 0) {
    const int n = H.n, W = H.W;
    if (nvec < n) nvec = n;
    if (W > CPR_MAX_W) return fail(c, OPMHIP_ANALYSIS_FAILED, "cpr: a row of a pressure-AMG level has %d entries (limit %d)", W, CPR_MAX_W);
    L.n = n; L.nnz = H.nnz; L.W = W; L.rm = H.rm;
    int rc;
    if ((rc = dev_upload(c, &L.d_ecol, H.ecol))) return rc;
    if ((rc = dev_upload(c, &L.d_rlen, H.rlen))) return rc;
    if ((rc = dev_upload(c, &L.d_diag, H.diag))) return rc;
    if ((rc = dev_alloc(c, &L.d_val, (size_t)W * n))) return rc;
    // the padding stays 0 for good.  On the context's stream, like the kernels that write the level's values behind it: a hipMemset - the
    // device's NULL stream, which a non-blocking stream is not ordered with - could land behind them (capi.cpp, alloc_system)
    OPMHIP_HIP(c, hipMemsetAsync(L.d_val, 0, (size_t)W * n * sizeof(double), c->stream));
    if ((rc = dev_alloc(c, &L.d_dinv, (size_t)n))) return rc;
    if ((rc = dev_alloc(c, &L.d_b, (size_t)nvec))) return rc;
    OPMHIP_HIP(c, hipMemsetAsync(L.d_b, 0, (size_t)nvec * sizeof(double), c->stream));
    if ((rc = dev_alloc(c, &L.d_x, (size_t)nvec))) return rc;
    OPMHIP_HIP(c, hipMemsetAsync(L.d_x, 0, (size_t)nvec * sizeof(double), c->stream));
    if ((rc = dev_alloc(c, &L.d_x2, (size_t)nvec))) return rc;
    OPMHIP_HIP(c, hipMemsetAsync(L.d_x2, 0, (size_t)nvec * sizeof(double), c->stream));
    if ((rc = dev_alloc(c, &L.d_r, (size_t)nvec))) return rc;
    OPMHIP_HIP(c, hipMemsetAsync(L.d_r, 0, (size_t)nvec * sizeof(double), c->stream));
    return OPMHIP_SUCCESS;
}
// True-IMPES weights (getQuasiImpesWeights.hpp:89-128): block[ii][jj] = d storage_ii / d x_jj / (V / dt), pressure column
// times 50e5, block^T w = e_p, w /= 1000 - from the intensive quantities of the state on the device.  The storage term is
// the one of the assembly kernel's diagonal lane (assemble.hip: same statements, same order); the 3 x 3 solve is the pivoted
// elimination of oracle/cpr.hpp: true_impes_weights_cell, statement by statement.  Field numbers and the field-major
// addressing of the cache: assemble.hip (F_S .. F_RS, Lay<EXT>, iq_at).
struct WAd { double v, d[3]; };
__device__ __forceinline__ WAd wad_load(const double* iq, int ncell, int f, int c) {
    const double* o = iq + ((size_t)f * ncell + c) * 4;
    return WAd{o[0], {o[1], o[2], o[3]}};
}
__device__ __forceinline__ WAd wad_mul(const WAd& a, const WAd& b) {
    const double u = a.v, w = b.v;
    return WAd{u * w, {a.d[0] * w + b.d[0] * u, a.d[1] * w + b.d[1] * u, a.d[2] * w + b.d[2] * u}};
}
__device__ __forceinline__ WAd wad_add(const WAd& a, const WAd& b) { return WAd{a.v + b.v, {a.d[0] + b.d[0], a.d[1] + b.d[1], a.d[2] + b.d[2]}}; }
template <bool EXT>
__global__ __launch_bounds__(256) void k_cpr_true_weights(int Nb, int ncell, int wet, const double* __restrict__ iq, const double* __restrict__ volume,
                                                          double dt, double* __restrict__ w) {
    constexpr int F_S = 0, F_B = 6, F_RS = 15, F_RV = 16, F_PORO = EXT ? 18 : 16;
    constexpr int WATER = 0, OIL = 1, GAS = 2, EQ_OIL = 0, EQ_WATER = 1, EQ_GAS = 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Nb) return;
    const WAd poro = wad_load(iq, ncell, F_PORO, i), Rs = wad_load(iq, ncell, F_RS, i);
    const WAd zero{0.0, {0.0, 0.0, 0.0}};
    WAd st[3] = {zero, zero, zero};
    const int comp[3] = {EQ_WATER, EQ_OIL, EQ_GAS};
#pragma unroll
    for (int ph = 0; ph < 3; ++ph) {   // computeStorage: surface volumes per bulk volume
        const WAd surfaceVolume = wad_mul(wad_mul(wad_load(iq, ncell, F_S + ph, i), wad_load(iq, ncell, F_B + ph, i)), poro);
        st[comp[ph]] = wad_add(st[comp[ph]], surfaceVolume);
        if (ph == OIL) st[EQ_GAS] = wad_add(st[EQ_GAS], wad_mul(Rs, surfaceVolume));
        if (EXT && ph == GAS && wet) st[EQ_OIL] = wad_add(st[EQ_OIL], wad_mul(wad_load(iq, ncell, F_RV, i), surfaceVolume));
    }
    (void)WATER;
    const double storage_scale = volume[i] / dt, pressure_scale = 50e5;
    double M[BS][BS + 1];
    for (int r = 0; r < BS; ++r) {          // M = block^T | e_p
        for (int c = 0; c < BS; ++c) {
            double v = st[c].d[r] / storage_scale;
            if (r == CPR_P) v = v * pressure_scale;
            M[r][c] = v;
        }
        M[r][BS] = (r == CPR_P) ? 1.0 : 0.0;
    }
#pragma unroll
    for (int k = 0; k < BS; ++k) {
        int piv = k;
        double best = fabs(M[k][k]);
#pragma unroll
        for (int r = k + 1; r < BS; ++r)
            if (fabs(M[r][k]) > best) { best = fabs(M[r][k]); piv = r; }
#pragma unroll
        for (int r = k + 1; r < BS; ++r)    // the swap, written without a run-time row index
            if (piv == r)
                for (int c = 0; c <= BS; ++c) { const double t = M[k][c]; M[k][c] = M[r][c]; M[r][c] = t; }
#pragma unroll
        for (int r = k + 1; r < BS; ++r) {
            const double f = M[r][k] / M[k][k];
            for (int c = k; c <= BS; ++c) M[r][c] = M[r][c] - f * M[k][c];
        }
    }
    double x[BS];
#pragma unroll
    for (int r = BS - 1; r >= 0; --r) {
        double s = M[r][BS];
        for (int c = r + 1; c < BS; ++c) s = s - M[r][c] * x[c];
        x[r] = s / M[r][r];
    }
    for (int r = 0; r < BS; ++r) w[(size_t)i * BS + r] = x[r] / 1000.0;
}
// the weights of this solve: handed in (kept), true-IMPES from the model's state, or quasi-IMPES from the matrix
static int cpr_weights(opmhip_ctx* c) {
    const Pattern& P = c->pat;
    CprDev& R = c->cpr;
    if (R.w_given) return OPMHIP_SUCCESS;
    if (c->cfg.preconditioner == OPMHIP_PRECOND_CPR_TRUEIMPES) {
        const AsmDev& A = c->asmb;
        if (!A.static_set || !A.d_iq || !(A.last_dt > 0.0))
            return fail(c, OPMHIP_NOT_READY, "cpr (true-IMPES weights): no assembled state on this context - call opmhip_assemble first or hand the weights in (opmhip_set_cpr_weights)");
        if (A.ext) hipLaunchKernelGGL(k_cpr_true_weights<true>, g256(P.Nb), dim3(256), 0, c->stream, P.Nb, P.Nloc, A.wet_gas ? 1 : 0, A.d_iq, A.d_volume, A.last_dt, R.d_w);
        else hipLaunchKernelGGL(k_cpr_true_weights<false>, g256(P.Nb), dim3(256), 0, c->stream, P.Nb, P.Nloc, 0, A.d_iq, A.d_volume, A.last_dt, R.d_w);
        return OPMHIP_SUCCESS;
    }
    hipLaunchKernelGGL(k_cpr_weights, g256(P.Nb), dim3(256), 0, c->stream, P.Nb, P.d_diag, c->d_A, R.d_w);
    return OPMHIP_SUCCESS;
}
// The rider of this solve's factorisation (FactorRider, solver.hip: k_ilu_factor<RIDER>): level 0's image is set up if it is not yet,
// weights that do not come from the matrix (true-IMPES from the state, or handed in) are formed now, quasi-IMPES weights ride along.
// OPMHIP_CPR_PVALS_SEPARATE=1 (measurement switch): the passes of their own, as before.
static int cpr_setup_level0(opmhip_ctx* c);
static bool cpr_gathering(const opmhip_ctx* c);
int cpr_factor_rider(opmhip_ctx* c, FactorRider* r) {
    CprDev& R = c->cpr;
    *r = FactorRider();
    static const bool separate = tuning_env("OPMHIP_CPR_PVALS_SEPARATE") != nullptr;
    if (separate || !use_cpr(c)) return OPMHIP_SUCCESS;
    int rc;
    if (!R.level0 && (rc = cpr_setup_level0(c))) return rc;
    const bool fromMatrix = !R.w_given && c->cfg.preconditioner != OPMHIP_PRECOND_CPR_TRUEIMPES;
    if (!fromMatrix && (rc = cpr_weights(c))) return rc;
    r->mode = fromMatrix ? 2 : 1;
    r->w = R.d_w;
    r->ap = R.lv[0].d_val;
    r->pcol = R.d_pcol;
    r->W = R.lv[0].W;
    r->ghostFrom = (c->pat.Nghost > 0 && !cpr_gathering(c)) ? c->pat.Nb : INT_MAX;
    // (CprDev::pvals_fresh is raised by the caller once the factorisation that carries this rider has run without an error: a solve that
    //  leaves before that must not make the next cpr_update skip its own pass over the matrix)
    return OPMHIP_SUCCESS;
}
// levels of up to this many rows are kept row-major and run the lane-group kernels (OPMHIP_CPR_LPR_ROWS: measurement switch)
static int cpr_lpr_rows() {
    static const int v = [] { const char* e = tuning_env("OPMHIP_CPR_LPR_ROWS"); return e ? std::atoi(e) : CPR_LPR_ROWS; }();
    return v;
}
static bool cpr_gathering(const opmhip_ctx* c);
// opmhip_config.cpr_amg_ilu_levels as it is in force: < 0 = the library's choice - level 0 where the block ILU0's ordering has few colours
// (a sweep of level 0 is one launch per colour: four launches per application with two colours, +5 ... +8 % Newton iterations/s on the
// 10^6-cell case; with the seven colours of a greedy-coloured corner-point grid 28, and Jacobi is 1.5 x faster there), Jacobi otherwise
static int cpr_ilu_levels(const opmhip_ctx* c) {
    const int v = c->cfg.cpr_amg_ilu_levels;
    return v >= 0 ? v : (c->pat.numColors <= 3 ? 1 : 0);
}
int cpr_ilu_levels_in_force(const opmhip_ctx* c) { return use_cpr(c) ? (cpr_gathering(c) ? 0 : cpr_ilu_levels(c)) : 0; }
// ---- level 0: belongs to the PATTERN (image, stencil form, block-vector work space): built once per context ------------------
static int cpr_setup_level0_body(opmhip_ctx* c);
// a set-up that fails half way (an ILU schedule the level cannot take, a failed allocation) gives back what it allocated: the retry of
// the next solve starts from a clean slate and nothing piles up in the context
static int cpr_setup_level0(opmhip_ctx* c) {
    CprDev& R = c->cpr;
    const size_t mark = c->allocs.size();
    const bool hadW = R.d_w != nullptr;
    const int rc = cpr_setup_level0_body(c);
    if (rc) {
        (void)hipStreamSynchronize(c->stream);
        while (c->allocs.size() > mark) { (void)hipFree(c->allocs.back()); c->allocs.pop_back(); }
        if (!hadW) R.d_w = nullptr;
        R.d_r = R.d_y = R.d_z = R.d_pcol = nullptr;
        R.lv.clear();
        R.level0 = false;
    }
    return rc;
}
static int cpr_setup_level0_body(opmhip_ctx* c) {
    const Pattern& P = c->pat;
    CprDev& R = c->cpr;
    int rc;
    // Decomposed runs: every subdomain has a CPR of its own, over its owned rows and columns - the pressure hierarchy, like the
    // block ILU0 (ParallelOverlappingILU0.hpp:439-494), leaves the couplings to ghost cells out; the Krylov method carries them.
    // (The reference's parallel CPR coarsens ACROSS the processes with Dune's parallel AMG; this one does not.)
    if (!R.d_w && (rc = dev_alloc(c, &R.d_w, (size_t)P.Nb * BS))) return rc;
    if ((rc = dev_alloc(c, &R.d_r, (size_t)P.Nb * BS))) return rc;
    if ((rc = dev_alloc(c, &R.d_y, (size_t)P.Nb * BS))) return rc;
    if ((rc = dev_alloc(c, &R.d_z, (size_t)P.Nb * BS))) return rc;
    HCsr A;
    A.n = P.Nb; A.rowptr = P.rowptr; A.col = P.col;
    R.lv.clear();
    R.lv.emplace_back();
    CprHostLevel H0;
    std::vector<int> pos;
    const bool spans = cpr_gathering(c);   // the pressure stage spans the ranks: level 0 keeps its couplings to ghost cells (columns Nb ..), its vectors their ghost entries
    (void)ell_image(A, H0, pos, false, spans ? P.Nloc : P.Nb);
    if ((rc = upload_ell(c, H0, R.lv[0], spans ? P.Nloc : P.Nb))) return rc;
    {   // level 0's ELL columns in stencil form (EllStencil), where the pattern has it: single domain, rows of <= 8 entries, <= 15 offsets per group of 32 rows
        static const bool off = [] { const char* e = tuning_env("OPMHIP_CPR_ELL_EXPLICIT"); return e && e[0] == '1'; }();   // A/B switch
        bool ok = !off && P.Nghost == 0 && R.lv[0].W <= 8;
        const int ng = (P.Nb + 31) / 32;
        std::vector<unsigned> word(ok ? P.Nb : 0, 0xFFFFFFFFu);
        std::vector<int> table(ok ? (size_t)16 * ng : 0, 0);
        for (int g = 0; g < ng && ok; ++g) {
            const int r0 = 32 * g, r1 = std::min(P.Nb, r0 + 32);
            std::vector<int> offs;
            for (int r = r0; r < r1; ++r)
                for (int k = P.rowptr[r]; k < P.rowptr[r + 1]; ++k) offs.push_back(P.col[k] - r);
            std::sort(offs.begin(), offs.end());
            offs.erase(std::unique(offs.begin(), offs.end()), offs.end());
            if (offs.size() > 15) { ok = false; break; }
            for (size_t q = 0; q < offs.size(); ++q) table[(size_t)16 * g + q] = offs[q];
            for (int r = r0; r < r1; ++r) {
                unsigned w = 0xFFFFFFFFu;
                for (int u = 0; u < P.rowptr[r + 1] - P.rowptr[r]; ++u) {
                    const int idx = (int)(std::lower_bound(offs.begin(), offs.end(), P.col[P.rowptr[r] + u] - r) - offs.begin());
                    w = (w & ~(0xFu << (4 * u))) | ((unsigned)idx << (4 * u));
                }
                word[r] = w;
            }
        }
        if (ok) {
            if ((rc = dev_upload(c, &R.lv[0].d_sword, word))) return rc;
            if ((rc = dev_upload(c, &R.lv[0].d_stable, table))) return rc;
        }
    }
    if (cpr_ilu_levels(c) > 0 && !spans) {   // ILU0 smoothing of level 0: in the stored order, the block ILU0's colours are this one's
        std::vector<int> posv(P.Nb), colour(P.Nb);
        for (int cc = 0; cc < P.numColors; ++cc)
            for (int p = P.colorPrefix[cc]; p < P.colorPrefix[cc + 1]; ++p) colour[p] = cc;
        std::iota(posv.begin(), posv.end(), 0);
        CprIluHost S;
        cpr_ilu_schedule(H0, posv, colour, P.numColors, S);
        if ((rc = cpr_upload_ilu(c, S, R.lv[0]))) return rc;
    }
    if ((rc = dev_alloc(c, &R.d_pcol, (size_t)3 * R.lv[0].W * P.Nb))) return rc;     // pressure columns of the blocks, ELL, component-major
    OPMHIP_HIP(c, hipMemsetAsync(R.d_pcol, 0, (size_t)3 * R.lv[0].W * P.Nb * sizeof(double), c->stream));   // the padding stays 0
    R.level0 = true;
    return OPMHIP_SUCCESS;
}
// The coarsening itself: A = the finest level of the hierarchy being built (CSR with values), pos = the place of every entry of A in
// that level's image on the device; natOf / atNat: see pairwise (level 0 of a reordered system, else NULL).  stopRows: a level of at
// most this many rows is the last one (CPR_COARSE_DIRECT: it is solved directly; a rank whose hierarchy is continued across the ranks
// stops at opmhip_config.cpr_gather_rows).  The last level's matrix stays in out.lastA / lastPos.
static void cpr_coarsen_host(HCsr A, std::vector<int> pos, const int* natOf, const int* atNat, double beta, int lprRows, int iluLevels, int stopRows, CprHostCoarse& out) {
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
#define CPR_T(acc, stmt) do { const double t_ = now(); stmt; acc += now() - t_; } while (0)
    out.lv.clear();
    CprHostLevel* cur = &out.l0;   // the level being coarsened (its transfer part is filled here)
    cur->n = A.n;
    int nlev = 1;
    while (true) {
        const bool last = A.n <= stopRows || nlev >= CPR_MAX_LEVELS;
        if (last) break;
        std::vector<int> a1, a2, g1p, g1i;
        int n1 = 0, n2 = 0;
        HCsr A1;
        for (int attempt = 0; attempt < 3; ++attempt) {
            const double b = attempt == 0 ? beta : 0.0;
            const bool lvl0 = nlev == 1;   // the finest level is stored in the ILU ordering: visit it in natural order
            CPR_T(out.tAgg, pairwise(A, b, attempt == 2, a1, n1, lvl0 ? natOf : nullptr, lvl0 ? atNat : nullptr));
            CPR_T(out.tGal, galerkin(A, a1, n1, A1, g1p, g1i));
            CPR_T(out.tAgg, pairwise(A1, b, attempt == 2, a2, n2));
            if (n2 <= (int)(0.5 * A.n)) break;
        }
        if (n2 >= (int)(0.8 * A.n)) break;   // coarsening stalls: this level is the coarsest
        std::vector<int> agg(A.n);
        for (int i = 0; i < A.n; ++i) agg[i] = a2[a1[i]];
        HCsr Ac;
        std::vector<int> gptr, gidx;
        CPR_T(out.tGal, galerkin(A, agg, n2, Ac, gptr, gidx));
        {   // a coarse level whose rows outgrow the ELL image (fault- and NNC-heavy patterns): stop here, this level is the coarsest
            int Wc = 1;
            for (int I = 0; I < Ac.n; ++I) Wc = std::max(Wc, Ac.rowptr[I + 1] - Ac.rowptr[I]);
            if (Wc > CPR_MAX_W) break;
        }
        for (int& g : gidx) g = pos[g];                       // gather lists address the fine level's ELL array
        std::vector<int> mptr(n2 + 1, 0), midx(A.n);
        for (int i = 0; i < A.n; ++i) mptr[agg[i] + 1]++;
        for (int I = 0; I < n2; ++I) mptr[I + 1] += mptr[I];
        {
            std::vector<int> wpos(mptr.begin(), mptr.end() - 1);
            for (int i = 0; i < A.n; ++i) midx[wpos[agg[i]]++] = i;
        }
        cur->nc = n2;
        {   // four-int member records (two pairwise passes: never more than four members) for cpr_restricted
            std::vector<int> mem4((size_t)4 * n2, -1);
            bool fits = true;
            for (int I = 0; I < n2 && fits; ++I) {
                fits = mptr[I + 1] - mptr[I] <= 4 && mptr[I + 1] > mptr[I];
                for (int q = mptr[I]; fits && q < mptr[I + 1]; ++q) mem4[(size_t)4 * I + (q - mptr[I])] = midx[q];
            }
            if (fits) cur->mem4 = std::move(mem4); else cur->mem4.clear();
        }
        cur->agg = std::move(agg); cur->mptr = std::move(mptr); cur->midx = std::move(midx);
        cur->gptr = std::move(gptr); cur->gidx = std::move(gidx);
        out.lv.emplace_back();
        std::vector<int> cposv;
        bool fitsW = true;
        const bool iluLevel = nlev < iluLevels;   // (level index nlev: the one being added)
        CPR_T(out.tImg, fitsW = ell_image(Ac, out.lv.back(), cposv, Ac.n <= lprRows && !iluLevel));
        if (!fitsW) { out.error = "cpr: a row of a pressure-AMG level outgrew the level image"; return; }
        if (iluLevel) {   // ILU0 smoothing: greedy multi-colouring of the level's graph, colour by colour
            std::vector<int> colour, posv;
            const int ncolours = cpr_greedy_colours(out.lv.back(), colour, posv);
            CPR_T(out.tImg, cpr_ilu_schedule(out.lv.back(), posv, colour, ncolours, out.lv.back().ilu));
            if (!out.lv.back().ilu.error.empty()) { out.error = out.lv.back().ilu.error; return; }
        }
        // (out.lv may have reallocated: cur is looked up again)
        CprHostLevel* fine = out.lv.size() == 1 ? &out.l0 : &out.lv[out.lv.size() - 2];
        fine->cpos = cposv;                                   // where the coarse entries go
        cur = &out.lv.back();
        pos = std::move(cposv);
        A = std::move(Ac);
        ++nlev;
    }
    out.lastA = std::move(A);
    out.lastPos = std::move(pos);
#undef CPR_T
}
// ---- everything below level 0's image, on the host: two passes of pairwise matching per level, Galerkin lists, level images.
//      Pure host work on its arguments (no context, no HIP call): may run on a thread of its own.  ell0: level 0's value image
//      (W0 x Nb, as on the device) of the pressure matrix the structure is built from.
static void cpr_build_coarse_host(const Pattern& P, const std::vector<double>& ell0, double beta, int lprRows, int iluLevels, int stopRows, CprHostCoarse& out) {
    HCsr A;
    A.n = P.Nb; A.rowptr = P.rowptr; A.col = P.col;
    CprHostLevel img0;
    std::vector<int> pos;   // ELL position of every CSR entry of the level being coarsened
    (void)ell_image(A, img0, pos, false, P.Nb);
    if (P.Nghost > 0) {   // the host copy the hierarchy is built from: owned columns only (pos follows the entries that stay)
        HCsr F;
        std::vector<int> fpos;
        F.n = P.Nb; F.rowptr.assign(P.Nb + 1, 0);
        for (int i = 0; i < P.Nb; ++i) {
            for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k)
                if (A.col[k] < P.Nb) { F.col.push_back(A.col[k]); fpos.push_back(pos[k]); }
            F.rowptr[i + 1] = (int)F.col.size();
        }
        A = std::move(F);
        pos = std::move(fpos);
    }
    const int nnz0 = (int)A.col.size();
    A.val.resize(nnz0);
    for (int k = 0; k < nnz0; ++k) A.val[k] = ell0[pos[k]];
    cpr_coarsen_host(std::move(A), std::move(pos), P.fromOrder.data(), P.toOrder.data(), beta, lprRows, iluLevels, stopRows, out);
}
// ---- the device side of it: transfer arrays of every level, images of the levels below level 0 --------------------------------
// gathered: R's last level is continued across the ranks (cpr_gather_setup), not solved by R
static int cpr_upload_coarse(opmhip_ctx* c, CprDev& R, const CprHostCoarse& H, bool gathered) {
    int rc;
    if (!H.error.empty()) return fail(c, OPMHIP_ANALYSIS_FAILED, "%s", H.error.c_str());
    auto transfer = [&](const CprHostLevel& h, CprLevelDev& L) -> int {
        if (h.agg.empty()) return OPMHIP_SUCCESS;   // the coarsest level
        L.nc = h.nc;
        if ((rc = dev_upload(c, &L.d_agg, h.agg))) return rc;
        if ((rc = dev_upload(c, &L.d_mptr, h.mptr))) return rc;
        if ((rc = dev_upload(c, &L.d_midx, h.midx))) return rc;
        if (!h.mem4.empty() && (rc = dev_upload(c, &L.d_mem4, h.mem4))) return rc;
        if ((rc = dev_upload(c, &L.d_gptr, h.gptr))) return rc;
        if ((rc = dev_upload(c, &L.d_gidx, h.gidx))) return rc;
        if ((rc = dev_upload(c, &L.d_cpos, h.cpos))) return rc;
        return OPMHIP_SUCCESS;
    };
    R.lv.resize(1);
    if ((rc = transfer(H.l0, R.lv[0]))) return rc;
    for (const CprHostLevel& h : H.lv) {
        R.lv.emplace_back();
        if ((rc = upload_ell(c, h, R.lv.back()))) return rc;
        if (h.ilu.ncol > 0 && (rc = cpr_upload_ilu(c, h.ilu, R.lv.back()))) return rc;
        if ((rc = transfer(h, R.lv.back()))) return rc;
    }
    R.coarse_direct = !gathered && R.lv.back().n <= CPR_COARSE_DIRECT;
    if (R.coarse_direct) {   // the direct solve keeps the factors in LDS: n x (n | 1) doubles of dynamic shared memory, more than the 64 KB a kernel gets unasked
        static const bool ldsOk = hipFuncSetAttribute(reinterpret_cast<const void*>(k_cpr_dense_solve), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                      CPR_COARSE_DIRECT * (CPR_COARSE_DIRECT | 1) * (int)sizeof(double)) == hipSuccess;
        if (!ldsOk && (size_t)R.lv.back().n * (R.lv.back().n | 1) * sizeof(double) > 65536)
            return fail(c, OPMHIP_DEVICE_ERROR, "cpr: the device refuses %d KB of dynamic shared memory for the coarsest level's direct solve", CPR_COARSE_DIRECT * (CPR_COARSE_DIRECT | 1) / 128);
    }
    if (R.coarse_direct && (rc = dev_alloc(c, &R.d_lu, (size_t)R.lv.back().n * R.lv.back().n + 1))) return rc;   // + 1: the pivot flag
    R.structured = true;
    return OPMHIP_SUCCESS;
}
// gives back what cpr_upload_coarse allocated: level 0 keeps its image (keep0; the joined hierarchy of a decomposed run goes as a whole),
// the levels below it go
static void cpr_release_coarse(opmhip_ctx* c, CprDev& R, bool keep0 = true) {
    (void)hipStreamSynchronize(c->stream);
    if (R.gather.glob) cpr_release_coarse(c, *R.gather.glob, false);
    {
        CprGatherDev& G = R.gather;
        dev_free(c, &G.d_recv); dev_free(c, &G.d_vsend); dev_free(c, &G.d_vrecv); dev_free(c, &G.d_unpad); dev_free(c, &G.d_vunpad); dev_free(c, &G.d_vpos);
        dev_free(c, &G.d_cagg); dev_free(c, &G.d_send); dev_free(c, &G.d_src); dev_free(c, &G.d_xptr); dev_free(c, &G.d_xidx); dev_free(c, &G.d_qrow); dev_free(c, &G.d_qentry); dev_free(c, &G.d_apg);
        G = CprGatherDev();
    }
    for (size_t l = 0; l < R.lv.size(); ++l) {
        CprLevelDev& L = R.lv[l];
        dev_free(c, &L.d_agg); dev_free(c, &L.d_mptr); dev_free(c, &L.d_midx); dev_free(c, &L.d_mem4); dev_free(c, &L.d_gptr); dev_free(c, &L.d_gidx); dev_free(c, &L.d_cpos);
        L.nc = 0;
        if (l == 0 && keep0) continue;
        dev_free(c, &L.d_ecol); dev_free(c, &L.d_rlen); dev_free(c, &L.d_diag);
        dev_free(c, &L.d_val); dev_free(c, &L.d_dinv); dev_free(c, &L.d_x2);
        dev_free(c, &L.d_b); dev_free(c, &L.d_x); dev_free(c, &L.d_r); dev_free(c, &L.d_sword); dev_free(c, &L.d_stable);
        cpr_free_ilu(c, L);
    }
    if (!R.lv.empty()) R.lv.resize(keep0 ? 1 : 0);
    dev_free(c, &R.d_lu);
    R.structured = false;
}
// level 0's values as they are on the device now (k_cpr_pvals ran): the matrix a structure is built from
static int cpr_download_level0(opmhip_ctx* c, std::vector<double>& ell) {
    const CprLevelDev& L0 = c->cpr.lv[0];
    ell.resize((size_t)L0.W * c->pat.Nb);
    OPMHIP_HIP(c, hipMemcpyAsync(ell.data(), L0.d_val, ell.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
    return OPMHIP_SUCCESS;
}
static int cpr_update_values(opmhip_ctx* c, CprDev& R);
static int cpr_upload_coarse(opmhip_ctx* c, CprDev& R, const CprHostCoarse& H, bool gathered);

// ---- decomposed runs: the hierarchy continued across the ranks ------------------------------------------------------------------
// The reference's parallel CPR keeps a coarse pressure problem that spans the processes (Dune's parallel AMG behind
// linalg/OwningTwoLevelPreconditioner.hpp; aggregates never cross a process boundary, the coarse operators carry the couplings between
// the processes).  Here: every rank coarsens its own subdomain down to its first level of at most opmhip_config.cpr_gather_rows rows
// (couplings to ghost cells left out of those levels: they smooth, and smoothing is local); that level's matrix rows - its own entries
// plus the Galerkin sums of the fine couplings between the aggregates of different subdomains - are gathered into ONE system that every
// rank holds, coarsens further and cycles on redundantly: one all-gather of a right-hand side per application, one of matrix values per
// solve.  oracle: orc_cpr_solve_blocks with gather_rows.
// cpr_gather_rows = 0 ("the default") means ON over the loopback communicator and OFF over RCCL: the spanning stage's collectives
// (ncclAllGather on the context stream, the w = 1 / w = 2 halo exchanges inside launch_cpr_apply, the joined level's set-up exchanges) have
// only ever run as threads on one GPU - no record of any round takes them with nranks > 1 over RCCL - and a default must not be a path
// that was never executed.  A positive cpr_gather_rows asks for it explicitly on either communicator.
static bool cpr_gathering(const opmhip_ctx* c) {
    if (c->comm.nranks <= 1 || c->comm.kind == COMM_NONE) return false;
    const int g = c->cfg.cpr_gather_rows;
    return g > 0 || (g == 0 && c->comm.kind != COMM_RCCL);
}
static int cpr_stop_rows(const opmhip_ctx* c) { return cpr_gathering(c) ? (c->cfg.cpr_gather_rows > 0 ? c->cfg.cpr_gather_rows : 100000) : CPR_COARSE_DIRECT; }
// `bytes` bytes of every rank, in rank order, on every rank (set-up only: staged through device buffers of its own)
static int cpr_host_allgather(opmhip_ctx* c, const void* src, size_t bytes, std::vector<char>& dst) {
    const int nr = c->comm.nranks;
    const size_t cnt = std::max<size_t>((bytes + 7) / 8, 1);
    std::vector<double> sb(cnt, 0.0), rb(cnt * nr, 0.0);
    if (bytes) std::memcpy(sb.data(), src, bytes);
    double *d_s = nullptr, *d_r = nullptr;
    if (hipMalloc((void**)&d_s, cnt * sizeof(double)) != hipSuccess || hipMalloc((void**)&d_r, cnt * nr * sizeof(double)) != hipSuccess) {
        if (d_s) (void)hipFree(d_s);
        // without its buffers this rank cannot enter the exchange and its peers wait in it - the one failure of the set-up that is not
        // agreed on first (cpr_gather_setup); the buffers are a few integers per row of the joined level
        return fail(c, OPMHIP_DEVICE_ERROR, "cpr: no device memory for the set-up exchange of the joined level");
    }
    int rc = [&]() -> int {
        OPMHIP_HIP(c, hipMemcpyAsync(d_s, sb.data(), cnt * sizeof(double), hipMemcpyHostToDevice, c->stream));
        int r2 = comm_allgather(c, d_s, d_r, cnt);
        if (r2) return r2;
        OPMHIP_HIP(c, hipMemcpyAsync(rb.data(), d_r, cnt * nr * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        return OPMHIP_SUCCESS;
    }();
    (void)hipFree(d_s); (void)hipFree(d_r);
    if (rc) return rc;
    dst.resize(bytes * nr);
    for (int r = 0; r < nr; ++r)
        if (bytes) std::memcpy(dst.data() + (size_t)r * bytes, rb.data() + (size_t)r * cnt, bytes);
    return OPMHIP_SUCCESS;
}
// after the rank's own hierarchy is on the device (its last level = the one that is gathered; H: the host side of it, with that level's matrix)
static int cpr_gather_setup(opmhip_ctx* c, const CprHostCoarse& H) {
    const Pattern& P = c->pat;
    CprDev& R = c->cpr;
    CprGatherDev& G = R.gather;
    const int nr = c->comm.nranks, me = c->comm.rank;
    // A failure of ONE rank between two exchanges must not leave the others waiting in the next one: the rank notes its first error
    // (text in c->err), goes on through every exchange with what it has, and the error flag travels with the next payload - all ranks
    // then leave together, the failed one with its own status, the others with a status that names it.
    int lerr = 0, rc;
    auto note = [&](int e) { if (e && !lerr) lerr = e; };
    // a HIP call between two exchanges goes through note() as well (OPMHIP_HIP would return at once, alone)
    auto hipnote = [&](hipError_t e, const char* what) { if (e != hipSuccess) note(fail(c, OPMHIP_DEVICE_ERROR, "%s failed: %s (cpr_gather_setup)", what, hipGetErrorString(e))); };
    auto first_failed = [&](const std::vector<char>& b, size_t stride, size_t at, const char* where) -> int {   // the flag at byte `at` of every rank's record
        for (int r = 0; r < nr; ++r) {
            int f;
            std::memcpy(&f, b.data() + (size_t)r * stride + at, sizeof f);
            if (f) return lerr ? lerr : fail(c, f, "cpr: rank %d failed in the set-up of the joined level (%s)", r, where);
        }
        return OPMHIP_SUCCESS;
    };
    if (P.Nghost > 0 && !c->comm.halo_set) note(fail(c, OPMHIP_NOT_READY, "cpr: the joined coarse level needs the halo lists (opmhip_set_halo) before the first solve"));
    std::vector<char> buf;
    const size_t g = R.lv.size() - 1;
    const int nloc = R.lv[g].n;
    // 1. sizes of everybody's slice, of everybody's subdomain
    const int mine[3] = {nloc, P.Nb, lerr};
    if ((rc = cpr_host_allgather(c, mine, sizeof mine, buf))) return rc;
    if ((rc = first_failed(buf, sizeof mine, 2 * sizeof(int), "halo lists"))) return rc;
    std::vector<int> offs(nr + 1, 0), foffs(nr + 1, 0);
    int maxn = 1;
    for (int r = 0; r < nr; ++r) {
        int v[3];
        std::memcpy(v, buf.data() + (size_t)r * sizeof v, sizeof v);
        offs[r + 1] = offs[r] + v[0];
        foffs[r + 1] = foffs[r] + v[1];
        maxn = std::max(maxn, v[0]);
    }
    // 2. aggregate (on the gathered level) of every owned cell
    std::vector<int> cagg(P.Nb);
    std::iota(cagg.begin(), cagg.end(), 0);
    for (size_t l = 0; l < g; ++l) {
        const std::vector<int>& a = l == 0 ? H.l0.agg : H.lv[l - 1].agg;
        for (int i = 0; i < P.Nb; ++i) cagg[i] = a[cagg[i]];
    }
    // 3. the same for the ghost cells, from their owners: row of the joined level, and the owner-side place of the cell (what orders the
    //    fine couplings inside one joined entry: subdomain by subdomain, each in its rank's internal order)
    std::vector<double> hv((size_t)2 * P.Nloc, 0.0);
    for (int i = 0; i < P.Nb; ++i) { hv[(size_t)2 * i] = (double)(offs[me] + cagg[i]); hv[(size_t)2 * i + 1] = (double)(foffs[me] + i); }
    {
        double* d_hv = c->d_stageV;   // 3 doubles per local cell, idle between the uploads and the solve: nothing to allocate
        hipnote(hipMemcpyAsync(d_hv, hv.data(), hv.size() * sizeof(double), hipMemcpyHostToDevice, c->stream), "hipMemcpyAsync");
        if ((rc = comm_halo_f64(c, d_hv, 2))) return rc;   // (a failed exchange is every rank's: the communicator reports it to all)
        hipnote(hipMemcpyAsync(hv.data(), d_hv, hv.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream), "hipMemcpyAsync");
        hipnote(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
    }
    // 4. my rows of the joined level: my last level's entries, and one entry per pair (my aggregate, an aggregate of another rank) that a
    //    fine coupling joins
    std::vector<int> qrow, qentry;
    struct Fine { int i; long long gf; int q; };
    std::vector<std::map<int, std::vector<Fine>>> cross(nloc);
    for (int i = 0; i < P.Nb; ++i)
        for (int k = P.rowptr[i]; k < P.rowptr[i + 1]; ++k) {
            const int gh = P.col[k];
            if (gh < P.Nb) continue;
            const int q = (int)qrow.size();
            qrow.push_back(i); qentry.push_back(k);
            cross[cagg[i]][(int)hv[(size_t)2 * gh]].push_back(Fine{i, (long long)hv[(size_t)2 * gh + 1], q});
        }
    G.nq = (int)qrow.size();
    std::vector<double> apg(std::max(G.nq, 1), 0.0);
    note(dev_upload(c, &G.d_qrow, qrow));
    note(dev_upload(c, &G.d_qentry, qentry));
    note(dev_alloc(c, &G.d_apg, (size_t)std::max(G.nq, 1)));
    if (G.nq > 0 && !lerr) {
        hipLaunchKernelGGL(k_cpr_ghost_pvals, g256(G.nq), dim3(256), 0, c->stream, G.nq, G.d_qrow, G.d_qentry, c->d_A, R.d_w, G.d_apg);
        hipnote(hipMemcpyAsync(apg.data(), G.d_apg, (size_t)G.nq * sizeof(double), hipMemcpyDeviceToHost, c->stream), "hipMemcpyAsync");
        hipnote(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
    }
    const HCsr& LA = H.lastA;
    std::vector<int> rowlen(nloc, 0), cols, src, xptr(1, 0), xidx;
    std::vector<double> vals;
    for (int I = 0; I < nloc; ++I) {
        auto cr = cross[I].begin();
        auto emit_cross = [&](int upto) {   // the entries towards other ranks whose joined column lies below `upto`
            for (; cr != cross[I].end() && cr->first < upto; ++cr) {
                std::vector<Fine>& f = cr->second;
                std::sort(f.begin(), f.end(), [](const Fine& a, const Fine& b) { return a.i != b.i ? a.i < b.i : a.gf < b.gf; });
                double sum = 0.0;
                for (const Fine& e : f) { sum += apg[e.q]; xidx.push_back(e.q); }
                cols.push_back(cr->first); vals.push_back(sum);
                src.push_back(-1 - (int)(xptr.size() - 1));
                xptr.push_back((int)xidx.size());
                ++rowlen[I];
            }
        };
        for (int k = LA.rowptr[I]; k < LA.rowptr[I + 1]; ++k) {
            emit_cross(offs[me] + LA.col[k]);
            cols.push_back(offs[me] + LA.col[k]); vals.push_back(LA.val[k]); src.push_back(H.lastPos[k]);
            ++rowlen[I];
        }
        emit_cross(INT_MAX);
    }
    G.nnzloc = (int)cols.size();
    // 5. everybody's rows
    const int mine5[2] = {G.nnzloc, lerr};
    if ((rc = cpr_host_allgather(c, mine5, sizeof mine5, buf))) return rc;
    if ((rc = first_failed(buf, sizeof mine5, sizeof(int), "boundary couplings"))) return rc;
    std::vector<int> nnzs(nr);
    int maxnnz = 1;
    for (int r = 0; r < nr; ++r) { std::memcpy(&nnzs[r], buf.data() + (size_t)r * sizeof mine5, sizeof(int)); maxnnz = std::max(maxnnz, nnzs[r]); }
    std::vector<char> bl, bc, bv;
    {
        std::vector<int> t(maxn, 0);
        std::copy(rowlen.begin(), rowlen.end(), t.begin());
        if ((rc = cpr_host_allgather(c, t.data(), (size_t)maxn * sizeof(int), bl))) return rc;
        std::vector<int> tc(maxnnz, 0);
        std::copy(cols.begin(), cols.end(), tc.begin());
        if ((rc = cpr_host_allgather(c, tc.data(), (size_t)maxnnz * sizeof(int), bc))) return rc;
        std::vector<double> tv(maxnnz, 0.0);
        std::copy(vals.begin(), vals.end(), tv.begin());
        if ((rc = cpr_host_allgather(c, tv.data(), (size_t)maxnnz * sizeof(double), bv))) return rc;
    }
    HCsr J;
    const int NG = offs[nr];
    J.n = NG;
    J.rowptr.assign(1, 0);
    std::vector<int> unpad, vunpad;
    for (int r = 0; r < nr; ++r) {
        const int* rl = reinterpret_cast<const int*>(bl.data() + (size_t)r * maxn * sizeof(int));
        const int* rcs = reinterpret_cast<const int*>(bc.data() + (size_t)r * maxnnz * sizeof(int));
        const double* rv = reinterpret_cast<const double*>(bv.data() + (size_t)r * maxnnz * sizeof(double));
        int e = 0;
        for (int I = 0; I < offs[r + 1] - offs[r]; ++I) {
            for (int t = 0; t < rl[I]; ++t, ++e) {
                if (rcs[e] < 0 || rcs[e] >= NG || e >= nnzs[r]) return fail(c, OPMHIP_ANALYSIS_FAILED, "cpr: the joined level's rows of rank %d are inconsistent", r);   // the same data on every rank: all leave here
                J.col.push_back(rcs[e]); J.val.push_back(rv[e]); vunpad.push_back(r * maxnnz + e);
            }
            J.rowptr.push_back((int)J.col.size());
            unpad.push_back(r * maxn + I);
        }
    }
    // 6. the hierarchy all ranks share: level 0 = the joined level (the same matrix on every rank, hence the same hierarchy)
    G.glob = std::make_shared<CprDev>();
    CprDev& JD = *G.glob;
    JD.omega = R.omega; JD.damp = R.damp; JD.beta = R.beta;
    CprHostLevel img0;
    std::vector<int> posg;
    if (!ell_image(J, img0, posg, NG <= cpr_lpr_rows())) return fail(c, OPMHIP_ANALYSIS_FAILED, "cpr: a row of the joined level has %d entries (limit %d): lower opmhip_config.cpr_gather_rows' level count or raise the limit", img0.W, CPR_MAX_W);
    JD.lv.emplace_back();
    // (from here on only device memory can fail, and on one rank alone: noted, agreed on at the end)
    note(upload_ell(c, img0, JD.lv[0]));
    CprHostCoarse HJ;
    cpr_coarsen_host(J, posg, nullptr, nullptr, R.beta, cpr_lpr_rows(), 0, CPR_COARSE_DIRECT, HJ);
    if (!lerr) note(cpr_upload_coarse(c, JD, HJ, false));
    // 7. what the solves need on the device
    G.nloc = nloc; G.off = offs[me]; G.NG = NG; G.maxn = maxn; G.maxnnz = maxnnz; G.nnzG = (int)J.col.size();
    if (!lerr) note(dev_alloc(c, &G.d_recv, (size_t)nr * maxn));
    if (!lerr) note(dev_alloc(c, &G.d_vsend, (size_t)maxnnz));
    if (!lerr) hipnote(hipMemsetAsync(G.d_vsend, 0, (size_t)maxnnz * sizeof(double), c->stream), "hipMemsetAsync");
    if (!lerr) note(dev_alloc(c, &G.d_vrecv, (size_t)nr * maxnnz));
    if (!lerr) note(dev_upload(c, &G.d_unpad, unpad));
    if (!lerr) note(dev_upload(c, &G.d_vunpad, vunpad));
    if (!lerr) note(dev_upload(c, &G.d_vpos, posg));
    if (!lerr) note(dev_upload(c, &G.d_src, src));
    if (!lerr) note(dev_upload(c, &G.d_xptr, xptr));
    if (!lerr) note(dev_upload(c, &G.d_xidx, xidx));
    if (!lerr) note(dev_upload(c, &G.d_cagg, cagg));
    if (!lerr) note(dev_alloc(c, &G.d_send, (size_t)maxn));   // my slice of the joined level's right-hand side: room for the largest slice
    if (!lerr) hipnote(hipMemsetAsync(G.d_send, 0, (size_t)maxn * sizeof(double), c->stream), "hipMemsetAsync");
    if ((rc = cpr_host_allgather(c, &lerr, sizeof lerr, buf))) return rc;
    if ((rc = first_failed(buf, sizeof lerr, 0, "uploads of the joined level"))) return rc;
    G.on = true;
    return OPMHIP_SUCCESS;
}
// per solve: my rows' values of the joined level's matrix, everybody's, the values of the shared hierarchy below it
static int cpr_gather_values(opmhip_ctx* c) {
    CprDev& R = c->cpr;
    CprGatherDev& G = R.gather;
    int rc;
    if (G.nq > 0) hipLaunchKernelGGL(k_cpr_ghost_pvals, g256(G.nq), dim3(256), 0, c->stream, G.nq, G.d_qrow, G.d_qentry, c->d_A, R.d_w, G.d_apg);
    if (G.nnzloc > 0) hipLaunchKernelGGL(k_cpr_gather_vals, g256(G.nnzloc), dim3(256), 0, c->stream, G.nnzloc, G.d_src, G.d_xptr, G.d_xidx, G.d_apg, R.lv.back().d_val, G.d_vsend);
    if ((rc = comm_allgather(c, G.d_vsend, G.d_vrecv, (size_t)G.maxnnz))) return rc;
    hipLaunchKernelGGL(k_cpr_scatter_vals, g256(G.nnzG), dim3(256), 0, c->stream, G.nnzG, G.d_vunpad, G.d_vpos, G.d_vrecv, G.glob->lv[0].d_val);
    return cpr_update_values(c, *G.glob);
}
// the structure from the pressure matrix now in level 0's image, synchronously; a set-up that fails half way gives back
// everything it allocated: a retry starts from a clean slate, nothing piles up
static int cpr_setup_coarse_now(opmhip_ctx* c) {
    CprDev& R = c->cpr;
    static const bool timing = tuning_env("OPMHIP_CPR_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    std::vector<double> ell;
    int rc;
    if ((rc = cpr_download_level0(c, ell))) return rc;
    CprHostCoarse H;
    cpr_build_coarse_host(c->pat, ell, R.beta, cpr_lpr_rows(), cpr_gathering(c) ? 0 : cpr_ilu_levels(c), cpr_stop_rows(c), H);
    const double t1 = now();
    const size_t mark = c->allocs.size();
    const bool gathered = cpr_gathering(c);
    rc = cpr_upload_coarse(c, R, H, gathered);
    if (gathered) {   // a rank whose own hierarchy failed must not leave its peers waiting in the joined level's set-up: agree first
        std::vector<char> flags;
        const int mine = rc;
        int r2 = cpr_host_allgather(c, &mine, sizeof mine, flags);
        for (int r = 0; !r2 && !rc && r < c->comm.nranks; ++r) {
            int f;
            std::memcpy(&f, flags.data() + (size_t)r * sizeof f, sizeof f);
            if (f) rc = fail(c, f, "cpr: rank %d failed to build its pressure hierarchy", r);
        }
        if (!rc) rc = r2;
    }
    if (!rc && gathered) rc = cpr_gather_setup(c, H);
    if (rc) {
        (void)hipStreamSynchronize(c->stream);
        R.gather = CprGatherDev();
        while (c->allocs.size() > mark) { (void)hipFree(c->allocs.back()); c->allocs.pop_back(); }
        for (size_t l = 0; l < R.lv.size(); ++l) {
            CprLevelDev& L = R.lv[l];
            L.d_agg = L.d_mptr = L.d_midx = L.d_mem4 = L.d_gptr = L.d_gidx = L.d_cpos = nullptr;
        }
        if (!R.lv.empty()) R.lv.resize(1);
        R.d_lu = nullptr;
        R.structured = false;
        return rc;
    }
    if (timing) std::fprintf(stderr, "opmhip cpr set-up: %.3f s (matching %.3f, Galerkin %.3f, level images %.3f, uploads %.3f), %zu levels\n", now() - t0, H.tAgg, H.tGal, H.tImg, now() - t1, R.lv.size());
    return OPMHIP_SUCCESS;
}
static int cpr_gather_values(opmhip_ctx* c);
// the values of a hierarchy below its level 0 (whose image holds the matrix already): 1 / diagonal, the ILU0 factors of the levels that
// smooth with them, Galerkin values level by level, the dense LU of the coarsest level
static bool cpr_ilu_active(const CprDev& R, size_t l);
static int cpr_update_values(opmhip_ctx* c, CprDev& R) {
    for (size_t l = 0; l < R.lv.size(); ++l) {
        CprLevelDev& L = R.lv[l];
        hipLaunchKernelGGL(k_cpr_dinv, g256(L.n), dim3(256), 0, c->stream, L.n, L.d_diag, L.d_val, L.d_dinv);
        if (cpr_ilu_active(R, l) && L.iluSimple) {   // scalar ILU0 of a level without triangles: the sweeps' images straight from the matrix, then the recurrence
            hipLaunchKernelGGL(k_cpr_ilu_pack, g256(L.n), dim3(256), 0, c->stream, L.n, L.W, L.iluMW, 0, L.iluWL, L.iluWU, L.d_ecol, L.d_rlen, L.d_diag, L.d_imask, L.d_val,
                               L.d_ilv, L.d_ilc, L.d_iuv, L.d_iuc, L.d_iud);
            for (size_t cc = 0; cc < L.iluNseq.size(); ++cc) {
                const int nseq = L.iluNseq[cc], wq = L.iluWl[cc];
                if (nseq == 0) continue;
                const dim3 grid((nseq + 63) / 64);
#define CPR_FS(WQ, G) hipLaunchKernelGGL((k_cpr_ilu_factor_simple<WQ, G>), grid, dim3(64), 0, c->stream, nseq, L.iluNsteps[cc], L.d_rowAt + L.iluOff[cc], L.n, wq, L.d_val, L.d_diag, L.d_ilc, L.d_tpos, L.d_ilv, (const double*)L.d_iud, L.d_iud)
                if (wq <= 1) CPR_FS(1, 12); else if (wq <= 2) CPR_FS(2, 12); else if (wq <= 4) CPR_FS(4, 10); else if (wq <= 6) CPR_FS(6, 8); else CPR_FS(8, 6);
#undef CPR_FS
            }
        } else
        if (cpr_ilu_active(R, l)) {   // scalar ILU0 of the level, colour by colour
            OPMHIP_HIP(c, hipMemcpyAsync(L.d_fval, L.d_val, (size_t)L.W * L.n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
            for (size_t cc = 0; cc < L.iluNseq.size(); ++cc) {
                if (L.iluNseq[cc] == 0) continue;
                if (L.rm) hipLaunchKernelGGL(k_cpr_ilu_factor<true>, g256(L.iluNseq[cc]), dim3(256), 0, c->stream, L.iluNseq[cc], L.iluNsteps[cc], L.d_rowAt + L.iluOff[cc], L.n, L.W, L.iluMW, L.iluWL, L.d_ecol, L.d_rlen, L.d_diag, L.d_imask, L.d_lorder, L.d_fval);
                else hipLaunchKernelGGL(k_cpr_ilu_factor<false>, g256(L.iluNseq[cc]), dim3(256), 0, c->stream, L.iluNseq[cc], L.iluNsteps[cc], L.d_rowAt + L.iluOff[cc], L.n, L.W, L.iluMW, L.iluWL, L.d_ecol, L.d_rlen, L.d_diag, L.d_imask, L.d_lorder, L.d_fval);
            }
            hipLaunchKernelGGL(k_cpr_ilu_pack, g256(L.n), dim3(256), 0, c->stream, L.n, L.W, L.iluMW, L.rm ? 1 : 0, L.iluWL, L.iluWU, L.d_ecol, L.d_rlen, L.d_diag, L.d_imask, L.d_fval,
                               L.d_ilv, L.d_ilc, L.d_iuv, L.d_iuc, L.d_iud);
        }
        if (l + 1 < R.lv.size()) {
            CprLevelDev& C = R.lv[l + 1];
            hipLaunchKernelGGL(k_cpr_galerkin, g256(C.nnz), dim3(256), 0, c->stream, C.nnz, L.d_gptr, L.d_gidx, L.d_cpos, L.d_val, C.d_val);
        }
    }
    if (R.coarse_direct) {
        const CprLevelDev& C = R.lv.back();
        // in LDS where the device grants n x n doubles of dynamic shared memory (gfx950: yes, up to n = 128), else in place in global memory
        static const bool ldsOk = hipFuncSetAttribute(reinterpret_cast<const void*>(k_cpr_dense_lu_lds), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                      CPR_COARSE_DIRECT * CPR_COARSE_DIRECT * (int)sizeof(double)) == hipSuccess;
        if (ldsOk) hipLaunchKernelGGL(k_cpr_dense_lu_lds, dim3(1), dim3(CPR_LU_THREADS), (size_t)C.n * C.n * sizeof(double), c->stream, C.n, C.W, C.rm ? 1 : 0, C.d_ecol, C.d_rlen, C.d_val, R.d_lu);
        else hipLaunchKernelGGL(k_cpr_dense_lu, dim3(1), dim3(256), 0, c->stream, C.n, C.W, C.rm ? 1 : 0, C.d_ecol, C.d_rlen, C.d_val, R.d_lu);
    }
    return OPMHIP_SUCCESS;
}
void cpr_shutdown(opmhip_ctx* c) { c->cpr.job.reset(); }
int cpr_update(opmhip_ctx* c, bool solveBoundary) {
    const Pattern& P = c->pat;
    CprDev& R = c->cpr;
    int rc;
    bool startAsync = false;
    if (R.structured && (solveBoundary || R.recreate)) {   // --cpr-reuse-setup (ISTLSolverEbos.hpp:401-426 shouldCreateSolver): build the structure anew from this matrix?
        const int mode = c->cfg.cpr_reuse_setup;
        bool anew = R.recreate || mode == 0 || (mode == 1 && c->asmb.assembled && c->asmb.last_iteration == 0) || (mode == 2 && c->last_solve_iterations > 10);
        if (mode == 2 && c->cfg.cpr_async_setup && !R.recreate && !cpr_gathering(c)) {
            // the rebuild beside the solves: a build that has finished is swapped in at this solve boundary; a new one is started
            // when the rule asks for it and none is under way; this solve goes on with the structure it has
            if (R.job && R.job->ready.load(std::memory_order_acquire)) {
                R.job->th.join();
                cpr_release_coarse(c, R);
                const size_t mark = c->allocs.size();
                rc = cpr_upload_coarse(c, R, R.job->result, false);
                R.job.reset();
                if (rc) {   // (cannot happen for a pattern that was set up once; if it does: back to a synchronous set-up below)
                    (void)hipStreamSynchronize(c->stream);
                    while (c->allocs.size() > mark) { (void)hipFree(c->allocs.back()); c->allocs.pop_back(); }
                    for (CprLevelDev& L : R.lv) L.d_agg = L.d_mptr = L.d_midx = L.d_mem4 = L.d_gptr = L.d_gidx = L.d_cpos = nullptr;
                    R.lv.resize(1); R.d_lu = nullptr; R.structured = false;
                }
            } else if (anew && !R.job) startAsync = true;
            anew = false;
        }
        if (anew) { R.job.reset(); cpr_release_coarse(c, R); }
    }
    R.recreate = false;
    if (!R.level0 && (rc = cpr_setup_level0(c))) return rc;
    const int ps = prof_begin(c, PROF_ILU_FACTOR);
    if (R.pvals_fresh) R.pvals_fresh = false;   // this solve's factorisation formed the weights and level 0's values as it staged the rows (FactorRider)
    else {
        if ((rc = cpr_weights(c))) { prof_end(c, ps); return rc; }
        hipLaunchKernelGGL(k_cpr_pvals, g256(P.Nb), dim3(256), 0, c->stream, P.Nb, R.lv[0].W, P.d_rowptr, (P.Nghost > 0 && !cpr_gathering(c)) ? P.d_col : (const int*)nullptr, c->d_A, R.d_w, R.lv[0].d_val, R.d_pcol);
    }
    if (!R.structured) {   // first solve, or a synchronous rebuild: from the pressure matrix just formed
        if ((rc = cpr_setup_coarse_now(c))) { prof_end(c, ps); return rc; }
    } else if (startAsync) {
        R.job = std::make_shared<CprAsyncJob>();
        std::shared_ptr<CprAsyncJob> job = R.job;
        auto ell = std::make_shared<std::vector<double>>();
        if ((rc = cpr_download_level0(c, *ell))) { R.job.reset(); prof_end(c, ps); return rc; }
        const Pattern* pat = &c->pat;   // outlives the job: cpr_release_structure / cpr_shutdown join it before the pattern goes
        const double beta = R.beta;
        const int lpr = cpr_lpr_rows(), iluLevels = cpr_ilu_levels(c);
        CprAsyncJob* raw = job.get();
        job->th = std::thread([raw, ell, pat, beta, lpr, iluLevels]() {
            cpr_build_coarse_host(*pat, *ell, beta, lpr, iluLevels, CPR_COARSE_DIRECT, raw->result);
            raw->ready.store(1, std::memory_order_release);
        });
    }
    if ((rc = cpr_update_values(c, R))) { prof_end(c, ps); return rc; }
    if (R.gather.on && (rc = cpr_gather_values(c))) { prof_end(c, ps); return rc; }
    prof_end(c, ps);
    OPMHIP_HIP(c, hipGetLastError());
    return OPMHIP_SUCCESS;
}

// is level l smoothed with its ILU0 (a level that has the schedule and is not the coarsest)?
static bool cpr_ilu_active(const CprDev& R, size_t l) { return l + 1 < R.lv.size() && R.lv[l].ilu; }
// v = (U^-1 L^-1) d with the level's ILU0: forward colour by colour, backward in reverse; out != NULL: out = add + v as the
// backward sweeps store (and, vfine, the block vector (0, out, 0))
template <bool BWD, int WQ>
static void cpr_ilu_launch(opmhip_ctx* c, const CprLevelDev& L, int cc, const double* d, double* v, const double* add, double* out, double* vfine) {
    const int nseq = L.iluNseq[cc], nsteps = L.iluNsteps[cc], wq = BWD ? L.iluWu[cc] : L.iluWl[cc];
    const int* rowAt = L.d_rowAt + L.iluOff[cc];
    const double* fv = BWD ? L.d_iuv : L.d_ilv;
    const int* fc = BWD ? L.d_iuc : L.d_ilc;
    if (L.iluFast[cc] && WQ > 0 && nsteps > 1) {
        constexpr int G = WQ <= 4 ? 12 : WQ <= 6 ? 10 : 6;   // (WQ = 6, G = 10: about 440 registers, no spills - and one wavefront per SIMD is all a colour offers)
        hipLaunchKernelGGL((k_cpr_ilu_sweep_fast<BWD, (WQ > 0 ? WQ : 1), G>), dim3((nseq + 63) / 64), dim3(64), 0, c->stream, nseq, nsteps, rowAt, L.n, wq, fv, fc, L.d_iud, d, (const double*)v, v, add, out, vfine, c->d_done);
    } else hipLaunchKernelGGL((k_cpr_ilu_sweep<BWD, WQ>), g256(nseq), dim3(256), 0, c->stream, nseq, nsteps, rowAt, L.n, wq, fv, fc, L.d_iud, d, v, add, out, vfine, c->d_done);
}
template <bool BWD>
static void cpr_ilu_colour(opmhip_ctx* c, const CprLevelDev& L, int cc, const double* d, double* v, const double* add, double* out, double* vfine) {
    if (L.iluNseq[cc] == 0) return;
    const int wq = BWD ? L.iluWu[cc] : L.iluWl[cc];
    if (wq <= 1) cpr_ilu_launch<BWD, 1>(c, L, cc, d, v, add, out, vfine);
    else if (wq <= 2) cpr_ilu_launch<BWD, 2>(c, L, cc, d, v, add, out, vfine);
    else if (wq <= 4) cpr_ilu_launch<BWD, 4>(c, L, cc, d, v, add, out, vfine);
    else if (wq <= 6) cpr_ilu_launch<BWD, 6>(c, L, cc, d, v, add, out, vfine);
    else if (wq <= 8) cpr_ilu_launch<BWD, 8>(c, L, cc, d, v, add, out, vfine);
    else cpr_ilu_launch<BWD, 0>(c, L, cc, d, v, add, out, vfine);
}
static void cpr_ilu_smooth(opmhip_ctx* c, const CprLevelDev& L, const double* d, double* v, const double* add, double* out, double* vfine) {
    const int nc = (int)L.iluNseq.size();
    for (int cc = 0; cc < nc; ++cc) cpr_ilu_colour<false>(c, L, cc, d, v, nullptr, nullptr, nullptr);
    for (int cc = nc - 1; cc >= 0; --cc) cpr_ilu_colour<true>(c, L, cc, d, v, add, out, vfine);
}
// does level l (> 0) form its right-hand side itself, from the finer level's residual (cpr_restricted)?  Then no restriction
// kernel runs between the two levels.  Lane-group levels do, and the dense solve of the coarsest level.
static bool cpr_forms_rhs(const CprDev& R, size_t l) {
    static const bool off = tuning_env("OPMHIP_CPR_UNFUSED") != nullptr;   // A/B switch: the restriction as a launch of its own
    if (off || l == 0 || l >= R.lv.size() || !R.lv[l - 1].d_mem4) return false;
    if (R.gather.on && l + 1 == R.lv.size()) return false;   // the level that is gathered: its right-hand side is what travels
    return R.lv[l].rm || (l + 1 == R.lv.size() && R.coarse_direct);
}
// does level l take its pre-smoothed iterate x = omega D^-1 b from the kernel that produces b (the restriction above it)?
static bool cpr_presmooth_rides(const CprDev& R, size_t l) {
    const CprLevelDev& L = R.lv[l];
    if (cpr_forms_rhs(R, l) || cpr_ilu_active(R, l)) return false;
    if (R.gather.on && l + 1 == R.lv.size()) return false;
    if (l + 1 == R.lv.size()) return !R.coarse_direct;   // Jacobi coarse "solve": starts with the same statement
    return !L.rm;                                         // lane-group levels form it on the fly inside k_cpr_down_lpr
}
// one V(1,1) cycle on level l from x = 0; returns the buffer that holds the level's result (fineOut != NULL on level 0: the
// result is written there as the block vector (0, x_p, 0) instead, and NULL comes back)
static const double* cpr_vcycle(opmhip_ctx* c, CprDev& R, size_t l, double* fineOut = nullptr) {
    CprLevelDev& L = R.lv[l];
    const double* done = c->d_done;
    const bool fused = cpr_forms_rhs(R, l);           // b = restriction of the finer level's residual, formed by this level's first kernel
    const bool havex = cpr_presmooth_rides(R, l);     // L.d_x = omega D^-1 b is there already
    const int4* mem4f = fused ? (const int4*)R.lv[l - 1].d_mem4 : nullptr;
    const double* rf = fused ? R.lv[l - 1].d_r : nullptr;
    if (l + 1 == R.lv.size()) {
        if (R.coarse_direct) {
            // (n rows of n | 1 doubles of dynamic shared memory: 129 KB at n = 128 - gfx950's 160 KB hold it; cpr_upload_coarse checked the attribute)
            hipLaunchKernelGGL(k_cpr_dense_solve, dim3(1), dim3(CPR_DENSE_THREADS), (size_t)L.n * (L.n | 1) * sizeof(double), c->stream, L.n, R.d_lu, L.d_b, L.d_x, (const int*)mem4f, rf, done);
            return L.d_x;
        }
        // could not coarsen further: Jacobi sweeps stand in for the coarse solve (oracle/cpr.hpp: 1 + 4)
        if (!havex && !fused) hipLaunchKernelGGL(k_cpr_presmooth, g256(L.n), dim3(256), 0, c->stream, L.n, R.omega, L.d_dinv, L.d_b, L.d_x, done);
        double *xin = L.d_x, *xout = L.d_x2;
        for (int sweep = 0; sweep < 4; ++sweep) {
            if (L.rm && fused && sweep == 0) hipLaunchKernelGGL(k_cpr_jacobi_lpr<true>, g256(L.n * CPR_LPR), dim3(256), 0, c->stream, L.n, L.W, R.omega, L.d_ecol, L.d_val, L.d_dinv, L.d_b, xin, xout, mem4f, rf, done);
            else if (L.rm) hipLaunchKernelGGL(k_cpr_jacobi_lpr<false>, g256(L.n * CPR_LPR), dim3(256), 0, c->stream, L.n, L.W, R.omega, L.d_ecol, L.d_val, L.d_dinv, L.d_b, xin, xout, (const int4*)nullptr, (const double*)nullptr, done);
            else hipLaunchKernelGGL(k_cpr_jacobi, g256(L.n), dim3(256), 0, c->stream, L.n, L.W, R.omega, L.d_ecol, L.d_val, L.d_dinv, L.d_b, xin, xout, done);
            std::swap(xin, xout);
        }
        return xin;
    }
    CprLevelDev& C = R.lv[l + 1];
    const bool ilu = cpr_ilu_active(R, l);
    const EllStencil S0{L.d_sword, L.d_stable};
    if (ilu) {   // pre-smoothing from x = 0 with the level's ILU0, then the residual
        cpr_ilu_smooth(c, L, L.d_b, L.d_x, nullptr, nullptr, nullptr);
        if (S0.word) hipLaunchKernelGGL(k_cpr_resid<true>, g256(L.n), dim3(256), 0, c->stream, L.n, L.W, L.d_ecol, L.d_val, L.d_b, L.d_x, L.d_r, done, S0);
        else hipLaunchKernelGGL(k_cpr_resid<false>, g256(L.n), dim3(256), 0, c->stream, L.n, L.W, L.d_ecol, L.d_val, L.d_b, L.d_x, L.d_r, done, S0);
    } else
    if (L.rm && fused) hipLaunchKernelGGL(k_cpr_down_lpr<true>, g256(L.n * CPR_LPR), dim3(256), 0, c->stream, L.n, L.W, R.omega, L.d_ecol, L.d_val, L.d_dinv, L.d_b, L.d_x, L.d_r, mem4f, rf, done);
    else if (L.rm) hipLaunchKernelGGL(k_cpr_down_lpr<false>, g256(L.n * CPR_LPR), dim3(256), 0, c->stream, L.n, L.W, R.omega, L.d_ecol, L.d_val, L.d_dinv, L.d_b, L.d_x, L.d_r, (const int4*)nullptr, (const double*)nullptr, done);
    else {   // large levels: x first (it rides in the kernel that produced b), then the residual with ONE gathered value per entry (0.250 -> 0.243 ms per cycle against the fused form, which gathers dinv and b)
        if (!havex) hipLaunchKernelGGL(k_cpr_presmooth, g256(L.n), dim3(256), 0, c->stream, L.n, R.omega, L.d_dinv, L.d_b, L.d_x, done);
        {
            const EllStencil S{L.d_sword, L.d_stable};
            if (S.word) hipLaunchKernelGGL(k_cpr_resid<true>, g256(L.n), dim3(256), 0, c->stream, L.n, L.W, L.d_ecol, L.d_val, L.d_b, L.d_x, L.d_r, done, S);
            else hipLaunchKernelGGL(k_cpr_resid<false>, g256(L.n), dim3(256), 0, c->stream, L.n, L.W, L.d_ecol, L.d_val, L.d_b, L.d_x, L.d_r, done, S);
        }
    }
    if (!cpr_forms_rhs(R, l + 1)) {
        const bool ride = cpr_presmooth_rides(R, l + 1);
        hipLaunchKernelGGL(k_cpr_restrict, g256(L.nc), dim3(256), 0, c->stream, L.nc, L.d_mptr, L.d_midx, L.d_r, C.d_b, R.omega,
                           ride ? C.d_dinv : (const double*)nullptr, ride ? C.d_x : (double*)nullptr, done);
    }
    const double* xc = cpr_vcycle(c, R, l + 1);
    if (ilu) {   // x' = x + damp P xc; r = b - A x'; x = x' + ILU0(r)
        hipLaunchKernelGGL(k_cpr_prolong, g256(L.n), dim3(256), 0, c->stream, L.n, R.damp, L.d_agg, xc, L.d_x, L.d_r, done);
        if (S0.word) hipLaunchKernelGGL(k_cpr_resid<true>, g256(L.n), dim3(256), 0, c->stream, L.n, L.W, L.d_ecol, L.d_val, L.d_b, L.d_r, L.d_x, done, S0);
        else hipLaunchKernelGGL(k_cpr_resid<false>, g256(L.n), dim3(256), 0, c->stream, L.n, L.W, L.d_ecol, L.d_val, L.d_b, L.d_r, L.d_x, done, S0);
        cpr_ilu_smooth(c, L, L.d_x, L.d_t, L.d_r, L.d_x2, fineOut);
        return fineOut ? nullptr : L.d_x2;
    }
    if (L.rm) hipLaunchKernelGGL(k_cpr_up_lpr, g256(L.n * CPR_LPR), dim3(256), 0, c->stream, L.n, L.W, R.omega, R.damp, L.d_ecol, L.d_val, L.d_dinv, L.d_agg, xc, L.d_b, L.d_x, L.d_x2, done);
    else {   // large levels: the prolonged iterate first (into the residual buffer, free by now), then one gathered value per entry
        hipLaunchKernelGGL(k_cpr_prolong, g256(L.n), dim3(256), 0, c->stream, L.n, R.damp, L.d_agg, xc, L.d_x, L.d_r, done);
        {
            const EllStencil S{L.d_sword, L.d_stable};
            if (S.word) hipLaunchKernelGGL(k_cpr_post<true>, g256(L.n), dim3(256), 0, c->stream, L.n, L.W, R.omega, L.d_ecol, L.d_val, L.d_dinv, L.d_b, L.d_r, L.d_x2, fineOut, done, S);
            else hipLaunchKernelGGL(k_cpr_post<false>, g256(L.n), dim3(256), 0, c->stream, L.n, L.W, R.omega, L.d_ecol, L.d_val, L.d_dinv, L.d_b, L.d_r, L.d_x2, fineOut, done, S);
        }
        if (fineOut) return nullptr;
    }
    return L.d_x2;
}

// The pressure stage of a rank whose CPR spans the ranks (cpr_gather_setup): level 0 smooths with the operator of the WHOLE system
// (its couplings to ghost cells are in its image; the iterates' ghost entries come from their owners), its residual is summed over
// the aggregates level by level down to the joined level, one V-cycle there on every rank, the result back per aggregate.  Returns
// x_p, ghost entries up to date (the post-smoothing residual d - A (0, x_p, 0) is the whole system's as well).  Three exchanges of one
// double per boundary cell and one all-gather of the joined level's right-hand side.  oracle: orc_cpr_solve_blocks, gather_rows >= 0.
static const double* cpr_gathered_cycle(opmhip_ctx* c, const double* d) {
    const Pattern& P = c->pat;
    CprDev& R = c->cpr;
    CprGatherDev& G = R.gather;
    CprDev& J = *G.glob;
    CprLevelDev& L = R.lv[0];
    const double* done = c->d_done;
    auto keep = [&](int rc) { if (rc && !R.apply_rc) R.apply_rc = rc; };
    hipLaunchKernelGGL(k_cpr_restrict_fine, g256(P.Nb), dim3(256), 0, c->stream, P.Nb, d, R.d_w, L.d_b, R.omega, L.d_dinv, L.d_x, done);   // b = r_p, x = omega D^-1 b
    keep(comm_halo_f64(c, L.d_x, 1));
    const EllStencil S{nullptr, nullptr};
    hipLaunchKernelGGL(k_cpr_resid<false>, g256(L.n), dim3(256), 0, c->stream, L.n, L.W, L.d_ecol, L.d_val, L.d_b, L.d_x, L.d_r, done, S);
    const size_t g = R.lv.size() - 1;
    const double* rl = L.d_r;
    for (size_t l = 0; l < g; ++l) {   // sums over the aggregates, level by level (no smoothing in between)
        CprLevelDev& F = R.lv[l];
        double* to = l + 1 == g ? G.d_send : R.lv[l + 1].d_b;
        hipLaunchKernelGGL(k_cpr_restrict, g256(F.nc), dim3(256), 0, c->stream, F.nc, F.d_mptr, F.d_midx, rl, to, R.omega, (const double*)nullptr, (double*)nullptr, done);
        rl = to;
    }
    if (g == 0) keep(hipMemcpyAsync(G.d_send, L.d_r, (size_t)L.n * sizeof(double), hipMemcpyDeviceToDevice, c->stream) == hipSuccess ? 0 : OPMHIP_DEVICE_ERROR);
    const int span = prof_span_begin(c, PROF_CPR_GATHER);   // the joined level: all-gather of its right-hand side, the cycle every rank runs on it
    keep(comm_allgather(c, G.d_send, G.d_recv, (size_t)G.maxn));
    {
        const bool ride = cpr_presmooth_rides(J, 0);
        hipLaunchKernelGGL(k_cpr_unpad, g256(G.NG), dim3(256), 0, c->stream, G.NG, G.d_unpad, G.d_recv, J.lv[0].d_b, J.omega,
                           ride ? J.lv[0].d_dinv : (const double*)nullptr, ride ? J.lv[0].d_x : (double*)nullptr, done);
    }
    const double* xG = cpr_vcycle(c, J, 0);
    prof_span_end(c, span);
    hipLaunchKernelGGL(k_cpr_prolong, g256(L.n), dim3(256), 0, c->stream, L.n, 1.0, G.d_cagg, xG + G.off, L.d_x, L.d_r, done);   // x' = x + (the joined level's result, per aggregate)
    keep(comm_halo_f64(c, L.d_r, 1));
    hipLaunchKernelGGL(k_cpr_post<false>, g256(L.n), dim3(256), 0, c->stream, L.n, L.W, R.omega, L.d_ecol, L.d_val, L.d_dinv, L.d_b, L.d_r, L.d_x2, (double*)nullptr, done, S);
    keep(comm_halo_f64(c, L.d_x2, 1));
    return L.d_x2;
}

// v = M_cpr^-1 d (TwoLevelMethodCpr::apply)
void launch_cpr_apply(opmhip_ctx* c, const double* d, double* v) {
    const Pattern& P = c->pat;
    CprDev& R = c->cpr;
    const int n = P.Nb * BS;
    const double* done = c->d_done;
    int ps = prof_begin(c, PROF_CPR_AMG);
    if (R.gather.on) {
        const double* xp = cpr_gathered_cycle(c, d);
        prof_end(c, ps);
        ps = prof_begin(c, PROF_VECTOR);
        const EllStencil S{nullptr, nullptr};
        hipLaunchKernelGGL(k_cpr_presid<false>, g256(P.Nb), dim3(256), 0, c->stream, P.Nb, R.lv[0].W, R.lv[0].d_ecol, R.d_pcol, d, xp, R.d_r, done, S);
        prof_end(c, ps);
        launch_ilu_apply(c, R.d_r, v, 1.0, nullptr, xp, R.d_z);
        return;
    }
    {
        const bool ride = cpr_presmooth_rides(R, 0);
        hipLaunchKernelGGL(k_cpr_restrict_fine, g256(P.Nb), dim3(256), 0, c->stream, P.Nb, d, R.d_w, R.lv[0].d_b, R.omega,
                           ride ? R.lv[0].d_dinv : (const double*)nullptr, ride ? R.lv[0].d_x : (double*)nullptr, done);
    }
    // v = (0, x_p, 0) + ILU0(d - A (0, x_p, 0)): the block vector (0, x_p, 0) itself is never formed - the residual kernel reads
    // x_p, and the backward sweeps of the smoother add their result to it as they store (second_result, solver.hip): the
    // expansion pass, the addition kernel and a 24-byte-per-row store of the post-smoothing are gone
    static const bool separate = tuning_env("OPMHIP_CPR_SEPARATE_ADD") != nullptr;   // A/B switch: the three steps as kernels of their own
    if (separate) {
        const bool direct = R.lv.size() > 1 && !R.lv[0].rm;   // level 0's post-smoother writes v = (0, x_p, 0) itself
        const double* xp = cpr_vcycle(c, R, 0, direct ? v : nullptr);
        if (xp) hipLaunchKernelGGL(k_cpr_prolong_fine, g256(n), dim3(256), 0, c->stream, P.Nb, xp, v, done);
        else xp = R.lv[0].d_x2;   // k_cpr_post left the pressure solution there as well
        prof_end(c, ps);
        ps = prof_begin(c, PROF_VECTOR);
        {
        const EllStencil S{R.lv[0].d_sword, R.lv[0].d_stable};
        if (S.word) hipLaunchKernelGGL(k_cpr_presid<true>, g256(P.Nb), dim3(256), 0, c->stream, P.Nb, R.lv[0].W, R.lv[0].d_ecol, R.d_pcol, d, xp, R.d_r, done, S);
        else hipLaunchKernelGGL(k_cpr_presid<false>, g256(P.Nb), dim3(256), 0, c->stream, P.Nb, R.lv[0].W, R.lv[0].d_ecol, R.d_pcol, d, xp, R.d_r, done, S);
    }
        prof_end(c, ps);
        launch_ilu_apply(c, R.d_r, R.d_z, 1.0);                           // fine smoother: ILU0, relaxation 1
        ps = prof_begin(c, PROF_VECTOR);
        hipLaunchKernelGGL(k_cpr_add, g256(n), dim3(256), 0, c->stream, n, v, R.d_z, done);
        prof_end(c, ps);
        return;
    }
    const double* xp = cpr_vcycle(c, R, 0);
    prof_end(c, ps);
    // post-smoothing on the updated residual r = d - A (0, x_p, 0)
    ps = prof_begin(c, PROF_VECTOR);
    {
        const EllStencil S{R.lv[0].d_sword, R.lv[0].d_stable};
        if (S.word) hipLaunchKernelGGL(k_cpr_presid<true>, g256(P.Nb), dim3(256), 0, c->stream, P.Nb, R.lv[0].W, R.lv[0].d_ecol, R.d_pcol, d, xp, R.d_r, done, S);
        else hipLaunchKernelGGL(k_cpr_presid<false>, g256(P.Nb), dim3(256), 0, c->stream, P.Nb, R.lv[0].W, R.lv[0].d_ecol, R.d_pcol, d, xp, R.d_r, done, S);
    }
    prof_end(c, ps);
    launch_ilu_apply(c, R.d_r, v, 1.0, nullptr, xp, R.d_z);               // fine smoother: ILU0, relaxation 1; v = (0, x_p, 0) + its result
}

// weights handed in (natural order, 3 per row) / back to computed ones
int cpr_set_weights(opmhip_ctx* c, const double* w) {
    const Pattern& P = c->pat;
    CprDev& R = c->cpr;
    if (!w) { R.w_given = false; return OPMHIP_SUCCESS; }
    int rc;
    if (!R.d_w && (rc = dev_alloc(c, &R.d_w, (size_t)P.Nb * BS))) return rc;
    std::vector<double> wi((size_t)P.Nb * BS);
    for (int i = 0; i < P.Nb; ++i)
        for (int k = 0; k < BS; ++k) wi[(size_t)P.toOrder[i] * BS + k] = w[(size_t)i * BS + k];
    OPMHIP_HIP(c, hipMemcpyAsync(R.d_w, wi.data(), wi.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
    R.w_given = true;
    return OPMHIP_SUCCESS;
}
// did the dense LU of the coarsest level meet a vanishing / non-finite pivot in the last cpr_update?  (synchronises)
bool cpr_coarse_pivot_failed(opmhip_ctx* c) {
    const CprDev& R = c->cpr.gather.on ? *c->cpr.gather.glob : c->cpr;   // decomposed run with a joined level: that hierarchy's coarsest level
    if (!R.structured || !R.coarse_direct || !R.d_lu) return false;
    double flag = 0.0;
    const size_t n = (size_t)R.lv.back().n;
    if (hipMemcpy(&flag, R.d_lu + n * n, sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return false;
    return flag != 0.0;
}
int cpr_level_sizes(const opmhip_ctx* c, int* n, int* nnz, int cap) {
    // the rank's own levels, then - decomposed runs with a joined level - the levels of the hierarchy all ranks share
    int L = 0;
    for (const CprLevelDev& lv : c->cpr.lv) { if (L < cap) { n[L] = lv.n; nnz[L] = lv.nnz; } ++L; }
    if (c->cpr.gather.on)
        for (const CprLevelDev& lv : c->cpr.gather.glob->lv) { if (L < cap) { n[L] = lv.n; nnz[L] = lv.nnz; } ++L; }
    return L;
}

}  // namespace opmhip
