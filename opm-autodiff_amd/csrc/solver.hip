// gfx950 kernels of the linear-solve half of the hot path: block-CSR SpMV, block ILU0 factor/apply over a
// level/colour schedule, standard-well operator, fused BiCGStab vector kernels with device-resident scalars.
//
// Design ("tile kernels"): one workgroup = one wavefront = one tile of <= TILE_ROWS (32) block rows.  The
// tile's 72-byte blocks form ONE contiguous byte range of the value array, so the wavefront streams that
// range into LDS with 16-byte-per-lane coalesced loads (HBM sees only full-line, unit-stride traffic), and then
// every lane walks its own row out of LDS in exactly the CPU's sequential operation order.  Lane stride in LDS
// is 63 doubles for the 7-point stencil: 63*2 mod 64 banks = 62, i.e. ds_read_b64 from 32 lanes hits 32
// distinct bank pairs - conflict free.  Built with -ffp-contract=off: a*b+c is never fused, so the factors and
// sweeps are bit-identical to the CPU restatement in the same ordering.
// Line-coloured orderings add the chain kernels (heavy: LDS-staged, software-pipelined over the steps of a chain tile;
// light: lane-private recurrences); the BiCGStab driver keeps the
// stopping rule on the device and runs the host one half iteration ahead of it.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <atomic>

#include <chrono>
#include <cmath>
#include <cstdlib>

#include "internal.hpp"

namespace opmhip {

// ============================== device helpers ==========================================================
// Stream n doubles (16-byte aligned source) into LDS.  All loads of a batch are issued before the first LDS write so
// that a wavefront keeps STAGE_DEPTH KiB in flight (Little's law: ~64 KB per CU are needed to cover HBM latency at
// full bandwidth, MI355X_MICROARCH.md "Persistent kernels" glossary: 'streaming' = 32 KiB per CU in flight).
#ifndef OPMHIP_STAGE_DEPTH
#define OPMHIP_STAGE_DEPTH 16
#endif
// 16-byte load of a value that is read exactly once per launch (matrix / factor streams): nontemporal, so that the stream
// does not push the vectors out of L2 and the Infinity Cache.  tools/probe/stream_probe.hip on MI355X: a 500 MB stream read
// in 16-KiB tiles reaches 6.2 TB/s with plain loads and 6.9 TB/s with nontemporal ones.
typedef double v2d_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ld_stream(const double2* p) {
#if defined(OPMHIP_NO_NT_LOADS)
    return *p;
#else
    const v2d_t t = __builtin_nontemporal_load(reinterpret_cast<const v2d_t*>(p));
    return make_double2(t.x, t.y);
#endif
}
// (round 6, measured and not kept: the sweeps' own once-read items - D^-1, the light sweeps' blocks, the right-hand side - as nontemporal
// loads: one M^-1 0.143 -> 0.153 / 0.152 / 0.142 ms, all three 0.171; profiles/r06_nt_operands_ab.txt)
// the backward sweeps' row sums: written once per application, read once by the product that follows
__device__ __forceinline__ void st_rowsum(double* p, double v) {
    __builtin_nontemporal_store(v, p);   // one M^-1 with the row sums back to back: 0.136 - 0.139 ms with plain stores, 0.132 - 0.133 nontemporal
}
__device__ __forceinline__ void stage_doubles(const double* __restrict__ src, double* __restrict__ dst, int n, int lane) {
    const double2* __restrict__ s2 = reinterpret_cast<const double2*>(src);
    double2* __restrict__ d2 = reinterpret_cast<double2*>(dst);
    const int n2 = n >> 1;
#if OPMHIP_STAGE_DEPTH == 0
#pragma unroll 8
    for (int i = lane; i < n2; i += 64) d2[i] = s2[i];
#else
    constexpr int U = OPMHIP_STAGE_DEPTH;
    for (int base = 0; base < n2; base += 64 * U) {
        double2 tmp[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + u * 64 + lane;
            tmp[u] = ld_stream(&s2[i < n2 ? i : n2 - 1]);  // clamped, unconditional: keeps tmp[] in registers
        }
        // keep the whole batch of loads ahead of the first LDS write: without these empty asm "uses" hipcc sinks every
        // load under its store's bounds check and serialises load -> s_waitcnt vmcnt(0) -> ds_write, 1 KiB at a time
#pragma unroll
        for (int u = 0; u < U; ++u) asm volatile("" : "+v"(tmp[u].x), "+v"(tmp[u].y));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + u * 64 + lane;
            if (i < n2) d2[i] = tmp[u];
        }
    }
#endif
    if ((n & 1) && lane == 0) dst[n - 1] = src[n - 1];
}
__device__ __forceinline__ void stage_ints(const int* __restrict__ src, int* __restrict__ dst, int n, int lane) {
#pragma unroll 4
    for (int i = lane; i < n; i += 64) dst[i] = src[i];
}
// Workgroup -> tile map.  Default: identity - measured on MI355X the round-robin dealing of consecutive workgroups
// over the 8 XCDs streams the matrix through all 8 L2s at once and was faster (SpMV 0.112 vs 0.125 ms at 100^3) than
// the XCD-contiguous map below, which keeps each XCD on one contiguous eighth of the tile range (b and b+8 share an
// XCD) so that gathered vector entries stay in one 4 MB L2; build with -DOPMHIP_XCD_MAP to get it.  Speed only; any
// placement gives the same result.
__device__ __forceinline__ int xcd_tile(int b, int nt) {
#if defined(OPMHIP_XCD_MAP)
    const int chunk = (nt + 7) >> 3;
    const int t = (b & 7) * chunk + (b >> 3);
    return t;  // may be >= nt for the padded tail: callers check
#else
    (void)nt;
    return b;
#endif
}

// y -= A x, y += A x, y = A x in dune-common DenseMatrix order (row outer, column inner)
__device__ __forceinline__ void blk_mmv(const double* A, const double x0, const double x1, const double x2, double* y) {
    y[0] -= A[0] * x0; y[0] -= A[1] * x1; y[0] -= A[2] * x2;
    y[1] -= A[3] * x0; y[1] -= A[4] * x1; y[1] -= A[5] * x2;
    y[2] -= A[6] * x0; y[2] -= A[7] * x1; y[2] -= A[8] * x2;
}
// The same with the block known to live in LDS.  The pointer carries its address space in its type, so that the compiler
// can neither treat it as "flat" nor merge this path with the global-memory one behind a single flat pointer (it does
// that to two branches that differ in nothing but the pointer): a flat load makes the wavefront wait for EVERY outstanding
// memory access - in the pipelined sweeps that is the prefetch of the next step.
typedef __attribute__((address_space(3))) const double lds_cdouble;
__device__ __forceinline__ void blk_mmv_lds(const double* Agen, const double x0, const double x1, const double x2, double* y) {
    lds_cdouble* A = (lds_cdouble*)Agen;
    y[0] -= A[0] * x0; y[0] -= A[1] * x1; y[0] -= A[2] * x2;
    y[1] -= A[3] * x0; y[1] -= A[4] * x1; y[1] -= A[5] * x2;
    y[2] -= A[6] * x0; y[2] -= A[7] * x1; y[2] -= A[8] * x2;
}
// y -= A x and, beside it, u += A x from the SAME rounded products: the backward sweep's row sums u_i = sum_{j>i} U_ij x_j (Pattern::ualias)
__device__ __forceinline__ void blk_mmv_u(const double* A, const double x0, const double x1, const double x2, double* y, double* u) {
    double p;
    p = A[0] * x0; y[0] -= p; u[0] += p; p = A[1] * x1; y[0] -= p; u[0] += p; p = A[2] * x2; y[0] -= p; u[0] += p;
    p = A[3] * x0; y[1] -= p; u[1] += p; p = A[4] * x1; y[1] -= p; u[1] += p; p = A[5] * x2; y[1] -= p; u[1] += p;
    p = A[6] * x0; y[2] -= p; u[2] += p; p = A[7] * x1; y[2] -= p; u[2] += p; p = A[8] * x2; y[2] -= p; u[2] += p;
}
__device__ __forceinline__ void blk_mmv_lds_u(const double* Agen, const double x0, const double x1, const double x2, double* y, double* u) {
    lds_cdouble* A = (lds_cdouble*)Agen;
    double p;
    p = A[0] * x0; y[0] -= p; u[0] += p; p = A[1] * x1; y[0] -= p; u[0] += p; p = A[2] * x2; y[0] -= p; u[0] += p;
    p = A[3] * x0; y[1] -= p; u[1] += p; p = A[4] * x1; y[1] -= p; u[1] += p; p = A[5] * x2; y[1] -= p; u[1] += p;
    p = A[6] * x0; y[2] -= p; u[2] += p; p = A[7] * x1; y[2] -= p; u[2] += p; p = A[8] * x2; y[2] -= p; u[2] += p;
}
__device__ __forceinline__ void blk_umv_lds(const double* Agen, const double x0, const double x1, const double x2, double* y) {
    lds_cdouble* A = (lds_cdouble*)Agen;
    y[0] += A[0] * x0; y[0] += A[1] * x1; y[0] += A[2] * x2;
    y[1] += A[3] * x0; y[1] += A[4] * x1; y[1] += A[5] * x2;
    y[2] += A[6] * x0; y[2] += A[7] * x1; y[2] += A[8] * x2;
}
__device__ __forceinline__ void blk_umv(const double* A, const double x0, const double x1, const double x2, double* y) {
    y[0] += A[0] * x0; y[0] += A[1] * x1; y[0] += A[2] * x2;
    y[1] += A[3] * x0; y[1] += A[4] * x1; y[1] += A[5] * x2;
    y[2] += A[6] * x0; y[2] += A[7] * x1; y[2] += A[8] * x2;
}
// C = A * B with the inner sum starting from 0.0 (DenseMatrix::rightmultiply / leftmultiply)
__device__ __forceinline__ void blk_mul(const double* A, const double* B, double* C) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            s += A[i * 3 + 0] * B[0 * 3 + j];
            s += A[i * 3 + 1] * B[1 * 3 + j];
            s += A[i * 3 + 2] * B[2 * 3 + j];
            C[i * 3 + j] = s;
        }
}
// closed-form inverse, expression tree of Opm::Detail::Inverter<3> (linalg/MatrixBlock.hpp:722-747)
__device__ __forceinline__ void blk_invert(const double* m, double* inv) {
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
// The tile kernels run ONE wavefront per workgroup.  A wavefront's LDS instructions execute in issue order and its
// global stores are visible to its own later loads once they have left the wave (same CU, write-through L1), so no
// s_barrier is needed between "lanes wrote LDS" and "other lanes read it" - only the compiler must keep the order.
// __syncthreads() would also work but it drains vmcnt(0), i.e. it waits for every prefetch that was just issued.
__device__ __forceinline__ void wave_sync() { asm volatile("" ::: "memory"); }
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;  // valid in lane 0
}

// ---- tile machinery ------------------------------------------------------------------------------------------
// A tile = rows [r0, r1) of one CSR triple (rowptr, col, val); one row per lane.  Memory operations are ordered so that
// as much as possible is in flight at once:
//   scalar: r0, r1, k0, k1                       (tile range)
//   vector: kb, ke of the lane's row             -> column indices of the row's first GCH blocks (straight from HBM/L2)
//   then, back to back: the tile's value stream (16 B per lane per load, OPMHIP_STAGE_DEPTH loads in flight) and the
//   3-double gathers of the vector entries those columns point at
//   values -> LDS, barrier, then every lane multiplies its blocks out of LDS in the row's sequential order.
struct TileCtx {
    int r0, r1, k0e, nb;
    bool staged;
};
#ifndef OPMHIP_GATHER_CHUNK
#define OPMHIP_GATHER_CHUNK 8
#endif
constexpr int GCH = OPMHIP_GATHER_CHUNK;

#define TILE_LDS __shared__ __attribute__((aligned(16))) double sval[(TILE_CAP_BLOCKS + 2) * BB];

// values of tile t -> LDS (used by the factorisation, which needs no vector gathers); contains a barrier
__device__ __forceinline__ TileCtx tile_stage_values(int t, const int* __restrict__ tile_row0, const int* __restrict__ rowptr,
                                                     const double* __restrict__ val, double* sval, int lane) {
    TileCtx T;
    T.r0 = tile_row0[t];
    T.r1 = tile_row0[t + 1];
    const int k0 = rowptr[T.r0], k1 = rowptr[T.r1];
    T.k0e = k0 & ~1;
    T.nb = k1 - T.k0e;
    T.staged = (T.nb <= TILE_CAP_BLOCKS + 1);
    if (T.staged && T.nb > 0) stage_doubles(val + (size_t)T.k0e * BB, sval, T.nb * BB, lane);
    wave_sync();
    return T;
}

// y -= A x / y += A x per block, see blk_mmv / blk_umv
template <bool SUB>
__device__ __forceinline__ void blk_apply(const double* A, const double* xx, double* acc) {
    if (SUB) blk_mmv(A, xx[0], xx[1], xx[2], acc); else blk_umv(A, xx[0], xx[1], xx[2], acc);
}
template <bool SUB>
__device__ __forceinline__ void blk_apply_lds(const double* A, const double* xx, double* acc) {
    if (SUB) blk_mmv_lds(A, xx[0], xx[1], xx[2], acc); else blk_umv_lds(A, xx[0], xx[1], xx[2], acc);
}

// Accumulates acc (+/-)= sum_k A_k x[col_k] over the lane's row of tile t, blocks taken in ascending (or, with
// reverse, descending) column order.  Returns the lane's row index or -1 for an idle lane.  Contains a barrier.
template <bool SUB>
__device__ __forceinline__ int tile_row_product(int tr0, int tr1, const int* __restrict__ rowptr,
                                                const int* __restrict__ col, const double* __restrict__ val,
                                                const double* __restrict__ x, double* sval, int lane, bool reverse, double* acc,
                                                TileCtx& T, const double* __restrict__ xlo = nullptr, int xsplit = 0,
                                                int tk0 = -1, int tk1 = -1, double xs = 1.0) {
    // vector entries of columns < xsplit are read from xlo instead of x (ILU sweeps: first colour's y equals d)
    // tk0, tk1 >= 0: the tile's entry range is known already (it came with the launch schedule): one dependent load less
    // before the value stream can be issued
    T.r0 = tr0;
    T.r1 = tr1;
    const int k0 = tk0 >= 0 ? tk0 : rowptr[T.r0], k1 = tk1 >= 0 ? tk1 : rowptr[T.r1];
    T.k0e = k0 & ~1;
    T.nb = k1 - T.k0e;
    T.staged = (T.nb <= TILE_CAP_BLOCKS + 1);
    const int r = T.r0 + lane;
    const bool active = r < T.r1;
    const int rr = active ? r : T.r1 - 1;
    const int kb = rowptr[rr];
    const int ke = active ? rowptr[rr + 1] : kb;
    const int nrow = ke - kb;
    // columns of the first chunk; slots beyond the row's length point at the lane's own row (always a valid vector
    // entry, never used in the sums) - a colour's first/last sweep has rows, even whole tiles, without any entry
    int cc[GCH];
#pragma unroll
    for (int u = 0; u < GCH; ++u) {
        const int k = reverse ? ke - 1 - u : kb + u;
        cc[u] = (u < nrow) ? col[k] : rr;
    }
#pragma unroll
    for (int u = 0; u < GCH; ++u) asm volatile("" : "+v"(cc[u]));
    // value stream + vector gathers, all issued before anything is waited for
    double xx[GCH][3];
    const double2* __restrict__ s2 = reinterpret_cast<const double2*>(val + (size_t)T.k0e * BB);
    double2* __restrict__ d2 = reinterpret_cast<double2*>(sval);
    const int n = T.nb * BB, n2 = n >> 1;
    constexpr int U = OPMHIP_STAGE_DEPTH;
    if (T.staged && n2 > 0) {
        double2 tmp[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = u * 64 + lane;
            tmp[u] = ld_stream(&s2[i < n2 ? i : n2 - 1]);
        }
#pragma unroll
        for (int u = 0; u < GCH; ++u) {
            const double* xc = (cc[u] < xsplit) ? &xlo[(size_t)cc[u] * BS] : &x[(size_t)cc[u] * BS];
            xx[u][0] = xc[0]; xx[u][1] = xc[1]; xx[u][2] = xc[2];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) asm volatile("" : "+v"(tmp[u].x), "+v"(tmp[u].y));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = u * 64 + lane;
            if (i < n2) d2[i] = tmp[u];
        }
        for (int base = 64 * U; base < n2; base += 64 * U) {  // tiles larger than one batch
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = base + u * 64 + lane;
                tmp[u] = ld_stream(&s2[i < n2 ? i : n2 - 1]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) asm volatile("" : "+v"(tmp[u].x), "+v"(tmp[u].y));
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = base + u * 64 + lane;
                if (i < n2) d2[i] = tmp[u];
            }
        }
        if ((n & 1) && lane == 0) sval[n - 1] = val[(size_t)T.k0e * BB + n - 1];
    } else {
#pragma unroll
        for (int u = 0; u < GCH; ++u) {
            const double* xc = (cc[u] < xsplit) ? &xlo[(size_t)cc[u] * BS] : &x[(size_t)cc[u] * BS];
            xx[u][0] = xc[0]; xx[u][1] = xc[1]; xx[u][2] = xc[2];
        }
    }
    wave_sync();
    if (!active) return -1;
    const bool stagedU = __builtin_amdgcn_readfirstlane((int)T.staged) != 0;   // the same in every lane: a scalar branch
    // first chunk
#pragma unroll
    for (int u = 0; u < GCH; ++u) {
        if (u < nrow) {
            const int k = reverse ? ke - 1 - u : kb + u;
            // two branches, not one pointer chosen between LDS and global memory: such a pointer is "flat", and a flat load
            // makes the wavefront wait for EVERY outstanding memory access, the prefetched next tile included
            const double xs3[3] = {xs * xx[u][0], xs * xx[u][1], xs * xx[u][2]};   // xs = 1 unless the vector is an unscaled M^-1 result
            if (stagedU) blk_apply_lds<SUB>(&sval[(k - T.k0e) * BB], xs3, acc);
            else blk_apply<SUB>(&val[(size_t)k * BB], xs3, acc);
        }
    }
    // rows longer than one chunk
    for (int done = GCH; done < nrow; done += GCH) {
#pragma unroll
        for (int u = 0; u < GCH; ++u) {
            const int q = (done + u < nrow) ? done + u : nrow - 1;
            const int k = reverse ? ke - 1 - q : kb + q;
            const int cq = col[k];
            const double* xc = (cq < xsplit) ? &xlo[(size_t)cq * BS] : &x[(size_t)cq * BS];
            xx[u][0] = xc[0]; xx[u][1] = xc[1]; xx[u][2] = xc[2];
        }
#pragma unroll
        for (int u = 0; u < GCH; ++u) {
            if (done + u < nrow) {
                const int k = reverse ? ke - 1 - (done + u) : kb + done + u;
                const double xs3[3] = {xs * xx[u][0], xs * xx[u][1], xs * xx[u][2]};
                if (stagedU) blk_apply_lds<SUB>(&sval[(k - T.k0e) * BB], xs3, acc);
                else blk_apply<SUB>(&val[(size_t)k * BB], xs3, acc);
            }
        }
    }
    return r;
}

// ============================== permutations =============================================================
__global__ void k_permute_blocks(int nnzb, const int* __restrict__ nnzMap, const double* __restrict__ nat,
                                 double* __restrict__ internal) {
    // internal block k <- natural block nnzMap[k]; one lane per scalar so that writes are unit stride
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)nnzb * BB) return;
    const int k = (int)(e / BB), q = (int)(e % BB);
    internal[e] = nat[(size_t)nnzMap[k] * BB + q];
}
__global__ void k_vec_to_internal(int Nb, const int* __restrict__ fromOrder, const double* __restrict__ nat,
                                  double* __restrict__ internal) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= Nb * BS) return;
    internal[e] = nat[(size_t)fromOrder[e / BS] * BS + e % BS];
}
__global__ void k_vec_to_natural(int Nb, const int* __restrict__ toOrder, const double* __restrict__ internal,
                                 double* __restrict__ nat) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= Nb * BS) return;
    nat[e] = internal[(size_t)toOrder[e / BS] * BS + e % BS];
}
// checkZeroDiagonal on the device copy (bda/BdaBridge.cpp:125-161)
__global__ void k_zero_diag_fix(int Nb, const int* __restrict__ diag, double* __restrict__ A) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= Nb * BS) return;
    double* v = &A[(size_t)diag[e / BS] * BB + (e % BS) * 4];
    if (*v == 0.0) *v = 1e-15;
}
// factors back into the block-CSR layout of the (reordered) matrix: lower = L, diagonal = D^-1, upper = U
__global__ void k_lu_to_bcrs(int Nb, const int* __restrict__ rowptr, const int* __restrict__ col,
                             const int* __restrict__ lrowptr, const int* __restrict__ urowptr,
                             const double* __restrict__ L, const double* __restrict__ U, const double* __restrict__ invD,
                             double* __restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= Nb) return;
    int li = lrowptr[p], ui = urowptr[p];
    for (int k = rowptr[p]; k < rowptr[p + 1]; ++k) {
        if (col[k] >= Nb) {  // ghost column: not part of the block-Jacobi ILU0
            for (int q = 0; q < BB; ++q) out[(size_t)k * BB + q] = 0.0;
            continue;
        }
        const double* src = (col[k] < p) ? &L[(size_t)(li++) * BB] : (col[k] == p) ? &invD[(size_t)p * BB] : &U[(size_t)(ui++) * BB];
        for (int q = 0; q < BB; ++q) out[(size_t)k * BB + q] = src[q];
    }
}

// ============================== SpMV ======================================================================
// y = A x in BCRSMatrix::mv order (y_i = 0, then umv block by block in ascending column order).
// NDOT = 1: part[t] = sum_rows y.w0            NDOT = 2: additionally part[npart+t] = sum_rows y.y
// Launch position b -> rows [sched[b].x, sched[b].y) with entries [sched[b].z, sched[b].w) (an empty range = padding).  The schedule is built on the host
// (reorder.cpp: build_schedules): consecutive workgroups land on consecutive XCDs, and the schedule gives every XCD runs of
// tiles that gather from the same stretch of the input vector, so that a line of x is fetched by one L2, not by five.
template <int NDOT>
__global__ __launch_bounds__(64) void k_spmv(const int4* __restrict__ sched, const int* __restrict__ rowptr,
                                             const int* __restrict__ col, const double* __restrict__ val,
                                             const double* __restrict__ x, double* __restrict__ y,
                                             const double* __restrict__ w0, double* __restrict__ part, int npart,
                                             const double* __restrict__ done, double xs) {
    TILE_LDS
    const int lane = threadIdx.x, t = blockIdx.x;
    if (*done != 0.0) return;
    const int4 rows = sched[t];   // r0, r1, rowptr[r0], rowptr[r1]
    if (rows.y <= rows.x) {  // padding of the schedule: its partial sums are zero
        if (NDOT >= 1 && lane == 0) { part[t] = 0.0; if (NDOT == 2) part[npart + t] = 0.0; }
        return;
    }
    TileCtx T;
    double acc[3] = {0.0, 0.0, 0.0};
    const int r = tile_row_product<false>(rows.x, rows.y, rowptr, col, val, x, sval, lane, false, acc, T, nullptr, 0, rows.z, rows.w, xs);
    if (r >= 0) {
        double* yr = &y[(size_t)r * BS];
        yr[0] = acc[0]; yr[1] = acc[1]; yr[2] = acc[2];
    }
    if (NDOT >= 1) {
        double d0 = 0.0, d1 = 0.0;
        if (r >= 0) {
            const double* w = &w0[(size_t)r * BS];
            d0 = acc[0] * w[0]; d0 += acc[1] * w[1]; d0 += acc[2] * w[2];
            if (NDOT == 2) { d1 = acc[0] * acc[0]; d1 += acc[1] * acc[1]; d1 += acc[2] * acc[2]; }
        }
        d0 = wave_sum(d0);
        if (NDOT == 2) d1 = wave_sum(d1);
        if (lane == 0) {
            part[t] = d0;
            if (NDOT == 2) part[npart + t] = d1;
        }
    }
}

// Pipelined SpMV: the same tiles, the same per-row arithmetic, but every workgroup (one wavefront) walks through a list of
// tiles - launch positions blockIdx.x, + gridDim.x, + 2 gridDim.x ... - in a software pipeline, so that the chain of
// dependent loads of a tile (schedule entry -> row bounds -> column indices -> vector gathers) runs one to three tiles
// AHEAD of the arithmetic and the next tile's 16-KiB value stream is always in flight while this tile is multiplied out
// of LDS.  In the one-tile-per-workgroup kernel a wavefront spends most of its ~9 us life waiting for those four hops one
// after the other and the launch moves 5.2 TB/s; a stream that is always in flight reaches 6.2-6.9 TB/s on this card
// (tools/probe/stream_probe.hip).  Used when no row is longer than PGCH blocks (the host checks), else k_spmv.
//   S(st+4): schedule entry (scalar)   A(st+3): row bounds   C(st+2): column indices
//   G(st+1): value stream + vector gathers                    X(st): values -> LDS, products, store
// The BiCGStab scalar products that follow a product ride in this kernel (NDOT = 1: y.w0, NDOT = 2: y.w0 and y.y): the third
// operand w0 is fetched in the stage that fetches a row's small items, every lane adds up the products of ITS rows over all
// its tiles and the wavefront is summed once, at the end - one partial sum (pair) per workgroup.  Measured in round 2: the
// product gets 4 % longer (a third operand stream of 24 MB), the two k_dots launches and their kernel boundaries go, +1.3 %
// Newton iterations/s at equal iteration counts.  (Per-tile wavefront reductions, the first way this was tried, cost 8 %.)
constexpr int PGCH = 8;
constexpr int PIPE_MAX_STEPS = 128;   // schedule entries of one workgroup, kept in LDS (2 KiB: eight workgroups per CU must still fit; the host sizes the grid accordingly)
// Written by the rules listed at chain_sweep (which see): the tile's schedule entries come out of LDS (one round of loads
// at the start instead of a scalar load per tile at the head of every dependency chain); stages A (row bounds, 3 tiles
// ahead), C (column indices, 2 ahead), M (vector gathers and the row bounds once more, 1 ahead), S (value stream, 1
// ahead); every load unconditional, into registers whose content is dead; nothing loaded is copied.
template <int NDOT>
__global__ __launch_bounds__(64) void k_spmv_pipe(int npos, const int4* __restrict__ sched, const int* __restrict__ rowptr,
                                                  const int* __restrict__ col, const double* __restrict__ val,
                                                  const double* __restrict__ x, double* __restrict__ y,
                                                  const double* __restrict__ w0, double* __restrict__ part, int npart,
                                                  const double* __restrict__ done, double xs) {
    TILE_LDS
    __shared__ int4 ssched[PIPE_MAX_STEPS];
    const int lane = threadIdx.x, G = gridDim.x;
    constexpr int U = 16;  // 16 x 64 lanes x 16 B = 16 KiB >= any staged tile
    const int nsteps = ((int)blockIdx.x < npos) ? (npos - (int)blockIdx.x + G - 1) / G : 0;   // <= PIPE_MAX_STEPS (host)
    if (nsteps <= 0) {   // a workgroup of the rounded-up grid without work: its partial sums are zero
        if (NDOT >= 1 && lane == 0) { part[blockIdx.x] = 0.0; if (NDOT == 2) part[npart + blockIdx.x] = 0.0; }
        return;
    }
    const double stop = *done;                      // read together with the schedule entries: one round trip, not two
    for (int i = lane; i < nsteps; i += 64) ssched[i] = sched[(int)blockIdx.x + i * G];
    if (stop != 0.0) return;
    wave_sync();
    struct StA { int kb, ke; };
    struct StS {
        int k0e, n, n2;
        double2 tmp[U];
    };
    struct StM {
        int kb, ke;
        double xx[PGCH][3];
        double ww[NDOT >= 1 ? 3 : 1];   // the second operand of the scalar product at the lane's row
    };
    auto clampst = [&](int st) { return st < nsteps ? st : nsteps - 1; };
    auto row_of = [&](int st, bool& active) {   // LDS only
        const int4 rows = ssched[clampst(st)];
        active = st < nsteps && rows.x + lane < rows.y;
        return (rows.x + lane < rows.y) ? rows.x + lane : (rows.y > rows.x ? rows.y - 1 : 0);
    };
    auto stageA = [&](int st, StA& a) {
        bool act;
        const int rr = row_of(st, act);
        a.kb = rowptr[rr];
        a.ke = rowptr[rr + 1];
    };
    auto stageC = [&](int st, const StA& a, int (&cc)[PGCH]) {   // a = row bounds of tile st
        bool act;
        const int rr = row_of(st, act);
        const int nrow = a.ke - a.kb;
        const int spare = a.kb > 0 ? a.kb - 1 : 0;
        (void)rr;
#pragma unroll
        for (int u = 0; u < PGCH; ++u) cc[u] = col[(u < nrow) ? a.kb + u : spare];   // idle slots: some valid entry (a valid row to gather)
    };
    auto stageGs = [&](int st, StS& b) {   // value stream of tile st; past the end: one line, read 16 times
        const int4 rows = ssched[clampst(st)];
        b.k0e = rows.z & ~1;
        const int nb = rows.w - b.k0e;
        b.n = (rows.y > rows.x) ? nb * BB : 0;
        b.n2 = b.n >> 1;
        const int lim = (st < nsteps && b.n2 > 0) ? b.n2 - 1 : 0;
        const double2* __restrict__ s2 = reinterpret_cast<const double2*>(val + (size_t)b.k0e * BB);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = u * 64 + lane;
            b.tmp[u] = ld_stream(&s2[i < lim ? i : lim]);
        }
    };
    auto stageGm = [&](int st, const int (&cc)[PGCH], StM& b) {   // cc = column indices of tile st
        bool act;
        const int rr = row_of(st, act);
        b.kb = rowptr[rr];
        b.ke = rowptr[rr + 1];
#pragma unroll
        for (int u = 0; u < PGCH; ++u) {
            const double* xc = &x[(size_t)cc[u] * BS];
            b.xx[u][0] = xc[0]; b.xx[u][1] = xc[1]; b.xx[u][2] = xc[2];
        }
        if (NDOT >= 1) {
            const double* wr = &w0[(size_t)rr * BS];
            b.ww[0] = wr[0]; b.ww[1] = wr[1]; b.ww[2] = wr[2];
        }
    };
    StA a;
    int cc[PGCH];
    StS sb;
    StM m;
    double sd0 = 0.0, sd1 = 0.0;   // the lane's share of y.w0 and y.y: its rows, tile after tile
    {   // prologue, in the loop's order of issue (stream, gathers, column indices, row bounds)
        StA a0, a1;
        int c0[PGCH];
        stageA(0, a0);
        stageA(1, a1);
        stageC(0, a0, c0);
        stageGs(0, sb);
        stageGm(0, c0, m);
        stageC(1, a1, cc);
        stageA(2, a);
    }
    double2* d2 = reinterpret_cast<double2*>(sval);
    for (int st = 0; st < nsteps; ++st) {
        // ---- X(st), part 1: this tile's values -> LDS
        const int k0e = sb.k0e, n = sb.n, n2 = sb.n2;
        if (n2 > 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = u * 64 + lane;
                if (i < n2) d2[i] = sb.tmp[u];
            }
            if ((n & 1) && lane == 0) sval[n - 1] = val[(size_t)k0e * BB + n - 1];
        }
        wave_sync();
        // ---- the next tile's value stream goes out now and flies during this tile's arithmetic
        stageGs(st + 1, sb);
        // ---- X(st), part 2: products in BCRSMatrix::mv order, store
        bool active;
        const int r = row_of(st, active);
        if (active) {
            const int kb = m.kb, nrow = m.ke - m.kb;
            double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll
            for (int u = 0; u < PGCH; ++u)
                if (u < nrow) blk_umv_lds(&sval[(kb + u - k0e) * BB], xs * m.xx[u][0], xs * m.xx[u][1], xs * m.xx[u][2], acc);   // xs = 1 unless x is an unscaled M^-1 result
            double* yr = &y[(size_t)r * BS];
            yr[0] = acc[0]; yr[1] = acc[1]; yr[2] = acc[2];
            if (NDOT >= 1) {
                double d0 = acc[0] * m.ww[0]; d0 += acc[1] * m.ww[1]; d0 += acc[2] * m.ww[2];
                sd0 += d0;
                if (NDOT == 2) { double d1 = acc[0] * acc[0]; d1 += acc[1] * acc[1]; d1 += acc[2] * acc[2]; sd1 += d1; }
            }
        }
        wave_sync();  // the LDS image may be overwritten
        // ---- the tiles further ahead, each stage into the registers it has just finished with
        __builtin_amdgcn_sched_barrier(0);
        stageGm(st + 1, cc, m);
        __builtin_amdgcn_sched_barrier(0);
        stageC(st + 2, a, cc);
        __builtin_amdgcn_sched_barrier(0);
        stageA(st + 3, a);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (NDOT >= 1) {
        sd0 = wave_sum(sd0);
        if (NDOT == 2) sd1 = wave_sum(sd1);
        if (lane == 0) { part[blockIdx.x] = sd0; if (NDOT == 2) part[npart + blockIdx.x] = sd1; }
    }
}

// The pipelined SpMV on the STENCIL form of its index streams (TileSet::stWord / stKoff / stTable): stage W (two tiles ahead)
// fetches a row's word of eight 4-bit table indices, its first-entry byte and - lanes 0..15 - the tile's offset table; stage
// M (one ahead) turns them into column indices (table entries travel lane to lane, no LDS, no memory) and gathers.  No row
// bounds, no column indices: 5 bytes per row instead of 44, and the gathers wait for ONE round of loads instead of two.
// Same tiles, same per-row arithmetic in the same order as k_spmv_pipe: the same bits.
// UADD (the "rest product", Pattern::ualias): val / sched / stWord ... describe the matrix WITHOUT its U part and the row's sum ends with
// + xs * uadd_i, uadd = the row sums sum_{j>i} U_ij x_j the backward sweep of the ILU0 application left behind (U == upper(A) bit for bit):
// y_i = (sum over the lower entries, the diagonal and the ghost columns, ascending, of A_ik (xs x_k)) + xs u_i.
template <int NDOT, bool UADD = false>
__global__ __launch_bounds__(64) void k_spmv_pipe_st(int npos, const int4* __restrict__ sched, const unsigned* __restrict__ stWord,
                                                     const unsigned char* __restrict__ stKoff, const int* __restrict__ stTable,
                                                     const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y,
                                                     const double* __restrict__ w0, double* __restrict__ part, int npart,
                                                     const double* __restrict__ done, double xs, const double* __restrict__ uadd = nullptr,
                                                     const double* __restrict__ w1 = nullptr) {
    // NDOT == 3 (opmhip_config.fused_reductions: one reduction per half iteration): y.w0, y.y and y.w1
    TILE_LDS
    __shared__ int4 ssched[PIPE_MAX_STEPS];
    const int lane = threadIdx.x, G = gridDim.x;
    constexpr int U = 16;
    const int nsteps = ((int)blockIdx.x < npos) ? (npos - (int)blockIdx.x + G - 1) / G : 0;
    if (nsteps <= 0) {
        if (NDOT >= 1 && lane == 0) { part[blockIdx.x] = 0.0; if (NDOT >= 2) part[npart + blockIdx.x] = 0.0; if (NDOT == 3) part[2 * npart + blockIdx.x] = 0.0; }
        return;
    }
    const double stop = *done;
    for (int i = lane; i < nsteps; i += 64) ssched[i] = sched[(int)blockIdx.x + i * G];
    if (stop != 0.0) return;
    wave_sync();
    struct StW { unsigned w; int koff, tab; };
    struct StS {
        int k0e, n, n2;
        double2 tmp[U];
    };
    struct StM {
        int kb, nrow;
        double xx[PGCH][3];
        double ww[NDOT >= 1 ? 3 : 1];
        double w2[NDOT == 3 ? 3 : 1];
        double uu[UADD ? 3 : 1];
    };
    auto clampst = [&](int st) { return st < nsteps ? st : nsteps - 1; };
    auto row_of = [&](int st, bool& active) {   // LDS only
        const int4 rows = ssched[clampst(st)];
        active = st < nsteps && rows.x + lane < rows.y;
        return (rows.x + lane < rows.y) ? rows.x + lane : (rows.y > rows.x ? rows.y - 1 : 0);
    };
    auto stageW = [&](int st, StW& a) {
        bool act;
        const int rr = row_of(st, act);
        a.w = stWord[rr];
        a.koff = stKoff[rr];
        a.tab = stTable[((size_t)blockIdx.x + (size_t)clampst(st) * G) * 16 + (lane & 15)];
    };
    auto stageGs = [&](int st, StS& b) {
        const int4 rows = ssched[clampst(st)];
        b.k0e = rows.z & ~1;
        const int nb = rows.w - b.k0e;
        b.n = (rows.y > rows.x) ? nb * BB : 0;
        b.n2 = b.n >> 1;
        const int lim = (st < nsteps && b.n2 > 0) ? b.n2 - 1 : 0;
        const double2* __restrict__ s2 = reinterpret_cast<const double2*>(val + (size_t)b.k0e * BB);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = u * 64 + lane;
            b.tmp[u] = ld_stream(&s2[i < lim ? i : lim]);
        }
    };
    auto stageGm = [&](int st, const StW& a, StM& b) {
        bool act;
        const int rr = row_of(st, act);
        const int4 rows = ssched[clampst(st)];
        b.kb = rows.z + a.koff;
        int cnt = 0;
#pragma unroll
        for (int u = 0; u < PGCH; ++u) {
            const int nib = (int)((a.w >> (4 * u)) & 15u);
            const int off = __shfl(a.tab, nib, 64);
            const int cc = nib != 15 ? rr + off : rr;   // idle slots: the row itself (a valid row to gather)
            cnt += nib != 15;
            const double* xc = &x[(size_t)cc * BS];
            b.xx[u][0] = xc[0]; b.xx[u][1] = xc[1]; b.xx[u][2] = xc[2];
        }
        b.nrow = cnt;
        // the operands every row reads once per launch (the scalar products' second vectors, the backward sweep's row sums): nontemporal, so
        // that they do not push the gathered vector out of L2 - the rest product 0.0792 -> 0.0758 ms, +1.9 % Newton its/s in alternation
        // (profiles/r06_nt_operands_ab.txt); the result's stores stay plain (the vector kernel behind the product reads them at once)
#ifdef OPMHIP_PLAIN_OPERANDS
#define OPMHIP_LD1(p) (*(p))
#else
#define OPMHIP_LD1(p) __builtin_nontemporal_load(p)
#endif
        if constexpr (NDOT >= 1) {
            const double* wr = &w0[(size_t)rr * BS];
            b.ww[0] = OPMHIP_LD1(wr); b.ww[1] = OPMHIP_LD1(wr + 1); b.ww[2] = OPMHIP_LD1(wr + 2);
        }
        if constexpr (NDOT == 3) {
            const double* wr = &w1[(size_t)rr * BS];
            b.w2[0] = OPMHIP_LD1(wr); b.w2[1] = OPMHIP_LD1(wr + 1); b.w2[2] = OPMHIP_LD1(wr + 2);
        }
        if constexpr (UADD) {
            const double* ur = &uadd[(size_t)rr * BS];
            b.uu[0] = OPMHIP_LD1(ur); b.uu[1] = OPMHIP_LD1(ur + 1); b.uu[2] = OPMHIP_LD1(ur + 2);
        }
#undef OPMHIP_LD1
    };
    StW wq;
    StS sb;
    StM m;
    double sd0 = 0.0, sd1 = 0.0, sd2 = 0.0;
    {   // prologue, in the loop's order of issue
        StW w0q;
        stageW(0, w0q);
        stageGs(0, sb);
        stageGm(0, w0q, m);
        stageW(1, wq);
    }
    double2* d2 = reinterpret_cast<double2*>(sval);
    for (int st = 0; st < nsteps; ++st) {
        const int k0e = sb.k0e, n = sb.n, n2 = sb.n2;
        if (n2 > 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = u * 64 + lane;
                if (i < n2) d2[i] = sb.tmp[u];
            }
            if ((n & 1) && lane == 0) sval[n - 1] = val[(size_t)k0e * BB + n - 1];
        }
        wave_sync();
        stageGs(st + 1, sb);
        bool active;
        const int r = row_of(st, active);
        if (active) {
            const int kb = m.kb, nrow = m.nrow;
            double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll
            for (int u = 0; u < PGCH; ++u)
                if (u < nrow) blk_umv_lds(&sval[(kb + u - k0e) * BB], xs * m.xx[u][0], xs * m.xx[u][1], xs * m.xx[u][2], acc);
            if constexpr (UADD) { acc[0] += xs * m.uu[0]; acc[1] += xs * m.uu[1]; acc[2] += xs * m.uu[2]; }
            double* yr = &y[(size_t)r * BS];
            yr[0] = acc[0]; yr[1] = acc[1]; yr[2] = acc[2];
            if constexpr (NDOT >= 1) {
                double d0 = acc[0] * m.ww[0]; d0 += acc[1] * m.ww[1]; d0 += acc[2] * m.ww[2];
                sd0 += d0;
                if (NDOT >= 2) { double d1 = acc[0] * acc[0]; d1 += acc[1] * acc[1]; d1 += acc[2] * acc[2]; sd1 += d1; }
                if constexpr (NDOT == 3) { double d2 = acc[0] * m.w2[0]; d2 += acc[1] * m.w2[1]; d2 += acc[2] * m.w2[2]; sd2 += d2; }
            }
        }
        wave_sync();
        __builtin_amdgcn_sched_barrier(0);
        stageGm(st + 1, wq, m);
        __builtin_amdgcn_sched_barrier(0);
        stageW(st + 2, wq);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (NDOT >= 1) {
        sd0 = wave_sum(sd0);
        if (NDOT >= 2) sd1 = wave_sum(sd1);
        if (NDOT == 3) sd2 = wave_sum(sd2);
        if (lane == 0) { part[blockIdx.x] = sd0; if (NDOT >= 2) part[npart + blockIdx.x] = sd1; if (NDOT == 3) part[2 * npart + blockIdx.x] = sd2; }
    }
}

// ============================== ILU0 apply ===============================================================
// One colour of M^-1 = [w] U^-1 L^-1 (linalg/ParallelOverlappingILU0.hpp:848-903).  Work vector vu holds y = L^-1 d and then
// the unscaled U^-1 y; v receives the final (scaled) result.  Three shapes, chosen per colour by the launcher:
//   SW_L  : forward sweep of a middle colour      vu_i = d_i - sum_{j<i} L_ij y_j                      (:867-879)
//   SW_LF : last colour, forward + backward fused  (its rows have no U entries): vu_i = D_i^-1 (d_i - sum L_ij y_j)
//   SW_UF : backward sweep                         vu_i = D_i^-1 (y_i - sum_{j>i} U_ij vu_j)            (:881-895)
// y of the FIRST colour is d itself (no L entries there), so gathers of columns < n0 read d and the first colour's
// backward sweep takes its right-hand side from d: the forward "copy" launch disappears.
// relax_mode 0 (CPU path): backward sweep walks columns in DESCENDING order (the reference's reversed CRS), vu stays
//   unscaled, v_i = w * vu_i  (the reference's trailing "v *= w", :899-901, folded into the store);
// relax_mode 1 (OpenCL, bda/openclKernels.cpp:301-383): ascending columns, vu_i = v_i = w * D_i^-1 (...).
enum { SW_L = 0, SW_LF = 1, SW_UF = 2 };
// The backward sweeps' second result vector v (v != vu): w * vu_i - or, addp != NULL (CPR: v = (0, x_p, 0) + ILU0(d - A (0, x_p, 0)),
// twolevelmethodcpr.hh:476-498, w = 1), that product added to the block vector (0, addp_i, 0): the statement of the addition
// kernel that used to follow the sweeps (v_i += z_i), folded into the store
template <int K> __device__ __forceinline__ double second_result(const double* __restrict__ addp, int r, double wout) {
    if (!addp) return wout;
    return (K == 1 ? addp[r] : 0.0) + wout;
}
template <int SHAPE>
__global__ __launch_bounds__(64) void k_ilu_sweep(int tile_begin, int ntc, int n0, int rhs_from_d, const int* __restrict__ tile_row0,
                                                  const int* __restrict__ prow, const int* __restrict__ pcol,
                                                  const double* __restrict__ P, const double* __restrict__ invD,
                                                  const double* __restrict__ d, double* __restrict__ vu, double* __restrict__ v,
                                                  const double* __restrict__ addp, int relax_mode, double w, const double* __restrict__ done) {
    TILE_LDS
    const int lane = threadIdx.x, tl = xcd_tile(blockIdx.x, ntc);
    if (tl >= ntc || *done != 0.0) return;
    const int t = tile_begin + tl;
    const int r0 = tile_row0[t], r1 = tile_row0[t + 1];
    const int rq = (r0 + lane < r1) ? r0 + lane : r1 - 1;
    const double* rsrc = (SHAPE != SW_UF || rhs_from_d) ? d : vu;
    double rhs[3] = {rsrc[(size_t)rq * BS], rsrc[(size_t)rq * BS + 1], rsrc[(size_t)rq * BS + 2]};
    double Di[BB];
    if (SHAPE != SW_L) {
#pragma unroll
        for (int q = 0; q < BB; ++q) Di[q] = invD[(size_t)rq * BB + q];
    }
    TileCtx T;
    const bool reverse = (SHAPE == SW_UF) && relax_mode == 0;
    const int r = tile_row_product<true>(r0, r1, prow, pcol, P, vu, sval, lane, reverse, rhs, T, d, n0);
    if (r < 0) return;
    if (SHAPE == SW_L) {
        vu[(size_t)r * BS] = rhs[0]; vu[(size_t)r * BS + 1] = rhs[1]; vu[(size_t)r * BS + 2] = rhs[2];
        return;
    }
    double out[3] = {0.0, 0.0, 0.0};
    blk_umv(Di, rhs[0], rhs[1], rhs[2], out);  // DenseMatrix::mv: y = 0, then accumulate
    if (relax_mode == 1) { out[0] = w * out[0]; out[1] = w * out[1]; out[2] = w * out[2]; }
    vu[(size_t)r * BS] = out[0]; vu[(size_t)r * BS + 1] = out[1]; vu[(size_t)r * BS + 2] = out[2];
    if (v != vu) { v[(size_t)r * BS] = second_result<0>(addp, r, w * out[0]); v[(size_t)r * BS + 1] = second_result<1>(addp, r, w * out[1]); v[(size_t)r * BS + 2] = second_result<2>(addp, r, w * out[2]); }
}

// Line-coloured orderings: one workgroup walks the steps of its chain-tile in order (forward for L, backward for U);
// a row may depend on earlier rows of its own chain, which the same workgroup wrote one step before - visible after
// the workgroup barrier because a CU's L1 sees the CU's own write-through stores.  Chains of other colours are
// finished (earlier launches).  No first-colour / last-colour shortcuts here: every colour has chain-internal L and U.
// A load that must see what ANOTHER lane of this workgroup stored to global memory a step earlier (the rare paths of the
// chain sweeps): agent-scope atomic load, served by L2, so that no stale line of the CU's L1 can answer it.
__device__ __forceinline__ double coherent_load(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
constexpr int CHAIN_MAX_STEPS = 128;
#ifndef OPMHIP_CHAIN_GATHER
#define OPMHIP_CHAIN_GATHER 6
#endif
#ifndef OPMHIP_CHAIN_STAGE
#define OPMHIP_CHAIN_STAGE 12
#endif
constexpr int CGCH = OPMHIP_CHAIN_GATHER;   // blocks of a row whose vector entries are fetched one step ahead
constexpr int CHAIN_STAGE = OPMHIP_CHAIN_STAGE;
// Descriptor record of one launch position of a chain kernel (reorder.cpp: build_schedules), copied to LDS with one
// coalesced load: [0] steps (0 = padding) [1] chain-tile [4 ..] first row of step 0..steps, then the first L entry and the
// first U entry of those rows (S1 = longest chain-tile + 1 numbers each).  Everything the sweep needs to issue its first
// loads - the old chain  schedule -> chain-tile -> tile -> row -> entry  cost four dependent round trips per workgroup.
constexpr int DESC_HEAD = 4;
constexpr int DESC_MAX = DESC_HEAD + 3 * (CHAIN_MAX_STEPS + 1) + 3;
__device__ __forceinline__ int load_desc(const int* __restrict__ desc, int dstride, int lane, int* sdesc) {
    const int* rec = desc + (size_t)blockIdx.x * dstride;
    for (int i = lane; i < dstride; i += 64) sdesc[i] = rec[i];
    wave_sync();
    return sdesc[0];
}
// the stencil form of one factor part's index streams (Pattern::swL / swU, reorder.cpp: build_sweep_stencils): per tile a table of <= 15
// column offsets (col - row), per row a word of eight 4-bit table indices in ascending column order (15 = no entry) and a byte (first
// entry - the tile's first entry) - as for the SpMV (k_spmv_pipe_st)
struct SweepStencil {
    const unsigned* __restrict__ word;
    const unsigned char* __restrict__ koff;
    const int* __restrict__ table;   // [16 * number of tiles]
};
template <int SHAPE, bool ST = false, bool UA = false>  // SW_L or SW_UF; ST: column indices and row bounds from the stencil form (q0 = the chain-tile's first tile); UA (SW_UF): the row sums sum_j U_ij x_j are left in ua (Pattern::ualias)
__device__ __forceinline__ void chain_sweep(const int nsteps, const int lane, double* sval, const int* srow0, const int* sk0,
                                            const int* __restrict__ prow,
                                            const int* __restrict__ pcol, const double* __restrict__ P,
                                            const double* __restrict__ invD, const double* d,
                                            double* vu, double* v, const double* __restrict__ addp, int relax_mode, double w,
                                            const SweepStencil S = SweepStencil{nullptr, nullptr, nullptr}, const int q0 = 0, double* __restrict__ ua = nullptr) {
    // nsteps <= CHAIN_MAX_STEPS (checked on the host); srow0[0..nsteps], sk0[0..nsteps]: first row / first entry of every
    // step, out of the chain-tile's descriptor record (LDS) - no dependent loads here
    const bool reverse = (SHAPE == SW_UF) && relax_mode == 0;
    constexpr int U = CHAIN_STAGE;  // a step of a 2-colour line ordering streams <= 32 rows x 5 blocks = 11.25 KiB: 12 x 1 KiB per wavefront covers it in one batch
    // Software pipeline over the steps (st counts in processing order; the backward sweep walks tiles q1-1 ... q0).
    // The chain of dependent loads  row bounds -> column indices -> vector gathers  is spread over three iterations so
    // that no iteration waits for more than the loads issued one iteration earlier:
    //   A(st+3): row bounds      C(st+2): column indices      G(st+1): value stream, vector gathers, right-hand side, D^-1
    //   X(st)  : values -> LDS, products in the row's sequential order, store
    auto tile_of = [&](int st) { return (SHAPE == SW_UF) ? nsteps - 1 - st : st; };
    // What a step needs is loaded in four stages, each into registers of its own:
    //   A  row bounds (3 steps ahead)   C  column indices (2 ahead)   M  the small per-row items: right-hand side, vector
    //   entries of the other columns, D^-1, and the row bounds once more (1 ahead)   S  the value stream (1 ahead)
    // Every load targets registers whose previous content is dead when the load is issued, and NO loaded register is ever
    // copied into another one that lives across the loop: a copy of a register whose load is still in flight makes the
    // wavefront wait for it - and the newest load is the last to arrive, so that wait drains everything.  (A pipeline
    // written with "next" and "current" copies of one struct, or one that hands the row bounds on from stage to stage, ends
    // every step in s_waitcnt vmcnt(0): rocprofv3 showed 36 % of the wavefront cycles in s_waitcnt.)  Hence: a stage never
    // passes loaded values on; what a later stage needs again it loads again (the row bounds: a cache hit) or receives
    // as something COMPUTED from them when they were consumed (the masks `late` and `mine`).
    struct StA { int kb, ke; };
    struct StS {
        int k0e, n, n2;
        bool staged;
        double2 tmp[U];
    };
    struct StM {
        int kb, ke;             // loaded
        unsigned late;          // bit u set: column u lies inside this chain-tile (written by an earlier step of this sweep)
        unsigned mine;          // bit u set: column u is the row this lane finished one step earlier (its result is in registers)
        double xx[CGCH][3];     // vector entries of the other columns
        double rhs[3], Di[BB];
    };
    const int ctR0 = srow0[0], ctR1 = srow0[nsteps];  // rows of this chain-tile
    auto clampst = [&](int st) { return st < nsteps ? st : nsteps - 1; };
    auto row_of = [&](int st, bool& active) {   // the lane's row in step st (clamped to the step's last row when idle): LDS only
        const int ti = tile_of(st);
        const int r0 = srow0[ti], r1 = srow0[ti + 1];
        active = r0 + lane < r1;
        return active ? r0 + lane : r1 - 1;
    };
    auto stageA = [&](int st, StA& a) {
        bool act;
        const int rr = row_of(st, act);
        a.kb = prow[rr];
        a.ke = prow[rr + 1];
    };
    // (loads are issued unconditionally, idle slots and steps past the end reading some valid entry again: a load under a
    //  condition gives the two paths different numbers of outstanding loads, and every wait after their join then has to
    //  assume the smaller one - it waits for more than it needs, in the end for everything)
    auto stageC = [&](int st, const StA& a, int (&cc)[CGCH], unsigned& valid) {   // a = row bounds of step st
        bool act;
        (void)row_of(st, act);
        const int nrow = act ? a.ke - a.kb : 0;
        const int spare = a.kb > 0 ? a.kb - 1 : 0;   // a valid entry whatever the row looks like (the array holds >= 1)
        valid = 0u;
#pragma unroll
        for (int u = 0; u < CGCH; ++u) {
            const int k = reverse ? a.ke - 1 - u : a.kb + u;
            cc[u] = pcol[(u < nrow) ? k : spare];
            if (u < nrow) valid |= 1u << u;
        }
    };
    // ST: one stage in the place of A and C - the row's word and byte and (lanes 0..15) the tile's offset table, two steps ahead
    struct StW { unsigned wd; int koff, tab; };
    auto stageW = [&](int st, StW& a) {
        bool act;
        const int rr = row_of(st, act);
        a.wd = S.word[rr];
        a.koff = S.koff[rr];
        a.tab = S.table[(size_t)(q0 + tile_of(st)) * 16 + (lane & 15)];
    };
    // ... and what stage M makes of it: row bounds, column indices (the table entries travel lane to lane), the mask of used slots
    auto decodeW = [&](int st, const StW& a, int& kb, int& ke, int (&cc)[CGCH], unsigned& valid) {
        bool act;
        const int rr = row_of(st, act);
        const unsigned none = a.wd & (a.wd >> 1) & (a.wd >> 2) & (a.wd >> 3) & 0x11111111u;   // bit 4u set: slot u holds no entry
        const int cnt = none ? (__builtin_ctz(none) >> 2) : 8;
        kb = sk0[tile_of(st)] + a.koff;
        ke = kb + cnt;
        const int nrow = act ? cnt : 0;
        valid = 0u;
#pragma unroll
        for (int u = 0; u < CGCH; ++u) {
            const int e = reverse ? cnt - 1 - u : u;
            const int nib = (int)((a.wd >> (4 * (e & 7))) & 15u);
            const int off = __shfl(a.tab, nib, 64);
            cc[u] = (u < nrow) ? rr + off : rr;
            if (u < nrow) valid |= 1u << u;
        }
    };
    auto stageGs = [&](int st, StS& b) {   // the value stream of step st; st == nsteps: nothing but one line, read 12 times
        const int ti = tile_of(clampst(st));
        const int k0 = sk0[ti], k1 = sk0[ti + 1];
        b.k0e = k0 & ~1;
        const int nb = k1 - b.k0e;
        b.staged = nb <= TILE_CAP_BLOCKS + 1;
        b.n = nb * BB;
        b.n2 = b.n >> 1;
        const int lim = (st < nsteps && b.staged && b.n2 > 0) ? b.n2 - 1 : 0;   // (an unstaged or empty step: the same single line)
        const double2* __restrict__ s2 = reinterpret_cast<const double2*>(P + (size_t)b.k0e * BB);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = u * 64 + lane;
            b.tmp[u] = ld_stream(&s2[i < lim ? i : lim]);
        }
    };
    auto stageGm = [&](int st, const int (&cc)[CGCH], unsigned valid, int prevRow, StM& b, int skb = 0, int ske = 0) {   // cc = column indices of step st
        bool act;
        const int rr = row_of(st, act);
        if (ST) { b.kb = skb; b.ke = ske; }
        else {
            b.kb = prow[rr];
            b.ke = prow[rr + 1];
        }
        const double* rsrc = (SHAPE == SW_L) ? d : vu;  // the lane's own row: no earlier step of this sweep writes it
        b.rhs[0] = rsrc[(size_t)rr * BS]; b.rhs[1] = rsrc[(size_t)rr * BS + 1]; b.rhs[2] = rsrc[(size_t)rr * BS + 2];
        b.late = 0u; b.mine = 0u;
#pragma unroll
        for (int u = 0; u < CGCH; ++u) {
            const int cq = cc[u];
            const bool used = (valid >> u) & 1u;
            const bool inside = used && cq >= ctR0 && cq < ctR1;
            if (inside) b.late |= 1u << u;
            if (used && cq == prevRow) b.mine |= 1u << u;
            const double* xc = &vu[(size_t)((!used || inside) ? rr : cq) * BS];  // harmless own-row address when unused or read later
            b.xx[u][0] = xc[0]; b.xx[u][1] = xc[1]; b.xx[u][2] = xc[2];
        }
        // the masks exist HERE: left to itself the compiler forms them where they are used (the next step), which keeps the
        // column indices alive past the loads that refill their registers - and costs the copies this pipeline avoids
        asm volatile("" : "+v"(b.late), "+v"(b.mine));
        if (SHAPE != SW_L) {
#pragma unroll
            for (int q = 0; q < BB; ++q) b.Di[q] = invD[(size_t)rr * BB + q];
        }
    };
    StA a;
    StW wq;
    int cc[CGCH];
    unsigned ccValid;
    StS sb;
    StM m;
    int myPrevRow = -1;           // the row this lane finished in the previous step, and its result
    double myPrev[3] = {0.0, 0.0, 0.0};
    if (ST) {   // prologue of the stencil form, in the loop's order of issue
        StW w0q;
        int c0[CGCH], kb0, ke0;
        unsigned v0;
        stageW(0, w0q);
        stageGs(0, sb);
        decodeW(0, w0q, kb0, ke0, c0, v0);
        stageGm(0, c0, v0, -1, m, kb0, ke0);
        stageW(clampst(1), wq);
    } else
    {   // prologue: the row bounds of the first three steps in ONE round of loads, the column indices of the first two in
        // the next, then the first value stream and the first step's small items - three dependent rounds instead of six
        // - and in the loop's own order of issue (stream, small items, column indices, row bounds): the compiler's
        // bookkeeping of outstanding loads merges this path with the loop's back edge, and where the two disagree about
        // which load is older it has to assume the worse
        StA a0, a1;
        int c0[CGCH];
        unsigned v0;
        stageA(0, a0);
        stageA(clampst(1), a1);
        stageC(0, a0, c0, v0);
        stageGs(0, sb);
        stageGm(0, c0, v0, -1, m);
        stageC(clampst(1), a1, cc, ccValid);
        stageA(clampst(2), a);
    }
    for (int st = 0; st < nsteps; ++st) {
        // ---- X(st), part 1: commit the prefetched values to LDS
        const int k0e = sb.k0e, n = sb.n, n2 = sb.n2;
        const bool staged = __builtin_amdgcn_readfirstlane((int)sb.staged) != 0;   // the same in every lane (it comes from the descriptor record)
        double2* d2 = reinterpret_cast<double2*>(sval);
        if (staged && n2 > 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = u * 64 + lane;
                if (i < n2) d2[i] = sb.tmp[u];
            }
            const double2* __restrict__ s2 = reinterpret_cast<const double2*>(P + (size_t)k0e * BB);
            for (int base = 64 * U; base < n2; base += 64)  // steps larger than one batch (irregular rows): plain copy
                if (base + lane < n2) d2[base + lane] = s2[base + lane];
            if ((n & 1) && lane == 0) sval[n - 1] = P[(size_t)k0e * BB + n - 1];
        }
        wave_sync();
        // ---- the next step's value stream goes out now (its registers are free again) and flies during this step's work
        stageGs(st + 1, sb);
        // ---- X(st), part 2: the few columns inside this chain-tile are read now (the previous step wrote them; the
        //      lane's own previous row comes straight from registers), products in the row's sequential order, store
        bool active;
        const int rrow = row_of(st, active);
        const int r = active ? rrow : -1;
        if (r >= 0) {
            const int kb = m.kb, ke = m.ke;
            const int nrow = ke - kb;
            const unsigned used = (nrow >= CGCH) ? ((1u << CGCH) - 1u) : ((1u << nrow) - 1u);
            // a column inside this chain-tile that is NOT the lane's own previous row must come from memory: make sure
            // the previous step's stores have completed first (never taken on a line-coloured 7-point stencil)
            const bool fromMem = (m.late & ~m.mine & used) != 0u;
            if (__any(fromMem || nrow > CGCH)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            double rhs[3] = {m.rhs[0], m.rhs[1], m.rhs[2]};
            double us[3] = {0.0, 0.0, 0.0};
#pragma unroll
            for (int u = 0; u < CGCH; ++u) {
                if (u < nrow) {
                    double x0 = m.xx[u][0], x1 = m.xx[u][1], x2 = m.xx[u][2];
                    const int k = reverse ? ke - 1 - u : kb + u;
                    if ((m.late >> u) & 1u) {
                        if ((m.mine >> u) & 1u) { x0 = myPrev[0]; x1 = myPrev[1]; x2 = myPrev[2]; }
                        else { const double* xc = &vu[(size_t)pcol[k] * BS]; x0 = coherent_load(xc); x1 = coherent_load(xc + 1); x2 = coherent_load(xc + 2); }
                    }
                    // LDS or global memory by a (scalar) branch, never by one pointer: a pointer that may be either is "flat",
                    // and a flat load waits for every outstanding memory access - the next step's prefetch included
                    if (UA) {
                        if (staged) blk_mmv_lds_u(&sval[(k - k0e) * BB], x0, x1, x2, rhs, us);
                        else blk_mmv_u(&P[(size_t)k * BB], x0, x1, x2, rhs, us);
                    } else {
                        if (staged) blk_mmv_lds(&sval[(k - k0e) * BB], x0, x1, x2, rhs);
                        else blk_mmv(&P[(size_t)k * BB], x0, x1, x2, rhs);
                    }
                }
            }
            for (int done = CGCH; done < nrow; done += CGCH) {  // rows longer than one chunk: everything read now
                double yy[CGCH][3];
#pragma unroll
                for (int u = 0; u < CGCH; ++u) {
                    const int q = (done + u < nrow) ? done + u : nrow - 1;
                    const int cq = pcol[reverse ? ke - 1 - q : kb + q];
                    const double* xc = &vu[(size_t)cq * BS];  // may be a row an earlier step of this workgroup wrote
                    yy[u][0] = coherent_load(xc); yy[u][1] = coherent_load(xc + 1); yy[u][2] = coherent_load(xc + 2);
                }
#pragma unroll
                for (int u = 0; u < CGCH; ++u) {
                    if (done + u < nrow) {
                        const int k = reverse ? ke - 1 - (done + u) : kb + done + u;
                        if (UA) {
                            if (staged) blk_mmv_lds_u(&sval[(k - k0e) * BB], yy[u][0], yy[u][1], yy[u][2], rhs, us);
                            else blk_mmv_u(&P[(size_t)k * BB], yy[u][0], yy[u][1], yy[u][2], rhs, us);
                        } else {
                            if (staged) blk_mmv_lds(&sval[(k - k0e) * BB], yy[u][0], yy[u][1], yy[u][2], rhs);
                            else blk_mmv(&P[(size_t)k * BB], yy[u][0], yy[u][1], yy[u][2], rhs);
                        }
                    }
                }
            }
            if (SHAPE == SW_L) {
                vu[(size_t)r * BS] = rhs[0]; vu[(size_t)r * BS + 1] = rhs[1]; vu[(size_t)r * BS + 2] = rhs[2];
                myPrev[0] = rhs[0]; myPrev[1] = rhs[1]; myPrev[2] = rhs[2];
            } else {
                double out[3] = {0.0, 0.0, 0.0};
                blk_umv(m.Di, rhs[0], rhs[1], rhs[2], out);
                if (relax_mode == 1) { out[0] = w * out[0]; out[1] = w * out[1]; out[2] = w * out[2]; }
                vu[(size_t)r * BS] = out[0]; vu[(size_t)r * BS + 1] = out[1]; vu[(size_t)r * BS + 2] = out[2];
                if (v != vu) { v[(size_t)r * BS] = second_result<0>(addp, r, w * out[0]); v[(size_t)r * BS + 1] = second_result<1>(addp, r, w * out[1]); v[(size_t)r * BS + 2] = second_result<2>(addp, r, w * out[2]); }
                if (UA) { st_rowsum(&ua[(size_t)r * BS], us[0]); st_rowsum(&ua[(size_t)r * BS + 1], us[1]); st_rowsum(&ua[(size_t)r * BS + 2], us[2]); }
                myPrev[0] = out[0]; myPrev[1] = out[1]; myPrev[2] = out[2];
            }
            myPrevRow = r;
        }
        wave_sync();  // this step's results are visible to the next step; LDS image may be overwritten
        // ---- the next step's small items into the registers this step has just finished with, then the column indices
        //      and row bounds further ahead into theirs (unconditionally, past the end with the last step's indices again -
        //      a few loads that hit the cache: a condition here would make each of these registers a merge of "old" and
        //      "newly loaded", which the compiler resolves with copies at the end of the loop body)
        //      The scheduling barriers keep each stage's last use of its input registers in front of the loads that refill
        //      them - interleaved by the instruction scheduler, old and new values would overlap and need copies again.
        __builtin_amdgcn_sched_barrier(0);
        if (ST) {
            int skb, ske;
            decodeW(clampst(st + 1), wq, skb, ske, cc, ccValid);
            stageGm(clampst(st + 1), cc, ccValid, myPrevRow, m, skb, ske);
            __builtin_amdgcn_sched_barrier(0);
            stageW(clampst(st + 2), wq);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            stageGm(clampst(st + 1), cc, ccValid, myPrevRow, m);
            __builtin_amdgcn_sched_barrier(0);
            stageC(clampst(st + 2), a, cc, ccValid);
            __builtin_amdgcn_sched_barrier(0);
            stageA(clampst(st + 3), a);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
// Light sweep of a chain tile: every row's part of the factor is at most ONE block, the one towards the row the same
// lane handled one step earlier (set_pattern verified it: Pattern::lightL / lightU).  Then a lane's rows form a private
// recurrence - no LDS image, no workgroup-level ordering - and what limits the sweep is the latency of eight dependent
// steps, so each lane keeps LIGHT_DEPTH steps of loads (row bounds one step further) in flight.
#ifndef OPMHIP_LIGHT_DEPTH
#define OPMHIP_LIGHT_DEPTH 2   // one M^-1 on one box: depth 1 0.1488 ms, 2 0.1460, 3 0.1480, 4 0.1477, 6 0.1653 (more than 63 loads outstanding)
#endif
constexpr int LIGHT_DEPTH = OPMHIP_LIGHT_DEPTH;
template <int SHAPE, bool UA = false>  // SW_L or SW_UF; UA: as in chain_sweep
__device__ __forceinline__ void chain_sweep_light(const int nsteps, const int lane, const int* srow0,
                                                  const int* __restrict__ prow,
                                                  const int* __restrict__ pcol, const double* __restrict__ P,
                                                  const double* __restrict__ invD, const double* d, double* vu, double* v,
                                                  const double* __restrict__ addp, int relax_mode, double w, double* __restrict__ ua = nullptr) {
    // Written by the rules chain_sweep's comment lists: every load unconditional, into registers whose content is dead,
    // nothing loaded ever copied, and no load consumed while it is the newest one outstanding (waiting for the newest
    // load means waiting for all of them - and, the counter being in order, waiting for a load issued k steps ago means
    // waiting for everything older, so the prefetch is only as deep as its shortest dependency).  The row bounds of step
    // st + 2 D are therefore requested D steps before the stage that turns them into addresses, into one of D register
    // pairs; steps past the end repeat one address.
    constexpr int D = LIGHT_DEPTH;
    auto tile_of = [&](int st) { return (SHAPE == SW_UF) ? nsteps - 1 - st : st; };
    struct StA { int kb, ke; };
    struct StB {
        int kb, ke;       // loaded again with the rest (a cache hit): what a stage needs later it does not inherit as a copy
        int cq;
        double blk[BB], rhs[3], Di[BB];
    };
    auto row_of = [&](int st, bool& active) {   // LDS only; past the end: the tile's first row for every lane (one line)
        if (st >= nsteps) { active = false; return srow0[0]; }
        const int ti = tile_of(st);
        const int r0 = srow0[ti], r1 = srow0[ti + 1];
        active = r0 + lane < r1;
        return active ? r0 + lane : r1 - 1;
    };
    auto stageA = [&](int st, StA& a) {
        bool act;
        const int rr = row_of(st, act);
        a.kb = prow[rr];
        a.ke = prow[rr + 1];
    };
    auto stageB = [&](int st, const StA& a, StB& b) {
        bool act;
        const int rr = row_of(st, act);
        const bool has = act && a.ke > a.kb;
        const int k = has ? a.kb : (a.kb > 0 ? a.kb - 1 : 0);   // any valid entry: loaded, not used
        b.kb = prow[rr];
        b.ke = prow[rr + 1];
        b.cq = pcol[k];
#pragma unroll
        for (int q = 0; q < BB; ++q) b.blk[q] = P[(size_t)k * BB + q];
        const double* rsrc = (SHAPE == SW_L) ? d : vu;
        b.rhs[0] = rsrc[(size_t)rr * BS]; b.rhs[1] = rsrc[(size_t)rr * BS + 1]; b.rhs[2] = rsrc[(size_t)rr * BS + 2];
        if (SHAPE != SW_L) {
#pragma unroll
            for (int q = 0; q < BB; ++q) b.Di[q] = invD[(size_t)rr * BB + q];
        }
    };
    StA aa[D];
    StB b[D];
    {   // prologue: row bounds of the first D steps in one round, their items in the next, then the D pairs further ahead
        StA a0[D];
#pragma unroll
        for (int u = 0; u < D; ++u) stageA(u, a0[u]);
#pragma unroll
        for (int u = 0; u < D; ++u) stageB(u, a0[u], b[u]);
#pragma unroll
        for (int u = 0; u < D; ++u) stageA(D + u, aa[u]);
    }
    double prev[3] = {0.0, 0.0, 0.0};  // result of the row this lane finished one step earlier
    for (int s0 = 0; s0 < nsteps; s0 += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int st = s0 + u;
            StB& c = b[u];
            bool active;
            const int rrow = row_of(st, active);
            if (active) {
                const int r = rrow;
                double rhs[3] = {c.rhs[0], c.rhs[1], c.rhs[2]};
                double us[3] = {0.0, 0.0, 0.0};
                if (c.ke > c.kb) {
                    if (UA) blk_mmv_u(c.blk, prev[0], prev[1], prev[2], rhs, us);
                    else blk_mmv(c.blk, prev[0], prev[1], prev[2], rhs);
                }
                if (SHAPE == SW_L) {
                    vu[(size_t)r * BS] = rhs[0]; vu[(size_t)r * BS + 1] = rhs[1]; vu[(size_t)r * BS + 2] = rhs[2];
                    prev[0] = rhs[0]; prev[1] = rhs[1]; prev[2] = rhs[2];
                } else {
                    double out[3] = {0.0, 0.0, 0.0};
                    blk_umv(c.Di, rhs[0], rhs[1], rhs[2], out);
                    if (relax_mode == 1) { out[0] = w * out[0]; out[1] = w * out[1]; out[2] = w * out[2]; }
                    vu[(size_t)r * BS] = out[0]; vu[(size_t)r * BS + 1] = out[1]; vu[(size_t)r * BS + 2] = out[2];
                    if (v != vu) { v[(size_t)r * BS] = second_result<0>(addp, r, w * out[0]); v[(size_t)r * BS + 1] = second_result<1>(addp, r, w * out[1]); v[(size_t)r * BS + 2] = second_result<2>(addp, r, w * out[2]); }
                    if (UA) { st_rowsum(&ua[(size_t)r * BS], us[0]); st_rowsum(&ua[(size_t)r * BS + 1], us[1]); st_rowsum(&ua[(size_t)r * BS + 2], us[2]); }
                    prev[0] = out[0]; prev[1] = out[1]; prev[2] = out[2];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            stageB(st + D, aa[u], c);                // row bounds requested D steps ago
            __builtin_amdgcn_sched_barrier(0);
            stageA(st + 2 * D, aa[u]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
template <int SHAPE, bool UA = false>
__global__ __launch_bounds__(64) void k_ilu_sweep_light(const int* __restrict__ desc, int dstride, int S1,
                                                        const int* __restrict__ prow,
                                                        const int* __restrict__ pcol, const double* __restrict__ P,
                                                        const double* __restrict__ invD, const double* d,
                                                        double* vu, double* v, const double* __restrict__ addp, int relax_mode, double w, const double* __restrict__ done,
                                                        double* __restrict__ ua = nullptr) {
    __shared__ int sdesc[DESC_MAX];
    const int lane = threadIdx.x;
    const double stop = *done;   // read with the descriptor record, tested after it: one round trip instead of two
    const int nsteps = load_desc(desc, dstride, lane, sdesc);
    if (stop != 0.0 || nsteps <= 0) return;
    (void)S1;
    chain_sweep_light<SHAPE, UA>(nsteps, lane, sdesc + DESC_HEAD, prow, pcol, P, invD, d, vu, v, addp, relax_mode, w, ua);
}
template <int SHAPE, bool ST = false, bool UA = false>
__global__ __launch_bounds__(64) void k_ilu_sweep_chain(const int* __restrict__ desc, int dstride, int S1,
                                                        const int* __restrict__ prow,
                                                        const int* __restrict__ pcol, const double* __restrict__ P,
                                                        const double* __restrict__ invD, const double* d,
                                                        double* vu, double* v, const double* __restrict__ addp, int relax_mode, double w, const double* __restrict__ done,
                                                        const SweepStencil S, double* __restrict__ ua = nullptr) {
    TILE_LDS
    __shared__ int sdesc[DESC_MAX];
    const int lane = threadIdx.x;
    const double stop = *done;   // read with the descriptor record, tested after it: one round trip instead of two
    const int nsteps = load_desc(desc, dstride, lane, sdesc);
    if (stop != 0.0 || nsteps <= 0) return;
    const int* srow0 = sdesc + DESC_HEAD;
    chain_sweep<SHAPE, ST, UA>(nsteps, lane, sval, srow0, srow0 + (SHAPE == SW_L ? 1 : 2) * S1, prow, pcol, P, invD, d, vu, v, addp, relax_mode, w, S, sdesc[2], ua);
}
// Last colour: its rows have U entries only inside their own chain-tile, so the backward sweep of a chain-tile can
// start the moment its forward sweep ends - one launch instead of two, and y never leaves the cache in between.
template <bool LIGHT_U, bool ST = false, bool UA = false>
__global__ __launch_bounds__(64) void k_ilu_sweep_chain_LU(const int* __restrict__ desc, int dstride, int S1,
                                                           const int* __restrict__ lrow,
                                                           const int* __restrict__ lcol, const double* __restrict__ L,
                                                           const int* __restrict__ urow, const int* __restrict__ ucol,
                                                           const double* __restrict__ Uv, const double* __restrict__ invD,
                                                           const double* d, double* vu, double* v, const double* __restrict__ addp, int relax_mode, double w,
                                                           const double* __restrict__ done, const SweepStencil SL, const SweepStencil SU, double* __restrict__ ua = nullptr) {
    TILE_LDS
    __shared__ int sdesc[DESC_MAX];
    const int lane = threadIdx.x;
    const double stop = *done;   // read with the descriptor record, tested after it: one round trip instead of two
    const int nsteps = load_desc(desc, dstride, lane, sdesc);
    if (stop != 0.0 || nsteps <= 0) return;
    const int* srow0 = sdesc + DESC_HEAD;
    chain_sweep<SW_L, ST>(nsteps, lane, sval, srow0, srow0 + S1, lrow, lcol, L, invD, d, vu, v, addp, relax_mode, w, SL, sdesc[2]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // y of this chain-tile is written before the backward sweep reads it
    wave_sync();
    if (LIGHT_U) chain_sweep_light<SW_UF, UA>(nsteps, lane, srow0, urow, ucol, Uv, invD, d, vu, v, addp, relax_mode, w, ua);
    else chain_sweep<SW_UF, ST, UA>(nsteps, lane, sval, srow0, srow0 + 2 * S1, urow, ucol, Uv, invD, d, vu, v, addp, relax_mode, w, SU, sdesc[2], ua);
}

// ============================== ILU0 factorisation =======================================================
// One colour of the left-looking block ILU0 with stored inverse (detail::ghost_last_bilu0_decomposition,
// linalg/ParallelOverlappingILU0.hpp:439-494): row i of A is staged in LDS and eliminated there against the
// already finished rows j < i (their U part and D_j^-1 live in HBM, written by earlier colours), then split into
// L, U and D^-1.
// RIDER (CPR, FactorRider): the staged, fixed-up row also leaves its pressure-column image and its entries of the pressure matrix behind
// - and with RIDER == 2 its quasi-IMPES weights - before the elimination touches it: the pass k_cpr_pvals / k_cpr_weights made over the
// whole Jacobian right after this kernel (1 142 MB of traffic, 0.20 ms per solve) rides here on values that are in LDS anyway.
template <int RIDER>
__global__ __launch_bounds__(64) void k_ilu_factor(const int* __restrict__ sched, const int* __restrict__ ct_first, const int* __restrict__ tile_row0,
                                                   const int* __restrict__ rowptr, const int* __restrict__ col,
                                                   const int* __restrict__ diag, const double* __restrict__ A,
                                                   const int* __restrict__ fdest, const int* __restrict__ lmatch, const int* __restrict__ urowptr,
                                                   const int* __restrict__ ucol, double* L, double* U, double* invD, double* Afix,
                                                   int Nb, int ellW, int ghostFrom, double* rw, double* __restrict__ rap, double* __restrict__ rpcol,
                                                   const int* __restrict__ rdest, const int* __restrict__ rrowptr, double* __restrict__ Rv) {
    TILE_LDS
    const int lane = threadIdx.x;
    // launch position -> chain-tile by the colour's XCD-aware schedule (reorder.cpp: build_schedules): the chain-tiles of one stretch of
    // the grid share an L2, so the D^-1 and U blocks that four neighbour rows of different workgroups gather are fetched once
    const int ct = sched[blockIdx.x];
    if (ct < 0) return;
    const int q0 = ct_first[ct], q1 = ct_first[ct + 1];
    constexpr int FU = 8;  // U-row columns of a neighbour fetched in one batch (longer rows: the merge below)
    __shared__ int scol[TILE_CAP_BLOCKS + 2], sdest[TILE_CAP_BLOCKS + 2], slm[TILE_CAP_BLOCKS + 2];
    __shared__ short ssrc[TILE_CAP_BLOCKS + 2];   // Rv != NULL: per block of the tile's stretch of the rest stream, its place in the staged tile
    for (int t = q0; t < q1; ++t) {  // steps of a chain-tile in order (a single step unless the ordering is line-coloured)
        const TileCtx T = tile_stage_values(t, tile_row0, rowptr, A, sval, lane);
        // the tile's column indices next to its values: the elimination searches them many times
        if (T.staged) {
            const int kk0 = rowptr[T.r0], kk1 = rowptr[T.r1];
            const int rb0 = Rv ? rrowptr[T.r0] : 0;
            for (int q = kk0 + lane; q < kk1; q += 64) {
                scol[q - T.k0e] = col[q]; sdest[q - T.k0e] = fdest[q]; slm[q - T.k0e] = lmatch[q];
                if (Rv) { const int rd = rdest[q]; if (rd >= 0) ssrc[rd - rb0] = (short)(q - T.k0e); }
            }
            if (kk0 > T.k0e && lane == 0) sdest[0] = -1;   // the alignment block in front of the tile belongs to the previous row
        }
        wave_sync();
        const int i = T.r0 + lane;
        if (Afix && i < T.r1) {   // the zero-diagonal fix of the device-assembled Jacobian (k_zero_diag_fix's statement), on the staged row and - rarely - in the matrix itself, which the operator reads later
            const int kd = diag[i];
            double* dblk = &sval[(kd - T.k0e) * BB];
#pragma unroll
            for (int dgn = 0; dgn < BS; ++dgn)
                if (dblk[dgn * 4] == 0.0) { dblk[dgn * 4] = 1e-15; Afix[(size_t)kd * BB + dgn * 4] = 1e-15; }
        }
        if (Rv && T.staged) {
            // Pattern::ualias: the matrix's own entries beside the U part - lower entries, diagonal, ghost columns - leave LDS BEFORE the
            // elimination touches them, as one stream the product after an M^-1 application reads in place of the whole matrix
            // (rows are stored one after the other in the rest stream as in the matrix: the tile's stretch of it is ONE range, written
            //  output-major - neighbouring lanes, neighbouring doubles, 16 bytes per lane where the range's alignment allows)
            wave_sync();
            const int rb = rrowptr[T.r0], nout = (rrowptr[T.r1] - rb) * BB;
            double* out = Rv + (size_t)rb * BB;
            const int h = (rb * BB) & 1;   // the range starts on an odd double: one leading single
            auto src = [&](int o) { const int ob = o / BB; return sval[(int)ssrc[ob] * BB + (o - ob * BB)]; };
            if (h && lane == 0 && nout > 0) out[0] = src(0);
            const int np2 = (nout - h) >> 1;
            double2* out2 = reinterpret_cast<double2*>(out + h);
            for (int p2 = lane; p2 < np2; p2 += 64) {   // nontemporal: read once, by the products of the solve (0.473 -> 0.460 ms per factorisation)
                v2d_t t;
                t.x = src(h + 2 * p2); t.y = src(h + 2 * p2 + 1);
                __builtin_nontemporal_store(t, reinterpret_cast<v2d_t*>(&out2[p2]));
            }
            if (((nout - h) & 1) && lane == 0) out[nout - 1] = src(nout - 1);
            wave_sync();
        }
        if (i < T.r1) {
            const int kb = rowptr[i], ke = rowptr[i + 1], kd = diag[i];
            // an over-long row (not staged) cannot be eliminated in LDS: such rows are rejected at set_pattern time
            double* row = &sval[(kb - T.k0e) * BB];
            const int* rcol = &scol[kb - T.k0e];
            const int* rlm = &slm[kb - T.k0e];
            const int n = ke - kb, nd = kd - kb;
            if (RIDER) {
                constexpr int PCOL = 1;   // pressure index inside a block (BlackOilIndices::pressureSwitchIdx; cpr.hip: CPR_P)
                double w0, w1, w2;
                if (RIDER == 2) {   // w_i = D_ii^-T e_p / max|.| (getQuasiImpesWeights.hpp:46-85): k_cpr_weights' statements on the staged diagonal block
                    double Dt[BB], inv[BB];
#pragma unroll
                    for (int r = 0; r < BS; ++r)
#pragma unroll
                        for (int cc = 0; cc < BS; ++cc) Dt[r * BS + cc] = row[nd * BB + cc * BS + r];
                    blk_invert(Dt, inv);
                    const double b0 = inv[0 * BS + PCOL], b1 = inv[1 * BS + PCOL], b2 = inv[2 * BS + PCOL];
                    double mx = 0.0;
                    mx = fmax(mx, fabs(b0)); mx = fmax(mx, fabs(b1)); mx = fmax(mx, fabs(b2));
                    w0 = b0 / mx; w1 = b1 / mx; w2 = b2 / mx;
                    rw[(size_t)i * BS] = w0; rw[(size_t)i * BS + 1] = w1; rw[(size_t)i * BS + 2] = w2;
                } else {
                    w0 = rw[(size_t)i * BS]; w1 = rw[(size_t)i * BS + 1]; w2 = rw[(size_t)i * BS + 2];
                }
                const size_t plane = (size_t)ellW * Nb;
                for (int a = 0; a < n; ++a) {   // k_cpr_pvals' statements: a_p = sum_r A[r][p] w[r], r ascending; a ghost coupling of a subdomain's own system: 0
                    const bool ghost = rcol[a] >= ghostFrom;
                    const double b0 = ghost ? 0.0 : row[a * BB + 0 * BS + PCOL], b1 = ghost ? 0.0 : row[a * BB + 1 * BS + PCOL], b2 = ghost ? 0.0 : row[a * BB + 2 * BS + PCOL];
                    double sp = 0.0;
                    sp += b0 * w0; sp += b1 * w1; sp += b2 * w2;
                    const size_t e = (size_t)a * Nb + i;
                    rap[e] = sp;
                    rpcol[e] = b0; rpcol[plane + e] = b1; rpcol[2 * plane + e] = b2;
                }
            }
            bool allfast = true;
            for (int a = 0; a < nd; ++a) allfast = allfast && rlm[a] != -2;
            // Every step of this row is a looked-up one (the rule on grids without triangles): the loads of step a + 1 - its D_j^-1
            // and its U block, addresses known from LDS - are issued before step a is worked off, two register sets taking turns.
            // The steps themselves stay in order: a step's L block may have been touched by an earlier one.
#define OPMHIP_FACTOR_LOAD(D_, U_, a_)                                                                  \
    do {                                                                                                \
        const int j_ = rcol[a_], lm_ = rlm[a_];                                                         \
        const size_t ub_ = lm_ >= 0 ? (size_t)(lm_ >> 6) * BB : 0;                                     \
        _Pragma("unroll") for (int q = 0; q < BB; ++q) { D_[q] = invD[(size_t)j_ * BB + q]; U_[q] = U[ub_ + q]; } \
    } while (0)
#define OPMHIP_FACTOR_STEP(D_, U_, a_)                                                                  \
    do {                                                                                                \
        const int lm_ = rlm[a_];                                                                        \
        double tmp_[BB], Lij_[BB];                                                                      \
        _Pragma("unroll") for (int q = 0; q < BB; ++q) tmp_[q] = row[(a_) * BB + q];                    \
        blk_mul(tmp_, D_, Lij_);                                                                        \
        _Pragma("unroll") for (int q = 0; q < BB; ++q) row[(a_) * BB + q] = Lij_[q];                    \
        if (lm_ >= 0) {                                                                                 \
            double Pm_[BB];                                                                             \
            blk_mul(Lij_, U_, Pm_);                                                                     \
            double* tgt_ = &row[(lm_ & 63) * BB];                                                       \
            _Pragma("unroll") for (int q = 0; q < BB; ++q) tgt_[q] -= Pm_[q];                           \
        }                                                                                               \
    } while (0)
            if (allfast && nd > 0) {
                double D0[BB], U0[BB], D1[BB], U1[BB];
                OPMHIP_FACTOR_LOAD(D0, U0, 0);
                for (int a = 0; a < nd; a += 2) {
                    if (a + 1 < nd) OPMHIP_FACTOR_LOAD(D1, U1, a + 1);
                    OPMHIP_FACTOR_STEP(D0, U0, a);
                    if (a + 1 < nd) {
                        if (a + 2 < nd) OPMHIP_FACTOR_LOAD(D0, U0, a + 2);
                        OPMHIP_FACTOR_STEP(D1, U1, a + 1);
                    }
                }
            }
#undef OPMHIP_FACTOR_LOAD
#undef OPMHIP_FACTOR_STEP
            for (int a = allfast ? nd : 0; a < nd; ++a) {
                const int j = rcol[a];
                const int lm = rlm[a];
                if (lm != -2) {
                    // the step's one update is known (Pattern::lmatch): D_j^-1 and the U block it needs come in ONE round of loads
                    // (no extent of row j's U part, no column list, no search) - the same products, the same subtraction
                    const bool has = lm >= 0;
                    const size_t ub = has ? (size_t)(lm >> 6) * BB : 0;
                    double Lij[BB], Dj[BB], tmp[BB], Ujk[BB];
#pragma unroll
                    for (int q = 0; q < BB; ++q) { tmp[q] = row[a * BB + q]; Dj[q] = invD[(size_t)j * BB + q]; Ujk[q] = U[ub + q]; }
                    blk_mul(tmp, Dj, Lij);  // A_ij * A_jj^-1
#pragma unroll
                    for (int q = 0; q < BB; ++q) row[a * BB + q] = Lij[q];
                    if (has) {
                        double Pm[BB];
                        blk_mul(Lij, Ujk, Pm);  // L_ij * A_jk
                        double* tgt = &row[(lm & 63) * BB];
#pragma unroll
                        for (int q = 0; q < BB; ++q) tgt[q] -= Pm[q];
                    }
                    continue;
                }
                double Lij[BB], Dj[BB], tmp[BB];
                // one round of loads: D_j^-1, the extent of row j's U part and (next) its first FU column indices -
                // instead of walking them one dependent load at a time
                const int jk0 = urowptr[j], jend = urowptr[j + 1];
#pragma unroll
                for (int q = 0; q < BB; ++q) { tmp[q] = row[a * BB + q]; Dj[q] = invD[(size_t)j * BB + q]; }
                int ucj[FU];
#pragma unroll
                for (int u = 0; u < FU; ++u) ucj[u] = (jk0 + u < jend) ? ucol[jk0 + u] : -1;
                blk_mul(tmp, Dj, Lij);  // A_ij * A_jj^-1
#pragma unroll
                for (int q = 0; q < BB; ++q) row[a * BB + q] = Lij[q];
                if (jend - jk0 <= FU) {
                    // every common column of (row i beyond a) and (U row of j); each match updates its own block of
                    // row i, so the order among matches does not matter
#pragma unroll
                    for (int u = 0; u < FU; ++u) {
                        const int cj = ucj[u];
                        if (cj < 0) continue;
                        for (int ik = a + 1; ik < n; ++ik) {
                            if (rcol[ik] == cj) {
                                double Ujk[BB], P[BB];
#pragma unroll
                                for (int q = 0; q < BB; ++q) Ujk[q] = U[(size_t)(jk0 + u) * BB + q];
                                blk_mul(Lij, Ujk, P);  // L_ij * A_jk
#pragma unroll
                                for (int q = 0; q < BB; ++q) row[ik * BB + q] -= P[q];
                                break;
                            }
                        }
                    }
                    continue;
                }
                int jk = jk0;
                int ik = a + 1;
                while (ik < n && jk < jend) {
                    const int ci = rcol[ik], cj = ucol[jk];
                    if (ci == cj) {
                        double Ujk[BB], P[BB];
#pragma unroll
                        for (int q = 0; q < BB; ++q) Ujk[q] = U[(size_t)jk * BB + q];
                        blk_mul(Lij, Ujk, P);  // L_ij * A_jk
#pragma unroll
                        for (int q = 0; q < BB; ++q) row[ik * BB + q] -= P[q];
                        ++ik; ++jk;
                    } else if (ci < cj) ++ik;
                    else ++jk;
                }
            }
            double dblk[BB], inv[BB];
#pragma unroll
            for (int q = 0; q < BB; ++q) dblk[q] = row[nd * BB + q];
            blk_invert(dblk, inv);
#pragma unroll
            for (int q = 0; q < BB; ++q) invD[(size_t)i * BB + q] = inv[q];
        }
        wave_sync();
        // The finished rows leave LDS together: every entry knows its place in L or U (Pattern::fdest), neighbouring lanes
        // write neighbouring doubles.  (One lane per row storing its own 6 + 6 blocks cost a third of the kernel: 63 store
        // instructions per lane, each scattering 8 bytes into 32 different rows.)
        if (T.staged) {
            const int n9 = T.nb * BB;
            for (int e = lane; e < n9; e += 64) {
                const int blk = e / BB, dst = sdest[blk];
                if (dst >= 0) L[(size_t)dst * BB + (e - blk * BB)] = sval[e];
                else if (dst <= -2) U[(size_t)(-2 - dst) * BB + (e - blk * BB)] = sval[e];
            }
        }
        __syncthreads();  // drains vmcnt: this step's factors (global stores) are complete before the next step reads them (measured: without it, wrong factors and 0.333 against 0.325 ms - the drain costs nothing)
    }
}

// ============================== standard wells ===========================================================
// y -= C^T (D^-1 (B x)) per well (bda/WellContributions.cu:36-126); one wavefront per well, any number of
// perforations (the CUDA kernel's 32-lane masks assume <= 2 blocks per warp pass, :82).
// (B x)_w added to (SUB: subtracted from) s on lanes 0..3, one well equation each, in the order of the CPU's per-perforation loop
// (wells/StandardWell_impl.hpp:1254-1275).  Walking the perforations one by one on four lanes costs a round trip to memory per
// perforation (100 completions: ~50 us per application, twice per BiCGStab iteration); here every lane forms the twelve products of one
// perforation - 64 perforations' loads in flight together - and lanes 0..3 then add the products up out of LDS in that same order: the
// same products, the same additions, the same bits.
template <bool SUB>
__device__ __forceinline__ double well_Bx(double s, int pb, int pe, const int* __restrict__ Bcols, const double* __restrict__ B,
                                          const double* __restrict__ x, double xs, int lane, double* prod /* LDS, 64 x 12 */) {
    for (int p0 = pb; p0 < pe; p0 += 64) {
        const int p = p0 + lane;
        if (p < pe) {
            const double* xb = &x[(size_t)Bcols[p] * 3];
            const double x0 = xs * xb[0], x1 = xs * xb[1], x2 = xs * xb[2];
            const double* Bp = &B[(size_t)p * 12];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                prod[lane * 12 + q * 3] = Bp[q * 3] * x0;
                prod[lane * 12 + q * 3 + 1] = Bp[q * 3 + 1] * x1;
                prod[lane * 12 + q * 3 + 2] = Bp[q * 3 + 2] * x2;
            }
        }
        __syncthreads();
        if (lane < 4) {
            const int m = pe - p0 < 64 ? pe - p0 : 64;
            for (int j = 0; j < m; ++j) {
                const double* pj = &prod[j * 12 + lane * 3];
                if (SUB) { s -= pj[0]; s -= pj[1]; s -= pj[2]; }
                else { s += pj[0]; s += pj[1]; s += pj[2]; }
            }
        }
        __syncthreads();
    }
    return s;
}
__global__ __launch_bounds__(64) void k_wells_apply(const int* __restrict__ vp, const int* __restrict__ Ccols,
                                                    const int* __restrict__ Bcols, const double* __restrict__ C,
                                                    const double* __restrict__ D, const double* __restrict__ B,
                                                    const double* __restrict__ x, double* __restrict__ y, double xs) {
    __shared__ double z1[4], z2[4], prod[64 * 12];
    const int w = blockIdx.x, lane = threadIdx.x;
    const int pb = vp[w], pe = vp[w + 1];
    // z1 = B x : lanes 0..3 own one well equation each
    const double bx = well_Bx<false>(0.0, pb, pe, Bcols, B, x, xs, lane, prod);
    if (lane < 4) z1[lane] = bx;
    __syncthreads();
    if (lane < 4) {
        double s = 0.0;
        for (int q = 0; q < 4; ++q) s += D[(size_t)w * 16 + lane * 4 + q] * z1[q];
        z2[lane] = s;
    }
    __syncthreads();
    for (int e = pb * 3 + lane; e < pe * 3; e += 64) {
        const int p = e / 3, c = e % 3;
        double s = 0.0;
        for (int j = 0; j < 4; ++j) s += C[(size_t)p * 12 + j * 3 + c] * z2[j];
        // wells do not normally share a cell; if two do, their workgroups meet here: add atomically (one add per
        // well and entry - the same bits as a plain update whenever the cell has a single well)
        atomicAdd(&y[(size_t)Ccols[p] * 3 + c], -s);
    }
}

// this rank's part of B x of every well, 4 doubles per well, for the sum over the ranks (distributed wells); the sums of k_wells_apply
__global__ __launch_bounds__(64) void k_wells_bx(const int* __restrict__ vp, const int* __restrict__ Bcols, const double* __restrict__ B,
                                                 const double* __restrict__ x, double xs, double* __restrict__ bx) {
    __shared__ double prod[64 * 12];
    const int w = blockIdx.x, lane = threadIdx.x;
    const double s = well_Bx<false>(0.0, vp[w], vp[w + 1], Bcols, B, x, xs, lane, prod);
    if (lane < 4) bx[(size_t)w * 4 + lane] = s;
}

// r -= C^T (D^-1 resWell) (StandardWell::apply(BVector& r), wells/StandardWell_impl.hpp:1283-1296) and
// xw = D^-1 (resWell - B x) (recoverSolutionWell, :1298-1311); one wavefront per well, sums in the CPU's order
__global__ __launch_bounds__(64) void k_wells_residual(const int* __restrict__ vp, const int* __restrict__ Ccols,
                                                       const double* __restrict__ C, const double* __restrict__ D,
                                                       const double* __restrict__ resWell, double* __restrict__ r) {
    __shared__ double z2[4];
    const int w = blockIdx.x, lane = threadIdx.x;
    const int pb = vp[w], pe = vp[w + 1];
    if (lane < 4) {
        double s = 0.0;
        for (int q = 0; q < 4; ++q) s += D[(size_t)w * 16 + lane * 4 + q] * resWell[(size_t)w * 4 + q];
        z2[lane] = s;
    }
    __syncthreads();
    for (int e = pb * 3 + lane; e < pe * 3; e += 64) {
        const int p = e / 3, c = e % 3;
        double s = 0.0;
        for (int j = 0; j < 4; ++j) s += C[(size_t)p * 12 + j * 3 + c] * z2[j];
        atomicAdd(&r[(size_t)Ccols[p] * 3 + c], -s);  // see k_wells_apply
    }
}
__global__ __launch_bounds__(64) void k_wells_recover(const int* __restrict__ vp, const int* __restrict__ Bcols,
                                                      const double* __restrict__ B, const double* __restrict__ D,
                                                      const double* __restrict__ resWell, const double* __restrict__ x,
                                                      const double* __restrict__ bxAll, double* __restrict__ xw) {
    __shared__ double z1[4];
    __shared__ double prod[64 * 12];
    const int w = blockIdx.x, lane = threadIdx.x;
    const int pb = vp[w], pe = vp[w + 1];
    double s = lane < 4 ? resWell[(size_t)w * 4 + lane] : 0.0;   // resWell -= B x, perforation by perforation (BCRSMatrix::mmv)
    if (bxAll) { if (lane < 4) s -= bxAll[(size_t)w * 4 + lane]; }   // a well shared by several ranks: the product summed over them, subtracted as a whole
    else s = well_Bx<true>(s, pb, pe, Bcols, B, x, 1.0, lane, prod);
    if (lane < 4) z1[lane] = s;
    __syncthreads();
    if (lane < 4) {
        double s = 0.0;
        for (int q = 0; q < 4; ++q) s += D[(size_t)w * 16 + lane * 4 + q] * z1[q];
        xw[(size_t)w * 4 + lane] = s;
    }
}

// A(Ccols[c], Bcols[b]) += -C_c^T (D^-1 B_b) for every perforation pair (c, b) of well w0 + blockIdx.x
// (StandardWell::addWellContributions, wells/StandardWell_impl.hpp:1688-1712; sums in the order of Detail::multMatrix
// and Detail::negativeMultMatrixTransposed, linalg/MatrixBlock.hpp:496-570).  entry[] holds, per pair, the position of
// the block in the device's block-CSR (found on the host).
// serial != 0: the well has two perforations in one cell, i.e. several of its pairs hit the same block - one lane then
// adds all pairs in the perforation order of the CPU loop (no race, same bits).
__global__ __launch_bounds__(64) void k_wells_add_to_matrix(int w0, int serial, const int* __restrict__ vp, const double* __restrict__ C,
                                                            const double* __restrict__ D, const double* __restrict__ B,
                                                            const int* __restrict__ pair_ptr, const int* __restrict__ entry,
                                                            double* __restrict__ A) {
    const int w = w0 + blockIdx.x;
    const int pb = vp[w], np = vp[w + 1] - pb;
    if (serial && threadIdx.x != 0) return;
    for (int q = serial ? 0 : (int)threadIdx.x; q < np * np; q += serial ? 1 : 64) {
        const int c = q / np, b = q % np;
        const double* Bb = &B[(size_t)(pb + b) * 12];
        const double* Cc = &C[(size_t)(pb + c) * 12];
        const double* Dw = &D[(size_t)w * 16];
        double tmp[4][3];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 3; ++j) {
                double s = 0.0;
                for (int k = 0; k < 4; ++k) s += Dw[i * 4 + k] * Bb[k * 3 + j];
                tmp[i][j] = s;
            }
        double* blk = &A[(size_t)entry[pair_ptr[w] + q] * BB];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double s = 0.0;
                for (int k = 0; k < 4; ++k) s += Cc[k * 3 + i] * tmp[k][j];
                blk[i * 3 + j] += -s;
            }
    }
}

// ============================== BiCGStab vector kernels ==================================================
constexpr int VB = 256;          // threads per block
constexpr int VPT = 8;           // doubles per thread
__device__ __forceinline__ void block_partials(double a, double b, double* part, int npart, int nsum) {
    __shared__ double sh[2][VB / 64];
    a = wave_sum(a);
    if (nsum > 1) b = wave_sum(b);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { sh[0][wv] = a; sh[1][wv] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = sh[0][0];
        for (int i = 1; i < VB / 64; ++i) s += sh[0][i];
        part[blockIdx.x] = s;
        if (nsum > 1) {
            double u = sh[1][0];
            for (int i = 1; i < VB / 64; ++i) u += sh[1][i];
            part[npart + blockIdx.x] = u;
        }
    }
}
// Vectors a BiCGStab iteration is done with once a vector kernel has read them (v after the p- and r-updates; x, M^-1 p, M^-1 s and t in
// k_bicg_upd2) are read - and x written - nontemporally: they then do not take the place of the vectors the next kernels come back for
// (p, r, s: the sweeps' right-hand sides) in L2 and the Infinity Cache.  One M^-1 0.145 -> 0.135 ms, +3 % Newton its/s in alternation
// (profiles/r06_nt_operands_ab.txt).
#ifdef OPMHIP_PLAIN_VEC
#define OPMHIP_VLD(p) (*(p))
#define OPMHIP_VST(v, p) (*(p) = (v))
#else
#define OPMHIP_VLD(p) __builtin_nontemporal_load(p)
#define OPMHIP_VST(v, p) __builtin_nontemporal_store(v, p)
#endif
// (measured and not kept: r in the p-update likewise, the factorisation's L stores likewise - nothing either way)
// r = rw = p = b, x = 0, partial b.b
__global__ __launch_bounds__(VB) void k_bicg_init(int n, const double* __restrict__ b, double* __restrict__ r,
                                                  double* __restrict__ rw, double* __restrict__ p, double* __restrict__ x,
                                                  double* __restrict__ v, double* __restrict__ part, int npart) {
    double s = 0.0;
    const int base = blockIdx.x * VB * VPT + threadIdx.x;
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
        const int e = base + u * VB;
        if (e < n) {
            const double be = b[e];
            r[e] = be; rw[e] = be; p[e] = be; x[e] = 0.0; v[e] = 0.0;
            s += be * be;
        }
    }
    block_partials(s, 0.0, part, npart, 1);
}
// p = (p - omega v) beta + r     (bda/openclKernels.cpp:130-153 "custom")
__global__ __launch_bounds__(VB) void k_bicg_pupdate(int n, const double* __restrict__ scal, double* __restrict__ p,
                                                     const double* __restrict__ v, const double* __restrict__ r) {
    if (scal[SC_DONE] != 0.0) return;
    const double omega = scal[SC_OMEGA], beta = scal[SC_BETA];
    const int base = blockIdx.x * VB * VPT + threadIdx.x;
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
        const int e = base + u * VB;
        if (e < n) p[e] = (p[e] - omega * OPMHIP_VLD(&v[e])) * beta + r[e];
    }
}
// r -= alpha v ; partial r.r.  The first half's "x += alpha pw" (bda/cusparseSolverBackend.cu:110) waits for the second
// half's update of x (k_bicg_upd2 adds both terms, in the reference's order: two passes over x less per iteration); a solve
// that meets the stopping rule right after a first half gets it from k_bicg_xhalf.
// SR (opmhip_config.fused_reductions): the norm of the new residual was formed from the product's scalar products BEFORE this kernel ran
// (finalize_scalars3), so there are no partial sums to leave - and a half iteration that met the stopping rule still carries out ITS update
// (`half` = its number; later, speculative halves return)
template <bool SR = false>
__global__ __launch_bounds__(VB) void k_bicg_upd1(int n, const double* __restrict__ scal, double* __restrict__ r,
                                                  const double* __restrict__ v, double* __restrict__ part, int npart, double half = -1.0) {
    if (scal[SC_DONE] != 0.0 && !(SR && scal[SC_DONEH] == half)) return;
    const double alpha = scal[SC_ALPHA];
    double s = 0.0;
    const int base = blockIdx.x * VB * VPT + threadIdx.x;
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
        const int e = base + u * VB;
        if (e < n) {
            const double re = r[e] - alpha * OPMHIP_VLD(&v[e]);
            r[e] = re;
            s += re * re;
        }
    }
    if (!SR) block_partials(s, 0.0, part, npart, 1);
}
// x += alpha pw, alone: the solve ended on a first half
__global__ __launch_bounds__(VB) void k_bicg_xhalf(int n, const double* __restrict__ scal, double* __restrict__ x, const double* __restrict__ pw, double ws) {
    const double alpha = scal[SC_ALPHA];
    const int base = blockIdx.x * VB * VPT + threadIdx.x;
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
        const int e = base + u * VB;
        if (e < n) x[e] += alpha * (ws * pw[e]);
    }
}
// x = (x + alpha pw) + omega s ; r -= omega t ; partials r.r and rw.r
template <bool SR = false>
__global__ __launch_bounds__(VB) void k_bicg_upd2(int n, const double* __restrict__ scal, double* __restrict__ x,
                                                  const double* __restrict__ pw, const double* __restrict__ sv, double* __restrict__ r,
                                                  const double* __restrict__ tv, const double* __restrict__ rw,
                                                  double* __restrict__ part, int npart, double ws, double half = -1.0) {
    // ws: pw and sv are M^-1 results still to be multiplied by the relaxation factor (1 when they already are, see bicgstab)
    if (scal[SC_DONE] != 0.0 && !(SR && scal[SC_DONEH] == half)) return;
    const double alpha = scal[SC_ALPHA], omega = scal[SC_OMEGA];
    double s = 0.0, q = 0.0;
    const int base = blockIdx.x * VB * VPT + threadIdx.x;
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
        const int e = base + u * VB;
        if (e < n) {
            const double xh = OPMHIP_VLD(&x[e]) + alpha * (ws * OPMHIP_VLD(&pw[e]));   // the first half's update
            OPMHIP_VST(xh + omega * (ws * OPMHIP_VLD(&sv[e])), &x[e]);
            const double re = r[e] - omega * OPMHIP_VLD(&tv[e]);
            r[e] = re;
            s += re * re;
            q += rw[e] * re;
        }
    }
    if (!SR) block_partials(s, q, part, npart, 2);
}
// plain dots for the path where wells modify y after the SpMV: part0 = a.b, part1 = a.a
__global__ __launch_bounds__(VB) void k_dots(int n, const double* __restrict__ a, const double* __restrict__ b,
                                             double* __restrict__ part, int npart, int nsum) {
    double s = 0.0, q = 0.0;
    const int base = blockIdx.x * VB * VPT + threadIdx.x;
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
        const int e = base + u * VB;
        if (e < n) { s += a[e] * b[e]; q += a[e] * a[e]; }
    }
    block_partials(s, q, part, npart, nsum);
}
// the three scalar products of a half iteration with fused reductions where they do not ride in the product's kernel (wells modify y after
// it; patterns without the stencil form): part0 = a.b, part1 = a.a, part2 = a.c; grid-stride, at most RED1_SINGLE_MAX workgroups
__global__ __launch_bounds__(VB) void k_dots3(int n, const double* __restrict__ a, const double* __restrict__ b, const double* __restrict__ c3,
                                              double* __restrict__ part, int npart) {
    __shared__ double sh[3][VB / 64];
    double s = 0.0, q = 0.0, t = 0.0;
    for (int base = blockIdx.x * VB * VPT + threadIdx.x; base < n; base += gridDim.x * VB * VPT) {
#pragma unroll
        for (int u = 0; u < VPT; ++u) {
            const int e = base + u * VB;
            if (e < n) { const double ae = a[e]; s += ae * b[e]; q += ae * ae; t += ae * c3[e]; }
        }
    }
    s = wave_sum(s); q = wave_sum(q); t = wave_sum(t);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { sh[0][wv] = s; sh[1][wv] = q; sh[2][wv] = t; }
    __syncthreads();
    if (threadIdx.x < 3) {
        double a0 = sh[threadIdx.x][0];
        for (int i = 1; i < VB / 64; ++i) a0 += sh[threadIdx.x][i];
        part[threadIdx.x * npart + blockIdx.x] = a0;
    }
}
// Long partial lists (one partial per 32-row tile = 31250 at 100^3) are reduced by RED1_BLOCKS workgroups, each summing one
// contiguous slice in a fixed order (k_reduce_finalize); short ones by a single workgroup (k_finalize / k_local_sums).
#ifndef OPMHIP_RED1_BLOCKS
#define OPMHIP_RED1_BLOCKS 128
#endif
constexpr int RED1_BLOCKS = OPMHIP_RED1_BLOCKS;
#ifndef OPMHIP_RED1_SINGLE_MAX
#define OPMHIP_RED1_SINGLE_MAX 2048
#endif
// partial lists up to this length go through one workgroup (k_finalize): 1465 partials of a 10^6-row vector kernel take it 5 us,
// the two-stage kernel with its ticket 7.7 (vector scopes 0.0285 -> 0.0260 ms on the bench); the 1953 chain-tile partials of
// of longer lists stay on the two-stage kernel
constexpr int RED1_SINGLE_MAX = OPMHIP_RED1_SINGLE_MAX;
// out[0], out[1] = the two sums of the partial lists, fixed order (input of the all-reduce in decomposed runs)
__global__ __launch_bounds__(VB) void k_local_sums(int count, const double* __restrict__ part, int npart, double* __restrict__ out) {
    __shared__ double sh[2][VB];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < count; i += VB) { a += part[i]; b += part[npart + i]; }
    sh[0][threadIdx.x] = a; sh[1][threadIdx.x] = b;
    __syncthreads();
    for (int o = VB / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { sh[0][threadIdx.x] += sh[0][threadIdx.x + o]; sh[1][threadIdx.x] += sh[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = sh[0][0]; out[1] = sh[1][0]; }
}
// Sum the partials in a fixed order and update the device-resident scalars.  One workgroup.
enum FinMode { FIN_INIT = 0, FIN_ALPHA = 1, FIN_NORM = 2, FIN_OMEGA = 3, FIN_NORM_RHO = 4,
               FIN_LOCAL = 5,     // decomposed runs: only leave the two local sums in scal[0], scal[1] (input of the all-reduce)
               FIN_SR1 = 6, FIN_SR2 = 7,   // opmhip_config.fused_reductions: ONE reduction of three sums per half iteration (finalize_scalars3)
               FIN_TRUE = 8 };             // ... and, once per solve, the norm of the residual vector itself (scal[SC_TMP1])
// FIN_NORM / FIN_NORM_RHO evaluate the stopping rule (norm < tol * norm_0, bda/cusparseSolverBackend.cu:115,151) on the
// device: they raise scal[SC_DONE] and leave (norm, norm_0, done) in the pinned host slot `hslot` for the host, which
// meanwhile has enqueued the next half iteration already.  Once the flag is up nothing changes any more.
__device__ __forceinline__ void finalize_scalars(int mode, double s0, double s1, double* __restrict__ scal, double tol, double* hslot, double seq) {
    switch (mode) {
        case FIN_LOCAL: scal[0] = s0; scal[1] = s1; return;
        case FIN_INIT:
            scal[SC_NORM0] = sqrt(s0); scal[SC_NORM] = sqrt(s0);
            scal[SC_RHO] = s0; scal[SC_RHOP] = 1.0; scal[SC_ALPHA] = 1.0; scal[SC_OMEGA] = 1.0; scal[SC_BETA] = 0.0;
            scal[SC_DONE] = 0.0;
            scal[SC_RR] = s0; scal[SC_RHOH] = s0; scal[SC_DONEH] = -1.0;
            break;
        case FIN_TRUE: scal[SC_TMP1] = sqrt(s0); return;
        case FIN_ALPHA: scal[SC_TMP1] = s0; scal[SC_ALPHA] = scal[SC_RHO] / s0; break;
        case FIN_NORM: scal[SC_NORM] = sqrt(s0); break;  // stopping rule: below
        case FIN_OMEGA: scal[SC_TMP1] = s0; scal[SC_TMP2] = s1; scal[SC_OMEGA] = s0 / s1; break;
        case FIN_NORM_RHO: {
            scal[SC_NORM] = sqrt(s0);
            const double rhop = scal[SC_RHO];
            scal[SC_RHOP] = rhop; scal[SC_RHO] = s1;
            scal[SC_BETA] = (s1 / rhop) * (scal[SC_ALPHA] / scal[SC_OMEGA]);
        } break;
    }
    if (hslot && (mode == FIN_NORM || mode == FIN_NORM_RHO)) {
        const double norm = scal[SC_NORM], norm0 = scal[SC_NORM0];
        const double stop = (norm < tol * norm0) ? 1.0 : 0.0;
        scal[SC_DONE] = stop;
        hslot[0] = norm; hslot[1] = norm0; hslot[2] = stop;
        __hip_atomic_store(&hslot[3], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);  // the host polls this one
    }
}
// One reduction per half iteration (opmhip_config.fused_reductions; the reference forms alpha, the norm, omega, the norm and rho from five
// scalar products in four reductions, bda/cusparseSolverBackend.cu:92, 120-127, 151-161).  The product's kernel leaves three sums:
//   first half,  v = A M^-1 p:  s0 = v.rw, s1 = v.v, s2 = v.r      alpha = rho / s0
//       |r - alpha v|^2 = r.r - 2 alpha s2 + alpha^2 s1            rw.(r - alpha v) = rho - alpha s0   (0 but for rounding)
//   second half, t = A M^-1 r:  s0 = t.r, s1 = t.t, s2 = t.rw      omega = s0 / s1
//       |r - omega t|^2 = r.r - 2 omega s0 + omega^2 s1            rho' = rw.(r - omega t) = (rho - alpha v.rw) - omega s2
// with r.r carried from half iteration to half iteration (SC_RR; negative results of the cancellation are clamped to 0).  The stopping
// rule is evaluated HERE, before the update kernels of the half run: a half that meets it is remembered (SC_DONEH) so that its own
// updates are still carried out.  oracle/linalg.hpp: bicgstab_fused_reductions states the same arithmetic.
__device__ __forceinline__ void finalize_scalars3(int mode, double s0, double s1, double s2, double* __restrict__ scal, double tol, double* hslot, double seq, double half) {
    const double rr = scal[SC_RR];
    double rrn;
    if (mode == FIN_SR1) {
        const double rho = scal[SC_RHO];
        const double alpha = rho / s0;
        scal[SC_TMP1] = s0;
        scal[SC_ALPHA] = alpha;
        rrn = (rr - 2.0 * alpha * s2) + alpha * alpha * s1;
        scal[SC_RHOH] = rho - alpha * s0;
    } else {
        const double omega = s0 / s1;
        scal[SC_TMP1] = s0; scal[SC_TMP2] = s1;
        scal[SC_OMEGA] = omega;
        rrn = (rr - 2.0 * omega * s0) + omega * omega * s1;
        const double rhop = scal[SC_RHO], rhon = scal[SC_RHOH] - omega * s2;
        scal[SC_RHOP] = rhop; scal[SC_RHO] = rhon;
        scal[SC_BETA] = (rhon / rhop) * (scal[SC_ALPHA] / omega);
    }
    if (!(rrn > 0.0)) rrn = (rrn != rrn) ? rrn : 0.0;   // (a NaN stays a NaN: the caller reports it)
    scal[SC_RR] = rrn;
    const double norm = sqrt(rrn), norm0 = scal[SC_NORM0];
    scal[SC_NORM] = norm;
    const double stop = (norm < tol * norm0) ? 1.0 : 0.0;
    if (stop != 0.0) scal[SC_DONEH] = half;
    scal[SC_DONE] = stop;
    if (hslot) {
        hslot[0] = norm; hslot[1] = norm0; hslot[2] = stop;
        __hip_atomic_store(&hslot[3], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ __launch_bounds__(VB) void k_finalize3(int mode, int count, const double* __restrict__ part, int npart,
                                                  double* __restrict__ scal, double tol, double* hslot, double seq, double half) {
    __shared__ double sh[3][VB];
    constexpr int FU = 8;
    double va[FU], vb[FU], vc[FU];
#pragma unroll
    for (int u = 0; u < FU; ++u) {
        const int i = (int)threadIdx.x + u * VB;
        const int j = i < count ? i : 0;
        va[u] = part[j]; vb[u] = part[npart + j]; vc[u] = part[2 * npart + j];
    }
    if (mode != FIN_LOCAL && scal[SC_DONE] != 0.0) {   // a speculative launch past the stopping point still answers the host
        if (threadIdx.x == 0 && hslot) {
            hslot[0] = scal[SC_NORM]; hslot[1] = scal[SC_NORM0]; hslot[2] = 1.0;
            __hip_atomic_store(&hslot[3], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    double a = 0.0, b = 0.0, c3 = 0.0;
#pragma unroll
    for (int u = 0; u < FU; ++u)
        if ((int)threadIdx.x + u * VB < count) { a += va[u]; b += vb[u]; c3 += vc[u]; }
    for (int i = (int)threadIdx.x + FU * VB; i < count; i += VB) { a += part[i]; b += part[npart + i]; c3 += part[2 * npart + i]; }
    sh[0][threadIdx.x] = a; sh[1][threadIdx.x] = b; sh[2][threadIdx.x] = c3;
    __syncthreads();
    for (int o = VB / 2; o >= 64; o >>= 1) {
        if ((int)threadIdx.x < o) { sh[0][threadIdx.x] += sh[0][threadIdx.x + o]; sh[1][threadIdx.x] += sh[1][threadIdx.x + o]; sh[2][threadIdx.x] += sh[2][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x >= 64) return;
    a = sh[0][threadIdx.x]; b = sh[1][threadIdx.x]; c3 = sh[2][threadIdx.x];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); b += __shfl_down(b, o, 64); c3 += __shfl_down(c3, o, 64); }
    if (threadIdx.x != 0) return;
    if (mode == FIN_LOCAL) { scal[0] = a; scal[1] = b; scal[2] = c3; return; }   // decomposed runs: the three local sums, input of the all-reduce
    finalize_scalars3(mode, a, b, c3, scal, tol, hslot, seq, half);
}
__global__ __launch_bounds__(VB) void k_finalize(int mode, int count, const double* __restrict__ part, int npart,
                                                 double* __restrict__ scal, double tol, double* hslot, double seq) {
    __shared__ double sh[2][VB];
    // The partial sums were written by the kernel before this one, from every XCD: each load is a trip to the fabric.  All
    // of a thread's partials (and the stop flag) are requested in ONE round and then added in the fixed order - thread t:
    // part[t], part[t + VB], ... - instead of one dependent round trip per addend (5.2 us -> the figure in DESIGN.md).
    constexpr int FU = 8;   // FU x VB = 2048 covers RED1_SINGLE_MAX; longer lists finish in the loop below
    double va[FU], vb[FU];
#pragma unroll
    for (int u = 0; u < FU; ++u) {
        const int i = (int)threadIdx.x + u * VB;
        const int j = i < count ? i : 0;
        va[u] = part[j]; vb[u] = part[npart + j];
    }
    const double stop = (mode != FIN_INIT && mode != FIN_TRUE) ? scal[SC_DONE] : 0.0;
    if (stop != 0.0) {
        // a speculative launch past the stopping point still answers the host, which may be polling this slot
        if (threadIdx.x == 0 && hslot && (mode == FIN_NORM || mode == FIN_NORM_RHO)) {
            hslot[0] = scal[SC_NORM]; hslot[1] = scal[SC_NORM0]; hslot[2] = 1.0;
            __hip_atomic_store(&hslot[3], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int u = 0; u < FU; ++u)
        if ((int)threadIdx.x + u * VB < count) { a += va[u]; b += vb[u]; }
    for (int i = (int)threadIdx.x + FU * VB; i < count; i += VB) { a += part[i]; b += part[npart + i]; }
    sh[0][threadIdx.x] = a; sh[1][threadIdx.x] = b;
    __syncthreads();
    // the tree sh[t] += sh[t + o], o = VB/2 ... 1: across wavefronts through LDS, inside the first one through lane shifts
    // (lane t takes lane t + o's value: the same additions in the same order)
    for (int o = VB / 2; o >= 64; o >>= 1) {
        if ((int)threadIdx.x < o) { sh[0][threadIdx.x] += sh[0][threadIdx.x + o]; sh[1][threadIdx.x] += sh[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x >= 64) return;
    a = sh[0][threadIdx.x]; b = sh[1][threadIdx.x];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); b += __shfl_down(b, o, 64); }
    if (threadIdx.x != 0) return;
    finalize_scalars(mode, a, b, scal, tol, hslot, seq);
}
// slice sums and k_finalize in one launch for long partial lists: the workgroup that finishes last (ticket counter)
// sums the RED1_BLOCKS slice sums in k_finalize's order and updates the scalars.  The slice sums cross XCDs inside one
// kernel, so they travel as agent-scope atomics (an XCD's L2 is not coherent with the others' for plain accesses).
static_assert(RED1_BLOCKS <= VB, "the last workgroup holds one slice sum per thread");
__global__ __launch_bounds__(VB) void k_reduce_finalize(int mode, int count, const double* __restrict__ part, int npart, double* out,
                                                        unsigned* ticket, double* __restrict__ scal, double tol, double* hslot, double seq) {
    __shared__ double sh[2][VB];
    __shared__ bool last;
    if (mode != FIN_INIT && mode != FIN_LOCAL && mode != FIN_TRUE && scal[SC_DONE] != 0.0) {
        if (blockIdx.x == 0 && threadIdx.x == 0 && hslot && (mode == FIN_NORM || mode == FIN_NORM_RHO)) {
            hslot[0] = scal[SC_NORM]; hslot[1] = scal[SC_NORM0]; hslot[2] = 1.0;
            __hip_atomic_store(&hslot[3], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    const int chunk = (count + RED1_BLOCKS - 1) / RED1_BLOCKS;
    const int b0 = blockIdx.x * chunk, e0 = min(count, b0 + chunk);
    double a0 = 0.0, a1 = 0.0;
    for (int i = b0 + threadIdx.x; i < e0; i += VB) { a0 += part[i]; a1 += part[npart + i]; }
    sh[0][threadIdx.x] = a0; sh[1][threadIdx.x] = a1;
    __syncthreads();
    for (int o = VB / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { sh[0][threadIdx.x] += sh[0][threadIdx.x + o]; sh[1][threadIdx.x] += sh[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        __hip_atomic_store(&out[blockIdx.x], sh[0][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&out[RED1_BLOCKS + blockIdx.x], sh[1][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = (t == gridDim.x - 1);
    }
    __syncthreads();
    if (!last) return;
    double a = 0.0, b = 0.0;
    if (threadIdx.x < RED1_BLOCKS) {
        a = __hip_atomic_load(&out[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        b = __hip_atomic_load(&out[RED1_BLOCKS + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    sh[0][threadIdx.x] = a; sh[1][threadIdx.x] = b;
    __syncthreads();
    for (int o = VB / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { sh[0][threadIdx.x] += sh[0][threadIdx.x + o]; sh[1][threadIdx.x] += sh[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    finalize_scalars(mode, sh[0][0], sh[1][0], scal, tol, hslot, seq);
}

// On-box streaming ceiling (opmhip_time_kernel, which = 4): every byte of the Jacobian's value array read once with 16-byte
// loads, eight in flight per lane, nothing else - what the HBM of THIS card delivers to a kernel that only streams, to set
// beside the 8 TB/s spec figure the roofline fractions are quoted against (SURVEY.md section 8d asks for both).
__global__ __launch_bounds__(256) void k_stream_read(size_t n2, const double2* __restrict__ src, double* __restrict__ sink) {
    // consecutive workgroups read consecutive 32-KiB pieces (the access shape of the tile kernels): 8 x 4 KiB rows per
    // workgroup iteration, all eight loads of a lane in flight before the first is used
    double s = 0.0;
    constexpr size_t PIECE = 256 * 8;  // double2 per workgroup iteration
    const size_t npiece = n2 / PIECE;
    for (size_t p = blockIdx.x; p < npiece; p += gridDim.x) {
        const double2* q = src + p * PIECE + threadIdx.x;
        double2 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = q[u * 256];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += t[u].x + t[u].y;
    }
    for (size_t i = npiece * PIECE + (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) { const double2 t = src[i]; s += t.x + t.y; }
    if (s == 1.2345678e300) sink[0] = s;  // never true: keeps the loads alive
}
void launch_stream_read(opmhip_ctx* c) {
    const size_t n2 = (size_t)c->pat.nnzb * BB / 2;
    hipLaunchKernelGGL(k_stream_read, dim3(256 * 8), dim3(256), 0, c->stream, n2, reinterpret_cast<const double2*>(c->d_A), c->d_part2);
}

// ============================== launchers ================================================================
static inline int cdiv(size_t a, size_t b) { return (int)((a + b - 1) / b); }
static inline int vec_blocks(int n) { return cdiv((size_t)n, (size_t)VB * VPT); }

void launch_permute_blocks(opmhip_ctx* c, const double* nat, double* internal) {
    const size_t n = (size_t)c->pat.nnzb * BB;
    hipLaunchKernelGGL(k_permute_blocks, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, c->pat.nnzb, c->pat.d_nnzMap, nat, internal);
}
// cells = -1: the owned rows (solver vectors); otherwise that many cells (state vectors include the ghosts)
void launch_vec_to_internal(opmhip_ctx* c, const double* nat, double* internal, int cells) {
    const int n = cells < 0 ? c->pat.Nb : cells;
    hipLaunchKernelGGL(k_vec_to_internal, dim3(cdiv((size_t)n * BS, 256)), dim3(256), 0, c->stream, n, c->pat.d_fromOrder, nat, internal);
}
void launch_vec_to_natural(opmhip_ctx* c, const double* internal, double* nat, int cells) {
    const int n = cells < 0 ? c->pat.Nb : cells;
    hipLaunchKernelGGL(k_vec_to_natural, dim3(cdiv((size_t)n * BS, 256)), dim3(256), 0, c->stream, n, c->pat.d_toOrder, internal, nat);
}
void launch_zero_diag_fix(opmhip_ctx* c) {
    hipLaunchKernelGGL(k_zero_diag_fix, dim3(cdiv((size_t)c->pat.Nb * BS, 256)), dim3(256), 0, c->stream, c->pat.Nb, c->pat.d_diag, c->d_A);
}
void launch_lu_to_natural(opmhip_ctx* c, double* d_out) {
    const Pattern& P = c->pat;
    hipLaunchKernelGGL(k_lu_to_bcrs, dim3(cdiv((size_t)P.Nb, 256)), dim3(256), 0, c->stream, P.Nb, P.d_rowptr, P.d_col, P.d_lrowptr,
                       P.d_urowptr, c->d_L, c->d_U, c->d_invD, d_out);
}
int launch_wells_apply(opmhip_ctx* c, const double* x, double* y, double xs) {
    const WellsDev& W = c->wells;
    if (W.num_wells <= 0) return OPMHIP_SUCCESS;
    if (W.distributed) {
        // a well shared by several subdomains: this rank's perforations give its part of B x, the parts are summed over the ranks, D^-1 and
        // C^T follow on every rank for its own cells (ParallelStandardWellB::mv, wells/WellHelpers.hpp:68-123; StandardWell::apply,
        // wells/StandardWell_impl.hpp:1254-1280)
        hipLaunchKernelGGL(k_wells_bx, dim3(W.num_wells), dim3(64), 0, c->stream, W.d_val_pointers, W.d_Bcols, W.d_B, x, xs, W.d_bx);
        const int rc = comm_allreduce(c, W.d_bx, W.num_wells * 4, 0);
        if (rc) return rc;
        hipLaunchKernelGGL(k_wells_residual, dim3(W.num_wells), dim3(64), 0, c->stream, W.d_val_pointers, W.d_Ccols, W.d_C, W.d_D, W.d_bx, y);
        return OPMHIP_SUCCESS;
    }
    hipLaunchKernelGGL(k_wells_apply, dim3(W.num_wells), dim3(64), 0, c->stream, W.d_val_pointers, W.d_Ccols, W.d_Bcols, W.d_C, W.d_D,
                       W.d_B, x, y, xs);
    return OPMHIP_SUCCESS;
}
void launch_wells_add_to_matrix(opmhip_ctx* c, int w0, int nw, int serial, const int* d_pair_ptr, const int* d_entry) {
    const WellsDev& W = c->wells;
    hipLaunchKernelGGL(k_wells_add_to_matrix, dim3(nw), dim3(64), 0, c->stream, w0, serial, W.d_val_pointers, W.d_C, W.d_D, W.d_B, d_pair_ptr, d_entry, c->d_A);
}
void launch_wells_residual(opmhip_ctx* c, const double* d_resWell, double* r) {
    const WellsDev& W = c->wells;
    if (W.num_wells <= 0) return;
    hipLaunchKernelGGL(k_wells_residual, dim3(W.num_wells), dim3(64), 0, c->stream, W.d_val_pointers, W.d_Ccols, W.d_C, W.d_D, d_resWell, r);
}
int launch_wells_recover(opmhip_ctx* c, const double* d_resWell, const double* x, double* d_xw) {
    const WellsDev& W = c->wells;
    if (W.num_wells <= 0) return OPMHIP_SUCCESS;
    if (W.distributed) {   // resWell - (sum over the ranks of B x): ParallelStandardWellB::mmv's branch for a shared well, wells/WellHelpers.hpp:136-141
        hipLaunchKernelGGL(k_wells_bx, dim3(W.num_wells), dim3(64), 0, c->stream, W.d_val_pointers, W.d_Bcols, W.d_B, x, 1.0, W.d_bx);
        const int rc = comm_allreduce(c, W.d_bx, W.num_wells * 4, 0);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_wells_recover, dim3(W.num_wells), dim3(64), 0, c->stream, W.d_val_pointers, W.d_Bcols, W.d_B, W.d_D, d_resWell, x,
                       W.distributed ? W.d_bx : nullptr, d_xw);
    return OPMHIP_SUCCESS;
}
#ifndef OPMHIP_SPMV_PIPE_WGS
#define OPMHIP_SPMV_PIPE_WGS 2048
#endif
constexpr int SPMV_PIPE_WGS = OPMHIP_SPMV_PIPE_WGS;  // resident single-wave workgroups the pipelined SpMV is sized for (256 CUs x 8)
static int spmv_pipe_env() {   // OPMHIP_SPMV_PIPE (tuning): 0 = off, n > 1 = workgroups the pipelined kernel is sized for
    static const int v = [] { const char* e = tuning_env("OPMHIP_SPMV_PIPE"); return e ? std::atoi(e) : -1; }();
    return v;
}
// cfg.spmv_pipe_wgs: resident workgroups the pipelined kernel is sized for (0 = default, < 0 = never use it)
static int spmv_pipe_wgs(const opmhip_ctx* c) {
    return c->cfg.spmv_pipe_wgs != 0 ? c->cfg.spmv_pipe_wgs : (spmv_pipe_env() > 1 ? spmv_pipe_env() : SPMV_PIPE_WGS);
}
static bool spmv_pipelined(const opmhip_ctx* c) {
    const int w = spmv_pipe_wgs(c);
    return c->pat.maxRowBlocks <= PGCH && w > 0 && c->pat.tiles.nsched > w && spmv_pipe_env() != 0;
}
// opmhip_config.half_product resolved: > 0 wherever the pattern allows it (Pattern::ualias, a line-coloured ordering, the rest's stencil
// form - RestSched::on); 0, the library's choice: there, where the system is large enough for the pipelined kernels (the size the form
// was measured at); < 0 never.  Subdomains of a decomposed run: the interior tiles take the form, the boundary tiles (rows with ghost
// columns) keep the whole product - each row's sum is one or the other, as the schedules say.
bool half_product_wanted(const opmhip_ctx* c) {
    const Pattern& P = c->pat;
    if (!P.rest.on || c->cfg.half_product < 0 || use_cpr(c)) return false;   // (CPR: the product follows the two-level application, not a sweep)
    if (c->cfg.half_product > 0) return true;
    static const bool off = [] { const char* e = tuning_env("OPMHIP_HALF_PRODUCT"); return e && e[0] == '0'; }();   // A/B switch
    return !off && spmv_pipelined(c);
}
// the scalar products ride in the product's kernel unless wells modify y after it (then k_dots forms them afterwards)
static bool spmv_dots_env() {   // OPMHIP_DOTS_SEPARATE=1 (tuning / A-B measurements): k_dots behind every product, as with wells
    static const bool v = [] { const char* e = tuning_env("OPMHIP_DOTS_SEPARATE"); return e && std::atoi(e) != 0; }();
    return v;
}
static bool spmv_dots_separate(const opmhip_ctx* c) { return c->wells.any() || spmv_dots_env(); }
// one launch over the schedule positions [p0, p0 + np): the pipelined kernel where the pattern allows it, else one tile per
// workgroup.  Partial sums go to part[pofs ...]; returns how many were written (0 with ndot == 0).
static int launch_spmv_part(opmhip_ctx* c, int p0, int np, const double* x, double* y, int ndot, const double* w0, double xs, int pofs, int cls,
                            const double* uadd = nullptr, const double* w1 = nullptr) {
    const Pattern& P = c->pat;
    if (np <= 0) return 0;
    double* part = c->d_part + pofs;
    // timed by its own dispatch (kernel begin to kernel end), which is what bench.py's roofline quotes
    int es = -1, ee = -1;
    const bool timed = prof_kernel_scope(c, cls, &es, &ee);
    hipEvent_t e0 = timed ? c->prof.ev[es] : nullptr, e1 = timed ? c->prof.ev[ee] : nullptr;
    if (uadd) {
        // the rest product (Pattern::ualias): the pipelined kernel on the schedule, the index streams and the values of the matrix WITHOUT
        // its U part, the row sums of the backward sweep added at the end of every row.  The grid as for the full product: every workgroup
        // resident, all ending together; small systems: one tile per workgroup.
        const RestSched& R = P.rest;
        const int4* rsched = reinterpret_cast<const int4*>(R.d_sched) + p0;
        static const int restWgs = [] { const char* e = tuning_env("OPMHIP_REST_WGS"); return e ? std::atoi(e) : 0; }();   // measurement switch
        const int pipeWgs = restWgs > 0 ? restWgs : std::max(8, spmv_pipe_wgs(c));
        const int steps = std::min((np + pipeWgs - 1) / pipeWgs, PIPE_MAX_STEPS);
        const int grid = 8 * (((np + steps - 1) / steps + 7) / 8);
        const int* tab = R.d_table + (size_t)p0 * 16;
        if (ndot == 0)
            hipExtLaunchKernelGGL((k_spmv_pipe_st<0, true>), dim3(grid), dim3(64), 0, c->stream, e0, e1, 0, np, rsched, R.d_word, R.d_koff, tab, c->d_R, x, y, w0, part, c->npart, c->d_done, xs, uadd, (const double*)nullptr);
        else if (ndot == 1)
            hipExtLaunchKernelGGL((k_spmv_pipe_st<1, true>), dim3(grid), dim3(64), 0, c->stream, e0, e1, 0, np, rsched, R.d_word, R.d_koff, tab, c->d_R, x, y, w0, part, c->npart, c->d_done, xs, uadd, (const double*)nullptr);
        else if (ndot == 2)
            hipExtLaunchKernelGGL((k_spmv_pipe_st<2, true>), dim3(grid), dim3(64), 0, c->stream, e0, e1, 0, np, rsched, R.d_word, R.d_koff, tab, c->d_R, x, y, w0, part, c->npart, c->d_done, xs, uadd, (const double*)nullptr);
        else
            hipExtLaunchKernelGGL((k_spmv_pipe_st<3, true>), dim3(grid), dim3(64), 0, c->stream, e0, e1, 0, np, rsched, R.d_word, R.d_koff, tab, c->d_R, x, y, w0, part, c->npart, c->d_done, xs, uadd, w1);
        return ndot > 0 ? grid : 0;
    }
    const int4* sched = reinterpret_cast<const int4*>(P.tiles.d_spmvSched) + p0;
    if (spmv_pipelined(c)) {
        // every workgroup walks through ceil(np / grid) launch positions; the grid is sized so that all workgroups are
        // resident at once and end together, and is a multiple of 8 (a workgroup stays on "its" XCD column of the schedule)
        const int pipeWgs = spmv_pipe_wgs(c);
        const int steps = std::min((np + pipeWgs - 1) / pipeWgs, PIPE_MAX_STEPS);   // beyond that: more workgroups than are resident
        const int grid = 8 * (((np + steps - 1) / steps + 7) / 8);
        static const bool explicitIdx = [] { const char* e = tuning_env("OPMHIP_SPMV_EXPLICIT"); return e && e[0] == '1'; }();   // A/B switch: the explicit index streams
        const bool inInt = p0 < P.tiles.nschedInt, inBnd = p0 + np > P.tiles.nschedInt;   // which parts of the schedule this launch covers
        if ((!inInt || P.tiles.stencilPart[0]) && (!inBnd || P.tiles.stencilPart[1]) && !explicitIdx) {
            const int* tab = P.tiles.d_stTable + (size_t)p0 * 16;
            if (ndot == 0)
                hipExtLaunchKernelGGL((k_spmv_pipe_st<0, false>), dim3(grid), dim3(64), 0, c->stream, e0, e1, 0, np, sched, P.tiles.d_stWord, P.tiles.d_stKoff, tab, c->d_A, x, y, w0, part, c->npart, c->d_done, xs, (const double*)nullptr, (const double*)nullptr);
            else if (ndot == 1)
                hipExtLaunchKernelGGL((k_spmv_pipe_st<1, false>), dim3(grid), dim3(64), 0, c->stream, e0, e1, 0, np, sched, P.tiles.d_stWord, P.tiles.d_stKoff, tab, c->d_A, x, y, w0, part, c->npart, c->d_done, xs, (const double*)nullptr, (const double*)nullptr);
            else if (ndot == 2)
                hipExtLaunchKernelGGL((k_spmv_pipe_st<2, false>), dim3(grid), dim3(64), 0, c->stream, e0, e1, 0, np, sched, P.tiles.d_stWord, P.tiles.d_stKoff, tab, c->d_A, x, y, w0, part, c->npart, c->d_done, xs, (const double*)nullptr, (const double*)nullptr);
            else
                hipExtLaunchKernelGGL((k_spmv_pipe_st<3, false>), dim3(grid), dim3(64), 0, c->stream, e0, e1, 0, np, sched, P.tiles.d_stWord, P.tiles.d_stKoff, tab, c->d_A, x, y, w0, part, c->npart, c->d_done, xs, (const double*)nullptr, w1);
            return ndot > 0 ? grid : 0;
        }
        if (ndot == 0)
            hipExtLaunchKernelGGL(k_spmv_pipe<0>, dim3(grid), dim3(64), 0, c->stream, e0, e1, 0, np, sched, P.d_rowptr, P.d_col, c->d_A, x, y, w0, part, c->npart, c->d_done, xs);
        else if (ndot == 1)
            hipExtLaunchKernelGGL(k_spmv_pipe<1>, dim3(grid), dim3(64), 0, c->stream, e0, e1, 0, np, sched, P.d_rowptr, P.d_col, c->d_A, x, y, w0, part, c->npart, c->d_done, xs);
        else
            hipExtLaunchKernelGGL(k_spmv_pipe<2>, dim3(grid), dim3(64), 0, c->stream, e0, e1, 0, np, sched, P.d_rowptr, P.d_col, c->d_A, x, y, w0, part, c->npart, c->d_done, xs);
        return ndot > 0 ? grid : 0;
    }
    if (ndot == 0)
        hipExtLaunchKernelGGL(k_spmv<0>, dim3(np), dim3(64), 0, c->stream, e0, e1, 0, sched, P.d_rowptr, P.d_col, c->d_A, x, y, w0, part, c->npart, c->d_done, xs);
    else if (ndot == 1)
        hipExtLaunchKernelGGL(k_spmv<1>, dim3(np), dim3(64), 0, c->stream, e0, e1, 0, sched, P.d_rowptr, P.d_col, c->d_A, x, y, w0, part, c->npart, c->d_done, xs);
    else
        hipExtLaunchKernelGGL(k_spmv<2>, dim3(np), dim3(64), 0, c->stream, e0, e1, 0, sched, P.d_rowptr, P.d_col, c->d_A, x, y, w0, part, c->npart, c->d_done, xs);
    return ndot > 0 ? np : 0;
}
// Multisegment wells: y -= C^T (D^-1 (B (xs x))) on the HOST, by the caller's objects (opmhip_wells.ms_apply) - x and y to pinned memory in
// the natural order, the callback, y back: the round trip the reference's back-ends make after every product
// (bda/WellContributions.cu:160-187; MultisegmentWellContribution::apply, bda/MultisegmentWellContribution.cpp:70-110).  The stream is
// drained twice per product; a deck with multisegment wells pays what it pays in the reference.
static int ms_wells_apply(opmhip_ctx* c, const double* x, double* y, double xs) {
    WellsDev& W = c->wells;
    const size_t n = (size_t)c->pat.Nb * BS;
    launch_vec_to_natural(c, x, c->d_stageV);
    OPMHIP_HIP(c, hipMemcpyAsync(W.h_x, c->d_stageV, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    launch_vec_to_natural(c, y, c->d_stageV);
    OPMHIP_HIP(c, hipMemcpyAsync(W.h_y, c->d_stageV, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
    if (xs != 1.0)   // x arrives without the relaxation factor of M^-1 (launch_ilu_apply's `unscaled`): one rounded product per entry, as on the device
        for (size_t i = 0; i < n; ++i) W.h_x[i] *= xs;
    W.ms_apply(W.ms_user, W.h_x, W.h_y);
    OPMHIP_HIP(c, hipMemcpyAsync(c->d_stageV, W.h_y, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    launch_vec_to_internal(c, c->d_stageV, y);
    OPMHIP_HIP(c, hipStreamSynchronize(c->stream));   // h_y may be written again by the next product
    return OPMHIP_SUCCESS;
}
// y = A x (+ wells) and the partial sums of the scalar products: ndot 0 none, 1 y.w0, 2 y.w0 and y.y.
// exchange (decomposed runs): the ghost entries of x are refreshed first - Dune's copyOwnerToAll in front of the operator
// (linalg/WellOperators.hpp:127-138, ParallelOverlappingILU0.hpp:897) - on the halo stream, WHILE the interior tiles (no
// ghost column in any of their rows) are multiplied on the main stream; the boundary tiles follow when the ghosts are in.
// Every row's sum is formed by the same statements in the same order whichever launch it is in: the same bits as one launch.
// uadd != NULL: the rest product - x is the (unscaled) result of the ILU0 application that has just left its row sums in uadd:
// y_i = sum_rest A_ik (xs x_k) + xs uadd_i  (Pattern::ualias; the same wells, halo exchange and scalar products around it)
// ndot == 3 (fused reductions): y.w0, y.y and y.w1 - riding in the product's kernel where every launch of it is the pipelined stencil kernel,
// else formed by k_dots3 behind it
int launch_spmv(opmhip_ctx* c, double* x, double* y, int ndot, const double* w0, double xs, bool exchange, const double* uadd, const double* w1) {
    const Pattern& P = c->pat;
    const bool wells = c->wells.num_wells > 0;
    static const bool explicitIdx3 = [] { const char* e = tuning_env("OPMHIP_SPMV_EXPLICIT"); return e && e[0] == '1'; }();
    const bool bndPlain = uadd && P.Nghost > 0 && P.tiles.nsched > P.tiles.nschedInt;   // half-product form in a subdomain: its boundary tiles run the whole product
    const bool plain3 = spmv_pipelined(c) && !explicitIdx3;                              // the plain pipelined stencil kernel can carry three sums
    const bool rides3 = uadd ? (!bndPlain || (plain3 && P.tiles.stencilPart[1]))
                             : (plain3 && P.tiles.stencilPart[0] && (P.tiles.nsched == P.tiles.nschedInt || P.tiles.stencilPart[1]));
    const int fused = (spmv_dots_separate(c) || (ndot == 3 && !rides3)) ? 0 : ndot;
    const bool halo = exchange && c->comm.halo_set && c->comm.nneigh > 0;
    if (exchange && !halo && c->comm.nranks > 1) comm_halo_bystander(c);   // a subdomain that touches no other: nothing to exchange, but the peers' exchange counts this rank in (loopback)
    // (half-product form in a subdomain with ghost columns: the rest schedule holds the interior tiles only; the boundary tiles are the
    //  whole product's, positions [nschedInt, nsched) of the matrix's own schedule)
    const int nInt = uadd ? P.rest.nschedInt : P.tiles.nschedInt;
    const int bnd0 = (uadd && !bndPlain) ? P.rest.nschedInt : P.tiles.nschedInt;
    const int nBnd = (uadd && !bndPlain) ? P.rest.nsched - P.rest.nschedInt : P.tiles.nsched - P.tiles.nschedInt;
    const double* uBnd = bndPlain ? nullptr : uadd;
    int rc, cnt = 0;
    if (halo && nBnd > 0) {
        // main: ev_x (x complete) -> interior tiles ........................ wait ev_h -> boundary tiles
        // halo:         wait ev_x -> pack -> send / receive -> ev_h
        if ((rc = comm_halo_begin(c, x))) return rc;     // records ev_x on the main stream; the exchange itself is behind it on the halo stream
        cnt += launch_spmv_part(c, 0, nInt, x, y, fused, w0, xs, cnt, PROF_SPMV, uadd, w1);
        if ((rc = comm_halo_end(c))) return rc;
        cnt += launch_spmv_part(c, bnd0, nBnd, x, y, fused, w0, xs, cnt, PROF_SPMV_BOUNDARY, uBnd, w1);
    } else {
        if (halo && (rc = comm_halo_f64(c, x, BS))) return rc;
        cnt += launch_spmv_part(c, 0, nInt, x, y, fused, w0, xs, cnt, PROF_SPMV, uadd, w1);
        if (nBnd > 0) cnt += launch_spmv_part(c, bnd0, nBnd, x, y, fused, w0, xs, cnt, bndPlain ? PROF_SPMV_BOUNDARY : PROF_SPMV, uBnd, w1);
    }
    if (c->wells.num_ms > 0 && (rc = ms_wells_apply(c, x, y, xs))) return rc;   // in front of the standard wells, bda/WellContributions.cu:160-187
    if (wells && (rc = launch_wells_apply(c, x, y, xs))) return rc;
    if (fused == 0 && ndot > 0) {
        const int n = P.Nb * BS;
        const int ps = prof_begin(c, PROF_VECTOR);   // a scope of its own: its bytes are counted under "vector"
        if (ndot == 3) {
            cnt = std::min(vec_blocks(n), RED1_SINGLE_MAX);
            hipLaunchKernelGGL(k_dots3, dim3(cnt), dim3(VB), 0, c->stream, n, y, w0, w1, c->d_part, c->npart);
        } else {
            hipLaunchKernelGGL(k_dots, dim3(vec_blocks(n)), dim3(VB), 0, c->stream, n, y, w0, c->d_part, c->npart, ndot);
            cnt = vec_blocks(n);
        }
        prof_end(c, ps);
    }
    c->last_dot_count = cnt;
    return OPMHIP_SUCCESS;
}
static int dot_count(opmhip_ctx* c) { return c->last_dot_count; }  // how many partials the last launch_spmv left behind
// fix_zero_diagonal: exact zeros on the diagonal of a diagonal block become 1e-15 on the way (bda/BdaBridge.cpp:125-161) - in the factors'
// input and in the matrix: the separate pass over the diagonal blocks (38 us, 198 MB of traffic for 24 MB of data) is gone from the solve
void launch_ilu_factor(opmhip_ctx* c, bool fix_zero_diagonal, const FactorRider* rider) {
    const Pattern& P = c->pat;
    const int ps = prof_begin(c, PROF_ILU_FACTOR);
    const FactorRider none;
    const FactorRider& R = rider ? *rider : none;
    for (int col = 0; col < P.numColors; ++col) {
        const int off = P.tiles.ctSchedOff[col], npos = P.tiles.ctSchedOff[col + 1] - off;
        if (npos <= 0) continue;
#define OPMHIP_FACTOR_LAUNCH(RID)                                                                                                                          \
    hipLaunchKernelGGL(k_ilu_factor<RID>, dim3(npos), dim3(64), 0, c->stream, P.tiles.d_ctSched + off, P.tiles.d_ctFirst, P.tiles.d_row0, P.d_rowptr, P.d_col, \
                       P.d_diag, c->d_A, P.d_fdest, P.d_lmatch, P.d_urowptr, P.d_ucol, c->d_L, c->d_U, c->d_invD, fix_zero_diagonal ? c->d_A : (double*)nullptr, \
                       P.Nb, R.W, R.ghostFrom, R.w, R.ap, R.pcol, c->half_product ? P.d_rdest : (const int*)nullptr, c->half_product ? P.d_rrowptr : (const int*)nullptr, \
                       c->half_product ? c->d_R : (double*)nullptr)
        if (R.mode == 1) OPMHIP_FACTOR_LAUNCH(1); else if (R.mode == 2) OPMHIP_FACTOR_LAUNCH(2); else OPMHIP_FACTOR_LAUNCH(0);
#undef OPMHIP_FACTOR_LAUNCH
    }
    prof_end(c, ps);
}
// unscaled != NULL: with post-scaling the sweeps leave U^-1 L^-1 d in v WITHOUT the factor w and report the factor in *unscaled
// (1 when there is none to apply): whoever reads v next multiplies on the fly - w * v_i is one rounded product either way -
// and the second result vector (24 bytes per row written by the backward sweeps) never exists.
// usum != NULL (chained orderings, Pattern::ualias): the backward sweeps also leave u_i = sum_{j>i} U_ij x_j there, x = the sweep's own
// (unscaled) result - the upper part of A x, bit for bit, since U == upper(A)
void launch_ilu_apply(opmhip_ctx* c, const double* d, double* v, double w_override, double* unscaled, const double* addp, double* work, double* usum) {
    const Pattern& P = c->pat;
    const int ps = prof_begin(c, PROF_ILU_APPLY);
    const int C = P.numColors, mode = c->cfg.relax_mode;
    const double w = w_override > 0.0 ? w_override : c->cfg.ilu_relaxation;   // CPR's fine smoother runs with relaxation 1
    // post-scale with w != 1 keeps the unscaled sweep vector apart from the scaled result
    const bool post = mode == OPMHIP_RELAX_POST_SCALE && w != 1.0;
    // addp (CPR, w = 1): the sweeps run in `work`, v receives (0, addp, 0) + the result
    double* vu = addp ? work : (post && !unscaled) ? c->d_vu : v;
    if (unscaled) *unscaled = post ? w : 1.0;
    const int n0 = P.colorPrefix[1];  // rows of the first colour: their y is d
    auto grid = [](int n) { return dim3(8 * ((n + 7) / 8)); };
    if (P.chained) {
        // per colour: the descriptor records of its launch schedule (one per position, padding included) and their count
        const int ds = P.tiles.descStride, S1 = P.tiles.descS1;
        auto desc = [&](int col) { return P.tiles.d_ctDesc + (size_t)P.tiles.ctSchedOff[col] * ds; };
        auto npos = [&](int col) { return P.tiles.ctSchedOff[col + 1] - P.tiles.ctSchedOff[col]; };
        static const bool explicitIdx = [] { const char* e = tuning_env("OPMHIP_SWEEP_EXPLICIT"); return e && e[0] == '1'; }();   // A/B switch
        const bool st = P.sweepStencil && !explicitIdx;   // column indices and row bounds of the heavy sweeps from the stencil form
        const SweepStencil SL{P.d_swWord[0], P.d_swKoff[0], P.d_swTable[0]}, SU{P.d_swWord[1], P.d_swKoff[1], P.d_swTable[1]};
        for (int col = 0; col < C - 1; ++col) {
            const int nct = npos(col);
            if (nct <= 0) continue;
            if (P.lightL[col])
                hipLaunchKernelGGL((k_ilu_sweep_light<SW_L, false>), dim3(nct), dim3(64), 0, c->stream, desc(col), ds, S1, P.d_lrowptr,
                                   P.d_lcol, c->d_L, c->d_invD, d, vu, v, addp, mode, w, c->d_done, (double*)nullptr);
            else if (st)
                hipLaunchKernelGGL((k_ilu_sweep_chain<SW_L, true, false>), dim3(nct), dim3(64), 0, c->stream, desc(col), ds, S1, P.d_lrowptr,
                                   P.d_lcol, c->d_L, c->d_invD, d, vu, v, addp, mode, w, c->d_done, SL, (double*)nullptr);
            else
                hipLaunchKernelGGL((k_ilu_sweep_chain<SW_L, false, false>), dim3(nct), dim3(64), 0, c->stream, desc(col), ds, S1, P.d_lrowptr,
                                   P.d_lcol, c->d_L, c->d_invD, d, vu, v, addp, mode, w, c->d_done, SL, (double*)nullptr);
        }
        {
            const int nct = npos(C - 1);
            if (nct > 0) {
#define OPMHIP_LU_LAUNCH(LIGHT, STF, UAF)                                                                                                   \
    hipLaunchKernelGGL((k_ilu_sweep_chain_LU<LIGHT, STF, UAF>), dim3(nct), dim3(64), 0, c->stream, desc(C - 1), ds, S1, P.d_lrowptr, P.d_lcol, c->d_L, \
                       P.d_urowptr, P.d_ucol, c->d_U, c->d_invD, d, vu, v, addp, mode, w, c->d_done, SL, SU, usum)
                if (usum) {
                    if (P.lightU[C - 1]) { if (st) OPMHIP_LU_LAUNCH(true, true, true); else OPMHIP_LU_LAUNCH(true, false, true); }
                    else { if (st) OPMHIP_LU_LAUNCH(false, true, true); else OPMHIP_LU_LAUNCH(false, false, true); }
                } else {
                    if (P.lightU[C - 1]) { if (st) OPMHIP_LU_LAUNCH(true, true, false); else OPMHIP_LU_LAUNCH(true, false, false); }
                    else { if (st) OPMHIP_LU_LAUNCH(false, true, false); else OPMHIP_LU_LAUNCH(false, false, false); }
                }
#undef OPMHIP_LU_LAUNCH
            }
        }
        for (int col = C - 2; col >= 0; --col) {
            const int nct = npos(col);
            if (nct <= 0) continue;
            if (usum) {
                if (P.lightU[col])
                    hipLaunchKernelGGL((k_ilu_sweep_light<SW_UF, true>), dim3(nct), dim3(64), 0, c->stream, desc(col), ds, S1, P.d_urowptr,
                                       P.d_ucol, c->d_U, c->d_invD, d, vu, v, addp, mode, w, c->d_done, usum);
                else if (st)
                    hipLaunchKernelGGL((k_ilu_sweep_chain<SW_UF, true, true>), dim3(nct), dim3(64), 0, c->stream, desc(col), ds, S1, P.d_urowptr,
                                       P.d_ucol, c->d_U, c->d_invD, d, vu, v, addp, mode, w, c->d_done, SU, usum);
                else
                    hipLaunchKernelGGL((k_ilu_sweep_chain<SW_UF, false, true>), dim3(nct), dim3(64), 0, c->stream, desc(col), ds, S1, P.d_urowptr,
                                       P.d_ucol, c->d_U, c->d_invD, d, vu, v, addp, mode, w, c->d_done, SU, usum);
            } else if (P.lightU[col])
                hipLaunchKernelGGL((k_ilu_sweep_light<SW_UF, false>), dim3(nct), dim3(64), 0, c->stream, desc(col), ds, S1, P.d_urowptr,
                                   P.d_ucol, c->d_U, c->d_invD, d, vu, v, addp, mode, w, c->d_done, (double*)nullptr);
            else if (st)
                hipLaunchKernelGGL((k_ilu_sweep_chain<SW_UF, true, false>), dim3(nct), dim3(64), 0, c->stream, desc(col), ds, S1, P.d_urowptr,
                                   P.d_ucol, c->d_U, c->d_invD, d, vu, v, addp, mode, w, c->d_done, SU, (double*)nullptr);
            else
                hipLaunchKernelGGL((k_ilu_sweep_chain<SW_UF, false, false>), dim3(nct), dim3(64), 0, c->stream, desc(col), ds, S1, P.d_urowptr,
                                   P.d_ucol, c->d_U, c->d_invD, d, vu, v, addp, mode, w, c->d_done, SU, (double*)nullptr);
        }
        prof_end(c, ps);
        return;
    }
    for (int col = 1; col < C; ++col) {
        const int tb = P.tiles.colorTile[col], nt = P.tiles.colorTile[col + 1] - tb;
        if (nt <= 0) continue;
        if (col < C - 1)
            hipLaunchKernelGGL(k_ilu_sweep<SW_L>, grid(nt), dim3(64), 0, c->stream, tb, nt, n0, 0, P.tiles.d_row0, P.d_lrowptr, P.d_lcol, c->d_L,
                               c->d_invD, d, vu, v, addp, mode, w, c->d_done);
        else
            hipLaunchKernelGGL(k_ilu_sweep<SW_LF>, grid(nt), dim3(64), 0, c->stream, tb, nt, n0, 0, P.tiles.d_row0, P.d_lrowptr, P.d_lcol, c->d_L,
                               c->d_invD, d, vu, v, addp, mode, w, c->d_done);
    }
    for (int col = (C > 1 ? C - 2 : 0); col >= 0; --col) {
        const int tb = P.tiles.colorTile[col], nt = P.tiles.colorTile[col + 1] - tb;
        if (nt <= 0) continue;
        // gathers only reach later colours (>= n0 rows in), so the d/vu split of the gather is inert here (n0 = 0)
        hipLaunchKernelGGL(k_ilu_sweep<SW_UF>, grid(nt), dim3(64), 0, c->stream, tb, nt, 0, col == 0 ? 1 : 0, P.tiles.d_row0, P.d_urowptr, P.d_ucol,
                           c->d_U, c->d_invD, d, vu, v, addp, mode, w, c->d_done);
    }
    prof_end(c, ps);
}
// rb_half >= 0: this launch evaluates the stopping rule of half iteration rb_half and reports into its ring slot
static int finalize(opmhip_ctx* c, int mode, int count, int rb_half = -1) {
    const double tol = c->cfg.tolerance;
    double* hslot = nullptr;
    double seq = 0.0;
    if (rb_half >= 0) {
        const int sl = rb_half % opmhip_ctx::RB_SLOTS;
        hslot = c->d_ring + (size_t)sl * opmhip_ctx::RB_DOUBLES;
        seq = (c->rb_seq += 1.0);
        c->rb_want[sl] = seq;
    }
    if (c->comm.nranks > 1) {
        // local sums -> one small all-reduce -> scalars (the reference all-reduces one double per scalar product
        // through OwnerOverlapCopyCommunication; here the two sums of a half iteration travel together)
        // long partial lists: slice sums and their fixed-order total in ONE launch (the last workgroup to finish folds
        // the slices, as in the single-GPU path) instead of k_reduce_stage1 + k_local_sums
        const int span = prof_span_begin(c, PROF_ALLREDUCE);   // local sums -> all-reduce -> the sums are on the device
        if (count > RED1_SINGLE_MAX)
            hipLaunchKernelGGL(k_reduce_finalize, dim3(RED1_BLOCKS), dim3(VB), 0, c->stream, (int)FIN_LOCAL, count, c->d_part, c->npart, c->d_part2,
                               reinterpret_cast<unsigned*>(c->d_part2 + 2 * RED1_BLOCKS), c->comm.d_red, tol, (double*)nullptr, 0.0);
        else
            hipLaunchKernelGGL(k_local_sums, dim3(1), dim3(VB), 0, c->stream, count, c->d_part, c->npart, c->comm.d_red);
        // a failed all-reduce would leave garbage in alpha / omega / the norm: stop the solve, the caller reports it
        c->comm.reduce_span_open = true;
        const int rc = comm_allreduce(c, c->comm.d_red, 2, 0);
        c->comm.reduce_span_open = false;
        prof_span_end(c, span);
        if (rc) return rc;
        hipLaunchKernelGGL(k_finalize, dim3(1), dim3(VB), 0, c->stream, mode, 1, c->comm.d_red, 1, c->d_scal, tol, hslot, seq);
        return OPMHIP_SUCCESS;
    }
    if (count > RED1_SINGLE_MAX) {
        hipLaunchKernelGGL(k_reduce_finalize, dim3(RED1_BLOCKS), dim3(VB), 0, c->stream, mode, count, c->d_part, c->npart, c->d_part2,
                           reinterpret_cast<unsigned*>(c->d_part2 + 2 * RED1_BLOCKS), c->d_scal, tol, hslot, seq);
        return OPMHIP_SUCCESS;
    }
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(VB), 0, c->stream, mode, count, c->d_part, c->npart, c->d_scal, tol, hslot, seq);
    return OPMHIP_SUCCESS;
}
// opmhip_config.fused_reductions: the ONE reduction of a half iteration (three sums) and its scalar stage; decomposed runs: local sums ->
// one all-reduce of three doubles -> scalars
static int finalize3(opmhip_ctx* c, int mode, int count, int rb_half) {
    const double tol = c->cfg.tolerance;
    const int sl = rb_half % opmhip_ctx::RB_SLOTS;
    double* hslot = c->d_ring + (size_t)sl * opmhip_ctx::RB_DOUBLES;
    const double seq = (c->rb_seq += 1.0);
    c->rb_want[sl] = seq;
    if (c->comm.nranks > 1) {
        const int span = prof_span_begin(c, PROF_ALLREDUCE);
        hipLaunchKernelGGL(k_finalize3, dim3(1), dim3(VB), 0, c->stream, (int)FIN_LOCAL, count, c->d_part, c->npart, c->comm.d_red, tol, (double*)nullptr, 0.0, 0.0);
        c->comm.reduce_span_open = true;
        const int rc = comm_allreduce(c, c->comm.d_red, 3, 0);
        c->comm.reduce_span_open = false;
        prof_span_end(c, span);
        if (rc) return rc;
        hipLaunchKernelGGL(k_finalize3, dim3(1), dim3(VB), 0, c->stream, mode, 1, c->comm.d_red, 1, c->d_scal, tol, hslot, seq, (double)rb_half);
        return OPMHIP_SUCCESS;
    }
    hipLaunchKernelGGL(k_finalize3, dim3(1), dim3(VB), 0, c->stream, mode, count, c->d_part, c->npart, c->d_scal, tol, hslot, seq, (double)rb_half);
    return OPMHIP_SUCCESS;
}
// the vector kernels of one BiCGStab iteration, for opmhip_time_kernel
void launch_vector_kernels_once(opmhip_ctx* c) {
    const int n = c->pat.Nb * BS, nb = vec_blocks(n);
    hipLaunchKernelGGL(k_bicg_pupdate, dim3(nb), dim3(VB), 0, c->stream, n, c->d_scal, c->d_p, c->d_v, c->d_r);
    hipLaunchKernelGGL(k_bicg_upd1<false>, dim3(nb), dim3(VB), 0, c->stream, n, c->d_scal, c->d_r, c->d_v, c->d_part, c->npart, -1.0);
    (void)finalize(c, FIN_NORM, nb);  // timing helper, single-rank contexts only
    hipLaunchKernelGGL(k_bicg_upd2<false>, dim3(nb), dim3(VB), 0, c->stream, n, c->d_scal, c->d_x, c->d_pw, c->d_s, c->d_r, c->d_t, c->d_rw, c->d_part, c->npart, c->minv_scale, -1.0);
    (void)finalize(c, FIN_NORM_RHO, nb);
}

// ============================== BiCGStab driver ==========================================================
// Recurrence, half-iteration counter and stopping rule of bda/cusparseSolverBackend.cu:60-184.  Scalars stay on the
// device and so does the stopping rule (k_finalize): the host enqueues half iteration h + 1 BEFORE it looks at the
// outcome of half iteration h, so the queue never drains (the reference reads back seven scalars per iteration, each
// blocking, :78-158).  If h met the stopping rule, the kernels of h + 1 see the flag and return at once; their profile
// scopes are voided.  `it` counts exactly as in the reference: 0.5 per half iteration, maxit + 0.5 = not converged.
static int read_scalars(opmhip_ctx* c) {
    OPMHIP_HIP(c, hipMemcpyAsync(c->h_pinned, c->d_scal, SC_COUNT * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
    return OPMHIP_SUCCESS;
}
// part: HALF_ALL, or the half iteration in two pieces - HALF_PRECOND (the communication-free head: the p-update and the preconditioner
// application) and HALF_REST (operator with its halo exchange, scalar products with their all-reduces, updates).  A CPR whose pressure
// stage spans the ranks (cpr.gather.on) communicates INSIDE its application - three halo exchanges and an all-gather - so it belongs to
// the rest: enqueued ahead of the stopping rule it would cost every solve one round of collectives past its last half iteration.
enum { HALF_ALL = 0, HALF_PRECOND = 1, HALF_REST = 2 };
static int enqueue_half(opmhip_ctx* c, int h, int part = HALF_ALL) {
    const Pattern& P = c->pat;
    const int n = P.Nb * BS, nb = vec_blocks(n);
    // The p-update and the (r, x)-update are kernels of their own: riding in the first colour's light sweep of the
    // preconditioner application that follows them (round 1) they made that sweep 0.017 ms longer and the vector scopes as
    // much shorter - the same Newton iteration rate (76.5 / 78.1 against 78.0 / 76.3 its/s on one box).
    const bool cpr = use_cpr(c);
    // Pattern::ualias: the backward sweeps of the ILU0 application leave the upper part of the product that follows behind as row sums;
    // that product then streams the matrix without its U part (d_R, written by this solve's factorisation)
    const bool hp = !cpr && c->half_product;
    const bool sr = c->cfg.fused_reductions > 0;   // one reduction of three sums per half iteration (finalize_scalars3)
    const bool precTalks = cpr && c->cpr.gather.on;                  // the preconditioner posts collectives
    const bool head = part != HALF_REST, rest = part != HALF_PRECOND;
    const bool applyNow = part == HALF_ALL || (precTalks ? part == HALF_REST : part == HALF_PRECOND);
    int rc, ps;
    if ((h & 1) == 0) {  // first half: p, y = M^-1 p, v = A y, alpha, x += alpha y, r -= alpha v, |r|
        if (head && h > 0) {
            ps = prof_begin(c, PROF_VECTOR);
            hipLaunchKernelGGL(k_bicg_pupdate, dim3(nb), dim3(VB), 0, c->stream, n, c->d_scal, c->d_p, c->d_v, c->d_r);
            prof_end(c, ps);
        }
        if (applyNow) {
            if (cpr) launch_cpr_apply(c, c->d_p, c->d_pw);
            else launch_ilu_apply(c, c->d_p, c->d_pw, -1.0, &c->minv_scale, nullptr, nullptr, hp ? c->d_usum : nullptr);   // d_pw without the relaxation factor: its readers apply it
            if (cpr && c->cpr.apply_rc) { rc = c->cpr.apply_rc; c->cpr.apply_rc = 0; return rc; }   // a collective of the joined coarse level failed
        }
        if (!rest) return OPMHIP_SUCCESS;
        if (sr) {   // one reduction: v.rw, v.v, v.r ride in the product; alpha, the new norm and the stopping rule follow at once
            if ((rc = launch_spmv(c, c->d_pw, c->d_v, 3, c->d_rw, c->minv_scale, true, hp ? c->d_usum : nullptr, c->d_r))) return rc;
            ps = prof_begin(c, PROF_VECTOR);
            if ((rc = finalize3(c, FIN_SR1, dot_count(c), h))) return rc;
            hipLaunchKernelGGL(k_bicg_upd1<true>, dim3(nb), dim3(VB), 0, c->stream, n, c->d_scal, c->d_r, c->d_v, c->d_part, c->npart, (double)h);
            prof_end(c, ps);
            return OPMHIP_SUCCESS;
        }
        if ((rc = launch_spmv(c, c->d_pw, c->d_v, 1, c->d_rw, c->minv_scale, true, hp ? c->d_usum : nullptr))) return rc;  // with copyOwnerToAll before the operator (ParallelOverlappingILU0.hpp:897)
        ps = prof_begin(c, PROF_VECTOR);
        if ((rc = finalize(c, FIN_ALPHA, dot_count(c)))) return rc;
        hipLaunchKernelGGL(k_bicg_upd1<false>, dim3(nb), dim3(VB), 0, c->stream, n, c->d_scal, c->d_r, c->d_v, c->d_part, c->npart, -1.0);
        if ((rc = finalize(c, FIN_NORM, nb, h))) return rc;
        prof_end(c, ps);
    } else {             // second half: z = M^-1 r, t = A z, omega, x += omega z, r -= omega t, |r|, rho, beta
        if (applyNow) {
            if (cpr) launch_cpr_apply(c, c->d_r, c->d_s);
            else launch_ilu_apply(c, c->d_r, c->d_s, -1.0, &c->minv_scale, nullptr, nullptr, hp ? c->d_usum : nullptr);
            if (cpr && c->cpr.apply_rc) { rc = c->cpr.apply_rc; c->cpr.apply_rc = 0; return rc; }
        }
        if (!rest) return OPMHIP_SUCCESS;
        if (sr) {   // t.r, t.t, t.rw: omega, the new norm, rho and beta from one reduction
            if ((rc = launch_spmv(c, c->d_s, c->d_t, 3, c->d_r, c->minv_scale, true, hp ? c->d_usum : nullptr, c->d_rw))) return rc;
            ps = prof_begin(c, PROF_VECTOR);
            if ((rc = finalize3(c, FIN_SR2, dot_count(c), h))) return rc;
            hipLaunchKernelGGL(k_bicg_upd2<true>, dim3(nb), dim3(VB), 0, c->stream, n, c->d_scal, c->d_x, c->d_pw, c->d_s, c->d_r, c->d_t, c->d_rw, c->d_part, c->npart, c->minv_scale, (double)h);
            prof_end(c, ps);
            return OPMHIP_SUCCESS;
        }
        if ((rc = launch_spmv(c, c->d_s, c->d_t, 2, c->d_r, c->minv_scale, true, hp ? c->d_usum : nullptr))) return rc;
        ps = prof_begin(c, PROF_VECTOR);
        if ((rc = finalize(c, FIN_OMEGA, dot_count(c)))) return rc;
        hipLaunchKernelGGL(k_bicg_upd2<false>, dim3(nb), dim3(VB), 0, c->stream, n, c->d_scal, c->d_x, c->d_pw, c->d_s, c->d_r, c->d_t, c->d_rw, c->d_part, c->npart, c->minv_scale, -1.0);
        if ((rc = finalize(c, FIN_NORM_RHO, nb, h))) return rc;
        prof_end(c, ps);
    }
    return OPMHIP_SUCCESS;
}
// the host's side of the read-back ring: spin on the sequence number of half iteration h's slot
static int wait_half(opmhip_ctx* c, int h, double* norm, double* norm_0) {
    const int sl = h % opmhip_ctx::RB_SLOTS;
    const volatile double* slot = c->h_ring + (size_t)sl * opmhip_ctx::RB_DOUBLES;
    const double want = c->rb_want[sl];
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned long spin = 1;; ++spin) {
        if (slot[3] == want) break;
        if ((spin & 0xFFFFu) == 0) {  // every ~65k polls: is the stream still alive?
            const hipError_t e = hipStreamQuery(c->stream);
            if (e == hipSuccess) {
                if (slot[3] == want) break;
                return fail(c, OPMHIP_DEVICE_ERROR, "BiCGStab: the stream drained without the stopping-rule record of half iteration %d", h);
            }
            if (e != hipErrorNotReady) return fail(c, OPMHIP_DEVICE_ERROR, "BiCGStab: %s", hipGetErrorString(e));
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120))
                return fail(c, OPMHIP_DEVICE_ERROR, "BiCGStab: no stopping-rule record of half iteration %d after 120 s", h);
        }
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#elif defined(__aarch64__)
        asm volatile("yield" ::: "memory");
#endif
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    *norm = slot[0];
    *norm_0 = slot[1];
    return OPMHIP_SUCCESS;
}
int bicgstab(opmhip_ctx* c, opmhip_result* res) {
    const Pattern& P = c->pat;
    const int n = P.Nb * BS, nb = vec_blocks(n);
    const int maxit = c->cfg.maxit;
    const double tol = c->cfg.tolerance;
    struct SolveScope {  // the tile kernels watch the solve's flag only while the solve runs; profile scopes share events
        opmhip_ctx* c;
        explicit SolveScope(opmhip_ctx* c_) : c(c_) {
            c->d_done = c->d_scal + SC_DONE;
            Profiler& Q = c->prof;
            Q.solve_no += 1;
            Q.suspended = Q.enabled && Q.every > 1 && (Q.solve_no % Q.every) != 0;
            Q.lazy = c->comm.nranks == 1;  // decomposed runs: halo exchanges sit between the scopes, keep them out
        }
        ~SolveScope() {
            c->d_done = c->d_scal + SC_ZERO;
            c->prof.lazy = false;
            prof_flush(c);
            c->prof.suspended = false;
        }
    } scope(c);
    int rc;
    c->minv_scale = 1.0;   // CPR delivers finished vectors; the ILU0 application sets its factor
    hipLaunchKernelGGL(k_bicg_init, dim3(nb), dim3(VB), 0, c->stream, n, c->d_b, c->d_r, c->d_rw, c->d_p, c->d_x, c->d_v, c->d_part, c->npart);
    if ((rc = finalize(c, FIN_INIT, nb))) return rc;
    // the reference's loop "for (it = 0.5; it < maxit; it += 0.5) { first half; it += 0.5; second half }" runs the half
    // iterations h = 0 .. 2 maxit - 1 with it = (h + 1) / 2 after half h
    const int nhalves = 2 * maxit;
    double norm = 0.0, norm_0 = 0.0;
    float it = maxit + 0.5f;
    size_t mark_next = c->prof.used;  // first profile scope of the half iteration in flight beyond the one being waited for
    // Decomposed runs: only the communication-free head of half iteration h + 1 (p-update, preconditioner application: 0.14 ms and
    // more of device work, enough to cover the host's look at h) is enqueued ahead of the stopping rule of h; its operator, with
    // the halo exchange, and its all-reduces follow once the (all-reduced, hence rank-independent) norm says the solve goes on.  A
    // solve that stops therefore posts NO collective beyond the stopping point - enqueueing the whole half ahead cost every solve
    // two halo exchanges and two all-reduces on a latency-bound path (round-3 review).  With a CPR that spans the ranks the
    // head shrinks to the p-update (enqueue_half: precTalks): the device then idles for the host's look at h in every half iteration -
    // the price of not posting that application's three halo exchanges and its all-gather once too often per solve.
    const bool split = c->comm.nranks > 1;
    if (nhalves > 0 && (rc = enqueue_half(c, 0))) return rc;
    for (int h = 0; h < nhalves; ++h) {
        mark_next = c->prof.used;
        if (h + 1 < nhalves && (rc = enqueue_half(c, h + 1, split ? HALF_PRECOND : HALF_ALL))) return rc;
        if ((rc = wait_half(c, h, &norm, &norm_0))) return rc;
        if (!std::isfinite(norm)) {  // a singular pivot: nothing can converge any more (the caller reports it)
            prof_flush(c);
            break;
        }
        if (norm < tol * norm_0) {
            it = 0.5f * (float)(h + 1);
            prof_flush(c);
            for (size_t i = mark_next; i < c->prof.used; ++i) c->prof.cls[i] = -1;  // half h + 1 ran as no-ops
            // stopped on a first half: its update of x was left to the second half, which will not come
            if ((h & 1) == 0) hipLaunchKernelGGL(k_bicg_xhalf, dim3(nb), dim3(VB), 0, c->stream, n, c->d_scal, c->d_x, c->d_pw, c->minv_scale);
            break;
        }
        if (split && h + 1 < nhalves && (rc = enqueue_half(c, h + 1, HALF_REST))) return rc;
    }
    if (nhalves <= 0) {
        if ((rc = read_scalars(c))) return rc;
        norm = norm_0 = c->h_pinned[SC_NORM0];
    }
    bool drifted = false;
    if (c->cfg.fused_reductions > 0 && nhalves > 0 && std::isfinite(norm)) {
        // the norms above came out of a recurrence (finalize_scalars3); the residual VECTOR was updated explicitly all along: its own norm,
        // once per solve (one more reduction), is what is reported - and a solve whose recurred norm met the tolerance while the vector's is
        // more than twice it is reported as not converged (the caller then falls back as after any failure, linalg/ISTLSolverEbos.hpp:277-289)
        hipLaunchKernelGGL(k_dots, dim3(nb), dim3(VB), 0, c->stream, n, c->d_r, c->d_r, c->d_part, c->npart, 1);
        if ((rc = finalize(c, FIN_TRUE, nb))) return rc;
        if ((rc = read_scalars(c))) return rc;
        const double truth = c->h_pinned[SC_TMP1];
        drifted = it != (maxit + 0.5f) && !(truth < 2.0 * tol * norm_0);
        norm = truth;
    }
    OPMHIP_HIP(c, hipGetLastError());
    res->it = it;
    res->iterations = (int)std::fmin(it, (float)maxit);
    res->reduction = norm / norm_0;
    res->conv_rate = std::pow(res->reduction, 1.0 / it);
    res->converged = (it != (maxit + 0.5f)) && !drifted;
    return OPMHIP_SUCCESS;
}

}  // namespace opmhip
