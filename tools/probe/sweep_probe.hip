// tools/probe/sweep_probe.hip - the access shape of the heavy chain sweeps, stream only: `nct` one-wavefront workgroups, each
// walking `nsteps` dependent steps of one contiguous 12-KiB chunk (a step's factor blocks), the result of a step feeding
// the next.  How fast can that shape be read, and does a deeper prefetch through LDS-DMA (global_load_lds, no registers)
// buy anything?  Development probe (not product code); numbers in DESIGN.md section 4.
//   reg1   : what csrc/solver.hip's chain_sweep does - step s+1 into registers while step s is consumed out of LDS
//   dma<N> : N LDS buffers per wavefront, steps s+1 .. s+N-1 in flight by LDS-DMA while step s is consumed
//   "+ both, coherent": what ONE launch for all colours of an ILU0 application would have to do instead of three (review item 3 of
//            round 2): the gathered vector entries were written by other workgroups of the same launch, so they are read with
//            agent-scope (sc1) loads and written with sc1 stores, and every workgroup polls one flag word per step first
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int STEP2 = 768;   // double2 per step: 12 KiB (a 7-point chain step is 11.25 KiB)
typedef double v2d_t __attribute__((ext_vector_type(2)));

// what a step's arithmetic looks like to the memory system: lanes 0..31 read "their row's" 5 blocks (45 doubles) out of the
// LDS image and fold them into a value that depends on the previous step's
__device__ __forceinline__ double consume(const double* img, int lane, double carry) {
    double s = carry * 1e-3;
    if (lane < 32) {
        const double* row = img + lane * 45;
#pragma unroll
        for (int q = 0; q < 45; ++q) s += row[q];
    }
    return s;
}

__global__ __launch_bounds__(64) void sweep_reg1(int nsteps, const double2* __restrict__ src, double* sink) {
    __shared__ __attribute__((aligned(16))) double2 img[STEP2];
    const int lane = threadIdx.x;
    const double2* base = src + (size_t)blockIdx.x * nsteps * STEP2;
    double2 tmp[12];
#pragma unroll
    for (int u = 0; u < 12; ++u) {
        const v2d_t t = __builtin_nontemporal_load(reinterpret_cast<const v2d_t*>(&base[u * 64 + lane]));
        tmp[u] = make_double2(t.x, t.y);
    }
    double carry = 0.0;
    for (int s = 0; s < nsteps; ++s) {
#pragma unroll
        for (int u = 0; u < 12; ++u) img[u * 64 + lane] = tmp[u];
        asm volatile("" ::: "memory");
        if (s + 1 < nsteps) {
            const double2* nx = base + (size_t)(s + 1) * STEP2;
#pragma unroll
            for (int u = 0; u < 12; ++u) {
                const v2d_t t = __builtin_nontemporal_load(reinterpret_cast<const v2d_t*>(&nx[u * 64 + lane]));
                tmp[u] = make_double2(t.x, t.y);
            }
        }
        carry = consume(reinterpret_cast<const double*>(img), lane, carry);
        asm volatile("" ::: "memory");
    }
    if (carry == 1.2345e300) sink[0] = carry;
}


// reg1 plus the small accesses of the real sweep: per step and lane (32 rows) NCOL gathered vector entries of 3 doubles, a
// right-hand side of 3 and a D^-1 block of 9 - as 8-byte loads (MODE 0: 3 + 3 + 9 instructions per column / row, what
// chain_sweep issues) or as 16 + 8 / 4 x 16 + 8 byte loads (MODE 1: the same bytes in fewer instructions)
template <int NCOL, int MODE, int FLAGS = 0>
__global__ __launch_bounds__(64) void sweep_reg1_small(int nsteps, int nrows, const double2* __restrict__ src, const double* __restrict__ vec,
                                                       const double* __restrict__ dinv, double* sink, const int* __restrict__ idx = nullptr, double* __restrict__ outv = nullptr) {
    __shared__ __attribute__((aligned(16))) double2 img[STEP2];
    const int lane = threadIdx.x;
    const double2* base = src + (size_t)blockIdx.x * nsteps * STEP2;
    double2 tmp[12];
#pragma unroll
    for (int u = 0; u < 12; ++u) {
        const v2d_t t = __builtin_nontemporal_load(reinterpret_cast<const v2d_t*>(&base[u * 64 + lane]));
        tmp[u] = make_double2(t.x, t.y);
    }
    double carry = 0.0;
    double g[NCOL + 1][3], D[9];
    auto small = [&](int s) {
        // rows of a step are consecutive; column u of a row lies a fixed, far offset away (another chain-tile)
        const int row = ((blockIdx.x * nsteps + s) * 32 + (lane & 31)) % nrows;
        int colv[NCOL + 1];
#pragma unroll
        for (int u = 0; u <= NCOL; ++u) colv[u] = (row + u * 40009) % nrows;
        if (FLAGS & 1) {   // the column of every gather comes out of memory first (two dependent rounds: row bounds, columns)
            const int kb = idx[row], ke = idx[row + 1];
#pragma unroll
            for (int u = 0; u <= NCOL; ++u) colv[u] = (colv[u] + idx[(kb + u) % nrows] + (ke & 0)) % nrows;
        }
#pragma unroll
        for (int u = 0; u <= NCOL; ++u) {
            const size_t c = (size_t)colv[u] * 3;
            if (FLAGS & 4) {
                g[u][0] = __hip_atomic_load(vec + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                g[u][1] = __hip_atomic_load(vec + c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                g[u][2] = __hip_atomic_load(vec + c + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (MODE == 0) { g[u][0] = vec[c]; g[u][1] = vec[c + 1]; g[u][2] = vec[c + 2]; }
            else {
                double2 a; double b;
                __builtin_memcpy(&a, vec + c, 16); b = vec[c + 2];
                g[u][0] = a.x; g[u][1] = a.y; g[u][2] = b;
            }
        }
        const size_t d = (size_t)row * 9;
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 9; ++q) D[q] = dinv[d + q];
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) { double2 a; __builtin_memcpy(&a, dinv + d + 2 * q, 16); D[2 * q] = a.x; D[2 * q + 1] = a.y; }
            D[8] = dinv[d + 8];
        }
    };
    if (FLAGS & 8) {   // a dependency flag of another workgroup, already set: the price of looking, not of waiting
        while (__hip_atomic_load(idx + (blockIdx.x * 13) % nrows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {}
    }
    small(0);
    for (int s = 0; s < nsteps; ++s) {
#pragma unroll
        for (int u = 0; u < 12; ++u) img[u * 64 + lane] = tmp[u];
        double acc = 0.0;
#pragma unroll
        for (int u = 0; u <= NCOL; ++u) acc += g[u][0] + g[u][1] + g[u][2];
#pragma unroll
        for (int q = 0; q < 9; ++q) acc += D[q];
        asm volatile("" ::: "memory");
        if (s + 1 < nsteps) {
            const double2* nx = base + (size_t)(s + 1) * STEP2;
#pragma unroll
            for (int u = 0; u < 12; ++u) {
                const v2d_t t = __builtin_nontemporal_load(reinterpret_cast<const v2d_t*>(&nx[u * 64 + lane]));
                tmp[u] = make_double2(t.x, t.y);
            }
            small(s + 1);
        }
        carry = consume(reinterpret_cast<const double*>(img), lane, carry) + acc;
        if ((FLAGS & 2) && lane < 32) {   // the step's results: two vectors of 3 doubles per row
            const size_t o = (size_t)(((blockIdx.x * nsteps + s) * 32 + lane) % nrows) * 3;
            if (FLAGS & 4) {
                for (int q = 0; q < 3; ++q) {
                    __hip_atomic_store(outv + o + q, carry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(outv + (size_t)nrows * 3 + o + q, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                outv[o] = carry; outv[o + 1] = carry; outv[o + 2] = carry;
                outv[(size_t)nrows * 3 + o] = acc; outv[(size_t)nrows * 3 + o + 1] = acc; outv[(size_t)nrows * 3 + o + 2] = acc;
            }
        }
        asm volatile("" ::: "memory");
    }
    if (carry == 1.2345e300) sink[0] = carry;
}

template <int NBUF, int AUX>
__device__ __forceinline__ void dma_step(const double2* g, double2* buf, int lane) {
#pragma unroll
    for (int u = 0; u < 12; ++u) __builtin_amdgcn_global_load_lds(g + u * 64 + lane, buf + u * 64, 16, 0, AUX);
}
template <int NBUF> __device__ __forceinline__ void wait_oldest();
template <> __device__ __forceinline__ void wait_oldest<2>() { asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_oldest<3>() { asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_oldest<4>() { asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); }

template <int NBUF, int AUX>
__global__ __launch_bounds__(64) void sweep_dma(int nsteps, const double2* __restrict__ src, double* sink) {
    __shared__ __attribute__((aligned(16))) double2 img[NBUF][STEP2];
    const int lane = threadIdx.x;
    const double2* base = src + (size_t)blockIdx.x * nsteps * STEP2;
    // NBUF - 1 steps ahead; past the end the last step is fetched again so that the wait counts stay uniform
    for (int p = 0; p < NBUF - 1; ++p) dma_step<NBUF, AUX>(base + (size_t)(p < nsteps ? p : nsteps - 1) * STEP2, img[p % NBUF], lane);
    double carry = 0.0;
    for (int s = 0; s < nsteps; ++s) {
        const int nx = s + NBUF - 1;
        dma_step<NBUF, AUX>(base + (size_t)(nx < nsteps ? nx : nsteps - 1) * STEP2, img[nx % NBUF], lane);
        wait_oldest<NBUF>();   // step s has landed; NBUF - 1 younger ones are still flying
        carry = consume(reinterpret_cast<const double*>(img[s % NBUF]), lane, carry);
        asm volatile("" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (carry == 1.2345e300) sink[0] = carry;
}

int main(int argc, char** argv) {
    const int nsteps = 8;
    const int ncts[] = {1953, 2604, 3906};
    const size_t maxct = 3906;
    const size_t n2 = maxct * nsteps * STEP2;
    double2 *a, *b;
    double* sink;
    CK(hipMalloc(&a, n2 * sizeof(double2)));
    CK(hipMalloc(&b, n2 * sizeof(double2)));
    CK(hipMalloc(&sink, 64));
    const int nrows = 1000000;
    double *vec, *dinv;
    CK(hipMalloc(&vec, (size_t)nrows * 3 * 8 + 64));
    CK(hipMalloc(&dinv, (size_t)nrows * 9 * 8 + 64));
    CK(hipMemset(vec, 0, (size_t)nrows * 3 * 8 + 64));
    CK(hipMemset(dinv, 0, (size_t)nrows * 9 * 8 + 64));
    int* idx;
    double* outv;
    CK(hipMalloc(&idx, (size_t)(nrows + 8) * 4));
    CK(hipMemset(idx, 0, (size_t)(nrows + 8) * 4));
    CK(hipMalloc(&outv, (size_t)nrows * 6 * 8 + 64));
    CK(hipMemset(a, 0, n2 * sizeof(double2)));
    CK(hipMemset(b, 0, n2 * sizeof(double2)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](const char* name, int nct, auto launch) {
        for (int w = 0; w < 3; ++w) launch(w & 1 ? a : b);
        CK(hipDeviceSynchronize());
        const int reps = 20;
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch(r & 1 ? a : b);   // two buffers alternate: 2 x 190 .. 380 MB, nothing survives in the Infinity Cache at 1953 x 2
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = (double)nct * nsteps * STEP2 * 16.0;
        printf("%-20s %5d workgroups  %8.1f us  %7.1f GB/s\n", name, nct, 1e3 * ms / reps, bytes * reps / (ms * 1e6));
    };
    for (int nct : ncts) {
        run("reg1", nct, [&](const double2* p) { hipLaunchKernelGGL(sweep_reg1, dim3(nct), dim3(64), 0, 0, nsteps, p, sink); });
        run("reg1+small 8B", nct, [&](const double2* p) { hipLaunchKernelGGL((sweep_reg1_small<5, 0>), dim3(nct), dim3(64), 0, 0, nsteps, nrows, p, vec, dinv, sink); });
        run("reg1+small 16B", nct, [&](const double2* p) { hipLaunchKernelGGL((sweep_reg1_small<5, 1>), dim3(nct), dim3(64), 0, 0, nsteps, nrows, p, vec, dinv, sink); });
        run("  + indices", nct, [&](const double2* p) { hipLaunchKernelGGL((sweep_reg1_small<5, 0, 1>), dim3(nct), dim3(64), 0, 0, nsteps, nrows, p, vec, dinv, sink, idx, outv); });
        run("  + stores", nct, [&](const double2* p) { hipLaunchKernelGGL((sweep_reg1_small<5, 0, 2>), dim3(nct), dim3(64), 0, 0, nsteps, nrows, p, vec, dinv, sink, idx, outv); });
        run("  + both", nct, [&](const double2* p) { hipLaunchKernelGGL((sweep_reg1_small<5, 0, 3>), dim3(nct), dim3(64), 0, 0, nsteps, nrows, p, vec, dinv, sink, idx, outv); });
        run("  + both, coherent", nct, [&](const double2* p) { hipLaunchKernelGGL((sweep_reg1_small<5, 0, 3 | 4>), dim3(nct), dim3(64), 0, 0, nsteps, nrows, p, vec, dinv, sink, idx, outv); });
        run("  + both, coh.+flag", nct, [&](const double2* p) { hipLaunchKernelGGL((sweep_reg1_small<5, 0, 3 | 4 | 8>), dim3(nct), dim3(64), 0, 0, nsteps, nrows, p, vec, dinv, sink, idx, outv); });
        // three launches of a third of the work each against one launch of all of it: what the launch boundaries cost in this shape
        if (nct == 1953) {
            run("3 x (651) + both", nct, [&](const double2* p) { for (int q = 0; q < 3; ++q) hipLaunchKernelGGL((sweep_reg1_small<5, 0, 3>), dim3(651), dim3(64), 0, 0, nsteps, nrows, p + (size_t)q * 651 * nsteps * STEP2, vec, dinv, sink, idx, outv); });
        }
        run("dma<2>", nct, [&](const double2* p) { hipLaunchKernelGGL((sweep_dma<2, 0>), dim3(nct), dim3(64), 0, 0, nsteps, p, sink); });
        run("dma<2> nt", nct, [&](const double2* p) { hipLaunchKernelGGL((sweep_dma<2, 2>), dim3(nct), dim3(64), 0, 0, nsteps, p, sink); });
        run("dma<3> nt", nct, [&](const double2* p) { hipLaunchKernelGGL((sweep_dma<3, 2>), dim3(nct), dim3(64), 0, 0, nsteps, p, sink); });
        run("dma<4> nt", nct, [&](const double2* p) { hipLaunchKernelGGL((sweep_dma<4, 2>), dim3(nct), dim3(64), 0, 0, nsteps, p, sink); });
    }
    CK(hipGetLastError());
    return 0;
}
