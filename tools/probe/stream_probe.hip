// tools/probe/stream_probe.hip - what HBM read bandwidth does this card give to different access shapes?
// Development probe (not product code): decides how the tile kernels should issue their loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void read_gs(size_t n2, const double2* __restrict__ src, double* sink) {
    double s = 0.0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < n2; i += 8 * stride) {
        double2 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += t[u].x + t[u].y;
    }
    if (s == 1.2345e300) sink[0] = s;
}
__global__ __launch_bounds__(256) void read_piece(size_t n2, const double2* __restrict__ src, double* sink) {
    double s = 0.0;
    constexpr size_t PIECE = 256 * 8;
    const size_t npiece = n2 / PIECE;
    for (size_t p = blockIdx.x; p < npiece; p += gridDim.x) {
        const double2* q = src + p * PIECE + threadIdx.x;
        double2 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = q[u * 256];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += t[u].x + t[u].y;
    }
    if (s == 1.2345e300) sink[0] = s;
}
// one wave per workgroup, one contiguous 16 KiB tile per workgroup (the SpMV's shape), U loads of 16 B in flight per lane
template <int U, bool LDS, bool NT>
__global__ __launch_bounds__(64) void read_tile(size_t ntiles, const double2* __restrict__ src, double* sink) {
    __shared__ double2 sh[LDS ? 64 * U : 1];
    const size_t t = blockIdx.x;
    if (t >= ntiles) return;
    const double2* q = src + t * (64 * U) + threadIdx.x;
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (NT) { v[u].x = __builtin_nontemporal_load(&q[u * 64].x); v[u].y = __builtin_nontemporal_load(&q[u * 64].y); }
        else v[u] = q[u * 64];
    }
    double s = 0.0;
    if (LDS) {
#pragma unroll
        for (int u = 0; u < U; ++u) sh[u * 64 + threadIdx.x] = v[u];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < U; ++u) { const double2 w = sh[u * 64 + (threadIdx.x ^ 1)]; s += w.x + w.y; }
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u) s += v[u].x + v[u].y;
    }
    if (s == 1.2345e300) sink[0] = s;
}
// persistent: W waves per workgroup... one wave per workgroup, G workgroups, each loops over tiles with the next tile's
// loads issued before the current one is consumed
template <int U>
__global__ __launch_bounds__(64) void read_persist(size_t ntiles, const double2* __restrict__ src, double* sink) {
    double s = 0.0;
    size_t t = blockIdx.x;
    if (t >= ntiles) return;
    double2 v[U], w[U];
    const double2* q = src + t * (64 * U) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = q[u * 64];
    for (t += gridDim.x; t < ntiles; t += gridDim.x) {
        q = src + t * (64 * U) + threadIdx.x;
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = q[u * 64];
#pragma unroll
        for (int u = 0; u < U; ++u) s += v[u].x + v[u].y;
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = w[u];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) s += v[u].x + v[u].y;
    if (s == 1.2345e300) sink[0] = s;
}
__global__ __launch_bounds__(256) void copy_piece(size_t n2, const double2* __restrict__ src, double2* __restrict__ dst) {
    constexpr size_t PIECE = 256 * 8;
    const size_t npiece = n2 / PIECE;
    for (size_t p = blockIdx.x; p < npiece; p += gridDim.x) {
        const double2* q = src + p * PIECE + threadIdx.x;
        double2 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = q[u * 256];
#pragma unroll
        for (int u = 0; u < 8; ++u) dst[p * PIECE + threadIdx.x + u * 256] = t[u];
    }
}

int main(int argc, char** argv) {
    const size_t bytes = (argc > 1 ? atol(argv[1]) : 500) * 1000000ul;
    const size_t n2 = bytes / 16;
    double2 *a, *b; double* sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipStream_t st; CK(hipStreamCreate(&st));
    auto timeit = [&](const char* name, auto launch, double bytes_per) {
        for (int r = 0; r < 2; ++r) { launch(a); launch(b); }
        CK(hipStreamSynchronize(st));
        const int reps = 10;
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) { launch(a); launch(b); }   // alternate two buffers: nothing survives in the Infinity Cache
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double per = ms / (2 * reps);
        printf("%-44s %8.2f us  %6.2f TB/s\n", name, per * 1e3, bytes_per / per / 1e9);
        fflush(stdout);
    };
    for (int g : {1024, 2048, 4096, 8192})
        timeit(("read_gs 256thr grid " + std::to_string(g)).c_str(), [&](double2* p) { hipLaunchKernelGGL(read_gs, dim3(g), dim3(256), 0, st, n2, p, sink); }, (double)bytes);
    for (int g : {1024, 2048, 4096, 8192, 16384})
        timeit(("read_piece 256thr grid " + std::to_string(g)).c_str(), [&](double2* p) { hipLaunchKernelGGL(read_piece, dim3(g), dim3(256), 0, st, n2, p, sink); }, (double)bytes);
    {
        const size_t nt16 = n2 / (64 * 16), nt8 = n2 / (64 * 8), nt32 = n2 / (64 * 32);
        timeit("read_tile U=16 (16 KiB/wave) regs", [&](double2* p) { hipLaunchKernelGGL((read_tile<16, false, false>), dim3(nt16), dim3(64), 0, st, nt16, p, sink); }, (double)bytes);
        timeit("read_tile U=16 via LDS", [&](double2* p) { hipLaunchKernelGGL((read_tile<16, true, false>), dim3(nt16), dim3(64), 0, st, nt16, p, sink); }, (double)bytes);
        timeit("read_tile U=16 regs nontemporal", [&](double2* p) { hipLaunchKernelGGL((read_tile<16, false, true>), dim3(nt16), dim3(64), 0, st, nt16, p, sink); }, (double)bytes);
        timeit("read_tile U=8 (8 KiB/wave) regs", [&](double2* p) { hipLaunchKernelGGL((read_tile<8, false, false>), dim3(nt8), dim3(64), 0, st, nt8, p, sink); }, (double)bytes);
        timeit("read_tile U=32 (32 KiB/wave) regs", [&](double2* p) { hipLaunchKernelGGL((read_tile<32, false, false>), dim3(nt32), dim3(64), 0, st, nt32, p, sink); }, (double)bytes);
        for (int g : {2048, 4096, 8192})
            timeit(("read_persist U=16 grid " + std::to_string(g)).c_str(), [&](double2* p) { hipLaunchKernelGGL((read_persist<16>), dim3(g), dim3(64), 0, st, nt16, p, sink); }, (double)bytes);
        for (int g : {2048, 4096, 8192})
            timeit(("read_persist U=8 grid " + std::to_string(g)).c_str(), [&](double2* p) { hipLaunchKernelGGL((read_persist<8>), dim3(g), dim3(64), 0, st, nt8, p, sink); }, (double)bytes);
    }
    for (int g : {2048, 4096})
        timeit(("copy_piece grid " + std::to_string(g) + " (read+write bytes)").c_str(), [&](double2* p) { hipLaunchKernelGGL(copy_piece, dim3(g), dim3(256), 0, st, n2, p, p == a ? b : a); }, 2.0 * bytes);
    return 0;
}
