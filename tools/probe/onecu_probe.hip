// tools/probe/onecu_probe.hip - what ONE workgroup of 1024 threads (one CU) can do per microsecond: the measurement behind the
// question "do the small levels of the CPR V-cycle belong in one launch of one workgroup?" (DESIGN.md section 5b).
// Development probe (not product code).
//   (a) the floor of a launch: an empty kernel, back to back;
//   (b) a phase of a single-workgroup cycle: one workgroup streams B bytes that sit in L2 (read twice before timing), 16-byte loads,
//       8 in flight per lane - the rate at which one CU takes a level's matrix in;
//   (c) the same bytes by a grid of 256-thread workgroups, one per 4 KiB (what the per-level kernels of today do);
//   (d) P phases of B bytes inside ONE launch of one workgroup, __syncthreads() between them, against P launches of (c).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_empty() {}
__global__ __launch_bounds__(1024) void k_one_wg(size_t n2, int phases, const double2* __restrict__ src, double* sink) {
    double s = 0.0;
    for (int p = 0; p < phases; ++p) {
        size_t i = threadIdx.x;
        for (; i + 7 * 1024 < n2; i += 8 * 1024) {
            double2 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = src[i + u * 1024];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += t[u].x + t[u].y;
        }
        for (; i < n2; i += 1024) s += src[i].x + src[i].y;
        __syncthreads();
    }
    if (s == 1.2345e300) sink[0] = s;
}
__global__ __launch_bounds__(256) void k_grid(size_t n2, const double2* __restrict__ src, double* sink) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    double s = 0.0;
    if (i < n2) s = src[i].x + src[i].y;
    if (s == 1.2345e300) sink[0] = s;
}
int main() {
    const size_t cap = 8u << 20;
    double2* d; double* sink;
    CK(hipMalloc(&d, cap)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(d, 0, cap));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto fn, int reps) {
        fn(); fn();
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) fn();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return 1e3 * ms / reps;
    };
    printf("(a) empty launch, back to back: %.2f us\n", time([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0); }, 200));
    for (size_t kb : {64, 128, 256, 512, 1024, 2048, 4096}) {
        const size_t n2 = kb * 1024 / 16;
        const double one = time([&] { hipLaunchKernelGGL(k_one_wg, dim3(1), dim3(1024), 0, 0, n2, 1, d, sink); }, 100);
        const double grid = time([&] { hipLaunchKernelGGL(k_grid, dim3((n2 + 255) / 256), dim3(256), 0, 0, n2, d, sink); }, 100);
        const double p8 = time([&] { hipLaunchKernelGGL(k_one_wg, dim3(1), dim3(1024), 0, 0, n2, 8, d, sink); }, 50);
        const double g8 = time([&] { for (int p = 0; p < 8; ++p) hipLaunchKernelGGL(k_grid, dim3((n2 + 255) / 256), dim3(256), 0, 0, n2, d, sink); }, 50);
        printf("%5zu KiB: one workgroup %.2f us (%.0f GB/s)   grid %.2f us   | 8 phases in one workgroup %.2f us (%.2f us per phase beyond the first launch)   8 grid launches %.2f us\n",
               kb, one, kb * 1.024 / one, grid, p8, (p8 - one) / 7, g8);
    }
    return 0;
}
