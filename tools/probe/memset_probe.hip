// Is hipMemset (the NULL stream) complete when it returns?  And is a kernel on a non-blocking stream ordered behind it?
// hipcc --offload-arch=gfx950 -O2 tools/probe/memset_probe.hip -o tools/probe/bin/memset_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_fill(double* p, size_t n, double v) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t big = (size_t)1 << 31;   // 2 GiB: a fill of ~0.5 ms at least
    char* a = nullptr;
    CK(hipMalloc(&a, big));
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; ++rep) {
        const double t0 = now();
        CK(hipMemset(a, 0, big));
        const double t1 = now();
        CK(hipDeviceSynchronize());
        const double t2 = now();
        std::printf("hipMemset of 2 GiB: the call %.3f ms, the synchronise behind it %.3f ms\n", 1e3 * (t1 - t0), 1e3 * (t2 - t1));
    }
    // a small array: NULL-stream fill with zeros, then at once a kernel on a non-blocking stream that writes ones; which one is last?
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t n = 1 << 16;
    double* v = nullptr;
    CK(hipMalloc(&v, n * sizeof(double)));
    std::vector<double> h(n);
    int wiped = 0;
    for (int rep = 0; rep < 200; ++rep) {
        CK(hipMemset(a, 1, big));                 // keeps the NULL stream busy
        CK(hipMemset(v, 0, n * sizeof(double)));  // the fill in question, behind it on the NULL stream
        hipLaunchKernelGGL(k_fill, dim3((n + 255) / 256), dim3(256), 0, s, v, n, 1.0);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), v, n * sizeof(double), hipMemcpyDeviceToHost));
        size_t zeros = 0;
        for (double x : h) zeros += x == 0.0;
        if (zeros) ++wiped;
    }
    std::printf("kernel on a non-blocking stream right after a NULL-stream hipMemset of the same array: the fill came last in %d of 200 rounds\n", wiped);
    return 0;
}
