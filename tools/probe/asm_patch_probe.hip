// tools/probe/asm_patch_probe.hip - round 5, review item 8: the memory shape of an assembly that does NOT run in the ILU order.
// k_assemble's tiles are nine consecutive rows of the line-coloured order; every off-diagonal entry gathers its neighbour's 13 flux
// fields from the field-major cache (2.5 GB of L2 requests per launch, tools/probe/asm_probe.hip: 250 of the mapping's 520 us).  The
// other mapping: a workgroup per compact PATCH of 4 x 4 x 4 cells, the intensive-quantity cache stored patch-major (a patch's 17 fields
// x 64 cells x 32 B are one contiguous 34 KB run), 144 of its 192 interior faces inside the patch (their neighbour records come out of
// LDS), the 96 halo cells gathered from the six neighbouring patches' runs, and the rows of J written where the ILU order puts them
// (7 blocks = 504 contiguous bytes per row, rows of one patch far apart: chains of 10 along z, columns coloured like a checkerboard,
// 64 chains per chain-tile - the line colouring of a 100^3 grid).  No arithmetic: what the time of THIS is says whether the mapping
// is worth the change of layout it needs (every other kernel that touches the cache - k_iq_update, k_newton_update, convergence,
// true-IMPES weights, hysteresis - would have to follow the patch-major order or gather).  Development probe (not product code).
//   hipcc --offload-arch=gfx950 -O3 -o asm_patch_probe tools/probe/asm_patch_probe.hip && ./asm_patch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NX = 100, NCELL = NX * NX * NX, PD = 4, PC = PD * PD * PD, NPX = NX / PD, NPATCH = NPX * NPX * NPX, IQF = 17, NFLUX = 13, F0 = 3, THREADS = 512;

// patch-local cell c = ck + 4 ci + 16 cj (z fastest); entry e = 7 c + slot, slot 3 = the diagonal, 0..2 = -y -x -z, 4..6 = +z +x +y
template <bool GATHER, bool OUT, int WAVES>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
void k_patch(const double* __restrict__ iqp, const double* __restrict__ geo, const double* __restrict__ ent, const double* __restrict__ pre,
             const int* __restrict__ perm, double* __restrict__ A, double* __restrict__ resid) {
    __shared__ __attribute__((aligned(16))) double sI[IQF * PC * 4];      // 34 816 B: own records, field-major inside the patch; later the patch's blocks
    __shared__ double sgeo[2 * PC];
    __shared__ int sperm[PC];
    const int tid = threadIdx.x, p = blockIdx.x;
    const int pk = p % NPX, pi = (p / NPX) % NPX, pj = p / (NPX * NPX);
    {   // round A: own records, one contiguous run
        const double2* g2 = reinterpret_cast<const double2*>(iqp + (size_t)p * IQF * PC * 4);
        double2* s2 = reinterpret_cast<double2*>(sI);
        for (int i = tid; i < IQF * PC * 2; i += THREADS) s2[i] = g2[i];
    }
    if (tid < 2 * PC) sgeo[tid] = geo[(size_t)p * 2 * PC + tid];
    if (tid < PC) sperm[tid] = perm[(size_t)p * PC + tid];
    const bool act = tid < 7 * PC;
    const int c = act ? tid / 7 : 0, slot = act ? tid % 7 : 3;
    const int ck = c % PD, ci = (c / PD) % PD, cj = c / (PD * PD);
    const double2 ta = act ? reinterpret_cast<const double2*>(ent)[(size_t)p * 7 * PC + tid] : make_double2(0.0, 0.0);   // transmissibility, area
    // the neighbour: inside the patch (LDS), in a neighbouring patch (gather), or outside the grid
    int nk = ck, ni = ci, nj = cj;
    if (slot == 0) nj--; else if (slot == 1) ni--; else if (slot == 2) nk--; else if (slot == 4) nk++; else if (slot == 5) ni++; else if (slot == 6) nj++;
    int qk = pk, qi = pi, qj = pj;
    if (nk < 0) { nk += PD; qk--; } else if (nk >= PD) { nk -= PD; qk++; }
    if (ni < 0) { ni += PD; qi--; } else if (ni >= PD) { ni -= PD; qi++; }
    if (nj < 0) { nj += PD; qj--; } else if (nj >= PD) { nj -= PD; qj++; }
    const bool outside = qk < 0 || qk >= NPX || qi < 0 || qi >= NPX || qj < 0 || qj >= NPX;
    const bool inPatch = qk == pk && qi == pi && qj == pj;
    const int nc = nk + PD * ni + PD * PD * nj;
    double q[NFLUX * 4];
#pragma unroll
    for (int i = 0; i < NFLUX * 4; ++i) q[i] = 0.0;
    if (act && slot != 3 && !outside && !inPatch && GATHER) {   // round B: a halo cell's flux fields from the neighbouring patch's run
        const int qp = qk + NPX * (qi + NPX * qj);
        const double2* g2 = reinterpret_cast<const double2*>(iqp + (size_t)qp * IQF * PC * 4);
#pragma unroll
        for (int f = 0; f < NFLUX; ++f) {
            const double2 a = g2[((F0 + f) * PC + nc) * 2], b = g2[((F0 + f) * PC + nc) * 2 + 1];
            q[4 * f] = a.x; q[4 * f + 1] = a.y; q[4 * f + 2] = b.x; q[4 * f + 3] = b.y;
        }
    } else if (act && slot == 3) {
#pragma unroll
        for (int e = 0; e < 10; ++e) q[4 * e] = pre[((size_t)p * PC + c) * 10 + e];
    }
    __syncthreads();
    if (act && slot != 3 && inPatch) {   // a face inside the patch: the neighbour's record out of LDS
#pragma unroll
        for (int f = 0; f < NFLUX; ++f)
#pragma unroll
            for (int u = 0; u < 4; ++u) q[4 * f + u] = sI[((F0 + f) * PC + nc) * 4 + u];
    }
    double s = ta.x + ta.y + sgeo[c] + sgeo[PC + c];
#pragma unroll
    for (int i = 0; i < NFLUX * 4; ++i) s += q[i];
    for (int f = 0; f < IQF; ++f) s += sI[(f * PC + c) * 4 + (tid & 3)];
    __syncthreads();
    double* sblk = sI;   // 7 x 64 blocks of 9 doubles = 32 256 B
    if (act)
        for (int e = 0; e < 9; ++e) sblk[tid * 9 + e] = s + e;
    __syncthreads();
    if (OUT) {
        if (act && slot == 3)
            for (int e = 0; e < 3; ++e) resid[(size_t)sperm[c] * 3 + e] = s;
        for (int i = tid; i < 63 * PC; i += THREADS) {   // a row's 504 bytes by consecutive lanes, the rows where the ILU order has them
            const int row = i / 63, off = i - row * 63;
            A[(size_t)sperm[row] * 63 + off] = sblk[i];
        }
    } else if (s == 1.2345e300) A[0] = s;
}

template <bool GATHER, bool OUT, int WAVES>
static void run(const char* name, const double* iqp, const double* geo, const double* ent, const double* pre, const int* perm, double* A, double* resid, double bytes) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_patch<GATHER, OUT, WAVES>), dim3(NPATCH), dim3(THREADS), 0, 0, iqp, geo, ent, pre, perm, A, resid);
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k_patch<GATHER, OUT, WAVES>), dim3(NPATCH), dim3(THREADS), 0, 0, iqp, geo, ent, pre, perm, A, resid);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-66s %8.1f us   %7.1f GB/s of k_assemble's algorithmic bytes\n", name, 1e3 * ms / reps, bytes / (ms / reps * 1e-3) / 1e9);
}

int main() {
    // where the ILU order puts a cell's row: chains of 10 along z, columns coloured like a checkerboard, 64 chains per chain-tile, a
    // chain-tile's rows step by step
    std::vector<int> perm((size_t)NPATCH * PC);
    const int chainsPerColour = NX * NX / 2 * (NX / 10), fullTiles = chainsPerColour / 64, lastWidth = chainsPerColour - fullTiles * 64;
    for (int p = 0; p < NPATCH; ++p) {
        const int pk = p % NPX, pi = (p / NPX) % NPX, pj = p / (NPX * NPX);
        for (int c = 0; c < PC; ++c) {
            const int k = pk * PD + c % PD, i = pi * PD + (c / PD) % PD, j = pj * PD + c / (PD * PD);
            const int colour = (i + j) & 1, chain = ((j * NX + i) / 2) * (NX / 10) + k / 10, step = k % 10;
            const int tile = chain / 64, idx = chain % 64, width = tile < fullTiles ? 64 : lastWidth;
            perm[(size_t)p * PC + c] = colour * (NCELL / 2) + tile * 640 + step * width + idx;
        }
    }
    {   // a bijection?
        std::vector<char> seen(NCELL, 0);
        for (int v : perm) { if (v < 0 || v >= NCELL || seen[v]) { printf("perm is not a bijection\n"); return 1; } seen[v] = 1; }
    }
    double *d_iq, *d_geo, *d_ent, *d_pre, *d_A, *d_res;
    int* d_perm;
    CK(hipMalloc(&d_iq, (size_t)IQF * NCELL * 32)); CK(hipMemset(d_iq, 0, (size_t)IQF * NCELL * 32));
    CK(hipMalloc(&d_geo, (size_t)NCELL * 16)); CK(hipMemset(d_geo, 0, (size_t)NCELL * 16));
    CK(hipMalloc(&d_ent, (size_t)NCELL * 7 * 16)); CK(hipMemset(d_ent, 0, (size_t)NCELL * 7 * 16));
    CK(hipMalloc(&d_pre, (size_t)NCELL * 80)); CK(hipMemset(d_pre, 0, (size_t)NCELL * 80));
    CK(hipMalloc(&d_A, (size_t)NCELL * 504)); CK(hipMalloc(&d_res, (size_t)NCELL * 24));
    CK(hipMalloc(&d_perm, perm.size() * 4)); CK(hipMemcpy(d_perm, perm.data(), perm.size() * 4, hipMemcpyHostToDevice));
    const double nnz = 6940000.0, bytes = 85.0 * NCELL + 12.0 * nnz + 72.0 * nnz + 24.0 * NCELL;   // bench.py: alg_bytes["assemble"] (692 MB)
    printf("%d patches of %d cells, %d threads per workgroup; k_assemble's algorithmic bytes %.1f MB\n", NPATCH, PC, THREADS, bytes / 1e6);
#define RUN(G, O, W, name) run<G, O, W>(name, d_iq, d_geo, d_ent, d_pre, d_perm, d_A, d_res, bytes)
    RUN(true, true, 2, "patch mapping: all rounds and stores, 2 wavefronts per SIMD");
    RUN(true, true, 4, "patch mapping: all rounds and stores, 4 wavefronts per SIMD");
    RUN(false, true, 4, "  without the halo gather");
    RUN(true, false, 4, "  without the rows of J (scattered 504-byte runs) and the residual");
    RUN(false, false, 4, "  without both");
    RUN(true, true, 6, "patch mapping: all rounds and stores, 6 wavefronts per SIMD");
    return 0;
}
