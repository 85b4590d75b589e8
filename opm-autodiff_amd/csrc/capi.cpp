// C-ABI of libopmhip.so (include/opmhip.h): argument checking, device memory, call ordering.  No exceptions
// cross this file's boundary: every entry point is wrapped and maps failures to opmhip_status codes.
#include <cstdlib>
#include <chrono>
#include <unordered_map>
#include <algorithm>
#include <cstring>
#include <new>

#include "internal.hpp"

using namespace opmhip;

namespace {
thread_local std::string g_err = "";

double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int alloc_system(opmhip_ctx* c) {
    const Pattern& P = c->pat;
    const size_t n = (size_t)P.Nloc * BS;  // vectors carry the ghost cells at their tail (SpMV / assembly gathers)
    int rc;
    // Two blocks of slack behind the value arrays: the pipelined kernels issue their prefetch loads unconditionally (a load
    // under a condition costs them their precise waits, see chain_sweep), and for an empty step or a step past the end that
    // is one 16-byte line at the step's base address - which for the last, empty step is the end of the array.
    constexpr size_t SLACK = 2 * BB;
    if ((rc = dev_alloc(c, &c->d_A, (size_t)P.nnzb * BB + SLACK))) return rc;
    if ((rc = dev_alloc(c, &c->d_L, (size_t)P.nl * BB + SLACK))) return rc;
    if ((rc = dev_alloc(c, &c->d_U, (size_t)P.nu * BB + SLACK))) return rc;
    OPMHIP_HIP(c, hipMemsetAsync(c->d_A + (size_t)P.nnzb * BB, 0, SLACK * sizeof(double), c->stream));
    OPMHIP_HIP(c, hipMemsetAsync(c->d_L + (size_t)P.nl * BB, 0, SLACK * sizeof(double), c->stream));
    OPMHIP_HIP(c, hipMemsetAsync(c->d_U + (size_t)P.nu * BB, 0, SLACK * sizeof(double), c->stream));
    if ((rc = dev_alloc(c, &c->d_invD, (size_t)P.Nb * BB))) return rc;
    c->half_product = half_product_wanted(c);
    if (c->half_product) {   // the matrix beside its U part (written by the factorisation) and the backward sweeps' row sums
        if ((rc = dev_alloc(c, &c->d_R, (size_t)P.nr * BB + SLACK))) return rc;
        OPMHIP_HIP(c, hipMemsetAsync(c->d_R, 0, ((size_t)P.nr * BB + SLACK) * sizeof(double), c->stream));
        if ((rc = dev_alloc(c, &c->d_usum, n))) return rc;
        OPMHIP_HIP(c, hipMemsetAsync(c->d_usum, 0, n * sizeof(double), c->stream));
    }
    double** vecs[] = {&c->d_b, &c->d_x, &c->d_r, &c->d_rw, &c->d_p, &c->d_v, &c->d_s, &c->d_t, &c->d_pw, &c->d_vu, &c->d_stageV};
    for (double** v : vecs) {
        if ((rc = dev_alloc(c, v, n))) return rc;
        OPMHIP_HIP(c, hipMemsetAsync(*v, 0, n * sizeof(double), c->stream));
    }
    if ((rc = dev_alloc(c, &c->d_stageA, (size_t)P.nnzb * BB))) return rc;
    if ((rc = dev_alloc(c, &c->d_scal, (size_t)SC_COUNT))) return rc;
    OPMHIP_HIP(c, hipMemsetAsync(c->d_scal, 0, SC_COUNT * sizeof(double), c->stream));
    const int vb = (int)((n + 2047) / 2048);
    c->npart = std::max(std::max(P.tiles.ntiles(), P.tiles.nsched), vb) + 1;
    if ((rc = dev_alloc(c, &c->d_part, (size_t)3 * c->npart))) return rc;   // three lists of partial sums (the third: opmhip_config.fused_reductions)
    OPMHIP_HIP(c, hipMemsetAsync(c->d_part, 0, (size_t)3 * c->npart * sizeof(double), c->stream));
    if ((rc = dev_alloc(c, &c->d_part2, (size_t)1024))) return rc;
    OPMHIP_HIP(c, hipMemsetAsync(c->d_part2, 0, 1024 * sizeof(double), c->stream));
    if (!c->h_pinned) OPMHIP_HIP(c, hipHostMalloc((void**)&c->h_pinned, SC_COUNT * sizeof(double)));
    if (!c->h_ring) {
    OPMHIP_HIP(c, hipHostMalloc((void**)&c->h_ring, opmhip_ctx::RB_SLOTS * opmhip_ctx::RB_DOUBLES * sizeof(double), hipHostMallocMapped));
    OPMHIP_HIP(c, hipHostGetDevicePointer((void**)&c->d_ring, c->h_ring, 0));
    std::memset(c->h_ring, 0, opmhip_ctx::RB_SLOTS * opmhip_ctx::RB_DOUBLES * sizeof(double));
    }
    c->d_done = c->d_scal + SC_ZERO;
    // Every fill above went to the context's stream - the one every kernel that touches these arrays runs on.  (Until round 6 they were
    // hipMemset calls: work for the device's NULL stream that returns to the host at once (2 us for a fill of 0.33 ms) and with which a
    // non-blocking stream is not ordered - tools/probe/memset_probe.hip, profiles/r06_memset_probe.txt: with the NULL stream busy, a
    // kernel launched on a non-blocking stream right behind such a fill of the same array is overwritten by it in 199 rounds of 200.
    // The one such fill inside a solve, the zeroing of a new AMG level's value array in cpr.hip's upload_ell, is what one run in nine of
    // the test suite met: eight contexts of one process busy on one GPU, a non-finite norm in the first solve of
    // tests/test_gpu_dd.py::test_dd_cpr_pressure_stage_across_the_ranks[8-...], never reproduced in isolation.)
    OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
    return OPMHIP_SUCCESS;
}

// nothing to do for this list?  (A shared list - opmhip_wells.distributed in a decomposed run - always goes through upload_wells, where the
// ranks compare its length: a rank that stepped out here with no wells would leave the others waiting in that comparison.)
static bool no_standard_wells(const opmhip_ctx* c, const opmhip_wells* w) {
    return !w || (w->num_wells <= 0 && !(w->distributed != 0 && c->comm.nranks > 1));
}

// the rank-local part of upload_wells: everything that can fail on this rank alone (validation, allocation, copies)
static int upload_wells_local(opmhip_ctx* c, const opmhip_wells* w) {
    WellsDev& W = c->wells;
    if (w->num_ms_wells > 0) {
        // multisegment wells stay with the caller (their D^-1 is a sparse LU on the host, bda/MultisegmentWellContribution.cpp:35-62):
        // what the library needs is the callback and two pinned vectors for the round trip (bda/WellContributions.cu:160-187)
        if (!w->ms_apply) return fail(c, OPMHIP_INVALID_ARGUMENT, "wells: num_ms_wells = %d but ms_apply == NULL", w->num_ms_wells);
        if (c->comm.nranks > 1) return fail(c, OPMHIP_INVALID_ARGUMENT, "wells: multisegment wells through the host callback are not supported in decomposed runs");
        const size_t bytes = (size_t)c->pat.Nb * BS * sizeof(double);
        if (!W.h_x) OPMHIP_HIP(c, hipHostMalloc((void**)&W.h_x, bytes));
        if (!W.h_y) OPMHIP_HIP(c, hipHostMalloc((void**)&W.h_y, bytes));
        W.num_ms = w->num_ms_wells;
        W.ms_apply = w->ms_apply;
        W.ms_user = w->ms_user;
    }
    if (w->num_wells <= 0) return OPMHIP_SUCCESS;
    if (!w->val_pointers || !w->Dnnzs) return fail(c, OPMHIP_INVALID_ARGUMENT, "wells: null array");
    const int nw = w->num_wells, np = w->val_pointers[nw];
    if (np < 0) return fail(c, OPMHIP_INVALID_ARGUMENT, "wells: bad val_pointers");
    // (a rank of a decomposed run may hold none of the perforations of the wells it shares with others: np == 0, no arrays to look at)
    if (np > 0 && (!w->Ccols || !w->Bcols || !w->Cnnzs || !w->Bnnzs)) return fail(c, OPMHIP_INVALID_ARGUMENT, "wells: null array");
    // cell indices arrive in natural order; the device works in the internal order
    std::vector<int> cc(np), bc(np);
    for (int p = 0; p < np; ++p) {
        if (w->Ccols[p] < 0 || w->Ccols[p] >= c->pat.Nb || w->Bcols[p] < 0 || w->Bcols[p] >= c->pat.Nb)
            return fail(c, OPMHIP_INVALID_ARGUMENT, "wells: perforation %d cell out of range", p);
        cc[p] = c->pat.toOrder[w->Ccols[p]];
        bc[p] = c->pat.toOrder[w->Bcols[p]];
    }
    int rc;
    // device arrays grow geometrically and are then reused: wells are rebuilt for every solve
    // (linalg/ISTLSolverEbos.hpp:265-272)
    // what of the list differs from what the device already holds (a Newton iteration hands the same list over three times)
    auto same_i = [](const std::vector<int>& h, const int* p, size_t n) { return h.size() == n && (n == 0 || std::memcmp(h.data(), p, n * sizeof(int)) == 0); };
    auto same_d = [](const std::vector<double>& h, const double* p, size_t n) { return h.size() == n && (n == 0 || std::memcmp(h.data(), p, n * sizeof(double)) == 0); };
    const bool sVp = same_i(W.h_vp, w->val_pointers, (size_t)nw + 1), sD = same_d(W.h_D, w->Dnnzs, (size_t)nw * 16);
    const bool sCc = same_i(W.h_cc, cc.data(), (size_t)np), sBc = same_i(W.h_bc, bc.data(), (size_t)np);
    const bool sC = same_d(W.h_C, w->Cnnzs, (size_t)np * 12), sB = same_d(W.h_B, w->Bnnzs, (size_t)np * 12);
    if (sVp && sD && sCc && sBc && sC && sB && (size_t)nw <= W.cap_wells && (size_t)np <= W.cap_perf) {
        W.nperf = np;
        return OPMHIP_SUCCESS;
    }
    // earlier kernels on the context's (non-blocking) stream may still read the well arrays the copies below replace
    OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
    if ((size_t)nw > W.cap_wells) {
        const size_t cap = std::max((size_t)nw, 2 * W.cap_wells);
        dev_free(c, &W.d_val_pointers); dev_free(c, &W.d_D); dev_free(c, &W.d_res); dev_free(c, &W.d_xw); dev_free(c, &W.d_bx);
        W.cap_wells = 0;
        if ((rc = dev_alloc(c, &W.d_val_pointers, cap + 1))) return rc;
        if ((rc = dev_alloc(c, &W.d_D, cap * 16))) return rc;
        if ((rc = dev_alloc(c, &W.d_res, cap * 4))) return rc;
        if ((rc = dev_alloc(c, &W.d_xw, cap * 4))) return rc;
        if ((rc = dev_alloc(c, &W.d_bx, cap * 4))) return rc;
        W.cap_wells = cap;
        W.h_vp.clear(); W.h_D.clear();     // new arrays: nothing of the old content is there
    }
    if ((size_t)np > W.cap_perf) {
        const size_t cap = std::max((size_t)np, 2 * W.cap_perf);
        dev_free(c, &W.d_Ccols); dev_free(c, &W.d_Bcols); dev_free(c, &W.d_C); dev_free(c, &W.d_B);
        W.cap_perf = 0;
        if ((rc = dev_alloc(c, &W.d_Ccols, cap))) return rc;
        if ((rc = dev_alloc(c, &W.d_Bcols, cap))) return rc;
        if ((rc = dev_alloc(c, &W.d_C, cap * 12))) return rc;
        if ((rc = dev_alloc(c, &W.d_B, cap * 12))) return rc;
        W.cap_perf = cap;
        W.h_cc.clear(); W.h_bc.clear(); W.h_C.clear(); W.h_B.clear();
    }
    // a failed copy leaves the record of that array empty: the next call copies it again
    auto put_i = [&](int* d, std::vector<int>& h, const int* p, size_t n) -> int {
        h.clear();
        if (n > 0) OPMHIP_HIP(c, hipMemcpy(d, p, n * sizeof(int), hipMemcpyHostToDevice));
        h.assign(p, p + n);
        return OPMHIP_SUCCESS;
    };
    auto put_d = [&](double* d, std::vector<double>& h, const double* p, size_t n) -> int {
        h.clear();
        if (n > 0) OPMHIP_HIP(c, hipMemcpy(d, p, n * sizeof(double), hipMemcpyHostToDevice));
        h.assign(p, p + n);
        return OPMHIP_SUCCESS;
    };
    if (!same_i(W.h_vp, w->val_pointers, (size_t)nw + 1) && (rc = put_i(W.d_val_pointers, W.h_vp, w->val_pointers, (size_t)nw + 1))) return rc;
    if (!same_d(W.h_D, w->Dnnzs, (size_t)nw * 16) && (rc = put_d(W.d_D, W.h_D, w->Dnnzs, (size_t)nw * 16))) return rc;
    if (!same_i(W.h_cc, cc.data(), (size_t)np) && (rc = put_i(W.d_Ccols, W.h_cc, cc.data(), (size_t)np))) return rc;
    if (!same_i(W.h_bc, bc.data(), (size_t)np) && (rc = put_i(W.d_Bcols, W.h_bc, bc.data(), (size_t)np))) return rc;
    if (!same_d(W.h_C, w->Cnnzs, (size_t)np * 12) && (rc = put_d(W.d_C, W.h_C, w->Cnnzs, (size_t)np * 12))) return rc;
    if (!same_d(W.h_B, w->Bnnzs, (size_t)np * 12) && (rc = put_d(W.d_B, W.h_B, w->Bnnzs, (size_t)np * 12))) return rc;
    W.nperf = np;
    return OPMHIP_SUCCESS;
}

int upload_wells(opmhip_ctx* c, const opmhip_wells* w) {
    WellsDev& W = c->wells;
    W.num_wells = 0;
    W.num_ms = 0;
    W.ms_apply = nullptr;
    W.ms_user = nullptr;
    W.distributed = false;
    if (!w) return OPMHIP_SUCCESS;
    const bool shared = w->distributed != 0 && c->comm.nranks > 1;
    // Everything that can fail on this rank alone runs FIRST; a shared list (every call that takes one is collective) then settles the
    // outcome in ONE max-reduction - the list's length as max(n) and max(-n), and whether some rank failed - so that the ranks leave
    // together: with INVALID_ARGUMENT if they hold different lists or one of them could not take its part, instead of one rank returning
    // alone and its peers waiting in the first all-reduce of the solve that follows.  (What the reduction cannot catch: a rank that
    // passes NULL or distributed = 0 where the others pass a shared list never enters it.)
    int local = (w->num_ms_wells < 0 || w->num_wells < 0) ? fail(c, OPMHIP_INVALID_ARGUMENT, "wells: negative well count") : OPMHIP_SUCCESS;
    if (!local && shared && !c->d_scal) local = fail(c, OPMHIP_NOT_READY, "wells: distributed wells before the pattern is set");
    if (!local) local = upload_wells_local(c, w);
    if (shared && c->d_scal) {
        double h[3] = {(double)w->num_wells, -(double)w->num_wells, local ? 1.0 : 0.0};
        double* d = c->d_scal + SC_TMP1;   // SC_TMP1, SC_TMP2, SC_NORM: scratch between solves
        static_assert(SC_TMP2 == SC_TMP1 + 1 && SC_NORM == SC_TMP1 + 2, "three consecutive scratch scalars");
        const std::string mine = c->err;   // this rank's own failure text survives the exchange
        OPMHIP_HIP(c, hipMemcpyAsync(d, h, sizeof h, hipMemcpyHostToDevice, c->stream));
        int rca = comm_allreduce(c, d, 3, 1);
        if (rca) return rca;
        OPMHIP_HIP(c, hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, c->stream));
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        if (local) { c->err = mine; W.num_ms = 0; return local; }
        if (h[2] != 0.0) { W.num_ms = 0; return fail(c, OPMHIP_INVALID_ARGUMENT, "wells: another rank could not take its part of the shared list"); }
        if (h[0] != -h[1]) {
            W.num_ms = 0;
            return fail(c, OPMHIP_INVALID_ARGUMENT, "wells: distributed = 1 but the ranks hold lists of %d to %d wells (this rank: %d) - every rank must hand over the same wells in the same order",
                        (int)-h[1], (int)h[0], w->num_wells);
        }
    } else if (local) {
        W.num_ms = 0;
        return local;
    }
    W.num_wells = w->num_wells > 0 ? w->num_wells : 0;
    W.distributed = shared;
    return OPMHIP_SUCCESS;
}

// opmhip_config.pin_host_arrays: the caller's array becomes page-locked and mapped for DMA the first time its address is seen (the
// reference's CUDA back-end copies from the same fixed addresses every solve, bda/cusparseSolverBackend.cu:314-338; Flow allocates them
// once, bda/BdaBridge.cpp:199-232).  A refused registration is remembered and the plain copy stays.
void pin_host_range(opmhip_ctx* c, const void* p, size_t bytes) {
    if (!c->cfg.pin_host_arrays || !p || bytes == 0) return;
    for (const auto& r : c->pinned)
        if (r.p == p) return;
    const hipError_t e = hipHostRegister(const_cast<void*>(p), bytes, hipHostRegisterDefault);
    if (e != hipSuccess) (void)hipGetLastError();   // (clears the sticky error: the fallback is the ordinary copy)
    c->pinned.push_back({p, e == hipSuccess ? bytes : 0});
    if (c->cfg.verbosity > 0) std::fprintf(stderr, "opmhip: host range %p (%zu bytes) %s\n", p, bytes, e == hipSuccess ? "registered for DMA" : "could not be registered: plain copies");
}

int upload_system(opmhip_ctx* c, const double* vals, const double* b) {
    const Pattern& P = c->pat;
    pin_host_range(c, vals, (size_t)P.nnzb * BB * sizeof(double));
    pin_host_range(c, b, (size_t)P.Nb * BS * sizeof(double));
    if (vals) {
        OPMHIP_HIP(c, hipMemcpyAsync(c->d_stageA, vals, (size_t)P.nnzb * BB * sizeof(double), hipMemcpyHostToDevice, c->stream));
        launch_permute_blocks(c, c->d_stageA, c->d_A);
        c->factored = false;
    }
    if (b) {
        OPMHIP_HIP(c, hipMemcpyAsync(c->d_stageV, b, (size_t)P.Nb * BS * sizeof(double), hipMemcpyHostToDevice, c->stream));
        launch_vec_to_internal(c, c->d_stageV, c->d_b);
    }
    if (vals && c->cfg.zero_diag_fix) launch_zero_diag_fix(c);
    OPMHIP_HIP(c, hipGetLastError());
    if (vals) c->system_loaded = true;
    return OPMHIP_SUCCESS;
}

int vec_in(opmhip_ctx* c, const double* h, double* d_internal) {
    OPMHIP_HIP(c, hipMemcpyAsync(c->d_stageV, h, (size_t)c->pat.Nb * BS * sizeof(double), hipMemcpyHostToDevice, c->stream));
    launch_vec_to_internal(c, c->d_stageV, d_internal);
    return OPMHIP_SUCCESS;
}
int vec_out(opmhip_ctx* c, const double* d_internal, double* h) {
    launch_vec_to_natural(c, d_internal, c->d_stageV);
    OPMHIP_HIP(c, hipMemcpyAsync(h, c->d_stageV, (size_t)c->pat.Nb * BS * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
    return OPMHIP_SUCCESS;
}

template <class F>
int guarded(opmhip_ctx* c, F&& f) {
    try {
        return f();
    } catch (const std::bad_alloc&) {
        return fail(c, OPMHIP_UNKNOWN_ERROR, "out of host memory");
    } catch (const std::exception& e) {
        return fail(c, OPMHIP_UNKNOWN_ERROR, "exception: %s", e.what());
    } catch (...) {
        return fail(c, OPMHIP_UNKNOWN_ERROR, "unknown exception");
    }
}
}  // namespace

extern "C" {

int opmhip_abi_version(void) { return OPMHIP_ABI_VERSION; }

void opmhip_default_config(opmhip_config* cfg) {
    if (!cfg) return;
    std::memset(cfg, 0, sizeof *cfg);
    cfg->abi_version = OPMHIP_ABI_VERSION;
    cfg->device_id = 0;
    cfg->verbosity = 0;
    cfg->maxit = 200;             // linalg/FlowLinearSolverParameters.hpp:150-154
    cfg->tolerance = 1e-2;        // :142-146
    cfg->ilu_relaxation = 0.9;    // :147-149
    cfg->relax_mode = OPMHIP_RELAX_POST_SCALE;
    // The ordering this library was measured with: line colouring on large structured grids, the greedy colouring
    // elsewhere.  The reference's accelerator default "graph_coloring" (bda/BdaBridge.cpp:72-73, Jones-Plassmann) stays available under
    // its name and costs a factor of three on the 10^6-cell case; the plug-in maps Flow's untouched --opencl-ilu-reorder to "auto".
    cfg->reorder = OPMHIP_REORDER_AUTO;
    cfg->cpr_amg_ilu_levels = -1; // CPR: level 0 of the pressure AMG smooths with ILU0 where the block ordering has <= 3 colours
    cfg->zero_diag_fix = 1;
    cfg->cpr_reuse_setup = 3;     // CprReuseSetup (FlowLinearSolverParameters.hpp:212-214): never recreate
}

int opmhip_create(const opmhip_config* cfg, opmhip_ctx** out) {
    if (!out) { g_err = "opmhip_create: out == NULL"; return OPMHIP_INVALID_ARGUMENT; }
    *out = nullptr;
    if (!cfg || cfg->abi_version != OPMHIP_ABI_VERSION) { g_err = "opmhip_create: config missing or ABI version mismatch"; return OPMHIP_INVALID_ARGUMENT; }
    if (cfg->maxit < 1 || !(cfg->tolerance > 0.0)) { g_err = "opmhip_create: maxit/tolerance out of range"; return OPMHIP_INVALID_ARGUMENT; }
    if (cfg->preconditioner < OPMHIP_PRECOND_ILU0 || cfg->preconditioner > OPMHIP_PRECOND_CPR_TRUEIMPES) { g_err = "opmhip_create: unknown preconditioner"; return OPMHIP_INVALID_ARGUMENT; }
    if (cfg->chain_length < 0) { g_err = "opmhip_create: chain_length < 0"; return OPMHIP_INVALID_ARGUMENT; }
    if (cfg->cpr_reuse_setup < 0 || cfg->cpr_reuse_setup > 3) { g_err = "opmhip_create: cpr_reuse_setup must be 0 .. 3"; return OPMHIP_INVALID_ARGUMENT; }
    if (cfg->cpr_async_setup < 0 || cfg->cpr_async_setup > 1) { g_err = "opmhip_create: cpr_async_setup must be 0 or 1"; return OPMHIP_INVALID_ARGUMENT; }
    if (cfg->fused_reductions < 0 || cfg->fused_reductions > 1) { g_err = "opmhip_create: fused_reductions must be 0 or 1"; return OPMHIP_INVALID_ARGUMENT; }
    // the recurred |r|^2 = r.r - 2 a v.r + a^2 v.v carries an absolute error of eps |r_0|^2: the norm it yields has a floor of sqrt(eps) |r_0| =
    // 1.5e-8 |r_0| and a stopping rule below it would never be met
    if (cfg->fused_reductions && cfg->tolerance < 1e-6) { g_err = "opmhip_create: fused_reductions needs tolerance >= 1e-6 (the recurred residual norm cannot resolve less than sqrt(eps) |r_0|)"; return OPMHIP_INVALID_ARGUMENT; }
    if (cfg->pin_host_arrays < 0 || cfg->pin_host_arrays > 1) { g_err = "opmhip_create: pin_host_arrays must be 0 or 1"; return OPMHIP_INVALID_ARGUMENT; }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_err = std::string("opmhip_create: no HIP device visible (") + hipGetErrorString(e) + "); libopmhip has no CPU fallback";
        return OPMHIP_NO_DEVICE;
    }
    if (cfg->device_id < 0 || cfg->device_id >= ndev) { g_err = "opmhip_create: device_id out of range"; return OPMHIP_INVALID_ARGUMENT; }
    opmhip_ctx* c = new (std::nothrow) opmhip_ctx();
    if (!c) { g_err = "opmhip_create: out of memory"; return OPMHIP_UNKNOWN_ERROR; }
    c->cfg = *cfg;
    c->device = cfg->device_id;
    if ((e = hipSetDevice(c->device)) != hipSuccess || (e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&c->ev0)) != hipSuccess || (e = hipEventCreate(&c->ev1)) != hipSuccess) {
        g_err = std::string("opmhip_create: ") + hipGetErrorString(e);
        delete c;
        return OPMHIP_DEVICE_ERROR;
    }
    *out = c;
    return OPMHIP_SUCCESS;
}

void opmhip_destroy(opmhip_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    cpr_shutdown(c);
    comm_release(c);
    for (const auto& r : c->pinned)
        if (r.bytes) (void)hipHostUnregister(const_cast<void*>(r.p));
    for (void* p : c->allocs) (void)hipFree(p);
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    if (c->wells.h_x) (void)hipHostFree(c->wells.h_x);
    if (c->wells.h_y) (void)hipHostFree(c->wells.h_y);
    if (c->h_ring) (void)hipHostFree(c->h_ring);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (hipEvent_t e : c->prof.ev) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* opmhip_last_error(const opmhip_ctx* c) { return c ? c->err.c_str() : g_err.c_str(); }

int opmhip_synchronize(opmhip_ctx* c) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    OPMHIP_HIP(c, hipSetDevice(c->device));
    OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
    return OPMHIP_SUCCESS;
}

int opmhip_set_pattern(opmhip_ctx* c, int Nb, int nnzb, const int* rows, const int* cols) {
    return opmhip_set_pattern_dd(c, Nb, 0, nnzb, rows, cols);
}

int opmhip_set_pattern_dd(opmhip_ctx* c, int Nb, int Nghost, int nnzb, const int* rows, const int* cols) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        if (c->pattern_set) return fail(c, OPMHIP_NOT_READY, "set_pattern: the pattern of a context is immutable (linalg/ISTLSolverEbos.hpp:216-219)");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        int rc = build_pattern(c, Nb, Nghost, nnzb, rows, cols);
        if (rc) return rc;
        for (int i = 0; i < Nb; ++i)
            if (rows[i + 1] - rows[i] > TILE_CAP_BLOCKS)
                return fail(c, OPMHIP_ANALYSIS_FAILED, "row %d has %d blocks; the ILU0 tile kernel holds at most %d", i, rows[i + 1] - rows[i], TILE_CAP_BLOCKS);
        if ((rc = alloc_system(c))) return rc;
        c->pattern_set = true;
        return OPMHIP_SUCCESS;
    });
}

int opmhip_upload_system(opmhip_ctx* c, const double* vals, const double* b) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        if (!c->pattern_set) return fail(c, OPMHIP_NOT_READY, "upload_system before set_pattern");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        int rc = upload_system(c, vals, b);
        if (rc) return rc;
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        return OPMHIP_SUCCESS;
    });
}

int opmhip_solve_system(opmhip_ctx* c, int N, int nnz, int dim, double* vals, const int* rows, const int* cols, const double* b,
                        const opmhip_wells* wells, opmhip_result* res) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        if (!res) return fail(c, OPMHIP_INVALID_ARGUMENT, "solve_system: res == NULL");
        std::memset(res, 0, sizeof *res);
        if (dim != 3) return fail(c, OPMHIP_INVALID_ARGUMENT, "solve_system: only block size 3 is supported (bda/BdaBridge.cpp:207-211)");
        if (N <= 0 || N % 3 || nnz <= 0 || nnz % 9) return fail(c, OPMHIP_INVALID_ARGUMENT, "solve_system: N/nnz not multiples of the block size");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        const double t0 = now();
        if (!c->pattern_set) {
            if (!rows || !cols) return fail(c, OPMHIP_NOT_READY, "solve_system: first call needs rows/cols");
            int rc = opmhip_set_pattern(c, N / 3, nnz / 9, rows, cols);
            if (rc) return rc == OPMHIP_INVALID_ARGUMENT ? rc : OPMHIP_ANALYSIS_FAILED;
        }
        const Pattern& P = c->pat;
        if (N != P.Nb * 3 || nnz != P.nnzb * 9) return fail(c, OPMHIP_INVALID_ARGUMENT, "solve_system: size differs from the pattern set earlier");
        if (!vals && !c->system_loaded) return fail(c, OPMHIP_NOT_READY, "solve_system: vals == NULL but no matrix is resident on the device");
        int rc;
        if ((rc = upload_system(c, vals, b))) return rc;
        static const bool zfixSeparate = [] { const char* e = tuning_env("OPMHIP_ZFIX_SEPARATE"); return e && e[0] == '1'; }();   // A/B switch
        const bool zfix = !vals && c->cfg.zero_diag_fix;   // device-assembled Jacobian: the same fix-up as the uploaded one gets
        if (zfix && zfixSeparate) launch_zero_diag_fix(c);
        if ((rc = upload_wells(c, wells))) return rc;
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        const double t1 = now();
        FactorRider rider;   // CPR: weights and level 0's values of the pressure hierarchy are formed while the rows are in LDS
        if ((rc = cpr_factor_rider(c, &rider))) return rc;
        launch_ilu_factor(c, zfix && !zfixSeparate, &rider);  // ... applied as the rows are staged
        OPMHIP_HIP(c, hipGetLastError());
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        c->factored = true;
        c->cpr.pvals_fresh = rider.mode != 0;   // only now: the factorisation that formed weights and level 0's values has completed
        if (use_cpr(c)) {   // weights, pressure matrix, AMG values (the hierarchy's structure is built at the first solve)
            if ((rc = cpr_update(c))) return rc;
            OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        }
        const double t2 = now();
        if ((rc = bicgstab(c, res))) return rc;
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        const double t3 = now();
        c->have_result = true;
        c->last_solve_iterations = res->iterations;
        res->t_copy = t1 - t0;
        res->t_factor = t2 - t1;
        res->t_solve = t3 - t2;
        res->elapsed = t3 - t0;
        res->num_colors = P.numColors;
        if (c->cfg.verbosity > 0)
            std::fprintf(stderr, "opmhip: converged %d, it %.1f, reduction %.3e, copy %.3f ms, factor %.3f ms, solve %.3f ms\n",
                         res->converged, res->it, res->reduction, 1e3 * res->t_copy, 1e3 * res->t_factor, 1e3 * res->t_solve);
        // a singular pivot shows up as a non-finite norm: report it the way the reference's create_preconditioner does
        if (!std::isfinite(res->reduction)) {
            res->converged = 0;
            if (use_cpr(c) && cpr_coarse_pivot_failed(c))
                return fail(c, OPMHIP_CREATE_PRECONDITIONER_FAILED, "CPR: the dense LU of the coarsest pressure level met a zero or non-finite pivot");
            return fail(c, OPMHIP_CREATE_PRECONDITIONER_FAILED, "non-finite residual norm (singular diagonal block in ILU0?)");
        }
        return OPMHIP_SUCCESS;
    });
}

int opmhip_wells_apply_residual(opmhip_ctx* c, const opmhip_wells* wells, const double* res_well) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        if (!c->system_loaded && !c->asmb.assembled) return fail(c, OPMHIP_NOT_READY, "wells_apply_residual: no residual on the device");
        if (no_standard_wells(c, wells)) return OPMHIP_SUCCESS;
        if (wells->num_wells > 0 && !res_well) return fail(c, OPMHIP_INVALID_ARGUMENT, "wells_apply_residual: null res_well");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        int rc;
        if ((rc = upload_wells(c, wells))) return rc;
        if (wells->num_wells <= 0) return OPMHIP_SUCCESS;   // an empty shared list: the ranks have agreed that it is empty
        OPMHIP_HIP(c, hipMemcpyAsync(c->wells.d_res, res_well, (size_t)wells->num_wells * 4 * sizeof(double), hipMemcpyHostToDevice, c->stream));
        launch_wells_residual(c, c->wells.d_res, c->d_b);
        OPMHIP_HIP(c, hipGetLastError());
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));  // res_well is the caller's again
        c->wells.num_wells = c->wells.num_ms = 0;  // the operator form is set per solve_system call
        return OPMHIP_SUCCESS;
    });
}

int opmhip_add_well_contributions(opmhip_ctx* c, const opmhip_wells* wells) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        if (!c->system_loaded && !c->asmb.assembled) return fail(c, OPMHIP_NOT_READY, "add_well_contributions: no matrix on the device");
        // A well shared by several subdomains has blocks -C_c^T D^-1 B_b with c and b on different ranks: they are in no rank's pattern
        // (the reference's matrix-add mode keeps every well inside one process, ebos/eclbasevanguard.hh:148-151 with
        // --matrix-add-well-contributions).  Refused on every rank alike - the flag and the communicator are the same everywhere - before
        // any collective is entered; shared wells go through the operator form (opmhip_solve_system's `wells`).
        if (wells && wells->distributed != 0 && c->comm.nranks > 1)
            return fail(c, OPMHIP_INVALID_ARGUMENT, "add_well_contributions: a shared list (distributed = 1) cannot be written into the matrix - the blocks that couple "
                                                    "perforations of different subdomains are in no rank's pattern; hand it to opmhip_solve_system instead");
        if (no_standard_wells(c, wells)) return OPMHIP_SUCCESS;
        OPMHIP_HIP(c, hipSetDevice(c->device));
        int rc;
        if ((rc = upload_wells(c, wells))) return rc;
        if (wells->num_wells <= 0) return OPMHIP_SUCCESS;
        const Pattern& P = c->pat;
        const int nw = wells->num_wells;
        // position of every (c, b) block in the device's block-CSR; wells that touch a common block are serialised
        std::vector<int> pair_ptr(nw + 1, 0), entry;
        std::vector<char> shares(nw, 0), dup(nw, 0);
        std::unordered_map<int, int> owner;  // entry -> first well that writes it
        for (int w = 0; w < nw; ++w) {
            const int pb = wells->val_pointers[w], np = wells->val_pointers[w + 1] - pb;
            for (int a = 0; a < np; ++a)
                for (int b = 0; b < np; ++b) {
                    const int row = P.toOrder[wells->Ccols[pb + a]], colv = P.toOrder[wells->Bcols[pb + b]];
                    const int* lo = P.col.data() + P.rowptr[row];
                    const int* hi = P.col.data() + P.rowptr[row + 1];   // (not &P.col[...]: one past the end for the last row)
                    const int* it = std::lower_bound(lo, hi, colv);
                    if (it == hi || *it != colv)
                        return fail(c, OPMHIP_INVALID_ARGUMENT, "add_well_contributions: block (%d, %d) of well %d is not in the pattern",
                                    wells->Ccols[pb + a], wells->Bcols[pb + b], w);
                    const int e = (int)(it - P.col.data());
                    entry.push_back(e);
                    auto ins = owner.emplace(e, w);
                    if (!ins.second) {
                        if (ins.first->second != w) shares[w] = 1;
                        else dup[w] = 1;  // two perforations of this well in one cell: several pairs of ONE launch hit this block
                    }
                }
            pair_ptr[w + 1] = (int)entry.size();
        }
        struct Scratch {  // freed on every path out of here
            int *pp = nullptr, *en = nullptr;
            ~Scratch() { if (pp) (void)hipFree(pp); if (en) (void)hipFree(en); }
        } S;
        OPMHIP_HIP(c, hipMalloc((void**)&S.pp, pair_ptr.size() * sizeof(int)));
        OPMHIP_HIP(c, hipMalloc((void**)&S.en, std::max<size_t>(1, entry.size()) * sizeof(int)));
        OPMHIP_HIP(c, hipMemcpyAsync(S.pp, pair_ptr.data(), pair_ptr.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
        OPMHIP_HIP(c, hipMemcpyAsync(S.en, entry.data(), entry.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
        // runs of wells without shared blocks go out as one launch; a well that shares a block with an earlier one starts
        // a new launch, so that the additions to that block happen in well order (the order of well_container_)
        // a well with two perforations in one cell goes out alone, one lane adding its pairs in perforation order
        int w0 = 0;
        for (int w = 1; w <= nw; ++w)
            if (w == nw || shares[w] || dup[w] || dup[w - 1]) {
                launch_wells_add_to_matrix(c, w0, w - w0, (w - w0 == 1 && dup[w0]) ? 1 : 0, S.pp, S.en);
                w0 = w;
            }
        OPMHIP_HIP(c, hipGetLastError());
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        c->factored = false;
        c->wells.num_wells = c->wells.num_ms = 0;  // these wells now live in the matrix: no operator form left behind for later SpMVs
        return OPMHIP_SUCCESS;
    });
}

int opmhip_get_rhs(opmhip_ctx* c, double* b) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        if (!c->system_loaded && !c->asmb.assembled) return fail(c, OPMHIP_NOT_READY, "get_rhs: no right-hand side on the device");
        if (!b) return fail(c, OPMHIP_INVALID_ARGUMENT, "get_rhs: null array");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        int rc;
        if ((rc = vec_out(c, c->d_b, b))) return rc;
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        return OPMHIP_SUCCESS;
    });
}

int opmhip_wells_recover_solution(opmhip_ctx* c, const opmhip_wells* wells, const double* res_well, double* xw) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        if (!c->have_result) return fail(c, OPMHIP_NOT_READY, "wells_recover_solution before a solve");
        if (no_standard_wells(c, wells)) return OPMHIP_SUCCESS;
        if (wells->num_wells > 0 && (!res_well || !xw)) return fail(c, OPMHIP_INVALID_ARGUMENT, "wells_recover_solution: null array");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        int rc;
        if ((rc = upload_wells(c, wells))) return rc;
        if (wells->num_wells <= 0) return OPMHIP_SUCCESS;
        OPMHIP_HIP(c, hipMemcpyAsync(c->wells.d_res, res_well, (size_t)wells->num_wells * 4 * sizeof(double), hipMemcpyHostToDevice, c->stream));
        if ((rc = launch_wells_recover(c, c->wells.d_res, c->d_x, c->wells.d_xw))) return rc;
        OPMHIP_HIP(c, hipMemcpyAsync(xw, c->wells.d_xw, (size_t)wells->num_wells * 4 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        OPMHIP_HIP(c, hipGetLastError());
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        c->wells.num_wells = c->wells.num_ms = 0;
        return OPMHIP_SUCCESS;
    });
}

int opmhip_get_result(opmhip_ctx* c, double* x) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        if (!x) return fail(c, OPMHIP_INVALID_ARGUMENT, "get_result: x == NULL");
        if (!c->have_result) return fail(c, OPMHIP_NOT_READY, "get_result before a solve");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        pin_host_range(c, x, (size_t)c->pat.Nb * BS * sizeof(double));
        return vec_out(c, c->d_x, x);
    });
}

int opmhip_spmv(opmhip_ctx* c, const double* x, double* y) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        if (!x || !y) return fail(c, OPMHIP_INVALID_ARGUMENT, "spmv: null vector");
        if (!c->system_loaded) return fail(c, OPMHIP_NOT_READY, "spmv before a matrix was uploaded");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        int rc;
        if ((rc = vec_in(c, x, c->d_pw))) return rc;
        if ((rc = launch_spmv(c, c->d_pw, c->d_v, 0, nullptr, 1.0, true))) return rc;
        OPMHIP_HIP(c, hipGetLastError());
        return vec_out(c, c->d_v, y);
    });
}

int opmhip_ilu0_factor(opmhip_ctx* c, double* lu_out) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        if (!c->system_loaded) return fail(c, OPMHIP_NOT_READY, "ilu0_factor before a matrix was uploaded");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        launch_ilu_factor(c);
        OPMHIP_HIP(c, hipGetLastError());
        c->factored = true;
        if (lu_out) {
            launch_lu_to_natural(c, c->d_stageA);
            OPMHIP_HIP(c, hipMemcpyAsync(lu_out, c->d_stageA, (size_t)c->pat.nnzb * BB * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        }
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        return OPMHIP_SUCCESS;
    });
}

int opmhip_ilu0_apply(opmhip_ctx* c, const double* d, double* v) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        if (!d || !v) return fail(c, OPMHIP_INVALID_ARGUMENT, "ilu0_apply: null vector");
        if (!c->factored) return fail(c, OPMHIP_NOT_READY, "ilu0_apply before ilu0_factor");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        int rc;
        if ((rc = vec_in(c, d, c->d_p))) return rc;
        launch_ilu_apply(c, c->d_p, c->d_pw);
        OPMHIP_HIP(c, hipGetLastError());
        return vec_out(c, c->d_pw, v);
    });
}

int opmhip_preconditioned_product(opmhip_ctx* c, const double* d, double* t, double* z) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        if (!d || !t) return fail(c, OPMHIP_INVALID_ARGUMENT, "preconditioned_product: null vector");
        if (!c->factored) return fail(c, OPMHIP_NOT_READY, "preconditioned_product before ilu0_factor");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        int rc;
        if ((rc = vec_in(c, d, c->d_p))) return rc;
        // BiCGStab's statements for one half iteration (solver.hip: enqueue_half): the sweeps leave the vector without the relaxation factor,
        // the product applies it; with the half-product form the backward sweeps' row sums go along
        double scale = 1.0;
        double* us = c->half_product ? c->d_usum : nullptr;
        launch_ilu_apply(c, c->d_p, c->d_pw, -1.0, &scale, nullptr, nullptr, us);
        if ((rc = launch_spmv(c, c->d_pw, c->d_v, 0, nullptr, scale, true, us))) return rc;
        OPMHIP_HIP(c, hipGetLastError());
        if ((rc = vec_out(c, c->d_v, t))) return rc;
        if (z) {   // M^-1 d itself, relaxation factor applied (one rounded product per entry, as its readers form it)
            if ((rc = vec_out(c, c->d_pw, z))) return rc;
            if (scale != 1.0)
                for (size_t i = 0; i < (size_t)c->pat.Nb * BS; ++i) z[i] = scale * z[i];
        }
        return OPMHIP_SUCCESS;
    });
}

int opmhip_set_cpr_weights(opmhip_ctx* c, const double* weights) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        if (!use_cpr(c)) return fail(c, OPMHIP_NOT_READY, "set_cpr_weights: the context was not created with the CPR preconditioner");
        if (!c->pattern_set) return fail(c, OPMHIP_NOT_READY, "set_cpr_weights before set_pattern");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        return cpr_set_weights(c, weights);
    });
}

int opmhip_cpr_recreate(opmhip_ctx* c) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        if (!use_cpr(c)) return fail(c, OPMHIP_NOT_READY, "cpr_recreate: the context was not created with the CPR preconditioner");
        c->cpr.recreate = true;
        return OPMHIP_SUCCESS;
    });
}

int opmhip_get_cpr_weights(opmhip_ctx* c, double* weights) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        if (!weights) return fail(c, OPMHIP_INVALID_ARGUMENT, "get_cpr_weights: null array");
        if (!use_cpr(c) || !c->cpr.d_w) return fail(c, OPMHIP_NOT_READY, "get_cpr_weights: no CPR set-up on this context yet");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        return vec_out(c, c->cpr.d_w, weights);
    });
}

int opmhip_cpr_apply(opmhip_ctx* c, const double* d, double* v) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        if (!d || !v) return fail(c, OPMHIP_INVALID_ARGUMENT, "cpr_apply: null vector");
        if (!use_cpr(c)) return fail(c, OPMHIP_NOT_READY, "cpr_apply: the context was not created with the CPR preconditioner");
        if (!c->factored) return fail(c, OPMHIP_NOT_READY, "cpr_apply before ilu0_factor (the fine smoother's factors)");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        int rc;
        if ((rc = cpr_update(c, false))) return rc;
        if ((rc = vec_in(c, d, c->d_p))) return rc;
        launch_cpr_apply(c, c->d_p, c->d_pw);
        if (c->cpr.apply_rc) { rc = c->cpr.apply_rc; c->cpr.apply_rc = 0; return rc; }
        OPMHIP_HIP(c, hipGetLastError());
        return vec_out(c, c->d_pw, v);
    });
}

int opmhip_get_ordering(opmhip_ctx* c, int* toOrder, int* fromOrder, int* rowsPerColor) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    if (!c->pattern_set) return fail(c, OPMHIP_NOT_READY, "get_ordering before set_pattern");
    const Pattern& P = c->pat;
    if (toOrder) std::memcpy(toOrder, P.toOrder.data(), P.Nb * sizeof(int));
    if (fromOrder) std::memcpy(fromOrder, P.fromOrder.data(), P.Nb * sizeof(int));
    if (rowsPerColor)
        for (int k = 0; k < P.numColors; ++k) rowsPerColor[k] = P.colorPrefix[k + 1] - P.colorPrefix[k];
    return P.numColors;
}

int opmhip_get_ordering_info(opmhip_ctx* c, int info[4]) {
    if (!c || !info) return OPMHIP_INVALID_ARGUMENT;
    if (!c->pattern_set) return fail(c, OPMHIP_NOT_READY, "get_ordering_info before set_pattern");
    const Pattern& P = c->pat;
    info[0] = P.kindInForce;
    info[1] = P.chainLen;
    info[2] = P.numColors;
    info[3] = cpr_ilu_levels_in_force(c);
    return OPMHIP_SUCCESS;
}

int opmhip_get_product_form(opmhip_ctx* c, int info[4]) {
    if (!c || !info) return OPMHIP_INVALID_ARGUMENT;
    if (!c->pattern_set) return fail(c, OPMHIP_NOT_READY, "get_product_form before set_pattern");
    const Pattern& P = c->pat;
    info[0] = (c->half_product && !use_cpr(c)) ? 1 : 0;
    info[1] = P.ualias ? 1 : 0;
    info[2] = P.nr;
    if (P.Nghost > 0 && P.rest.on) {   // a subdomain: the interior tiles stream their rows without the U part, the boundary tiles whole rows
        long long blocks = 0;
        for (int b = 0; b < P.rest.nsched; ++b) blocks += P.rest.sched[4 * b + 3] - P.rest.sched[4 * b + 2];
        for (int p = P.tiles.nschedInt; p < P.tiles.nsched; ++p) blocks += P.tiles.spmvSched[4 * p + 3] - P.tiles.spmvSched[4 * p + 2];
        info[2] = (int)blocks;
    }
    info[3] = P.rest.on ? P.rest.nsched : 0;
    return OPMHIP_SUCCESS;
}

int opmhip_profile_enable(opmhip_ctx* c, int on) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        OPMHIP_HIP(c, hipSetDevice(c->device));
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        Profiler& P = c->prof;
        P.enabled = on != 0;
        P.every = on > 1 ? on : 1;   // on = k > 1: the linear-solver scopes of every k-th solve only
        P.solve_no = 0;
        P.suspended = false;
        P.pending = -1;
        P.used = 0;
        P.ev_used = 0;
        for (int k = 0; k < PROF_COUNT; ++k) { P.total_ms[k] = 0.0; P.count[k] = 0; }
        return OPMHIP_SUCCESS;
    });
}

int opmhip_profile_get(opmhip_ctx* c, int cls, long long* launches, double* total_ms) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        if (cls < 0 || cls >= PROF_COUNT || !launches || !total_ms) return fail(c, OPMHIP_INVALID_ARGUMENT, "profile_get: bad arguments");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        Profiler& P = c->prof;
        prof_flush(c);
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        if (c->comm.hstream) OPMHIP_HIP(c, hipStreamSynchronize(c->comm.hstream));   // halo spans are stamped on the halo stream
        for (size_t i = 0; i < P.used; ++i) {
            float ms = 0.f;
            if (P.cls[i] >= 0 && P.e1[i] >= 0 && hipEventElapsedTime(&ms, P.ev[P.e0[i]], P.ev[P.e1[i]]) == hipSuccess) {
                P.total_ms[P.cls[i]] += ms;
                P.count[P.cls[i]] += 1;
            }
        }
        P.used = 0;
        P.ev_used = 0;
        *launches = P.count[cls];
        *total_ms = P.total_ms[cls];
        return OPMHIP_SUCCESS;
    });
}

int opmhip_cpr_levels(opmhip_ctx* c, int* n, int* nnz, int cap) {
    if (!c || !n || !nnz || cap < 1) return OPMHIP_INVALID_ARGUMENT;
    return cpr_level_sizes(c, n, nnz, cap);
}

int opmhip_time_kernel(opmhip_ctx* c, int which, int reps, double* ms_per_launch) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    return guarded(c, [&]() -> int {
        if (!ms_per_launch || reps < 1 || which < 0 || which > 9) return fail(c, OPMHIP_INVALID_ARGUMENT, "time_kernel: bad arguments");
        if (which >= 7 && !c->half_product) return fail(c, OPMHIP_NOT_READY, "time_kernel: the half-product form is not in force on this context");
        if (!c->system_loaded) return fail(c, OPMHIP_NOT_READY, "time_kernel before a matrix was uploaded");
        if (which != 2 && which != 4 && !c->factored) return fail(c, OPMHIP_NOT_READY, "time_kernel: factor first");
        OPMHIP_HIP(c, hipSetDevice(c->device));
        auto once = [&]() {
            switch (which) {
                case 0: (void)launch_spmv(c, c->d_pw, c->d_v, 0, nullptr); break;
                case 1: launch_ilu_apply(c, c->d_p, c->d_pw); break;
                case 2: launch_ilu_factor(c); break;
                case 3: launch_vector_kernels_once(c); break;
                case 4: launch_stream_read(c); break;
                case 5: (void)launch_spmv(c, c->d_pw, c->d_v, 1, c->d_rw); break;   // with the partial sums of y.w0
                case 6: (void)launch_spmv(c, c->d_pw, c->d_v, 2, c->d_r); break;    // ... and of y.y
                case 7: (void)launch_spmv(c, c->d_pw, c->d_v, 0, nullptr, 1.0, false, c->d_usum); break;   // the rest product
                case 8: (void)launch_spmv(c, c->d_pw, c->d_v, 2, c->d_r, 1.0, false, c->d_usum); break;    // ... with two scalar products
                case 9: launch_ilu_apply(c, c->d_p, c->d_pw, -1.0, nullptr, nullptr, nullptr, c->d_usum); break;   // M^-1 with the row sums stored
            }
        };
        once();  // warm
        OPMHIP_HIP(c, hipEventRecord(c->ev0, c->stream));
        for (int i = 0; i < reps; ++i) once();
        OPMHIP_HIP(c, hipEventRecord(c->ev1, c->stream));
        OPMHIP_HIP(c, hipEventSynchronize(c->ev1));
        OPMHIP_HIP(c, hipGetLastError());
        float ms = 0.f;
        OPMHIP_HIP(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
        *ms_per_launch = (double)ms / reps;
        if (which == 2) c->factored = true;
        return OPMHIP_SUCCESS;
    });
}

}  // extern "C"
