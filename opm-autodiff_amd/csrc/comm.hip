// Domain-decomposed (restricted additive Schwarz / block-Jacobi ILU0) variant: what the reference does over MPI with
// Dune::OwnerOverlapCopyCommunication (linalg/ISTLSolverEbos.hpp:101-105): `copyOwnerToAll` halo updates
// (linalg/ParallelOverlappingILU0.hpp:897,906-911; linalg/WellOperators.hpp:200-240) and scalar all-reduces inside the
// scalar products, plus the convergence reduction `comm.sum(7) / comm.max(3)` (flow/BlackoilModelEbos.hpp:599-603).
// Here: one context per GPU, one process per GPU; halos travel with ncclSend/ncclRecv grouped per neighbour and the
// scalars with one small ncclAllReduce, all on the context's stream (RCCL over xGMI).  RCCL is dlopen'ed so that the
// library also loads where no RCCL exists (single-GPU use).  A second backend, "loopback", connects several contexts of
// ONE process (one host thread each) through device-to-device copies and a host barrier: it exists so that the whole
// decomposition logic can be tested on a single GPU (tests/test_gpu_dd.py).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <rccl/rccl.h>

#include <map>
#include <mutex>

#include "internal.hpp"

namespace opmhip {

// ---- RCCL entry points, resolved at run time -------------------------------------------------------------------
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;
static std::mutex g_rccl_mutex;

static const char* load_rccl() {
    std::lock_guard<std::mutex> lk(g_rccl_mutex);
    if (g_rccl.lib) return nullptr;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    void* h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);  // the copy the host process already loaded, if any
    for (int i = 0; !h && i < 3; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!h) return "librccl.so not found (dlopen)";
#define SYM(f, name) *(void**)(&g_rccl.f) = dlsym(h, name); if (!g_rccl.f) return "RCCL symbol missing: " name;
    SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
    SYM(AllReduce, "ncclAllReduce") SYM(AllGather, "ncclAllGather") SYM(Send, "ncclSend") SYM(Recv, "ncclRecv") SYM(GroupStart, "ncclGroupStart")
    SYM(GroupEnd, "ncclGroupEnd") SYM(GetErrorString, "ncclGetErrorString")
    SYM(CommCount, "ncclCommCount") SYM(CommUserRank, "ncclCommUserRank") SYM(CommCuDevice, "ncclCommCuDevice")
#undef SYM
    g_rccl.lib = h;
    return nullptr;
}

// ---- loopback group: contexts of one process ---------------------------------------------------------------------
struct LoopGroup {
    int nranks = 0, joined = 0;
    pthread_barrier_t barrier;
    std::vector<opmhip_ctx*> members;
    std::vector<std::vector<double>> scratch;  // [rank][n of the largest all-reduce so far, >= 16]
};
static std::map<std::string, LoopGroup*> g_groups;
static std::mutex g_groups_mutex;

// ---- kernels -----------------------------------------------------------------------------------------------------
__global__ void k_pack_f64(int n, int w, const int* __restrict__ idx, const double* __restrict__ vec, double* __restrict__ buf) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * w) return;
    buf[e] = vec[(size_t)idx[e / w] * w + e % w];
}
__global__ void k_pack_u8(int n, const int* __restrict__ idx, const unsigned char* __restrict__ vec, unsigned char* __restrict__ buf) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) buf[e] = vec[idx[e]];
}

#define NCCLCHK(c, call)                                                                                              \
    do {                                                                                                              \
        ncclResult_t r_ = (call);                                                                                     \
        if (r_ != ncclSuccess) return fail(c, OPMHIP_DEVICE_ERROR, "%s failed: %s", #call, g_rccl.GetErrorString(r_)); \
    } while (0)

// sum (op 0) / max (op 1) of n doubles over all ranks, in place, result identical on every rank (n: a handful of scalars; num_wells x 4
// for wells shared between subdomains)
static int allreduce_body(opmhip_ctx* c, double* d_buf, int n, int op);
int comm_allreduce(opmhip_ctx* c, double* d_buf, int n, int op) {
    if (c->comm.nranks <= 1) return OPMHIP_SUCCESS;
    const int span = c->comm.reduce_span_open ? -1 : prof_span_begin(c, PROF_ALLREDUCE);   // a caller's span already covers its local sums and this
    const int rc = allreduce_body(c, d_buf, n, op);
    prof_span_end(c, span);
    return rc;
}
static int allreduce_body(opmhip_ctx* c, double* d_buf, int n, int op) {
    CommDev& C = c->comm;
    if (C.kind == COMM_RCCL) {
        NCCLCHK(c, g_rccl.AllReduce(d_buf, d_buf, (size_t)n, ncclDouble, op == 0 ? ncclSum : ncclMax, (ncclComm_t)C.nccl, c->stream));
        return OPMHIP_SUCCESS;
    }
    // loopback: an error on this rank's stream must not strand the peers inside a barrier - remember it, keep the
    // barrier protocol, report it afterwards
    LoopGroup* G = (LoopGroup*)C.group;
    if (G->scratch[C.rank].size() < (size_t)n) G->scratch[C.rank].resize(n);   // this rank's own slot; the peers read it between the two barriers only
    int rc = [&]() -> int {
        OPMHIP_HIP(c, hipMemcpyAsync(G->scratch[C.rank].data(), d_buf, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        return OPMHIP_SUCCESS;
    }();
    pthread_barrier_wait(&G->barrier);
    double few[16];
    std::vector<double> many(n > 16 ? n : 0);
    double* acc = n > 16 ? many.data() : few;
    for (int i = 0; i < n; ++i) {
        double a = G->scratch[0][i];
        for (int r = 1; r < C.nranks; ++r) a = (op == 0) ? a + G->scratch[r][i] : (a > G->scratch[r][i] ? a : G->scratch[r][i]);
        acc[i] = a;  // fixed rank order: every rank forms the same bits
    }
    pthread_barrier_wait(&G->barrier);
    if (rc) return rc;
    OPMHIP_HIP(c, hipMemcpyAsync(d_buf, acc, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
    return OPMHIP_SUCCESS;
}

// d_recv[r * count .. (r + 1) * count) = rank r's d_send[0 .. count) on every rank (count doubles per rank, the same on all ranks): the
// right-hand sides, matrix values and - at set-up - the patterns of the pressure hierarchy's level that is continued across the ranks
// (cpr.hip).  On the context's stream; RCCL: one ncclAllGather; loopback: device copies between two barriers.
int comm_allgather(opmhip_ctx* c, const double* d_send, double* d_recv, size_t count) {
    CommDev& C = c->comm;
    if (C.nranks <= 1 || C.kind == COMM_NONE) {
        if (count > 0) OPMHIP_HIP(c, hipMemcpyAsync(d_recv, d_send, count * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        return OPMHIP_SUCCESS;
    }
    if (C.kind == COMM_RCCL) {
        NCCLCHK(c, g_rccl.AllGather(d_send, d_recv, count, ncclDouble, (ncclComm_t)C.nccl, c->stream));
        return OPMHIP_SUCCESS;
    }
    LoopGroup* G = (LoopGroup*)C.group;
    C.ag_send = d_send;
    int rc = (hipStreamSynchronize(c->stream) == hipSuccess) ? OPMHIP_SUCCESS : fail(c, OPMHIP_DEVICE_ERROR, "allgather: the stream failed before the exchange");
    pthread_barrier_wait(&G->barrier);   // every rank's contribution is complete and its address published
    if (!rc) rc = [&]() -> int {
        for (int r = 0; r < C.nranks && count > 0; ++r)
            OPMHIP_HIP(c, hipMemcpyAsync(d_recv + (size_t)r * count, G->members[r]->comm.ag_send, count * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        return OPMHIP_SUCCESS;
    }();
    pthread_barrier_wait(&G->barrier);   // nobody overwrites its contribution before every copy is done
    return rc;
}

// ghost entries of `vec` (w doubles per cell, internal order, ghosts at cells Nb..) <- owners' values; on stream s (default:
// the context's).  A rank may be its own neighbour (periodic coupling; also how a single GPU exercises the RCCL path).
// loopback: a rank without neighbours still meets its peers at the exchange's two barriers (they count every rank of the group)
static bool loopback_bystander(CommDev& C);
void comm_halo_bystander(opmhip_ctx* c) { (void)loopback_bystander(c->comm); }
static bool loopback_bystander(CommDev& C) {
    if (C.kind != COMM_LOOPBACK || C.nranks <= 1 || (C.halo_set && C.nneigh > 0)) return false;
    LoopGroup* G = (LoopGroup*)C.group;
    pthread_barrier_wait(&G->barrier);
    pthread_barrier_wait(&G->barrier);
    return true;
}
static int halo_f64_body(opmhip_ctx* c, double* vec, int w, hipStream_t s);
int comm_halo_f64(opmhip_ctx* c, double* vec, int w, hipStream_t s) {
    CommDev& C = c->comm;
    if (loopback_bystander(C)) return OPMHIP_SUCCESS;
    if (!C.halo_set || C.nneigh == 0 || C.kind == COMM_NONE) return OPMHIP_SUCCESS;
    if (!s) s = c->stream;
    const int span = prof_span_begin(c, PROF_HALO, s);   // pack -> exchange -> ghosts in, on the stream the exchange runs on
    const int rc = halo_f64_body(c, vec, w, s);
    prof_span_end(c, span, s);
    return rc;
}
static int halo_f64_body(opmhip_ctx* c, double* vec, int w, hipStream_t s) {
    CommDev& C = c->comm;
    const int nsend = C.send_ptr[C.nneigh];
    if (nsend > 0)
        hipLaunchKernelGGL(k_pack_f64, dim3((nsend * w + 255) / 256), dim3(256), 0, s, nsend, w, C.d_send_idx, vec, C.d_sendbuf);
    double* ghost0 = vec + (size_t)c->pat.Nb * w;
    if (C.kind == COMM_RCCL) {
        // a failing call must not leave the group open: remember the first error, always reach ncclGroupEnd
        NCCLCHK(c, g_rccl.GroupStart());
        ncclResult_t err = ncclSuccess;
        for (int q = 0; q < C.nneigh && err == ncclSuccess; ++q) {
            err = g_rccl.Send(C.d_sendbuf + (size_t)C.send_ptr[q] * w, (size_t)(C.send_ptr[q + 1] - C.send_ptr[q]) * w, ncclDouble, C.neigh[q], (ncclComm_t)C.nccl, s);
            if (err == ncclSuccess)
                err = g_rccl.Recv(ghost0 + (size_t)C.recv_ptr[q] * w, (size_t)(C.recv_ptr[q + 1] - C.recv_ptr[q]) * w, ncclDouble, C.neigh[q], (ncclComm_t)C.nccl, s);
        }
        const ncclResult_t endErr = g_rccl.GroupEnd();
        if (err == ncclSuccess) err = endErr;
        if (err != ncclSuccess) return fail(c, OPMHIP_DEVICE_ERROR, "halo exchange (ncclSend/ncclRecv) failed: %s", g_rccl.GetErrorString(err));
        return OPMHIP_SUCCESS;
    }
    // loopback: errors are remembered and reported after the SECOND barrier, so that no peer waits for ever
    LoopGroup* G = (LoopGroup*)C.group;
    int rc = (hipStreamSynchronize(s) == hipSuccess) ? OPMHIP_SUCCESS : fail(c, OPMHIP_DEVICE_ERROR, "halo exchange: packing failed");
    pthread_barrier_wait(&G->barrier);  // every rank's send buffer is packed
    if (!rc) rc = [&]() -> int {
        for (int q = 0; q < C.nneigh; ++q) {
            const opmhip_ctx* peer = G->members[C.neigh[q]];
            const CommDev& PC = peer->comm;
            int me = -1;
            for (int t = 0; t < PC.nneigh; ++t)
                if (PC.neigh[t] == C.rank) me = t;
            if (me < 0 || PC.send_ptr[me + 1] - PC.send_ptr[me] != C.recv_ptr[q + 1] - C.recv_ptr[q])
                return fail(c, OPMHIP_INVALID_ARGUMENT, "halo lists of ranks %d and %d do not match", C.rank, C.neigh[q]);
            OPMHIP_HIP(c, hipMemcpyAsync(ghost0 + (size_t)C.recv_ptr[q] * w, PC.d_sendbuf + (size_t)PC.send_ptr[me] * w,
                                         (size_t)(C.recv_ptr[q + 1] - C.recv_ptr[q]) * w * sizeof(double), hipMemcpyDeviceToDevice, s));
        }
        OPMHIP_HIP(c, hipStreamSynchronize(s));
        return OPMHIP_SUCCESS;
    }();
    pthread_barrier_wait(&G->barrier);  // nobody repacks before every copy is done
    return rc;
}

// The exchange in front of an operator application, on the halo stream: begin() orders it behind everything the main
// stream holds so far (the input vector is complete) and returns with the exchange enqueued (RCCL) or done (loopback: the
// host takes part in it) - the caller launches the interior tiles' product on the main stream BEFORE calling begin() would
// be too early (x must be complete), so it launches them right AFTER ev_x is recorded: begin() records ev_x first thing,
// and whatever the main stream gets afterwards runs beside the exchange.  end(): the main stream waits for the ghosts.
int comm_halo_begin(opmhip_ctx* c, double* vec) {
    CommDev& C = c->comm;
    OPMHIP_HIP(c, hipEventRecord(C.ev_x, c->stream));
    OPMHIP_HIP(c, hipStreamWaitEvent(C.hstream, C.ev_x, 0));
    C.halo_vec = vec;
    // Every begin() is followed by an end() before anything else touches the communicator or d_sendbuf (launch_spmv is the only caller
    // of the pair).  A failure on the way must not leave halo-stream work unordered against what the main stream does next - packs into
    // the same send buffer, all-reduces on the same communicator: the main stream is made to wait for the halo stream before the error
    // goes back, and no exchange stays pending.
    auto fail_ordered = [&](int rc) {
        C.halo_vec = nullptr;
        if (hipEventRecord(C.ev_h, C.hstream) != hipSuccess || hipStreamWaitEvent(c->stream, C.ev_h, 0) != hipSuccess) (void)hipStreamSynchronize(C.hstream);
        return rc;
    };
    if (C.kind == COMM_RCCL) {   // asynchronous: enqueue now, the interior product is launched behind this call and overlaps on the device
        const int rc = comm_halo_f64(c, vec, BS, C.hstream);
        if (rc) return fail_ordered(rc);
        if (hipEventRecord(C.ev_h, C.hstream) != hipSuccess) return fail_ordered(fail(c, OPMHIP_DEVICE_ERROR, "halo exchange: hipEventRecord failed"));
        C.halo_vec = nullptr;
    }
    return OPMHIP_SUCCESS;   // loopback: the host-driven exchange happens in end(), after the interior launch is on its way
}
int comm_halo_end(opmhip_ctx* c) {
    CommDev& C = c->comm;
    if (C.halo_vec) {   // loopback: barriers and device copies driven from this thread, on the halo stream, while the main stream works
        const int rc = comm_halo_f64(c, C.halo_vec, BS, C.hstream);
        C.halo_vec = nullptr;
        if (rc) { (void)hipStreamSynchronize(C.hstream); return rc; }   // nothing of the halo stream is left in flight behind the error
        OPMHIP_HIP(c, hipEventRecord(C.ev_h, C.hstream));
    }
    OPMHIP_HIP(c, hipStreamWaitEvent(c->stream, C.ev_h, 0));
    return OPMHIP_SUCCESS;
}

int comm_halo_u8(opmhip_ctx* c, unsigned char* vec) {
    CommDev& C = c->comm;
    if (loopback_bystander(C)) return OPMHIP_SUCCESS;
    if (!C.halo_set || C.nneigh == 0 || C.kind == COMM_NONE) return OPMHIP_SUCCESS;
    const int nsend = C.send_ptr[C.nneigh];
    if (nsend > 0) hipLaunchKernelGGL(k_pack_u8, dim3((nsend + 255) / 256), dim3(256), 0, c->stream, nsend, C.d_send_idx, vec, C.d_sendbuf_u8);
    unsigned char* ghost0 = vec + c->pat.Nb;
    if (C.kind == COMM_RCCL) {
        NCCLCHK(c, g_rccl.GroupStart());
        ncclResult_t err = ncclSuccess;
        for (int q = 0; q < C.nneigh && err == ncclSuccess; ++q) {
            err = g_rccl.Send(C.d_sendbuf_u8 + C.send_ptr[q], (size_t)(C.send_ptr[q + 1] - C.send_ptr[q]), ncclUint8, C.neigh[q], (ncclComm_t)C.nccl, c->stream);
            if (err == ncclSuccess)
                err = g_rccl.Recv(ghost0 + C.recv_ptr[q], (size_t)(C.recv_ptr[q + 1] - C.recv_ptr[q]), ncclUint8, C.neigh[q], (ncclComm_t)C.nccl, c->stream);
        }
        const ncclResult_t endErr = g_rccl.GroupEnd();
        if (err == ncclSuccess) err = endErr;
        if (err != ncclSuccess) return fail(c, OPMHIP_DEVICE_ERROR, "halo exchange (ncclSend/ncclRecv) failed: %s", g_rccl.GetErrorString(err));
        return OPMHIP_SUCCESS;
    }
    LoopGroup* G = (LoopGroup*)C.group;
    int rc = (hipStreamSynchronize(c->stream) == hipSuccess) ? OPMHIP_SUCCESS : fail(c, OPMHIP_DEVICE_ERROR, "halo exchange: packing failed");
    pthread_barrier_wait(&G->barrier);
    if (!rc) rc = [&]() -> int {
        for (int q = 0; q < C.nneigh; ++q) {
            const CommDev& PC = G->members[C.neigh[q]]->comm;
            int me = -1;
            for (int t = 0; t < PC.nneigh; ++t)
                if (PC.neigh[t] == C.rank) me = t;
            if (me < 0 || PC.send_ptr[me + 1] - PC.send_ptr[me] != C.recv_ptr[q + 1] - C.recv_ptr[q])
                return fail(c, OPMHIP_INVALID_ARGUMENT, "halo lists of ranks %d and %d do not match", C.rank, C.neigh[q]);
            OPMHIP_HIP(c, hipMemcpyAsync(ghost0 + C.recv_ptr[q], PC.d_sendbuf_u8 + PC.send_ptr[me], (size_t)(C.recv_ptr[q + 1] - C.recv_ptr[q]),
                                         hipMemcpyDeviceToDevice, c->stream));
        }
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        return OPMHIP_SUCCESS;
    }();
    pthread_barrier_wait(&G->barrier);
    return rc;
}

void comm_release(opmhip_ctx* c) {
    CommDev& C = c->comm;
    if (C.hstream) { (void)hipStreamSynchronize(C.hstream); (void)hipStreamDestroy(C.hstream); C.hstream = nullptr; }
    if (C.ev_x) { (void)hipEventDestroy(C.ev_x); C.ev_x = nullptr; }
    if (C.ev_h) { (void)hipEventDestroy(C.ev_h); C.ev_h = nullptr; }
    if (C.kind == COMM_RCCL && C.nccl && g_rccl.CommDestroy) (void)g_rccl.CommDestroy((ncclComm_t)C.nccl);
    C.nccl = nullptr;
    C.kind = COMM_NONE;
}

}  // namespace opmhip

using namespace opmhip;

extern "C" {

int opmhip_comm_unique_id(char* id128) {
    if (!id128) return OPMHIP_INVALID_ARGUMENT;
    const char* e = load_rccl();
    if (e) return OPMHIP_DEVICE_ERROR;
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return OPMHIP_DEVICE_ERROR;
    std::memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return OPMHIP_SUCCESS;
}

int opmhip_comm_init_rccl(opmhip_ctx* c, int nranks, int rank, const char* id128) {
    if (!c || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return OPMHIP_INVALID_ARGUMENT;
    const char* e = load_rccl();
    if (e) return fail(c, OPMHIP_DEVICE_ERROR, "comm_init_rccl: %s", e);
    OPMHIP_HIP(c, hipSetDevice(c->device));
    ncclUniqueId id;
    std::memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t comm;
    NCCLCHK(c, g_rccl.CommInitRank(&comm, nranks, id, rank));
    c->comm.kind = COMM_RCCL;
    c->comm.nccl = comm;
    c->comm.nranks = nranks;
    c->comm.rank = rank;
    return OPMHIP_SUCCESS;
}

int opmhip_comm_init_loopback(opmhip_ctx* c, int nranks, int rank, const char* group_name) {
    if (!c || !group_name || nranks < 1 || rank < 0 || rank >= nranks) return OPMHIP_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lk(g_groups_mutex);
    LoopGroup*& G = g_groups[group_name];
    if (!G) {
        G = new LoopGroup();
        G->nranks = nranks;
        G->members.assign(nranks, nullptr);
        G->scratch.assign(nranks, std::vector<double>(16, 0.0));
        pthread_barrier_init(&G->barrier, nullptr, (unsigned)nranks);
    }
    if (G->nranks != nranks || G->members[rank]) return fail(c, OPMHIP_INVALID_ARGUMENT, "comm_init_loopback: group '%s' mismatch", group_name);
    G->members[rank] = c;
    c->comm.kind = COMM_LOOPBACK;
    c->comm.group = G;
    c->comm.nranks = nranks;
    c->comm.rank = rank;
    return OPMHIP_SUCCESS;
}

// one RCCL all-reduce of two doubles on the context's stream, whatever the rank count: checks that librccl could be
// loaded, the communicator works and the stream ordering holds (usable with nranks == 1 on a single GPU)
int opmhip_comm_selftest(opmhip_ctx* c, double* sum_out) {
    if (!c || !sum_out) return OPMHIP_INVALID_ARGUMENT;
    if (c->comm.kind != COMM_RCCL) return fail(c, OPMHIP_NOT_READY, "comm_selftest: no RCCL communicator");
    OPMHIP_HIP(c, hipSetDevice(c->device));
    double* d = nullptr;
    int rc = dev_alloc(c, &d, (size_t)2);
    if (rc) return rc;
    const double h[2] = {1.0 + c->comm.rank, 2.0};
    OPMHIP_HIP(c, hipMemcpyAsync(d, h, sizeof h, hipMemcpyHostToDevice, c->stream));
    NCCLCHK(c, g_rccl.AllReduce(d, d, 2, ncclDouble, ncclSum, (ncclComm_t)c->comm.nccl, c->stream));
    double o[2];
    OPMHIP_HIP(c, hipMemcpyAsync(o, d, sizeof o, hipMemcpyDeviceToHost, c->stream));
    OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
    sum_out[0] = o[0];
    sum_out[1] = o[1];
    {   // ... and one ncclAllGather (the CPR pressure stage that spans the ranks gathers with it): every rank's two doubles, this rank's slice checked
        double* g = nullptr;
        if ((rc = dev_alloc(c, &g, (size_t)2 * c->comm.nranks))) return rc;
        NCCLCHK(c, g_rccl.AllGather(d, g, 2, ncclDouble, (ncclComm_t)c->comm.nccl, c->stream));
        double back[2] = {0.0, 0.0};
        OPMHIP_HIP(c, hipMemcpyAsync(back, g + (size_t)2 * c->comm.rank, sizeof back, hipMemcpyDeviceToHost, c->stream));
        OPMHIP_HIP(c, hipStreamSynchronize(c->stream));
        dev_free(c, &g);
        if (back[0] != o[0] || back[1] != o[1]) return fail(c, OPMHIP_DEVICE_ERROR, "comm_selftest: ncclAllGather returned (%g, %g) for this rank's (%g, %g)", back[0], back[1], o[0], o[1]);
    }
    dev_free(c, &d);
    return OPMHIP_SUCCESS;
}

int opmhip_comm_info(opmhip_ctx* c, int* info4) {
    if (!c || !info4) return OPMHIP_INVALID_ARGUMENT;
    info4[0] = c->comm.nranks; info4[1] = c->comm.rank; info4[2] = c->device; info4[3] = c->comm.kind;
    if (c->comm.kind == COMM_RCCL) {
        NCCLCHK(c, g_rccl.CommCount((ncclComm_t)c->comm.nccl, &info4[0]));
        NCCLCHK(c, g_rccl.CommUserRank((ncclComm_t)c->comm.nccl, &info4[1]));
        NCCLCHK(c, g_rccl.CommCuDevice((ncclComm_t)c->comm.nccl, &info4[2]));
    }
    return OPMHIP_SUCCESS;
}

int opmhip_set_cell_global_ids(opmhip_ctx* c, const long long* gids) {
    if (!c || !gids) return OPMHIP_INVALID_ARGUMENT;
    if (!c->pattern_set) return fail(c, OPMHIP_NOT_READY, "set_cell_global_ids before set_pattern");
    c->pat.gids.assign(gids, gids + c->pat.Nloc);
    return OPMHIP_SUCCESS;
}

int opmhip_set_halo(opmhip_ctx* c, long long global_cells, int nneigh, const int* neigh_rank, const int* send_ptr,
                    const int* send_cells, const int* recv_ptr) {
    if (!c) return OPMHIP_INVALID_ARGUMENT;
    if (!c->pattern_set) return fail(c, OPMHIP_NOT_READY, "set_halo before set_pattern");
    if (global_cells < c->pat.Nb || nneigh < 0 || (nneigh > 0 && (!neigh_rank || !send_ptr || !send_cells || !recv_ptr)))
        return fail(c, OPMHIP_INVALID_ARGUMENT, "set_halo: bad arguments");
    OPMHIP_HIP(c, hipSetDevice(c->device));
    CommDev& C = c->comm;
    C.global_cells = global_cells;
    C.nneigh = nneigh;
    C.neigh.assign(neigh_rank, neigh_rank + nneigh);
    C.send_ptr.assign(1, 0);
    C.recv_ptr.assign(1, 0);
    if (nneigh > 0) {
        C.send_ptr.assign(send_ptr, send_ptr + nneigh + 1);
        C.recv_ptr.assign(recv_ptr, recv_ptr + nneigh + 1);
    }
    const int nsend = C.send_ptr[nneigh];
    if (C.recv_ptr[nneigh] != c->pat.Nghost) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_halo: recv ranges cover %d ghost cells, pattern has %d", C.recv_ptr[nneigh], c->pat.Nghost);
    std::vector<int> idx(nsend);
    for (int q = 0; q < nsend; ++q) {
        if (send_cells[q] < 0 || send_cells[q] >= c->pat.Nb) return fail(c, OPMHIP_INVALID_ARGUMENT, "set_halo: send cell %d is not an owned cell", send_cells[q]);
        idx[q] = c->pat.toOrder[send_cells[q]];  // natural local id -> internal position
    }
    int rc;
    if ((rc = dev_upload(c, &C.d_send_idx, idx))) return rc;
    if ((rc = dev_alloc(c, &C.d_sendbuf, (size_t)std::max(nsend, 1) * 3))) return rc;
    if ((rc = dev_alloc(c, &C.d_sendbuf_u8, (size_t)std::max(nsend, 1)))) return rc;
    if ((rc = dev_alloc(c, &C.d_red, (size_t)16))) return rc;
    if (!C.hstream) {
        OPMHIP_HIP(c, hipStreamCreateWithFlags(&C.hstream, hipStreamNonBlocking));
        OPMHIP_HIP(c, hipEventCreateWithFlags(&C.ev_x, hipEventDisableTiming));
        OPMHIP_HIP(c, hipEventCreateWithFlags(&C.ev_h, hipEventDisableTiming));
    }
    C.halo_set = true;
    return OPMHIP_SUCCESS;
}

}  // extern "C"
