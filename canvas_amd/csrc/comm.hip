// Multi-GPU plumbing: one process per GPU, chromosomes sharded across ranks (the path has no data-path exchange:
// CanvasBin, CBS and the HMM are per-chromosome tasks in the reference — CanvasBin.cs:539, CBSRunner.cs:147,
// HiddenMarkovModelsRunner.cs:51).  The ONLY collective is one RCCL all-gather (over xGMI) of the per-rank segment
// boundary records at the end of CanvasPartition, so that every rank can number segments in file order (a30, Q17).
// The buffers are KBs: latency-bound, the 7 x 153 GB/s links are irrelevant at this size.
#include "common.hpp"
#include <rccl/rccl.h>

#define CANVAS_NCCL_TRY(ctx, expr)                                                           \
    do {                                                                                     \
        ncclResult_t r_ = (expr);                                                            \
        if (r_ != ncclSuccess) { (ctx)->err = std::string(#expr) + ": " + ncclGetErrorString(r_); return CANVAS_ERR_COMM; } \
    } while (0)

__global__ void k_pack_boundaries(const int32_t* __restrict__ local, int nlocal, int maxPer, int32_t* __restrict__ send) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) send[0] = nlocal;
    if (i < maxPer) send[1 + i] = i < nlocal ? local[i] : 0;
}

// all-gather between device buffers on the context's stream.  RCCL when the ranks share a communicator (one GPU per rank, xGMI); the host-callback transport when
// they do not (ranks that share a GPU, hosts that bring their own MPI): device -> pinned host -> callback -> device, synchronous.
int32_t cvx_allgather(canvas_ctx* ctx, const void* d_send, void* d_recv, size_t bytes) {
    ProfScope ps(ctx, "allgather");                     // (hipEvents on the library's stream around every collective: bench.py --gpus N reports their sum per pass)
    if (ctx->nranks == 1 && !ctx->comm) { CANVAS_HIP_TRY(ctx, hipMemcpyAsync(d_recv, d_send, bytes, hipMemcpyDefault, ctx->stream)); return CANVAS_OK; }
    if (ctx->comm) { CANVAS_NCCL_TRY(ctx, ncclAllGather(d_send, d_recv, bytes, ncclUint8, (ncclComm_t)ctx->comm, ctx->stream)); return CANVAS_OK; }
    if (!ctx->host_allgather) CANVAS_FAIL(ctx, CANVAS_ERR_COMM, "no communicator: call canvas_comm_init or canvas_comm_init_host first");
    const size_t need = bytes * (size_t)(ctx->nranks + 1);
    if (need * 2 > ctx->comm_pin_bytes) {      // (two areas of `need` bytes each)
        if (ctx->comm_pin) { CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); CANVAS_HIP_TRY(ctx, hipHostFree(ctx->comm_pin)); ctx->comm_pin = nullptr; ctx->comm_pin_bytes = 0; }
        CANVAS_HIP_TRY(ctx, hipHostMalloc(&ctx->comm_pin, need * 4, hipHostMallocDefault)); ctx->comm_pin_bytes = need * 4;
    }
    // two staging areas, used in turn: the upload of this exchange may still be running when the next exchange fills the OTHER area, and that exchange's own
    // synchronisation (below, same stream) has passed it by the time this area comes round again — one synchronisation per exchange instead of two
    ctx->comm_pin_flip ^= 1;
    char* hs = (char*)ctx->comm_pin + (ctx->comm_pin_flip ? ctx->comm_pin_bytes / 2 : 0); char* hr = hs + bytes;
    // (hipMemcpyDefault: a rank that announces a failed device reservation sends from / receives into pinned host memory, sharded.hip)
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(hs, d_send, bytes, hipMemcpyDefault, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->host_allgather(ctx->host_allgather_user, hs, (int64_t)bytes, hr) != 0) CANVAS_FAIL(ctx, CANVAS_ERR_COMM, "host all-gather callback failed");
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(d_recv, hr, bytes * (size_t)ctx->nranks, hipMemcpyDefault, ctx->stream));
    return CANVAS_OK;
}

extern "C" {

int32_t canvas_comm_init_host(canvas_ctx* ctx, int32_t rank, int32_t nranks, canvas_host_allgather_fn fn, void* user) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (nranks < 1 || rank < 0 || rank >= nranks || (nranks > 1 && !fn)) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_comm_init_host: bad arguments");
    if (ctx->comm_parent) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_comm_init_host: inside a sub-communicator (canvas_comm_restore first)");
    if (ctx->comm) { CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); (void)ncclCommDestroy((ncclComm_t)ctx->comm); }      // the host transport replaces an RCCL communicator: it is released, not dropped
    ctx->comm = nullptr; ctx->rank = rank; ctx->nranks = nranks; ctx->host_allgather = fn; ctx->host_allgather_user = user;
    return CANVAS_OK;
}

int32_t canvas_comm_unique_id(void* h_id128) {
    if (!h_id128) return CANVAS_ERR_INVALID;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return CANVAS_ERR_COMM;
    memcpy(h_id128, &id, 128);
    return CANVAS_OK;
}

int32_t canvas_comm_init(canvas_ctx* ctx, int32_t rank, int32_t nranks, const void* h_id128) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (nranks < 1 || rank < 0 || rank >= nranks || !h_id128) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_comm_init: bad arguments");
    if (ctx->comm_parent) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_comm_init: inside a sub-communicator (canvas_comm_restore first)");
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, h_id128, 128);
    ncclComm_t comm;
    CANVAS_NCCL_TRY(ctx, ncclCommInitRank(&comm, nranks, id, rank));
    if (ctx->comm) { CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); (void)ncclCommDestroy((ncclComm_t)ctx->comm); }      // re-initialisation: the previous communicator is released
    ctx->comm = (void*)comm; ctx->rank = rank; ctx->nranks = nranks;
    return CANVAS_OK;
}

// Sub-communicators (BASELINE configs[3]: samples x chromosome groups): the ranks that pass the same color form a communicator of their own, ordered by key, and every
// sharded call that follows runs inside it — a trio on 8 GPUs is three groups of 3 + 3 + 2 ranks, each sharding its sample's chromosomes.  The parent stays alive and
// comes back with canvas_comm_restore (the bin-size exchange and the bin intersection of a pedigree span all samples).  RCCL communicators only: a host-callback
// transport brings its own group with canvas_comm_init_host (canvas_amd/parallel.py: init_host_comm(group=...)).
int32_t canvas_comm_split(canvas_ctx* ctx, int32_t color, int32_t key) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (!ctx->comm) CANVAS_FAIL(ctx, CANVAS_ERR_COMM, "canvas_comm_split: no RCCL communicator (canvas_comm_init first; the host transport takes its group from canvas_comm_init_host)");
    if (ctx->comm_parent) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_comm_split: already inside a sub-communicator (canvas_comm_restore first)");
    if (color < 0) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_comm_split: color must not be negative");
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ncclComm_t sub = nullptr;
    CANVAS_NCCL_TRY(ctx, ncclCommSplit((ncclComm_t)ctx->comm, color, key, &sub, nullptr));
    int r = 0, n = 0;
    { const ncclResult_t e1 = ncclCommUserRank(sub, &r), e2 = e1 == ncclSuccess ? ncclCommCount(sub, &n) : e1;
      if (e2 != ncclSuccess) { (void)ncclCommDestroy(sub); ctx->err = std::string("canvas_comm_split: rank / size of the sub-communicator: ") + ncclGetErrorString(e2); return CANVAS_ERR_COMM; } }
    ctx->comm_parent = ctx->comm; ctx->rank_parent = ctx->rank; ctx->nranks_parent = ctx->nranks;
    ctx->comm = (void*)sub; ctx->rank = r; ctx->nranks = n;
    return CANVAS_OK;
}
int32_t canvas_comm_restore(canvas_ctx* ctx) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (!ctx->comm_parent) return CANVAS_OK;
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    (void)ncclCommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = ctx->comm_parent; ctx->rank = ctx->rank_parent; ctx->nranks = ctx->nranks_parent; ctx->comm_parent = nullptr;
    return CANVAS_OK;
}
int32_t canvas_comm_rank(canvas_ctx* ctx, int32_t* h_rank, int32_t* h_nranks) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (h_rank) *h_rank = ctx->rank;
    if (h_nranks) *h_nranks = ctx->nranks;
    return CANVAS_OK;
}

int32_t canvas_allgather_boundaries(canvas_ctx* ctx, const int32_t* d_local, int32_t nlocal, int32_t max_per_rank,
                                    int32_t* d_all, int32_t* h_counts) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (nlocal < 0 || max_per_rank < nlocal || !d_all) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_allgather_boundaries: bad arguments");
    return cvx_allgather_boundaries_status(ctx, d_local, nlocal, max_per_rank, d_all, h_counts);
}

}  // extern "C"

// canvas_destroy: the sub-communicator (if the context dies inside one) and the communicator itself
void cvx_comm_destroy(canvas_ctx* ctx) {
    if (ctx->comm_parent) { if (ctx->comm) (void)ncclCommDestroy((ncclComm_t)ctx->comm); ctx->comm = ctx->comm_parent; ctx->comm_parent = nullptr; }
    if (ctx->comm) { (void)ncclCommDestroy((ncclComm_t)ctx->comm); ctx->comm = nullptr; }
}

// the gather itself; nlocal < 0: this rank has failed and announces its (negative) error code in the count slot, with no records (canvas_sample_pipeline_sharded)
int32_t cvx_allgather_boundaries_status(canvas_ctx* ctx, const int32_t* d_local, int32_t nlocal, int32_t max_per_rank, int32_t* d_all, int32_t* h_counts) {
    if (max_per_rank < nlocal || !d_all) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_allgather_boundaries: bad arguments");
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int rec = 1 + max_per_rank;
    int32_t rc = canvas_ws_reserve(ctx, (size_t)rec * 4 + 256); if (rc) return rc;
    int32_t* send = (int32_t*)ctx->ws;
    hipLaunchKernelGGL(k_pack_boundaries, dim3((rec + 255) / 256), dim3(256), 0, ctx->stream, d_local, nlocal, max_per_rank, send);
    rc = cvx_allgather(ctx, send, d_all, (size_t)rec * 4); if (rc) return rc;
    if (h_counts)          // the count slot of every rank's slice: one strided copy
        CANVAS_HIP_TRY(ctx, hipMemcpy2DAsync(h_counts, 4, d_all, (size_t)rec * 4, 4, (size_t)ctx->nranks, hipMemcpyDeviceToHost, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CANVAS_OK;
}


