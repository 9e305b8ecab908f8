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

extern "C" {

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
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, h_id128, 128);
    ncclComm_t comm;
    CANVAS_NCCL_TRY(ctx, ncclCommInitRank(&comm, nranks, id, rank));
    ctx->comm = (void*)comm; ctx->rank = rank; ctx->nranks = nranks;
    return CANVAS_OK;
}

int32_t canvas_allgather_boundaries(canvas_ctx* ctx, const int32_t* d_local, int32_t nlocal, int32_t max_per_rank,
                                    int32_t* d_all, int32_t* h_counts) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (nlocal < 0 || max_per_rank < nlocal || !d_all) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_allgather_boundaries: bad arguments");
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int rec = 1 + max_per_rank;
    int32_t rc = canvas_ws_reserve(ctx, (size_t)rec * 4 + 256); if (rc) return rc;
    int32_t* send = (int32_t*)ctx->ws;
    hipLaunchKernelGGL(k_pack_boundaries, dim3((rec + 255) / 256), dim3(256), 0, ctx->stream, d_local, nlocal, max_per_rank, send);
    if (ctx->nranks == 1 || !ctx->comm) {
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(d_all, send, (size_t)rec * 4, hipMemcpyDeviceToDevice, ctx->stream));
    } else {
        CANVAS_NCCL_TRY(ctx, ncclAllGather(send, d_all, rec, ncclInt32, (ncclComm_t)ctx->comm, ctx->stream));
    }
    if (h_counts) {
        for (int r = 0; r < ctx->nranks; r++) CANVAS_HIP_TRY(ctx, hipMemcpyAsync(&h_counts[r], d_all + (size_t)r * rec, 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CANVAS_OK;
}

}  // extern "C"
