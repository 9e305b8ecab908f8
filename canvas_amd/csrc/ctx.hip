// Context, memory and stream plumbing of libcanvas_hip.so (include/canvas_hip.h "context" section).
#include "common.hpp"

std::atomic<long long> g_cvx_mail_awaits{0}, g_cvx_mail_waited{0};       // common.hpp: cvx_mail_await

extern "C" {

// process-wide: [0] how often a result that a kernel wrote straight into pinned host memory was looked at behind its synchronisation, [1] how often that first look found the
// previous contents (the synchronisation had returned before the stores had landed) and the library polled the result's sequence word until it arrived
int32_t canvas_stale_reads(int64_t* h_out2) {
    if (!h_out2) return CANVAS_ERR_INVALID;
    h_out2[0] = g_cvx_mail_awaits.load(); h_out2[1] = g_cvx_mail_waited.load();
    return CANVAS_OK;
}

#ifndef CANVAS_SRC_HASH
#define CANVAS_SRC_HASH "unhashed-build-0000000000000000"
#endif
// the hash of the sources this library was compiled from (canvas_amd/build.py::source_hash): build() reads it back from the file's bytes and
// recompiles on a mismatch, so a shipped binary always corresponds to the sources next to it
__attribute__((used)) static const char src_hash_marker[] = "CANVAS_SRC_HASH=" CANVAS_SRC_HASH;
const char* canvas_version(void) { return "canvas_hip 0.2 (gfx950) src=" CANVAS_SRC_HASH; }

canvas_ctx* canvas_create(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return nullptr;  // no CPU fallback
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    canvas_ctx* ctx = new canvas_ctx();
    ctx->device = device;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return nullptr; }
    ctx->own_stream = true;
    return ctx;
}

void canvas_destroy(canvas_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    cvx_comm_destroy(ctx);
    if (ctx->side) { (void)hipStreamSynchronize(ctx->side); (void)hipStreamDestroy(ctx->side); }
    if (ctx->copy) { (void)hipStreamSynchronize(ctx->copy); (void)hipStreamDestroy(ctx->copy); }
    for (auto e : ctx->up_ev) (void)hipEventDestroy(e);
    if (ctx->up_fence) (void)hipEventDestroy(ctx->up_fence);
    if (ctx->side_ev) (void)hipEventDestroy(ctx->side_ev);
    if (ctx->side_ev2) (void)hipEventDestroy(ctx->side_ev2);
    if (ctx->up2_stage) (void)hipFree(ctx->up2_stage);
    if (ctx->side_pin) (void)hipHostFree(ctx->side_pin);
    if (ctx->wv_pin) (void)hipHostFree(ctx->wv_pin);
    if (ctx->wv_main) (void)hipStreamDestroy(ctx->wv_main);
    if (ctx->wv_chain) (void)hipStreamDestroy(ctx->wv_chain);
    if (ctx->wv_sub) (void)hipStreamDestroy(ctx->wv_sub);
    if (ctx->wv_sub2) (void)hipStreamDestroy(ctx->wv_sub2);
    if (ctx->wv_copy) (void)hipStreamDestroy(ctx->wv_copy);
    if (ctx->wv_ev_in) (void)hipEventDestroy(ctx->wv_ev_in);
    if (ctx->wv_ev_x) (void)hipEventDestroy(ctx->wv_ev_x);
    if (ctx->wv_fgh) (void)hipFree(ctx->wv_fgh);
    if (ctx->covq_pin) (void)hipHostFree(ctx->covq_pin);
    if (ctx->covq_dev) (void)hipFree(ctx->covq_dev);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->gc_arena) (void)hipFree(ctx->gc_arena);
    if (ctx->cg_state) (void)hipFree(ctx->cg_state);
    if (ctx->bin_dev) (void)hipFree(ctx->bin_dev);
    if (ctx->bin_ev) (void)hipEventDestroy(ctx->bin_ev);
    if (ctx->shard_ws) (void)hipFree(ctx->shard_ws);
    if (ctx->comm_pin) (void)hipHostFree(ctx->comm_pin);
    if (ctx->misc_pin) (void)hipHostFree(ctx->misc_pin);
    if (ctx->sel_ws) (void)hipFree(ctx->sel_ws);
    if (ctx->sel_hist) (void)hipFree(ctx->sel_hist);
    if (ctx->sel_pin) (void)hipHostFree(ctx->sel_pin);
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* canvas_last_error(canvas_ctx* ctx) { return ctx ? ctx->err.c_str() : "no context (no usable GPU: this library has no CPU fallback)"; }

int32_t canvas_set_one_shot(canvas_ctx* ctx, int32_t on) {
    if (!ctx) return CANVAS_ERR_INVALID;
    ctx->one_shot = on ? 1 : 0;
    return CANVAS_OK;
}
int32_t canvas_set_stream(canvas_ctx* ctx, void* hip_stream) {
    if (!ctx) return CANVAS_ERR_INVALID;
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (hip_stream) {
        if (ctx->own_stream && ctx->stream) CANVAS_HIP_TRY(ctx, hipStreamDestroy(ctx->stream));
        ctx->stream = (hipStream_t)hip_stream; ctx->own_stream = false;
    } else if (!ctx->own_stream) {
        CANVAS_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)); ctx->own_stream = true;
    }
    return CANVAS_OK;
}

int32_t canvas_synchronize(canvas_ctx* ctx) {
    if (!ctx) return CANVAS_ERR_INVALID;
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CANVAS_OK;
}

void* canvas_device_malloc(canvas_ctx* ctx, int64_t bytes) {
    if (!ctx || bytes < 0) return nullptr;
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, (size_t)(bytes > 0 ? bytes : 1));
    if (e != hipSuccess) { ctx->err = std::string("hipMalloc: ") + hipGetErrorString(e); return nullptr; }
    return p;
}
int32_t canvas_device_free(canvas_ctx* ctx, void* d_ptr) {
    if (!ctx) return CANVAS_ERR_INVALID;
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    CANVAS_HIP_TRY(ctx, hipFree(d_ptr));
    return CANVAS_OK;
}
int32_t canvas_memcpy_h2d(canvas_ctx* ctx, void* d_dst, const void* h_src, int64_t bytes) {
    if (!ctx || bytes < 0) return CANVAS_ERR_INVALID;
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(d_dst, h_src, (size_t)bytes, hipMemcpyHostToDevice, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CANVAS_OK;
}
int32_t canvas_memcpy_d2h(canvas_ctx* ctx, void* h_dst, const void* d_src, int64_t bytes) {
    if (!ctx || bytes < 0) return CANVAS_ERR_INVALID;
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(h_dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CANVAS_OK;
}

int32_t canvas_host_register(canvas_ctx* ctx, void* h_ptr, int64_t bytes) {
    if (!ctx || !h_ptr || bytes <= 0) return CANVAS_ERR_INVALID;
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    CANVAS_HIP_TRY(ctx, hipHostRegister(h_ptr, (size_t)bytes, hipHostRegisterDefault));
    return CANVAS_OK;
}
int32_t canvas_host_unregister(canvas_ctx* ctx, void* h_ptr) {
    if (!ctx || !h_ptr) return CANVAS_ERR_INVALID;
    CANVAS_HIP_TRY(ctx, hipHostUnregister(h_ptr));
    return CANVAS_OK;
}

// The per-base arrays always come from host memory (CanvasBin.LoadIntermediateData, CanvasBin.cs:965-969): chromosome after chromosome goes out on a
// copy stream of its own, each followed by an event, and the binning call that follows sweeps a chromosome as soon as it has arrived.
int32_t canvas_upload_genome_begin(canvas_ctx* ctx, int32_t nchr, const int64_t* h_len, const uint8_t* const* h_bases, uint8_t* const* d_bases,
                                   const uint64_t* const* h_mask, uint64_t* const* d_mask, const uint8_t* const* h_hits, uint8_t* const* d_hits) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (nchr <= 0 || !h_len || !d_bases || !d_mask || !d_hits) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_upload_genome_begin: bad arguments");
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->copy) CANVAS_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy, hipStreamNonBlocking));
    while ((int)ctx->up_ev.size() < nchr) { hipEvent_t e; CANVAS_HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming)); ctx->up_ev.push_back(e); }
    // the destinations may still be read by work queued on the compute stream (the previous pass): the copies start after it
    if (!ctx->up_fence) CANVAS_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->up_fence, hipEventDisableTiming));
    CANVAS_HIP_TRY(ctx, hipEventRecord(ctx->up_fence, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipStreamWaitEvent(ctx->copy, ctx->up_fence, 0));
    ctx->up_bases.assign(nchr, nullptr); ctx->up_mask.assign(nchr, nullptr); ctx->up_hits.assign(nchr, nullptr);
    for (int c = 0; c < nchr; c++) {
        if (h_len[c] <= 0) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_upload_genome_begin: chromosome length must be positive");
        const size_t L = (size_t)h_len[c], mbytes = (size_t)((h_len[c] + 63) / 64) * 8;
        if (h_bases && h_bases[c]) CANVAS_HIP_TRY(ctx, hipMemcpyAsync(d_bases[c], h_bases[c], L, hipMemcpyHostToDevice, ctx->copy));
        if (h_mask && h_mask[c]) CANVAS_HIP_TRY(ctx, hipMemcpyAsync(d_mask[c], h_mask[c], mbytes, hipMemcpyHostToDevice, ctx->copy));
        if (h_hits && h_hits[c]) CANVAS_HIP_TRY(ctx, hipMemcpyAsync(d_hits[c], h_hits[c], L, hipMemcpyHostToDevice, ctx->copy));
        CANVAS_HIP_TRY(ctx, hipEventRecord(ctx->up_ev[c], ctx->copy));
        ctx->up_bases[c] = d_bases[c]; ctx->up_mask[c] = d_mask[c]; ctx->up_hits[c] = d_hits[c];
    }
    ctx->up_active = true;
    return CANVAS_OK;
}
// the packed planes of bin_packed.hpp: 16 B (reference) + 32 B (hits) per 64 positions, whole tiles; h_ref may be NULL (reference planes already resident)
int32_t canvas_upload_packed_begin(canvas_ctx* ctx, int32_t nchr, const int64_t* h_len, const uint64_t* const* h_ref, uint64_t* const* d_ref,
                                   const uint64_t* const* h_hit_planes, uint64_t* const* d_hit_planes) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (nchr <= 0 || !h_len || !d_ref || !d_hit_planes) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_upload_packed_begin: bad arguments");
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->copy) CANVAS_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy, hipStreamNonBlocking));
    while ((int)ctx->up_ev.size() < nchr) { hipEvent_t e; CANVAS_HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming)); ctx->up_ev.push_back(e); }
    if (!ctx->up_fence) CANVAS_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->up_fence, hipEventDisableTiming));
    CANVAS_HIP_TRY(ctx, hipEventRecord(ctx->up_fence, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipStreamWaitEvent(ctx->copy, ctx->up_fence, 0));
    ctx->up_bases.assign(nchr, nullptr); ctx->up_mask.assign(nchr, nullptr); ctx->up_hits.assign(nchr, nullptr);
    for (int c = 0; c < nchr; c++) {
        if (h_len[c] <= 0) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_upload_packed_begin: chromosome length must be positive");
        const size_t words = (size_t)((h_len[c] + 4095) / 4096) * 64;
        if (h_ref && h_ref[c]) CANVAS_HIP_TRY(ctx, hipMemcpyAsync(d_ref[c], h_ref[c], words * 16, hipMemcpyHostToDevice, ctx->copy));
        if (h_hit_planes && h_hit_planes[c]) CANVAS_HIP_TRY(ctx, hipMemcpyAsync(d_hit_planes[c], h_hit_planes[c], words * 32, hipMemcpyHostToDevice, ctx->copy));
        CANVAS_HIP_TRY(ctx, hipEventRecord(ctx->up_ev[c], ctx->copy));
        ctx->up_bases[c] = d_ref[c]; ctx->up_mask[c] = d_ref[c]; ctx->up_hits[c] = d_hit_planes[c];
    }
    ctx->up_active = true;
    return CANVAS_OK;
}
int32_t canvas_upload_genome_wait(canvas_ctx* ctx) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (ctx->copy) CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->copy));
    ctx->up_active = false;
    return CANVAS_OK;
}

int32_t canvas_profile_enable(canvas_ctx* ctx, int32_t on) {
    if (!ctx) return CANVAS_ERR_INVALID;
    ctx->prof = on < 0 ? 0 : (on > 2 ? 1 : on);
    return CANVAS_OK;
}
int32_t canvas_profile_get(canvas_ctx* ctx, const char* name, double* h_ms_total, int32_t* h_launches, int32_t reset) {
    if (!ctx || !name) return CANVAS_ERR_INVALID;
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (auto& sl : ctx->slots) {
        if (sl.name != name) continue;
        for (size_t i = 0; i + 1 < sl.ev.size(); i += 2) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, sl.ev[i], sl.ev[i + 1]) == hipSuccess) sl.ms += ms;
            (void)hipEventDestroy(sl.ev[i]); (void)hipEventDestroy(sl.ev[i + 1]);
        }
        sl.ev.clear();
        if (h_ms_total) *h_ms_total = sl.ms;
        if (h_launches) *h_launches = sl.launches;
        if (reset) { sl.ms = 0; sl.launches = 0; }
        return CANVAS_OK;
    }
    if (h_ms_total) *h_ms_total = 0;
    if (h_launches) *h_launches = 0;
    return CANVAS_OK;
}

}  // extern "C"
