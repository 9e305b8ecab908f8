// Shared host-side plumbing of libcanvas_hip.so (context, error handling, device workspace).
// Everything in csrc/ is compiled for gfx950 only (hipcc --offload-arch=gfx950 -ffp-contract=off): wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <chrono>
#include <thread>
#include <memory>
#include <string>
#include <vector>
#include "../../include/canvas_hip.h"

struct canvas_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    std::string err;
    // growable device scratch (bytes); reused across calls, never shrinks
    void* ws = nullptr;
    size_t ws_bytes = 0;
    // pinned host staging for small D2H results
    void* pin = nullptr;
    size_t pin_bytes = 0;
    // side stream for work that is off the critical path (per-chromosome MAD of CanvasClean), its fork event and a pinned result buffer
    hipStream_t side = nullptr;
    hipEvent_t side_ev = nullptr, side_ev2 = nullptr;
    double* side_pin = nullptr;        // 65536 results + 65544 int64 run starts
    // persistent scratch of the radix select (select.hpp): device blob and pinned staging blob
    void* sel_ws = nullptr; size_t sel_ws_bytes = 0;
    void* sel_hist = nullptr; size_t sel_hist_bytes = 0;   // replicated histograms: all zero between calls (k_select_pick clears what it reads)
    void* sel_pin = nullptr; size_t sel_pin_bytes = 0;
    // small pinned staging area for host->device parameter tables (async copies from pinned memory need no synchronisation
    // to protect the source); bump-allocated, wrapped with a synchronisation when full
    char* misc_pin = nullptr; size_t misc_off = 0;
    // pending genome upload (canvas_upload_genome_begin): copy stream, one "arrived" event per chromosome and the destination tables the next
    // binning call is matched against
    hipStream_t copy = nullptr;
    std::vector<hipEvent_t> up_ev;
    hipEvent_t up_fence = nullptr;     // compute stream -> copy stream: the destinations may still be read by the previous pass
    void* up2_stage = nullptr; size_t up2_stage_bytes = 0;      // canvas_upload_packed2_begin: the two-bit wire form of the hit planes before its expansion
    std::vector<const void*> up_bases, up_mask, up_hits;
    bool up_active = false;
    void* bin_dev = nullptr; hipEvent_t bin_ev = nullptr;      // bin_tail.hpp: arrival tickets + the sample's decisions (BinDev) in device memory; event behind their D2H copy
    long long gcw_total = 0; void* gcw_stats_dev = nullptr;     // last GCContentWeighted binning: {bins whose weighted count was decided from the exact sum's interval, bins replayed in the reference's order} (in gc_arena)
    void* gc_arena = nullptr; size_t gc_arena_bytes = 0;   // GCContentWeighted binning: read-GC profile of every position + GC prefix array (grow-only)
    size_t clean_ws_end = 0;          // clean_fast.hpp: bytes of ctx->ws the last clean_batch_enqueue carved (what is enqueued behind it must not alias them: a second phase may follow)
    bool clean_cq_failed = false, clean_cq_skip = false;   // clean_fast.hpp: a sample's counting selects gave up (it is redone with the radix selects)
    int one_shot = 0;                   // canvas_set_one_shot: the host makes one call per method and exits — staging through pinned host memory is not worth its pinning
    std::shared_ptr<void> cbs_cache;     // cbs.hip: device / pinned buffers of the arc-search and permutation engines, kept between calls (a call used to spend tens of ms in hipMalloc / hipHostMalloc)
    std::shared_ptr<void> hmm_pool;      // hmm.hip: helper threads that fill the negative-binomial emission tables of a sample
    long long cg_deferred = 0;           // clean_gc_only.hpp: chunks k_cg_apply has put aside so far (moved by k_cg_fixup): 0 on a device the grid has to itself
    void* cg_state = nullptr; unsigned cg_epoch = 0;     // clean_gc_only.hpp: tickets / genome row / chunk flags of the -g-only stage (zero between calls), call counter
    std::shared_ptr<void> clean_batch;   // clean_fast.hpp: the batch that clean_batch_enqueue queued (consumed by clean_batch_finish)
    void* comm = nullptr;  // ncclComm_t
    int32_t (*gcw_reduce)(void* user, unsigned long long* v, int n) = nullptr; void* gcw_reduce_user = nullptr;      // element-wise sum over the ranks of a sharded GCContentWeighted binning (fragment means, read-GC profile)
    void* comm_parent = nullptr; int rank_parent = 0, nranks_parent = 1;      // canvas_comm_split: the communicator the sub-communicator was split from
    int rank = 0, nranks = 1;
    // host-callback transport of the collectives (canvas_comm_init_host): used when the ranks cannot form an RCCL communicator
    int32_t (*host_allgather)(void* user, const void* send, int64_t bytes_per_rank, void* recv) = nullptr;
    void* host_allgather_user = nullptr;
    void* comm_pin = nullptr; size_t comm_pin_bytes = 0; int comm_pin_flip = 0;
    // persistent buffers of the chromosome-sharded pipeline (sharded.hip)
    void* shard_ws = nullptr; size_t shard_ws_bytes = 0;
    long long shard_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long cbs_dev[6] = {0, 0, 0, 0, 0, 0};
    long long cbs_cache_stats[6] = {0, 0, 0, 0, 0, 0};      // last CBS call: draws read out of the stream cache, generated inside batches, generated by the cache's producer, states fetched for host code; bytes mapped, words held
    long long cbs_tpermp[2] = {0, 0};          // last canvas_cbs call: edge tests (TPermP) run by the device kernel, swaps of all edge tests
    long long cbs_tailp[2] = {0, 0};   // last CBS call: TailP decisions taken from the device series / recomputed by the host series   // counters of the device permutation engine (canvas_cbs_device_stats)
    long long wv_levels = 0, wv_redone = 0;   // last canvas_wavelets call: tree levels processed, nodes recomputed by the exact chain
    void* covq_dev = nullptr; void* covq_pin = nullptr;   // pipeline.hip: counters / result of the genome-wide coverage quartiles counted while the coverage is quantised (hmm.hip)
    hipStream_t wv_main = nullptr, wv_chain = nullptr, wv_sub = nullptr, wv_sub2 = nullptr;
    hipStream_t wv_copy = nullptr; hipEvent_t wv_ev_in = nullptr, wv_ev_x = nullptr;      // canvas_wavelets: the host copy of the coverage travels next to the first kernels of the call
    void* wv_fgh = nullptr; int wv_fgh_len = 0;          // canvas_wavelets: the step coefficients of every (node length, position) of the short nodes, computed once  // canvas_wavelets: streams confined to disjoint sets of compute units (the exact chains keep theirs to themselves)
    int wv_streams_tried = 0; unsigned wv_calls = 1;
    void* wv_pin = nullptr; size_t wv_pin_bytes = 0;   // pinned arena of canvas_wavelets (host copy of the coverage + staging lists), kept between calls
    long long wv_stats[4] = {0, 0, 0, 0};     // ... long nodes decided from the closed form / sent to the chain undecided / chained for their coefficient; closed form in use
    unsigned mail_seq = 0;      // sequence numbers of the results kernels write straight into pinned host memory (cvx_mail_*, below)
    unsigned covq_seq = 0;      // ... the one the pending quartile result (covq_pin) will carry
    double hmm_dispersion = 1e9;      // IQR / median of the coverage PerSampleHMM was last set up for (hmm.hip: the first speculative attempt's lead-in)
    int hmm_retry = 0;     // chromosomes that needed the second speculative attempt (longer lead-ins) in the last HMM call
    int hmm_redo = 0;      // chromosomes recomputed sequentially by the last canvas_hmm_per_sample (speculation failures)
    // profiling: hipEvent pairs around named kernels
    int prof = 0;          // 0 off; 1 every named scope; 2 only the scopes of the dominant (roofline) kernel of a stage: a scope costs two barrier packets, ~10 us of the pass
    struct ProfSlot { std::string name; std::vector<hipEvent_t> ev; double ms = 0; int launches = 0; };
    std::vector<ProfSlot> slots;
};

// scoped event pair: records start now and stop at destruction (on ctx->stream, or on the stream given) when profiling is enabled
struct ProfScope {
    canvas_ctx* ctx; int slot = -1; hipEvent_t a = nullptr, b = nullptr; hipStream_t st = nullptr;
    ProfScope(canvas_ctx* c, const char* name, bool dominant = false, hipStream_t on = nullptr) : ctx(c), st(on ? on : c->stream) {
        if (!c->prof || (c->prof == 2 && !dominant)) return;
        for (size_t i = 0; i < c->slots.size(); i++) if (c->slots[i].name == name) slot = (int)i;
        if (slot < 0) { c->slots.push_back({}); slot = (int)c->slots.size() - 1; c->slots[slot].name = name; }
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { slot = -1; return; }
        (void)hipEventRecord(a, st);
    }
    ~ProfScope() {
        if (slot < 0) return;
        (void)hipEventRecord(b, st);
        ctx->slots[slot].ev.push_back(a); ctx->slots[slot].ev.push_back(b); ctx->slots[slot].launches++;
    }
};

#define CANVAS_HIP_TRY(ctx, expr)                                                                   \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                         \
            return CANVAS_ERR_HIP;                                                                  \
        }                                                                                           \
    } while (0)

#define CANVAS_FAIL(ctx, code, msg) do { (ctx)->err = (msg); return (code); } while (0)

// Every CANVAS_* environment switch of the library is a test or diagnostic hook (A/B forms of a kernel, forced fallbacks, timing printouts).  None of them is
// read unless CANVAS_TEST_HOOKS=1 is set as well (tests/conftest.py and the tools/ scripts set it): a production process cannot be steered off the default path by a stray variable.
static inline const char* cvx_hook(const char* name) { static const bool on = getenv("CANVAS_TEST_HOOKS") != nullptr; return on ? getenv(name) : nullptr; }

// ---- results a kernel writes STRAIGHT into pinned host memory ("mailboxes": bin size and totals, quartiles, segment count, Wavelets reports).  That a stream or an event has
// completed does not by itself put such stores in front of the host's reads on this platform: in round 4 a Wavelets report was read before it had arrived about once in eight
// process starts (gpurun_out/wv_fail_*.log) — silently wrong breakpoints.  So every mailbox carries a sequence word: the kernel stores the payload, fences at system scope, and
// stores the call's sequence number LAST with release semantics; the host resets the word before it enqueues the kernel, and after the synchronisation compares it with the number
// it handed out, polling until it matches (cvx_mail_await).  canvas_stale_reads reports how often a mailbox was looked at and how often the look came too early.
// Results that travel by hipMemcpyAsync(DeviceToHost) + synchronisation are ordinary DMA copies and need none of this.
extern std::atomic<long long> g_cvx_mail_awaits, g_cvx_mail_waited;       // ctx.hip
#if defined(__HIPCC__)
// ONE thread calls this, after every payload store of its workgroup is complete: the storing threads have executed __threadfence_system() and a barrier lies in between
__device__ __forceinline__ void cvx_mail_publish(unsigned* seqWord, unsigned seq) {
    __threadfence_system();
    __hip_atomic_store(seqWord, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
#endif
// a fresh number for one mailbox use; the word itself is reset here so that nothing an earlier use left there can pass for this one (call BEFORE the kernel is enqueued)
static inline unsigned cvx_mail_arm(canvas_ctx* ctx, volatile unsigned* seqWord) {
    unsigned s = ++ctx->mail_seq; if (s == 0) s = ++ctx->mail_seq;
    *seqWord = 0u;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    return s;
}
// after the stream / event the kernel was enqueued on has been waited for: returns when the mailbox carries `expect`
static inline int32_t cvx_mail_await(canvas_ctx* ctx, const volatile unsigned* seqWord, unsigned expect, const char* what) {
    g_cvx_mail_awaits.fetch_add(1, std::memory_order_relaxed);
    if (*seqWord != expect) {
        g_cvx_mail_waited.fetch_add(1, std::memory_order_relaxed);
        if (cvx_hook("CANVAS_MAIL_TRACE")) fprintf(stderr, "canvas: %s: the synchronisation returned before the result had arrived in host memory (sequence %u, expected %u): polling\n", what, *seqWord, expect);
        const auto t0 = std::chrono::steady_clock::now();
        while (*seqWord != expect) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 20.0) { ctx->err = std::string(what) + ": a result written to pinned host memory did not arrive"; return CANVAS_ERR_HIP; }
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#else
            std::this_thread::yield();
#endif
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return CANVAS_OK;
}

// ensure ctx->ws has at least `bytes`
static inline int32_t canvas_ws_reserve(canvas_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->ws_bytes) return CANVAS_OK;
    if (ctx->ws) { CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); CANVAS_HIP_TRY(ctx, hipFree(ctx->ws)); ctx->ws = nullptr; ctx->ws_bytes = 0; }
    size_t want = bytes + bytes / 4 + (1u << 20);
    CANVAS_HIP_TRY(ctx, hipMalloc(&ctx->ws, want));
    ctx->ws_bytes = want;
    return CANVAS_OK;
}
#define CANVAS_MISC_PIN_BYTES (256u << 10)
// copy `bytes` from host memory to the device through the pinned staging area: fully asynchronous, `src` may be a stack buffer
static inline int32_t canvas_h2d_small(canvas_ctx* ctx, void* d_dst, const void* src, size_t bytes) {
    if (bytes > CANVAS_MISC_PIN_BYTES) { CANVAS_HIP_TRY(ctx, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream)); CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); return CANVAS_OK; }
    if (!ctx->misc_pin) CANVAS_HIP_TRY(ctx, hipHostMalloc((void**)&ctx->misc_pin, CANVAS_MISC_PIN_BYTES, hipHostMallocDefault));
    const size_t need = (bytes + 63) & ~size_t(63);
    if (ctx->misc_off + need > CANVAS_MISC_PIN_BYTES) { CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); ctx->misc_off = 0; }
    char* p = ctx->misc_pin + ctx->misc_off; ctx->misc_off += need;
    memcpy(p, src, bytes);
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(d_dst, p, bytes, hipMemcpyHostToDevice, ctx->stream));
    return CANVAS_OK;
}
static inline int32_t canvas_side_init(canvas_ctx* ctx) {
    if (ctx->side) return CANVAS_OK;
    CANVAS_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
    CANVAS_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->side_ev, hipEventDisableTiming));
    CANVAS_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->side_ev2, hipEventDisableTiming));
    CANVAS_HIP_TRY(ctx, hipHostMalloc((void**)&ctx->side_pin, (65536 + 65544) * sizeof(double), hipHostMallocDefault));
    return CANVAS_OK;
}
static inline int32_t canvas_pin_reserve(canvas_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->pin_bytes) return CANVAS_OK;
    if (ctx->pin) { CANVAS_HIP_TRY(ctx, hipHostFree(ctx->pin)); ctx->pin = nullptr; ctx->pin_bytes = 0; }
    size_t want = bytes + 4096;
    CANVAS_HIP_TRY(ctx, hipHostMalloc(&ctx->pin, want, hipHostMallocDefault));
    ctx->pin_bytes = want;
    return CANVAS_OK;
}

// ---- internal cross-file entry points (hidden: not part of the C ABI)
#define CVX_INTERNAL __attribute__((visibility("hidden")))
// all-gather of bytes_per_rank bytes between device buffers on ctx->stream: RCCL (ncclAllGather over xGMI), the host-callback transport, or a copy when nranks == 1
CVX_INTERNAL int32_t cvx_allgather(canvas_ctx* ctx, const void* d_send, void* d_recv, size_t bytes_per_rank);
// canvas_allgather_boundaries with a status: nlocal < 0 announces a failed rank (its error code travels in the count slot, no records)
CVX_INTERNAL int32_t cvx_allgather_boundaries_status(canvas_ctx* ctx, const int32_t* d_local, int32_t nlocal, int32_t max_per_rank, int32_t* d_all, int32_t* h_counts);
CVX_INTERNAL void cvx_comm_destroy(canvas_ctx* ctx);      // comm.hip: releases the (sub-)communicator of a context that is being destroyed
// canvas_quantize_f2 fused with the counting of the genome-wide quartiles PerSampleHMM starts from (the same sweep); the result travels to *h_covq_out (pinned, valid after the
// next synchronisation of ctx->stream) and is handed to cvx_hmm_per_sample_preq, which then needs neither the counting sweep nor a round trip of its own
CVX_INTERNAL int32_t cvx_quantize_f2_covq(canvas_ctx* ctx, const float* d_count, int64_t n, double* d_cov, const void** h_covq_out);
CVX_INTERNAL int32_t cvx_quant_covq_enqueue(canvas_ctx* ctx, const float* d_count, int64_t n, const unsigned long long* d_n, double* d_cov, const void** h_covq_out);
// pipeline.hip: CanvasClean, the F2 hand-off (with the quartile counters) and the chromosome offsets enqueued back to back with the bin count read on the device: ONE
// synchronisation for the three stages; samples CanvasClean does not finish in its first device-driven phase take the stage-by-stage calls
CVX_INTERNAL int32_t cvx_clean_f2_offsets(canvas_ctx* ctx, int64_t n, int32_t* d_chr, int32_t* d_start, int32_t* d_stop, float* d_count, int32_t* d_gc, int32_t nchr,
                                          const uint8_t* h_chr_is_autosome, const uint8_t* h_chr_is_y, uint32_t flags, int32_t min_bins_per_gc, double* d_cov,
                                          double* h_local_sd_out, int64_t* h_n_out, int32_t* h_info, int64_t* h_chr_offset, const void** h_covq_out);
CVX_INTERNAL int32_t cvx_hmm_per_sample_preq(canvas_ctx* ctx, int32_t nchr, const double* d_cov, const int64_t* h_chr_offset, int32_t* d_state, const void* h_covq);
CVX_INTERNAL int32_t cvx_hmm_per_sample_segments(canvas_ctx* ctx, int32_t nchr, const double* d_cov, const int64_t* h_chr_offset, int32_t* d_state, const void* h_covq,
                                                 const int32_t* d_start, const int32_t* d_stop, int32_t max_inter_bin_dist, int32_t* d_segment_id, int64_t* h_nsegments);
// canvas_bin_sample on the chromosomes this rank owns, with the bin size decided by `hook` from the per-chromosome (#hit > 0, popcount(mask), possible positions
// in front of the first non-'n' base): the hook is where the sharded pipeline exchanges the rate pairs so that every rank derives the same size
typedef int32_t (*cvx_bin_size_hook)(void* user, int nchr, const long long* obs, const long long* pop, const long long* popBefore, int32_t* binSizeOut);
CVX_INTERNAL int32_t cvx_bin_sample_hooked(canvas_ctx* ctx, int32_t nchr, const uint8_t* const* d_bases, const uint64_t* const* d_mask, const uint8_t* const* d_hits, const int64_t* h_len,
                                           int32_t mode, cvx_bin_size_hook hook, void* user, int32_t* d_chr, int32_t* d_start, int32_t* d_stop, int32_t* d_gc, float* d_count, int64_t cap,
                                           int64_t* h_nbins_per_chr, int64_t* h_nbins_total, const int64_t* h_pos0_packed = nullptr, const int16_t* const* d_fraglen = nullptr);
// canvas_hmm_per_sample on a subset of chromosomes (d_cov / h_chr_offset: the subset, contiguous) with the genome-wide quartiles taken from d_cov_all[0, n_all)
// CanvasPartition -m CBS / -m Wavelets for the chromosomes h_mask selects (NULL: all); genome-wide inputs (seeds in file order, trimmed SD, coverage variability) always use the whole coverage
CVX_INTERNAL int32_t cvx_cbs_masked(canvas_ctx* ctx, int32_t nchr, const double* d_cov, const int64_t* h_chr_offset, double alpha, uint32_t nperm, int32_t undo, double undo_sd,
                                    const uint8_t* h_mask, std::vector<std::vector<int>>& segs, int64_t* h_stats);
CVX_INTERNAL int32_t cvx_wavelets_masked(canvas_ctx* ctx, int32_t nchr, const double* d_cov, const int64_t* h_chr_offset, int32_t is_germline, double threshold_lower, double threshold_upper,
                                         double mad_factor, int32_t variability_window, int32_t min_size, const uint8_t* h_mask, int32_t* h_breakpoints, int64_t cap, int64_t* h_bp_offset);
CVX_INTERNAL int32_t cvx_hmm_per_sample_subset(canvas_ctx* ctx, int32_t nchr, const double* d_cov, const int64_t* h_chr_offset, const double* d_cov_all, int64_t n_all, int32_t* d_state,
                                               const void* h_covq = nullptr);      // h_covq: the quartile counters of cvx_quantize_f2_covq over d_cov_all (saves the counting sweep and its round trip)

// (Polling the stream with hipStreamQuery instead of blocking in hipStreamSynchronize was tried for the short waits of a pass: no gain on the pass time, and
// hipStreamQuery reported completion early on streams that wait for another stream's event — results arrived before their kernels.  Every wait blocks.)
// entry points that read per-base arrays without the per-chromosome overlap wait for a pending canvas_upload_genome_begin as a whole
static inline int32_t canvas_upload_fence(canvas_ctx* ctx) {
    if (ctx->up_active) { CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->copy)); ctx->up_active = false; }
    return CANVAS_OK;
}

// bump allocator over the workspace
struct WsCarver {
    char* base; size_t off = 0;
    explicit WsCarver(void* p) : base((char*)p) {}
    template <class T> T* take(size_t n) { off = (off + 255) & ~size_t(255); T* r = (T*)(base + off); off += n * sizeof(T); return r; }
};
struct WsSizer {
    size_t off = 0;
    template <class T> void take(size_t n) { off = (off + 255) & ~size_t(255); off += n * sizeof(T); }
};

// ---- device helpers ------------------------------------------------------------------------------------------
#define WAVE 64
// A pointer that a kernel reads out of a table in device memory (BinChrom, CfArgs, PermReq ...) is a generic pointer to the compiler: every access through it becomes a
// flat_load / flat_store.  A flat instruction counts on vmcnt AND lgkmcnt, and because LDS and memory answer out of order the only wait the compiler can put behind one is
// s_waitcnt vmcnt(0) lgkmcnt(0): every LDS read of the kernel then also waits for every load still in flight (a prefetch for the next round is waited for at the first LDS
// access of this one) and for every store.  Kernel ARGUMENTS are inferred to be global; table entries are not, so the kernels convert them once: gptr<T> is T* in the global
// address space, and the accesses come out as global_load / global_store with their own counter.  (A cast to the global address space and back is folded away, and
// __builtin_assume(!__builtin_amdgcn_is_shared(p)) is not picked up by this toolchain, so the TYPE has to be carried to the access.  tools/flat_census.sh lists the kernels
// that still hold flat instructions.)
template <class T> using gptr = __attribute__((address_space(1))) T*;
template <class T> __device__ __forceinline__ gptr<T> as_global(T* p) { return (gptr<T>)p; }
// 16-byte loads through a gptr (uint4 / ulonglong2 are class types whose copy constructors take generic references: the load goes through a builtin vector type)
typedef unsigned int canvas_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long canvas_u64x2 __attribute__((ext_vector_type(2)));
template <class T> __device__ __forceinline__ uint4 gload_uint4(gptr<T> p) { const canvas_u32x4 v = *reinterpret_cast<gptr<const canvas_u32x4>>(p); return make_uint4(v.x, v.y, v.z, v.w); }
typedef unsigned int canvas_u32x2 __attribute__((ext_vector_type(2)));
template <class T> __device__ __forceinline__ uint2 gload_uint2(gptr<T> p) { const canvas_u32x2 v = *reinterpret_cast<gptr<const canvas_u32x2>>(p); return make_uint2(v.x, v.y); }
template <class T> __device__ __forceinline__ void gstore_uint2(gptr<T> p, uint2 v) { canvas_u32x2 w; w.x = v.x; w.y = v.y; *reinterpret_cast<gptr<canvas_u32x2>>(p) = w; }
template <class T> __device__ __forceinline__ void gstore_uint4(gptr<T> p, uint4 v) { canvas_u32x4 w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w; *reinterpret_cast<gptr<canvas_u32x4>>(p) = w; }
template <class T> __device__ __forceinline__ ulonglong2 gload_ulonglong2(gptr<T> p) { const canvas_u64x2 v = *reinterpret_cast<gptr<const canvas_u64x2>>(p); return make_ulonglong2(v.x, v.y); }
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// inclusive wave scan (64 lanes) with DPP: Hillis-Steele inside each 16-lane row (row_shr 1,2,4,8), then the row totals are
// propagated with row_bcast:15 (rows 1,3) and row_bcast:31 (rows 2,3).  6 VALU adds, no LDS-crossbar round trips.
__device__ __forceinline__ uint32_t dpp_add_u32(uint32_t x, uint32_t src, int ctrl_dummy) { return x + src; }
#define CANVAS_DPP(x, ctrl, row_mask) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), (ctrl), (row_mask), 0xF, true))
__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t v) {
    v += CANVAS_DPP(v, 0x111, 0xF);      // row_shr:1
    v += CANVAS_DPP(v, 0x112, 0xF);      // row_shr:2
    v += CANVAS_DPP(v, 0x114, 0xF);      // row_shr:4
    v += CANVAS_DPP(v, 0x118, 0xF);      // row_shr:8
    v += CANVAS_DPP(v, 0x142, 0xA);      // row_bcast:15 into rows 1 and 3
    v += CANVAS_DPP(v, 0x143, 0xC);      // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ uint32_t wave_reduce_add_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ unsigned long long wave_reduce_add_u64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
