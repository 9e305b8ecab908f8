// CanvasBin per-chromosome preparation steps (CanvasBin.BinOneGenomicInterval, CanvasBin.cs:765-792) as streaming kernels:
//   k_mask_from_fasta      InitializeAlignmentArrays (:183-200): possible[i] = char.IsUpper(referenceBases[i])      1.125 B/base
//   k_mask_exclude         ExcludeTagsOverlappingFilterFile (:668-692): clear the bits of every BED interval          O(interval bases / 64)
//   k_screen_hits          ScreenObservedTags (:699-716): observed[i] = 0 where !possible[i]                          2.125 B/base
// (The BAM loop that fills the hit array, :207-275, is host I/O and stays in the C# module.)
#include "common.hpp"

__global__ void __launch_bounds__(256) k_mask_from_fasta(const uint8_t* __restrict__ bases, int64_t len, uint64_t* __restrict__ mask) {
    // one 64-bit mask word (64 bases) per thread; 4 x 16-byte loads
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t p0 = w * 64;
    if (p0 >= len) return;
    uint64_t m = 0;
    if (p0 + 64 <= len) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint4 v = *reinterpret_cast<const uint4*>(bases + p0 + q * 16);
            uint32_t ws[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                // per byte: 'A' <= b <= 'Z'  (char.IsUpper on ASCII)
                uint32_t x = ws[k];
                uint32_t ge = ((x | 0x80808080u) - 0x41414141u) & ~x;          // bit7 set iff b >= 0x41 and b < 0x80
                uint32_t le = ((0x5A5A5A5Au | 0x80808080u) - (x & 0x7F7F7F7Fu)) & ~x;   // bit7 set iff (b & 0x7f) <= 0x5A and b < 0x80
                uint32_t up = (ge & le & 0x80808080u) >> 7;                     // 0x01 per upper-case byte
                uint32_t bits = ((up * 0x01020408u) >> 24) & 0xFu;
                m |= (uint64_t)bits << (q * 16 + k * 4);
            }
        }
    } else {
        for (int i = 0; i < 64 && p0 + i < len; i++) { uint8_t b = bases[p0 + i]; if (b >= 'A' && b <= 'Z') m |= 1ull << i; }
    }
    mask[w] = m;
}

struct Interval { int64_t a, b; };
__global__ void __launch_bounds__(256) k_mask_exclude(const Interval* __restrict__ iv, int64_t len, unsigned long long* __restrict__ mask) {
    Interval I = iv[blockIdx.x];
    int64_t a = I.a < 0 ? 0 : I.a, b = I.b > len ? len : I.b;
    if (a >= b) return;
    const int64_t w0 = a >> 6, w1 = (b - 1) >> 6;
    for (int64_t w = w0 + threadIdx.x; w <= w1; w += 256) {
        unsigned long long keep = 0;
        if (w == w0 && (a & 63)) keep |= (1ull << (a & 63)) - 1ull;                    // bits below a stay
        if (w == w1 && (b & 63)) keep |= ~((1ull << (b & 63)) - 1ull);                 // bits at/above b stay
        atomicAnd(&mask[w], keep);
    }
}

__global__ void __launch_bounds__(256) k_screen_hits(uint8_t* __restrict__ hits, const uint64_t* __restrict__ mask, int64_t len) {
    // 16 positions per thread
    const int64_t p = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (p >= len) return;
    const uint32_t m16 = (uint32_t)((mask[p >> 6] >> (p & 63)) & 0xFFFFull);
    if (p + 16 <= len) {
        uint4 v = *reinterpret_cast<uint4*>(hits + p);
        uint32_t ws[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) { uint32_t x = __umul24((m16 >> (4 * k)) & 0xFu, 0x204081u) & 0x01010101u; ws[k] &= (x << 8) - x; }
        *reinterpret_cast<uint4*>(hits + p) = make_uint4(ws[0], ws[1], ws[2], ws[3]);
    } else {
        for (int i = 0; i < 16 && p + i < len; i++) if (!((m16 >> i) & 1u)) hits[p + i] = 0;
    }
}

extern "C" {

int32_t canvas_mask_from_fasta(canvas_ctx* ctx, const uint8_t* d_bases, int64_t len, uint64_t* d_mask) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (len <= 0 || !d_bases || !d_mask) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_mask_from_fasta: bad arguments");
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    { int32_t rcf = canvas_upload_fence(ctx); if (rcf) return rcf; }
    int64_t words = (len + 63) / 64;
    hipLaunchKernelGGL(k_mask_from_fasta, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, ctx->stream, d_bases, len, d_mask);
    CANVAS_HIP_TRY(ctx, hipGetLastError());
    return CANVAS_OK;
}

int32_t canvas_mask_exclude_intervals(canvas_ctx* ctx, uint64_t* d_mask, int64_t len, int32_t n, const int32_t* h_start, const int32_t* h_stop) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (len <= 0 || !d_mask || n < 0 || (n > 0 && (!h_start || !h_stop))) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_mask_exclude_intervals: bad arguments");
    if (n == 0) return CANVAS_OK;
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    { int32_t rcf = canvas_upload_fence(ctx); if (rcf) return rcf; }
    std::vector<Interval> iv(n);
    for (int i = 0; i < n; i++) iv[i] = Interval{h_start[i], h_stop[i]};
    int32_t rc = canvas_ws_reserve(ctx, (size_t)n * sizeof(Interval) + 256); if (rc) return rc;
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(ctx->ws, iv.data(), (size_t)n * sizeof(Interval), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_mask_exclude, dim3(n), dim3(256), 0, ctx->stream, (const Interval*)ctx->ws, len, (unsigned long long*)d_mask);
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    CANVAS_HIP_TRY(ctx, hipGetLastError());
    return CANVAS_OK;
}

int32_t canvas_screen_hits(canvas_ctx* ctx, uint8_t* d_hits, const uint64_t* d_mask, int64_t len) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (len <= 0 || !d_hits || !d_mask) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_screen_hits: bad arguments");
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    { int32_t rcf = canvas_upload_fence(ctx); if (rcf) return rcf; }
    int64_t groups = (len + 15) / 16;
    hipLaunchKernelGGL(k_screen_hits, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, ctx->stream, d_hits, d_mask, len);
    CANVAS_HIP_TRY(ctx, hipGetLastError());
    return CANVAS_OK;
}

}  // extern "C"
