// CanvasNormalize, ratio path: weighted-average reference over the control samples, ratio of a sample to the reference
// (LSNorm / Raw), ratios back to counts.  reference: CanvasNormalize/WeightedAverageReferenceGenerator.cs:28-70, BinCounts.cs:36-60,
// LSNormRatioCalculator.cs:20-48, RawRatioCalculator.cs:21-46, CanvasNormalizeUtilities.cs:23-33.
// Element-wise work plus two medians: the medians are exact order statistics from the radix select of CanvasClean (select.hpp), the
// dropped bins are removed with a block-count / scan / scatter compaction.  Which bins are "on target" comes from the caller (the
// manifest parser stays on the host side of the boundary).
#include "common.hpp"
#include "select.hpp"
#include <vector>

__global__ void __launch_bounds__(256) k_norm_keys_f64(const double* __restrict__ v, const int32_t* __restrict__ idx, int64_t n, unsigned long long* __restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) keys[i] = key_of_double(v[idx ? idx[i] : i]);
}
__global__ void __launch_bounds__(256) k_norm_keys_f32(const float* __restrict__ v, const int32_t* __restrict__ idx, int64_t n, unsigned long long* __restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) keys[i] = key_of_double((double)v[idx ? idx[i] : i]);
}
#define NORM_MAX_SAMPLES 64
struct NormPtrs { const double* c[NORM_MAX_SAMPLES]; double w[NORM_MAX_SAMPLES]; };
__global__ void __launch_bounds__(256) k_norm_weighted(NormPtrs P, int nsamples, int64_t n, double* __restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    double w = 0;
    for (int s = 0; s < nsamples; s++) w += P.w[s] * P.c[s][j];      // left to right, product rounded before the add (-ffp-contract=off)
    out[j] = w;
}
__device__ __forceinline__ bool norm_keep(float r, int mode, double minRef, double maxRef) {
    return mode == 0 ? !(r < 1.0f) : !((double)r < minRef) && !((double)r > maxRef);
}
__global__ void __launch_bounds__(256) k_norm_count(const float* __restrict__ ref, int64_t n, int mode, double minRef, double maxRef, uint32_t* __restrict__ blockCnt) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool keep = i < n && norm_keep(ref[i], mode, minRef, maxRef);
    const unsigned long long b = __ballot(keep);
    __shared__ uint32_t s[4];
    if (lane_id() == 0) s[threadIdx.x >> 6] = (uint32_t)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) blockCnt[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}
__global__ void __launch_bounds__(1024) k_norm_scan(uint32_t* __restrict__ blockCnt, int64_t nblocks, long long* __restrict__ total) {
    __shared__ uint32_t sw[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < nblocks; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const uint32_t v = i < nblocks ? blockCnt[i] : 0u;
        const uint32_t inc = wave_inclusive_scan_u32(v);
        if (lane_id() == 63) sw[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t before = 0, all = 0;
        for (int w = 0; w < 16; w++) { if (w < (int)(threadIdx.x >> 6)) before += sw[w]; all += sw[w]; }
        const uint32_t c = carry;
        if (i < nblocks) blockCnt[i] = c + before + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + all;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}
__global__ void __launch_bounds__(256) k_norm_ratio(const float* __restrict__ sample, const float* __restrict__ ref, const int32_t* __restrict__ ploidy, int64_t n, int mode,
                                                    double minRef, double maxRef, double lsf, const uint32_t* __restrict__ blockOff, int32_t* __restrict__ keepIdx,
                                                    float* __restrict__ ratio, float* __restrict__ count) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool keep = i < n && norm_keep(ref[i], mode, minRef, maxRef);
    const unsigned long long b = __ballot(keep);
    __shared__ uint32_t s[4];
    if (lane_id() == 0) s[threadIdx.x >> 6] = (uint32_t)__popcll(b);
    __syncthreads();
    if (!keep) return;
    uint32_t off = blockOff[blockIdx.x];
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) off += s[w];
    off += (uint32_t)__popcll(b & ((1ull << lane_id()) - 1ull));
    const float q = sample[i] / ref[i];                                   // float / float, as in the C#
    const double rt = mode == 0 ? (double)q * lsf : (double)q;
    const float rf = (float)rt;
    const double factor = 40.0 * (double)(ploidy ? ploidy[i] : 2) / 2.0;  // CanvasDiploidBinRatioFactor * ploidy / 2.0
    keepIdx[off] = (int32_t)i; ratio[off] = rf; count[off] = (float)((double)rf * factor);
}

// SortedList<double>.Median() of the n keys (two order statistics)
static int32_t norm_median(canvas_ctx* ctx, const unsigned long long* dKeys, int64_t n, double& med) {
    if (n <= 0) { med = 0; return CANVAS_OK; }
    std::vector<SelQuery> qs; std::vector<unsigned long long> res;
    if (n & 1) qs.push_back({0, 0, n / 2}); else { qs.push_back({0, 0, n / 2 - 1}); qs.push_back({0, 0, n / 2}); }
    int32_t rc = radix_select<unsigned long long>(ctx, dKeys, 1, std::vector<int64_t>{0, n}, qs, res); if (rc) return rc;
    med = (n & 1) ? host_double_of_key(res[0]) : (host_double_of_key(res[0]) + host_double_of_key(res[1])) / 2;
    return CANVAS_OK;
}

extern "C" {

int32_t canvas_normalize_reference(canvas_ctx* ctx, int32_t nsamples, const double* const* h_d_counts, int64_t n, const int32_t* d_on_target_idx, int64_t n_on_target,
                                   double* d_weighted, double* h_weights) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (nsamples <= 0 || nsamples > NORM_MAX_SAMPLES || !h_d_counts || n <= 0 || !d_weighted || (d_on_target_idx && n_on_target <= 0)) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_normalize_reference: bad arguments");
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t nk = d_on_target_idx ? n_on_target : n;
    int32_t rc = canvas_ws_reserve(ctx, (size_t)nk * 8 + 4096); if (rc) return rc;
    unsigned long long* dKeys = (unsigned long long*)ctx->ws;
    NormPtrs P;
    double weightSum = 0;
    for (int s = 0; s < nsamples; s++) {
        hipLaunchKernelGGL(k_norm_keys_f64, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, ctx->stream, h_d_counts[s], d_on_target_idx, nk, dKeys);
        double median; rc = norm_median(ctx, dKeys, nk, median); if (rc) return rc;
        P.c[s] = h_d_counts[s]; P.w[s] = median > 0 ? 1.0 / median : 0;
        weightSum += P.w[s];
    }
    for (int s = 0; s < nsamples; s++) { P.w[s] /= weightSum; if (h_weights) h_weights[s] = P.w[s]; }
    hipLaunchKernelGGL(k_norm_weighted, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, P, nsamples, n, d_weighted);
    CANVAS_HIP_TRY(ctx, hipGetLastError());
    return CANVAS_OK;
}

int32_t canvas_normalize_ratio(canvas_ctx* ctx, int64_t n, const float* d_sample, const float* d_reference, const int32_t* d_on_target_idx, int64_t n_on_target,
                               int32_t mode, double min_ref, double max_ref, const int32_t* d_ploidy, int32_t* d_keep_idx, float* d_ratio, float* d_count,
                               int64_t* h_n_out, double* h_library_size_factor) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (n <= 0 || !d_sample || !d_reference || !d_keep_idx || !d_ratio || !d_count || !h_n_out || (mode != 0 && mode != 1) || (d_on_target_idx && n_on_target <= 0))
        CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_normalize_ratio: bad arguments");
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t nk = d_on_target_idx ? n_on_target : n, nblocks = (n + 255) / 256;
    WsSizer sz; sz.take<unsigned long long>((size_t)nk); sz.take<uint32_t>((size_t)nblocks); sz.take<long long>(1);
    int32_t rc = canvas_ws_reserve(ctx, sz.off + 4096); if (rc) return rc;
    WsCarver ws(ctx->ws);
    unsigned long long* dKeys = ws.take<unsigned long long>((size_t)nk); uint32_t* dBlock = ws.take<uint32_t>((size_t)nblocks); long long* dTotal = ws.take<long long>(1);
    double lsf = 1;
    if (mode == 0) {                                  // LSNormRatioCalculator.cs:29-31
        double sm, rm;
        hipLaunchKernelGGL(k_norm_keys_f32, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, ctx->stream, d_sample, d_on_target_idx, nk, dKeys);
        rc = norm_median(ctx, dKeys, nk, sm); if (rc) return rc;
        hipLaunchKernelGGL(k_norm_keys_f32, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, ctx->stream, d_reference, d_on_target_idx, nk, dKeys);
        rc = norm_median(ctx, dKeys, nk, rm); if (rc) return rc;
        lsf = (sm > 0 && rm > 0) ? rm / sm : 1;
    }
    if (h_library_size_factor) *h_library_size_factor = lsf;
    hipLaunchKernelGGL(k_norm_count, dim3((unsigned)nblocks), dim3(256), 0, ctx->stream, d_reference, n, mode, min_ref, max_ref, dBlock);
    hipLaunchKernelGGL(k_norm_scan, dim3(1), dim3(1024), 0, ctx->stream, dBlock, nblocks, dTotal);
    hipLaunchKernelGGL(k_norm_ratio, dim3((unsigned)nblocks), dim3(256), 0, ctx->stream, d_sample, d_reference, d_ploidy, n, mode, min_ref, max_ref, lsf, dBlock, d_keep_idx, d_ratio, d_count);
    long long total = 0;
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(&total, dTotal, sizeof total, hipMemcpyDeviceToHost, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    CANVAS_HIP_TRY(ctx, hipGetLastError());
    *h_n_out = total;
    return CANVAS_OK;
}

}
