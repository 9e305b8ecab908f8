// Evenness metric of CanvasPartition (--evenness-metric-file): SegmentationInput.GetEvennessScore / reportScoresByWindow
// (CanvasPartition/Segmentation.cs:260-296), computed by WaveletsRunner.Run before the segmentation (WaveletsRunner.cs:58-67) and written as
// "#evenness\t<double>" (CanvasCommon/IO.cs:88-98).  CanvasRunner asks for it on every Somatic-WGS run (CanvasRunner.cs:958-960).
//
// Per window of w - 1 consecutive coverage values (w = 10000 for the IQR test, w = EvennessScoreWindow for the median):
//     S   = the values summed in list order (LINQ Sum / Average of doubles)            -> one sequential FP64 chain per window
//     avg = S / (w - 1)
//     score = sum_{k = 0 .. floor(avg)} (double)#{x >= k} / S, accumulated in k order   -> #{x >= k} = suffix sums of a histogram of floor(x)
// The reference recounts the window for every k (O(w * avg) comparisons per window, ~1e9 per WGS sample); here one workgroup per window stages the
// window through LDS once for the chain and once for the histogram.  The chain is what bounds the kernel: (w - 1) dependent FP64 adds on one lane
// (~0.35 ms for w = 100000), all windows of the genome side by side.  Bit-exact: same operand order for S, exact integer counts, same order over k.
#include "common.hpp"
#include <algorithm>
#include <cmath>

#define EV_CHUNK 1024
#define EV_MAXK 8192         // histogram rows in LDS (32 KB): floor(average coverage of a window) must stay below this

struct EvWindow { int64_t begin; int32_t count; int32_t pad; };

__global__ void __launch_bounds__(256) k_evenness_windows(const double* __restrict__ cov, const EvWindow* __restrict__ wins, double* __restrict__ score, int* __restrict__ status) {
    __shared__ double buf[2][EV_CHUNK];
    __shared__ uint32_t hist[EV_MAXK + 1];
    __shared__ double sSum; __shared__ int sK;
    const EvWindow W = wins[blockIdx.x];
    const int tid = threadIdx.x;
    // ---- S: chunks are staged by all threads (coalesced), thread 0 adds them in list order while the next chunk is in flight
    double acc = 0.0;
    const int nchunks = (W.count + EV_CHUNK - 1) / EV_CHUNK;
    for (int i = tid; i < EV_CHUNK && i < W.count; i += 256) buf[0][i] = cov[W.begin + i];
    __syncthreads();
    for (int c = 0; c < nchunks; c++) {
        const int cur = c & 1, base = c * EV_CHUNK, len = min(EV_CHUNK, W.count - base);
        if (c + 1 < nchunks) { const int nb = base + EV_CHUNK; for (int i = tid; i < EV_CHUNK && nb + i < W.count; i += 256) buf[cur ^ 1][i] = cov[W.begin + nb + i]; }
        if (tid == 0) {
            const double* b = buf[cur];
            int i = 0;
            for (; i + 8 <= len; i += 8) { const double v0 = b[i], v1 = b[i + 1], v2 = b[i + 2], v3 = b[i + 3], v4 = b[i + 4], v5 = b[i + 5], v6 = b[i + 6], v7 = b[i + 7];
                acc += v0; acc += v1; acc += v2; acc += v3; acc += v4; acc += v5; acc += v6; acc += v7; }
            for (; i < len; i++) acc += b[i];
        }
        __syncthreads();
    }
    if (tid == 0) {
        sSum = acc;
        const double average = acc / (double)W.count;
        int K = -1;                                         // the loop "coverageBin <= average" runs for k = 0..K; not at all for a negative or NaN average
        if (average >= 0.0) { K = average >= (double)EV_MAXK ? EV_MAXK : (int)floor(average); }
        sK = K;
    }
    __syncthreads();
    const int K = sK;
    if (K >= EV_MAXK) { if (tid == 0) { *status = 1; score[blockIdx.x] = 0.0; } return; }
    if (K < 0) { if (tid == 0) score[blockIdx.x] = 0.0; return; }       // tmpEvenness stays 0 (finite: the reference adds it)
    // ---- histogram of floor(x) clamped to K: x >= k  <=>  floor(x) >= k for integer k; negative values and NaN match no k
    for (int i = tid; i <= K; i += 256) hist[i] = 0;
    __syncthreads();
    for (int i = tid; i < W.count; i += 256) {
        const double v = cov[W.begin + i];
        if (v >= 0.0) { const int f = v >= (double)K ? K : (int)floor(v); atomicAdd(&hist[f], 1u); }
    }
    __syncthreads();
    if (tid == 0) {
        const double S = sSum;
        uint32_t suffix = 0;
        for (int k = K; k >= 0; k--) { suffix += hist[k]; hist[k] = suffix; }       // hist[k] = #{x >= k}
        double tmpEvenness = 0.0;
        for (int k = 0; k <= K; k++) tmpEvenness += (double)(int)hist[k] / S;       // int / double, k ascending (Segmentation.cs:286-289)
        score[blockIdx.x] = tmpEvenness;
    }
}

// Utilities.Quartiles (CanvasCommon/Utilities.cs:361-419), float arithmetic; needs at least 2 values
static void ev_quartiles(std::vector<float> v, float& q1, float& q3) {
    std::sort(v.begin(), v.end());
    const int iSize = (int)v.size(), iMid = iSize / 2;
    if (iSize % 2 == 0) {
        const int mm = iMid / 2;
        if (iMid % 2 == 0) { q1 = (v[mm - 1] + v[mm]) / 2; q3 = (v[iMid + mm - 1] + v[iMid + mm]) / 2; }
        else { q1 = v[mm]; q3 = v[mm + iMid]; }
    } else if ((iSize - 1) % 4 == 0) { const int n = (iSize - 1) / 4; q1 = (v[n - 1] * 0.25f) + (v[n] * 0.75f); q3 = (v[3 * n] * 0.75f) + (v[3 * n + 1] * 0.25f); }
    else { const int n = (iSize - 3) / 4; q1 = (v[n] * 0.75f) + (v[n + 1] * 0.25f); q3 = (v[3 * n + 1] * 0.25f) + (v[3 * n + 2] * 0.75f); }
}

extern "C" int32_t canvas_evenness_score(canvas_ctx* ctx, int32_t nchr, const double* d_cov, const int64_t* h_chr_offset, int32_t window_size, double* h_score, int32_t* h_valid) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (nchr <= 0 || !h_chr_offset || !h_score || !h_valid || window_size < 2) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_evenness_score: bad arguments");
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    *h_valid = 0; *h_score = 0.0;
    const int windowSizeIQR = 10000; const double IQRthreshold = 0.015;
    std::vector<EvWindow> wins; size_t nIqr = 0;
    for (int pass = 0; pass < 2; pass++) {
        const int w = pass == 0 ? windowSizeIQR : window_size;
        for (int c = 0; c < nchr; c++) {
            const int64_t n = h_chr_offset[c + 1] - h_chr_offset[c];
            for (int64_t index = 0; index < n - w; index += w) wins.push_back({h_chr_offset[c] + index, w - 1, 0});
        }
        if (pass == 0) nIqr = wins.size();
    }
    if (nIqr < 2 || wins.size() == nIqr) return CANVAS_OK;            // Quartiles / Median throw in the reference: no score (WaveletsRunner.cs:58-67)
    if (!d_cov) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_evenness_score: no coverage");
    const size_t nw = wins.size();
    WsSizer sz; sz.take<EvWindow>(nw); sz.take<double>(nw); sz.take<int>(1);
    int32_t rc = canvas_ws_reserve(ctx, sz.off + 4096); if (rc) return rc;
    WsCarver ws(ctx->ws);
    EvWindow* dWins = ws.take<EvWindow>(nw); double* dScore = ws.take<double>(nw); int* dStatus = ws.take<int>(1);
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dWins, wins.data(), nw * sizeof(EvWindow), hipMemcpyHostToDevice, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipMemsetAsync(dStatus, 0, 4, ctx->stream));
    hipLaunchKernelGGL(k_evenness_windows, dim3((unsigned)nw), dim3(256), 0, ctx->stream, d_cov, dWins, dScore, dStatus);
    std::vector<double> sc(nw); int status = 0;
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(sc.data(), dScore, nw * 8, hipMemcpyDeviceToHost, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(&status, dStatus, 4, hipMemcpyDeviceToHost, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    CANVAS_HIP_TRY(ctx, hipGetLastError());
    if (status) CANVAS_FAIL(ctx, CANVAS_ERR_UNSUPPORTED, "canvas_evenness_score: a window's average coverage is 8192 or more");
    std::vector<float> iqr; std::vector<double> med;
    for (size_t i = 0; i < nw; i++) { const double v = sc[i]; if (std::isinf(v) || std::isnan(v)) continue; if (i < nIqr) iqr.push_back((float)v); else med.push_back(v); }
    if (iqr.size() < 2 || med.empty()) return CANVAS_OK;
    float q1, q3; ev_quartiles(iqr, q1, q3);
    std::sort(med.begin(), med.end());
    const size_t m = med.size();
    const double median = (m % 2) ? med[m / 2] : (med[m / 2 - 1] + med[m / 2]) / 2;
    *h_score = (q3 - q1 > IQRthreshold) ? q3 * 100.0 : median * 100.0;
    *h_valid = 1;
    return CANVAS_OK;
}
