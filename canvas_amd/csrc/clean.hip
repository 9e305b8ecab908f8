// CanvasClean on MI355X: CanvasClean.Main (CanvasClean/CanvasClean.cs:415-533) over the whole-genome bin SoA.
//
// Layout in HBM: struct-of-arrays in file order — chr i32, start i32, stop i32, gc i32, count f32 (SampleGenomicBin.Count is
// float), plus CountDeviation f64 kept in the workspace.  All arrays fit the 256 MiB Infinity Cache at WGS sizes (5.4 M bins
// x 28 B = 150 MB), so the stages are launch/latency bound rather than HBM bound; every stage is a streaming pass.
//
// Every reduction that feeds a decision is an exact order statistic (radix select, select.hpp) or an integer histogram, and
// every per-bin floating expression is evaluated element-wise with the reference's operand order and types
// (-ffp-contract=off), so MedianByGC cleaning is bit-identical to the CPU oracle.
//
// Host logic (this file, between launches) only does what the reference does once per file on <= 101 GC buckets / <= a few
// hundred chromosomes: thresholds, medians of two selected values, quartile interpolation (Utilities.cs:361-419).
//
// GC buckets holding fewer than 100 autosomal bins (reachable with -w < 100 on < 10100 bins, and in LOESS mode where no GC
// strip runs) take the neighbour-weighted quantiles of CanvasClean.cs:107-132: the few bucket ranges involved are copied to the
// host and weighted there (a per-file O(hundreds) computation next to the thresholds above).
// Not built (returns CANVAS_ERR_UNSUPPORTED): manifests (-t).
#include "common.hpp"
#include <chrono>
#include "select.hpp"
#include "loess.hpp"
#include <algorithm>
#include <cmath>
#include <thread>

#define NGC 101
#define CBLK 2048   // elements per compaction block

struct Soa { int32_t *chr, *start, *stop, *gc; float* count; double* dev; };
struct GSoa { gptr<int32_t> chr, start, stop, gc; gptr<float> count; gptr<double> dev; };      // the same arrays as global-address-space pointers (common.hpp: as_global)
__device__ __forceinline__ GSoa as_global(const Soa& s) { return GSoa{as_global(s.chr), as_global(s.start), as_global(s.stop), as_global(s.gc), as_global(s.count), as_global(s.dev)}; }

// ---------------------------------------------------------------- flag kernels
__global__ void __launch_bounds__(256) k_keys_size(const int32_t* __restrict__ start, const int32_t* __restrict__ stop, int64_t n, uint32_t* __restrict__ keys) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) keys[i] = (uint32_t)(stop[i] - start[i]) ^ 0x80000000u;   // order-preserving image of int32
}
// threshold = the selected order statistic, read from the select's device result (no host round trip)
__global__ void __launch_bounds__(256) k_flags_size(const int32_t* __restrict__ start, const int32_t* __restrict__ stop, int64_t n, const unsigned long long* __restrict__ dKey, uint8_t* __restrict__ flags) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int32_t thresh = (int32_t)((uint32_t)dKey[0] ^ 0x80000000u);
    if (i < n) flags[i] = (stop[i] - start[i]) <= thresh;       // CanvasClean.cs:349-352
}
// SignificantlyDifferent (CanvasClean.cs:363-381)
__device__ __forceinline__ bool sig_diff(float a, float b) {
    double mu = ((double)a + (double)b) / 2;
    if (a + b == 0) return false;
    double da = (double)a - mu, db = (double)b - mu;
    double chi2 = (da * da + db * db) / mu;
    return chi2 > 6.635;
}
// RemoveOutliers (CanvasClean.cs:387-413)
// dN != NULL: the bin count is still on the device (the previous compaction was not synchronised); the grid covers an upper bound
__global__ void __launch_bounds__(256) k_flags_outlier(const int32_t* __restrict__ chr, const float* __restrict__ count, int64_t n, const unsigned long long* __restrict__ dN, uint8_t* __restrict__ flags) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (dN) n = (int64_t)*dN;
    if (i >= n) return;
    bool hasPrev = i > 0, hasNext = i < n - 1;
    int32_t c = chr[i];
    bool prevSame = hasPrev && chr[i - 1] == c, nextSame = hasNext && chr[i + 1] == c;
    bool keep;
    if ((hasPrev && !prevSame) && (hasNext && !nextSame)) keep = false;
    else {
        float v = count[i];
        keep = (prevSame && !sig_diff(v, count[i - 1])) || (nextSame && !sig_diff(v, count[i + 1])) || (!hasPrev && !hasNext);
    }
    flags[i] = keep;
}
__global__ void __launch_bounds__(256) k_flags_gc(const int32_t* __restrict__ gc, int64_t n, const uint8_t* __restrict__ keepGc, uint8_t* __restrict__ flags) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) flags[i] = keepGc[gc[i]];
}
// RemoveBinsWithExtremeLocalSD (CanvasClean.cs:308-322): drop when CountDeviation > threshold*2.0 (localSDaverage > 5 checked on host)
__global__ void __launch_bounds__(256) k_flags_localsd(const double* __restrict__ dev, int64_t n, double limit, uint8_t* __restrict__ flags) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) flags[i] = !(dev[i] > limit);
}

// ---------------------------------------------------------------- stable compaction (count / scan / scatter)
__global__ void __launch_bounds__(256) k_block_count(const uint8_t* __restrict__ flags, int64_t n, uint32_t* __restrict__ blockCnt, const unsigned long long* __restrict__ dN = nullptr) {
    __shared__ uint32_t sh[4];
    if (dN) n = (int64_t)*dN;
    int64_t base = (int64_t)blockIdx.x * CBLK;
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < CBLK / 256; j++) { int64_t i = base + j * 256 + threadIdx.x; if (i < n) c += flags[i]; }
    c = wave_reduce_add_u32(c);
    if (lane_id() == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blockCnt[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ void __launch_bounds__(1024) k_scan_blocks(uint32_t* __restrict__ blockCnt, int nblocks, unsigned long long* __restrict__ total) {
    __shared__ uint32_t sh[17];
    uint32_t carry = 0;
    for (int base = 0; base < nblocks; base += 1024) {
        int i = base + threadIdx.x;
        uint32_t v = i < nblocks ? blockCnt[i] : 0;
        uint32_t inc = wave_inclusive_scan_u32(v);
        int w = threadIdx.x >> 6;
        if (lane_id() == 63) sh[w] = inc;
        __syncthreads();
        if (threadIdx.x == 0) { uint32_t s = 0; for (int k = 0; k < 16; k++) { uint32_t t = sh[k]; sh[k] = s; s += t; } sh[16] = s; }
        __syncthreads();
        if (i < nblocks) blockCnt[i] = carry + sh[w] + inc - v;
        carry += sh[16];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}
__global__ void __launch_bounds__(256) k_scatter(const uint8_t* __restrict__ flags, const uint32_t* __restrict__ blockOff, int64_t n, Soa src, Soa dst, const unsigned long long* __restrict__ dN = nullptr) {
    __shared__ uint32_t sh[4];
    if (dN) n = (int64_t)*dN;
    int64_t base = (int64_t)blockIdx.x * CBLK;
    uint32_t running = blockOff[blockIdx.x];
    for (int j = 0; j < CBLK / 256; j++) {
        int64_t i = base + j * 256 + threadIdx.x;
        uint32_t f = (i < n) ? flags[i] : 0;
        uint32_t inc = wave_inclusive_scan_u32(f);
        if (lane_id() == 63) sh[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t woff = 0, tot = 0;
        for (int k = 0; k < 4; k++) { if (k < (int)(threadIdx.x >> 6)) woff += sh[k]; tot += sh[k]; }
        if (f) {
            uint32_t d = running + woff + inc - 1;
            dst.chr[d] = src.chr[i]; dst.start[d] = src.start[i]; dst.stop[d] = src.stop[i]; dst.gc[d] = src.gc[i];
            dst.count[d] = src.count[i]; dst.dev[d] = src.dev[i];
        }
        running += tot;
        __syncthreads();
    }
}

// ---------------------------------------------------------------- GC histogram / grouping
// hist[0..100] = autosomal bins per GC value (what the reference's filters look at), hist[101..201] = the other bins: with both the host
// knows how many bins a GC strip keeps without counting flags on the device
__global__ void __launch_bounds__(256) k_gc_hist(const int32_t* __restrict__ chr, const int32_t* __restrict__ gc, const uint8_t* __restrict__ isAuto, int64_t n, uint32_t* __restrict__ hist) {
    __shared__ uint32_t lh[2 * NGC];
    if (threadIdx.x < 2 * NGC) lh[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        atomicAdd(&lh[(isAuto[chr[i]] ? 0 : NGC) + gc[i]], 1u);
    __syncthreads();
    if (threadIdx.x < 2 * NGC && lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], lh[threadIdx.x]);
}
// scatter the indices of autosomal bins into per-GC contiguous groups (order inside a group is irrelevant: only order statistics are taken)
__global__ void __launch_bounds__(256) k_group_by_gc(const int32_t* __restrict__ chr, const int32_t* __restrict__ gc, const uint8_t* __restrict__ isAuto, int64_t n,
                                                     const uint32_t* __restrict__ segOff, uint32_t* __restrict__ cursor, uint32_t* __restrict__ gidx) {
    __shared__ uint32_t lcnt[NGC], lbase[NGC];
    if (threadIdx.x < NGC) lcnt[threadIdx.x] = 0;
    __syncthreads();
    int64_t base = (int64_t)blockIdx.x * CBLK;
    uint32_t myRank[CBLK / 256];
    int myGc[CBLK / 256];
#pragma unroll
    for (int j = 0; j < CBLK / 256; j++) {
        int64_t i = base + j * 256 + threadIdx.x;
        myGc[j] = -1;
        if (i < n && isAuto[chr[i]]) { myGc[j] = gc[i]; myRank[j] = atomicAdd(&lcnt[myGc[j]], 1u); }
    }
    __syncthreads();
    if (threadIdx.x < NGC && lcnt[threadIdx.x]) lbase[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], lcnt[threadIdx.x]);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < CBLK / 256; j++) {
        int64_t i = base + j * 256 + threadIdx.x;
        if (myGc[j] >= 0) gidx[segOff[myGc[j]] + lbase[myGc[j]] + myRank[j]] = (uint32_t)i;
    }
}
__global__ void __launch_bounds__(256) k_gather_keys(const float* __restrict__ count, const uint32_t* __restrict__ gidx, int64_t m, uint32_t* __restrict__ keys) {
    int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < m) keys[j] = key_of_float(count[gidx[j]]);
}

// ---------------------------------------------------------------- element-wise normalisation
// NormalizeByGC apply (CanvasClean.cs:190-195): count = (float)(globalMedian * (double)count / median) when median > 0
__global__ void __launch_bounds__(256) k_apply_gc(float* __restrict__ count, const int32_t* __restrict__ gc, int64_t n, const double* __restrict__ medians, double globalMedian) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double median = medians[gc[i]];
    if (median > 0) count[i] = (float)(globalMedian * (double)count[i] / median);
}
// NormalizeVarianceByGC apply (CanvasClean.cs:84-94), float32 arithmetic
struct VarTab { float localIQR[NGC]; float med[NGC]; float globalIQR; };
__global__ void __launch_bounds__(256) k_apply_var(float* __restrict__ count, const int32_t* __restrict__ gc, int64_t n, const VarTab* __restrict__ tab) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int g = gc[i];
    float globalIQR = tab->globalIQR;
    float scaledLocalIqr = tab->localIQR[g] * 0.8f;
    if (globalIQR >= scaledLocalIqr) return;
    float iqrRatio = scaledLocalIqr / globalIQR;
    float m = tab->med[g];
    count[i] = m + (count[i] - m) / iqrRatio;
}

// ---------------------------------------------------------------- local SD (CanvasClean.cs:268-298)
// one thread per window of 20 consecutive count differences; sequential double arithmetic exactly as
// Utilities.StandardDeviation(double[], start, end) (Utilities.cs:246-262)
template <class PC, class PD>          // plain pointers, or gptr<> when they come out of a table in device memory (clean_fast.hpp)
__device__ __forceinline__ void local_sd_window(PC count, int64_t w, PD sd, PD dev) {
    int64_t s = w * 20;
    double d[20];
    float prev = count[s];
#pragma unroll
    for (int k = 0; k < 20; k++) { float nx = count[s + k + 1]; d[k] = (double)(nx - prev); prev = nx; }
    double sum = 0;
#pragma unroll
    for (int k = 0; k < 20; k++) sum += d[k];
    double mu = sum / 20;
    double s2 = 0;
#pragma unroll
    for (int k = 0; k < 20; k++) { double df = d[k] - mu; s2 += df * df; }
    double v = sqrt(s2 / 19);
    sd[w] = v;
#pragma unroll
    for (int k = 0; k < 20; k++) dev[s + k] = v;
}
__device__ __forceinline__ void local_sd_body(const float* __restrict__ count, int64_t nW, double* __restrict__ sd, double* __restrict__ dev, const unsigned long long* __restrict__ dN) {
    int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (dN) { const int64_t D = (int64_t)*dN - 1; nW = D >= 1 ? (D - 1) / 20 : 0; }      // the bin count is still on the device: grid = upper bound
    if (w >= nW) return;
    local_sd_window(count, w, sd, dev);
}
__global__ void __launch_bounds__(256) k_local_sd(const float* __restrict__ count, int64_t nW, double* __restrict__ sd, double* __restrict__ dev, const unsigned long long* __restrict__ dN = nullptr) {
    local_sd_body(count, nW, sd, dev, dN);
}
__global__ void __launch_bounds__(256) k_copy_soa(Soa src, Soa dst, int64_t n) {     // the five caller-visible columns in one launch
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    dst.chr[i] = src.chr[i]; dst.start[i] = src.start[i]; dst.stop[i] = src.stop[i]; dst.gc[i] = src.gc[i]; dst.count[i] = src.count[i];
}
// CountDeviation = -1 for every bin, and the range check the reference gets for free from its managed arrays: a gc outside 0..100 (or a
// chromosome index outside the table) would index past the GC tables below, where the C# throws IndexOutOfRangeException
__global__ void __launch_bounds__(256) k_init_validate(const int32_t* __restrict__ chr, const int32_t* __restrict__ gc, int64_t n, int nchr, double* __restrict__ dev, unsigned int* __restrict__ bad) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    dev[i] = -1.0;
    const int32_t g = gc[i], c = chr[i];
    if ((uint32_t)g > 100u || (uint32_t)c >= (uint32_t)nchr) *bad = 1u;
}
__global__ void __launch_bounds__(256) k_keys_f64(const double* __restrict__ v, int64_t n, unsigned long long* __restrict__ keys) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) keys[i] = key_of_double(v[i]);
}
// |sd - median(run)| keys; runStart (window index) sorted ascending
__global__ void __launch_bounds__(256) k_absdev_keys(const double* __restrict__ sd, int64_t nW, const int64_t* __restrict__ runStart, const double* __restrict__ runMedian,
                                                     int nruns, unsigned long long* __restrict__ keys) {
    int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= nW) return;
    int lo = 0, hi = nruns - 1;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (runStart[mid] <= w) lo = mid; else hi = mid - 1; }
    keys[w] = key_of_double(fabs(sd[w] - runMedian[lo]));
}
// chromosome run boundaries of the bin list: positions i with chr[i] != chr[i-1]
// pos[k] = (position << 20) | chromosome index (chromosome ids < 2^20, positions < 2^43), so one sort orders the records
__global__ void __launch_bounds__(256) k_run_bounds(const int32_t* __restrict__ chr, int64_t n, unsigned int* __restrict__ cnt, long long* __restrict__ pos, int cap, const unsigned long long* __restrict__ dN = nullptr) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (dN) n = (int64_t)*dN;
    if (i >= n) return;
    const int32_t c = chr[i];
    if (i == 0 || c != chr[i - 1]) { unsigned int k = atomicAdd(cnt, 1u); if ((int)k < cap) pos[k] = (long long)((i << 20) | (long long)(c & 0xFFFFF)); }
}

// Utilities.Mad per chromosome run of the window SDs (CanvasClean.cs:243-258, Utilities.cs Median/Mad): one workgroup per run
__device__ __forceinline__ void run_mad_body(const double* __restrict__ sd, const int64_t* __restrict__ runStart, double* __restrict__ outMad, const int* __restrict__ nrunsDev) {
    __shared__ uint32_t sH[2][256];
    __shared__ unsigned long long sPre[2], sK[2];
    if (nrunsDev && (int)blockIdx.x >= *nrunsDev) return;         // device-built run table: the grid is an upper bound
    const int64_t lo = runStart[blockIdx.x], hi = runStart[blockIdx.x + 1], cnt = hi - lo;
    if (cnt <= 0) { if (threadIdx.x == 0) outMad[blockIdx.x] = 0.0; return; }
    const unsigned long long r1 = (unsigned long long)(cnt / 2), r0 = (cnt % 2) ? r1 : r1 - 1;
    wg_select2([&](int64_t i) { return key_of_double(sd[i]); }, lo, hi, r0, r1, sH, sPre, sK);
    const double median = (cnt % 2) ? double_of_key(sPre[1]) : (double_of_key(sPre[0]) + double_of_key(sPre[1])) / 2;
    __syncthreads();
    wg_select2([&](int64_t i) { return key_of_double(fabs(sd[i] - median)); }, lo, hi, r0, r1, sH, sPre, sK);
    if (threadIdx.x == 0) outMad[blockIdx.x] = (cnt % 2) ? double_of_key(sPre[1]) : (double_of_key(sPre[0]) + double_of_key(sPre[1])) / 2;
}
__global__ void __launch_bounds__(1024) k_run_mad(const double* __restrict__ sd, const int64_t* __restrict__ runStart, double* __restrict__ outMad, const int* __restrict__ nrunsDev = nullptr) {
    run_mad_body(sd, runStart, outMad, nrunsDev);
}

// ---------------------------------------------------------------- host helpers (scalar logic of the reference)
static __host__ __device__ inline float median_from_two(float a, float b) { return (a + b) / 2; }   // SortedList<float>.Median(), even length

// Utilities.Quartiles (CanvasCommon/Utilities.cs:361-419): which order statistics are needed for length n
struct QuartIdx { int64_t idx[6]; int n; };
static __host__ __device__ QuartIdx quartile_indices(int64_t iSize) {
    QuartIdx q; q.n = 0;
    auto add = [&](int64_t v) { q.idx[q.n++] = v; };
    int64_t iMid = iSize / 2;
    if (iSize % 2 == 0) {
        int64_t mm = iMid / 2;
        add(iMid - 1); add(iMid);
        if (iMid % 2 == 0) { add(mm - 1); add(mm); add(iMid + mm - 1); add(iMid + mm); }
        else { add(mm); add(mm + iMid); }
    } else {
        add(iMid);
        if ((iSize - 1) % 4 == 0) { int64_t n = (iSize - 1) / 4; add(n - 1); add(n); add(3 * n); add(3 * n + 1); }
        else { int64_t n = (iSize - 3) / 4; add(n); add(n + 1); add(3 * n + 1); add(3 * n + 2); }
    }
    return q;
}
static __host__ __device__ void quartiles_from_values(int64_t iSize, const float* v, float& q1, float& q2, float& q3) {
    int64_t iMid = iSize / 2;
    if (iSize % 2 == 0) {
        q2 = (v[0] + v[1]) / 2;
        if (iMid % 2 == 0) { q1 = (v[2] + v[3]) / 2; q3 = (v[4] + v[5]) / 2; }
        else { q1 = v[2]; q3 = v[3]; }
    } else {
        q2 = v[0];
        if ((iSize - 1) % 4 == 0) { q1 = (v[1] * 0.25f) + (v[2] * 0.75f); q3 = (v[3] * 0.75f) + (v[4] * 0.25f); }
        else { q1 = (v[1] * 0.75f) + (v[2] * 0.25f); q3 = (v[3] * 0.25f) + (v[4] * 0.75f); }
    }
}

struct CleanState {
    canvas_ctx* ctx;
    int64_t n;
    Soa cur, alt;            // cur holds the live bins; alt = destination of the next compaction (pick_alt)
    Soa bufs[3];             // [0] the caller's arrays, [1] and [2] scratch
    uint8_t* flags; uint32_t* blockCnt; unsigned long long* dTotal;
    uint32_t* keys32; unsigned long long* keys64; uint32_t* gidx;
    uint8_t* dIsAuto;
};

static inline unsigned nblk(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

// Destination of the next compaction.  The result must end in the caller's arrays: the two filters that always run go caller -> scratch 1
// -> scratch 2 and the later, data-dependent compactions (GC strip, local-SD filter) aim at the caller's arrays, so the usual sequence
// (size filter, outlier filter, GC strip) needs no copy at the end.
static void pick_alt(CleanState& st, bool preferCaller) {
    const int ci = st.cur.chr == st.bufs[0].chr ? 0 : (st.cur.chr == st.bufs[1].chr ? 1 : 2);
    st.alt = (preferCaller && ci != 0) ? st.bufs[0] : st.bufs[ci == 1 ? 2 : 1];
}
// compaction cur -> alt by st.flags; swaps; returns new n (one sync)
static int32_t compact(CleanState& st, bool preferCaller) {
    canvas_ctx* ctx = st.ctx;
    if (st.n == 0) return CANVAS_OK;
    pick_alt(st, preferCaller);
    int nb = (int)nblk(st.n, CBLK);
    hipLaunchKernelGGL(k_block_count, dim3(nb), dim3(256), 0, ctx->stream, st.flags, st.n, st.blockCnt);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, ctx->stream, st.blockCnt, nb, st.dTotal);
    hipLaunchKernelGGL(k_scatter, dim3(nb), dim3(256), 0, ctx->stream, st.flags, st.blockCnt, st.n, st.cur, st.alt);
    unsigned long long tot = 0;
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(&tot, st.dTotal, 8, hipMemcpyDeviceToHost, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    CANVAS_HIP_TRY(ctx, hipGetLastError());
    std::swap(st.cur, st.alt);
    st.n = (int64_t)tot;
    return CANVAS_OK;
}

// the same without the host round trip: the new count goes to *dOut, the old one may itself still be on the device (dN); st.n stays an upper bound
static void compact_launch(CleanState& st, const unsigned long long* dN, unsigned long long* dOut, bool preferCaller) {
    canvas_ctx* ctx = st.ctx;
    pick_alt(st, preferCaller);
    int nb = (int)nblk(st.n, CBLK);
    hipLaunchKernelGGL(k_block_count, dim3(nb), dim3(256), 0, ctx->stream, st.flags, st.n, st.blockCnt, dN);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, ctx->stream, st.blockCnt, nb, dOut);
    hipLaunchKernelGGL(k_scatter, dim3(nb), dim3(256), 0, ctx->stream, st.flags, st.blockCnt, st.n, st.cur, st.alt, dN);
    std::swap(st.cur, st.alt);
}

struct GcGroups {              // autosomal bins grouped by GC
    uint32_t hist[NGC];
    std::vector<int64_t> segOff;   // NGC+1
    int64_t nauto;
};

static int32_t gc_histogram(CleanState& st, uint32_t* dHist, uint32_t* hist, uint32_t* histOther = nullptr) {
    canvas_ctx* ctx = st.ctx;
    uint32_t both[2 * NGC];
    CANVAS_HIP_TRY(ctx, hipMemsetAsync(dHist, 0, 2 * NGC * 4, ctx->stream));
    if (st.n > 0) hipLaunchKernelGGL(k_gc_hist, dim3(std::min<unsigned>(nblk(st.n, 256), 1024u)), dim3(256), 0, ctx->stream, st.cur.chr, st.cur.gc, st.dIsAuto, st.n, dHist);
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(both, dHist, sizeof both, hipMemcpyDeviceToHost, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(hist, both, NGC * 4);
    if (histOther) memcpy(histOther, both + NGC, NGC * 4);
    return CANVAS_OK;
}

// build gidx for the current bins (call after any compaction)
static int32_t group_by_gc(CleanState& st, GcGroups& g, uint32_t* dSegOff, uint32_t* dCursor) {
    canvas_ctx* ctx = st.ctx;
    g.segOff.assign(NGC + 1, 0);
    for (int i = 0; i < NGC; i++) g.segOff[i + 1] = g.segOff[i] + g.hist[i];
    g.nauto = g.segOff[NGC];
    uint32_t so[NGC + 1];
    for (int i = 0; i <= NGC; i++) so[i] = (uint32_t)g.segOff[i];
    { int32_t rc = canvas_h2d_small(ctx, dSegOff, so, sizeof so); if (rc) return rc; }
    CANVAS_HIP_TRY(ctx, hipMemsetAsync(dCursor, 0, NGC * 4, ctx->stream));
    if (st.n > 0) hipLaunchKernelGGL(k_group_by_gc, dim3(nblk(st.n, CBLK)), dim3(256), 0, ctx->stream, st.cur.chr, st.cur.gc, st.dIsAuto, st.n, dSegOff, dCursor, st.gidx);
    return CANVAS_OK;
}


// GetWeightedCounts (CanvasClean.cs:107-132) + Utilities.WeightedQuantiles (Utilities.cs:493-520) for GC buckets with fewer
// than 100 autosomal bins.  st.keys32 holds the grouped count keys (bucket by bucket, file order inside a bucket), so the
// reference's insertion order — bucket, bucket+1, bucket-1, bucket+2, ... — is a concatenation of contiguous key ranges.
// The ranges are copied to the host once (merged), the weighting itself is <= a few thousand elements per bucket.
struct WQuant { double q[3]; };
static int32_t weighted_quantiles_sparse(CleanState& st, const GcGroups& g, const std::vector<int>& buckets, const float* probs, int nprobs,
                                         std::vector<WQuant>& out) {
    canvas_ctx* ctx = st.ctx;
    out.assign(buckets.size(), WQuant{{0, 0, 0}});
    if (buckets.empty()) return CANVAS_OK;
    struct Piece { int gc; float w; };
    std::vector<std::vector<Piece>> plan(buckets.size());
    std::vector<uint8_t> need(NGC, 0);
    for (size_t b = 0; b < buckets.size(); b++) {
        const int gcBin = buckets[b];
        int64_t have = 0; int radius = 0; float weight = 1;
        while (have < 100) {
            int hi = gcBin + radius, lo = gcBin - radius;
            if (hi >= NGC && lo < 0) break;
            if (hi < NGC) { plan[b].push_back({hi, weight}); have += g.hist[hi]; }
            if (lo != hi && lo >= 0) { plan[b].push_back({lo, weight}); have += g.hist[lo]; }
            radius++; weight /= 2;
        }
        for (auto& p : plan[b]) if (g.hist[p.gc]) need[p.gc] = 1;
    }
    // download merged runs of needed buckets
    std::vector<uint32_t> hostKeys; std::vector<int64_t> hostAt(NGC, -1);
    for (int gc = 0; gc < NGC;) {
        if (!need[gc]) { gc++; continue; }
        int e = gc; while (e + 1 < NGC && (need[e + 1] || g.hist[e + 1] == 0)) e++;
        while (!need[e]) e--;
        int64_t lo = g.segOff[gc], hi = g.segOff[e + 1], at = (int64_t)hostKeys.size();
        hostKeys.resize(at + (hi - lo));
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(hostKeys.data() + at, st.keys32 + lo, (hi - lo) * 4, hipMemcpyDeviceToHost, ctx->stream));
        CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // hostKeys may reallocate on the next run
        for (int k = gc; k <= e; k++) hostAt[k] = at + (g.segOff[k] - lo);
        gc = e + 1;
    }
    struct WC { float v, w; };
    std::vector<WC> wc;
    for (size_t b = 0; b < buckets.size(); b++) {
        wc.clear();
        for (auto& p : plan[b]) for (uint32_t k = 0; k < g.hist[p.gc]; k++) wc.push_back({host_float_of_key(hostKeys[hostAt[p.gc] + k]), p.w});
        double acc = 0;
        for (auto& t : wc) acc += (double)t.w;                        // LINQ Sum<float>: double accumulator, float result
        const double totalWeight = (double)(float)acc;
        std::stable_sort(wc.begin(), wc.end(), [](const WC& a, const WC& c) { return a.v < c.v; });   // OrderBy is stable
        double cumulativeWeight = 0;
        for (auto& t : wc) {
            cumulativeWeight += (double)t.w;
            const double cumulativeProb = cumulativeWeight / totalWeight;
            for (int i = 0; i < nprobs; i++) if (cumulativeProb <= (double)probs[i]) out[b].q[i] = (double)t.v;
        }
    }
    return CANVAS_OK;
}

// NormalizeByGC (CanvasClean.cs:163-196) on the grouped autosomal counts
static int32_t normalize_by_gc(CleanState& st, const GcGroups& g, double* dMedians, bool emptyBucketsReadable) {
    canvas_ctx* ctx = st.ctx;
    if (g.nauto == 0) CANVAS_FAIL(ctx, CANVAS_ERR_UNSUPPORTED, "NormalizeByGC: no autosomal bins (the reference would throw on an empty median)");
    hipLaunchKernelGGL(k_gather_keys, dim3(nblk(g.nauto, 256)), dim3(256), 0, ctx->stream, st.cur.count, st.gidx, g.nauto, st.keys32);
    std::vector<SelQuery> qs;
    auto addMedian = [&](int lo, int hi, int64_t cnt) { if (cnt % 2) qs.push_back({lo, hi, cnt / 2}); else { qs.push_back({lo, hi, cnt / 2 - 1}); qs.push_back({lo, hi, cnt / 2}); } };
    addMedian(0, NGC - 1, g.nauto);
    std::vector<int> first(NGC, -1);
    std::vector<int> sparse;
    for (int gc = 0; gc < NGC; gc++) {
        if (g.hist[gc] >= 100) { first[gc] = (int)qs.size(); addMedian(gc, gc, g.hist[gc]); }
        else if (g.hist[gc] > 0 || emptyBucketsReadable) sparse.push_back(gc);     // CanvasClean.cs:178-187
    }
    std::vector<unsigned long long> res;
    int32_t rc = radix_select<uint32_t>(ctx, st.keys32, NGC, g.segOff, qs, res); if (rc) return rc;
    std::vector<WQuant> wq; const float half = 0.5f;
    rc = weighted_quantiles_sparse(st, g, sparse, &half, 1, wq); if (rc) return rc;
    auto med = [&](int at, int64_t cnt) -> double {
        if (cnt % 2) return (double)host_float_of_key((uint32_t)res[at]);
        return (double)median_from_two(host_float_of_key((uint32_t)res[at]), host_float_of_key((uint32_t)res[at + 1]));
    };
    double globalMedian = med(0, g.nauto);
    double medians[NGC];
    for (int gc = 0; gc < NGC; gc++) medians[gc] = first[gc] >= 0 ? med(first[gc], g.hist[gc]) : 0.0;   // unreadable empty buckets stay 0
    for (size_t b = 0; b < sparse.size(); b++) medians[sparse[b]] = wq[b].q[0];
    rc = canvas_h2d_small(ctx, dMedians, medians, sizeof medians); if (rc) return rc;
    hipLaunchKernelGGL(k_apply_gc, dim3(nblk(st.n, 256)), dim3(256), 0, ctx->stream, st.cur.count, st.cur.gc, st.n, dMedians, globalMedian);
    CANVAS_HIP_TRY(ctx, hipGetLastError());
    return CANVAS_OK;
}

// NormalizeVarianceByGC (CanvasClean.cs:34-97); returns whether counts were changed
static int32_t normalize_variance_by_gc(CleanState& st, const GcGroups& g, VarTab* dTab, bool& changed) {
    canvas_ctx* ctx = st.ctx;
    changed = false;
    hipLaunchKernelGGL(k_gather_keys, dim3(nblk(g.nauto, 256)), dim3(256), 0, ctx->stream, st.cur.count, st.gidx, g.nauto, st.keys32);
    std::vector<SelQuery> qs;
    std::vector<int> first(NGC + 1, -1);
    auto addQ = [&](int slot, int lo, int hi, int64_t cnt) { first[slot] = (int)qs.size(); QuartIdx qi = quartile_indices(cnt); for (int k = 0; k < qi.n; k++) qs.push_back({lo, hi, qi.idx[k]}); };
    if (g.nauto < 2) CANVAS_FAIL(ctx, CANVAS_ERR_UNSUPPORTED, "NormalizeVarianceByGC: fewer than 2 autosomal bins");
    addQ(NGC, 0, NGC - 1, g.nauto);
    for (int gc = 0; gc < NGC; gc++) if (g.hist[gc] >= 100) addQ(gc, gc, gc, g.hist[gc]);
    std::vector<unsigned long long> res;
    int32_t rc = radix_select<uint32_t>(ctx, st.keys32, NGC, g.segOff, qs, res); if (rc) return rc;
    std::vector<int> sparse;
    for (int gc = 0; gc < NGC; gc++) if (g.hist[gc] > 0 && g.hist[gc] < 100) sparse.push_back(gc);       // CanvasClean.cs:62-68
    std::vector<WQuant> wq; const float probs[3] = {0.25f, 0.5f, 0.75f};
    rc = weighted_quantiles_sparse(st, g, sparse, probs, 3, wq); if (rc) return rc;
    auto quart = [&](int slot, int64_t cnt, float& q1, float& q2, float& q3) {
        float v[6]; QuartIdx qi = quartile_indices(cnt);
        for (int k = 0; k < qi.n; k++) v[k] = host_float_of_key((uint32_t)res[first[slot] + k]);
        quartiles_from_values(cnt, v, q1, q2, q3);
    };
    float g1, g2, g3;
    quart(NGC, g.nauto, g1, g2, g3);
    VarTab tab;
    for (int gc = 0; gc < NGC; gc++) {
        if (g.hist[gc] == 0) { tab.localIQR[gc] = -1.0f; tab.med[gc] = -1.0f; }
        else if (g.hist[gc] >= 100) { float q1, q2, q3; quart(gc, g.hist[gc], q1, q2, q3); tab.med[gc] = q2; tab.localIQR[gc] = q3 - q1; }
    }
    for (size_t b = 0; b < sparse.size(); b++) { tab.med[sparse[b]] = (float)wq[b].q[1]; tab.localIQR[sparse[b]] = (float)(wq[b].q[2] - wq[b].q[0]); }
    tab.globalIQR = g3 - g1;
    int significant = 0;
    for (int i = 10; i < 90; i++) if (tab.globalIQR * 2.0f < tab.localIQR[i]) significant++;
    if (significant <= 0) return CANVAS_OK;
    rc = canvas_h2d_small(ctx, dTab, &tab, sizeof tab); if (rc) return rc;
    hipLaunchKernelGGL(k_apply_var, dim3(nblk(st.n, 256)), dim3(256), 0, ctx->stream, st.cur.count, st.cur.gc, st.n, dTab);
    changed = true;
    return CANVAS_OK;
}

// GetLocalStandardDeviation + GetLocalStandardDeviationAverage (CanvasClean.cs:243-298); Q8 kept
// GetLocalStandardDeviationAverage (CanvasClean.cs:243-298).  The value is only needed at the very end (RemoveBinsWithExtremeLocalSD and
// the metric file), so the per-chromosome MAD runs on the context's side stream while the GC stages continue on the main one:
// local_sd_begin enqueues it, local_sd_end waits for the result.
struct LocalSdPending { unsigned int nb = 0; std::vector<long long> brec; };
// part 1: window SDs and chromosome-run records; the bin count may still be on the device (dN).  No synchronisation: the records arrive
// with the caller's next one.
static int32_t local_sd_launch(CleanState& st, const unsigned long long* dN, double* dSd, unsigned int* dCnt, long long* dPos, LocalSdPending& P) {
    canvas_ctx* ctx = st.ctx;
    const int64_t Du = st.n - 1, nWu = Du >= 1 ? (Du - 1) / 20 : 0;        // upper bounds
    if (nWu > 0) hipLaunchKernelGGL(k_local_sd, dim3(nblk(nWu, 256)), dim3(256), 0, ctx->stream, st.cur.count, nWu, dSd, st.cur.dev, dN);
    CANVAS_HIP_TRY(ctx, hipMemsetAsync(dCnt, 0, 4, ctx->stream));
    hipLaunchKernelGGL(k_run_bounds, dim3(nblk(st.n, 256)), dim3(256), 0, ctx->stream, st.cur.chr, st.n, dCnt, dPos, 65536, dN);
    P.brec.assign(1024, 0);
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(&P.nb, dCnt, 4, hipMemcpyDeviceToHost, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(P.brec.data(), dPos, 1024 * 8, hipMemcpyDeviceToHost, ctx->stream));     // the usual case: all records in one go
    return CANVAS_OK;
}
// part 2 (after a synchronisation, st.n exact): runs of windows per chromosome, then the per-run MAD on the side stream
static int32_t local_sd_begin(CleanState& st, LocalSdPending& P, double* dSd, double* dRunMedian, int64_t* dRunStart, long long* dPos, int& nrunsOut) {
    canvas_ctx* ctx = st.ctx;
    const int64_t D = st.n - 1;
    const int64_t nW = D >= 1 ? (D - 1) / 20 : 0;    // windows with windowEnd = 20(w+1) < D
    if (nW <= 0) CANVAS_FAIL(ctx, CANVAS_ERR_UNSUPPORTED, "local SD: no complete window");
    const int cap = 65536;
    const unsigned int nb = P.nb; std::vector<long long>& brec = P.brec;
    if ((int)nb > cap) CANVAS_FAIL(ctx, CANVAS_ERR_UNSUPPORTED, "local SD: more than 65536 chromosome runs");
    if (nb > 1024) { brec.resize(nb); CANVAS_HIP_TRY(ctx, hipMemcpy(brec.data(), dPos, (size_t)nb * 8, hipMemcpyDeviceToHost)); }
    brec.resize(nb);
    std::sort(brec.begin(), brec.end());
    std::vector<long long> bpos(nb); std::vector<int32_t> bchr(nb);
    for (unsigned i = 0; i < nb; i++) { bpos[i] = brec[i] >> 20; bchr[i] = (int32_t)(brec[i] & 0xFFFFF); }
    bpos.push_back(st.n);
    std::vector<int64_t> runStart; std::vector<int32_t> runChr;
    for (unsigned r = 0; r < nb; r++) {
        int64_t w0 = (bpos[r] + 19) / 20, w1 = std::min<int64_t>((bpos[r + 1] + 19) / 20, nW);
        if (w0 >= w1) continue;
        if (!runChr.empty() && runChr.back() == bchr[r]) continue;   // adjacent windows with the same chromosome merge
        runStart.push_back(w0); runChr.push_back(bchr[r]);
    }
    const int nruns = (int)runStart.size();
    std::vector<int64_t> segOff(runStart); segOff.push_back(nW);
    segOff[0] = 0;
    // Mad of the window SDs per run (median, then median of |x - median|), one workgroup per run, on the side stream
    nrunsOut = nruns;
    if (nruns > 0) {
        int32_t rc = canvas_side_init(ctx); if (rc) return rc;
        int64_t* pinStarts = reinterpret_cast<int64_t*>(ctx->side_pin + 65536);      // k_local_sd finished at the synchronisation above
        memcpy(pinStarts, segOff.data(), (size_t)(nruns + 1) * 8);
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dRunStart, pinStarts, (nruns + 1) * 8, hipMemcpyHostToDevice, ctx->side));
        hipLaunchKernelGGL(k_run_mad, dim3(nruns), dim3(1024), 0, ctx->side, dSd, dRunStart, dRunMedian);
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(ctx->side_pin, dRunMedian, nruns * 8, hipMemcpyDeviceToHost, ctx->side));
    }
    return CANVAS_OK;
}
static int32_t local_sd_end(CleanState& st, int nruns, double& localSd) {
    canvas_ctx* ctx = st.ctx;
    if (nruns > 0) { CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->side)); CANVAS_HIP_TRY(ctx, hipGetLastError()); }
    double s = 0;
    for (int r = 0; r < nruns; r++) s += ctx->side_pin[r];        // List<double>.Average(): sequential sum / count
    localSd = s / (double)nruns;
    return CANVAS_OK;
}


// NormalizeByGC, LOESS flavour (CanvasClean.cs:145-152 -> LoessGCNormalizer): see loess.hpp
struct LoessModelDev { loess::Grouped G; std::vector<double> P; double medianY = 0; };
static int32_t loess_build_model(CleanState& st, const uint8_t* dIsY, int excludeY, uint8_t* dKey, double* dY, double* dYg, double* dP, uint32_t* dTileHist, uint32_t* dKeyTot,
                                 uint32_t* dGroupOff, double* dBlockSum, unsigned long long* dKeys64, unsigned int* dCnt, LoessModelDev& M) {
    canvas_ctx* ctx = st.ctx;
    const int64_t n = st.n;
    const int ntiles = (int)nblk(n, LO_TILE);
    hipLaunchKernelGGL(k_loess_keys, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, st.cur.count, st.cur.gc, st.cur.chr, dIsY, excludeY, n, dKey, dY);
    hipLaunchKernelGGL(k_loess_tile_hist, dim3(ntiles), dim3(256), 0, ctx->stream, dKey, n, dTileHist);
    hipLaunchKernelGGL(k_loess_col_scan, dim3(128), dim3(1024), 0, ctx->stream, dTileHist, ntiles, dKeyTot);
    uint32_t tot[128];
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(tot, dKeyTot, sizeof tot, hipMemcpyDeviceToHost, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    uint32_t goff[128]; int64_t acc = 0;
    for (int g = 0; g < LO_NGC; g++) { M.G.off[g] = acc; goff[g] = (uint32_t)acc; acc += tot[g]; }
    M.G.off[LO_NGC] = acc; M.G.n = acc;
    for (int g = LO_NGC; g < 128; g++) goff[g] = (uint32_t)acc;
    for (int g = 0; g < LO_NGC; g++) M.G.shift[g] = 0;
    if (M.G.n < 2) CANVAS_FAIL(ctx, CANVAS_ERR_UNSUPPORTED, "LOESS: fewer than 2 usable bins");
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dGroupOff, goff, sizeof goff, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_loess_scatter, dim3(ntiles), dim3(64), 0, ctx->stream, dKey, dY, n, dTileHist, dGroupOff, dYg);
    const int64_t m = M.G.n; const int nb = (int)nblk(m, LO_TILE);
    hipLaunchKernelGGL(k_dscan_block, dim3(nb), dim3(256), 0, ctx->stream, dYg, m, dBlockSum);
    hipLaunchKernelGGL(k_dscan_top, dim3(1), dim3(64), 0, ctx->stream, dBlockSum, nb);
    hipLaunchKernelGGL(k_dscan_write, dim3(nb), dim3(64), 0, ctx->stream, dYg, m, dBlockSum, dP);
    M.P.resize((size_t)m + 1);
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(M.P.data(), dP, (size_t)(m + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    // medianY = Utilities.Median(counts): exact order statistic of the model's y values (excluded elements carry +inf keys)
    CANVAS_HIP_TRY(ctx, hipMemsetAsync(dCnt, 0, 4, ctx->stream));
    hipLaunchKernelGGL(k_loess_ykeys, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, dKey, dY, n, dKeys64, dCnt);
    std::vector<SelQuery> qs;
    if (m % 2) qs.push_back({0, 0, m / 2}); else { qs.push_back({0, 0, m / 2 - 1}); qs.push_back({0, 0, m / 2}); }
    std::vector<unsigned long long> res;
    int32_t rc = radix_select<unsigned long long>(ctx, dKeys64, 1, std::vector<int64_t>{0, n}, qs, res); if (rc) return rc;
    M.medianY = (m % 2) ? host_double_of_key(res[0]) : (host_double_of_key(res[0]) + host_double_of_key(res[1])) / 2;
    M.G.P = M.P.data();
    return CANVAS_OK;
}

static int32_t normalize_by_gc_loess(CleanState& st, int nchr, const uint8_t* h_is_y) {
    canvas_ctx* ctx = st.ctx;
    const int64_t n = st.n;
    const int ntiles = (int)nblk(n, LO_TILE);
    size_t bytes = (size_t)n * (1 + 8 + 8 + 8 + 8) + (size_t)ntiles * 128 * 4 + (size_t)(ntiles + 8) * 8 + 8192 + nchr + 1024 * 8;
    char* arena = nullptr;
    CANVAS_HIP_TRY(ctx, hipMalloc((void**)&arena, bytes));
    struct Free { canvas_ctx* c; char* p; ~Free() { (void)hipStreamSynchronize(c->stream); (void)hipFree(p); } } fr{ctx, arena};
    WsCarver ws(arena);
    uint8_t* dKey = ws.take<uint8_t>(n); double* dY = ws.take<double>(n); double* dYg = ws.take<double>(n); double* dP = ws.take<double>(n + 1);
    unsigned long long* dKeys64 = ws.take<unsigned long long>(n); uint32_t* dTileHist = ws.take<uint32_t>((size_t)ntiles * 128); uint32_t* dKeyTot = ws.take<uint32_t>(128);
    uint32_t* dGroupOff = ws.take<uint32_t>(128); double* dBlockSum = ws.take<double>(ntiles + 8); unsigned int* dCnt = ws.take<unsigned int>(1);
    uint8_t* dIsY = ws.take<uint8_t>(nchr); double* dFit = ws.take<double>(256);
    std::vector<uint8_t> isY(nchr, 0);
    if (h_is_y) for (int c = 0; c < nchr; c++) isY[c] = h_is_y[c];
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dIsY, isY.data(), nchr, hipMemcpyHostToDevice, ctx->stream));
    // bandwidth search without chrY (LoessGCNormalizer.cs:63-68)
    LoessModelDev noY;
    int32_t rc = loess_build_model(st, dIsY, 1, dKey, dY, dYg, dP, dTileHist, dKeyTot, dGroupOff, dBlockSum, dKeys64, dCnt, noY); if (rc) return rc;
    double minBw = std::max(2.0 / (double)noY.G.n, 0.3), maxBw = std::min(1.0, 0.75);
    if (maxBw < minBw) maxBw = minBw;
    if (noY.G.minX() == 0) CANVAS_FAIL(ctx, CANVAS_ERR_UNSUPPORTED, "LOESS: a bin with GC = 0 makes the reference index past fittedByGC (LoessGCNormalizer.cs:74,113)");
    const double best = loess::golden_section([&](double b) { return loess::objective(b, noY.G, noY.medianY); }, minBw, maxBw);
    // final fit on all bins (LoessGCNormalizer.cs:70-75)
    LoessModelDev all;
    rc = loess_build_model(st, dIsY, 0, dKey, dY, dYg, dP, dTileHist, dKeyTot, dGroupOff, dBlockSum, dKeys64, dCnt, all); if (rc) return rc;
    const int minGC = all.G.minX(), maxGC = all.G.maxX();
    std::vector<double> fitted = loess::train_predict(all.G, best, minGC, maxGC);
    if (fitted.empty() || fitted.size() > 256) CANVAS_FAIL(ctx, CANVAS_ERR_UNSUPPORTED, "LOESS: degenerate GC range");
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dFit, fitted.data(), fitted.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_loess_apply, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, st.cur.count, st.cur.gc, n, dFit, (int)fitted.size(), minGC, all.medianY);
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    CANVAS_HIP_TRY(ctx, hipGetLastError());
    return CANVAS_OK;
}

#include "quantize.hpp"
#include "clean_fast.hpp"
#include "clean_gc_only.hpp"

extern "C" int32_t canvas_clean2(canvas_ctx* ctx, int64_t n, int32_t* d_chr, int32_t* d_start, int32_t* d_stop, float* d_count,
                                 int32_t* d_gc, int32_t nchr, const uint8_t* h_chr_is_autosome, const uint8_t* h_chr_is_y, uint32_t flags, int32_t min_bins_per_gc,
                                 double* h_local_sd_out, int64_t* h_n_out, int32_t* h_info) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (n < 0 || nchr <= 0 || nchr > (1 << 20) || !h_chr_is_autosome || !h_n_out) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_clean: bad arguments");
    const bool loessMode = (flags & CANVAS_CLEAN_LOESS) != 0;
    if (n >= 0x7FFFFFFFll) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "too many bins");
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    int32_t info[8] = {0};
    double localSd = -1.0;
    if (n == 0) { *h_n_out = 0; if (h_local_sd_out) *h_local_sd_out = -1.0; if (h_info) memcpy(h_info, info, sizeof info); return CANVAS_OK; }
    // MedianByGC with the default weighted-median setting: the whole stage is driven from the device (clean_fast.hpp), one synchronisation at the end.  The host-driven
    // path below stays for -m LOESS, -w < 100 (sparse GC buckets take the neighbour-weighted quantiles) and inputs with more than CF_MAXRUN chromosome runs;
    // CANVAS_CLEAN_HOST_DRIVEN=1 forces it (test hook: the two paths must agree bit for bit)
    if (!loessMode && min_bins_per_gc >= 100 && !cvx_hook("CANVAS_CLEAN_HOST_DRIVEN")) {
        bool handled = false;
        // -g alone on whole-number counts (BASELINE configs[1]): three launches, in place (clean_gc_only.hpp); anything it cannot decide exactly leaves the arrays untouched.
        // CANVAS_CLEAN_GENERAL_GC=1 (test hook) keeps the general chain
        if (flags == CANVAS_CLEAN_GCNORM && !cvx_hook("CANVAS_CLEAN_GENERAL_GC")) {
            char h1 = 0;
            int32_t rcg = clean_gc_only(ctx, 1, &n, &d_chr, &d_start, &d_stop, &d_count, &d_gc, nchr, h_chr_is_autosome, min_bins_per_gc, h_n_out, h_info, &h1);
            if (rcg) return rcg;
            if (h1) { if (h_local_sd_out) *h_local_sd_out = -1.0; return CANVAS_OK; }
        }
        int32_t rcf = clean_device_driven(ctx, n, d_chr, d_start, d_stop, d_count, d_gc, nchr, h_chr_is_autosome, flags, min_bins_per_gc, h_local_sd_out, h_n_out, h_info, &handled);
        if (rcf) return rcf;
        if (handled) return CANVAS_OK;
    }
    // workspace
    const int64_t nW0 = n / 20 + 2;
    WsSizer sz;
    sz.take<int32_t>(n); sz.take<int32_t>(n); sz.take<int32_t>(n); sz.take<int32_t>(n); sz.take<float>(n); sz.take<double>(n); sz.take<double>(n);
    sz.take<int32_t>(n); sz.take<int32_t>(n); sz.take<int32_t>(n); sz.take<int32_t>(n); sz.take<float>(n); sz.take<double>(n);
    sz.take<uint8_t>(n); sz.take<uint32_t>(n / CBLK + 2); sz.take<unsigned long long>(2); sz.take<uint32_t>(n); sz.take<unsigned long long>(nW0); sz.take<uint32_t>(n);
    sz.take<uint8_t>(nchr); sz.take<uint32_t>(2 * NGC); sz.take<uint32_t>(NGC + 1); sz.take<uint32_t>(NGC); sz.take<double>(NGC); sz.take<VarTab>(1);
    sz.take<uint8_t>(NGC); sz.take<double>(nW0); sz.take<double>(65536); sz.take<int64_t>(65536 + 1); sz.take<unsigned int>(1); sz.take<long long>(65536); sz.take<unsigned int>(1);
    int32_t rc = canvas_ws_reserve(ctx, sz.off + 8192); if (rc) return rc;
    WsCarver ws(ctx->ws);
    CleanState st; st.ctx = ctx; st.n = n;
    st.cur = Soa{d_chr, d_start, d_stop, d_gc, d_count, nullptr};
    st.alt.chr = ws.take<int32_t>(n); st.alt.start = ws.take<int32_t>(n); st.alt.stop = ws.take<int32_t>(n); st.alt.gc = ws.take<int32_t>(n);
    st.alt.count = ws.take<float>(n); st.alt.dev = ws.take<double>(n); st.cur.dev = ws.take<double>(n);
    st.bufs[0] = st.cur; st.bufs[1] = st.alt;
    st.bufs[2].chr = ws.take<int32_t>(n); st.bufs[2].start = ws.take<int32_t>(n); st.bufs[2].stop = ws.take<int32_t>(n); st.bufs[2].gc = ws.take<int32_t>(n);
    st.bufs[2].count = ws.take<float>(n); st.bufs[2].dev = ws.take<double>(n);
    st.flags = ws.take<uint8_t>(n); st.blockCnt = ws.take<uint32_t>(n / CBLK + 2); st.dTotal = ws.take<unsigned long long>(2);
    st.keys32 = ws.take<uint32_t>(n); st.keys64 = ws.take<unsigned long long>(nW0); st.gidx = ws.take<uint32_t>(n);
    st.dIsAuto = ws.take<uint8_t>(nchr);
    uint32_t* dHist = ws.take<uint32_t>(2 * NGC); uint32_t* dSegOff = ws.take<uint32_t>(NGC + 1); uint32_t* dCursor = ws.take<uint32_t>(NGC);
    double* dMedians = ws.take<double>(NGC); VarTab* dTab = ws.take<VarTab>(1); uint8_t* dKeepGc = ws.take<uint8_t>(NGC);
    double* dSd = ws.take<double>(nW0); double* dRunMedian = ws.take<double>(65536); int64_t* dRunStart = ws.take<int64_t>(65536 + 1);
    unsigned int* dCnt = ws.take<unsigned int>(1); long long* dPos = ws.take<long long>(65536);
    unsigned int* dBad = ws.take<unsigned int>(1);
    ProfScope psTotal(ctx, "clean_total");       // whole CanvasClean on the device timeline (kernels + the gaps of the host decisions)
    rc = canvas_h2d_small(ctx, st.dIsAuto, h_chr_is_autosome, nchr); if (rc) return rc;
    CANVAS_HIP_TRY(ctx, hipMemsetAsync(dBad, 0, 4, ctx->stream));
    hipLaunchKernelGGL(k_init_validate, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, st.cur.chr, st.cur.gc, n, nchr, st.cur.dev, dBad);   // CountDeviation = -1 (GenomicBin.cs:83)
    {   // the range check comes back before any kernel indexes a table with gc / chr (one 4-byte read; the stage has several synchronisations anyway)
        unsigned int bad = 0;
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(&bad, dBad, 4, hipMemcpyDeviceToHost, ctx->stream));
        CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (bad) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_clean: a bin has gc outside 0..100 or a chromosome index outside [0, nchr) (the reference throws IndexOutOfRangeException)");
    }

    // RemoveBigBins (CanvasClean.cs:328-355) and RemoveOutliers (CanvasClean.cs:387-413) without a host round trip in between: the size
    // threshold is read by the flag kernel from the select's device result, the count after the first compaction stays on the device and the
    // outlier kernels run on an upper-bound grid; both counts come back with ONE synchronisation.
    const unsigned long long* dN = nullptr; bool sized = false;
    unsigned long long* dTot2 = st.dTotal;           // two slots
    if (flags & CANVAS_CLEAN_FILTSIZE) {
        int64_t index = (int64_t)(0.98 * (double)st.n);
        if (index < st.n) {
            hipLaunchKernelGGL(k_keys_size, dim3(nblk(st.n, 256)), dim3(256), 0, ctx->stream, st.cur.start, st.cur.stop, st.n, st.keys32);
            std::vector<unsigned long long> res; const unsigned long long* dKey = nullptr;
            rc = radix_select<uint32_t>(ctx, st.keys32, 1, std::vector<int64_t>{0, st.n}, std::vector<SelQuery>{{0, 0, index}}, res, &dKey); if (rc) return rc;
            hipLaunchKernelGGL(k_flags_size, dim3(nblk(st.n, 256)), dim3(256), 0, ctx->stream, st.cur.start, st.cur.stop, st.n, dKey, st.flags);
            compact_launch(st, nullptr, dTot2 + 0, false);
            dN = dTot2 + 0; sized = true;
        }
    }
    bool outl = false;
    if ((flags & CANVAS_CLEAN_OUTLIERS) && st.n > 0) {
        hipLaunchKernelGGL(k_flags_outlier, dim3(nblk(st.n, 256)), dim3(256), 0, ctx->stream, st.cur.chr, st.cur.count, st.n, dN, st.flags);
        compact_launch(st, dN, dTot2 + 1, false);
        dN = dTot2 + 1; outl = true;
    }
    info[0] = (int32_t)st.n;
    // the local-SD kernels go into the same stretch (on the upper-bound grid); whether the metric applies (>= 50000 bins, CanvasClean.cs:483-486)
    // is decided once the count is back
    LocalSdPending lsdPending; bool lsdLaunched = false;
    if ((flags & CANVAS_CLEAN_LOCALSD) && st.n >= 50000) { rc = local_sd_launch(st, dN, dSd, dCnt, dPos, lsdPending); if (rc) return rc; lsdLaunched = true; }
    if (sized || outl || lsdLaunched) {
        unsigned long long tot[2] = {0, 0};
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(tot, dTot2, 16, hipMemcpyDeviceToHost, ctx->stream));
        CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        CANVAS_HIP_TRY(ctx, hipGetLastError());
        if (sized) info[0] = (int32_t)tot[0];
        if (sized || outl) st.n = (int64_t)(outl ? tot[1] : tot[0]);
    }
    info[1] = (int32_t)st.n;
    bool haveLocalSd = (flags & CANVAS_CLEAN_LOCALSD) && st.n >= 50000;   // CanvasClean.cs:483-486
    int sdRuns = 0;
    if (haveLocalSd) { rc = local_sd_begin(st, lsdPending, dSd, dRunMedian, dRunStart, dPos, sdRuns); if (rc) return rc; }
    if ((flags & CANVAS_CLEAN_GCNORM) && st.n > 0 && loessMode) {
        // -m LOESS: no GC strip (CanvasClean.cs:497-499); variance normalisation still uses the MedianByGC quartiles
        rc = normalize_by_gc_loess(st, nchr, h_chr_is_y); if (rc) return rc;
        if (haveLocalSd && st.n > 500000) {
            GcGroups g;
            rc = gc_histogram(st, dHist, g.hist); if (rc) return rc;
            rc = group_by_gc(st, g, dSegOff, dCursor); if (rc) return rc;
            bool changed = false;
            rc = normalize_variance_by_gc(st, g, dTab, changed); if (rc) return rc;
            info[4] = changed ? 1 : 0;
            if (changed) { rc = normalize_by_gc_loess(st, nchr, h_chr_is_y); if (rc) return rc; }
        }
    } else if ((flags & CANVAS_CLEAN_GCNORM) && st.n > 0) {
        // RemoveBinsWithExtremeGC (CanvasClean.cs:207-237)
        GcGroups g; uint32_t histOther[NGC];
        rc = gc_histogram(st, dHist, g.hist, histOther); if (rc) return rc;
        double totalCount = 0;
        for (int i = 0; i < NGC; i++) totalCount += g.hist[i];
        int averageCountPerGC = std::max(min_bins_per_gc, (int)(totalCount / NGC));
        int threshold = std::min(100, averageCountPerGC);
        uint8_t keep[NGC]; int64_t kept = 0; bool dropsAny = false;
        // survivors = bins of ANY chromosome whose GC value keeps enough autosomal bins; known from the two histograms, so "strippedBins.Count == 0
        // -> proceed without GC correction" (CanvasClean.cs:500-505) needs no device count
        for (int i = 0; i < NGC; i++) { keep[i] = (int)g.hist[i] >= threshold; if (!keep[i]) dropsAny = true; else kept += (int64_t)g.hist[i] + (int64_t)histOther[i]; }
        int nb = (int)nblk(st.n, CBLK);
        if (kept > 0 && dropsAny && kept < st.n) {
            rc = canvas_h2d_small(ctx, dKeepGc, keep, NGC); if (rc) return rc;
            hipLaunchKernelGGL(k_flags_gc, dim3(nblk(st.n, 256)), dim3(256), 0, ctx->stream, st.cur.gc, st.n, dKeepGc, st.flags);
            hipLaunchKernelGGL(k_block_count, dim3(nb), dim3(256), 0, ctx->stream, st.flags, st.n, st.blockCnt);
            hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, ctx->stream, st.blockCnt, nb, st.dTotal);
        }
        if (kept > 0) {
            if (dropsAny && kept < st.n) {
                pick_alt(st, true);
                hipLaunchKernelGGL(k_scatter, dim3(nb), dim3(256), 0, ctx->stream, st.flags, st.blockCnt, st.n, st.cur, st.alt);
                std::swap(st.cur, st.alt); st.n = kept;
            }
            for (int i = 0; i < NGC; i++) {
                if (!keep[i]) g.hist[i] = 0;
            }
            const bool emptyReadable = threshold <= 0;   // otherwise every bin of an empty autosomal bucket was stripped above
            rc = group_by_gc(st, g, dSegOff, dCursor); if (rc) return rc;
            rc = normalize_by_gc(st, g, dMedians, emptyReadable); if (rc) return rc;
            if (haveLocalSd && st.n > 500000) {     // CanvasClean.cs:512-519
                bool changed = false;
                rc = normalize_variance_by_gc(st, g, dTab, changed); if (rc) return rc;
                info[4] = changed ? 1 : 0;
                if (changed) { rc = normalize_by_gc(st, g, dMedians, emptyReadable); if (rc) return rc; }
            }
        }
    }
    info[2] = (int32_t)st.n;
    if (haveLocalSd) { rc = local_sd_end(st, sdRuns, localSd); if (rc) return rc; }
    if (haveLocalSd && localSd > 5.0 && st.n > 0) {    // RemoveBinsWithExtremeLocalSD (threshold 20 -> 40.0)
        hipLaunchKernelGGL(k_flags_localsd, dim3(nblk(st.n, 256)), dim3(256), 0, ctx->stream, st.cur.dev, st.n, 20 * 2.0, st.flags);
        rc = compact(st, true); if (rc) return rc;
    }
    info[3] = (int32_t)st.n;
    // results must end in the caller's arrays
    if (st.cur.chr != d_chr && st.n > 0) {
        Soa dst; dst.chr = d_chr; dst.start = d_start; dst.stop = d_stop; dst.gc = d_gc; dst.count = d_count; dst.dev = nullptr;
        hipLaunchKernelGGL(k_copy_soa, dim3(nblk(st.n, 256)), dim3(256), 0, ctx->stream, st.cur, dst, st.n);
    }
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    CANVAS_HIP_TRY(ctx, hipGetLastError());
    *h_n_out = st.n;
    if (h_local_sd_out) *h_local_sd_out = haveLocalSd ? localSd : -1.0;
    if (h_info) memcpy(h_info, info, sizeof info);
    return CANVAS_OK;
}

// ---------------------------------------------------------------- pedigree: bins present in every sample's cleaned list
// Utilities.MergeMultiSampleCleanedBedFile (CanvasCommon/Utilities.cs:834-920) + CanvasRunner.NormalizeCanvasClean (CanvasRunner.cs:883-903):
// bins are keyed by (chromosome, start); a bin survives when every sample has it ("if outlier is removed in one sample, remove it in
// all samples"), its stop is the one read last (the last sample's), its counts are the samples' counts in input order, and the output
// follows the first sample's order.  Every sample's list is sorted by (chromosome index, start) — CanvasBin / CanvasClean order — so the
// lookup is a binary search; unsorted input is rejected.
struct MergePtrs { const int32_t* chr[16]; const int32_t* start[16]; const int32_t* stop[16]; const float* count[16]; float* outCount[16]; long long n[16]; };
__device__ __forceinline__ unsigned long long merge_key(const int32_t* chr, const int32_t* start, long long i) { return ((unsigned long long)(uint32_t)chr[i] << 32) | (uint32_t)start[i]; }
__global__ void __launch_bounds__(256) k_merge_check_sorted(MergePtrs P, int nsamples, int* __restrict__ bad) {
    const int s = blockIdx.y;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (s >= nsamples || i + 1 >= P.n[s]) return;
    if (P.chr[s][i] < 0 || P.start[s][i] < 0 || merge_key(P.chr[s], P.start[s], i) >= merge_key(P.chr[s], P.start[s], i + 1)) *bad = 1;
}
__global__ void __launch_bounds__(256) k_merge_find(MergePtrs P, int nsamples, uint8_t* __restrict__ flags, int32_t* __restrict__ where /* [nsamples][n0] */) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n0 = P.n[0];
    if (i >= n0) return;
    const unsigned long long key = merge_key(P.chr[0], P.start[0], i);
    bool all = true;
    for (int s = 1; s < nsamples; s++) {
        long long lo = 0, hi = P.n[s];
        while (lo < hi) { long long mid = (lo + hi) >> 1; if (merge_key(P.chr[s], P.start[s], mid) < key) lo = mid + 1; else hi = mid; }
        const bool hit = lo < P.n[s] && merge_key(P.chr[s], P.start[s], lo) == key;
        where[(size_t)s * n0 + i] = hit ? (int32_t)lo : -1;
        all = all && hit;
    }
    flags[i] = all ? 1 : 0;
}
__global__ void __launch_bounds__(256) k_merge_scatter(MergePtrs P, int nsamples, const uint8_t* __restrict__ flags, const uint32_t* __restrict__ blockOff, const int32_t* __restrict__ where,
                                                       int32_t* __restrict__ oChr, int32_t* __restrict__ oStart, int32_t* __restrict__ oStop) {
    __shared__ uint32_t sh[4];
    const long long n0 = P.n[0];
    const long long base = (long long)blockIdx.x * CBLK;
    uint32_t running = blockOff[blockIdx.x];
    for (int j = 0; j < CBLK / 256; j++) {
        const long long i = base + j * 256 + threadIdx.x;
        const uint32_t f = (i < n0) ? flags[i] : 0;
        const uint32_t inc = wave_inclusive_scan_u32(f);
        if (lane_id() == 63) sh[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t woff = 0, tot = 0;
        for (int k = 0; k < 4; k++) { if (k < (int)(threadIdx.x >> 6)) woff += sh[k]; tot += sh[k]; }
        if (f) {
            const uint32_t d = running + woff + inc - 1;
            oChr[d] = P.chr[0][i]; oStart[d] = P.start[0][i];
            const long long last = nsamples > 1 ? (long long)where[(size_t)(nsamples - 1) * n0 + i] : i;
            oStop[d] = P.stop[nsamples - 1][last];
            P.outCount[0][d] = P.count[0][i];
            for (int s = 1; s < nsamples; s++) P.outCount[s][d] = P.count[s][where[(size_t)s * n0 + i]];
        }
        running += tot;
        __syncthreads();
    }
}
extern "C" int32_t canvas_merge_cleaned(canvas_ctx* ctx, int32_t nsamples, const int64_t* h_n, const int32_t* const* h_d_chr, const int32_t* const* h_d_start,
                                        const int32_t* const* h_d_stop, const float* const* h_d_count, int32_t* d_out_chr, int32_t* d_out_start, int32_t* d_out_stop,
                                        float* const* h_d_out_count, int64_t* h_n_out) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (nsamples <= 0 || nsamples > 16 || !h_n || !h_d_chr || !h_d_start || !h_d_stop || !h_d_count || !d_out_chr || !d_out_start || !d_out_stop || !h_d_out_count || !h_n_out)
        CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_merge_cleaned: bad arguments (1..16 samples)");
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    MergePtrs P; memset(&P, 0, sizeof P);
    int64_t nmax = 0;
    for (int s = 0; s < nsamples; s++) {
        if (h_n[s] < 0 || h_n[s] > 0x7FFFFFFFll) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_merge_cleaned: bin count out of range");
        P.chr[s] = h_d_chr[s]; P.start[s] = h_d_start[s]; P.stop[s] = h_d_stop[s]; P.count[s] = h_d_count[s]; P.outCount[s] = h_d_out_count[s]; P.n[s] = h_n[s];
        nmax = std::max(nmax, h_n[s]);
    }
    const int64_t n0 = h_n[0];
    *h_n_out = 0;
    if (n0 == 0) return CANVAS_OK;
    const int nb = (int)nblk(n0, CBLK);
    WsSizer sz; sz.take<uint8_t>(n0); sz.take<int32_t>((size_t)nsamples * n0); sz.take<uint32_t>(nb + 2); sz.take<unsigned long long>(1); sz.take<int>(1);
    int32_t rc = canvas_ws_reserve(ctx, sz.off + 4096); if (rc) return rc;
    WsCarver ws(ctx->ws);
    uint8_t* flags = ws.take<uint8_t>(n0); int32_t* where = ws.take<int32_t>((size_t)nsamples * n0); uint32_t* blockCnt = ws.take<uint32_t>(nb + 2);
    unsigned long long* dTotal = ws.take<unsigned long long>(1); int* dBad = ws.take<int>(1);
    CANVAS_HIP_TRY(ctx, hipMemsetAsync(dBad, 0, 4, ctx->stream));
    hipLaunchKernelGGL(k_merge_check_sorted, dim3(nblk(nmax, 256), nsamples), dim3(256), 0, ctx->stream, P, nsamples, dBad);
    hipLaunchKernelGGL(k_merge_find, dim3(nblk(n0, 256)), dim3(256), 0, ctx->stream, P, nsamples, flags, where);
    hipLaunchKernelGGL(k_block_count, dim3(nb), dim3(256), 0, ctx->stream, flags, n0, blockCnt);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, ctx->stream, blockCnt, nb, dTotal);
    hipLaunchKernelGGL(k_merge_scatter, dim3(nb), dim3(256), 0, ctx->stream, P, nsamples, flags, blockCnt, where, d_out_chr, d_out_start, d_out_stop);
    unsigned long long tot = 0; int bad = 0;
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(&tot, dTotal, 8, hipMemcpyDeviceToHost, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(&bad, dBad, 4, hipMemcpyDeviceToHost, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    CANVAS_HIP_TRY(ctx, hipGetLastError());
    if (bad) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_merge_cleaned: every sample must be sorted by (chromosome index, start) without duplicates");
    *h_n_out = (int64_t)tot;
    return CANVAS_OK;
}

__global__ void __launch_bounds__(256) k_quantize_f2(const float* __restrict__ count, int64_t n, double* __restrict__ cov, int generalOnly) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) cov[i] = quantize_f2_one(count[i], nullptr, generalOnly != 0);
}
extern "C" int32_t canvas_quantize_f2(canvas_ctx* ctx, const float* d_count, int64_t n, double* d_cov) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (n < 0 || (n > 0 && (!d_count || !d_cov))) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_quantize_f2: bad arguments");
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (n > 0) hipLaunchKernelGGL(k_quantize_f2, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, d_count, n, d_cov, cvx_hook("CANVAS_F2_GENERAL") ? 1 : 0);      // (test hook: the general digit arithmetic for every value)
    CANVAS_HIP_TRY(ctx, hipGetLastError());
    return CANVAS_OK;
}

// ---------------------------------------------------------------- chromosome offsets of a grouped bin list
__global__ void __launch_bounds__(256) k_chr_first(const int32_t* __restrict__ chr, int64_t n, int nchr, long long* __restrict__ first, const unsigned long long* __restrict__ nDev = nullptr) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (nDev) n = (int64_t)*nDev < n ? (int64_t)*nDev : n;       // (enqueued behind the stage that produces the bins: the grid covers an upper bound)
    if (i >= n) return;
    int c = chr[i];
    if (c >= 0 && c < nchr && (i == 0 || chr[i - 1] != c)) atomicMin(&first[c], (long long)i);
}
static int32_t offsets_from_first(canvas_ctx* ctx, const long long* f, int64_t n, int32_t nchr, int64_t* h_chr_offset) {
    h_chr_offset[nchr] = n;
    for (int c = nchr - 1; c >= 0; c--) h_chr_offset[c] = (f[c] >= 0 && f[c] < n) ? (int64_t)f[c] : h_chr_offset[c + 1];
    for (int c = 0; c < nchr; c++) if (h_chr_offset[c] > h_chr_offset[c + 1]) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "bins are not grouped by increasing chromosome index");
    return CANVAS_OK;
}
extern "C" int32_t canvas_chromosome_offsets(canvas_ctx* ctx, const int32_t* d_chr, int64_t n, int32_t nchr, int64_t* h_chr_offset) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (n < 0 || nchr <= 0 || !h_chr_offset || (n > 0 && !d_chr)) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_chromosome_offsets: bad arguments");
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    int32_t rc = canvas_ws_reserve(ctx, (size_t)nchr * 8 + 256); if (rc) return rc;
    rc = canvas_pin_reserve(ctx, (size_t)nchr * 8); if (rc) return rc;
    long long* dFirst = (long long*)ctx->ws;
    CANVAS_HIP_TRY(ctx, hipMemsetAsync(dFirst, 0x7F, (size_t)nchr * 8, ctx->stream));     // "not seen"
    if (n > 0) hipLaunchKernelGGL(k_chr_first, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, d_chr, n, nchr, dFirst);
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(ctx->pin, dFirst, (size_t)nchr * 8, hipMemcpyDeviceToHost, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return offsets_from_first(ctx, (const long long*)ctx->pin, n, nchr, h_chr_offset);
}

int32_t cvx_clean_f2_offsets(canvas_ctx* ctx, int64_t n, int32_t* d_chr, int32_t* d_start, int32_t* d_stop, float* d_count, int32_t* d_gc, int32_t nchr,
                             const uint8_t* h_chr_is_autosome, const uint8_t* h_chr_is_y, uint32_t flags, int32_t min_bins_per_gc, double* d_cov,
                             double* h_local_sd_out, int64_t* h_n_out, int32_t* h_info, int64_t* h_chr_offset, const void** h_covq_out) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (n < 0 || n >= 0x7FFFFFFFll || nchr <= 0 || nchr > (1 << 20) || !h_chr_is_autosome || !h_n_out || !h_chr_offset || !h_covq_out || !d_cov) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_sample_pipeline: bad arguments");
    *h_covq_out = nullptr;
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const bool fused = n > 0 && !(flags & CANVAS_CLEAN_LOESS) && min_bins_per_gc >= 100 && !cvx_hook("CANVAS_CLEAN_HOST_DRIVEN") && !cvx_hook("CANVAS_HMM_RADIX_SELECT") && !cvx_hook("CANVAS_PIPELINE_UNFUSED");
    bool cleaned = false;
    if (fused) {
        // everything the three stages send back lands in ctx->pin: reserved BEFORE the first copy is enqueued (a later, larger reservation would free the buffer under it)
        const size_t oFirst = (sizeof(CleanDev) + 255) & ~size_t(255);
        int32_t rc = canvas_pin_reserve(ctx, oFirst + (size_t)nchr * 8 + 256); if (rc) return rc;
        const bool useCq = clean_counting_selects() && !ctx->clean_cq_skip;
        ctx->clean_cq_failed = false;
        rc = clean_batch_enqueue(ctx, 1, &n, &d_chr, &d_start, &d_stop, &d_count, &d_gc, nchr, h_chr_is_autosome, flags, min_bins_per_gc, useCq); if (rc) return rc;
        const CleanPending& q = *std::static_pointer_cast<CleanPending>(ctx->clean_batch);
        const unsigned long long* dN = &q.dD->nFinal;          // 0 until the last compaction has run (and stays 0 when the sample leaves the first phase: nothing below does anything then)
        const void* hq = nullptr;
        // (measured and dropped: the F2 hand-off written by the last compaction itself + a counting sweep over its integer keys — the compaction grows from 39 to 61 us, the
        //  counting sweep takes 16: 10 us less per pass than k_quant_covq, at the price of a CanvasClean stage that is 21 us slower by its own account)
        rc = cvx_quant_covq_enqueue(ctx, d_count, n, dN, d_cov, &hq); if (rc) return rc;
        long long* dFirst = (long long*)((char*)ctx->ws + ctx->clean_ws_end);
        CANVAS_HIP_TRY(ctx, hipMemsetAsync(dFirst, 0x7F, (size_t)nchr * 8, ctx->stream));     // "not seen"
        hipLaunchKernelGGL(k_chr_first, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, d_chr, n, nchr, dFirst, dN);
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync((char*)ctx->pin + oFirst, dFirst, (size_t)nchr * 8, hipMemcpyDeviceToHost, ctx->stream));
        char handled = 0; bool second = false; double lsd = -1.0; int64_t nOut = 0; int32_t info[8] = {0};
        rc = clean_batch_finish(ctx, &lsd, &nOut, info, &handled, &second); if (rc) return rc;      // the one synchronisation
        if (handled) {
            cleaned = true;
            *h_n_out = nOut; if (h_local_sd_out) *h_local_sd_out = lsd; if (h_info) memcpy(h_info, info, sizeof info);
            if (!second) { *h_covq_out = hq; return offsets_from_first(ctx, (const long long*)((const char*)ctx->pin + oFirst), nOut, nchr, h_chr_offset); }
            // NormalizeVarianceByGC changed the counts: the last compaction ran in the second phase, after the quantisation had looked at an empty list
        }
    }
    if (!cleaned) { int32_t rc = canvas_clean2(ctx, n, d_chr, d_start, d_stop, d_count, d_gc, nchr, h_chr_is_autosome, h_chr_is_y, flags, min_bins_per_gc, h_local_sd_out, h_n_out, h_info); if (rc) return rc; }
    int32_t rc = cvx_quantize_f2_covq(ctx, d_count, *h_n_out, d_cov, h_covq_out); if (rc) return rc;
    return canvas_chromosome_offsets(ctx, d_chr, *h_n_out, nchr, h_chr_offset);
}

// A cohort through CanvasClean in one call.  The single-sample stage is a chain of ~55 launches on 134 MB that sit in the Infinity Cache, bound by launch latency and one-workgroup
// decision kernels.  The device-driven path is batch-native (clean_fast.hpp): the B samples share every launch (grid.y = B), so the chain is paid once per cohort and the
// working set (B x 134 MB) streams from HBM — the mode in which the stage's HBM roofline fraction means something (hard part 3).  Results per sample are exactly
// those of canvas_clean2.  (Round 2 first ran every sample on a stream and a context of its own: 0.57-1.1 ms per sample, bound by the host's ~80 launches per sample.)
extern "C" int32_t canvas_clean_batch(canvas_ctx* ctx, int32_t nsamples, const int64_t* h_n, int32_t* const* h_d_chr, int32_t* const* h_d_start, int32_t* const* h_d_stop, float* const* h_d_count,
                                      int32_t* const* h_d_gc, int32_t nchr, const uint8_t* h_chr_is_autosome, const uint8_t* h_chr_is_y, uint32_t flags, int32_t min_bins_per_gc,
                                      double* h_local_sd_out, int64_t* h_n_out, int32_t* h_info) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (nsamples <= 0 || nsamples > 64 || !h_n || !h_d_chr || !h_d_start || !h_d_stop || !h_d_count || !h_d_gc || !h_chr_is_autosome || !h_n_out) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_clean_batch: bad arguments (1..64 samples)");
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const bool fast = !(flags & CANVAS_CLEAN_LOESS) && min_bins_per_gc >= 100 && !cvx_hook("CANVAS_CLEAN_HOST_DRIVEN");
    int32_t rcAll = CANVAS_OK;
    std::vector<char> done(nsamples, 0);
    if (fast) {
        // the samples the device-driven path takes form ONE batch: every kernel of the stage runs once with grid.y = the number of samples (clean_fast.hpp)
        std::vector<int> idx;
        for (int s = 0; s < nsamples; s++) if (h_n[s] > 0 && h_n[s] < 0x7FFFFFFFll) idx.push_back(s);
        if (!idx.empty()) {
            const int B = (int)idx.size();
            std::vector<int64_t> n(B), nOut(B, 0); std::vector<int32_t*> c(B), st(B), sp(B), g(B); std::vector<float*> cnt(B);
            std::vector<double> lsd(B, -1.0); std::vector<int32_t> info((size_t)8 * B, 0); std::vector<char> handled(B, 0);
            for (int k = 0; k < B; k++) { const int s = idx[k]; n[k] = h_n[s]; c[k] = h_d_chr[s]; st[k] = h_d_start[s]; sp[k] = h_d_stop[s]; g[k] = h_d_gc[s]; cnt[k] = h_d_count[s]; }
            ctx->clean_cq_failed = false;
            if (flags == CANVAS_CLEAN_GCNORM && B <= CG_MAXB && !cvx_hook("CANVAS_CLEAN_GENERAL_GC")) {
                // -g alone: the three-launch stage of clean_gc_only.hpp with grid.y = B; samples it hands back go through the general chain one by one below
                int32_t rcg = clean_gc_only(ctx, B, n.data(), c.data(), st.data(), sp.data(), cnt.data(), g.data(), nchr, h_chr_is_autosome, min_bins_per_gc, nOut.data(), info.data(), handled.data());
                if (rcg) return rcg;
                for (int k = 0; k < B; k++) if (handled[k]) { const int s = idx[k]; done[s] = 1; h_n_out[s] = nOut[k]; if (h_local_sd_out) h_local_sd_out[s] = -1.0; if (h_info) memcpy(h_info + 8 * s, info.data() + 8 * k, 8 * sizeof(int32_t)); }
                std::vector<int> rest; for (int k = 0; k < B; k++) if (!handled[k]) rest.push_back(idx[k]);
                idx.swap(rest);
            }
        }
        if (!idx.empty()) {
            const int B = (int)idx.size();
            std::vector<int64_t> n(B), nOut(B, 0); std::vector<int32_t*> c(B), st(B), sp(B), g(B); std::vector<float*> cnt(B);
            std::vector<double> lsd(B, -1.0); std::vector<int32_t> info((size_t)8 * B, 0); std::vector<char> handled(B, 0);
            for (int k = 0; k < B; k++) { const int s = idx[k]; n[k] = h_n[s]; c[k] = h_d_chr[s]; st[k] = h_d_start[s]; sp[k] = h_d_stop[s]; g[k] = h_d_gc[s]; cnt[k] = h_d_count[s]; }
            int32_t rc = clean_batch_enqueue(ctx, B, n.data(), c.data(), st.data(), sp.data(), cnt.data(), g.data(), nchr, h_chr_is_autosome, flags, min_bins_per_gc, clean_counting_selects());
            if (rc == CANVAS_OK) rc = clean_batch_finish(ctx, lsd.data(), nOut.data(), info.data(), handled.data());
            if (rc) return rc;
            for (int k = 0; k < B; k++) if (handled[k]) {
                const int s = idx[k]; done[s] = 1; h_n_out[s] = nOut[k];
                if (h_local_sd_out) h_local_sd_out[s] = lsd[k];
                if (h_info) memcpy(h_info + 8 * s, info.data() + 8 * k, 8 * sizeof(int32_t));
            }
        }
    }
    ctx->clean_cq_skip = ctx->clean_cq_failed;       // samples the counting selects gave up on go straight to the radix selects
    struct Unskip { canvas_ctx* c; ~Unskip() { c->clean_cq_skip = false; } } unskip{ctx};
    for (int s = 0; s < nsamples; s++) {
        if (done[s]) continue;                      // empty sample, LOESS / -w < 100, or the device path handed the sample back (its arrays are untouched)
        const int32_t rc = canvas_clean2(ctx, h_n[s], h_d_chr[s], h_d_start[s], h_d_stop[s], h_d_count[s], h_d_gc[s], nchr, h_chr_is_autosome, h_chr_is_y, flags, min_bins_per_gc,
                                         h_local_sd_out ? h_local_sd_out + s : nullptr, h_n_out + s, h_info ? h_info + 8 * s : nullptr);
        if (rc && rcAll == CANVAS_OK) rcAll = rc;
    }
    return rcAll;
}

extern "C" int32_t canvas_clean(canvas_ctx* ctx, int64_t n, int32_t* d_chr, int32_t* d_start, int32_t* d_stop, float* d_count,
                                int32_t* d_gc, int32_t nchr, const uint8_t* h_chr_is_autosome, uint32_t flags, int32_t min_bins_per_gc,
                                double* h_local_sd_out, int64_t* h_n_out, int32_t* h_info) {
    return canvas_clean2(ctx, n, d_chr, d_start, d_stop, d_count, d_gc, nchr, h_chr_is_autosome, nullptr, flags, min_bins_per_gc, h_local_sd_out, h_n_out, h_info);
}
