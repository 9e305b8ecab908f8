// CanvasClean -m LOESS on MI355X: LoessGCNormalizer (CanvasClean/LoessGCNormalizer.cs:9-133) + LoessInterpolator
// (CanvasClean/LoessInterpolator.cs:61-301,358-492) + GoldenSectionSearch (CanvasCommon/Utilities.cs:1014-1044).
//
// MI355X-first redesign, not a port of the loops: the abscissa is the bin's GC percentage, so it takes <= 101 distinct values.
// After a STABLE grouping of y = log(count) by GC (one radix-rank pass) and one prefix scan over the grouped array, every
// tricube-weighted window sum of the reference (sum over ceil(bw*n) ~ 1-2 M points, 101 fits x 2 trainings x ~25 bandwidths)
// collapses to <= 101 range sums of that prefix array: P[b+1] - P[a] per GC group inside the window.  The whole golden-section
// search then costs O(25 x 101^2) scalar operations on the host; the GPU does the O(n) passes (log, group, scan, apply).
// The reference's window logic (updateBandwidthInterval / computeIntervals / Predict, including the (start, COUNT) Range quirk Q7
// and the stale-interval quirk of computeIntervals) is reproduced exactly on the grouped representation; only the association of
// the floating sums differs, which is what the LOESS tolerance (1e-5 relative, BASELINE.json) allows.  MedianByGC stays bit-exact.
#pragma once
#include "common.hpp"
#include "select.hpp"
#include <algorithm>
#include <cmath>
#include <limits>

#define LO_NGC 101
#define LO_TILE 2048

// y = log(count) as double for bins that enter the model (finite log, optionally not chrY); key = gc (0..100) or 127 = excluded
__global__ void __launch_bounds__(256) k_loess_keys(const float* __restrict__ count, const int32_t* __restrict__ gc, const int32_t* __restrict__ chr, const uint8_t* __restrict__ isY,
                                                    int excludeY, int64_t n, uint8_t* __restrict__ key, double* __restrict__ y) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double v = log((double)count[i]);                     // Math.Log(x) (CanvasClean.cs:149)
    bool use = !isinf(v) && !(excludeY && isY[chr[i]]);
    int g = gc[i];
    key[i] = (use && g >= 0 && g <= 100) ? (uint8_t)g : (uint8_t)127;
    y[i] = v;
}
// per-tile histogram of keys
__global__ void __launch_bounds__(256) k_loess_tile_hist(const uint8_t* __restrict__ key, int64_t n, uint32_t* __restrict__ tileHist /* [ntiles][128] */) {
    __shared__ uint32_t h[128];
    if (threadIdx.x < 128) h[threadIdx.x] = 0;
    __syncthreads();
    int64_t base = (int64_t)blockIdx.x * LO_TILE;
    for (int j = 0; j < LO_TILE / 256; j++) { int64_t i = base + j * 256 + threadIdx.x; if (i < n) atomicAdd(&h[key[i]], 1u); }
    __syncthreads();
    if (threadIdx.x < 128) tileHist[(size_t)blockIdx.x * 128 + threadIdx.x] = h[threadIdx.x];
}
// column-wise exclusive scan over tiles for each key: one workgroup per key (128 workgroups)
__global__ void __launch_bounds__(1024) k_loess_col_scan(uint32_t* __restrict__ tileHist, int ntiles, uint32_t* __restrict__ keyTotal) {
    __shared__ uint32_t sh[17];
    const int k = blockIdx.x;
    uint32_t carry = 0;
    for (int base = 0; base < ntiles; base += 1024) {
        int t = base + threadIdx.x;
        uint32_t v = t < ntiles ? tileHist[(size_t)t * 128 + k] : 0;
        uint32_t inc = wave_inclusive_scan_u32(v);
        int w = threadIdx.x >> 6;
        if (lane_id() == 63) sh[w] = inc;
        __syncthreads();
        if (threadIdx.x == 0) { uint32_t s = 0; for (int q = 0; q < 16; q++) { uint32_t tt = sh[q]; sh[q] = s; s += tt; } sh[16] = s; }
        __syncthreads();
        if (t < ntiles) tileHist[(size_t)t * 128 + k] = carry + sh[w] + inc - v;
        carry += sh[16];
        __syncthreads();
    }
    if (threadIdx.x == 0) keyTotal[k] = carry;
}
// stable scatter: element i of key g goes to groupOff[g] + (elements of key g before i).  Rank inside the tile = sequential walk of the
// tile by ONE wave in 64-element steps with ballot-based multi-split (lanes with the same key, lower lane first).
__global__ void __launch_bounds__(64) k_loess_scatter(const uint8_t* __restrict__ key, const double* __restrict__ y, int64_t n, const uint32_t* __restrict__ tileEx,
                                                      const uint32_t* __restrict__ groupOff, double* __restrict__ yGrouped) {
    __shared__ uint32_t run[128];
    const int l = threadIdx.x;
    run[l] = tileEx[(size_t)blockIdx.x * 128 + l]; run[l + 64] = tileEx[(size_t)blockIdx.x * 128 + 64 + l];
    __syncthreads();
    int64_t base = (int64_t)blockIdx.x * LO_TILE;
    for (int j = 0; j < LO_TILE / 64; j++) {
        int64_t i = base + j * 64 + l;
        int k = i < n ? key[i] : 128;                      // 128 = out of range (matches nobody real: 8 bits compared below)
        unsigned long long same = ~0ull;
#pragma unroll
        for (int b = 0; b < 8; b++) { unsigned long long bal = __ballot((k >> b) & 1); same &= ((k >> b) & 1) ? bal : ~bal; }
        unsigned long long lt = l == 0 ? 0ull : (~0ull >> (64 - l));
        uint32_t rankInWave = __popcll(same & lt), cntInWave = __popcll(same);
        uint32_t r0 = 0;
        if (k < 128) r0 = run[k];
        if (k < 127) yGrouped[(size_t)groupOff[k] + r0 + rankInWave] = y[i];
        __syncthreads();
        if (k < 128 && rankInWave == 0) run[k] = r0 + cntInWave;      // one lane per distinct key advances the running offset
        __syncthreads();
    }
}
// inclusive->exclusive prefix sums of doubles: P[j] = sum_{j' < j} v[j'], P has m + 1 entries (3-kernel scan)
__global__ void __launch_bounds__(256) k_dscan_block(const double* __restrict__ v, int64_t m, double* __restrict__ blockSum) {
    __shared__ double sh[4];
    int64_t base = (int64_t)blockIdx.x * LO_TILE;
    double s = 0;
    for (int j = 0; j < LO_TILE / 256; j++) { int64_t i = base + j * 256 + threadIdx.x; if (i < m) s += v[i]; }
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if (lane_id() == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) blockSum[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ void k_dscan_top(double* __restrict__ blockSum, int nb) {     // small: sequential
    if (threadIdx.x == 0 && blockIdx.x == 0) { double s = 0; for (int i = 0; i < nb; i++) { double t = blockSum[i]; blockSum[i] = s; s += t; } }
}
__global__ void __launch_bounds__(64) k_dscan_write(const double* __restrict__ v, int64_t m, const double* __restrict__ blockEx, double* __restrict__ P) {
    int64_t base = (int64_t)blockIdx.x * LO_TILE;
    double run = blockEx[blockIdx.x];
    const int l = threadIdx.x;
    for (int j = 0; j < LO_TILE / 64; j++) {
        int64_t i = base + j * 64 + l;
        double x = i < m ? v[i] : 0.0, inc = x;
        for (int d = 1; d < 64; d <<= 1) { double t = __shfl_up(inc, d, 64); if (l >= d) inc += t; }
        if (i < m) P[i] = run + inc - x;
        run += __shfl(inc, 63, 64);
    }
    if (blockIdx.x == gridDim.x - 1 && l == 0) P[m] = run;
}
// bin.Count = (float)Math.Exp(log(count) - fittedByGC[k] + medianY), k = clamp(gc - minGC) (LoessGCNormalizer.cs:83-88, CanvasClean.cs:150)
__global__ void __launch_bounds__(256) k_loess_apply(float* __restrict__ count, const int32_t* __restrict__ gc, int64_t n, const double* __restrict__ fitted, int nfit, int minGC, double medianY) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int k = gc[i] - minGC; k = k < 0 ? 0 : k; k = k > nfit - 1 ? nfit - 1 : k;
    double smoothed = log((double)count[i]) - fitted[k] + medianY;
    count[i] = (float)exp(smoothed);
}
__global__ void __launch_bounds__(256) k_loess_ykeys(const uint8_t* __restrict__ key, const double* __restrict__ y, int64_t n, unsigned long long* __restrict__ keys, unsigned int* __restrict__ cnt) {
    // compaction-free: excluded elements get +inf keys (sorted last); cnt = number of included elements
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    bool use = key[i] != 127;
    keys[i] = use ? key_of_double(y[i]) : ~0ull;
    if (use) atomicAdd(cnt, 1u);
}

// ------------------------------------------------------------------------------------------------ host: grouped LOESS arithmetic
namespace loess {

struct Grouped {                  // sorted-by-x view of the data: group g = all points with x == g, in file order
    int64_t n = 0;                // points in the model
    int64_t off[LO_NGC + 1];      // sorted index of the first point of group g
    const double* P = nullptr;    // prefix sums of y over the grouped array (n + 1 entries)
    // y' = y - shift[g] (second training of the objective); shift = 0 for the raw data
    double shift[LO_NGC];
    int val(int64_t idx) const { int lo = 0, hi = LO_NGC - 1; while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (off[mid] <= idx) lo = mid; else hi = mid - 1; } return lo; }
    int minX() const { for (int g = 0; g < LO_NGC; g++) if (off[g + 1] > off[g]) return g; return 0; }
    int maxX() const { for (int g = LO_NGC - 1; g >= 0; g--) if (off[g + 1] > off[g]) return g; return 0; }
};
struct Interval { double xmin, xmax; int64_t l, r; };

// updateBandwidthInterval (LoessInterpolator.cs:253-283) on the grouped array: the two while-loops advance one index at a time in the
// reference; here they jump over runs in which xval[left], xval[right] and xval[right+1] do not change.
static bool update_interval(double x, const Grouped& G, int64_t& l, int64_t& r) {
    bool updated = false;
    const int64_t n = G.n;
    // while (right < n-1 && x > xval[right]) { left++; right++; }
    if (r < n - 1 && x > (double)G.val(r)) {
        int g = (int)std::ceil(x); if (g < 0) g = 0;
        int64_t target = g > LO_NGC - 1 ? n : G.off[g];         // first index whose value >= x
        // skip empty groups: off[g] already is the first index with value >= g
        int64_t nr = std::min<int64_t>(n - 1, std::max<int64_t>(r, target));
        if (nr > r) { l += nr - r; r = nr; updated = true; }
    }
    // while (right < n-1 && xval[right+1] - x < x - xval[left]) { left++; right++; }
    while (r < n - 1) {
        int vl = G.val(l), vr1 = G.val(r + 1);
        if (!((double)vr1 - x < x - (double)vl)) break;
        int64_t d = std::min<int64_t>(G.off[vl + 1] - l, std::min<int64_t>(G.off[vr1 + 1] - (r + 1), n - 1 - r));   // steps during which both values stay
        if (d < 1) d = 1;
        l += d; r += d; updated = true;
    }
    return updated;
}
// computeIntervals (LoessInterpolator.cs:171-190) with xStep = 1
static std::vector<Interval> compute_intervals(const Grouped& G, int64_t bandwidthInPoints) {
    std::vector<Interval> iv;
    int64_t l = 0, r = bandwidthInPoints - 1;
    double xMin = -std::numeric_limits<double>::infinity();
    const double x0 = (double)G.minX(), x1 = (double)G.maxX();
    for (double x = x0; x <= x1; x += 1.0) {
        int64_t nl = l, nr = r;
        if (update_interval(x, G, nl, nr)) { iv.push_back({xMin, x, l, r}); xMin = x; l = nl; r = nr; }
    }
    iv.push_back({xMin, std::numeric_limits<double>::infinity(), l, r});
    return iv;
}
static inline double tricube(double x) { double t = 1 - x * x * x; return t * t * t; }
// computeCoefficients + predict (LoessInterpolator.cs:187-249) on the grouped representation
static double fit_at(double x, const Grouped& G, int64_t l, int64_t r) {
    const int vl = G.val(l), vr = G.val(r);
    const int edge = (x - (double)vl > (double)vr - x) ? vl : vr;
    const double denom = std::fabs(1.0 / ((double)edge - x));
    double sumW = 0, sumX = 0, sumXX = 0, sumY = 0, sumXY = 0;
    for (int g = vl; g <= vr; g++) {
        int64_t a = std::max(l, G.off[g]), b = std::min(r, G.off[g + 1] - 1);
        if (b < a) continue;
        double m = (double)(b - a + 1);
        double ys = (G.P[b + 1] - G.P[a]) - m * G.shift[g];
        double xk = (double)g;
        double w = tricube(std::fabs(x - xk) * denom);
        double xkw = xk * w;
        sumW += m * w; sumX += m * xkw; sumXX += m * (xk * xkw); sumY += ys * w; sumXY += ys * xkw;
    }
    double meanX = sumX / sumW, meanY = sumY / sumW, meanXY = sumXY / sumW, meanXX = sumXX / sumW;
    double beta = (meanXX == meanX * meanX) ? 0 : (meanXY - meanX * meanY) / (meanXX - meanX * meanX);
    double alpha = meanY - beta * meanX;
    return alpha + x * beta;
}
// LoessModel.Predict(Enumerable.Range(minGC, maxGC)) (LoessInterpolator.cs:419-443; Q7: `maxGC` is a COUNT)
static std::vector<double> predict_by_gc(const Grouped& G, const std::vector<Interval>& iv, int minGC, int maxGC) {
    std::vector<double> out(maxGC > 0 ? maxGC : 0);
    size_t idx = 0;
    for (int i = 0; i < maxGC; i++) {
        double x = (double)(minGC + i);
        while (idx < iv.size() - 1 && iv[idx].xmax <= x) idx++;
        out[i] = fit_at(x, G, iv[idx].l, iv[idx].r);
    }
    return out;
}
static std::vector<double> train_predict(const Grouped& G, double bandwidth, int minGC, int maxGC) {
    int64_t bw = (int64_t)std::ceil(bandwidth * (double)G.n);       // LoessInterpolator.cs:101
    return predict_by_gc(G, compute_intervals(G, bw), minGC, maxGC);
}
// LoessGCNormalizer.objective (LoessGCNormalizer.cs:98-131): SD of the second-pass fitted values over all points
static double objective(double bandwidth, Grouped G, double medianY) {
    const int minGC = G.minX(), maxGC = G.maxX();
    for (int g = 0; g < LO_NGC; g++) G.shift[g] = 0;
    std::vector<double> fit1 = train_predict(G, bandwidth, minGC, maxGC);
    // normalized[i] = counts[i] - fittedByGC[gc - minGC] + medianY   (index gc - minGC is always < maxGC count? the reference indexes the same way)
    for (int g = minGC; g <= maxGC; g++) { int k = g - minGC; double f = k < (int)fit1.size() ? fit1[k] : std::numeric_limits<double>::quiet_NaN(); G.shift[g] = f - medianY; }
    std::vector<double> fit2 = train_predict(G, bandwidth, minGC, maxGC);
    // Utilities.StandardDeviation(fitted) with fitted[i] = fit2[gc_i - minGC]
    double sum = 0; for (int g = minGC; g <= maxGC; g++) { double m = (double)(G.off[g + 1] - G.off[g]); if (m > 0) sum += m * fit2[g - minGC]; }
    double mu = sum / (double)G.n, s2 = 0;
    for (int g = minGC; g <= maxGC; g++) { double m = (double)(G.off[g + 1] - G.off[g]); if (m > 0) { double d = fit2[g - minGC] - mu; s2 += m * d * d; } }
    return std::sqrt(s2 / (double)(G.n - 1));
}
template <class F>
static double golden_section(F f, double a, double b, double tol = 1E-5) {       // Utilities.cs:1014-1044
    const double gr = 0.618034;
    double c = b - gr * (b - a), d = a + gr * (b - a), fc = f(c), fd = f(d);
    while (std::fabs(d - c) > tol) {
        if (fc < fd) { b = d; d = c; fd = fc; c = b - gr * (b - a); fc = f(c); }
        else { a = c; c = d; fc = fd; d = a + gr * (b - a); fd = f(d); }
    }
    return (b + a) / 2;
}

}  // namespace loess
