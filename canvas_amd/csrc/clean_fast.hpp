// CanvasClean without host round trips (included by clean.hip): CanvasClean.Main (CanvasClean/CanvasClean.cs:415-533) for the MedianByGC flavour with the default
// weighted-median setting (-w >= 100, which makes every GC bucket that survives RemoveBinsWithExtremeGC hold >= 100 autosomal bins: CanvasClean.cs:207-237,178-187).
//
// The stage is EIGHT launches and one synchronisation:
//   k_cf_init        zeroes the stage's counters and publishes the argument table (which arrives as a kernel argument: no copy engine, no memset node)
//   k_cf_size        exact per-size counts of the bins; the last workgroup to finish reads the 98th percentile off them (RemoveBigBins threshold, CanvasClean.cs:328-348)
//   k_cf_flags_ab    RemoveBigBins + RemoveOutliers flags in one pass + the GC histogram of the survivors; the last workgroup totals the block counts, takes the
//                    RemoveBinsWithExtremeGC decision (:207-237) and prepares the counting selects (window, tiles, ranks)
//   k_cf_scatter_ab  ONE compaction caller -> scratch (every workgroup sums the counts of the blocks in front of it: no scan kernel) + the autosomal keys grouped by GC
//   k_cf_hist_lsd    two roles in one grid: per-value counters of the grouped keys  |  window SDs of the compacted counts (:243-298); the last window workgroup builds
//                    the chromosome runs
//   k_cf_pick_mad    two roles: per-GC medians / quartile statistics from the counters, the weighted count of the normalised values, and in the last workgroup the genome's
//                    quartiles + the NormalizeVarianceByGC decision (:34-83)  |  median and MAD of every chromosome run's window SDs; the last one averages them
//   k_cf_flags_final GC strip + local-SD filter flags (CountDeviation is read from the window SDs: it is never stored per bin)
//   k_cf_scatter_final the last compaction, scratch -> caller's arrays, applying NormalizeByGC on the way out
// Every decision the reference takes once per file is taken by the LAST workgroup of the kernel that produces its input (an arrival ticket per kernel): what that
// workgroup reads from the others was published with device-scope atomics or write-through stores and is read back with sc1 loads (cf_ld / cf_st below), its results go
// to CleanDev with plain stores and are read by the NEXT kernel.  (Round 2 ran each decision as a one-workgroup kernel: ten launches of 5-12 us and their boundaries.)
// The caller's arrays are not written before the last kernel, so a case this path does not cover (more than CF_MAXRUN chromosome runs) is detected on the device, leaves the
// input intact and is handed to the host-driven path of clean.hip.  All per-bin arithmetic and all order statistics are the ones of that path: results are bit-identical.
//
// The stage is BATCH-NATIVE: every kernel takes a table of per-sample argument blocks (CfArgs, in device memory) and works on the block of blockIdx.y, so a cohort of B samples
// costs the same launches as one sample (grid.y = B).  One sample is a batch of one.
//
// Order statistics: the counts of a .binned file are two-decimal values, so the medians and quartiles are read off exact per-value counters (CfCq below) — one sweep and
// one pick per stage instead of four radix passes per select; the radix selects (select.hpp) remain for any other input and for the rare second NormalizeByGC.
// No kernel ends with an atomic on ONE address per workgroup (a serial chain at the memory side): the per-GC counters and cursors are replicated CF_HREP times.
#pragma once

#define CF_MAXQ 640          // 6 + 6 * 101 quartile queries (variance normalisation); 2 + 2 * 101 median queries
#define CF_MAXRUN 1024       // chromosome runs of the bin list handled on the device
#define CF_NPROB 4           // select problems of a sample: [1] medians, [2] quartiles, [3] medians after the variance normalisation ([0] unused)
#define CF_SZ_BINS 65536     // bin sizes counted exactly (a larger 98th percentile hands the sample to the host-driven path)
#define CF_HREP 16           // replicas of the per-GC counters the flag / scatter kernels add to (workgroup % CF_HREP picks one): 2 338 workgroups adding to ONE address
                             // are a serial chain at the memory side (measured: a single per-workgroup atomicAdd on one word cost k_cf_flags_ab 31 of its 82 us)
#define CF_SZ_LDS 4096       // ... of which the first CF_SZ_LDS are counted in LDS per workgroup (WGS bins are a few hundred to a few thousand positions)
#define CF_MADB 64           // workgroups of the run-MAD role (each takes runs r, r + CF_MADB, ...)
#define CF_BYVAL 4           // samples whose argument blocks travel as a kernel argument of k_cf_init (a larger cohort uses one small H2D copy)

struct CleanDev {
    unsigned long long nAB;          // bins after RemoveBigBins + RemoveOutliers
    unsigned long long nFinal;       // bins after the GC strip and the local-SD filter
    unsigned int nA;                 // bins after RemoveBigBins alone
    unsigned int bad;                // a gc outside 0..100 or a chromosome index outside the table
    unsigned int fallback;           // not covered on the device: the host-driven path takes over (the caller's arrays are untouched)
    unsigned int nRunRec;
    unsigned int sizeOn;             // RemoveBigBins applies (the filter is switched on and the percentile index lies inside the list)
    int32_t sizeThresh;              // its threshold (CanvasClean.cs:328-348)
    unsigned long long sizeBelow;    // bins with a negative size (they sort in front of every counted size)
    unsigned int nOver, pad1;        // bins of CF_SZ_BINS positions and more (kept in a list: the percentile is taken from it when it lies that far out)
    uint32_t hist[2 * NGC];          // [0..100] autosomal bins per GC, [101..201] the other bins (after the first compaction)
    uint32_t segOff[NGC + 1];        // grouped autosomal bins of the kept GC values
    uint8_t keepGc[NGC + 3];
    long long kept;                  // bins of any chromosome that survive the GC strip
    int gcActive, haveLocalSd, varActive, changed, nruns;
    int cqFail;                      // the counting selects could not decide (a count that is not a two-decimal value, an order statistic outside the window): fallback is set with it

    double medians[NGC];
    double globalMedian;
    VarTab tab;
    double localSd;
};
struct CfSel {                       // a select problem built on the device (select.hpp's tiles / queries)
    uint32_t hdr[4];                 // [0] tiles, [1] queries
    int32_t first[NGC + 1];          // first query of GC bucket g (slot NGC: the genome), -1 = none
    unsigned long long qk[CF_MAXQ], qprefix[CF_MAXQ];
    SelSegQ segq[NGC];
};
// Counting selects.  CanvasClean reads its counts from the F2 text CanvasBin (or CanvasNormalize) wrote, so every count is the float of a two-decimal value k / 100 and
// x -> k = llrint(100 x) is strictly increasing on them: an order statistic of the counts is an order statistic of the integers k, and those are read off exact counters
// per value (one sweep) instead of four radix passes.  The counters cover a window of CQW values around the sample's level (estimated from 33 strided counts), one row per GC
// bucket plus one for the genome; keys below the window are counted per row, keys above it are what is left.  Every key is checked (cq_value(k) == x, below) and every
// requested rank must fall inside the window: otherwise cqFail is raised and the sample is redone with the radix selects — the result never depends on the window.
#define CQW 16384            // counter slots per row: +-82 count units around the level
#define CQ_TILE 32768        // keys per workgroup of the counting sweep (the 64 KB of LDS counters are zeroed and flushed once per tile)
#define CQ_ROWS (NGC + 2)    // counter rows of a sample: one per GC bucket, [NGC] the genome, [NGC + 1] the weighted count of the normalised values
struct CfCq {
    int32_t lo; uint32_t bad, fail, ntiles;
    unsigned long long nbelow;       // normalised counts under the window of the weighted count
    uint32_t below[NGC + 1];         // keys under the window, per bucket and [NGC] for the genome
    uint32_t inWin[NGC + 1];         // keys inside it
    int32_t kq[NGC][6];              // the quartile order statistics of a bucket (as k), in quartile_indices order
};
// k -> cq_value(k) may be ANY non-decreasing map as long as the check and the reconstruction use the same one: a key is accepted only if cq_value(k) gives it back bit for
// bit, so accepted keys are in strictly increasing correspondence with their k.  (float)(k * 0.01) is that map (one multiplication; the division k / 100.0 made
// the counting sweep VALU-bound: 37 -> 31 us).  It reproduces every integer count — the read counts of a WGS .binned file — and the float.Parse of a two-decimal text except where
// the double product lies within an ulp of a float rounding boundary (~7e-9 of the values: about 2 % of 3 M-bin samples with fractional counts hold one and take the
// radix selects; tests/test_counting_key_map.py restates the map in numpy).
__device__ __forceinline__ float cq_value(long long k) { return (float)((double)k * 0.01); }
__device__ __forceinline__ bool cq_key(float x, long long& k) {
    const double y = (double)x * 100.0;
    if (!(y >= -0.5 && y < 1073741823.0)) { k = -1; return false; }            // (NaN fails the comparison too)
    const int ki = __double2int_rn(y);
    k = ki;
    return ki >= 0 && cq_value(ki) == x && !(ki == 0 && (__float_as_uint(x) >> 31));      // (-0.0 sorts in front of 0.0)
}
// Utilities.Quartiles (CanvasCommon/Utilities.cs:361-419) without arrays (a local array indexed by a loop variable lives in scratch memory, and a kernel that uses scratch
// is dispatched more slowly — 40 -> 60 us for k_cf_flags_ab): how many order statistics a list of iSize needs, and the k-th of them in quartile_indices' order
__device__ __forceinline__ int quart_n(int64_t iSize) { if (iSize % 2 == 0) return ((iSize / 2) % 2 == 0) ? 6 : 4; return 5; }
__device__ __forceinline__ int64_t quart_idx(int64_t iSize, int k) {
    const int64_t iMid = iSize / 2;
    if (iSize % 2 == 0) {
        const int64_t mm = iMid / 2;
        if (k == 0) return iMid - 1;
        if (k == 1) return iMid;
        if (iMid % 2 == 0) return k == 2 ? mm - 1 : (k == 3 ? mm : (k == 4 ? iMid + mm - 1 : iMid + mm));
        return k == 2 ? mm : mm + iMid;
    }
    if (k == 0) return iMid;
    if ((iSize - 1) % 4 == 0) { const int64_t n = (iSize - 1) / 4; return k == 1 ? n - 1 : (k == 2 ? n : (k == 3 ? 3 * n : 3 * n + 1)); }
    const int64_t n = (iSize - 3) / 4; return k == 1 ? n : (k == 2 ? n + 1 : (k == 3 ? 3 * n + 1 : 3 * n + 2));
}
// the scratch copy between the two compactions: the five columns of the survivors, GC as one byte; CountDeviation (GenomicBin.cs:83) is never materialised —
// the only reader (RemoveBinsWithExtremeLocalSD, :308-322) takes it from the window SD of the bin's window
struct Soa1 { int32_t *chr, *start, *stop; float* count; uint8_t* gc; };
struct GSoa1 { gptr<int32_t> chr, start, stop; gptr<float> count; gptr<uint8_t> gc; };
__device__ __forceinline__ GSoa1 as_global(const Soa1& s) { return GSoa1{as_global(s.chr), as_global(s.start), as_global(s.stop), as_global(s.count), as_global(s.gc)}; }
struct CfArgs {                      // one sample of the batch
    int64_t n;                       // bins handed in
    int32_t nb, nchr, minBinsPerGc, wantLsd, doSize, doOutlier;
    uint32_t flags, tilesUpper;
    int32_t useCq, padB;
    Soa caller; Soa1 S1;             // the caller's arrays; the scratch copy between the two compactions
    uint8_t* dFlags;
    unsigned long long* dBlk;        // per block of the input: survivors of both filters | survivors of the size filter << 32 (k_cf_flags_ab)
    uint32_t* dBlkF;                 // per block of the scratch copy: survivors of the last compaction (k_cf_flags_final)
    uint32_t* szHist; uint32_t* szOver; uint32_t szOverCap, padA; uint32_t* keysG; const uint8_t* isAuto;
    double* dSd; double* dRunMad; int64_t* dRunStart; long long* dPos;
    CleanDev* D; CfSel* P; SelTile* tiles; uint32_t* hist;
    CfCq* cq; uint32_t* cqHist; SelTile* cqTiles;      // the counting selects (below)
    uint32_t* repl;                  // [CF_HREP][2 * NGC] GC counts of the survivors per replica (k_cf_flags_ab), then [CF_HREP][NGC] write cursors into the grouped keys (its last workgroup -> k_cf_scatter_ab)
    unsigned long long* dbg;         // profiling hook (CANVAS_CLEAN_DEBUG_CLOCKS=1): [64] wall-clock stamps of selected workgroups, else nullptr
    uint32_t* tick;                  // [8][CF_TICK_WORDS] arrival tickets (cf_arrive_last): 0 k_cf_size, 1 k_cf_flags_ab, 2 window role of k_cf_hist_lsd, 3 / 4 pick and run role of k_cf_pick_mad
};
struct CfArgsPack { CfArgs a[CF_BYVAL]; uint8_t isAuto[256]; };
#define CF_SAMPLE const CfArgs& A = AA[blockIdx.y]
#define CF_STAMP(slot) do { if (A.dbg && threadIdx.x == 0) A.dbg[slot] = wall_clock64(); } while (0)

// ---------------------------------------------------------------- hand-off inside a launch
// A value one workgroup publishes for the last workgroup of the same launch: write-through (sc1) store / device-scope atomic on the producer's side (no release fence:
// the per-XCD L2s are not coherent and a CU's L1 is never refreshed by another CU's stores); the last workgroup issues ONE agent-scope acquire after its ticket
// (cf_arrive_last) and reads with plain loads, which the compiler keeps in flight together — sc1 loads (cf_ld) are issued one at a time: a tail of 40 of them per
// thread was 20 us.
template <class T> __device__ __forceinline__ void cf_st(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> __device__ __forceinline__ T cf_ld(const T* p) { return *p; }        // (after the acquire of cf_arrive_last)
__device__ __forceinline__ void cf_st_f64(double* p, double v) { cf_st(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v)); }
__device__ __forceinline__ double cf_ld_f64(const double* p) { return *p; }
// Arrival ticket (zero at launch): true in the workgroup that arrives last.  Every wave drains its own stores and atomics first, so whatever a workgroup published is
// in memory before its ticket is.  TWO LEVELS: device-scope atomics on one address are performed one after the other at the memory side (~7 ns each), and the
// workgroups of a streaming kernel all arrive within a few microseconds — 2 339 tickets on one word were 17 of k_cf_flags_ab's 53 us (measured by leaving the
// ticket out).  So workgroup idx counts on sub-counter idx % CF_TICK_SUB (each on a 128-byte line of its own), and the last arrival of a sub-counter counts on the
// top word: chains of expected / 32 + 32 instead of expected.
#define CF_TICK_SUB 32
#define CF_TICK_WORDS (32 * (1 + CF_TICK_SUB))          // words of one ticket: the top word and the sub-counters, 128 bytes apart
__device__ __forceinline__ bool cf_arrive_last(uint32_t* tick, uint32_t idx, uint32_t expected, int* sFlag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        int last = 0;
        if (expected <= 2 * CF_TICK_SUB) last = (__hip_atomic_fetch_add(tick, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == expected) ? 1 : 0;
        else {
            const uint32_t r = idx % CF_TICK_SUB, members = (expected - r + CF_TICK_SUB - 1) / CF_TICK_SUB;       // indices below `expected` that are r modulo CF_TICK_SUB
            // (the returned value is waited for before the top word is touched: every member's stores were drained before its own count)
            if (__hip_atomic_fetch_add(tick + 32 * (1 + r), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == members)
                last = (__hip_atomic_fetch_add(tick, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == CF_TICK_SUB) ? 1 : 0;
        }
        if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");           // buffer_inv sc1: this CU's L1 and the XCD's L2 hold nothing stale of what the others published
        *sFlag = last;
    }
    __syncthreads();
    return *sFlag != 0;
}
// sum of one uint32 / uint64 per thread over the workgroup (any multiple of 64 threads up to 1024); every thread gets the total.  Two barriers.
__device__ __forceinline__ unsigned long long cf_block_sum_u64(unsigned long long v, unsigned long long* sh16) {
    v = wave_reduce_add_u64(v);
    __syncthreads();
    if (lane_id() == 0) sh16[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned long long t = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) t += sh16[w];
    return t;
}
// exclusive prefix sum of one uint32 per thread over a 256-thread workgroup; *total = the sum.  Two barriers.
__device__ __forceinline__ uint32_t cf_excl_scan256(uint32_t v, uint32_t* sh4, uint32_t* total) {
    const uint32_t inc = wave_inclusive_scan_u32(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 63) sh4[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t off = 0, tot = 0;
    for (int w = 0; w < 4; w++) { if (w < (int)(threadIdx.x >> 6)) off += sh4[w]; tot += sh4[w]; }
    *total = tot;
    return inc - v + off;
}
// exclusive prefix sum of one uint32 per thread over a 128-thread workgroup (two waves); *total = the sum.  One barrier.
__device__ __forceinline__ uint32_t cf_excl_scan128(uint32_t v, uint32_t* sh2 /* [2] */, uint32_t* total) {
    const uint32_t inc = wave_inclusive_scan_u32(v);
    if ((threadIdx.x & 63) == 63) sh2[threadIdx.x >> 6] = inc;
    __syncthreads();
    *total = sh2[0] + sh2[1];
    return inc - v + (threadIdx.x >= 64 ? sh2[0] : 0u);
}

// ---------------------------------------------------------------- k_cf_init: the stage's zeros and its argument table
// zero: CleanDev blocks, size counters, replica counters, CfCq blocks, tickets, value counters of all samples (one contiguous region, 16-byte granules)
__global__ void __launch_bounds__(1024) k_cf_init(CfArgs* __restrict__ table, uint8_t* __restrict__ isAutoDev, const CfArgsPack pack, int npack, int nchr, uint4* __restrict__ zero, size_t nvec) {
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 1024) zero[i] = make_uint4(0u, 0u, 0u, 0u);
    if (blockIdx.x == 0 && npack > 0) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&pack.a[0]); uint32_t* dst = reinterpret_cast<uint32_t*>(table);
        const int words = npack * (int)(sizeof(CfArgs) / 4);
        for (int i = threadIdx.x; i < words; i += 1024) dst[i] = src[i];
        for (int i = threadIdx.x; i < nchr; i += 1024) isAutoDev[i] = pack.isAuto[i];
    }
}

// ---------------------------------------------------------------- RemoveBigBins threshold (CanvasClean.cs:328-348): the 98th percentile of the bin sizes
// Sizes are small integers, so the order statistic is read off an exact count per size (one sweep of start / stop, no key array, no radix passes): sizes below CF_SZ_LDS are
// counted in LDS per workgroup and flushed, sizes up to CF_SZ_BINS go to the global counters directly, larger ones only matter if the percentile itself is that large.
#define CF_SZ_GRID 256       // workgroups per sample in the size count
__device__ __forceinline__ void cf_size_count(const CfArgs& A, int32_t sz, uint32_t* lh, uint32_t& neg) {
    if (sz < 0) neg++;
    else if (sz < CF_SZ_LDS) atomicAdd(&lh[sz], 1u);
    else if (sz < CF_SZ_BINS) atomicAdd(&A.szHist[sz], 1u);
    else { const unsigned int k = atomicAdd(&A.D->nOver, 1u); if (k < A.szOverCap) cf_st(&A.szOver[k], (uint32_t)sz); }
}
__global__ void __launch_bounds__(1024) k_cf_size(const CfArgs* __restrict__ AA) {
    CF_SAMPLE;
    __shared__ uint32_t lh[CF_SZ_LDS];
    __shared__ int sLast;
    if (!A.doSize) return;                                                      // sizeOn stays 0 (k_cf_init)
    const int64_t n = A.n;
    for (int i = threadIdx.x; i < CF_SZ_LDS; i += 1024) lh[i] = 0;
    __syncthreads();
    const gptr<const int32_t> start = as_global(A.caller.start), stop = as_global(A.caller.stop);
    uint32_t neg = 0;
    const bool vec = (((uintptr_t)A.caller.start | (uintptr_t)A.caller.stop) & 15) == 0;
    const int64_t nvec = vec ? n / 4 : 0;
    for (int64_t v = (int64_t)blockIdx.x * 1024 + threadIdx.x; v < nvec; v += (int64_t)CF_SZ_GRID * 2048) {      // two 16-byte loads per column in flight
        const int64_t v2 = v + (int64_t)CF_SZ_GRID * 1024;
        const uint4 a0 = gload_uint4(start + 4 * v), b0 = gload_uint4(stop + 4 * v);
        uint4 a1 = make_uint4(0, 0, 0, 0), b1 = a1;
        if (v2 < nvec) { a1 = gload_uint4(start + 4 * v2); b1 = gload_uint4(stop + 4 * v2); }
        cf_size_count(A, (int32_t)b0.x - (int32_t)a0.x, lh, neg); cf_size_count(A, (int32_t)b0.y - (int32_t)a0.y, lh, neg);
        cf_size_count(A, (int32_t)b0.z - (int32_t)a0.z, lh, neg); cf_size_count(A, (int32_t)b0.w - (int32_t)a0.w, lh, neg);
        if (v2 < nvec) {
            cf_size_count(A, (int32_t)b1.x - (int32_t)a1.x, lh, neg); cf_size_count(A, (int32_t)b1.y - (int32_t)a1.y, lh, neg);
            cf_size_count(A, (int32_t)b1.z - (int32_t)a1.z, lh, neg); cf_size_count(A, (int32_t)b1.w - (int32_t)a1.w, lh, neg);
        }
    }
    for (int64_t i = 4 * nvec + (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += (int64_t)CF_SZ_GRID * 1024) cf_size_count(A, stop[i] - start[i], lh, neg);
    neg = wave_reduce_add_u32(neg);
    if (lane_id() == 0 && neg) atomicAdd(&A.D->sizeBelow, (unsigned long long)neg);
    __syncthreads();
    for (int i = threadIdx.x; i < CF_SZ_LDS; i += 1024) { const uint32_t v = lh[i]; if (v) atomicAdd(&A.szHist[i], v); }
    if (!cf_arrive_last(A.tick + 0 * CF_TICK_WORDS, blockIdx.x, CF_SZ_GRID, &sLast)) return;
    // ---- last workgroup: the size with cumulative count > index (the element at `index` of the sorted sizes)
    __shared__ unsigned long long sTot[1024 / 64];
    __shared__ int sFound;
    CleanDev* __restrict__ D = A.D;
    const int64_t index = (int64_t)(0.98 * (double)A.n);                       // CanvasClean.cs:339
    if (!(index < A.n)) return;                                                // the percentile index falls past the end: the filter is off (sizeOn stays 0)
    const unsigned long long want = (unsigned long long)index, below = cf_ld(&D->sizeBelow);
    const unsigned int nOver = cf_ld(&D->nOver);
    if (threadIdx.x == 0) sFound = 0;
    // the counters are scanned in two stretches of 1024 x k bins: [0, CF_SZ_LDS) first — where the bins of a WGS sample are — then the rest
    unsigned long long total = below;
    for (int part = 0; part < 2; part++) {
        const int lo = part == 0 ? 0 : CF_SZ_LDS, per = part == 0 ? CF_SZ_LDS / 1024 : (CF_SZ_BINS - CF_SZ_LDS) / 1024;      // 4, 60 bins per thread
        const uint32_t* __restrict__ h = A.szHist + lo + (size_t)threadIdx.x * per;
        unsigned long long mine = 0;
        for (int k = 0; k < per; k += 2) { const unsigned long long q = cf_ld(reinterpret_cast<const unsigned long long*>(h + k)); mine += (q & 0xFFFFFFFFull) + (q >> 32); }
        unsigned long long inc = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const unsigned long long o = __shfl_up(inc, d, 64); if ((int)lane_id() >= d) inc += o; }
        __syncthreads();                                                       // sTot of the previous stretch has been read by everybody
        if (lane_id() == 63) sTot[threadIdx.x >> 6] = inc;
        __syncthreads();
        unsigned long long before = total, sum = 0;
        for (int w = 0; w < 1024 / 64; w++) { if (w < (int)(threadIdx.x >> 6)) before += sTot[w]; sum += sTot[w]; }
        before += inc - mine;
        if (want >= before && want < before + mine) {
            unsigned long long cum = before;
            for (int k = 0; k < per; k++) { cum += cf_ld(h + k); if (want < cum) { D->sizeThresh = lo + (int)threadIdx.x * per + k; break; } }
            sFound = 1;
        }
        total += sum;
        __syncthreads();
        if (sFound) break;
    }
    if (threadIdx.x == 0) {
        D->sizeOn = 1u;
        if (want < below || (want >= total && nOver > A.szOverCap)) D->fallback = 1u;     // a negative percentile, or more large bins than the list holds: the host-driven path
    }
    if (!sFound && want >= total && nOver <= A.szOverCap) {
        // the percentile lies among the bins of CF_SZ_BINS positions and more (a heavy tail of bins across assembly gaps): exact order statistic of the list, in this workgroup
        __shared__ uint32_t sH[2][256];
        __shared__ unsigned long long sPre[2], sK[2];
        const uint32_t* __restrict__ ov = A.szOver;
        const unsigned long long t = want - total;
        __syncthreads();
        wg_select2([&](int64_t i) { return (unsigned long long)cf_ld(ov + i); }, 0, (int64_t)nOver, t, t, sH, sPre, sK);
        if (threadIdx.x == 0) D->sizeThresh = (int32_t)(uint32_t)sPre[0];
    }
}
// the select problems over the grouped keys, genome + every kept bucket: mode 0 = medians (NormalizeByGC, CanvasClean.cs:163-189), mode 1 = quartiles (NormalizeVarianceByGC, :34-66).
// gate: which CleanDev flag switches the problem on (0 gcActive, 1 varActive, 2 changed)
__global__ void __launch_bounds__(128) k_cf_sel_setup(const CfArgs* __restrict__ AA, int which, int mode, int gate) {
    CF_SAMPLE;
    __shared__ uint32_t so[NGC + 1];
    const CleanDev* D = A.D; CfSel* P = A.P + which; SelTile* tiles = A.tiles + (size_t)which * A.tilesUpper;
    const int t = threadIdx.x;
    const bool on = gate == 0 ? D->gcActive != 0 : (gate == 1 ? D->varActive != 0 : D->changed != 0);
    if (!on) { if (t == 0) { P->hdr[0] = 0; P->hdr[1] = 0; } return; }
    if (t <= NGC) so[t] = D->segOff[t];
    __syncthreads();
    auto ranksOf = [&](int64_t cnt, int64_t* ranks) -> int {
        if (cnt <= 0) return 0;
        if (mode == 0) { if (cnt % 2) { ranks[0] = cnt / 2; return 1; } ranks[0] = cnt / 2 - 1; ranks[1] = cnt / 2; return 2; }
        const QuartIdx qi = quartile_indices(cnt); for (int k = 0; k < qi.n; k++) ranks[k] = qi.idx[k]; return qi.n;
    };
    int64_t myRanks[6]; int myN = 0;
    if (t < NGC) myN = ranksOf((int64_t)so[t + 1] - (int64_t)so[t], myRanks);
    else if (t == NGC) myN = ranksOf((int64_t)so[NGC], myRanks);
    // query numbering: the genome's queries first (0 .. nG-1), then the buckets' in GC order; tile numbering: the buckets' tiles in GC order (two 128-thread scans)
    __shared__ uint32_t shQ[2], shT[2]; __shared__ int sNG;
    if (t == NGC) sNG = myN;
    uint32_t totQ, totT;
    const uint32_t exQ = cf_excl_scan128(t < NGC ? (uint32_t)myN : 0u, shQ, &totQ);          // (the barrier inside also publishes sNG)
    const uint32_t myTiles = t < NGC ? (so[t + 1] - so[t] + SEL_TILE - 1) / SEL_TILE : 0u;
    const uint32_t exT = cf_excl_scan128(myTiles, shT, &totT);
    const int nG = sNG;
    if (t == 0) { P->hdr[0] = totT; P->hdr[1] = (uint32_t)nG + totQ; }
    const int f = t < NGC ? nG + (int)exQ : 0;                                              // slot NGC (the genome) starts at query 0
    if (t <= NGC) {
        P->first[t] = myN > 0 ? f : -1;
        for (int k = 0; k < myN; k++) { P->qk[f + k] = (unsigned long long)myRanks[k]; P->qprefix[f + k] = 0ull; }
    }
    if (t < NGC) {
        SelSegQ Q; Q.nq = 0;
        if (so[t + 1] > so[t]) { for (int k = 0; k < nG; k++) Q.q[Q.nq++] = k; for (int k = 0; k < myN; k++) Q.q[Q.nq++] = f + k; }
        P->segq[t] = Q;
        uint32_t k = exT;
        for (int64_t b = so[t]; b < (int64_t)so[t + 1]; b += SEL_TILE) tiles[k++] = SelTile{t, b, min<int64_t>(b + SEL_TILE, (int64_t)so[t + 1])};
    }
}
// the radix passes of a device-built problem (grids are upper bounds: tiles <= n / SEL_TILE + NGC + 1, queries <= CF_MAXQ)
__global__ void __launch_bounds__(256) k_cf_select_hist(const CfArgs* __restrict__ AA, int which, int shift, int firstPass) {
    CF_SAMPLE;
    const CfSel* P = A.P + which;
    select_hist_body<uint32_t>(A.keysG, A.tiles + (size_t)which * A.tilesUpper, P->segq, P->qprefix, shift, firstPass, A.hist, CF_MAXQ, P->hdr);
}
__global__ void __launch_bounds__(64) k_cf_select_pick(const CfArgs* __restrict__ AA, int which, int firstPass) {
    CF_SAMPLE;
    CfSel* P = A.P + which;
    select_pick_body(A.hist, P->qprefix, P->qk, CF_MAXQ, firstPass, P->hdr);
}

// ---------------------------------------------------------------- RemoveBigBins + RemoveOutliers in one pass over the caller's arrays
// keepA(j) = size <= threshold (CanvasClean.cs:349-352); RemoveOutliers (:387-413) looks at the neighbours in the list RemoveBigBins left, i.e. at the nearest
// bins on either side that pass keepA.  Also the range check of gc / chr, the count after the size filter, and the block counts of the compaction.
// A thread takes four consecutive bins per round (16-byte loads; the neighbours inside the quad stay in registers, only the quad's outer neighbours are looked up in LDS).
#define CF_OFF 8             // LDS slot of the block's first bin: slot CF_OFF - 1 = the bin in front of the block, slot CF_OFF + L = the bin behind it
// the last workgroup of k_cf_flags_ab (256 threads): totals, RemoveBinsWithExtremeGC decision (CanvasClean.cs:207-237) and what follows from it, counting-select set-up
__device__ __forceinline__ void cf_decide_gc(const CfArgs& A) {
    __shared__ unsigned long long sh16[16];
    __shared__ uint32_t shA[4], shB[4], shC[4], shD[4], shT[4];
    __shared__ uint32_t so[NGC + 1];
    __shared__ long long sv[33];
    CleanDev* __restrict__ D = A.D; const uint32_t flags = A.flags; const int minBinsPerGc = A.minBinsPerGc;
    const int t = threadIdx.x;
    // ---- everything this workgroup reads from the others is requested first (one round of memory latency): block counts, GC counter replicas, the 33 sample counts
    unsigned long long mineT = 0;
    for (int j0 = t; j0 < A.nb; j0 += 8 * 256) {
        unsigned long long part[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int j = j0 + u * 256; part[u] = j < A.nb ? cf_ld(&A.dBlk[j]) : 0ull; }
#pragma unroll
        for (int u = 0; u < 8; u++) mineT += part[u];
    }
    uint32_t cr[CF_HREP]; uint32_t hA = 0, hO = 0;
#pragma unroll
    for (int r = 0; r < CF_HREP; r++) { cr[r] = 0; if (t < NGC) { cr[r] = cf_ld(&A.repl[r * (2 * NGC) + t]); hO += cf_ld(&A.repl[r * (2 * NGC) + NGC + t]); } }
    float sampleCount = 0.0f; bool haveSample = false;
    if (A.useCq && t < 33) { const int64_t i = (int64_t)((double)A.n * (t + 0.5) / 33.0); if (i < A.n) { sampleCount = as_global(A.caller.count)[i]; haveSample = true; } }
#pragma unroll
    for (int r = 0; r < CF_HREP; r++) hA += cr[r];
    // ---- bins after both filters / after RemoveBigBins alone: the two halves of the block counts
    CF_STAMP(7);
    const unsigned long long tot = cf_block_sum_u64(mineT, sh16);
    CF_STAMP(8);
    const long long nAB = (long long)(tot & 0xFFFFFFFFull);
    if (t == 0) { D->nAB = (unsigned long long)nAB; D->nA = (unsigned int)(tot >> 32); }
    // ---- GC histogram of the survivors (sum of the replicas)
    if (t < NGC) { D->hist[t] = hA; D->hist[NGC + t] = hO; }
    // the counts are integers below 2^32 and there are 101 of them: their double sum (CanvasClean.cs:219-222) is exact in any order
    uint32_t totalA; (void)cf_excl_scan256(hA, shA, &totalA);
    const bool consider = (flags & CANVAS_CLEAN_GCNORM) && nAB > 0;
    const int averageCountPerGC = max(minBinsPerGc, (int)((double)totalA / NGC));
    const int threshold = min(100, averageCountPerGC);
    bool kp = true;
    if (consider && t < NGC) kp = (int)hA >= threshold;
    // bins of any chromosome in the kept buckets (each term < 2^32, 101 terms: the 64-bit total is split over two 32-bit scans of the halves)
    const unsigned long long mine = (consider && t < NGC && kp) ? (unsigned long long)hA + (unsigned long long)hO : 0ull;
    uint32_t totLo, totHi; (void)cf_excl_scan256((uint32_t)(mine & 0xFFFFu), shB, &totLo); (void)cf_excl_scan256((uint32_t)(mine >> 16), shC, &totHi);
    const long long kept = (long long)totLo + ((long long)totHi << 16);
    const bool active = consider && kept > 0;                               // kept <= 0: "proceed without GC correction" (CanvasClean.cs:500-505)
    if (!active) kp = true;
    uint32_t totalKept; const uint32_t soff = cf_excl_scan256((active && t < NGC && kp) ? hA : 0u, shD, &totalKept);
    if (t < NGC) {
        D->keepGc[t] = kp ? 1 : 0; D->medians[t] = 0.0; D->segOff[t] = active ? soff : 0u; so[t] = active ? soff : 0u;
        // the bucket's stretch of the grouped keys is filled replica by replica (the order inside a bucket is irrelevant: only order statistics are taken from it)
        uint32_t at = soff;
#pragma unroll
        for (int r = 0; r < CF_HREP; r++) { A.repl[CF_HREP * (2 * NGC) + r * NGC + t] = at; at += cr[r]; }
    }
    const long long sKept = active ? kept : nAB;
    // NormalizeVarianceByGC runs for whole-genome samples only (CanvasClean.cs:512-519); the host enqueues its kernels when the INPUT has more than 500000 bins
    const bool haveLsd = A.wantLsd && nAB >= 50000;                           // what the window role of k_cf_hist_lsd will store in haveLocalSd (CanvasClean.cs:483-486)
    const bool varActive = active && haveLsd && sKept > 500000 && A.n > 500000;
    if (t == 0) {
        so[NGC] = active ? totalKept : 0u;
        D->segOff[NGC] = active ? totalKept : 0u; D->kept = sKept; D->gcActive = active ? 1 : 0; D->changed = 0; D->varActive = varActive ? 1 : 0;
    }
    CF_STAMP(9);
    if (!A.useCq) return;
    // ---- counting selects: window, sweep tiles and the marker / ranks the later kernels look at
    __syncthreads();
    CfCq* __restrict__ C = A.cq; CfSel* P1 = A.P + 1; CfSel* P2 = A.P + 2;
    const uint32_t total = so[NGC];
    if (!active || total == 0) return;                       // hdr[] and ntiles stay 0 (k_cf_init)
    // the sample's level: the middle one of 33 strided counts of the input that are two-decimal values (any estimate is correct: the window only decides whether the counters suffice)
    if (t < 33) { long long k = -1; if (haveSample && !cq_key(sampleCount, k)) k = -1; sv[t] = k; }
    __syncthreads();
    if (t < 33) {                                            // every lane ranks its own sample among the valid ones; the one in the middle sets the window
        const long long mineK = sv[t];
        int m = 0, rank = 0;
        for (int j = 0; j < 33; j++) { const long long o = sv[j]; if (o >= 0) { m++; if (o < mineK || (o == mineK && j < t)) rank++; } }
        if (m == 0) { if (t == 0) C->lo = 0; }
        else if (mineK >= 0 && rank == m / 2) C->lo = (int32_t)(mineK > CQW / 2 ? mineK - CQW / 2 : 0);
    }
    uint32_t totT;
    const uint32_t myTiles = t < NGC ? (so[t + 1] - so[t] + CQ_TILE - 1) / CQ_TILE : 0u;
    const uint32_t exT = cf_excl_scan256(myTiles, shT, &totT);
    if (t < NGC) { uint32_t k = exT; for (int64_t b = so[t]; b < (int64_t)so[t + 1]; b += CQ_TILE) A.cqTiles[k++] = SelTile{t, b, min<int64_t>(b + CQ_TILE, (int64_t)so[t + 1])}; }
    if (t == 0) {
        C->ntiles = totT;
        P1->hdr[0] = 0; P1->hdr[1] = 1;                      // "NormalizeByGC has been decided" for k_cf_scatter_final / k_cf_apply_gc
        P2->hdr[0] = 0; P2->hdr[1] = 0;
        if (varActive) { P2->hdr[1] = (uint32_t)quart_n((int64_t)total); P2->first[NGC] = 0; }
    }
    // the genome's quartile ranks (weighted count + resolve in k_cf_pick_mad), one thread each
    if (varActive && t < quart_n((int64_t)total)) { P2->qk[t] = (unsigned long long)quart_idx((int64_t)total, t); P2->qprefix[t] = 0ull; }
}
__global__ void __launch_bounds__(256) k_cf_flags_ab(const CfArgs* __restrict__ AA) {
    CF_SAMPLE;
    __shared__ uint32_t sh[8];
    // keepA, chromosome and count of this block's bins and of the bin on either side of it, for the neighbour searches that leave a thread's own eight bins; the search
    // leaves these slots only when the halo bin itself fails the size filter (slow path, global memory).
    __shared__ __attribute__((aligned(16))) uint8_t sA[CBLK + 2 * CF_OFF];
    __shared__ __attribute__((aligned(16))) int32_t sChr[CBLK + 2 * CF_OFF];
    __shared__ __attribute__((aligned(16))) float sCnt[CBLK + 2 * CF_OFF];
    __shared__ uint8_t sAuto[256];                        // isAuto of the first 256 chromosomes (more than that: read from the table)
    __shared__ uint32_t lh[2 * NGC];                      // GC histogram of the survivors (CanvasClean.cs:207-223): [0..100] autosomal, [101..201] the others
    __shared__ int sLast;
    const int64_t n = A.n;
    const int64_t base = (int64_t)blockIdx.x * CBLK;
    if (base >= n) return;
    const int L = (int)min<int64_t>(CBLK, n - base);
    if (blockIdx.x == 0) CF_STAMP(0);
    if (threadIdx.x < 2 * NGC) lh[threadIdx.x] = 0;
    const gptr<const uint8_t> isAuto = as_global(A.isAuto);         // (pointers out of the argument table: converted so that the accesses are global_load, not flat_load)
    const GSoa in = as_global(A.caller);
    const gptr<const int32_t> chr = in.chr, start = in.start, stop = in.stop, gc = in.gc; const gptr<const float> count = in.count;
    const gptr<uint8_t> flags = as_global(A.dFlags); CleanDev* __restrict__ D = A.D;
    const int nchr = A.nchr, doOutlier = A.doOutlier;
    const bool doSize = D->sizeOn != 0;                   // off when the filter is, or when the percentile index falls past the end
    const int32_t thresh = doSize ? D->sizeThresh : 0;
    sAuto[threadIdx.x] = (int)threadIdx.x < nchr ? isAuto[threadIdx.x] : 0;
    const bool vec = ((((uintptr_t)A.caller.chr | (uintptr_t)A.caller.start | (uintptr_t)A.caller.stop | (uintptr_t)A.caller.gc | (uintptr_t)A.caller.count) & 15) == 0);
    uint32_t nSize = 0, bad = 0;
    // ---- eight consecutive bins per thread (two 16-byte loads per column, all in flight together)
    const int li0 = 8 * (int)threadIdx.x;
    int32_t rc[8], rg[8], sz[8]; float rv[8];
    if (vec && li0 + 7 < L) {
        const int64_t i = base + li0;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint4 c = gload_uint4(chr + i + 4 * h), g = gload_uint4(gc + i + 4 * h), s = gload_uint4(start + i + 4 * h), e = gload_uint4(stop + i + 4 * h), v = gload_uint4(count + i + 4 * h);
            rc[4 * h] = (int32_t)c.x; rc[4 * h + 1] = (int32_t)c.y; rc[4 * h + 2] = (int32_t)c.z; rc[4 * h + 3] = (int32_t)c.w;
            rg[4 * h] = (int32_t)g.x; rg[4 * h + 1] = (int32_t)g.y; rg[4 * h + 2] = (int32_t)g.z; rg[4 * h + 3] = (int32_t)g.w;
            sz[4 * h] = (int32_t)e.x - (int32_t)s.x; sz[4 * h + 1] = (int32_t)e.y - (int32_t)s.y; sz[4 * h + 2] = (int32_t)e.z - (int32_t)s.z; sz[4 * h + 3] = (int32_t)e.w - (int32_t)s.w;
            rv[4 * h] = __uint_as_float(v.x); rv[4 * h + 1] = __uint_as_float(v.y); rv[4 * h + 2] = __uint_as_float(v.z); rv[4 * h + 3] = __uint_as_float(v.w);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; e++) {
            rc[e] = -1; rg[e] = 0; rv[e] = 0.0f; sz[e] = 0;
            if (li0 + e < L) { const int64_t i = base + li0 + e; rc[e] = chr[i]; rg[e] = gc[i]; rv[e] = count[i]; sz[e] = stop[i] - start[i]; }
        }
    }
    uint32_t aM = 0;                                      // bit e: bin li0 + e exists and passes the size filter
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const bool in = li0 + e < L;
        if (in && ((uint32_t)rg[e] > 100u || (uint32_t)rc[e] >= (uint32_t)nchr)) bad = 1;
        if (in && (!doSize || sz[e] <= thresh)) aM |= 1u << e;
    }
    nSize = (uint32_t)__builtin_popcount(aM);
    {
        unsigned long long a8 = 0;
#pragma unroll
        for (int e = 0; e < 8; e++) a8 |= (unsigned long long)((aM >> e) & 1u) << (8 * e);
        *reinterpret_cast<unsigned long long*>(&sA[CF_OFF + li0]) = a8;
        *reinterpret_cast<int4*>(&sChr[CF_OFF + li0]) = make_int4(rc[0], rc[1], rc[2], rc[3]); *reinterpret_cast<int4*>(&sChr[CF_OFF + li0 + 4]) = make_int4(rc[4], rc[5], rc[6], rc[7]);
        *reinterpret_cast<float4*>(&sCnt[CF_OFF + li0]) = make_float4(rv[0], rv[1], rv[2], rv[3]); *reinterpret_cast<float4*>(&sCnt[CF_OFF + li0 + 4]) = make_float4(rv[4], rv[5], rv[6], rv[7]);
    }
    const bool hasLeft = base > 0, hasRight = base + L < n;
    if (threadIdx.x < 2) {
        const int64_t i = threadIdx.x == 0 ? base - 1 : base + L;
        uint8_t a = 0; int32_t c = -1; float v = 0.0f;
        if (threadIdx.x == 0 ? hasLeft : hasRight) { c = chr[i]; v = count[i]; a = (!doSize || (stop[i] - start[i]) <= thresh) ? 1 : 0; }
        const int slot = threadIdx.x == 0 ? CF_OFF - 1 : CF_OFF + L;
        sA[slot] = a; sChr[slot] = c; sCnt[slot] = v;
    }
    __syncthreads();
    if (blockIdx.x == 0) CF_STAMP(1);
    // ---- RemoveOutliers (CanvasClean.cs:387-413) on the list RemoveBigBins left: a bin is compared with the nearest bins on either side that pass the size filter.
    // keep = okPrev || okNext || (no neighbour at all), ok = same chromosome && !SignificantlyDifferent; the test is symmetric in its two counts, so a pair of
    // neighbours inside the thread's eight bins is evaluated ONCE (it is the left bin's next-test and the right bin's prev-test): nine divisions per thread, not sixteen
    uint32_t keepM = aM;
    if (doOutlier && aM) {
        uint32_t hasPrevM = 0, okPM = 0, hasNextM = 0, okNM = 0;
        int pe = -1; int32_t cpv = 0; float vpv = 0.0f;       // the previous bin of this thread that passes the size filter
#pragma unroll
        for (int e = 0; e < 8; e++) {
            if (!((aM >> e) & 1u)) continue;
            if (pe >= 0) {
                hasPrevM |= 1u << e; hasNextM |= 1u << pe;
                if (cpv == rc[e] && !sig_diff(rv[e], vpv)) { okPM |= 1u << e; okNM |= 1u << pe; }
            } else {
                int ps = CF_OFF + li0 + e - 1;
                while (ps >= CF_OFF && !sA[ps]) ps--;
                bool hasPrev; int32_t cp = -1; float vp = 0.0f;
                if (ps >= CF_OFF || !hasLeft || sA[CF_OFF - 1]) { hasPrev = ps >= CF_OFF || hasLeft; if (hasPrev) { cp = sChr[ps]; vp = sCnt[ps]; } }
                else {                                    // the bin in front of the block fails the size filter too: keep looking in global memory
                    int64_t p = base - 2;
                    while (p >= 0 && (stop[p] - start[p]) > thresh) p--;
                    hasPrev = p >= 0; if (hasPrev) { cp = chr[p]; vp = count[p]; }
                }
                if (hasPrev) { hasPrevM |= 1u << e; if (cp == rc[e] && !sig_diff(rv[e], vp)) okPM |= 1u << e; }
            }
            pe = e; cpv = rc[e]; vpv = rv[e];
        }
        {   // the last of them looks to the right of the thread's bins
            int qs = CF_OFF + li0 + pe + 1;
            while (qs < CF_OFF + L && !sA[qs]) qs++;
            bool hasNext; int32_t cq = -1; float vq = 0.0f;
            if (qs < CF_OFF + L || !hasRight || sA[CF_OFF + L]) { hasNext = qs < CF_OFF + L || hasRight; if (hasNext) { cq = sChr[qs]; vq = sCnt[qs]; } }
            else {
                int64_t q = base + L + 1;
                while (q < n && (stop[q] - start[q]) > thresh) q++;
                hasNext = q < n; if (hasNext) { cq = chr[q]; vq = count[q]; }
            }
            if (hasNext) { hasNextM |= 1u << pe; if (cq == cpv && !sig_diff(vpv, vq)) okNM |= 1u << pe; }
        }
        keepM = aM & (okPM | okNM | ~(hasPrevM | hasNextM));
    }
    const uint32_t nKeepT = (uint32_t)__builtin_popcount(keepM);
    {
        unsigned long long f8 = 0;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const uint32_t k = (keepM >> e) & 1u;
            f8 |= (unsigned long long)k << (8 * e);
            if (k) {                                          // out-of-range input is reported through D->bad and nothing is returned: clamped here so that no table is overrun
                const int32_t cc = (uint32_t)rc[e] >= (uint32_t)nchr ? 0 : rc[e];
                const uint8_t au = cc < 256 ? sAuto[cc] : isAuto[cc];
                atomicAdd(&lh[(au ? 0 : NGC) + ((uint32_t)rg[e] > 100u ? 100 : rg[e])], 1u);
            }
        }
        if (li0 + 7 < L) *reinterpret_cast<gptr<unsigned long long>>(flags + base + li0) = f8;
        else for (int e = 0; e < 8; e++) if (li0 + e < L) flags[base + li0 + e] = (uint8_t)((f8 >> (8 * e)) & 1ull);
    }
    if (blockIdx.x == 0) CF_STAMP(2);
    uint32_t nKeep = wave_reduce_add_u32(nKeepT); nSize = wave_reduce_add_u32(nSize);
    if (lane_id() == 0) { sh[threadIdx.x >> 6] = nKeep; sh[4 + (threadIdx.x >> 6)] = nSize; }
    if (bad) D->bad = 1u;
    __syncthreads();
    if (threadIdx.x == 0) cf_st(&A.dBlk[blockIdx.x], (unsigned long long)(sh[0] + sh[1] + sh[2] + sh[3]) | ((unsigned long long)(sh[4] + sh[5] + sh[6] + sh[7]) << 32));
    if (threadIdx.x < 2 * NGC && lh[threadIdx.x]) atomicAdd(&A.repl[(blockIdx.x % CF_HREP) * (2 * NGC) + threadIdx.x], lh[threadIdx.x]);
    if (blockIdx.x == 0) CF_STAMP(3);
    const bool last = cf_arrive_last(A.tick + 1 * CF_TICK_WORDS, blockIdx.x, (uint32_t)A.nb, &sLast);
    if (blockIdx.x == 0) CF_STAMP(4);
    if (last) { CF_STAMP(5); cf_decide_gc(A); __syncthreads(); CF_STAMP(6); }
}
// sum of the counts of the blocks in front of this one (every workgroup of a scatter kernel does this for itself: the counts are a few KB in L2, and the stage has no
// scan kernel); lo32: the counts are the low halves of 64-bit records
template <bool LO32, class T>
__device__ __forceinline__ uint32_t cf_block_offset(const T* __restrict__ cnt, int nbefore, unsigned long long* sh16) {
    unsigned long long mine = 0;
    for (int j = threadIdx.x; j < nbefore; j += blockDim.x) mine += LO32 ? ((unsigned long long)cnt[j] & 0xFFFFFFFFull) : (unsigned long long)cnt[j];
    return (uint32_t)cf_block_sum_u64(mine, sh16);
}
// the compaction itself: caller's arrays -> scratch SoA, and — the GC strip is decided by then — the order-preserving keys of the autosomal survivors with a kept GC
// value, grouped by GC (order inside a bucket is irrelevant: only order statistics are taken).  All flags of the block are scanned first (one barrier), then every
// load of the survivors is in flight at once.
__global__ void __launch_bounds__(256) k_cf_scatter_ab(const CfArgs* __restrict__ AA) {
    CF_SAMPLE;
    __shared__ unsigned long long sh16[16];
    __shared__ uint32_t shW[CBLK / 256][4];
    __shared__ uint32_t lcnt[NGC], lbase[NGC];
    __shared__ uint8_t sKeepGc[NGC];
    const int64_t n = A.n;
    const int64_t base = (int64_t)blockIdx.x * CBLK;
    if (base >= n) return;
    const gptr<const uint8_t> flags = as_global(A.dFlags), isAuto = as_global(A.isAuto);
    const GSoa src = as_global(A.caller); const GSoa1 dst = as_global(A.S1); const int nchr = A.nchr;
    CleanDev* __restrict__ D = A.D;
    const bool group = D->gcActive != 0;
    if (threadIdx.x < NGC) { lcnt[threadIdx.x] = 0; sKeepGc[threadIdx.x] = D->keepGc[threadIdx.x]; }
    uint32_t f[CBLK / 256], inc[CBLK / 256];
#pragma unroll
    for (int j = 0; j < CBLK / 256; j++) { const int64_t i = base + j * 256 + threadIdx.x; f[j] = (i < n) ? flags[i] : 0; }
    // the five columns of EVERY bin of the block are requested now, whatever its flag (2 % of them are dropped): a load under `if (flag)` is waited for where the condition
    // ends, one element at a time, and the block offset below (a few thousand counts + two barriers) runs while these travel
    int32_t lc[CBLK / 256], lg[CBLK / 256], ls[CBLK / 256], le[CBLK / 256]; float lv[CBLK / 256];
    if (base + CBLK <= n) {
#pragma unroll
        for (int j = 0; j < CBLK / 256; j++) { const int64_t i = base + j * 256 + threadIdx.x; lc[j] = src.chr[i]; lg[j] = src.gc[i]; ls[j] = src.start[i]; le[j] = src.stop[i]; lv[j] = src.count[i]; }
    } else {
#pragma unroll
        for (int j = 0; j < CBLK / 256; j++) {
            const int64_t i = base + j * 256 + threadIdx.x;
            lc[j] = 0; lg[j] = 0; ls[j] = 0; le[j] = 0; lv[j] = 0.0f;
            if (i < n) { lc[j] = src.chr[i]; lg[j] = src.gc[i]; ls[j] = src.start[i]; le[j] = src.stop[i]; lv[j] = src.count[i]; }
        }
    }
    const uint32_t blockOff = cf_block_offset<true>(A.dBlk, (int)blockIdx.x, sh16);          // (its barriers also publish lcnt / sKeepGc)
#pragma unroll
    for (int j = 0; j < CBLK / 256; j++) { inc[j] = wave_inclusive_scan_u32(f[j]); if (lane_id() == 63) shW[j][threadIdx.x >> 6] = inc[j]; }
    __syncthreads();
    uint32_t running = blockOff;
    uint32_t myRank[CBLK / 256], myKey[CBLK / 256]; int myGc[CBLK / 256];
    const int w = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < CBLK / 256; j++) {
        uint32_t woff = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { const uint32_t v = shW[j][k]; if (k < w) woff += v; tot += v; }
        myGc[j] = -1;
        if (f[j]) {
            const uint32_t d = running + woff + inc[j] - 1;
            // out-of-range input is reported through D->bad (k_cf_flags_ab) and nothing is returned; the scratch copy holds clamped values so that no later kernel indexes past a table
            const int32_t c0 = lc[j], g0 = lg[j];
            const int32_t g = (uint32_t)g0 > 100u ? 100 : g0, c = (uint32_t)c0 >= (uint32_t)nchr ? 0 : c0;
            const float v = lv[j];
            dst.chr[d] = c; dst.start[d] = ls[j]; dst.stop[d] = le[j]; dst.gc[d] = (uint8_t)g; dst.count[d] = v;
            if (group && isAuto[c] && sKeepGc[g]) { myGc[j] = g; myKey[j] = key_of_float(v); myRank[j] = atomicAdd(&lcnt[g], 1u); }
        }
        running += tot;
    }
    if (!group) return;
    __syncthreads();
    if (threadIdx.x < NGC && lcnt[threadIdx.x]) lbase[threadIdx.x] = atomicAdd(&A.repl[CF_HREP * (2 * NGC) + (blockIdx.x % CF_HREP) * NGC + threadIdx.x], lcnt[threadIdx.x]);      // absolute position in keysG
    __syncthreads();
    const gptr<uint32_t> keysG = as_global(A.keysG);
#pragma unroll
    for (int j = 0; j < CBLK / 256; j++) if (myGc[j] >= 0) keysG[lbase[myGc[j]] + myRank[j]] = myKey[j];
}

// ---------------------------------------------------------------- k_cf_hist_lsd: counting sweep of the grouped keys | window SDs (CanvasClean.cs:243-298)
// Role 1, workgroups [0, nHist): one sweep of the grouped keys, counters per value in LDS, flushed into the bucket's row and the genome's.
// Role 2, workgroups [nHist, nHist + nLsd): one thread per window of 20 count differences (Utilities.StandardDeviation, CanvasClean.cs:262-298) + the chromosome boundaries
// among the window's bins (the run records that GetLocalStandardDeviationAverage's per-chromosome grouping needs); the last of these workgroups sorts the records and
// derives the runs of windows per chromosome exactly as local_sd_begin does on the host.  Neither role reads what the other writes.
__device__ __forceinline__ void cq_count_key(uint32_t key, long long lo, uint32_t* lw, uint32_t& below, uint32_t& bad) {
    long long k;
    if (!cq_key(float_of_key(key), k)) { bad = 1; return; }
    if (k < lo) below++;
    else if (k - lo < CQW) atomicAdd(&lw[k - lo], 1u);
}
__device__ __forceinline__ void cf_runs_build(const CfArgs& A, long long* s /* [CF_MAXRUN] */, long long* raw /* [CF_MAXRUN] */) {
    CleanDev* __restrict__ D = A.D; const long long* __restrict__ recs = A.dPos; int64_t* __restrict__ runStart = A.dRunStart;
    const unsigned long long nAB = D->nAB;
    const int have = A.wantLsd && nAB >= 50000ull;                         // CanvasClean.cs:483-486
    if (threadIdx.x == 0) D->haveLocalSd = have;
    if (!have) { if (threadIdx.x == 0) D->nruns = 0; return; }
    const unsigned int nb = cf_ld(&D->nRunRec);
    if (nb > CF_MAXRUN) { if (threadIdx.x == 0) { D->fallback = 1u; D->nruns = 0; } return; }
    const int t = threadIdx.x;
    raw[t] = t < (int)nb ? (long long)cf_ld(reinterpret_cast<const unsigned long long*>(recs + t)) : 0x7FFFFFFFFFFFFFFFll;
    __syncthreads();
    // rank sort: every record counts the records in front of it (independent broadcast reads of LDS)
    if (t < (int)nb) {
        const long long mine = raw[t];
        int rank = 0;
        for (int j = 0; j < (int)nb; j++) { const long long o = raw[j]; rank += (o < mine || (o == mine && j < t)) ? 1 : 0; }
        s[rank] = mine;
    }
    __syncthreads();
    if (t == 0) {
        const int64_t n = (int64_t)nAB, Dn = n - 1, nW = Dn >= 1 ? (Dn - 1) / 20 : 0;
        int nruns = 0; int32_t lastChr = -1; bool any = false;
        for (unsigned r = 0; r < nb; r++) {
            const int64_t pos = s[r] >> 20, posNext = r + 1 < nb ? (s[r + 1] >> 20) : n;
            const int32_t c = (int32_t)(s[r] & 0xFFFFF);
            const int64_t w0 = (pos + 19) / 20, w1 = min((posNext + 19) / 20, nW);
            if (w0 >= w1) continue;
            if (any && lastChr == c) continue;                             // adjacent windows with the same chromosome merge
            runStart[nruns++] = w0; lastChr = c; any = true;
        }
        if (nruns > 0) runStart[0] = 0;
        runStart[nruns] = nW;
        D->nruns = nruns;
    }
}
__global__ void __launch_bounds__(1024) k_cf_hist_lsd(const CfArgs* __restrict__ AA, int nHist, int nLsd) {
    CF_SAMPLE;
    __shared__ __attribute__((aligned(16))) uint32_t lw[CQW];       // counters of the sweep role; scratch of the window role's last workgroup (its first 16 KB)
    int* const sLast = reinterpret_cast<int*>(&lw[CQW - 1]);
    __shared__ uint32_t sBelowW;
    uint32_t* const sBelow = &sBelowW;
    if ((int)blockIdx.x < nHist) {
        CfCq* __restrict__ C = A.cq;
        if (blockIdx.x >= C->ntiles) return;
        const SelTile T = A.cqTiles[blockIdx.x];
        const long long lo = C->lo;
        for (int i = threadIdx.x; i < CQW / 4; i += 1024) reinterpret_cast<uint4*>(lw)[i] = make_uint4(0u, 0u, 0u, 0u);
        if (threadIdx.x == 0) *sBelow = 0;
        __syncthreads();
        const gptr<const uint32_t> keys = as_global(A.keysG);
        uint32_t below = 0, bad = 0;
        // the tile's 16-byte aligned middle with 16-byte loads (all of a thread's loads in flight), its unaligned ends key by key
        const int64_t a0 = min<int64_t>((T.begin + 3) & ~(int64_t)3, T.end), nv = (T.end - a0) / 4, a1 = a0 + 4 * nv;
        uint4 kk[CQ_TILE / 4096];
#pragma unroll
        for (int u = 0; u < CQ_TILE / 4096; u++) { const int64_t v = (int64_t)u * 1024 + threadIdx.x; kk[u] = v < nv ? gload_uint4(keys + a0 + 4 * v) : make_uint4(0u, 0u, 0u, 0u); }
#pragma unroll
        for (int u = 0; u < CQ_TILE / 4096; u++) {
            if ((int64_t)u * 1024 + threadIdx.x >= nv) break;
            cq_count_key(kk[u].x, lo, lw, below, bad); cq_count_key(kk[u].y, lo, lw, below, bad); cq_count_key(kk[u].z, lo, lw, below, bad); cq_count_key(kk[u].w, lo, lw, below, bad);
        }
        if (threadIdx.x < 8) {
            const int64_t i = threadIdx.x < 4 ? T.begin + threadIdx.x : a1 + (threadIdx.x - 4);
            if (threadIdx.x < 4 ? i < a0 : i < T.end) cq_count_key(keys[i], lo, lw, below, bad);
        }
        // keys under the window: summed over the workgroup first (LDS), ONE pair of device atomics per workgroup — a wave of 2 048 keys nearly always holds one, and
        // 16 x ntiles atomics on the genome's word are performed one after the other at the memory side
        below = wave_reduce_add_u32(below);
        if ((threadIdx.x & 63) == 0 && below) atomicAdd(sBelow, below);
        if (bad) C->bad = 1u;
        __syncthreads();
        if (threadIdx.x == 0 && *sBelow) { atomicAdd(&C->below[T.seg], *sBelow); atomicAdd(&C->below[NGC], *sBelow); }
        uint32_t* __restrict__ row = A.cqHist + (size_t)T.seg * CQW; uint32_t* __restrict__ all = A.cqHist + (size_t)NGC * CQW;
        for (int i = threadIdx.x; i < CQW; i += 1024) { const uint32_t v = lw[i]; if (v) { atomicAdd(&row[i], v); atomicAdd(&all[i], v); } }
        return;
    }
    // ---- window role
    if (!A.wantLsd) return;
    const int64_t w = (int64_t)((int)blockIdx.x - nHist) * 1024 + threadIdx.x;
    const int64_t nAB = (int64_t)A.D->nAB, Dn = nAB - 1, nW = Dn >= 1 ? (Dn - 1) / 20 : 0;
    if (w <= nW) {
        const int64_t lo = w * 20, hi = w < nW ? lo + 20 : nAB;         // thread nW takes the tail (bins behind the last window: CountDeviation stays -1, GenomicBin.cs:83)
        const GSoa1 S1 = as_global(A.S1);
        if (w < nW) {
            // Utilities.StandardDeviation(double[], start, end) (Utilities.cs:246-262): sequential double arithmetic.  The window's 20 counts and chromosomes are five
            // 16-byte loads each (lo is a multiple of 20 and the scratch arrays are 256-byte aligned), all in flight together; the bin in front of the window and
            // the one behind it are two more
            const gptr<const float> count = S1.count;
            float x[21];
#pragma unroll
            for (int q = 0; q < 5; q++) { const uint4 v4 = gload_uint4(count + lo + 4 * q); x[4 * q] = __uint_as_float(v4.x); x[4 * q + 1] = __uint_as_float(v4.y); x[4 * q + 2] = __uint_as_float(v4.z); x[4 * q + 3] = __uint_as_float(v4.w); }
            x[20] = count[lo + 20];
            int32_t cw[21];
            cw[0] = lo > 0 ? S1.chr[lo - 1] : -1;
#pragma unroll
            for (int q = 0; q < 5; q++) { const uint4 v4 = gload_uint4(S1.chr + lo + 4 * q); cw[4 * q + 1] = (int32_t)v4.x; cw[4 * q + 2] = (int32_t)v4.y; cw[4 * q + 3] = (int32_t)v4.z; cw[4 * q + 4] = (int32_t)v4.w; }
#pragma unroll
            for (int k = 0; k < 20; k++) {
                if (lo + k == 0 || cw[k + 1] != cw[k]) {
                    const unsigned int at = atomicAdd(&A.D->nRunRec, 1u);
                    if (at < 65536u) cf_st(reinterpret_cast<unsigned long long*>(A.dPos + at), (unsigned long long)(((lo + k) << 20) | (long long)(cw[k + 1] & 0xFFFFF)));
                }
            }
            double d[20];
#pragma unroll
            for (int k = 0; k < 20; k++) d[k] = (double)(x[k + 1] - x[k]);
            double sum = 0;
#pragma unroll
            for (int k = 0; k < 20; k++) sum += d[k];
            const double mu = sum / 20;
            double s2 = 0;
#pragma unroll
            for (int k = 0; k < 20; k++) { const double df = d[k] - mu; s2 += df * df; }
            as_global(A.dSd)[w] = sqrt(s2 / 19);
        }
        const gptr<const int32_t> chr = S1.chr;
        int32_t prevC = lo > 0 ? chr[lo - 1] : -1;
        if (w == nW) for (int64_t i = lo; i < hi; i++) {                  // the tail behind the last window (the windows did their own bins above)
            const int32_t c = chr[i];
            if (i == 0 || c != prevC) { const unsigned int k = atomicAdd(&A.D->nRunRec, 1u); if (k < 65536u) cf_st(reinterpret_cast<unsigned long long*>(A.dPos + k), (unsigned long long)((i << 20) | (long long)(c & 0xFFFFF))); }
            prevC = c;
        }
    }
    if (cf_arrive_last(A.tick + 2 * CF_TICK_WORDS, blockIdx.x - (uint32_t)nHist, (uint32_t)nLsd, sLast)) cf_runs_build(A, reinterpret_cast<long long*>(lw), reinterpret_cast<long long*>(lw) + CF_MAXRUN);
}

// ---------------------------------------------------------------- k_cf_pick_mad: order statistics from the counters | median and MAD of the window SDs per chromosome run
// Exact order statistics r0 <= r1 of up to MAD_REG x 1024 values held in registers (or streamed from memory), inside one 1024-thread workgroup: MSB radix passes that stop
// as soon as both ranks have at most MAD_CAND candidates left, which are then ranked against each other in LDS.  The first digit is sign + exponent (12 bits): the window SDs
// of a chromosome — and their absolute deviations — sit in a handful of binades, so that pass separates them whatever outliers (an all-zero window, a copy-number edge)
// the run holds, and because its digits are that concentrated a wave first adds them up with ballots (one LDS atomic per distinct digit and wave); the second digit
// (8 mantissa bits) is spread and usually ends the search: two counting sweeps + one collecting sweep per select.
// (Round 2 ran eight full 8-bit passes per select with every thread adding to one or two LDS counters in the exponent passes: 89 us for chr1.)
#define MAD_REG 20
#define MAD_CAND 256
#define MAD_BITS0 12
struct MadShared { uint32_t sH[2][1 << MAD_BITS0]; unsigned long long sPre[2], sK[2], sCand[2][MAD_CAND]; uint32_t sCnt[2], sNc[2], sScan[2][16]; };
template <class KeyFn, class Pass0Fn>
__device__ __forceinline__ void wg_select2_fast(KeyFn forEachKey /* (callback(key)) */, Pass0Fn pass0 /* (hist, shift) -> true if it counted the first digit itself */, int64_t cnt, unsigned long long rank0, unsigned long long rank1, MadShared& S, unsigned long long* out /* [2], LDS */, unsigned long long* dbg = nullptr) {
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    __syncthreads();
    if (tid == 0) { S.sPre[0] = 0; S.sPre[1] = 0; S.sK[0] = rank0; S.sK[1] = rank1; S.sCnt[0] = S.sCnt[1] = (uint32_t)min<int64_t>(cnt, 0xFFFFFFFFll); }
    __syncthreads();
    int done = 0;                                                                      // leading bits of the answers that are known
    unsigned long long p0 = 0, p1 = 0;
    for (int pass = 0; done < 64; pass++) {
        p0 = S.sPre[0]; p1 = S.sPre[1];
        const unsigned long long k0 = S.sK[0], k1 = S.sK[1];
        if (S.sCnt[0] <= MAD_CAND && S.sCnt[1] <= MAD_CAND) break;
        const int bits = pass == 0 ? MAD_BITS0 : min(10, 64 - done), shift = 64 - done - bits, nb = 1 << bits;
        const bool same = p0 == p1;
        for (int i = tid; i < 2 * nb; i += 1024) S.sH[i >= nb ? 1 : 0][i & (nb - 1)] = 0;
        __syncthreads();
        if (pass == 0 && pass0(S.sH[0], shift)) {
        } else if (pass == 0) {
            forEachKey([&](unsigned long long key) {
                const uint32_t dgt = (uint32_t)(key >> shift);
                unsigned long long todo = __ballot(1);
#pragma unroll 1
                for (int it = 0; it < 6 && todo; it++) {
                    const int leader = __builtin_ctzll(todo);
                    const uint32_t dl = (uint32_t)__builtin_amdgcn_readlane((int)dgt, leader);
                    const unsigned long long eq = __ballot(dgt == dl) & todo;
                    if (l == leader) atomicAdd(&S.sH[0][dl], (uint32_t)__builtin_popcountll(eq));
                    todo &= ~eq;
                }
                if ((todo >> l) & 1ull) atomicAdd(&S.sH[0][dgt], 1u);
            });
        } else {
            forEachKey([&](unsigned long long key) {
                const uint32_t dgt = (uint32_t)(key >> shift) & (uint32_t)(nb - 1);
                const unsigned long long hiPart = key >> (64 - done);
                if (hiPart == p0) atomicAdd(&S.sH[0][dgt], 1u);
                if (!same && hiPart == p1) atomicAdd(&S.sH[1][dgt], 1u);
            });
        }
        __syncthreads();
        // the digit that holds each rank: thread t sums bins 4t .. 4t + 3 (the rows are 4 * 1024 bins long at most), block scan, the thread whose stretch covers the rank narrows it
        uint32_t c[2][4], sum[2], inc[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const uint32_t* h = S.sH[same ? 0 : q];
#pragma unroll
            for (int e = 0; e < 4; e++) c[q][e] = 4 * tid + e < nb ? h[4 * tid + e] : 0u;
            sum[q] = c[q][0] + c[q][1] + c[q][2] + c[q][3];
            inc[q] = wave_inclusive_scan_u32(sum[q]);
            if (l == 63) S.sScan[q][w] = inc[q];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; q++) {
            uint32_t woff = 0;
            for (int ww = 0; ww < w; ww++) woff += S.sScan[q][ww];
            const uint32_t ex = woff + inc[q] - sum[q];
            const unsigned long long k = q == 0 ? k0 : k1;
            if (k >= ex && k < (unsigned long long)ex + sum[q]) {
                uint32_t r = (uint32_t)(k - ex), dgt, cc;
                if (r < c[q][0]) { dgt = 0; cc = c[q][0]; } else if (r < c[q][0] + c[q][1]) { dgt = 1; r -= c[q][0]; cc = c[q][1]; }
                else if (r < c[q][0] + c[q][1] + c[q][2]) { dgt = 2; r -= c[q][0] + c[q][1]; cc = c[q][2]; } else { dgt = 3; r -= c[q][0] + c[q][1] + c[q][2]; cc = c[q][3]; }
                S.sPre[q] = ((q == 0 ? p0 : p1) << bits) | (unsigned long long)(4 * tid + dgt);
                S.sK[q] = r; S.sCnt[q] = cc;
            }
        }
        done += bits;
        __syncthreads();
        if (dbg && tid == 0) dbg[pass] = wall_clock64();
    }
    if (done == 64) { if (tid == 0) { out[0] = S.sPre[0]; out[1] = S.sPre[1]; } return; }      // every bit resolved (ties heavier than MAD_CAND): the prefix is the key
    // the candidates of both ranks, ranked against each other
    if (dbg && tid == 0) dbg[8] = wall_clock64();
    if (tid == 0) { S.sNc[0] = 0; S.sNc[1] = 0; }
    __syncthreads();
    const bool same = p0 == p1;
    forEachKey([&](unsigned long long key) {
        const unsigned long long hiPart = done == 0 ? 0ull : key >> (64 - done);
        if (hiPart == p0) { const uint32_t at = atomicAdd(&S.sNc[0], 1u); if (at < MAD_CAND) S.sCand[0][at] = key; }
        if (!same && hiPart == p1) { const uint32_t at = atomicAdd(&S.sNc[1], 1u); if (at < MAD_CAND) S.sCand[1][at] = key; }
    });
    __syncthreads();
    if (dbg && tid == 0) dbg[9] = wall_clock64();
    for (int q = 0; q < (same ? 1 : 2); q++) {
        const uint32_t nc = min(S.sNc[q], (uint32_t)MAD_CAND);
        if ((uint32_t)tid < nc) {
            const unsigned long long mine = S.sCand[q][tid];
            uint32_t less = 0;
#pragma unroll 8
            for (uint32_t j = 0; j < nc; j++) { const unsigned long long ok = S.sCand[q][j]; less += (ok < mine || (ok == mine && j < (uint32_t)tid)) ? 1u : 0u; }
            if ((unsigned long long)less == S.sK[q]) out[q] = mine;
            if (same && (unsigned long long)less == S.sK[1]) out[1] = mine;
        }
    }
}
// median and MAD of the window SDs of chromosome run r (CanvasClean.cs:243-258, Utilities.cs Median / Mad)
template <bool REG>
__device__ __forceinline__ void cf_run_mad(const gptr<const double> sd, int64_t lo, int64_t hi, MadShared& S, unsigned long long* sOut /* LDS [2] */, double* outMad, unsigned long long* dbg) {
    const int64_t cnt = hi - lo;
    const unsigned long long r1 = (unsigned long long)(cnt / 2), r0 = (cnt % 2) ? r1 : r1 - 1;
    // register flavour: thread t holds the KEYS of values lo + t, lo + t + 1024, ...  Only the keys are kept — the value is double_of_key(key) — and the second select rewrites them in place: with the values AND both key sets live
    // (the compiler hoists the key computations out of the sweeps) the role needed 144 registers, spilled, and every sweep waited for scratch loads.
    unsigned long long key[MAD_REG];
    const int64_t mine0 = lo + threadIdx.x;                                           // values mine0 + k * 1024 (coalesced; the order of the values is irrelevant)
    const int nMine = REG ? (int)max<int64_t>(0, (hi - mine0 + 1023) / 1024) : 0;
    if (REG) {
#pragma unroll
        for (int k = 0; k < MAD_REG; k++) key[k] = k < nMine ? key_of_double(sd[mine0 + (int64_t)k * 1024]) : 0ull;
    }
    double median = 0.0; bool second = false;
    auto sweep = [&](auto&& body) {
        if (REG) {
#pragma unroll
            for (int k = 0; k < MAD_REG; k++) { if (k < nMine) body(key[k]); __builtin_amdgcn_sched_barrier(0); }      // one value at a time: interleaved, the unrolled bodies spill
        } else if (!second) { for (int64_t i = lo + threadIdx.x; i < hi; i += 1024) body(key_of_double(sd[i])); }
        else for (int64_t i = lo + threadIdx.x; i < hi; i += 1024) body(key_of_double(fabs(sd[i] - median)));
    };
    // first digit (sign + exponent) of the register flavour: a thread counts the values that share the digit of its first one, a wave adds those counts up for the
    // lanes that agree with its first lane, and only what is left goes to the LDS counters one by one (the digits of a run are a handful of binades: 20 atomics per thread
    // on two or three LDS words were 12 us per select)
    auto pass0 = [&](uint32_t* h, int shift) -> bool {
        if (!REG) return false;
        const int l = threadIdx.x & 63;
        // the thread's two most likely digits (the first value's, and the first one that differs from it) are counted in registers
        const uint32_t d0 = (uint32_t)(key[0] >> shift);
        uint32_t d1 = d0, c0 = 0, c1 = 0;
#pragma unroll
        for (int k = 0; k < MAD_REG; k++) {
            if (k < nMine) {
                const uint32_t d = (uint32_t)(key[k] >> shift);
                if (d == d0) c0++;
                else { if (d1 == d0) d1 = d; if (d == d1) c1++; else atomicAdd(&h[d], 1u); }
            }
        }
        // ... and added up over the wave for the two digits its first active lane holds
        const bool active = nMine > 0;
        const unsigned long long todo = __ballot(active);
        if (todo) {
            const int leader = __builtin_ctzll(todo);
            const uint32_t da = (uint32_t)__builtin_amdgcn_readlane((int)d0, leader), db = (uint32_t)__builtin_amdgcn_readlane((int)d1, leader);
            uint32_t ca = 0, cb = 0;
            if (active) {
                if (d0 == da) { ca += c0; c0 = 0; } else if (d0 == db) { cb += c0; c0 = 0; }
                if (c1) { if (d1 == da) { ca += c1; c1 = 0; } else if (d1 == db && db != da) { cb += c1; c1 = 0; } }
            }
            ca = wave_reduce_add_u32(ca); cb = wave_reduce_add_u32(cb);
            if (l == leader) { atomicAdd(&h[da], ca); if (cb) atomicAdd(&h[db], cb); }
            if (active && c0) atomicAdd(&h[d0], c0);
            if (active && c1) atomicAdd(&h[d1], c1);
        }
        return true;
    };
    if (dbg && threadIdx.x == 0) dbg[31] = wall_clock64();
    wg_select2_fast(sweep, pass0, cnt, r0, r1, S, sOut, dbg ? dbg + 40 : nullptr);
    __syncthreads();
    if (dbg && threadIdx.x == 0) dbg[32] = wall_clock64();
    median = (cnt % 2) ? double_of_key(sOut[1]) : (double_of_key(sOut[0]) + double_of_key(sOut[1])) / 2;
    second = true;
    if (REG) {
#pragma unroll
        for (int k = 0; k < MAD_REG; k++) key[k] = key_of_double(fabs(double_of_key(key[k]) - median));
    }
    __syncthreads();
    wg_select2_fast(sweep, pass0, cnt, r0, r1, S, sOut);
    __syncthreads();
    if (dbg && threadIdx.x == 0) dbg[33] = wall_clock64();
    if (threadIdx.x == 0) cf_st_f64(outMad, (cnt % 2) ? double_of_key(sOut[1]) : (double_of_key(sOut[0]) + double_of_key(sOut[1])) / 2);
    __syncthreads();
}
// NormalizeByGC as a function of k for bucket g (what k_cf_xform_gc does to a key)
__device__ __forceinline__ float cq_normalised(long long k, double median, double globalMedian) {
    const float x = cq_value(k);
    return median > 0 ? (float)(globalMedian * (double)x / median) : x;
}
// The genome's quartiles of the normalised counts (CanvasClean.cs:34-66) by counting once more.  The items are the counters: value = the normalised count of the slot,
// weight = the counter.  The normalised values are not on a grid, but v -> floor((v - L0) * 100) is non-decreasing, so one weighted count over CQW bins of 0.01 around the
// genome's median locates, for every rank, the bin that holds it and the rank inside that bin; the bin's few candidates — per bucket the slots whose value
// falls into it, found by bisection because the value is monotone in the slot — are then ordered exactly by their float keys.  Keys below / above a
// bucket's window count as smaller / larger than everything, which the decision checks against the answers.
__device__ __forceinline__ double cq_nbin_origin(double globalMedian) { return globalMedian - (double)CQW / 200.0; }
__device__ __forceinline__ long long cq_nbin(float v, double origin) { return (long long)floor(((double)v - origin) * 100.0); }
#define CQ_NCAND 1024        // candidates of one bin (a bucket contributes about median / globalMedian slots per bin)
// ranks[0 .. nr) of a row of CQW counters held 16 consecutive per thread -> sK[q] = the slot that holds rank q.  False (in every thread) when a rank lies outside the window.
__device__ __forceinline__ bool cq_pick_ranks(const uint32_t (&c)[16], const long long* ranks /* LDS */, int nr, int64_t below, uint32_t* swave, long long* sK, int* sFail, uint32_t* inWinOut) {
    const int t = threadIdx.x;
    uint32_t s = 0;
#pragma unroll
    for (int u = 0; u < 16; u++) s += c[u];
    const uint32_t inc = wave_inclusive_scan_u32(s);
    __syncthreads();
    if ((t & 63) == 63) swave[t >> 6] = inc;
    if (t == 0) *sFail = 0;
    __syncthreads();
    uint32_t woff = 0, inWin = 0;
    for (int w = 0; w < 16; w++) { if (w < (t >> 6)) woff += swave[w]; inWin += swave[w]; }
    const uint32_t ex = woff + inc - s;
    for (int q = 0; q < nr; q++) {
        const int64_t r = ranks[q] - below;
        if (r < 0 || r >= (int64_t)inWin) { if (t == 0) *sFail = 1; continue; }
        if (r >= (int64_t)ex && r < (int64_t)ex + s) {
            uint32_t left = (uint32_t)(r - ex); int u = 0; bool found = false;
#pragma unroll
            for (int k = 0; k < 16; k++) { if (!found) { if (left < c[k]) { u = k; found = true; } else left -= c[k]; } }       // (no dynamic index into the register array)
            sK[q] = 16 * t + u;
        }
    }
    __syncthreads();
    *inWinOut = inWin;
    return *sFail == 0;
}
__device__ __forceinline__ void cq_load_row(const uint32_t* __restrict__ rowBase, uint32_t (&c)[16]) {
    const uint4* __restrict__ row = reinterpret_cast<const uint4*>(rowBase) + 4 * threadIdx.x;     // 16 consecutive counters per thread
#pragma unroll
    for (int u = 0; u < 4; u++) { const uint4 v = row[u]; c[4 * u] = v.x; c[4 * u + 1] = v.y; c[4 * u + 2] = v.z; c[4 * u + 3] = v.w; }
}
// the last workgroup of the pick role: the genome's quartiles of the normalised counts from the weighted count, then the NormalizeVarianceByGC decision (CanvasClean.cs:34-83)
__device__ __forceinline__ void cf_resolve_var(const CfArgs& A, uint32_t* lw, uint32_t* swave) {
    __shared__ int sBin[6], sFailR[6], sig, sFailD;
    __shared__ uint32_t sR[6], sKeyOut[6];
    __shared__ unsigned int sN[6];
    CfSel* __restrict__ P = A.P + 2; CleanDev* __restrict__ D = A.D; const CfCq* __restrict__ C = A.cq;
    const int t = threadIdx.x;
    const int nq = min((int)P->hdr[1], 6);
    // ---- the bin and the rank inside it, for every rank: one scan of the weighted count
    uint32_t c[16];
    {
        const unsigned long long* __restrict__ row = reinterpret_cast<const unsigned long long*>(A.cqHist + (size_t)(NGC + 1) * CQW) + 8 * t;
#pragma unroll
        for (int u = 0; u < 8; u++) { const unsigned long long q = cf_ld(row + u); c[2 * u] = (uint32_t)q; c[2 * u + 1] = (uint32_t)(q >> 32); }
    }
    uint32_t s = 0;
#pragma unroll
    for (int u = 0; u < 16; u++) s += c[u];
    const uint32_t inc = wave_inclusive_scan_u32(s);
    __syncthreads();
    if ((t & 63) == 63) swave[t >> 6] = inc;
    if (t < 6) { sBin[t] = -1; sN[t] = 0; sFailR[t] = 0; sKeyOut[t] = 0u; }
    if (t == 0) { sig = 0; sFailD = 0; }
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < (t >> 6); w++) woff += swave[w];
    const uint32_t ex = woff + inc - s;
    const long long nbelow = (long long)cf_ld(&C->nbelow);
    for (int q = 0; q < nq; q++) {
        const long long r = (long long)P->qk[q] - nbelow;
        if (r >= (long long)ex && r < (long long)ex + s) {
            uint32_t left = (uint32_t)(r - ex); int u = 0; bool found = false;
#pragma unroll
            for (int k = 0; k < 16; k++) { if (!found) { if (left < c[k]) { u = k; found = true; } else left -= c[k]; } }
            sBin[q] = 16 * t + u; sR[q] = left;
        }
    }
    __syncthreads();
    CF_STAMP(21);
    // ---- the candidates of every bin: per bucket the slots whose normalised value falls into it (keys at lw[q * 2048 + i], weights at lw[q * 2048 + 1024 + i]);
    // one thread per (rank, bucket)
    const double globalMedian = cf_ld_f64(&D->globalMedian);
    const long long lo = C->lo;
    if (t < 6 * NGC) {
        const int q = t / NGC, bkt = t - q * NGC;
        const int64_t cntB = (int64_t)D->segOff[bkt + 1] - (int64_t)D->segOff[bkt];
        const int bin = q < nq ? sBin[q] : -1;
        if (cntB > 0 && bin >= 0) {
            const double medianB = cf_ld_f64(&D->medians[bkt]), origin = cq_nbin_origin(globalMedian);
            int a = 0, b = CQW;                           // first slot whose value falls into bin `bin` or a later one
            while (a < b) { const int mid = (a + b) >> 1; if (cq_nbin(cq_normalised(lo + mid, medianB, globalMedian), origin) < (long long)bin) a = mid + 1; else b = mid; }
            for (int j = a; j < CQW; j++) {
                const float v = cq_normalised(lo + j, medianB, globalMedian);
                if (cq_nbin(v, origin) != (long long)bin) break;
                const uint32_t w = A.cqHist[(size_t)bkt * CQW + j];
                if (w) { const unsigned int at = atomicAdd(&sN[q], 1u); if (at < CQ_NCAND) { lw[q * 2048 + at] = key_of_float(v); lw[q * 2048 + 1024 + at] = w; } else sFailR[q] = 1; }
            }
        }
    }
    __syncthreads();
    CF_STAMP(22);
    {   // the candidate whose weights [less, less + w) cover the rank inside the bin: 170 threads per rank
        const int q = t / 170, idx = t - q * 170;
        if (q < nq && sBin[q] >= 0 && !sFailR[q]) {     // (a rank outside the bins, or too many candidates: key 0 makes the decision give the sample up)
            const unsigned int n = sN[q];
            const uint32_t rq = sR[q];
            for (unsigned int i = idx; i < n; i += 170) {
                const uint32_t key = lw[q * 2048 + i];
                unsigned long long less = 0;
#pragma unroll 8
                for (unsigned int o = 0; o < n; o++) { const uint32_t ko = lw[q * 2048 + o]; if (ko < key || (ko == key && o < i)) less += lw[q * 2048 + 1024 + o]; }
                if ((unsigned long long)rq >= less && (unsigned long long)rq < less + lw[q * 2048 + 1024 + i]) sKeyOut[q] = key;
            }
        }
    }
    __syncthreads();
    CF_STAMP(23);
    if (t < nq) P->qprefix[t] = (unsigned long long)sKeyOut[t];
    // ---- NormalizeVarianceByGC decision from the buckets' k statistics and the genome's keys
    const int64_t total = (int64_t)D->segOff[NGC];
    const int64_t cntT = t < NGC ? (int64_t)D->segOff[t + 1] - (int64_t)D->segOff[t] : 0;
    const double medianT = (t < NGC && cntT > 0) ? cf_ld_f64(&D->medians[t]) : 0.0;
    const int gn = quart_n(total);
    float gv[6]; uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
    for (int k = 0; k < 6; k++) { gv[k] = 0.0f; if (k < gn) { const uint32_t key = sKeyOut[k]; gv[k] = float_of_key(key); kmin = min(kmin, key); kmax = max(kmax, key); } }
    float g1, g2, g3;
    quartiles_from_values(total, gv, g1, g2, g3);
    const float globalIQR = g3 - g1;
    if (t < NGC) {
        float liqr = -1.0f, med = -1.0f;
        if (cntT > 0) {
            const int qn = quart_n(cntT);
            float v[6];
#pragma unroll
            for (int k = 0; k < 6; k++) v[k] = k < qn ? cq_normalised((long long)cf_ld(&C->kq[t][k]), medianT, globalMedian) : 0.0f;
            float q1, q2, q3; quartiles_from_values(cntT, v, q1, q2, q3);
            med = q2; liqr = q3 - q1;
            // the genome's answers are right only if every key that was left out of this bucket's window lies on the side it was counted on
            const uint32_t below = C->below[t], above = (uint32_t)(cntT - (int64_t)below - (int64_t)cf_ld(&C->inWin[t]));
            if (below && key_of_float(cq_normalised(lo, medianT, globalMedian)) > kmin) atomicOr(&sFailD, 1);
            if (above && key_of_float(cq_normalised(lo + CQW - 1, medianT, globalMedian)) < kmax) atomicOr(&sFailD, 1);
        }
        D->tab.localIQR[t] = liqr; D->tab.med[t] = med;
        if (t >= 10 && t < 90 && globalIQR * 2.0f < liqr) atomicAdd(&sig, 1);
    }
    __syncthreads();
    if (t == 0) {
        if (kmin == 0u || kmax == 0xFFFFFFFFu || sFailD) { A.cq->fail = 1u; D->cqFail = 1; D->fallback = 1u; D->changed = 0; return; }
        D->tab.globalIQR = globalIQR; D->changed = sig > 0 ? 1 : 0;
    }
}
// run role of k_cf_pick_mad
__device__ __forceinline__ void cf_role_mad(const CfArgs* __restrict__ AA, uint32_t* lw, unsigned long long* sOut, int* sLastP, int nPick) {
    CF_SAMPLE;
    MadShared& S = *reinterpret_cast<MadShared*>(lw);
    CleanDev* __restrict__ D = A.D;
    const int t = threadIdx.x;
    // ---- run role: median and MAD of the window SDs of runs b, b + CF_MADB, ... (Utilities.Mad per chromosome, CanvasClean.cs:243-258); the last workgroup averages them
    if (!A.wantLsd) return;
    const int b = (int)blockIdx.x - nPick, nruns = D->nruns;
    const gptr<const double> sd = as_global(A.dSd);
    if (b == 0) CF_STAMP(30);
    for (int r = b; r < nruns; r += CF_MADB) {
        const int64_t lo = A.dRunStart[r], hi = A.dRunStart[r + 1], cnt = hi - lo;
        if (cnt <= 0) { if (t == 0) cf_st_f64(&A.dRunMad[r], 0.0); }
        else if (cnt <= (int64_t)MAD_REG * 1024) cf_run_mad<true>(sd, lo, hi, S, sOut, &A.dRunMad[r], r == 0 ? A.dbg : nullptr);
        else cf_run_mad<false>(sd, lo, hi, S, sOut, &A.dRunMad[r], r == 0 ? A.dbg : nullptr);
    }
    if (b == 0) CF_STAMP(35);
    if (!cf_arrive_last(A.tick + 4 * CF_TICK_WORDS, (uint32_t)b, (uint32_t)CF_MADB, sLastP) || t != 0) return;
    if (!D->haveLocalSd) { D->localSd = -1.0; return; }
    double s = 0;
    for (int r = 0; r < nruns; r++) s += cf_ld_f64(&A.dRunMad[r]);               // List<double>.Average(): sequential sum / count
    D->localSd = s / (double)nruns;
}
__global__ void __launch_bounds__(1024) k_cf_pick_mad(const CfArgs* __restrict__ AA, int nPick) {
    CF_SAMPLE;
    __shared__ __attribute__((aligned(16))) uint32_t lw[CQW];       // pick role: the weighted count, then the last workgroup's candidate lists; run role: MadShared
    MadShared& S = *reinterpret_cast<MadShared*>(lw);
    __shared__ uint32_t swave[16];
    __shared__ long long sK[8], sRanks[8];
    __shared__ unsigned long long sOut[2];
    __shared__ int sFail, sLast;
    CleanDev* __restrict__ D = A.D;
    const int t = threadIdx.x;
    if ((int)blockIdx.x < nPick) {
        // ---- workgroup g: the order statistics of bucket g (g == NGC: of the genome) from its counters — the NormalizeByGC medians (CanvasClean.cs:170-189) and, for the
        // variance normalisation, the bucket's quartile statistics as k (a bucket is scaled by one factor, so they keep their ranks) and its share of the weighted count
        CfCq* __restrict__ C = A.cq;
        if (!A.useCq || A.P[1].hdr[1] == 0) return;
        const int g = blockIdx.x;
        if (g == 0 && t == 0 && C->bad) { C->fail = 1u; D->cqFail = 1; D->fallback = 1u; }
        const bool var = D->varActive != 0;
        const int64_t total = (int64_t)D->segOff[NGC];
        const int64_t cnt = g < NGC ? (int64_t)D->segOff[g + 1] - (int64_t)D->segOff[g] : total;
        const long long lo = C->lo;
        bool ok = true; double globalMedian = 0.0;
        if (g == 41) CF_STAMP(10);
        if ((g == NGC || (var && cnt > 0)) && total > 0) {              // the genome's median (every workgroup that goes on to the weighted count takes it for itself)
            uint32_t cG[16]; cq_load_row(A.cqHist + (size_t)NGC * CQW, cG);
            const int nr = (total % 2) ? 1 : 2;
            __syncthreads();
            if (t == 0) { if (total % 2) sRanks[0] = total / 2; else { sRanks[0] = total / 2 - 1; sRanks[1] = total / 2; } }
            uint32_t inWin;
            ok = cq_pick_ranks(cG, sRanks, nr, (int64_t)C->below[NGC], swave, sK, &sFail, &inWin);          // (its first barrier publishes sRanks)
            if (ok) globalMedian = nr == 1 ? (double)cq_value(lo + sK[0]) : (double)median_from_two(cq_value(lo + sK[0]), cq_value(lo + sK[1]));
            if (g == NGC && t == 0) {
                cf_st(&C->inWin[NGC], inWin);
                if (!ok) { C->fail = 1u; D->cqFail = 1; D->fallback = 1u; } else cf_st_f64(&D->globalMedian, globalMedian);
            }
        }
        if (g == 41) CF_STAMP(11);
        if (g < NGC && cnt > 0) {
            uint32_t c[16]; cq_load_row(A.cqHist + (size_t)g * CQW, c);
            const int nMed = (cnt % 2) ? 1 : 2, nQ = var ? quart_n(cnt) : 0, nr = nMed + nQ;
            __syncthreads();
            if (t == 0) { if (cnt % 2) sRanks[0] = cnt / 2; else { sRanks[0] = cnt / 2 - 1; sRanks[1] = cnt / 2; } }
            if (t < nQ) sRanks[nMed + t] = quart_idx(cnt, t);
            uint32_t inWin;
            const unsigned long long belowG = (unsigned long long)C->below[g];
            const bool okOwn = cq_pick_ranks(c, sRanks, nr, (int64_t)belowG, swave, sK, &sFail, &inWin);
            double median = 0.0;
            if (okOwn) median = nMed == 1 ? (double)cq_value(lo + sK[0]) : (double)median_from_two(cq_value(lo + sK[0]), cq_value(lo + sK[1]));
            if (t == 0) {
                cf_st(&C->inWin[g], inWin);
                if (!okOwn) { C->fail = 1u; D->cqFail = 1; D->fallback = 1u; }
                else { cf_st_f64(&D->medians[g], median); for (int k = nMed; k < nr; k++) cf_st(&C->kq[g][k - nMed], (int32_t)(lo + sK[k])); }
            }
            if (g == 41) CF_STAMP(12);
            if (var && ok && okOwn) {
                // this bucket's share of the weighted count: value = the normalised count of a slot, weight = its counter
                for (int i = t; i < CQW / 4; i += 1024) reinterpret_cast<uint4*>(lw)[i] = make_uint4(0u, 0u, 0u, 0u);
                __syncthreads();
                const double origin = cq_nbin_origin(globalMedian);
                unsigned long long below = t == 0 ? belowG : 0ull;
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    const uint32_t w = c[u];
                    if (!w) continue;
                    const long long b = cq_nbin(cq_normalised(lo + 16 * t + u, median, globalMedian), origin);
                    if (b < 0) below += w; else if (b < CQW) atomicAdd(&lw[b], w);
                }
                below = wave_reduce_add_u64(below);
                if ((t & 63) == 0 && below) atomicAdd(&C->nbelow, below);
                __syncthreads();
                uint32_t* __restrict__ all = A.cqHist + (size_t)(NGC + 1) * CQW;
                for (int i = t; i < CQW; i += 1024) { const uint32_t v = lw[i]; if (v) atomicAdd(&all[i], v); }
            }
        }
        if (g == 41) CF_STAMP(13);
        if (!var) return;
        const bool lastP = cf_arrive_last(A.tick + 3 * CF_TICK_WORDS, blockIdx.x, (uint32_t)nPick, &sLast);
        if (g == 41) CF_STAMP(14);
        if (lastP) { CF_STAMP(20); cf_resolve_var(A, lw, swave); __syncthreads(); CF_STAMP(25); }
        return;
    }
    cf_role_mad(AA, lw, sOut, &sLast, nPick);
}

// ---------------------------------------------------------------- the radix-select flavour of the decisions, and the second phase (NormalizeVarianceByGC changed the counts)
// NormalizeByGC decision: genome median and per-GC medians from the selected keys (CanvasClean.cs:170-189)
__global__ void __launch_bounds__(128) k_cf_dec_e(const CfArgs* __restrict__ AA, int which) {
    CF_SAMPLE;
    const CfSel* __restrict__ P = A.P + which; CleanDev* __restrict__ D = A.D;
    if (P->hdr[1] == 0) return;
    const int t = threadIdx.x;
    auto med = [&](int slot, int64_t cnt) -> double {
        const int at = P->first[slot];
        if (cnt % 2) return (double)float_of_key((uint32_t)P->qprefix[at]);
        return (double)median_from_two(float_of_key((uint32_t)P->qprefix[at]), float_of_key((uint32_t)P->qprefix[at + 1]));
    };
    if (t < NGC) { const int64_t cnt = (int64_t)D->segOff[t + 1] - (int64_t)D->segOff[t]; D->medians[t] = (cnt > 0 && P->first[t] >= 0) ? med(t, cnt) : 0.0; }
    if (t == NGC) D->globalMedian = med(NGC, (int64_t)D->segOff[NGC]);
}
#define CF_EPT 4             // bins / keys per thread of the element-wise normalisation kernels (the per-GC tables are staged in LDS once per workgroup)
__global__ void __launch_bounds__(256) k_cf_apply_gc(const CfArgs* __restrict__ AA, int which) {
    CF_SAMPLE;
    __shared__ double sMed[NGC];
    const CleanDev* __restrict__ D = A.D;
    if (A.P[which].hdr[1] == 0) return;
    const int64_t n = (int64_t)D->nAB, base = (int64_t)blockIdx.x * (256 * CF_EPT);
    if (base >= n) return;
    if (threadIdx.x < NGC) sMed[threadIdx.x] = D->medians[threadIdx.x];
    __syncthreads();
    float* __restrict__ count = A.S1.count; const uint8_t* __restrict__ gc = A.S1.gc;
    const double globalMedian = D->globalMedian;
#pragma unroll
    for (int j = 0; j < CF_EPT; j++) {
        const int64_t i = base + j * 256 + threadIdx.x;
        if (i >= n) break;
        const double median = sMed[gc[i]];
        if (median > 0) count[i] = (float)(globalMedian * (double)count[i] / median);        // CanvasClean.cs:190-195
    }
}
// the same normalisation applied to the grouped keys (the order statistics of the next step are taken from the updated counts)
// bucket of grouped position p: the buckets are contiguous, so a workgroup's first position is located by bisection and the rest walk forward
__device__ __forceinline__ int cf_bucket_walk(const uint32_t* sSeg, uint32_t p, int b) {
    while (b < NGC - 1 && p >= sSeg[b + 1]) b++;
    return b;
}
__device__ __forceinline__ int cf_bucket_of(const uint32_t* segOff, uint32_t p) {
    int lo = 0, hi = NGC - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (segOff[mid] <= p) lo = mid; else hi = mid - 1; }
    return lo;
}
__global__ void __launch_bounds__(256) k_cf_xform_gc(const CfArgs* __restrict__ AA, int nextProblem) {
    CF_SAMPLE;
    __shared__ uint32_t sSeg[NGC + 1];
    __shared__ double sMed[NGC];
    const CleanDev* __restrict__ D = A.D;
    if (A.P[nextProblem].hdr[1] == 0) return;
    const uint32_t total = D->segOff[NGC], base = blockIdx.x * (256u * CF_EPT);
    if (base >= total) return;
    if (threadIdx.x <= NGC) sSeg[threadIdx.x] = D->segOff[threadIdx.x];
    if (threadIdx.x < NGC) sMed[threadIdx.x] = D->medians[threadIdx.x];
    __syncthreads();
    uint32_t* __restrict__ keysG = A.keysG;
    const double globalMedian = D->globalMedian;
    int b = cf_bucket_of(sSeg, base);
#pragma unroll
    for (int j = 0; j < CF_EPT; j++) {
        const uint32_t p = base + j * 256u + threadIdx.x;
        if (p >= total) break;
        b = cf_bucket_walk(sSeg, p, b);
        const double median = sMed[b];
        if (median > 0) keysG[p] = key_of_float((float)(globalMedian * (double)float_of_key(keysG[p]) / median));
    }
}
// NormalizeVarianceByGC decision (CanvasClean.cs:34-83)
__global__ void __launch_bounds__(128) k_cf_dec_f(const CfArgs* __restrict__ AA, int which) {
    CF_SAMPLE;
    __shared__ int sig;
    const CfSel* __restrict__ P = A.P + which; CleanDev* __restrict__ D = A.D;
    if (P->hdr[1] == 0) return;
    const int t = threadIdx.x;
    if (t == 0) sig = 0;
    __syncthreads();
    auto quart = [&](int slot, int64_t cnt, float& q1, float& q2, float& q3) {
        float v[6]; const QuartIdx qi = quartile_indices(cnt);
        for (int k = 0; k < qi.n; k++) v[k] = float_of_key((uint32_t)P->qprefix[P->first[slot] + k]);
        quartiles_from_values(cnt, v, q1, q2, q3);
    };
    float g1, g2, g3;
    quart(NGC, (int64_t)D->segOff[NGC], g1, g2, g3);
    const float globalIQR = g3 - g1;
    if (t < NGC) {
        const int64_t cnt = (int64_t)D->segOff[t + 1] - (int64_t)D->segOff[t];
        float liqr = -1.0f, med = -1.0f;
        if (cnt > 0) { float q1, q2, q3; quart(t, cnt, q1, q2, q3); med = q2; liqr = q3 - q1; }
        D->tab.localIQR[t] = liqr; D->tab.med[t] = med;
        if (t >= 10 && t < 90 && globalIQR * 2.0f < liqr) atomicAdd(&sig, 1);
    }
    __syncthreads();
    if (t == 0) { D->tab.globalIQR = globalIQR; D->changed = sig > 0 ? 1 : 0; }
}
__global__ void __launch_bounds__(256) k_cf_apply_var(const CfArgs* __restrict__ AA) {
    CF_SAMPLE;
    __shared__ float sIqr[NGC], sMedF[NGC];
    const CleanDev* __restrict__ D = A.D;
    if (!D->changed) return;
    const int64_t n = (int64_t)D->nAB, base = (int64_t)blockIdx.x * (256 * CF_EPT);
    if (base >= n) return;
    if (threadIdx.x < NGC) { sIqr[threadIdx.x] = D->tab.localIQR[threadIdx.x]; sMedF[threadIdx.x] = D->tab.med[threadIdx.x]; }
    __syncthreads();
    float* __restrict__ count = A.S1.count; const uint8_t* __restrict__ gc = A.S1.gc;
    const float globalIQR = D->tab.globalIQR;
#pragma unroll
    for (int j = 0; j < CF_EPT; j++) {
        const int64_t i = base + j * 256 + threadIdx.x;
        if (i >= n) break;
        const int g = gc[i];
        const float scaledLocalIqr = sIqr[g] * 0.8f;
        if (globalIQR >= scaledLocalIqr) continue;
        const float iqrRatio = scaledLocalIqr / globalIQR, m = sMedF[g];
        count[i] = m + (count[i] - m) / iqrRatio;                                            // CanvasClean.cs:84-94
    }
}
__global__ void __launch_bounds__(256) k_cf_xform_var(const CfArgs* __restrict__ AA) {
    CF_SAMPLE;
    __shared__ uint32_t sSeg[NGC + 1];
    __shared__ float sIqr[NGC], sMedF[NGC];
    const CleanDev* __restrict__ D = A.D;
    if (!D->changed) return;
    const uint32_t total = D->segOff[NGC], base = blockIdx.x * (256u * CF_EPT);
    if (base >= total) return;
    if (threadIdx.x <= NGC) sSeg[threadIdx.x] = D->segOff[threadIdx.x];
    if (threadIdx.x < NGC) { sIqr[threadIdx.x] = D->tab.localIQR[threadIdx.x]; sMedF[threadIdx.x] = D->tab.med[threadIdx.x]; }
    __syncthreads();
    uint32_t* __restrict__ keysG = A.keysG;
    const float globalIQR = D->tab.globalIQR;
    int b = cf_bucket_of(sSeg, base);
#pragma unroll
    for (int j = 0; j < CF_EPT; j++) {
        const uint32_t p = base + j * 256u + threadIdx.x;
        if (p >= total) break;
        b = cf_bucket_walk(sSeg, p, b);
        const float scaledLocalIqr = sIqr[b] * 0.8f;
        if (globalIQR >= scaledLocalIqr) continue;
        const float iqrRatio = scaledLocalIqr / globalIQR, m = sMedF[b];
        keysG[p] = key_of_float(m + (float_of_key(keysG[p]) - m) / iqrRatio);
    }
}

// ---------------------------------------------------------------- last compaction: GC strip (CanvasClean.cs:226-235) + RemoveBinsWithExtremeLocalSD (:308-322) -> caller's arrays
// CountDeviation of a bin = the SD of its window of 20 (CanvasClean.cs:268-298), -1 behind the last window (GenomicBin.cs:83): read from the window SDs, eight bins per thread
__global__ void __launch_bounds__(256) k_cf_flags_final(const CfArgs* __restrict__ AA) {
    CF_SAMPLE;
    __shared__ uint32_t sh[4];
    __shared__ uint8_t sKeepGc[NGC + 3];
    const CleanDev* __restrict__ D = A.D;
    const int64_t n = (int64_t)D->nAB;
    const int64_t base = (int64_t)blockIdx.x * CBLK;
    if (base >= n) return;
    const gptr<const uint8_t> gc = as_global(A.S1.gc); const gptr<const double> sd = as_global(A.dSd); const gptr<uint8_t> flags = as_global(A.dFlags);
    const bool sdFilter = D->haveLocalSd && D->localSd > 5.0;
    const int64_t Dn = n - 1, nW = Dn >= 1 ? (Dn - 1) / 20 : 0;
    if (threadIdx.x < NGC) sKeepGc[threadIdx.x] = D->keepGc[threadIdx.x];
    __syncthreads();
    const int64_t i0 = base + 8 * (int64_t)threadIdx.x;
    uint32_t c = 0;
    if (i0 + 7 < n) {
        const unsigned long long g8 = *reinterpret_cast<gptr<const unsigned long long>>(gc + i0);
        const int64_t w0 = i0 / 20, w1 = (i0 + 7) / 20;
        bool ex0 = false, ex1 = false;
        if (sdFilter) { ex0 = w0 < nW && sd[w0] > 20 * 2.0; ex1 = w1 == w0 ? ex0 : (w1 < nW && sd[w1] > 20 * 2.0); }
        unsigned long long out = 0;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const bool keep = sKeepGc[(g8 >> (8 * e)) & 0xFF] && !(((i0 + e) / 20 == w0) ? ex0 : ex1);
            out |= (unsigned long long)(keep ? 1 : 0) << (8 * e); c += keep;
        }
        *reinterpret_cast<gptr<unsigned long long>>(flags + i0) = out;
    } else {
        for (int e = 0; e < 8; e++) {
            const int64_t i = i0 + e;
            if (i >= n) break;
            const int64_t w = i / 20;
            const bool keep = sKeepGc[gc[i]] && !(sdFilter && w < nW && sd[w] > 20 * 2.0);
            flags[i] = keep; c += keep;
        }
    }
    c = wave_reduce_add_u32(c);
    if (lane_id() == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) A.dBlkF[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ void __launch_bounds__(256) k_cf_scatter_final(const CfArgs* __restrict__ AA, int secondPhase) {
    CF_SAMPLE;
    __shared__ unsigned long long sh16[16];
    __shared__ uint32_t shW[CBLK / 256][4];
    CleanDev* __restrict__ D = A.D;
    if (D->fallback || D->bad) return;                                        // the caller's arrays stay as they were
    if (D->changed && !secondPhase) return;                                   // the variance normalisation changed the counts: the host enqueues the second NormalizeByGC, then this kernel again
    const int64_t n = (int64_t)D->nAB;
    const int64_t base = (int64_t)blockIdx.x * CBLK;
    if (base >= n) return;
    const gptr<const uint8_t> flags = as_global(A.dFlags); const GSoa1 src = as_global(A.S1); const GSoa dst = as_global(A.caller);
    // first phase: NormalizeByGC has only been decided, not applied to the scratch counts — nothing between here and there reads them — so it is applied while
    // the survivors are copied out.  (When the second phase runs, clean_batch_finish applies it to the scratch counts first: NormalizeVarianceByGC works on normalised counts.)
    const bool normalise = !secondPhase && (A.flags & CANVAS_CLEAN_GCNORM) && A.P[1].hdr[1] != 0;
    __shared__ double sMed[NGC];
    if (normalise && threadIdx.x < NGC) sMed[threadIdx.x] = D->medians[threadIdx.x];
    const double globalMedian = D->globalMedian;
    uint32_t f[CBLK / 256], inc[CBLK / 256];
#pragma unroll
    for (int j = 0; j < CBLK / 256; j++) { const int64_t i = base + j * 256 + threadIdx.x; f[j] = (i < n) ? flags[i] : 0; }
    // (as in k_cf_scatter_ab: every bin's columns requested at once, in front of the block offset)
    int32_t lc[CBLK / 256], lg[CBLK / 256], ls[CBLK / 256], le[CBLK / 256]; float lv[CBLK / 256];
    if (base + CBLK <= n) {
#pragma unroll
        for (int j = 0; j < CBLK / 256; j++) { const int64_t i = base + j * 256 + threadIdx.x; lc[j] = src.chr[i]; lg[j] = src.gc[i]; ls[j] = src.start[i]; le[j] = src.stop[i]; lv[j] = src.count[i]; }
    } else {
#pragma unroll
        for (int j = 0; j < CBLK / 256; j++) {
            const int64_t i = base + j * 256 + threadIdx.x;
            lc[j] = 0; lg[j] = 0; ls[j] = 0; le[j] = 0; lv[j] = 0.0f;
            if (i < n) { lc[j] = src.chr[i]; lg[j] = src.gc[i]; ls[j] = src.start[i]; le[j] = src.stop[i]; lv[j] = src.count[i]; }
        }
    }
    const uint32_t blockOff = cf_block_offset<false>(A.dBlkF, (int)blockIdx.x, sh16);       // (its barriers also publish sMed)
#pragma unroll
    for (int j = 0; j < CBLK / 256; j++) { inc[j] = wave_inclusive_scan_u32(f[j]); if (lane_id() == 63) shW[j][threadIdx.x >> 6] = inc[j]; }
    __syncthreads();
    uint32_t running = blockOff;
    const int w = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < CBLK / 256; j++) {
        uint32_t woff = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { const uint32_t v = shW[j][k]; if (k < w) woff += v; tot += v; }
        if (f[j]) {
            const uint32_t d = running + woff + inc[j] - 1;
            const int32_t g = lg[j];
            float v = lv[j];
            if (normalise) { const double median = sMed[g]; if (median > 0) v = (float)(globalMedian * (double)v / median); }        // CanvasClean.cs:190-195, applied on the way out
            dst.chr[d] = lc[j]; dst.start[d] = ls[j]; dst.stop[d] = le[j]; dst.gc[d] = g; dst.count[d] = v;
        }
        running += tot;
    }
    if (base + CBLK >= n && threadIdx.x == 0) D->nFinal = running;            // the last block's end = the number of bins that are left
}

// ---------------------------------------------------------------- host side
struct CleanPending { int B; unsigned gxN, gxB, gxT; bool anyLsd, anyVar, anyGc, useCq; CfArgs* dArgs; CleanDev* dD; std::vector<CfArgs> h; };

static void cf_select_passes(canvas_ctx* ctx, const CfArgs* dArgs, int B, unsigned gxT, int which) {
    for (int shift = 24; shift >= 0; shift -= 8) {
        hipLaunchKernelGGL(k_cf_select_hist, dim3(gxT, B), dim3(256), 0, ctx->stream, dArgs, which, shift, shift == 24 ? 1 : 0);
        hipLaunchKernelGGL(k_cf_select_pick, dim3(CF_MAXQ, B), dim3(64), 0, ctx->stream, dArgs, which, shift == 24 ? 1 : 0);
    }
}
// the medians of NormalizeByGC on problem `which` (1: first time, 3: after the variance normalisation) with the radix selects; apply = scale the scratch counts now
// (second phase) instead of in the last compaction (first phase)
static void cf_gc_medians(canvas_ctx* ctx, const CfArgs* args, int B, unsigned gxN, unsigned gxT, int which, int gate, bool apply) {
    hipLaunchKernelGGL(k_cf_sel_setup, dim3(1, B), dim3(128), 0, ctx->stream, args, which, 0, gate);
    cf_select_passes(ctx, args, B, gxT, which);
    hipLaunchKernelGGL(k_cf_dec_e, dim3(1, B), dim3(128), 0, ctx->stream, args, which);
    if (apply) hipLaunchKernelGGL(k_cf_apply_gc, dim3((gxN + CF_EPT - 1) / CF_EPT, B), dim3(256), 0, ctx->stream, args, which);
}

// CANVAS_CLEAN_RADIX_SELECT=1 switches the counting selects off (test hook: both ways must agree)
static inline bool clean_counting_selects() { return cvx_hook("CANVAS_CLEAN_RADIX_SELECT") == nullptr; }
// Enqueues the whole stage for B samples on ctx->stream (no synchronisation): the CleanDev blocks arrive in ctx->pin.  clean_batch_finish waits for them.
static int32_t clean_batch_enqueue(canvas_ctx* ctx, int B, const int64_t* h_n, int32_t* const* d_chr, int32_t* const* d_start, int32_t* const* d_stop, float* const* d_count, int32_t* const* d_gc,
                                   int32_t nchr, const uint8_t* h_chr_is_autosome, uint32_t flags, int32_t min_bins_per_gc, bool useCq) {
    useCq = useCq && (flags & CANVAS_CLEAN_GCNORM);
    const size_t cqWords = useCq ? (size_t)B * CQ_ROWS * CQW : 0;
    WsSizer sz;
    sz.take<CfArgs>(B); sz.take<uint8_t>(nchr); sz.take<CleanDev>(B); sz.take<uint32_t>((size_t)B * CF_SZ_BINS); sz.take<uint32_t>((size_t)B * CF_HREP * 3 * NGC); sz.take<CfCq>(B); sz.take<uint32_t>((size_t)B * 8 * CF_TICK_WORDS); sz.take<uint32_t>(cqWords + 4);
    int64_t nMax = 0; bool anyLsd = false, anyVar = false;
    for (int s = 0; s < B; s++) {
        const int64_t n = h_n[s], nW0 = n / 20 + 2, nb = nblk(n, CBLK); const size_t tilesUpper = (size_t)(n / SEL_TILE + NGC + 1);
        nMax = std::max(nMax, n);
        sz.take<int32_t>(n); sz.take<int32_t>(n); sz.take<int32_t>(n); sz.take<float>(n); sz.take<uint8_t>(n + 16);
        sz.take<uint8_t>(n + 16); sz.take<unsigned long long>(nb + 2); sz.take<uint32_t>(nb + 2); sz.take<uint32_t>(n); sz.take<uint32_t>(n / 8 + 1024); sz.take<double>(nW0); sz.take<double>(CF_MAXRUN + 8);
        sz.take<int64_t>(CF_MAXRUN + 8); sz.take<long long>(65536); sz.take<CfSel>(CF_NPROB); sz.take<SelTile>(tilesUpper * CF_NPROB); sz.take<SelTile>((size_t)(n / CQ_TILE + NGC + 1)); sz.take<unsigned long long>(64);
    }
    int32_t rc = canvas_ws_reserve(ctx, sz.off + 8192 + (size_t)nchr * 8 + 1024); if (rc) return rc;      // (+ the chromosome-offset table cvx_clean_f2_offsets enqueues behind the stage)
    const size_t histPer = (size_t)CF_MAXQ * 1024 * SEL_REP, histBytes = histPer * (size_t)B;
    if (histBytes > ctx->sel_hist_bytes) {
        if (ctx->sel_hist) { CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); CANVAS_HIP_TRY(ctx, hipFree(ctx->sel_hist)); ctx->sel_hist = nullptr; ctx->sel_hist_bytes = 0; }
        CANVAS_HIP_TRY(ctx, hipMalloc(&ctx->sel_hist, histBytes)); ctx->sel_hist_bytes = histBytes;
        CANVAS_HIP_TRY(ctx, hipMemsetAsync(ctx->sel_hist, 0, ctx->sel_hist_bytes, ctx->stream));       // zero once: k_select_pick clears every row it has read
    }
    WsCarver ws(ctx->ws);
    // two adjacent groups: what the host sends (the argument table) and what starts as zero (k_cf_init)
    CfArgs* dArgs = ws.take<CfArgs>(B); uint8_t* dIsAuto = ws.take<uint8_t>(nchr);
    CleanDev* dD = ws.take<CleanDev>(B); uint32_t* dSz = ws.take<uint32_t>((size_t)B * CF_SZ_BINS); uint32_t* dRepl = ws.take<uint32_t>((size_t)B * CF_HREP * 3 * NGC); CfCq* dCq = ws.take<CfCq>(B);
    uint32_t* dTick = ws.take<uint32_t>((size_t)B * 8 * CF_TICK_WORDS); uint32_t* dCqHist = ws.take<uint32_t>(cqWords + 4);
    CleanPending pend; pend.useCq = useCq; pend.B = B; pend.dArgs = dArgs; pend.dD = dD; pend.h.resize(B);
    unsigned gxT = 1, gxTcq = 1;
    for (int s = 0; s < B; s++) {
        const int64_t n = h_n[s], nW0 = n / 20 + 2; const int nb = (int)nblk(n, CBLK); const unsigned tilesUpper = (unsigned)(n / SEL_TILE + NGC + 1);
        CfArgs& A = pend.h[s];
        memset(&A, 0, sizeof A);
        A.n = n; A.nb = nb; A.nchr = nchr; A.minBinsPerGc = min_bins_per_gc; A.flags = flags; A.tilesUpper = tilesUpper; A.useCq = useCq ? 1 : 0;
        A.wantLsd = ((flags & CANVAS_CLEAN_LOCALSD) && n >= 50000) ? 1 : 0;
        A.doSize = (flags & CANVAS_CLEAN_FILTSIZE) ? 1 : 0; A.doOutlier = (flags & CANVAS_CLEAN_OUTLIERS) ? 1 : 0;
        A.caller = Soa{d_chr[s], d_start[s], d_stop[s], d_gc[s], d_count[s], nullptr};
        A.S1.chr = ws.take<int32_t>(n); A.S1.start = ws.take<int32_t>(n); A.S1.stop = ws.take<int32_t>(n); A.S1.count = ws.take<float>(n); A.S1.gc = ws.take<uint8_t>(n + 16);
        A.dFlags = ws.take<uint8_t>(n + 16); A.dBlk = ws.take<unsigned long long>(nb + 2); A.dBlkF = ws.take<uint32_t>(nb + 2); A.keysG = ws.take<uint32_t>(n); A.szOverCap = (uint32_t)(n / 8 + 1024); A.szOver = ws.take<uint32_t>(A.szOverCap);
        A.dSd = ws.take<double>(nW0); A.dRunMad = ws.take<double>(CF_MAXRUN + 8); A.dRunStart = ws.take<int64_t>(CF_MAXRUN + 8); A.dPos = ws.take<long long>(65536);
        A.P = ws.take<CfSel>(CF_NPROB); A.tiles = ws.take<SelTile>((size_t)tilesUpper * CF_NPROB);
        A.cqTiles = ws.take<SelTile>((size_t)(n / CQ_TILE + NGC + 1)); A.cq = dCq + s; A.cqHist = dCqHist + (size_t)s * CQ_ROWS * CQW;
        gxTcq = std::max(gxTcq, (unsigned)(n / CQ_TILE + NGC + 1));
        A.isAuto = dIsAuto; A.repl = dRepl + (size_t)s * CF_HREP * 3 * NGC; A.szHist = dSz + (size_t)s * CF_SZ_BINS; A.D = dD + s; A.hist = (uint32_t*)((char*)ctx->sel_hist + histPer * (size_t)s);
        A.tick = dTick + (size_t)s * 8 * CF_TICK_WORDS;
        { unsigned long long* dbg = ws.take<unsigned long long>(64); A.dbg = cvx_hook("CANVAS_CLEAN_DEBUG_CLOCKS") ? dbg : nullptr; }
        gxT = std::max(gxT, tilesUpper);
        anyLsd = anyLsd || A.wantLsd; anyVar = anyVar || (A.wantLsd && n > 500000);
    }
    const unsigned gxN = (unsigned)nblk(nMax, 256), gxB = (unsigned)nblk(nMax, CBLK);
    pend.gxN = gxN; pend.gxB = gxB; pend.gxT = gxT; pend.anyLsd = anyLsd; pend.anyVar = anyVar; pend.anyGc = (flags & CANVAS_CLEAN_GCNORM) != 0;
    ProfScope psTotal(ctx, "clean_total");
    // ---- the zeros of the stage (CleanDev blocks, size counters, replica counters, CfCq blocks, tickets, value counters) and the argument table, in one launch
    {
        const size_t zeroBytes = ((size_t)((char*)(dCqHist + cqWords) - (char*)dD) + 15) & ~size_t(15);
        CfArgsPack pack;                                     // (the kernel argument is copied at launch)
        int npack = 0;
        if (B <= CF_BYVAL && nchr <= 256) {
            memset(&pack, 0, sizeof pack);
            memcpy(pack.a, pend.h.data(), (size_t)B * sizeof(CfArgs)); memcpy(pack.isAuto, h_chr_is_autosome, nchr);
            npack = B;
        } else {
            std::vector<char> up((size_t)((char*)(dIsAuto + nchr) - (char*)dArgs), 0);
            memcpy(up.data(), pend.h.data(), (size_t)B * sizeof(CfArgs)); memcpy(up.data() + ((char*)dIsAuto - (char*)dArgs), h_chr_is_autosome, nchr);
            rc = canvas_h2d_small(ctx, dArgs, up.data(), up.size()); if (rc) return rc;
        }
        const unsigned gz = (unsigned)std::min<size_t>(512, (zeroBytes / 16 + 1023) / 1024 + 1);
        hipLaunchKernelGGL(k_cf_init, dim3(gz), dim3(1024), 0, ctx->stream, dArgs, dIsAuto, pack, npack, nchr, (uint4*)dD, zeroBytes / 16);
    }
    // ---- RemoveBigBins threshold (CanvasClean.cs:328-348): the 98th percentile of the bin sizes from exact per-size counts, left on the device
    if (flags & CANVAS_CLEAN_FILTSIZE) hipLaunchKernelGGL(k_cf_size, dim3(CF_SZ_GRID, B), dim3(1024), 0, ctx->stream, dArgs);        // sizeOn stays 0 without the filter
    // ---- size filter + outlier filter: one compaction, caller -> S1; the GC strip decision and the set-up of the counting selects ride on the flag kernel's last workgroup
    hipLaunchKernelGGL(k_cf_flags_ab, dim3(gxB, B), dim3(256), 0, ctx->stream, dArgs);
    hipLaunchKernelGGL(k_cf_scatter_ab, dim3(gxB, B), dim3(256), 0, ctx->stream, dArgs);
    if (!useCq && (flags & CANVAS_CLEAN_GCNORM)) {
        // NormalizeByGC on the grouped keys with the radix selects
        cf_gc_medians(ctx, dArgs, B, gxN, gxT, 1, 0, false);
        if (anyVar) {
            // NormalizeVarianceByGC (CanvasClean.cs:512-519): quartiles of the normalised counts; if it changes anything, NormalizeByGC once more
            hipLaunchKernelGGL(k_cf_sel_setup, dim3(1, B), dim3(128), 0, ctx->stream, dArgs, 2, 1, 1);
            hipLaunchKernelGGL(k_cf_xform_gc, dim3((gxN + CF_EPT - 1) / CF_EPT, B), dim3(256), 0, ctx->stream, dArgs, 2);
            cf_select_passes(ctx, dArgs, B, gxT, 2);
            hipLaunchKernelGGL(k_cf_dec_f, dim3(1, B), dim3(128), 0, ctx->stream, dArgs, 2);
        }
    }
    // ---- counting sweep | window SDs and chromosome runs (CanvasClean.cs:243-298); then order statistics and the variance decision | per-run MADs and their average.
    // ... the variance normalisation rarely changes anything: the last compaction below is enqueued on the assumption that it does not; when it did, that compaction does
    // nothing for the sample and clean_batch_finish enqueues the variance scaling, the second NormalizeByGC and the compaction for it (one more synchronisation, in that case only)
    const int nHist = useCq ? (int)gxTcq : 0, nLsd = anyLsd ? (int)nblk(nMax / 20 + 2, 1024) : 0;
    const int nPick = useCq ? NGC + 1 : 0, nMad = anyLsd ? CF_MADB : 0;
    if (cvx_hook("CANVAS_CLEAN_SPLIT_ROLES")) {               // profiling hook: every role as a launch of its own (same kernels, same results)
        if (nHist) hipLaunchKernelGGL(k_cf_hist_lsd, dim3(nHist, B), dim3(1024), 0, ctx->stream, dArgs, nHist, 0);
        if (nLsd) hipLaunchKernelGGL(k_cf_hist_lsd, dim3(nLsd, B), dim3(1024), 0, ctx->stream, dArgs, 0, nLsd);
        if (nPick) hipLaunchKernelGGL(k_cf_pick_mad, dim3(nPick, B), dim3(1024), 0, ctx->stream, dArgs, nPick);
        if (nMad) hipLaunchKernelGGL(k_cf_pick_mad, dim3(nMad, B), dim3(1024), 0, ctx->stream, dArgs, 0);
    } else {
        if (nHist + nLsd > 0) hipLaunchKernelGGL(k_cf_hist_lsd, dim3(nHist + nLsd, B), dim3(1024), 0, ctx->stream, dArgs, nHist, nLsd);
        if (nPick + nMad > 0) hipLaunchKernelGGL(k_cf_pick_mad, dim3(nPick + nMad, B), dim3(1024), 0, ctx->stream, dArgs, nPick);
    }
    // ---- last compaction into the caller's arrays
    hipLaunchKernelGGL(k_cf_flags_final, dim3(gxB, B), dim3(256), 0, ctx->stream, dArgs);
    hipLaunchKernelGGL(k_cf_scatter_final, dim3(gxB, B), dim3(256), 0, ctx->stream, dArgs, 0);
    rc = canvas_pin_reserve(ctx, (size_t)B * sizeof(CleanDev)); if (rc) return rc;
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(ctx->pin, dD, (size_t)B * sizeof(CleanDev), hipMemcpyDeviceToHost, ctx->stream));
    ctx->clean_ws_end = (ws.off + 255) & ~size_t(255);
    ctx->clean_batch = std::make_shared<CleanPending>(std::move(pend));
    return CANVAS_OK;
}
// Waits for the batch.  handled[s] = false: the host-driven path has to take sample s over (nothing of it was modified).  Per-sample outputs as canvas_clean2.
static int32_t clean_batch_finish(canvas_ctx* ctx, double* h_local_sd_out, int64_t* h_n_out, int32_t* h_info, char* handled, bool* secondPhase = nullptr) {
    std::shared_ptr<CleanPending> pp = std::static_pointer_cast<CleanPending>(ctx->clean_batch);
    ctx->clean_batch.reset();
    if (!pp) CANVAS_FAIL(ctx, CANVAS_ERR_HIP, "canvas_clean: no pending batch");
    const CleanPending& q = *pp;
    const int B = q.B;
    for (int s = 0; s < B; s++) handled[s] = 0;
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    CANVAS_HIP_TRY(ctx, hipGetLastError());
    if (q.h[0].dbg) {                                                      // profiling hook: wall-clock stamps (100 MHz) of selected workgroups, relative to the first one
        unsigned long long st[64];
        CANVAS_HIP_TRY(ctx, hipMemcpy(st, q.h[0].dbg, sizeof st, hipMemcpyDeviceToHost));
        fprintf(stderr, "clean stamps (us):");
        for (int i = 0; i < 64; i++) if (st[i]) fprintf(stderr, " [%d]%.2f", i, (double)(long long)(st[i] - st[0]) / 100.0);
        fprintf(stderr, "\n");
    }
    bool again = false;
    for (int s = 0; s < B; s++) {
        const CleanDev& H = ((const CleanDev*)ctx->pin)[s];
        if (!(H.changed && !H.fallback && !H.bad)) continue;
        // NormalizeVarianceByGC changed the counts of this sample (CanvasClean.cs:512-519): scale them, NormalizeByGC again on the new counts, then the last compaction
        const CfArgs* a = q.dArgs + s;
        const unsigned gxN = (unsigned)nblk(q.h[s].n, 256), gxB = (unsigned)nblk(q.h[s].n, CBLK), gxT = q.h[s].tilesUpper;
        hipLaunchKernelGGL(k_cf_apply_gc, dim3((gxN + CF_EPT - 1) / CF_EPT, 1), dim3(256), 0, ctx->stream, a, 1);      // the first NormalizeByGC, deferred until now (medians of problem 1 are still in CleanDev)
        hipLaunchKernelGGL(k_cf_apply_var, dim3((gxN + CF_EPT - 1) / CF_EPT, 1), dim3(256), 0, ctx->stream, a);
        if (q.useCq) hipLaunchKernelGGL(k_cf_xform_gc, dim3((gxN + CF_EPT - 1) / CF_EPT, 1), dim3(256), 0, ctx->stream, a, 2);       // the counting selects left the grouped keys as they were
        hipLaunchKernelGGL(k_cf_xform_var, dim3((gxN + CF_EPT - 1) / CF_EPT, 1), dim3(256), 0, ctx->stream, a);
        cf_gc_medians(ctx, a, 1, gxN, gxT, 3, 2, true);
        hipLaunchKernelGGL(k_cf_flags_final, dim3(gxB, 1), dim3(256), 0, ctx->stream, a);
        hipLaunchKernelGGL(k_cf_scatter_final, dim3(gxB, 1), dim3(256), 0, ctx->stream, a, 1);
        again = true;
    }
    if (secondPhase) *secondPhase = again;
    if (again) {
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(ctx->pin, q.dD, (size_t)B * sizeof(CleanDev), hipMemcpyDeviceToHost, ctx->stream));
        CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        CANVAS_HIP_TRY(ctx, hipGetLastError());
    }
    for (int s = 0; s < B; s++) {
        const CleanDev& H = ((const CleanDev*)ctx->pin)[s];
        if (H.bad) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_clean: a bin has gc outside 0..100 or a chromosome index outside [0, nchr) (the reference throws IndexOutOfRangeException)");
        if (H.cqFail) ctx->clean_cq_failed = true;                         // the caller redoes the sample with the radix selects
        if (H.fallback) continue;                                          // handled[s] stays 0: nothing was written to the caller's arrays
        handled[s] = 1;
        h_n_out[s] = (int64_t)H.nFinal;
        if (h_local_sd_out) h_local_sd_out[s] = H.haveLocalSd ? H.localSd : -1.0;
        if (h_info) {
            int32_t info[8] = {0};
            info[0] = (int32_t)H.nA; info[1] = (int32_t)H.nAB; info[2] = (int32_t)(H.gcActive ? H.kept : (long long)H.nAB); info[3] = (int32_t)H.nFinal; info[4] = H.changed; info[5] = q.useCq ? 1 : 0;
            memcpy(h_info + 8 * s, info, sizeof info);
        }
    }
    return CANVAS_OK;
}
// one sample = a batch of one
static int32_t clean_device_driven(canvas_ctx* ctx, int64_t n, int32_t* d_chr, int32_t* d_start, int32_t* d_stop, float* d_count, int32_t* d_gc, int32_t nchr, const uint8_t* h_chr_is_autosome,
                                   uint32_t flags, int32_t min_bins_per_gc, double* h_local_sd_out, int64_t* h_n_out, int32_t* h_info, bool* handled) {
    *handled = false;
    char h = 0; double lsd = -1.0; int64_t nOut = 0; int32_t info[8] = {0};
    for (int attempt = 0; attempt < 2; attempt++) {
        const bool useCq = attempt == 0 && clean_counting_selects() && !ctx->clean_cq_skip;
        ctx->clean_cq_failed = false;
        int32_t rc = clean_batch_enqueue(ctx, 1, &n, &d_chr, &d_start, &d_stop, &d_count, &d_gc, nchr, h_chr_is_autosome, flags, min_bins_per_gc, useCq); if (rc) return rc;
        rc = clean_batch_finish(ctx, &lsd, &nOut, info, &h); if (rc) return rc;
        if (h || !ctx->clean_cq_failed) break;                              // (a sample the counting selects gave up on is redone once, with the radix selects)
    }
    if (h) { *handled = true; *h_n_out = nOut; if (h_local_sd_out) *h_local_sd_out = lsd; if (h_info) memcpy(h_info, info, sizeof info); }
    return CANVAS_OK;
}
