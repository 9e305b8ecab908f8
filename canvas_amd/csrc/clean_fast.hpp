// CanvasClean without host round trips (included by clean.hip): CanvasClean.Main (CanvasClean/CanvasClean.cs:415-533) for the MedianByGC flavour with the default
// weighted-median setting (-w >= 100, which makes every GC bucket that survives RemoveBinsWithExtremeGC hold >= 100 autosomal bins: CanvasClean.cs:207-237,178-187).
//
// Every decision the reference takes once per file — the size threshold, the GC strip, the per-GC medians, whether the variance normalisation applies and changes
// anything, the local-SD filter — is taken by a one-workgroup kernel that leaves its result in device memory (CleanDev); the kernels that follow read it from there
// and the kernels of a branch that is not taken find an empty problem and return.  The host enqueues the whole stage in one go and synchronises once, at the end.
//   caller's arrays --(size filter + outlier filter: ONE compaction)--> scratch SoA  ...in-place normalisation...  --(GC strip + local-SD filter: ONE compaction)--> caller's arrays
// The caller's arrays are not written before the last kernel, so a case this path does not cover (more than CF_MAXRUN chromosome runs) is detected on the device, leaves the
// input intact and is handed to the host-driven path of clean.hip.  All per-bin arithmetic and all order statistics are the ones of that path: results are bit-identical.
//
// The stage is BATCH-NATIVE: every kernel takes a table of per-sample argument blocks (CfArgs, in device memory) and works on the block of blockIdx.y, so a cohort of B samples
// costs the same ~30 launches as one sample (grid.y = B) and no sample's launch latency is paid B times.  One sample is a batch of one.
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
// per value (one sweep) instead of four radix passes.  The counters cover a window of CQW values around the sample's level (estimated from 33 strided keys), one row per GC
// bucket plus one for the genome; keys below the window are counted per row, keys above it are what is left.  Every key is checked (cq_value(k) == x, below) and every
// requested rank must fall inside the window: otherwise cqFail is raised and the sample is redone with the radix selects — the result never depends on the window.
#define CQW 16384            // counter slots per row: +-82 count units around the level
#define CQ_TILE 32768        // keys per workgroup of the counting sweep (the 64 KB of LDS counters are zeroed and flushed once per tile)
struct CfCq {
    int32_t lo; uint32_t bad, fail, ntiles;
    unsigned long long nbelow;       // normalised counts under the window of k_cq_nhist
    uint32_t below[NGC + 1];         // keys under the window, per bucket and [NGC] for the genome
    uint32_t inWin[NGC + 1];         // keys inside it
    int32_t kq[NGC][6];              // the quartile order statistics of a bucket (as k), in quartile_indices order
};
// k -> cq_value(k) may be ANY non-decreasing map as long as the check and the reconstruction use the same one: a key is accepted only if cq_value(k) gives it back bit for
// bit, so accepted keys are in strictly increasing correspondence with their k.  (float)(k * 0.01) is that map (one multiplication; the division k / 100.0 made
// k_cq_hist VALU-bound: 37 -> 31 us).  It reproduces every integer count — the read counts of a WGS .binned file — and the float.Parse of a two-decimal text except where
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
struct CfArgs {                      // one sample of the batch
    int64_t n;                       // bins handed in
    int32_t nb, nchr, minBinsPerGc, wantLsd, doSize, doOutlier;
    uint32_t flags, tilesUpper;
    Soa caller, S1;                  // the caller's arrays; the scratch copy between the two compactions
    uint8_t* dFlags; uint32_t* dBlk; uint32_t* szHist; uint32_t* szOver; uint32_t szOverCap, padA; uint32_t* keysG; const uint8_t* isAuto;
    double* dSd; double* dRunMad; int64_t* dRunStart; long long* dPos;
    CleanDev* D; CfSel* P; SelTile* tiles; uint32_t* hist;
    CfCq* cq; uint32_t* cqHist; SelTile* cqTiles;      // the counting selects (below)
    uint32_t* repl;                  // [CF_HREP][2 * NGC] GC counts of the survivors per replica (k_cf_flags_ab), then [CF_HREP][NGC] write cursors into the grouped keys (k_cf_dec_gc -> k_cf_scatter_ab)
};
#define CF_SAMPLE const CfArgs& A = AA[blockIdx.y]

// ---------------------------------------------------------------- RemoveBigBins threshold (CanvasClean.cs:328-348): the 98th percentile of the bin sizes
// Sizes are small integers, so the order statistic is read off an exact count per size (one sweep of start / stop, no key array, no radix passes): sizes below CF_SZ_LDS are
// counted in LDS per workgroup and flushed, sizes up to CF_SZ_BINS go to the global counters directly, larger ones only matter if the percentile itself is that large.
#define CF_SZ_GRID 128       // workgroups per sample in the size count: each takes n / 128 bins, so its LDS counters absorb many bins per distinct size before the flush
__global__ void __launch_bounds__(1024) k_cf_size_hist(const CfArgs* __restrict__ AA) {
    CF_SAMPLE;
    __shared__ uint32_t lh[CF_SZ_LDS];
    if (!A.doSize) return;
    const int64_t n = A.n;
    for (int i = threadIdx.x; i < CF_SZ_LDS; i += 1024) lh[i] = 0;
    __syncthreads();
    const gptr<const int32_t> start = as_global(A.caller.start), stop = as_global(A.caller.stop);
    uint32_t neg = 0;
    for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += (int64_t)CF_SZ_GRID * 1024) {
        const int32_t sz = stop[i] - start[i];
        if (sz < 0) neg++;
        else if (sz < CF_SZ_LDS) atomicAdd(&lh[sz], 1u);
        else if (sz < CF_SZ_BINS) atomicAdd(&A.szHist[sz], 1u);
        else { const unsigned int k = atomicAdd(&A.D->nOver, 1u); if (k < A.szOverCap) A.szOver[k] = (uint32_t)sz; }
    }
    neg = wave_reduce_add_u32(neg);
    if (lane_id() == 0 && neg) atomicAdd(&A.D->sizeBelow, (unsigned long long)neg);
    __syncthreads();
    for (int i = threadIdx.x; i < CF_SZ_LDS; i += 1024) { const uint32_t v = lh[i]; if (v) atomicAdd(&A.szHist[i], v); }
}
// one workgroup per sample: the size with cumulative count > index (the element at `index` of the sorted sizes)
__global__ void __launch_bounds__(1024) k_cf_size_pick(const CfArgs* __restrict__ AA) {
    CF_SAMPLE;
    __shared__ unsigned long long sTot[1024 / 64];
    __shared__ int sFound;
    CleanDev* __restrict__ D = A.D;
    const int64_t index = (int64_t)(0.98 * (double)A.n);                       // CanvasClean.cs:339
    const bool on = A.doSize && index < A.n;
    if (!on) { if (threadIdx.x == 0) D->sizeOn = 0u; return; }
    const unsigned long long want = (unsigned long long)index, below = D->sizeBelow;
    if (threadIdx.x == 0) sFound = 0;
    // the counters are scanned in two stretches of 4 x 1024 x k bins (16-byte loads): [0, CF_SZ_LDS) first — where the bins of a WGS sample are — then the rest
    unsigned long long total = below;
    for (int part = 0; part < 2; part++) {
        const int lo = part == 0 ? 0 : CF_SZ_LDS, per = part == 0 ? CF_SZ_LDS / 1024 : (CF_SZ_BINS - CF_SZ_LDS) / 1024;      // 4, 60 bins per thread
        const uint32_t* __restrict__ h = A.szHist + lo + (size_t)threadIdx.x * per;
        unsigned long long mine = 0;
        for (int k = 0; k < per; k += 4) { const uint4 q = *reinterpret_cast<const uint4*>(h + k); mine += (unsigned long long)q.x + q.y + q.z + q.w; }
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
            for (int k = 0; k < per; k++) { cum += h[k]; if (want < cum) { D->sizeThresh = lo + (int)threadIdx.x * per + k; break; } }
            sFound = 1;
        }
        total += sum;
        __syncthreads();
        if (sFound) break;
    }
    if (threadIdx.x == 0) {
        D->sizeOn = 1u;
        if (want < below || (want >= total && D->nOver > A.szOverCap)) D->fallback = 1u;     // a negative percentile, or more large bins than the list holds: the host-driven path
    }
    if (!sFound && want >= total && D->nOver <= A.szOverCap) {
        // the percentile lies among the bins of CF_SZ_BINS positions and more (a heavy tail of bins across assembly gaps): exact order statistic of the list, in this workgroup
        __shared__ uint32_t sH[2][256];
        __shared__ unsigned long long sPre[2], sK[2];
        const uint32_t* __restrict__ ov = A.szOver;
        const unsigned long long t = want - total;
        __syncthreads();
        wg_select2([&](int64_t i) { return (unsigned long long)ov[i]; }, 0, (int64_t)D->nOver, t, t, sH, sPre, sK);
        if (threadIdx.x == 0) D->sizeThresh = (int32_t)(uint32_t)sPre[0];
    }
}
// exclusive prefix sum of one uint32 per thread over a 128-thread workgroup (two waves); *total = the sum.  One barrier.
__device__ __forceinline__ uint32_t cf_excl_scan128(uint32_t v, uint32_t* sh2 /* [2] */, uint32_t* total) {
    const uint32_t inc = wave_inclusive_scan_u32(v);
    if ((threadIdx.x & 63) == 63) sh2[threadIdx.x >> 6] = inc;
    __syncthreads();
    *total = sh2[0] + sh2[1];
    return inc - v + (threadIdx.x >= 64 ? sh2[0] : 0u);
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
__global__ void __launch_bounds__(256) k_cf_flags_ab(const CfArgs* __restrict__ AA) {
    CF_SAMPLE;
    __shared__ uint32_t sh[8];
    // keepA, chromosome and count of this block's bins and of the bin on either side of it: slot 0 = bin base - 1, slots 1 .. L = the block, slot L + 1 = bin base + L.
    // The neighbour search below runs on these slots with 32-bit indices; it leaves them only when the halo bin itself fails the size filter (slow path, global memory).
    __shared__ uint8_t sA[CBLK + 2];
    __shared__ int32_t sChr[CBLK + 2];
    __shared__ float sCnt[CBLK + 2];
    __shared__ uint8_t sGc[CBLK];
    __shared__ uint8_t sAuto[256];                        // isAuto of the first 256 chromosomes (more than that: read from the table)
    __shared__ uint32_t lh[2 * NGC];                      // GC histogram of the survivors (CanvasClean.cs:207-223): [0..100] autosomal, [101..201] the others
    const int64_t n = A.n;
    const int64_t base = (int64_t)blockIdx.x * CBLK;
    if (base >= n) return;
    const int L = (int)min<int64_t>(CBLK, n - base);
    if (threadIdx.x < 2 * NGC) lh[threadIdx.x] = 0;
    const gptr<const uint8_t> isAuto = as_global(A.isAuto);         // (pointers out of the argument table: converted so that the accesses are global_load, not flat_load)
    const GSoa in = as_global(A.caller);
    const gptr<const int32_t> chr = in.chr, start = in.start, stop = in.stop, gc = in.gc; const gptr<const float> count = in.count;
    const gptr<uint8_t> flags = as_global(A.dFlags); CleanDev* __restrict__ D = A.D;
    const int nchr = A.nchr, doOutlier = A.doOutlier;
    const bool doSize = D->sizeOn != 0;                   // off when the filter is, or when the percentile index falls past the end
    const int32_t thresh = doSize ? D->sizeThresh : 0;
    sAuto[threadIdx.x] = (int)threadIdx.x < nchr ? isAuto[threadIdx.x] : 0;
    uint32_t nKeep = 0, nSize = 0, bad = 0;
#pragma unroll
    for (int j = 0; j < CBLK / 256; j++) {
        const int li = j * 256 + threadIdx.x;
        uint8_t a = 0; int32_t c = -1; float v = 0.0f; int32_t g = 0;
        if (li < L) {
            const int64_t i = base + li;
            c = chr[i]; v = count[i]; g = gc[i];
            if ((uint32_t)g > 100u || (uint32_t)c >= (uint32_t)nchr) bad = 1;
            a = (!doSize || (stop[i] - start[i]) <= thresh) ? 1 : 0;
        }
        sA[1 + li] = a; sChr[1 + li] = c; sCnt[1 + li] = v; sGc[li] = (uint8_t)((uint32_t)g > 100u ? 100 : g);
        nSize += a;
    }
    const bool hasLeft = base > 0, hasRight = base + L < n;
    if (threadIdx.x < 2) {
        const int64_t i = threadIdx.x == 0 ? base - 1 : base + L;
        uint8_t a = 0; int32_t c = -1; float v = 0.0f;
        if (threadIdx.x == 0 ? hasLeft : hasRight) { c = chr[i]; v = count[i]; a = (!doSize || (stop[i] - start[i]) <= thresh) ? 1 : 0; }
        const int slot = threadIdx.x == 0 ? 0 : L + 1;
        sA[slot] = a; sChr[slot] = c; sCnt[slot] = v;
    }
    __syncthreads();
#pragma unroll 2
    for (int j = 0; j < CBLK / 256; j++) {
        const int li = j * 256 + threadIdx.x;
        if (li >= L) continue;
        const int s = li + 1;
        bool keep = sA[s] != 0;
        const int32_t c = sChr[s];
        if (keep && doOutlier) {
            // nearest bins on either side that pass the size filter (RemoveOutliers runs on the list RemoveBigBins left, CanvasClean.cs:387-413)
            int ps = s - 1, qs = s + 1;
            while (ps >= 1 && !sA[ps]) ps--;
            while (qs <= L && !sA[qs]) qs++;
            bool hasPrev, hasNext; int32_t cp = -1, cq = -1; float vp = 0.0f, vq = 0.0f;
            if (ps >= 1 || !hasLeft || sA[0]) { hasPrev = ps >= 1 || hasLeft; if (hasPrev) { cp = sChr[ps]; vp = sCnt[ps]; } }
            else {                                        // the bin in front of the block fails the size filter too: keep looking in global memory
                int64_t p = base - 2;
                while (p >= 0 && (stop[p] - start[p]) > thresh) p--;
                hasPrev = p >= 0; if (hasPrev) { cp = chr[p]; vp = count[p]; }
            }
            if (qs <= L || !hasRight || sA[L + 1]) { hasNext = qs <= L || hasRight; if (hasNext) { cq = sChr[qs]; vq = sCnt[qs]; } }
            else {
                int64_t q = base + L + 1;
                while (q < n && (stop[q] - start[q]) > thresh) q++;
                hasNext = q < n; if (hasNext) { cq = chr[q]; vq = count[q]; }
            }
            const bool prevSame = hasPrev && cp == c, nextSame = hasNext && cq == c;
            if ((hasPrev && !prevSame) && (hasNext && !nextSame)) keep = false;
            else {
                const float v = sCnt[s];
                keep = (prevSame && !sig_diff(v, vp)) || (nextSame && !sig_diff(v, vq)) || (!hasPrev && !hasNext);
            }
        }
        flags[base + li] = keep;
        nKeep += keep;
        if (keep) {                                       // out-of-range input is reported through D->bad and nothing is returned: clamped here so that no table is overrun
            const int32_t cc = (uint32_t)c >= (uint32_t)nchr ? 0 : c;
            const uint8_t au = cc < 256 ? sAuto[cc] : isAuto[cc];
            atomicAdd(&lh[(au ? 0 : NGC) + sGc[li]], 1u);
        }
    }
    nKeep = wave_reduce_add_u32(nKeep); nSize = wave_reduce_add_u32(nSize);
    if (lane_id() == 0) { sh[threadIdx.x >> 6] = nKeep; sh[4 + (threadIdx.x >> 6)] = nSize; }
    if (bad) D->bad = 1u;
    __syncthreads();
    if (threadIdx.x == 0) { A.dBlk[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3]; A.dBlk[A.nb + 2 + blockIdx.x] = sh[4] + sh[5] + sh[6] + sh[7]; }      // (k_cf_scan_blocks sums the second column into nA)
    if (threadIdx.x < 2 * NGC && lh[threadIdx.x]) atomicAdd(&A.repl[(blockIdx.x % CF_HREP) * (2 * NGC) + threadIdx.x], lh[threadIdx.x]);
}
// exclusive scan of the block counts: phase 0 over the blocks of the input (total -> nAB), phase 1 over the blocks of the nAB surviving bins (total -> nFinal)
__global__ void __launch_bounds__(1024) k_cf_scan_blocks(const CfArgs* __restrict__ AA, int phase) {
    CF_SAMPLE;
    __shared__ uint32_t sh[17];
    uint32_t* __restrict__ blockCnt = A.dBlk;
    const int nblocks = phase == 0 ? A.nb : (int)(((int64_t)A.D->nAB + CBLK - 1) / CBLK);
    uint32_t carry = 0;
    for (int base = 0; base < nblocks; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t v = i < nblocks ? blockCnt[i] : 0;
        const uint32_t inc = wave_inclusive_scan_u32(v);
        const int w = threadIdx.x >> 6;
        if (lane_id() == 63) sh[w] = inc;
        __syncthreads();
        if (threadIdx.x == 0) { uint32_t s = 0; for (int k = 0; k < 16; k++) { const uint32_t tt = sh[k]; sh[k] = s; s += tt; } sh[16] = s; }
        __syncthreads();
        if (i < nblocks) blockCnt[i] = carry + sh[w] + inc - v;
        carry += sh[16];
        __syncthreads();
    }
    if (threadIdx.x == 0) { if (phase == 0) A.D->nAB = carry; else A.D->nFinal = carry; }
    if (phase == 0) {                                     // bins after RemoveBigBins alone: the second column of k_cf_flags_ab's block counts
        uint32_t v = 0;
        for (int i = threadIdx.x; i < nblocks; i += 1024) v += blockCnt[A.nb + 2 + i];
        v = wave_reduce_add_u32(v);
        if (lane_id() == 0) sh[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) { uint32_t t = 0; for (int k = 0; k < 16; k++) t += sh[k]; A.D->nA = t; }
    }
}
// the compaction itself: caller's arrays -> scratch SoA, and — the GC strip is decided by then (k_cf_dec_gc) — the order-preserving keys of the autosomal survivors with a kept GC
// value, grouped by GC (order inside a bucket is irrelevant: only order statistics are taken).  CountDeviation is written by k_cf_local_sd (or never read).
__global__ void __launch_bounds__(256) k_cf_scatter_ab(const CfArgs* __restrict__ AA) {
    CF_SAMPLE;
    __shared__ uint32_t sh[4];
    __shared__ uint32_t lcnt[NGC], lbase[NGC];
    __shared__ uint8_t sKeepGc[NGC];
    const int64_t n = A.n;
    const int64_t base = (int64_t)blockIdx.x * CBLK;
    if (base >= n) return;
    const gptr<const uint8_t> flags = as_global(A.dFlags), isAuto = as_global(A.isAuto);
    const GSoa src = as_global(A.caller), dst = as_global(A.S1); const int nchr = A.nchr;
    CleanDev* __restrict__ D = A.D;
    const bool group = D->gcActive != 0;
    if (threadIdx.x < NGC) { lcnt[threadIdx.x] = 0; sKeepGc[threadIdx.x] = D->keepGc[threadIdx.x]; }
    __syncthreads();
    uint32_t running = A.dBlk[blockIdx.x];
    uint32_t myRank[CBLK / 256], myKey[CBLK / 256]; int myGc[CBLK / 256];
#pragma unroll
    for (int j = 0; j < CBLK / 256; j++) {
        const int64_t i = base + j * 256 + threadIdx.x;
        const uint32_t f = (i < n) ? flags[i] : 0;
        const uint32_t inc = wave_inclusive_scan_u32(f);
        if (lane_id() == 63) sh[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t woff = 0, tot = 0;
        for (int k = 0; k < 4; k++) { if (k < (int)(threadIdx.x >> 6)) woff += sh[k]; tot += sh[k]; }
        myGc[j] = -1;
        if (f) {
            const uint32_t d = running + woff + inc - 1;
            // out-of-range input is reported through D->bad (k_cf_flags_ab) and nothing is returned; the scratch copy holds clamped values so that no later kernel indexes past a table
            const int32_t c0 = src.chr[i], g0 = src.gc[i];
            const int32_t g = (uint32_t)g0 > 100u ? 100 : g0, c = (uint32_t)c0 >= (uint32_t)nchr ? 0 : c0;
            const float v = src.count[i];
            dst.chr[d] = c; dst.start[d] = src.start[i]; dst.stop[d] = src.stop[i]; dst.gc[d] = g; dst.count[d] = v;
            if (group && isAuto[c] && sKeepGc[g]) { myGc[j] = g; myKey[j] = key_of_float(v); myRank[j] = atomicAdd(&lcnt[g], 1u); }
        }
        running += tot;
        __syncthreads();
    }
    if (!group) return;
    if (threadIdx.x < NGC && lcnt[threadIdx.x]) lbase[threadIdx.x] = atomicAdd(&A.repl[CF_HREP * (2 * NGC) + (blockIdx.x % CF_HREP) * NGC + threadIdx.x], lcnt[threadIdx.x]);      // absolute position in keysG
    __syncthreads();
    const gptr<uint32_t> keysG = as_global(A.keysG);
#pragma unroll
    for (int j = 0; j < CBLK / 256; j++) if (myGc[j] >= 0) keysG[lbase[myGc[j]] + myRank[j]] = myKey[j];
}

// ---------------------------------------------------------------- local SD (CanvasClean.cs:243-298)
// (a variant that staged 5120 bins per workgroup through LDS for coalesced loads / stores was measured: 39 us against 27 us — the strided accesses hit in L2)
// one thread per window of 20 count differences (Utilities.StandardDeviation, CanvasClean.cs:262-298) + the chromosome boundaries among the window's bins (the run records that
// GetLocalStandardDeviationAverage's per-chromosome grouping needs) + CountDeviation = -1 (GenomicBin.cs:83) for the bins behind the last window
__global__ void __launch_bounds__(256) k_cf_local_sd(const CfArgs* __restrict__ AA) {
    CF_SAMPLE;
    if (!A.wantLsd) return;
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t nAB = (int64_t)A.D->nAB, Dn = nAB - 1, nW = Dn >= 1 ? (Dn - 1) / 20 : 0;
    if (w > nW) return;
    const int64_t lo = w * 20, hi = w < nW ? lo + 20 : nAB;             // thread nW takes the tail
    const GSoa S1 = as_global(A.S1);
    if (w < nW) local_sd_window(gptr<const float>(S1.count), w, as_global(A.dSd), S1.dev);
    else for (int64_t i = lo; i < hi; i++) S1.dev[i] = -1.0;
    const gptr<const int32_t> chr = S1.chr;
    int32_t prev = lo > 0 ? chr[lo - 1] : -1;
    for (int64_t i = lo; i < hi; i++) {
        const int32_t c = chr[i];
        if (i == 0 || c != prev) { const unsigned int k = atomicAdd(&A.D->nRunRec, 1u); if (k < 65536u) A.dPos[k] = (long long)((i << 20) | (long long)(c & 0xFFFFF)); }
        prev = c;
    }
}
// one workgroup: sorts the (position << 20 | chromosome) records of k_cf_local_sd, derives the runs of windows per chromosome exactly as local_sd_begin does on the host
__global__ void __launch_bounds__(1024) k_cf_runs_build(const CfArgs* __restrict__ AA) {
    CF_SAMPLE;
    __shared__ long long s[CF_MAXRUN];
    CleanDev* __restrict__ D = A.D; const long long* __restrict__ recs = A.dPos; int64_t* __restrict__ runStart = A.dRunStart;
    const unsigned long long nAB = D->nAB;
    const int have = A.wantLsd && nAB >= 50000ull;                         // CanvasClean.cs:483-486
    if (threadIdx.x == 0) D->haveLocalSd = have;
    if (!have) { if (threadIdx.x == 0) D->nruns = 0; return; }
    const unsigned int nb = D->nRunRec;
    if (nb > CF_MAXRUN) { if (threadIdx.x == 0) { D->fallback = 1u; D->nruns = 0; } return; }
    const int t = threadIdx.x;
    __shared__ long long raw[CF_MAXRUN];
    raw[t] = t < (int)nb ? recs[t] : 0x7FFFFFFFFFFFFFFFll;
    __syncthreads();
    // rank sort: every record counts the records in front of it (independent broadcast reads of LDS).  A one-thread insertion sort of the ~24 records of a sorted
    // file was a chain of dependent LDS accesses (10 of the kernel's 17 us), the bitonic network 55 barriers.
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
__global__ void __launch_bounds__(1024) k_cf_run_mad(const CfArgs* __restrict__ AA) {
    CF_SAMPLE;
    if (!A.wantLsd) return;
    run_mad_body(A.dSd, A.dRunStart, A.dRunMad, &A.D->nruns);
}
__global__ void k_cf_lsd_avg(const CfArgs* __restrict__ AA) {
    CF_SAMPLE;
    if (threadIdx.x || blockIdx.x) return;
    CleanDev* D = A.D;
    if (!D->haveLocalSd) { D->localSd = -1.0; return; }
    double s = 0;
    for (int r = 0; r < D->nruns; r++) s += A.dRunMad[r];                  // List<double>.Average(): sequential sum / count
    D->localSd = s / (double)D->nruns;
}

// ---------------------------------------------------------------- RemoveBinsWithExtremeGC decision (CanvasClean.cs:207-237) and what follows from it
__global__ void __launch_bounds__(128) k_cf_dec_gc(const CfArgs* __restrict__ AA) {
    CF_SAMPLE;
    __shared__ uint32_t shA[2], shB[2], shC[2], shD[2];
    CleanDev* __restrict__ D = A.D; const uint32_t flags = A.flags; const int minBinsPerGc = A.minBinsPerGc;
    const int t = threadIdx.x;
    const long long nAB = (long long)D->nAB;
    uint32_t hA = 0, hO = 0;
    if (t < NGC) for (int r = 0; r < CF_HREP; r++) { hA += A.repl[r * (2 * NGC) + t]; hO += A.repl[r * (2 * NGC) + NGC + t]; }
    if (t < NGC) { D->hist[t] = hA; D->hist[NGC + t] = hO; }
    // the counts are integers below 2^32 and there are 101 of them: their double sum (CanvasClean.cs:219-222) is exact in any order
    uint32_t totalA; (void)cf_excl_scan128(hA, shA, &totalA);
    const bool consider = (flags & CANVAS_CLEAN_GCNORM) && nAB > 0;
    const int averageCountPerGC = max(minBinsPerGc, (int)((double)totalA / NGC));
    const int threshold = min(100, averageCountPerGC);
    bool kp = true;
    if (consider && t < NGC) kp = (int)hA >= threshold;
    // bins of any chromosome in the kept buckets (each term < 2^32, 101 terms: the 64-bit total is split over two 32-bit scans of the halves)
    const unsigned long long mine = (consider && t < NGC && kp) ? (unsigned long long)hA + (unsigned long long)hO : 0ull;
    uint32_t totLo, totHi; (void)cf_excl_scan128((uint32_t)(mine & 0xFFFFu), shB, &totLo); (void)cf_excl_scan128((uint32_t)(mine >> 16), shC, &totHi);
    const long long kept = (long long)totLo + ((long long)totHi << 16);
    const bool active = consider && kept > 0;                               // kept <= 0: "proceed without GC correction" (CanvasClean.cs:500-505)
    if (!active) kp = true;
    uint32_t totalKept; const uint32_t so = cf_excl_scan128((active && t < NGC && kp) ? hA : 0u, shD, &totalKept);
    if (t < NGC) {
        D->keepGc[t] = kp ? 1 : 0; D->medians[t] = 0.0; D->segOff[t] = active ? so : 0u;
        // the bucket's stretch of the grouped keys is filled replica by replica (the order inside a bucket is irrelevant: only order statistics are taken from it)
        uint32_t at = so;
        for (int r = 0; r < CF_HREP; r++) { A.repl[CF_HREP * (2 * NGC) + r * NGC + t] = at; at += A.repl[r * (2 * NGC) + t]; }
    }
    if (t == 0) {
        const long long sKept = active ? kept : nAB;
        D->segOff[NGC] = active ? totalKept : 0u; D->kept = sKept; D->gcActive = active ? 1 : 0; D->changed = 0;
        // NormalizeVarianceByGC runs for whole-genome samples only (CanvasClean.cs:512-519); the host enqueues its kernels when the INPUT has more than 500000 bins
        const bool haveLsd = A.wantLsd && nAB >= 50000;                       // what k_cf_runs_build will store in haveLocalSd (CanvasClean.cs:483-486)
        D->varActive = (active && haveLsd && sKept > 500000 && A.n > 500000) ? 1 : 0;
    }
}
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
    float* __restrict__ count = A.S1.count; const int32_t* __restrict__ gc = A.S1.gc;
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
// ---------------------------------------------------------------- the counting selects (see CfCq)
// window, sweep tiles and the marker / queries the later kernels look at; gated like k_cf_sel_setup(…, gate 0)
__global__ void __launch_bounds__(128) k_cq_setup(const CfArgs* __restrict__ AA) {
    CF_SAMPLE;
    __shared__ uint32_t so[NGC + 1];
    __shared__ long long sv[33];
    __shared__ uint32_t shT[2];
    const CleanDev* D = A.D; CfCq* C = A.cq; CfSel* P1 = A.P + 1; CfSel* P2 = A.P + 2;
    const int t = threadIdx.x;
    if (t <= NGC) so[t] = D->segOff[t];
    __syncthreads();
    const uint32_t total = so[NGC];
    if (!D->gcActive || total == 0) { if (t == 0) { P1->hdr[0] = 0; P1->hdr[1] = 0; P2->hdr[0] = 0; P2->hdr[1] = 0; C->ntiles = 0; } return; }
    if (t < 33) { const uint32_t i = (uint32_t)((double)total * (t + 0.5) / 33.0); long long k = -1; if (i < total && !cq_key(float_of_key(A.keysG[i]), k)) k = -1; sv[t] = k; }
    __syncthreads();
    if (t < 33) {                                            // every lane ranks its own sample among the valid ones; the one in the middle sets the window
        const long long mine = sv[t];
        int m = 0, rank = 0;
        for (int j = 0; j < 33; j++) { const long long o = sv[j]; if (o >= 0) { m++; if (o < mine || (o == mine && j < t)) rank++; } }
        if (m == 0) { if (t == 0) C->lo = 0; }
        else if (mine >= 0 && rank == m / 2) C->lo = (int32_t)(mine > CQW / 2 ? mine - CQW / 2 : 0);
    }
    uint32_t totT;
    const uint32_t myTiles = t < NGC ? (so[t + 1] - so[t] + CQ_TILE - 1) / CQ_TILE : 0u;
    const uint32_t exT = cf_excl_scan128(myTiles, shT, &totT);
    if (t < NGC) { uint32_t k = exT; for (int64_t b = so[t]; b < (int64_t)so[t + 1]; b += CQ_TILE) A.cqTiles[k++] = SelTile{t, b, min<int64_t>(b + CQ_TILE, (int64_t)so[t + 1])}; }
    if (t == 0) {
        C->ntiles = totT;
        P1->hdr[0] = 0; P1->hdr[1] = 1;                      // "NormalizeByGC has been decided" for k_cf_scatter_final / k_cf_apply_gc
        P2->hdr[0] = 0; P2->hdr[1] = 0;
        if (D->varActive) {                                  // the genome's quartile ranks (k_cq_nhist / k_cq_nresolve)
            const QuartIdx qi = quartile_indices((int64_t)total);
            for (int k = 0; k < qi.n; k++) { P2->qk[k] = (unsigned long long)qi.idx[k]; P2->qprefix[k] = 0ull; }
            P2->hdr[1] = (uint32_t)qi.n; P2->first[NGC] = 0;
        }
    }
}
// one sweep of the grouped keys: counters per value in LDS, flushed into the bucket's row and the genome's
__global__ void __launch_bounds__(1024) k_cq_hist(const CfArgs* __restrict__ AA) {
    CF_SAMPLE;
    __shared__ uint32_t lw[CQW];
    CfCq* __restrict__ C = A.cq;
    if (blockIdx.x >= C->ntiles) return;
    const SelTile T = A.cqTiles[blockIdx.x];
    const long long lo = C->lo;
    for (int i = threadIdx.x; i < CQW; i += 1024) lw[i] = 0;
    __syncthreads();
    const gptr<const uint32_t> keys = as_global(A.keysG);
    uint32_t below = 0, bad = 0;
    for (int64_t i0 = T.begin + threadIdx.x; i0 < T.end; i0 += 4 * 1024) {
        uint32_t kk[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int64_t i = i0 + (int64_t)u * 1024; kk[u] = i < T.end ? keys[i] : 0u; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (i0 + (int64_t)u * 1024 >= T.end) break;
            long long k;
            if (!cq_key(float_of_key(kk[u]), k)) { bad = 1; continue; }
            if (k < lo) below++;
            else if (k - lo < CQW) atomicAdd(&lw[k - lo], 1u);
        }
    }
    below = wave_reduce_add_u32(below);
    if ((threadIdx.x & 63) == 0 && below) { atomicAdd(&C->below[T.seg], below); atomicAdd(&C->below[NGC], below); }
    if (bad) C->bad = 1u;
    __syncthreads();
    uint32_t* __restrict__ row = A.cqHist + (size_t)T.seg * CQW; uint32_t* __restrict__ all = A.cqHist + (size_t)NGC * CQW;
    for (int i = threadIdx.x; i < CQW; i += 1024) { const uint32_t v = lw[i]; if (v) { atomicAdd(&row[i], v); atomicAdd(&all[i], v); } }
}
// workgroup g: the order statistics of bucket g (g == NGC: of the genome) from its counters — the NormalizeByGC medians (CanvasClean.cs:170-189) and, for the variance
// normalisation, the bucket's quartile statistics as k (a bucket is scaled by one factor, so they keep their ranks)
__global__ void __launch_bounds__(1024) k_cq_pick(const CfArgs* __restrict__ AA) {
    CF_SAMPLE;
    __shared__ uint32_t swave[16];
    __shared__ long long sK[8];
    __shared__ int sFail;
    CleanDev* __restrict__ D = A.D; CfCq* __restrict__ C = A.cq;
    if (A.P[1].hdr[1] == 0) return;
    const int g = blockIdx.x, t = threadIdx.x;
    if (g == 0 && t == 0 && C->bad) { C->fail = 1u; D->cqFail = 1; D->fallback = 1u; }
    const int64_t cnt = g < NGC ? (int64_t)D->segOff[g + 1] - (int64_t)D->segOff[g] : (int64_t)D->segOff[NGC];
    if (cnt <= 0) return;
    int64_t ranks[8]; int nr = 0;
    if (cnt % 2) ranks[nr++] = cnt / 2; else { ranks[nr++] = cnt / 2 - 1; ranks[nr++] = cnt / 2; }
    const int nMed = nr;
    if (g < NGC && D->varActive) { const QuartIdx qi = quartile_indices(cnt); for (int k = 0; k < qi.n; k++) ranks[nr++] = qi.idx[k]; }
    const uint4* __restrict__ row = reinterpret_cast<const uint4*>(A.cqHist + (size_t)g * CQW) + 4 * t;     // 16 consecutive counters per thread
    uint32_t c[16];
#pragma unroll
    for (int u = 0; u < 4; u++) { const uint4 v = row[u]; c[4 * u] = v.x; c[4 * u + 1] = v.y; c[4 * u + 2] = v.z; c[4 * u + 3] = v.w; }
    if (g == NGC) {                                       // the genome's row has served its purpose: cleared, k_cq_nhist counts the normalised values in it
        uint4* wr = reinterpret_cast<uint4*>(A.cqHist + (size_t)NGC * CQW) + 4 * t;
#pragma unroll
        for (int u = 0; u < 4; u++) wr[u] = make_uint4(0u, 0u, 0u, 0u);
    }
    uint32_t s = 0;
#pragma unroll
    for (int u = 0; u < 16; u++) s += c[u];
    const uint32_t inc = wave_inclusive_scan_u32(s);
    if ((t & 63) == 63) swave[t >> 6] = inc;
    if (t == 0) sFail = 0;
    __syncthreads();
    uint32_t woff = 0, inWin = 0;
    for (int w = 0; w < 16; w++) { if (w < (t >> 6)) woff += swave[w]; inWin += swave[w]; }
    const uint32_t ex = woff + inc - s;
    const int64_t below = (int64_t)C->below[g];
    const long long lo = C->lo;
    for (int q = 0; q < nr; q++) {
        const int64_t r = ranks[q] - below;
        if (r < 0 || r >= (int64_t)inWin) { if (t == 0) sFail = 1; continue; }
        if (r >= (int64_t)ex && r < (int64_t)ex + s) {
            uint32_t left = (uint32_t)(r - ex); int u = 0;
            while (left >= c[u]) { left -= c[u]; u++; }
            sK[q] = lo + 16 * t + u;
        }
    }
    __syncthreads();
    if (t != 0) return;
    C->inWin[g] = inWin;
    if (sFail) { C->fail = 1u; D->cqFail = 1; D->fallback = 1u; return; }
    const double med = nMed == 1 ? (double)cq_value(sK[0]) : (double)median_from_two(cq_value(sK[0]), cq_value(sK[1]));
    if (g < NGC) { D->medians[g] = med; for (int k = nMed; k < nr; k++) C->kq[g][k - nMed] = (int32_t)sK[k]; }
    else D->globalMedian = med;
}
// NormalizeByGC as a function of k for bucket g (what k_cf_xform_gc does to a key)
__device__ __forceinline__ float cq_normalised(long long k, double median, double globalMedian) {
    const float x = cq_value(k);
    return median > 0 ? (float)(globalMedian * (double)x / median) : x;
}
// The genome's quartiles of the normalised counts (CanvasClean.cs:34-66) by counting once more.  The items are the counters: value = the normalised count of the slot,
// weight = the counter.  The normalised values are not on a grid, but v -> floor((v - L0) * 100) is non-decreasing, so one weighted count over CQW bins of 0.01 around the
// genome's median locates, for every rank, the bin that holds it and the rank inside that bin (k_cq_nhist); the bin's few candidates — per bucket the slots whose value
// falls into it, found by bisection because the value is monotone in the slot — are then ordered exactly by their float keys (k_cq_nresolve).  Keys below / above a
// bucket's window count as smaller / larger than everything, which k_cq_dec_f checks against the answers.  (This replaced four weighted radix passes + picks: 70 -> 20 us.)
__device__ __forceinline__ double cq_nbin_origin(double globalMedian) { return globalMedian - (double)CQW / 200.0; }
__device__ __forceinline__ long long cq_nbin(float v, double origin) { return (long long)floor(((double)v - origin) * 100.0); }
__global__ void __launch_bounds__(1024) k_cq_nhist(const CfArgs* __restrict__ AA) {
    CF_SAMPLE;
    __shared__ uint32_t lw[CQW];
    const CfSel* __restrict__ P = A.P + 2; const CleanDev* __restrict__ D = A.D; CfCq* __restrict__ C = A.cq;
    if (P->hdr[1] == 0) return;
    const int g = blockIdx.x;
    const int64_t cnt = (int64_t)D->segOff[g + 1] - (int64_t)D->segOff[g];
    if (cnt <= 0) return;
    for (int i = threadIdx.x; i < CQW; i += 1024) lw[i] = 0;
    __syncthreads();
    const double median = D->medians[g], globalMedian = D->globalMedian, origin = cq_nbin_origin(globalMedian);
    const long long lo = C->lo;
    const gptr<const uint32_t> row = as_global(A.cqHist) + (size_t)g * CQW;
    unsigned long long below = threadIdx.x == 0 ? (unsigned long long)C->below[g] : 0ull;
    for (int j = threadIdx.x; j < CQW; j += 1024) {
        const uint32_t w = row[j];
        if (!w) continue;
        const long long b = cq_nbin(cq_normalised(lo + j, median, globalMedian), origin);
        if (b < 0) below += w; else if (b < CQW) atomicAdd(&lw[b], w);
    }
    below = wave_reduce_add_u64(below);
    if ((threadIdx.x & 63) == 0 && below) atomicAdd(&C->nbelow, below);
    __syncthreads();
    uint32_t* __restrict__ all = A.cqHist + (size_t)NGC * CQW;
    for (int i = threadIdx.x; i < CQW; i += 1024) { const uint32_t v = lw[i]; if (v) atomicAdd(&all[i], v); }
}
#define CQ_NCAND 1024        // candidates of one bin (a bucket contributes about median / globalMedian slots per bin)
__global__ void __launch_bounds__(1024) k_cq_nresolve(const CfArgs* __restrict__ AA) {
    CF_SAMPLE;
    __shared__ uint32_t swave[16];
    __shared__ uint32_t sKey[CQ_NCAND], sW[CQ_NCAND];
    __shared__ unsigned int sN; __shared__ int sBin, sFail; __shared__ uint32_t sR;
    CfSel* __restrict__ P = A.P + 2; const CleanDev* __restrict__ D = A.D; const CfCq* __restrict__ C = A.cq;
    const int q = blockIdx.x, t = threadIdx.x;
    if ((uint32_t)q >= P->hdr[1]) return;
    const uint4* __restrict__ row = reinterpret_cast<const uint4*>(A.cqHist + (size_t)NGC * CQW) + 4 * t;
    uint32_t c[16];
#pragma unroll
    for (int u = 0; u < 4; u++) { const uint4 v = row[u]; c[4 * u] = v.x; c[4 * u + 1] = v.y; c[4 * u + 2] = v.z; c[4 * u + 3] = v.w; }
    uint32_t s = 0;
#pragma unroll
    for (int u = 0; u < 16; u++) s += c[u];
    const uint32_t inc = wave_inclusive_scan_u32(s);
    if ((t & 63) == 63) swave[t >> 6] = inc;
    if (t == 0) { sN = 0; sBin = -1; sFail = 0; }
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < (t >> 6); w++) woff += swave[w];
    const uint32_t ex = woff + inc - s;
    const long long r = (long long)P->qk[q] - (long long)C->nbelow;
    if (r >= (long long)ex && r < (long long)ex + s) {
        uint32_t left = (uint32_t)(r - ex); int u = 0;
        while (left >= c[u]) { left -= c[u]; u++; }
        sBin = 16 * t + u; sR = left;
    }
    __syncthreads();
    const int bin = sBin;
    if (bin < 0) { if (t == 0) P->qprefix[q] = 0ull; return; }          // the rank lies outside the bins: key 0 makes k_cq_dec_f give the sample up
    if (t < NGC) {
        const int64_t cnt = (int64_t)D->segOff[t + 1] - (int64_t)D->segOff[t];
        if (cnt > 0) {
            const double median = D->medians[t], globalMedian = D->globalMedian, origin = cq_nbin_origin(globalMedian);
            const long long lo = C->lo;
            int a = 0, b = CQW;                           // first slot whose value falls into bin `bin` or a later one
            while (a < b) { const int mid = (a + b) >> 1; if (cq_nbin(cq_normalised(lo + mid, median, globalMedian), origin) < (long long)bin) a = mid + 1; else b = mid; }
            for (int j = a; j < CQW; j++) {
                const float v = cq_normalised(lo + j, median, globalMedian);
                if (cq_nbin(v, origin) != (long long)bin) break;
                const uint32_t w = A.cqHist[(size_t)t * CQW + j];
                if (w) { const unsigned int at = atomicAdd(&sN, 1u); if (at < CQ_NCAND) { sKey[at] = key_of_float(v); sW[at] = w; } else sFail = 1; }
            }
        }
    }
    __syncthreads();
    if (sFail) { if (t == 0) P->qprefix[q] = 0ull; return; }
    const unsigned int n = sN;
    const uint32_t rq = sR;
    if ((unsigned int)t < n) {                            // the candidate whose weights [less, less + w) cover the rank inside the bin
        const uint32_t key = sKey[t];
        unsigned long long less = 0;
        for (unsigned int o = 0; o < n; o++) { const uint32_t ko = sKey[o]; if (ko < key || (ko == key && o < (unsigned int)t)) less += sW[o]; }
        if ((unsigned long long)rq >= less && (unsigned long long)rq < less + sW[t]) P->qprefix[q] = (unsigned long long)key;
    }
}
// NormalizeVarianceByGC decision (CanvasClean.cs:34-83) from the buckets' k statistics and the genome's selected keys
__global__ void __launch_bounds__(128) k_cq_dec_f(const CfArgs* __restrict__ AA) {
    CF_SAMPLE;
    __shared__ int sig, sFail;
    const CfSel* __restrict__ P = A.P + 2; CleanDev* __restrict__ D = A.D; const CfCq* __restrict__ C = A.cq;
    if (P->hdr[1] == 0) return;
    const int t = threadIdx.x;
    if (t == 0) { sig = 0; sFail = 0; }
    __syncthreads();
    const int64_t total = (int64_t)D->segOff[NGC];
    const QuartIdx gi = quartile_indices(total);
    float gv[6]; uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
    for (int k = 0; k < gi.n; k++) { const uint32_t key = (uint32_t)P->qprefix[k]; gv[k] = float_of_key(key); kmin = min(kmin, key); kmax = max(kmax, key); }
    float g1, g2, g3;
    quartiles_from_values(total, gv, g1, g2, g3);
    const float globalIQR = g3 - g1;
    const double globalMedian = D->globalMedian;
    const long long lo = C->lo;
    if (t < NGC) {
        const int64_t cnt = (int64_t)D->segOff[t + 1] - (int64_t)D->segOff[t];
        float liqr = -1.0f, med = -1.0f;
        if (cnt > 0) {
            const double median = D->medians[t];
            const QuartIdx qi = quartile_indices(cnt);
            float v[6];
            for (int k = 0; k < qi.n; k++) v[k] = cq_normalised(C->kq[t][k], median, globalMedian);
            float q1, q2, q3; quartiles_from_values(cnt, v, q1, q2, q3);
            med = q2; liqr = q3 - q1;
            // the genome's answers are right only if every key that was left out of this bucket's window lies on the side it was counted on
            const uint32_t below = C->below[t], above = (uint32_t)(cnt - (int64_t)below - (int64_t)C->inWin[t]);
            if (below && key_of_float(cq_normalised(lo, median, globalMedian)) > kmin) atomicOr(&sFail, 1);
            if (above && key_of_float(cq_normalised(lo + CQW - 1, median, globalMedian)) < kmax) atomicOr(&sFail, 1);
        }
        D->tab.localIQR[t] = liqr; D->tab.med[t] = med;
        if (t >= 10 && t < 90 && globalIQR * 2.0f < liqr) atomicAdd(&sig, 1);
    }
    __syncthreads();
    if (t == 0) {
        if (kmin == 0u || kmax == 0xFFFFFFFFu || sFail) { A.cq->fail = 1u; D->cqFail = 1; D->fallback = 1u; D->changed = 0; return; }
        D->tab.globalIQR = globalIQR; D->changed = sig > 0 ? 1 : 0;
    }
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
    float* __restrict__ count = A.S1.count; const int32_t* __restrict__ gc = A.S1.gc;
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
__global__ void __launch_bounds__(256) k_cf_flags_final(const CfArgs* __restrict__ AA) {
    CF_SAMPLE;
    __shared__ uint32_t sh[4];
    const CleanDev* __restrict__ D = A.D;
    const int64_t n = (int64_t)D->nAB;
    const int64_t base = (int64_t)blockIdx.x * CBLK;
    if (base >= n) return;
    const gptr<const int32_t> gc = as_global(A.S1.gc); const gptr<const double> dev = as_global(A.S1.dev); const gptr<uint8_t> flags = as_global(A.dFlags);
    const bool sdFilter = D->haveLocalSd && D->localSd > 5.0;
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < CBLK / 256; j++) {
        const int64_t i = base + j * 256 + threadIdx.x;
        if (i >= n) continue;
        const bool keep = D->keepGc[gc[i]] && !(sdFilter && dev[i] > 20 * 2.0);
        flags[i] = keep; c += keep;
    }
    c = wave_reduce_add_u32(c);
    if (lane_id() == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) A.dBlk[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ void __launch_bounds__(256) k_cf_scatter_final(const CfArgs* __restrict__ AA, int secondPhase) {
    CF_SAMPLE;
    __shared__ uint32_t sh[4];
    const CleanDev* __restrict__ D = A.D;
    if (D->fallback || D->bad) return;                                        // the caller's arrays stay as they were
    if (D->changed && !secondPhase) return;                                   // the variance normalisation changed the counts: the host enqueues the second NormalizeByGC, then this kernel again
    const int64_t n = (int64_t)D->nAB;
    const int64_t base = (int64_t)blockIdx.x * CBLK;
    if (base >= n) return;
    const gptr<const uint8_t> flags = as_global(A.dFlags); const GSoa src = as_global(A.S1), dst = as_global(A.caller);
    // first phase: NormalizeByGC has only been decided (k_cf_dec_e), not applied to the scratch counts — nothing between here and there reads them — so it is applied while
    // the survivors are copied out.  (When the second phase runs, clean_batch_finish applies it to the scratch counts first: NormalizeVarianceByGC works on normalised counts.)
    const bool normalise = !secondPhase && (A.flags & CANVAS_CLEAN_GCNORM) && A.P[1].hdr[1] != 0;
    __shared__ double sMed[NGC];
    if (normalise && threadIdx.x < NGC) sMed[threadIdx.x] = D->medians[threadIdx.x];
    const double globalMedian = D->globalMedian;
    __syncthreads();
    uint32_t running = A.dBlk[blockIdx.x];
    for (int j = 0; j < CBLK / 256; j++) {
        const int64_t i = base + j * 256 + threadIdx.x;
        const uint32_t f = (i < n) ? flags[i] : 0;
        const uint32_t inc = wave_inclusive_scan_u32(f);
        if (lane_id() == 63) sh[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t woff = 0, tot = 0;
        for (int k = 0; k < 4; k++) { if (k < (int)(threadIdx.x >> 6)) woff += sh[k]; tot += sh[k]; }
        if (f) {
            const uint32_t d = running + woff + inc - 1;
            const int32_t g = src.gc[i];
            float v = src.count[i];
            if (normalise) { const double median = sMed[g]; if (median > 0) v = (float)(globalMedian * (double)v / median); }        // CanvasClean.cs:190-195, applied on the way out
            dst.chr[d] = src.chr[i]; dst.start[d] = src.start[i]; dst.stop[d] = src.stop[i]; dst.gc[d] = g; dst.count[d] = v;
        }
        running += tot;
        __syncthreads();
    }
}

// ---------------------------------------------------------------- host side
struct CleanPending { int B; unsigned gxN, gxB, gxT; bool anyLsd, anyVar, anyGc, useCq; CfArgs* dArgs; CleanDev* dD; std::vector<CfArgs> h; };

static void cf_select_passes(canvas_ctx* ctx, const CfArgs* dArgs, int B, unsigned gxT, int which) {
    for (int shift = 24; shift >= 0; shift -= 8) {
        hipLaunchKernelGGL(k_cf_select_hist, dim3(gxT, B), dim3(256), 0, ctx->stream, dArgs, which, shift, shift == 24 ? 1 : 0);
        hipLaunchKernelGGL(k_cf_select_pick, dim3(CF_MAXQ, B), dim3(64), 0, ctx->stream, dArgs, which, shift == 24 ? 1 : 0);
    }
}
// NormalizeByGC on problem `which` (1: first time, 3: after the variance normalisation) and the last compaction; `args`: the whole batch or one sample's block
// the medians of NormalizeByGC on problem `which` (1: first time, 3: after the variance normalisation); apply = scale the scratch counts now (second phase) instead of in
// the last compaction (first phase)
static void cf_gc_medians(canvas_ctx* ctx, const CfArgs* args, int B, unsigned gxN, unsigned gxT, int which, int gate, bool apply) {
    hipLaunchKernelGGL(k_cf_sel_setup, dim3(1, B), dim3(128), 0, ctx->stream, args, which, 0, gate);
    cf_select_passes(ctx, args, B, gxT, which);
    hipLaunchKernelGGL(k_cf_dec_e, dim3(1, B), dim3(128), 0, ctx->stream, args, which);
    if (apply) hipLaunchKernelGGL(k_cf_apply_gc, dim3((gxN + CF_EPT - 1) / CF_EPT, B), dim3(256), 0, ctx->stream, args, which);
}

// CANVAS_CLEAN_RADIX_SELECT=1 switches the counting selects off (test hook: both ways must agree)
static inline bool clean_counting_selects() { return getenv("CANVAS_CLEAN_RADIX_SELECT") == nullptr; }
// Enqueues the whole stage for B samples on ctx->stream (no synchronisation): the CleanDev blocks arrive in ctx->pin.  clean_batch_finish waits for them.
static int32_t clean_batch_enqueue(canvas_ctx* ctx, int B, const int64_t* h_n, int32_t* const* d_chr, int32_t* const* d_start, int32_t* const* d_stop, float* const* d_count, int32_t* const* d_gc,
                                   int32_t nchr, const uint8_t* h_chr_is_autosome, uint32_t flags, int32_t min_bins_per_gc, bool useCq) {
    useCq = useCq && (flags & CANVAS_CLEAN_GCNORM);
    const size_t cqWords = useCq ? (size_t)B * (NGC + 1) * CQW : 0;
    WsSizer sz;
    sz.take<CfArgs>(B); sz.take<uint8_t>(nchr); sz.take<CleanDev>(B); sz.take<uint32_t>((size_t)B * CF_SZ_BINS); sz.take<uint32_t>((size_t)B * CF_HREP * 3 * NGC); sz.take<CfCq>(B); sz.take<uint32_t>(cqWords);
    int64_t nMax = 0; bool anyLsd = false, anyVar = false;
    for (int s = 0; s < B; s++) {
        const int64_t n = h_n[s], nW0 = n / 20 + 2, nb = nblk(n, CBLK); const size_t tilesUpper = (size_t)(n / SEL_TILE + NGC + 1);
        nMax = std::max(nMax, n);
        sz.take<int32_t>(n); sz.take<int32_t>(n); sz.take<int32_t>(n); sz.take<int32_t>(n); sz.take<float>(n); sz.take<double>(n);
        sz.take<uint8_t>(n); sz.take<uint32_t>(2 * (nb + 2)); sz.take<uint32_t>(n); sz.take<uint32_t>(n / 8 + 1024); sz.take<double>(nW0); sz.take<double>(CF_MAXRUN + 8);
        sz.take<int64_t>(CF_MAXRUN + 8); sz.take<long long>(65536); sz.take<CfSel>(CF_NPROB); sz.take<SelTile>(tilesUpper * CF_NPROB); sz.take<SelTile>((size_t)(n / CQ_TILE + NGC + 1));
    }
    int32_t rc = canvas_ws_reserve(ctx, sz.off + 8192); if (rc) return rc;
    const size_t histPer = (size_t)CF_MAXQ * 1024 * SEL_REP, histBytes = histPer * (size_t)B;
    if (histBytes > ctx->sel_hist_bytes) {
        if (ctx->sel_hist) { CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); CANVAS_HIP_TRY(ctx, hipFree(ctx->sel_hist)); ctx->sel_hist = nullptr; ctx->sel_hist_bytes = 0; }
        CANVAS_HIP_TRY(ctx, hipMalloc(&ctx->sel_hist, histBytes)); ctx->sel_hist_bytes = histBytes;
        CANVAS_HIP_TRY(ctx, hipMemsetAsync(ctx->sel_hist, 0, ctx->sel_hist_bytes, ctx->stream));       // zero once: k_select_pick clears every row it has read
    }
    WsCarver ws(ctx->ws);
    // two adjacent groups: what the host sends (one copy) and what starts as zero (one memset)
    CfArgs* dArgs = ws.take<CfArgs>(B); uint8_t* dIsAuto = ws.take<uint8_t>(nchr);
    CleanDev* dD = ws.take<CleanDev>(B); uint32_t* dSz = ws.take<uint32_t>((size_t)B * CF_SZ_BINS); uint32_t* dRepl = ws.take<uint32_t>((size_t)B * CF_HREP * 3 * NGC); CfCq* dCq = ws.take<CfCq>(B); uint32_t* dCqHist = ws.take<uint32_t>(cqWords);
    CleanPending pend; pend.useCq = useCq; pend.B = B; pend.dArgs = dArgs; pend.dD = dD; pend.h.resize(B);
    unsigned gxT = 1, gxTcq = 1;
    for (int s = 0; s < B; s++) {
        const int64_t n = h_n[s], nW0 = n / 20 + 2; const int nb = (int)nblk(n, CBLK); const unsigned tilesUpper = (unsigned)(n / SEL_TILE + NGC + 1);
        CfArgs& A = pend.h[s];
        A.n = n; A.nb = nb; A.nchr = nchr; A.minBinsPerGc = min_bins_per_gc; A.flags = flags; A.tilesUpper = tilesUpper;
        A.wantLsd = ((flags & CANVAS_CLEAN_LOCALSD) && n >= 50000) ? 1 : 0;
        A.doSize = (flags & CANVAS_CLEAN_FILTSIZE) ? 1 : 0; A.doOutlier = (flags & CANVAS_CLEAN_OUTLIERS) ? 1 : 0;
        A.caller = Soa{d_chr[s], d_start[s], d_stop[s], d_gc[s], d_count[s], nullptr};
        A.S1.chr = ws.take<int32_t>(n); A.S1.start = ws.take<int32_t>(n); A.S1.stop = ws.take<int32_t>(n); A.S1.gc = ws.take<int32_t>(n); A.S1.count = ws.take<float>(n); A.S1.dev = ws.take<double>(n);
        A.dFlags = ws.take<uint8_t>(n); A.dBlk = ws.take<uint32_t>(2 * (nb + 2)); A.keysG = ws.take<uint32_t>(n); A.szOverCap = (uint32_t)(n / 8 + 1024); A.padA = 0; A.szOver = ws.take<uint32_t>(A.szOverCap);
        A.dSd = ws.take<double>(nW0); A.dRunMad = ws.take<double>(CF_MAXRUN + 8); A.dRunStart = ws.take<int64_t>(CF_MAXRUN + 8); A.dPos = ws.take<long long>(65536);
        A.P = ws.take<CfSel>(CF_NPROB); A.tiles = ws.take<SelTile>((size_t)tilesUpper * CF_NPROB);
        A.cqTiles = ws.take<SelTile>((size_t)(n / CQ_TILE + NGC + 1)); A.cq = dCq + s; A.cqHist = dCqHist + (size_t)s * (NGC + 1) * CQW;
        gxTcq = std::max(gxTcq, (unsigned)(n / CQ_TILE + NGC + 1));
        A.isAuto = dIsAuto; A.repl = dRepl + (size_t)s * CF_HREP * 3 * NGC; A.szHist = dSz + (size_t)s * CF_SZ_BINS; A.D = dD + s; A.hist = (uint32_t*)((char*)ctx->sel_hist + histPer * (size_t)s);
        gxT = std::max(gxT, tilesUpper);
        anyLsd = anyLsd || A.wantLsd; anyVar = anyVar || (A.wantLsd && n > 500000);
    }
    const unsigned gxN = (unsigned)nblk(nMax, 256), gxB = (unsigned)nblk(nMax, CBLK);
    pend.gxN = gxN; pend.gxB = gxB; pend.gxT = gxT; pend.anyLsd = anyLsd; pend.anyVar = anyVar; pend.anyGc = (flags & CANVAS_CLEAN_GCNORM) != 0;
    ProfScope psTotal(ctx, "clean_total");
    {
        std::vector<char> up((size_t)((char*)(dIsAuto + nchr) - (char*)dArgs), 0);
        memcpy(up.data(), pend.h.data(), (size_t)B * sizeof(CfArgs)); memcpy(up.data() + ((char*)dIsAuto - (char*)dArgs), h_chr_is_autosome, nchr);
        rc = canvas_h2d_small(ctx, dArgs, up.data(), up.size()); if (rc) return rc;
    }
    CANVAS_HIP_TRY(ctx, hipMemsetAsync(dD, 0, (size_t)((char*)(dCqHist + cqWords) - (char*)dD), ctx->stream));      // CleanDev blocks, size counters, CfCq blocks, value counters
    // ---- RemoveBigBins threshold (CanvasClean.cs:328-348): the 98th percentile of the bin sizes from exact per-size counts, left on the device
    if (flags & CANVAS_CLEAN_FILTSIZE) {
        hipLaunchKernelGGL(k_cf_size_hist, dim3(CF_SZ_GRID, B), dim3(1024), 0, ctx->stream, dArgs);
        hipLaunchKernelGGL(k_cf_size_pick, dim3(1, B), dim3(1024), 0, ctx->stream, dArgs);        // sizeOn stays 0 (the memset of the CleanDev blocks) without the filter
    }
    // ---- size filter + outlier filter: one compaction, caller -> S1
    // (a one-bin-per-thread variant of the flag kernel was measured: 68 us against 53 us for the staged one)
    hipLaunchKernelGGL(k_cf_flags_ab, dim3(gxB, B), dim3(256), 0, ctx->stream, dArgs);
    hipLaunchKernelGGL(k_cf_scan_blocks, dim3(1, B), dim3(1024), 0, ctx->stream, dArgs, 0);
    hipLaunchKernelGGL(k_cf_dec_gc, dim3(1, B), dim3(128), 0, ctx->stream, dArgs);             // the GC strip decision (the histogram of the survivors came with the flags)
    hipLaunchKernelGGL(k_cf_scatter_ab, dim3(gxB, B), dim3(256), 0, ctx->stream, dArgs);
    // ---- local SD (CanvasClean.cs:243-298): window SDs and chromosome runs on the main stream, the per-run MAD on the side stream
    if (anyLsd) hipLaunchKernelGGL(k_cf_local_sd, dim3((unsigned)nblk(nMax / 20 + 2, 256), B), dim3(256), 0, ctx->stream, dArgs);
    hipLaunchKernelGGL(k_cf_runs_build, dim3(1, B), dim3(CF_MAXRUN), 0, ctx->stream, dArgs);
    if (anyLsd) {
        rc = canvas_side_init(ctx); if (rc) return rc;
        if (!ctx->side_ev2) CANVAS_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->side_ev2, hipEventDisableTiming));
        CANVAS_HIP_TRY(ctx, hipEventRecord(ctx->side_ev, ctx->stream));
        CANVAS_HIP_TRY(ctx, hipStreamWaitEvent(ctx->side, ctx->side_ev, 0));
        hipLaunchKernelGGL(k_cf_run_mad, dim3(CF_MAXRUN, B), dim3(1024), 0, ctx->side, dArgs);
    }
    // ---- NormalizeByGC on the grouped keys the compaction left
    if (useCq) {
        // counting selects (CfCq): one sweep + one pick for every median and every bucket's quartiles; the genome's quartiles of the normalised counts by a second
        // (weighted) count over the counters and an exact resolve of the bin each rank falls into
        hipLaunchKernelGGL(k_cq_setup, dim3(1, B), dim3(128), 0, ctx->stream, dArgs);
        hipLaunchKernelGGL(k_cq_hist, dim3(gxTcq, B), dim3(1024), 0, ctx->stream, dArgs);
        hipLaunchKernelGGL(k_cq_pick, dim3(NGC + 1, B), dim3(1024), 0, ctx->stream, dArgs);
        if (anyVar) {
            hipLaunchKernelGGL(k_cq_nhist, dim3(NGC, B), dim3(1024), 0, ctx->stream, dArgs);
            hipLaunchKernelGGL(k_cq_nresolve, dim3(6, B), dim3(1024), 0, ctx->stream, dArgs);
            hipLaunchKernelGGL(k_cq_dec_f, dim3(1, B), dim3(128), 0, ctx->stream, dArgs);
        }
    } else if (flags & CANVAS_CLEAN_GCNORM) {
        cf_gc_medians(ctx, dArgs, B, gxN, gxT, 1, 0, false);
        if (anyVar) {
            // NormalizeVarianceByGC (CanvasClean.cs:512-519): quartiles of the normalised counts; if it changes anything, NormalizeByGC once more
            hipLaunchKernelGGL(k_cf_sel_setup, dim3(1, B), dim3(128), 0, ctx->stream, dArgs, 2, 1, 1);
            hipLaunchKernelGGL(k_cf_xform_gc, dim3((gxN + CF_EPT - 1) / CF_EPT, B), dim3(256), 0, ctx->stream, dArgs, 2);
            cf_select_passes(ctx, dArgs, B, gxT, 2);
            hipLaunchKernelGGL(k_cf_dec_f, dim3(1, B), dim3(128), 0, ctx->stream, dArgs, 2);
            // ... which it rarely does: the last compaction below is enqueued on the assumption that it does not; when k_cf_dec_f says it did, that compaction does nothing for
            // the sample and clean_batch_finish enqueues the variance scaling, the second NormalizeByGC and the compaction for it (one more synchronisation, in that case only)
        }
    }
    // ---- local-SD average, last compaction into the caller's arrays
    if (anyLsd) {
        CANVAS_HIP_TRY(ctx, hipEventRecord(ctx->side_ev2, ctx->side));
        CANVAS_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->side_ev2, 0));
    }
    hipLaunchKernelGGL(k_cf_lsd_avg, dim3(1, B), dim3(64), 0, ctx->stream, dArgs);
    hipLaunchKernelGGL(k_cf_flags_final, dim3(gxB, B), dim3(256), 0, ctx->stream, dArgs);
    hipLaunchKernelGGL(k_cf_scan_blocks, dim3(1, B), dim3(1024), 0, ctx->stream, dArgs, 1);
    hipLaunchKernelGGL(k_cf_scatter_final, dim3(gxB, B), dim3(256), 0, ctx->stream, dArgs, 0);
    rc = canvas_pin_reserve(ctx, (size_t)B * sizeof(CleanDev)); if (rc) return rc;
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(ctx->pin, dD, (size_t)B * sizeof(CleanDev), hipMemcpyDeviceToHost, ctx->stream));
    ctx->clean_batch = std::make_shared<CleanPending>(std::move(pend));
    return CANVAS_OK;
}
// Waits for the batch.  handled[s] = false: the host-driven path has to take sample s over (nothing of it was modified).  Per-sample outputs as canvas_clean2.
static int32_t clean_batch_finish(canvas_ctx* ctx, double* h_local_sd_out, int64_t* h_n_out, int32_t* h_info, char* handled) {
    std::shared_ptr<CleanPending> pp = std::static_pointer_cast<CleanPending>(ctx->clean_batch);
    ctx->clean_batch.reset();
    if (!pp) CANVAS_FAIL(ctx, CANVAS_ERR_HIP, "canvas_clean: no pending batch");
    const CleanPending& q = *pp;
    const int B = q.B;
    for (int s = 0; s < B; s++) handled[s] = 0;
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    CANVAS_HIP_TRY(ctx, hipGetLastError());
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
        hipLaunchKernelGGL(k_cf_scan_blocks, dim3(1, 1), dim3(1024), 0, ctx->stream, a, 1);
        hipLaunchKernelGGL(k_cf_scatter_final, dim3(gxB, 1), dim3(256), 0, ctx->stream, a, 1);
        again = true;
    }
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
