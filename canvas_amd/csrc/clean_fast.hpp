// CanvasClean without host round trips (included by clean.hip): CanvasClean.Main (CanvasClean/CanvasClean.cs:415-533) for the MedianByGC flavour with the default
// weighted-median setting (-w >= 100, which makes every GC bucket that survives RemoveBinsWithExtremeGC hold >= 100 autosomal bins: CanvasClean.cs:207-237,178-187).
//
// Every decision the reference takes once per file — the size threshold, the GC strip, the per-GC medians, whether the variance normalisation applies and changes
// anything, the local-SD filter — is taken by a one-workgroup kernel that leaves its result in device memory (CleanDev); the kernels that follow read it from there
// and the kernels of a branch that is not taken find an empty problem and return.  The host enqueues the whole stage in one go and synchronises once, at the end.
//   caller's arrays --(size filter + outlier filter: ONE compaction)--> scratch SoA  ...in-place normalisation...  --(GC strip + local-SD filter: ONE compaction)--> caller's arrays
// The caller's arrays are not written before the last kernel, so a case this path does not cover (more than CF_MAXRUN chromosome runs) is detected on the device, leaves the
// input intact and is handed to the host-driven path of clean.hip.  All per-bin arithmetic and all order statistics are the ones of that path: results are bit-identical.
#pragma once

#define CF_MAXQ 640          // 6 + 6 * 101 quartile queries (variance normalisation); 2 + 2 * 101 median queries
#define CF_MAXRUN 1024       // chromosome runs of the bin list handled on the device

struct CleanDev {
    unsigned long long nAB;          // bins after RemoveBigBins + RemoveOutliers
    unsigned long long nFinal;       // bins after the GC strip and the local-SD filter
    unsigned int nA;                 // bins after RemoveBigBins alone
    unsigned int bad;                // a gc outside 0..100 or a chromosome index outside the table
    unsigned int fallback;           // not covered on the device: the host-driven path takes over (the caller's arrays are untouched)
    unsigned int nRunRec;
    uint32_t hist[2 * NGC];          // [0..100] autosomal bins per GC, [101..201] the other bins (after the first compaction)
    uint32_t segOff[NGC + 1];        // grouped autosomal bins of the kept GC values
    uint32_t cursor[NGC];
    uint8_t keepGc[NGC + 3];
    long long kept;                  // bins of any chromosome that survive the GC strip
    int gcActive, haveLocalSd, varActive, changed, nruns, pad0;
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

// ---------------------------------------------------------------- RemoveBigBins + RemoveOutliers in one pass over the caller's arrays
// keepA(j) = size <= threshold (CanvasClean.cs:349-352); RemoveOutliers (:387-413) looks at the neighbours in the list RemoveBigBins left, i.e. at the nearest
// bins on either side that pass keepA.  Also the range check of gc / chr, the count after the size filter, and the block counts of the compaction.
__global__ void __launch_bounds__(256) k_cf_flags_ab(const int32_t* __restrict__ chr, const int32_t* __restrict__ start, const int32_t* __restrict__ stop, const int32_t* __restrict__ gc,
                                                     const float* __restrict__ count, int64_t n, int nchr, const unsigned long long* __restrict__ dKey, int doOutlier,
                                                     uint8_t* __restrict__ flags, uint32_t* __restrict__ blockCnt, CleanDev* __restrict__ D) {
    __shared__ uint32_t sh[8];
    __shared__ uint8_t sA[CBLK];                          // keepA, chromosome and count of this block's bins: the neighbour search reads them from LDS
    __shared__ int32_t sChr[CBLK];
    __shared__ float sCnt[CBLK];
    const bool doSize = dKey != nullptr;
    const int32_t thresh = doSize ? (int32_t)((uint32_t)dKey[0] ^ 0x80000000u) : 0;
    const int64_t base = (int64_t)blockIdx.x * CBLK;
    uint32_t nKeep = 0, nSize = 0, bad = 0;
#pragma unroll
    for (int j = 0; j < CBLK / 256; j++) {
        const int64_t i = base + j * 256 + threadIdx.x;
        uint8_t a = 0; int32_t c = -1; float v = 0.0f;
        if (i < n) {
            c = chr[i]; v = count[i];
            if ((uint32_t)gc[i] > 100u || (uint32_t)c >= (uint32_t)nchr) bad = 1;
            a = (!doSize || (stop[i] - start[i]) <= thresh) ? 1 : 0;
        }
        sA[j * 256 + threadIdx.x] = a; sChr[j * 256 + threadIdx.x] = c; sCnt[j * 256 + threadIdx.x] = v;
        nSize += a;
    }
    __syncthreads();
    const int64_t blockEnd = base + CBLK;
    auto keepA = [&](int64_t j) -> bool { return (j >= base && j < blockEnd) ? sA[j - base] != 0 : (!doSize || (stop[j] - start[j]) <= thresh); };
    auto chrAt = [&](int64_t j) -> int32_t { return (j >= base && j < blockEnd) ? sChr[j - base] : chr[j]; };
    auto cntAt = [&](int64_t j) -> float { return (j >= base && j < blockEnd) ? sCnt[j - base] : count[j]; };
#pragma unroll 2
    for (int j = 0; j < CBLK / 256; j++) {
        const int64_t i = base + j * 256 + threadIdx.x;
        if (i >= n) continue;
        bool keep = sA[j * 256 + threadIdx.x] != 0;
        if (keep && doOutlier) {
            const int32_t c = sChr[j * 256 + threadIdx.x];
            int64_t p = i - 1, q = i + 1;
            while (p >= 0 && !keepA(p)) p--;
            while (q < n && !keepA(q)) q++;
            const bool hasPrev = p >= 0, hasNext = q < n;
            const bool prevSame = hasPrev && chrAt(p) == c, nextSame = hasNext && chrAt(q) == c;
            if ((hasPrev && !prevSame) && (hasNext && !nextSame)) keep = false;
            else {
                const float v = sCnt[j * 256 + threadIdx.x];
                keep = (prevSame && !sig_diff(v, cntAt(p))) || (nextSame && !sig_diff(v, cntAt(q))) || (!hasPrev && !hasNext);
            }
        }
        flags[i] = keep;
        nKeep += keep;
    }
    nKeep = wave_reduce_add_u32(nKeep); nSize = wave_reduce_add_u32(nSize);
    if (lane_id() == 0) { sh[threadIdx.x >> 6] = nKeep; sh[4 + (threadIdx.x >> 6)] = nSize; }
    if (bad) D->bad = 1u;
    __syncthreads();
    if (threadIdx.x == 0) { blockCnt[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3]; atomicAdd(&D->nA, sh[4] + sh[5] + sh[6] + sh[7]); }
}
// the compaction itself: caller's arrays -> scratch SoA, CountDeviation = -1 (GenomicBin.cs:83), and the GC histogram of what survives (CanvasClean.cs:207-223)
__global__ void __launch_bounds__(256) k_cf_scatter_ab(const uint8_t* __restrict__ flags, const uint32_t* __restrict__ blockOff, int64_t n, Soa src, Soa dst, const uint8_t* __restrict__ isAuto,
                                                       int nchr, CleanDev* __restrict__ D) {
    __shared__ uint32_t sh[4];
    __shared__ uint32_t lh[2 * NGC];
    if (threadIdx.x < 2 * NGC) lh[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * CBLK;
    uint32_t running = blockOff[blockIdx.x];
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
            // out-of-range input is reported through D->bad (k_cf_flags_ab) and nothing is returned; the scratch copy holds clamped values so that no later kernel indexes past a table
            const int32_t c0 = src.chr[i], g0 = src.gc[i];
            const int32_t g = (uint32_t)g0 > 100u ? 100 : g0, c = (uint32_t)c0 >= (uint32_t)nchr ? 0 : c0;
            dst.chr[d] = c; dst.start[d] = src.start[i]; dst.stop[d] = src.stop[i]; dst.gc[d] = g; dst.count[d] = src.count[i]; dst.dev[d] = -1.0;
            atomicAdd(&lh[(isAuto[c] ? 0 : NGC) + g], 1u);
        }
        running += tot;
        __syncthreads();
    }
    if (threadIdx.x < 2 * NGC && lh[threadIdx.x]) atomicAdd(&D->hist[threadIdx.x], lh[threadIdx.x]);
}

// ---------------------------------------------------------------- chromosome runs of the window SDs (GetLocalStandardDeviationAverage, CanvasClean.cs:243-258)
// one workgroup: sorts the (position << 20 | chromosome) records of k_run_bounds, derives the runs of windows per chromosome exactly as local_sd_begin does on the host
__global__ void __launch_bounds__(1024) k_cf_runs_build(const long long* __restrict__ recs, const unsigned int* __restrict__ nrecDev, int64_t* __restrict__ runStart, CleanDev* __restrict__ D, int wantLocalSd) {
    __shared__ long long s[CF_MAXRUN];
    const unsigned long long nAB = D->nAB;
    const int have = wantLocalSd && nAB >= 50000ull;                       // CanvasClean.cs:483-486
    if (threadIdx.x == 0) D->haveLocalSd = have;
    if (!have) { if (threadIdx.x == 0) D->nruns = 0; return; }
    const unsigned int nb = *nrecDev;
    if (nb > CF_MAXRUN) { if (threadIdx.x == 0) { D->fallback = 1u; D->nruns = 0; } return; }
    const int t = threadIdx.x;
    s[t] = t < (int)nb ? recs[t] : 0x7FFFFFFFFFFFFFFFll;
    __syncthreads();
    for (int k = 2; k <= CF_MAXRUN; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int ixj = t ^ j;
            if (ixj > t) { const long long a = s[t], b = s[ixj]; const bool up = (t & k) == 0; if ((a > b) == up) { s[t] = b; s[ixj] = a; } }
            __syncthreads();
        }
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
__global__ void k_cf_lsd_avg(const double* __restrict__ runMad, CleanDev* __restrict__ D) {
    if (threadIdx.x || blockIdx.x) return;
    if (!D->haveLocalSd) { D->localSd = -1.0; return; }
    double s = 0;
    for (int r = 0; r < D->nruns; r++) s += runMad[r];                     // List<double>.Average(): sequential sum / count
    D->localSd = s / (double)D->nruns;
}

// ---------------------------------------------------------------- RemoveBinsWithExtremeGC decision (CanvasClean.cs:207-237) and what follows from it
__global__ void __launch_bounds__(128) k_cf_dec_gc(uint32_t flags, int minBinsPerGc, CleanDev* __restrict__ D) {
    __shared__ uint32_t hA[NGC], hO[NGC], so[NGC + 1];
    __shared__ uint8_t kp[NGC];
    __shared__ long long sKept; __shared__ int sActive;
    const int t = threadIdx.x;
    const long long nAB = (long long)D->nAB;
    if (t < NGC) { hA[t] = D->hist[t]; hO[t] = D->hist[NGC + t]; }
    __syncthreads();
    if (t == 0) {                                        // 101-element loops over LDS: the decision the reference takes once per file
        sActive = 0; sKept = nAB;
        for (int i = 0; i < NGC; i++) kp[i] = 1;
        for (int i = 0; i <= NGC; i++) so[i] = 0;
        if ((flags & CANVAS_CLEAN_GCNORM) && nAB > 0) {
            double totalCount = 0;
            for (int i = 0; i < NGC; i++) totalCount += hA[i];
            const int averageCountPerGC = max(minBinsPerGc, (int)(totalCount / NGC));
            const int threshold = min(100, averageCountPerGC);
            long long kept = 0;
            for (int i = 0; i < NGC; i++) { const bool k = (int)hA[i] >= threshold; kp[i] = k; if (k) kept += (long long)hA[i] + (long long)hO[i]; }
            if (kept <= 0) { for (int i = 0; i < NGC; i++) kp[i] = 1; }            // "proceed without GC correction" (CanvasClean.cs:500-505)
            else {
                sKept = kept; sActive = 1;
                uint32_t acc = 0;
                for (int i = 0; i < NGC; i++) { so[i] = acc; if (kp[i]) acc += hA[i]; }
                so[NGC] = acc;
            }
        }
    }
    __syncthreads();
    if (t < NGC) { D->keepGc[t] = kp[t]; D->cursor[t] = 0; D->medians[t] = 0.0; D->segOff[t] = so[t]; }
    if (t == 0) {
        D->segOff[NGC] = so[NGC]; D->kept = sKept; D->gcActive = sActive; D->changed = 0;
        D->varActive = (sActive && D->haveLocalSd && sKept > 500000) ? 1 : 0;                    // CanvasClean.cs:512-519
    }
}
// grouped keys of the autosomal bins with a kept GC value (order inside a bucket is irrelevant: only order statistics are taken)
__global__ void __launch_bounds__(256) k_cf_group_keys(const int32_t* __restrict__ chr, const int32_t* __restrict__ gc, const float* __restrict__ count, const uint8_t* __restrict__ isAuto,
                                                       int64_t nUpper, CleanDev* __restrict__ D, uint32_t* __restrict__ keysG) {
    __shared__ uint32_t lcnt[NGC], lbase[NGC];
    if (!D->gcActive) return;
    const int64_t n = (int64_t)D->nAB;
    const int64_t base = (int64_t)blockIdx.x * CBLK;
    if (base >= n) return;
    if (threadIdx.x < NGC) lcnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t myRank[CBLK / 256]; int myGc[CBLK / 256]; uint32_t myKey[CBLK / 256];
#pragma unroll
    for (int j = 0; j < CBLK / 256; j++) {
        const int64_t i = base + j * 256 + threadIdx.x;
        myGc[j] = -1;
        if (i < n) { const int g = gc[i]; if (isAuto[chr[i]] && D->keepGc[g]) { myGc[j] = g; myKey[j] = key_of_float(count[i]); myRank[j] = atomicAdd(&lcnt[g], 1u); } }
    }
    __syncthreads();
    if (threadIdx.x < NGC && lcnt[threadIdx.x]) lbase[threadIdx.x] = atomicAdd(&D->cursor[threadIdx.x], lcnt[threadIdx.x]);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < CBLK / 256; j++) if (myGc[j] >= 0) keysG[D->segOff[myGc[j]] + lbase[myGc[j]] + myRank[j]] = myKey[j];
}
// the select problem over the grouped keys: mode 0 = medians (NormalizeByGC, CanvasClean.cs:163-189), mode 1 = quartiles (NormalizeVarianceByGC, :34-66); genome + every kept bucket.
// gate: which CleanDev flag switches the problem on (0 gcActive, 1 varActive, 2 changed)
__global__ void __launch_bounds__(128) k_cf_sel_setup(int mode, int gate, const CleanDev* __restrict__ D, CfSel* __restrict__ P, SelTile* __restrict__ tiles) {
    __shared__ uint32_t so[NGC + 1], tileBase[NGC + 1];
    __shared__ int qFirst[NGC + 2];                     // first query of slot s (slot NGC = the genome, placed FIRST: queries 0 .. nrG-1), prefix sums
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
    if (t <= NGC) qFirst[t] = myN;                      // counts first
    __syncthreads();
    if (t == 0) {
        const int nG = qFirst[NGC];
        int acc = nG; uint32_t tacc = 0;
        for (int s2 = 0; s2 < NGC; s2++) { const int c = qFirst[s2]; qFirst[s2] = acc; acc += c; tileBase[s2] = tacc; tacc += (so[s2 + 1] - so[s2] + SEL_TILE - 1) / SEL_TILE; }
        qFirst[NGC] = 0; qFirst[NGC + 1] = nG;
        tileBase[NGC] = tacc;
        P->hdr[0] = tacc; P->hdr[1] = (uint32_t)acc;
    }
    __syncthreads();
    const int nG = qFirst[NGC + 1];
    if (t <= NGC) {
        const int f = qFirst[t];
        P->first[t] = myN > 0 ? f : -1;
        for (int k = 0; k < myN; k++) { P->qk[f + k] = (unsigned long long)myRanks[k]; P->qprefix[f + k] = 0ull; }
    }
    if (t < NGC) {
        SelSegQ Q; Q.nq = 0;
        if (so[t + 1] > so[t]) { for (int k = 0; k < nG; k++) Q.q[Q.nq++] = k; for (int k = 0; k < myN; k++) Q.q[Q.nq++] = qFirst[t] + k; }
        P->segq[t] = Q;
        uint32_t k = tileBase[t];
        for (int64_t b = so[t]; b < (int64_t)so[t + 1]; b += SEL_TILE) tiles[k++] = SelTile{t, b, min<int64_t>(b + SEL_TILE, (int64_t)so[t + 1])};
    }
}
// NormalizeByGC decision: genome median and per-GC medians from the selected keys (CanvasClean.cs:170-189)
__global__ void __launch_bounds__(128) k_cf_dec_e(const CfSel* __restrict__ P, CleanDev* __restrict__ D) {
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
__global__ void __launch_bounds__(256) k_cf_apply_gc(float* __restrict__ count, const int32_t* __restrict__ gc, int64_t nUpper, const CleanDev* __restrict__ D, const CfSel* __restrict__ P) {
    if (P->hdr[1] == 0) return;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)D->nAB) return;
    const double median = D->medians[gc[i]];
    if (median > 0) count[i] = (float)(D->globalMedian * (double)count[i] / median);         // CanvasClean.cs:190-195
}
// the same normalisation applied to the grouped keys (the order statistics of the next step are taken from the updated counts)
__device__ __forceinline__ int cf_bucket_of(const uint32_t* __restrict__ segOff, uint32_t p) {
    int lo = 0, hi = NGC - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (segOff[mid] <= p) lo = mid; else hi = mid - 1; }
    return lo;
}
__global__ void __launch_bounds__(256) k_cf_xform_gc(uint32_t* __restrict__ keysG, const CleanDev* __restrict__ D, const CfSel* __restrict__ nextProblem) {
    if (nextProblem->hdr[1] == 0) return;
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= D->segOff[NGC]) return;
    const double median = D->medians[cf_bucket_of(D->segOff, p)];
    if (median > 0) keysG[p] = key_of_float((float)(D->globalMedian * (double)float_of_key(keysG[p]) / median));
}
// NormalizeVarianceByGC decision (CanvasClean.cs:34-83)
__global__ void __launch_bounds__(128) k_cf_dec_f(const CfSel* __restrict__ P, CleanDev* __restrict__ D) {
    __shared__ int sig;
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
__global__ void __launch_bounds__(256) k_cf_apply_var(float* __restrict__ count, const int32_t* __restrict__ gc, int64_t nUpper, const CleanDev* __restrict__ D) {
    if (!D->changed) return;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)D->nAB) return;
    const int g = gc[i];
    const float globalIQR = D->tab.globalIQR, scaledLocalIqr = D->tab.localIQR[g] * 0.8f;
    if (globalIQR >= scaledLocalIqr) return;
    const float iqrRatio = scaledLocalIqr / globalIQR, m = D->tab.med[g];
    count[i] = m + (count[i] - m) / iqrRatio;                                                // CanvasClean.cs:84-94
}
__global__ void __launch_bounds__(256) k_cf_xform_var(uint32_t* __restrict__ keysG, const CleanDev* __restrict__ D) {
    if (!D->changed) return;
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= D->segOff[NGC]) return;
    const int g = cf_bucket_of(D->segOff, p);
    const float globalIQR = D->tab.globalIQR, scaledLocalIqr = D->tab.localIQR[g] * 0.8f;
    if (globalIQR >= scaledLocalIqr) return;
    const float iqrRatio = scaledLocalIqr / globalIQR, m = D->tab.med[g];
    keysG[p] = key_of_float(m + (float_of_key(keysG[p]) - m) / iqrRatio);
}

// ---------------------------------------------------------------- last compaction: GC strip (CanvasClean.cs:226-235) + RemoveBinsWithExtremeLocalSD (:308-322) -> caller's arrays
__global__ void __launch_bounds__(256) k_cf_flags_final(const int32_t* __restrict__ gc, const double* __restrict__ dev, int64_t nUpper, const CleanDev* __restrict__ D,
                                                        uint8_t* __restrict__ flags, uint32_t* __restrict__ blockCnt) {
    __shared__ uint32_t sh[4];
    const int64_t n = (int64_t)D->nAB;
    const bool sdFilter = D->haveLocalSd && D->localSd > 5.0;
    const int64_t base = (int64_t)blockIdx.x * CBLK;
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
    if (threadIdx.x == 0) blockCnt[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ void __launch_bounds__(256) k_cf_scatter_final(const uint8_t* __restrict__ flags, const uint32_t* __restrict__ blockOff, int64_t nUpper, Soa src, Soa dst, const CleanDev* __restrict__ D, int secondPhase) {
    __shared__ uint32_t sh[4];
    if (D->fallback || D->bad) return;                                        // the caller's arrays stay as they were
    if (D->changed && !secondPhase) return;                                   // the variance normalisation changed the counts: the host enqueues the second NormalizeByGC, then this kernel again
    const int64_t n = (int64_t)D->nAB;
    const int64_t base = (int64_t)blockIdx.x * CBLK;
    if (base >= n) return;
    uint32_t running = blockOff[blockIdx.x];
    for (int j = 0; j < CBLK / 256; j++) {
        const int64_t i = base + j * 256 + threadIdx.x;
        const uint32_t f = (i < n) ? flags[i] : 0;
        const uint32_t inc = wave_inclusive_scan_u32(f);
        if (lane_id() == 63) sh[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t woff = 0, tot = 0;
        for (int k = 0; k < 4; k++) { if (k < (int)(threadIdx.x >> 6)) woff += sh[k]; tot += sh[k]; }
        if (f) { const uint32_t d = running + woff + inc - 1; dst.chr[d] = src.chr[i]; dst.start[d] = src.start[i]; dst.stop[d] = src.stop[i]; dst.gc[d] = src.gc[i]; dst.count[d] = src.count[i]; }
        running += tot;
        __syncthreads();
    }
}
// k_scan_blocks over a device-side element count
__global__ void __launch_bounds__(1024) k_cf_scan_blocks(uint32_t* __restrict__ blockCnt, const CleanDev* __restrict__ D, unsigned long long* __restrict__ total) {
    __shared__ uint32_t sh[17];
    const int nblocks = (int)(((int64_t)D->nAB + CBLK - 1) / CBLK);
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
    if (threadIdx.x == 0) *total = carry;
}

// the four radix passes of a device-built select problem (grids are upper bounds: tiles <= n / SEL_TILE + NGC + 1, queries <= CF_MAXQ)
static void cf_select_passes(canvas_ctx* ctx, const uint32_t* keysG, SelTile* dTiles, CfSel* P, int64_t nUpper) {
    const unsigned tilesUpper = (unsigned)(nUpper / SEL_TILE + NGC + 1);
    uint32_t* dHist = (uint32_t*)ctx->sel_hist;
    for (int shift = 24; shift >= 0; shift -= 8) {
        hipLaunchKernelGGL((k_select_hist<uint32_t>), dim3(tilesUpper), dim3(256), 0, ctx->stream, keysG, dTiles, P->segq, P->qprefix, shift, shift == 24 ? 1 : 0, dHist, CF_MAXQ, P->hdr);
        hipLaunchKernelGGL(k_select_pick, dim3(CF_MAXQ), dim3(64), 0, ctx->stream, dHist, P->qprefix, P->qk, CF_MAXQ, shift == 24 ? 1 : 0, P->hdr);
    }
}

struct CleanPending { int64_t n; int nb; unsigned tilesUpper; Soa S1, caller; uint8_t* dFlags; uint32_t* dBlk; uint32_t* keysG; SelTile* dTiles; CfSel* P; CleanDev* D; };
// Enqueues the whole stage on ctx->stream (no synchronisation): the CleanDev block arrives in ctx->pin.  clean_device_driven_finish waits for it.
static int32_t clean_device_driven_enqueue(canvas_ctx* ctx, int64_t n, int32_t* d_chr, int32_t* d_start, int32_t* d_stop, float* d_count, int32_t* d_gc, int32_t nchr, const uint8_t* h_chr_is_autosome,
                                           uint32_t flags, int32_t min_bins_per_gc) {
    const int64_t nW0 = n / 20 + 2;
    const int nb = (int)nblk(n, CBLK);
    const unsigned tilesUpper = (unsigned)(n / SEL_TILE + NGC + 1);
    WsSizer sz;
    sz.take<int32_t>(n); sz.take<int32_t>(n); sz.take<int32_t>(n); sz.take<int32_t>(n); sz.take<float>(n); sz.take<double>(n);
    sz.take<uint8_t>(n); sz.take<uint32_t>(2 * (nb + 2)); sz.take<uint32_t>(n); sz.take<uint32_t>(n); sz.take<uint8_t>(nchr); sz.take<double>(nW0); sz.take<double>(CF_MAXRUN + 8);
    sz.take<int64_t>(CF_MAXRUN + 8); sz.take<long long>(65536); sz.take<CleanDev>(1); sz.take<CfSel>(3); sz.take<SelTile>((size_t)tilesUpper * 3);
    int32_t rc = canvas_ws_reserve(ctx, sz.off + 8192); if (rc) return rc;
    const size_t histBytes = (size_t)CF_MAXQ * 1024 * SEL_REP;
    if (histBytes > ctx->sel_hist_bytes) {
        if (ctx->sel_hist) { CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); CANVAS_HIP_TRY(ctx, hipFree(ctx->sel_hist)); ctx->sel_hist = nullptr; ctx->sel_hist_bytes = 0; }
        CANVAS_HIP_TRY(ctx, hipMalloc(&ctx->sel_hist, histBytes)); ctx->sel_hist_bytes = histBytes;
        CANVAS_HIP_TRY(ctx, hipMemsetAsync(ctx->sel_hist, 0, ctx->sel_hist_bytes, ctx->stream));
    }
    WsCarver ws(ctx->ws);
    Soa caller{d_chr, d_start, d_stop, d_gc, d_count, nullptr};
    Soa S1; S1.chr = ws.take<int32_t>(n); S1.start = ws.take<int32_t>(n); S1.stop = ws.take<int32_t>(n); S1.gc = ws.take<int32_t>(n); S1.count = ws.take<float>(n); S1.dev = ws.take<double>(n);
    uint8_t* dFlags = ws.take<uint8_t>(n); uint32_t* dBlk = ws.take<uint32_t>(2 * (nb + 2)); uint32_t* keys32 = ws.take<uint32_t>(n); uint32_t* keysG = ws.take<uint32_t>(n);
    uint8_t* dIsAuto = ws.take<uint8_t>(nchr); double* dSd = ws.take<double>(nW0); double* dRunMad = ws.take<double>(CF_MAXRUN + 8); int64_t* dRunStart = ws.take<int64_t>(CF_MAXRUN + 8);
    long long* dPos = ws.take<long long>(65536); CleanDev* D = ws.take<CleanDev>(1); CfSel* P = ws.take<CfSel>(3); SelTile* dTiles = ws.take<SelTile>((size_t)tilesUpper * 3);
    ProfScope psTotal(ctx, "clean_total");
    rc = canvas_h2d_small(ctx, dIsAuto, h_chr_is_autosome, nchr); if (rc) return rc;
    CANVAS_HIP_TRY(ctx, hipMemsetAsync(D, 0, sizeof(CleanDev), ctx->stream));
    // ---- RemoveBigBins threshold (CanvasClean.cs:328-348): the 98th percentile of the bin sizes, left on the device
    const unsigned long long* dKey = nullptr;
    if (flags & CANVAS_CLEAN_FILTSIZE) {
        const int64_t index = (int64_t)(0.98 * (double)n);
        if (index < n) {
            hipLaunchKernelGGL(k_keys_size, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, d_start, d_stop, n, keys32);
            std::vector<unsigned long long> res;
            rc = radix_select<uint32_t>(ctx, keys32, 1, std::vector<int64_t>{0, n}, std::vector<SelQuery>{{0, 0, index}}, res, &dKey); if (rc) return rc;
        }
    }
    // ---- size filter + outlier filter: one compaction, caller -> S1
    // (a one-bin-per-thread variant of this kernel was measured: 68 us against 53 us for the staged one)
    hipLaunchKernelGGL(k_cf_flags_ab, dim3(nb), dim3(256), 0, ctx->stream, d_chr, d_start, d_stop, d_gc, d_count, n, nchr, dKey, (flags & CANVAS_CLEAN_OUTLIERS) ? 1 : 0, dFlags, dBlk, D);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, ctx->stream, dBlk, nb, &D->nAB);
    hipLaunchKernelGGL(k_cf_scatter_ab, dim3(nb), dim3(256), 0, ctx->stream, dFlags, dBlk, n, caller, S1, dIsAuto, nchr, D);
    // ---- local SD (CanvasClean.cs:243-298): window SDs and chromosome runs on the main stream, the per-run MAD on the side stream
    const bool wantLsd = (flags & CANVAS_CLEAN_LOCALSD) && n >= 50000;
    if (wantLsd) {
        const int64_t nWu = (n - 2) / 20;
        if (nWu > 0) hipLaunchKernelGGL(k_local_sd, dim3(nblk(nWu, 256)), dim3(256), 0, ctx->stream, S1.count, nWu, dSd, S1.dev, &D->nAB);
        hipLaunchKernelGGL(k_run_bounds, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, S1.chr, n, &D->nRunRec, dPos, 65536, &D->nAB);
        hipLaunchKernelGGL(k_cf_runs_build, dim3(1), dim3(CF_MAXRUN), 0, ctx->stream, dPos, &D->nRunRec, dRunStart, D, 1);
        rc = canvas_side_init(ctx); if (rc) return rc;
        if (!ctx->side_ev2) CANVAS_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->side_ev2, hipEventDisableTiming));
        CANVAS_HIP_TRY(ctx, hipEventRecord(ctx->side_ev, ctx->stream));
        CANVAS_HIP_TRY(ctx, hipStreamWaitEvent(ctx->side, ctx->side_ev, 0));
        hipLaunchKernelGGL(k_run_mad, dim3(CF_MAXRUN), dim3(1024), 0, ctx->side, dSd, dRunStart, dRunMad, &D->nruns);
    }
    // ---- GC strip decision, grouping, NormalizeByGC
    hipLaunchKernelGGL(k_cf_dec_gc, dim3(1), dim3(128), 0, ctx->stream, flags, min_bins_per_gc, D);
    if (flags & CANVAS_CLEAN_GCNORM) {
        hipLaunchKernelGGL(k_cf_group_keys, dim3(nb), dim3(256), 0, ctx->stream, S1.chr, S1.gc, S1.count, dIsAuto, n, D, keysG);
        hipLaunchKernelGGL(k_cf_sel_setup, dim3(1), dim3(128), 0, ctx->stream, 0, 0, D, P + 0, dTiles);
        cf_select_passes(ctx, keysG, dTiles, P + 0, n);
        hipLaunchKernelGGL(k_cf_dec_e, dim3(1), dim3(128), 0, ctx->stream, P + 0, D);
        hipLaunchKernelGGL(k_cf_apply_gc, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, S1.count, S1.gc, n, D, P + 0);
        if (wantLsd && n > 500000) {
            // NormalizeVarianceByGC (CanvasClean.cs:512-519): quartiles of the normalised counts; if it changes anything, NormalizeByGC once more
            hipLaunchKernelGGL(k_cf_sel_setup, dim3(1), dim3(128), 0, ctx->stream, 1, 1, D, P + 1, dTiles + tilesUpper);
            hipLaunchKernelGGL(k_cf_xform_gc, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, keysG, D, P + 1);
            cf_select_passes(ctx, keysG, dTiles + tilesUpper, P + 1, n);
            hipLaunchKernelGGL(k_cf_dec_f, dim3(1), dim3(128), 0, ctx->stream, P + 1, D);
            // ... which it rarely does: the last compaction below is enqueued on the assumption that it does not; when k_cf_dec_f says it did, that compaction does nothing
            // and clean_device_driven_finish enqueues the variance scaling, the second NormalizeByGC and the compaction (one more synchronisation, in that case only)
        }
    }
    // ---- local-SD average, last compaction into the caller's arrays
    if (wantLsd) {
        CANVAS_HIP_TRY(ctx, hipEventRecord(ctx->side_ev2, ctx->side));
        CANVAS_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->side_ev2, 0));
    }
    hipLaunchKernelGGL(k_cf_lsd_avg, dim3(1), dim3(64), 0, ctx->stream, dRunMad, D);
    hipLaunchKernelGGL(k_cf_flags_final, dim3(nb), dim3(256), 0, ctx->stream, S1.gc, S1.dev, n, D, dFlags, dBlk);
    hipLaunchKernelGGL(k_cf_scan_blocks, dim3(1), dim3(1024), 0, ctx->stream, dBlk, D, &D->nFinal);
    hipLaunchKernelGGL(k_cf_scatter_final, dim3(nb), dim3(256), 0, ctx->stream, dFlags, dBlk, n, S1, caller, D, 0);
    rc = canvas_pin_reserve(ctx, sizeof(CleanDev)); if (rc) return rc;
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(ctx->pin, D, sizeof(CleanDev), hipMemcpyDeviceToHost, ctx->stream));
    CleanPending pend{n, nb, tilesUpper, S1, caller, dFlags, dBlk, keysG, dTiles, P, D};
    ctx->clean_pending.assign((const char*)&pend, (const char*)&pend + sizeof pend);
    return CANVAS_OK;
}
// returns CANVAS_OK and sets *handled = false when the host-driven path has to take over (nothing was modified)
static int32_t clean_device_driven_finish(canvas_ctx* ctx, double* h_local_sd_out, int64_t* h_n_out, int32_t* h_info, bool* handled) {
    *handled = false;
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    CANVAS_HIP_TRY(ctx, hipGetLastError());
    if (((const CleanDev*)ctx->pin)->changed && !((const CleanDev*)ctx->pin)->fallback && !((const CleanDev*)ctx->pin)->bad && ctx->clean_pending.size() == sizeof(CleanPending)) {
        // NormalizeVarianceByGC changed the counts (CanvasClean.cs:512-519): scale them, NormalizeByGC again on the new counts, then the last compaction
        CleanPending q; memcpy(&q, ctx->clean_pending.data(), sizeof q);
        const int64_t n = q.n;
        hipLaunchKernelGGL(k_cf_apply_var, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, q.S1.count, q.S1.gc, n, q.D);
        hipLaunchKernelGGL(k_cf_xform_var, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, q.keysG, q.D);
        hipLaunchKernelGGL(k_cf_sel_setup, dim3(1), dim3(128), 0, ctx->stream, 0, 2, q.D, q.P + 2, q.dTiles + 2 * (size_t)q.tilesUpper);
        cf_select_passes(ctx, q.keysG, q.dTiles + 2 * (size_t)q.tilesUpper, q.P + 2, n);
        hipLaunchKernelGGL(k_cf_dec_e, dim3(1), dim3(128), 0, ctx->stream, q.P + 2, q.D);
        hipLaunchKernelGGL(k_cf_apply_gc, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, q.S1.count, q.S1.gc, n, q.D, q.P + 2);
        hipLaunchKernelGGL(k_cf_flags_final, dim3(q.nb), dim3(256), 0, ctx->stream, q.S1.gc, q.S1.dev, n, q.D, q.dFlags, q.dBlk);
        hipLaunchKernelGGL(k_cf_scan_blocks, dim3(1), dim3(1024), 0, ctx->stream, q.dBlk, q.D, &q.D->nFinal);
        hipLaunchKernelGGL(k_cf_scatter_final, dim3(q.nb), dim3(256), 0, ctx->stream, q.dFlags, q.dBlk, n, q.S1, q.caller, q.D, 1);
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(ctx->pin, q.D, sizeof(CleanDev), hipMemcpyDeviceToHost, ctx->stream));
        CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        CANVAS_HIP_TRY(ctx, hipGetLastError());
    }
    ctx->clean_pending.clear();
    const CleanDev& H = *(const CleanDev*)ctx->pin;
    if (H.bad) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_clean: a bin has gc outside 0..100 or a chromosome index outside [0, nchr) (the reference throws IndexOutOfRangeException)");
    if (H.fallback) return CANVAS_OK;                                  // *handled stays false: nothing was written to the caller's arrays
    *handled = true;
    *h_n_out = (int64_t)H.nFinal;
    if (h_local_sd_out) *h_local_sd_out = H.haveLocalSd ? H.localSd : -1.0;
    if (h_info) {
        int32_t info[8] = {0};
        info[0] = (int32_t)H.nA; info[1] = (int32_t)H.nAB; info[2] = (int32_t)(H.gcActive ? H.kept : (long long)H.nAB); info[3] = (int32_t)H.nFinal; info[4] = H.changed;
        memcpy(h_info, info, sizeof info);
    }
    return CANVAS_OK;
}
static int32_t clean_device_driven(canvas_ctx* ctx, int64_t n, int32_t* d_chr, int32_t* d_start, int32_t* d_stop, float* d_count, int32_t* d_gc, int32_t nchr, const uint8_t* h_chr_is_autosome,
                                   uint32_t flags, int32_t min_bins_per_gc, double* h_local_sd_out, int64_t* h_n_out, int32_t* h_info, bool* handled) {
    *handled = false;
    int32_t rc = clean_device_driven_enqueue(ctx, n, d_chr, d_start, d_stop, d_count, d_gc, nchr, h_chr_is_autosome, flags, min_bins_per_gc); if (rc) return rc;
    return clean_device_driven_finish(ctx, h_local_sd_out, h_n_out, h_info, handled);
}
