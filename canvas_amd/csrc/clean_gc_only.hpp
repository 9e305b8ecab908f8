// CanvasClean -g alone (BASELINE configs[1]): RemoveBinsWithExtremeGC + NormalizeByGC (CanvasClean/CanvasClean.cs:163-196,207-237,497-505) on the whole-number counts of a
// .binned file, in THREE launches and one synchronisation.  (The general chain of clean_fast.hpp serves this configuration in seven launches — two compactions through a
// scratch copy, grouped keys, a counting sweep — at 0.11 ms for 2.5 M bins; a first three-launch attempt in round 4 counted per (GC, count) with device atomics and lost.)
//
//   k_cg_count    every workgroup holds its chunk of the bins (<= 16 384) in registers and counts the autosomal ones per (GC, count) in LDS — 101 rows x 128 counts, 16-bit
//                 halves, no atomic leaves the CU — and writes the table as a slab of its own (26 KB, coalesced) with the per-GC totals of the chunk beside it.
//   k_cg_medians  every workgroup sums the chunks' per-GC totals (all of them the same 210 KB, out of the L2) and takes the RemoveBinsWithExtremeGC decision (:207-237) for
//                 itself — no workgroup waits for another; one workgroup per GC bucket then sums its row over the slabs, reads the bucket's median off the counters and adds
//                 the row into the genome row; one more workgroup turns the chunk totals into the chunks' output offsets.
//   k_cg_apply    IN PLACE: every workgroup loads its whole chunk into registers, says so (a flag per chunk), reads the global median off the genome row while the loads are
//                 in flight, waits for the chunks its output range reaches back into, and stores the surviving bins — normalised (:189-195) — at their final position.
//                 No workgroup can wait for ever, whatever else runs on the device: the wait for a chunk in front is BOUNDED (~0.3 ms), and a workgroup whose wait
//                 runs out puts its bins — they are in its registers — aside in a spill area instead of at their final place, says so (defer[chunk]) and LEAVES, which
//                 frees its CU for the workgroups it was waiting for; k_cg_fixup moves the spilled chunks to their place when the launch is over (every chunk has been
//                 read by then).  So the chunk can be the workgroup's index while the grid fits the device (round 5's fast path: on an empty device whoever is waited
//                 for is running) without the hang that path had when another process's kernels, a CU mask or a second spinning kernel kept part of the grid from
//                 being resident; a grid larger than the device takes its chunks by ticket (a workgroup then only waits for workgroups that have started).  The ticket
//                 for every grid cost 3.5 us of 47 (256 atomics on one word); the deferral costs nothing until it is needed.
// (A first version took the decisions in the LAST workgroup of k_cg_count / k_cg_medians: 14 + 9 us of tails behind an arrival ticket — measured with CANVAS_CG_CUT, the timing
// hook below.  Every decision is a function of a few hundred words; recomputing it where it is needed costs less than handing it over.)
// (Statistics and apply as ONE launch — the first 102 tickets the statistics roles, the others chunk workgroups that load their chunk and then wait for the roles' count — was
// built and measured: 42.6 us against 12 + 20 for the two launches.  The roles hold 51 CUs that the chunks then get a round later, and their chains of dependent round trips
// run slower beside 50 MB of streaming loads than alone.  A second form — the roles run BY the first 102 chunk workgroups, behind their own chunk loads, so that nobody holds a CU
// without a chunk — took 36.3 us: the roles' dependent round trips again, now behind the loads of their own workgroup as well.  The statistics want the device to themselves.)
// Nothing is written to the caller's arrays unless every decision could be taken exactly: counts that are not whole numbers, an order statistic outside the counter
// window, a bucket threshold below 100 (the neighbour-weighted medians) raise `fail`, the arrays stay as they were and the general chain takes the sample.
// Batch-native like clean_fast.hpp: blockIdx.y = sample; the argument blocks travel as a kernel argument.
#pragma once

#define CG_T 1024                      // threads of a workgroup
#define CG_NW (CG_T / 64)              // ... waves
#define CG_W 128                       // counts a row of the counters covers: [lo, lo + CG_W) (256: 13 MB of slabs per 2.5 M bins written and read again, +3 us)
#define CG_ROWW (CG_W / 2)             // ... as words of two 16-bit halves
#define CG_SLAB (NGC * CG_ROWW)        // words of a workgroup's table
#define CG_NG (CG_T / CG_ROWW)          // thread groups of k_cg_medians that share the slabs between them
#define CG_LB (256 / CG_NG)            // slabs a thread of k_cg_medians asks for at once (256 chunks in flight per workgroup)
#define CG_LROW (CG_ROWW + 1)           // ... row stride of its LDS form: with a power of two the bank of a counter would not depend on the GC value at all (lanes of equal count, different GC: conflicts)
#define CG_BPT 16                      // bins a thread holds at most
#define CG_CHUNK_MAX (CG_T * CG_BPT)
#define CG_MAXB 8                      // samples of a batch
#define CG_MAXG 2048                   // chunks of a sample
#define CG_TABW 208                    // words of a chunk's totals: [0..100] per GC: autosomal bins | bins of the other chromosomes << 16 (a chunk holds <= 16 384 bins),
                                       // [101..201] autosomal counts below the window per GC, [202] flags, [203] the window start
#define CG_TAB_BELOW NGC
#define CG_TAB_FLAGS (2 * NGC)
#define CG_TAB_LO (2 * NGC + 1)

#define CG_FAIL_INDEX 1u               // a gc outside 0..100 or a chromosome index outside the table (an error, not a fallback)
#define CG_FAIL_VALUE 2u               // a count that is not a whole number in [0, 2^30)
#define CG_FAIL_THRESHOLD 4u           // buckets with fewer than 100 autosomal bins survive: their medians are neighbour-weighted quantiles (CanvasClean.cs:178-187)
#define CG_FAIL_WINDOW 8u              // a median lies outside the counter window
#define CG_SPIN_LIMIT (1u << 13)       // s_sleep(1) rounds (64 clocks each, ~35 ns with the load) a workgroup waits for the chunks in front of it before it defers: ~0.3 ms

struct CgDev {
    double medians[NGC]; double globalMedian;
    unsigned long long nFinal;         // bins that survive (= n when nothing is stripped)
    unsigned long long nAuto;          // autosomal bins that survive (the list globalMedian is taken from)
    uint32_t fail, failWin, active, threshold; int32_t lo; uint32_t failApply;       // failWin: raised by bucket workgroups of k_cg_medians; failApply: by k_cg_apply (only the host reads it)
    uint32_t nDeferred, padD;          // chunks k_cg_apply put aside (k_cg_fixup moves them)
    uint8_t keep[NGC + 3];
    uint32_t off[CG_MAXG + 1];         // output offset of every chunk
};
struct CgState {
    uint32_t tick[4];                  // [2]: chunk ticket of k_cg_apply (back to zero when the last chunk is handed out)
    unsigned long long ghBelow;        // autosomal counts of the kept buckets below the window    } zeroed by k_cg_count, filled by k_cg_medians,
    uint32_t gh[CG_W];                 // ... inside it, per count                                  } read by k_cg_apply
    uint32_t ready[CG_MAXG];           // k_cg_apply: epoch of the last call in which chunk w was read completely (never reset: epochs only grow)
};
struct CgArgs {
    int64_t n; int32_t nchr, G, chunk, bpt; uint32_t epoch, padA;
    int32_t *chr, *start, *stop, *gc; float* count;
    uint32_t* slab;                    // [G][CG_SLAB]
    uint32_t* tab;                     // [G][CG_TABW]
    int32_t *spChr, *spStart, *spStop, *spGc; float* spCount;      // spill area of k_cg_apply (n entries per column; a deferred chunk's survivors from b0 on)
    uint32_t* defer;                   // [G] 0, or 1 + the number of bins the chunk put aside (zeroed by k_cg_count)
    CgDev* D; CgState* S;
};
struct CgPack { CgArgs a[CG_MAXB]; uint8_t isAuto[256]; int32_t minBinsPerGc, cut, ticket; uint32_t spin; };      // cut: timing hook (CANVAS_CG_CUT: the kernels stop early; results are void)

__global__ void __launch_bounds__(CG_T) k_cg_count(const CgPack P) {
    const CgArgs& A = P.a[blockIdx.y];
    const int w = (int)blockIdx.x, t = (int)threadIdx.x;
    if (w >= A.G) return;
    if (t == 0) A.defer[w] = 0u;
    __shared__ uint32_t sH[NGC * CG_LROW];
    __shared__ uint32_t sOther[NGC], sBelow[NGC], sAbove[NGC], sBad;
    __shared__ uint8_t sAuto[256];
    __shared__ float sSamp[33];
    __shared__ int sLo;
    const gptr<const int32_t> chr = as_global((const int32_t*)A.chr), gc = as_global((const int32_t*)A.gc); const gptr<const float> cnt = as_global((const float*)A.count);
    const int64_t b0 = (int64_t)w * A.chunk;
    const int len = (int)min((int64_t)A.chunk, A.n - b0);
    // the level of the sample: the median of 33 strided counts (the same 33 in every workgroup); the window starts CG_W / 2 below it
    float samp = 0.0f;
    if (t < 33) samp = cnt[(int64_t)t * A.n / 33];
    // the chunk, all loads in flight together
    int c[CG_BPT], g[CG_BPT]; float x[CG_BPT];
#pragma unroll
    for (int j = 0; j < CG_BPT; j++) {
        c[j] = 0; g[j] = 0; x[j] = 0.0f;
        if (j < A.bpt) { const int i = j * CG_T + t; if (i < len) { c[j] = chr[b0 + i]; g[j] = gc[b0 + i]; x[j] = cnt[b0 + i]; } }
    }
    for (int i = t; i < NGC * CG_LROW; i += CG_T) sH[i] = 0u;
    if (t < NGC) { sOther[t] = 0u; sBelow[t] = 0u; sAbove[t] = 0u; }
    if (t < 256) sAuto[t] = t < A.nchr ? P.isAuto[t] : (uint8_t)0;
    if (t == 0) sBad = 0u;
    if (w == 0) {                                                  // the genome row of this call starts from zero (k_cg_medians adds to it)
        if (t < CG_W) A.S->gh[t] = 0u;
        if (t == 0) { A.S->ghBelow = 0ull; A.D->failWin = 0u; }
    }
    if (t < 33) sSamp[t] = samp;
    if (P.cut == 1) { if (c[0] + g[0] + (int)x[0] + c[CG_BPT - 1] + g[5] + (int)x[7] == -12345) A.tab[0] = 1; return; }
    __syncthreads();
    if (t < 64) {
        const float xs = t < 33 ? sSamp[t] : 0.0f;
        int rank = -1;
        if (t < 33) { rank = 0; for (int j = 0; j < 33; j++) { const float y = sSamp[j]; rank += (y < xs || (y == xs && j < t)) ? 1 : 0; } }
        const unsigned long long m = __ballot(rank == 16);
        if (t == 0) { int lo = 0; if (m) { const float lv = sSamp[__builtin_ctzll(m)]; if (lv >= (float)(CG_W / 2) && lv < 1.0e9f) lo = (int)lv - CG_W / 2; } sLo = lo; }
    }
    __syncthreads();
    const int lo = sLo;
    uint32_t bad = 0u;
#pragma unroll
    for (int j = 0; j < CG_BPT; j++) {
        if (j < A.bpt && j * CG_T + t < len) {
            if ((unsigned)g[j] >= (unsigned)NGC || (unsigned)c[j] >= (unsigned)A.nchr) bad |= CG_FAIL_INDEX;
            else if (!sAuto[c[j]]) atomicAdd(&sOther[g[j]], 1u);
            else {
                const float xv = x[j];
                const int v = (xv >= 0.0f && xv < 1073741824.0f) ? (int)xv : -1;
                if (v < 0 || (float)v != xv || (__float_as_uint(xv) >> 31)) bad |= CG_FAIL_VALUE;            // (NaN fails the range test; -0.0 is not a count CanvasBin writes)
                else {
                    const int d = v - lo;
                    if (d < 0) atomicAdd(&sBelow[g[j]], 1u);
                    else if (d >= CG_W) atomicAdd(&sAbove[g[j]], 1u);
                    else atomicAdd(&sH[g[j] * CG_LROW + (d >> 1)], 1u << (16 * (d & 1)));                    // (a chunk holds <= 16 384 bins: a half cannot overflow)
                }
            }
        }
    }
    if (bad) atomicOr(&sBad, bad);
    __syncthreads();
    if (P.cut == 2) { if (sH[t] == 0xFFFFFFFFu) A.tab[0] = 1; return; }
    // the table as this workgroup's slab; the row totals (+ the counts outside the window) = autosomal bins per GC
    uint32_t* slab = A.slab + (size_t)w * CG_SLAB;
    for (int i = t; i < CG_SLAB; i += CG_T) slab[i] = sH[(i / CG_ROWW) * CG_LROW + (i & (CG_ROWW - 1))];
    if (P.cut == 3) return;
    uint32_t* tab = A.tab + (size_t)w * CG_TABW;
    if (t < NGC * 8) {
        const int gg = t >> 3, part = t & 7;
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < CG_ROWW / 8; k++) { const uint32_t wd = sH[gg * CG_LROW + part * (CG_ROWW / 8) + k]; s += (wd & 0xFFFFu) + (wd >> 16); }
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        if (part == 0) { tab[gg] = (s + sBelow[gg] + sAbove[gg]) | (sOther[gg] << 16); tab[CG_TAB_BELOW + gg] = sBelow[gg]; }
    }
    if (t == 0) { tab[CG_TAB_FLAGS] = sBad; tab[CG_TAB_LO] = (uint32_t)lo; }
}

// value of 0-based rank r in the sorted counts described by `below` counts under the window and the prefix sums of the window's counters (thread v < CG_W holds counter
// v's exclusive / inclusive sum): the thread whose counter covers r stores it
__device__ __forceinline__ void cg_pick(unsigned long long r, unsigned long long below, unsigned long long excl, unsigned long long inc, int v, int lo, int* out) {
    if (r >= below + excl && r < below + inc) *out = lo + v;
}
// prefix sums of one counter per thread t < CG_W (the others pass 0) over the workgroup; sWave[CG_NW] scratch.  Two barriers (the first one also lets go of sWave's last use).
__device__ __forceinline__ void cg_scan_row(uint32_t h, unsigned long long* sWave, unsigned long long& excl, unsigned long long& inc) {
    const int t = (int)threadIdx.x;
    const uint32_t wi = wave_inclusive_scan_u32(h);
    __syncthreads();
    if ((t & 63) == 63) sWave[t >> 6] = wi;
    __syncthreads();
    unsigned long long base = 0;
    for (int k = 0; k < (t >> 6) && k < CG_W / 64; k++) base += sWave[k];
    excl = base + wi - h; inc = base + wi;
}
// the RemoveBinsWithExtremeGC decision (CanvasClean.cs:207-237) from the chunks' totals: every thread of the workgroup gets the same answer.
// sRed[CG_NW][208], sCnt[208] scratch; afterwards sCnt[0..100] = autosomal bins per GC, [101..201] the other bins, sKeep[] = the buckets that stay.
struct CgDecision { uint32_t fail; bool active; int threshold; unsigned long long kept, keptAuto; };
__device__ __forceinline__ CgDecision cg_decide(const CgArgs& A, int minBinsPerGc, uint32_t (*sRed)[208], uint32_t* sCnt, uint8_t* sKeep, unsigned long long* sDec /* [3] */) {
    const int t = (int)threadIdx.x, wv = t >> 6, l = t & 63;
    const gptr<const uint32_t> tab = as_global((const uint32_t*)A.tab);
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, fl = 0;
    // a wave per chunk row (coalesced: lane l takes GC l and GC l + 64); sixteen rows = every load of a 256-chunk sample in flight at once (a dependent round trip to the
    // L2 / memory is ~2 us: the first version with four rows per round spent 8 us here).  A row index past the end is clamped for the load and masked afterwards: a load under
    // a condition is waited for where the condition ends, one round trip each.
    for (int s0 = wv; s0 < A.G; s0 += 16 * CG_NW) {
        uint32_t r[16][2];
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const gptr<const uint32_t> row = tab + (size_t)min(s0 + u * CG_NW, A.G - 1) * CG_TABW;
            r[u][0] = row[l]; r[u][1] = row[l + 64];               // (lanes 37..63 of the second load read the counts below the window and drop them)
        }
        asm volatile("" ::: "memory");      // (every load above is issued before the first one is waited for)
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const uint32_t m = s0 + u * CG_NW < A.G ? 0xFFFFFFFFu : 0u, x0 = r[u][0] & m, x1 = (l + 64 < NGC ? r[u][1] : 0u) & m;
            a0 += x0 & 0xFFFFu; a1 += x0 >> 16; a2 += x1 & 0xFFFFu; a3 += x1 >> 16;
        }
        // the chunks' flags: one more word per row (lane u takes row u of this round)
        { const int u = l & 15; const int s = s0 + u * CG_NW; if (l < 16 && s < A.G) fl |= tab[(size_t)s * CG_TABW + CG_TAB_FLAGS]; }
    }
    sRed[wv][l] = a0; sRed[wv][NGC + l] = a1; if (l + 64 < NGC) { sRed[wv][l + 64] = a2; sRed[wv][NGC + l + 64] = a3; }
    { uint32_t f2 = fl; for (int d = 1; d < 64; d <<= 1) f2 |= __shfl_xor(f2, d, 64); if (l == 63) sRed[wv][2 * NGC] = f2; }
    __syncthreads();
    if (t < 2 * NGC + 1) { uint32_t s = 0; if (t < 2 * NGC) { for (int k = 0; k < CG_NW; k++) s += sRed[k][t]; } else { for (int k = 0; k < CG_NW; k++) s |= sRed[k][2 * NGC]; } sCnt[t] = s; }
    __syncthreads();
    // counts[] and totalCount of CanvasClean.cs:214-222 (integers: exact in any order); threshold = Math.Min(100, Math.Max(-w, (int)(totalCount / 101))).  One wave decides
    // (lane l: GC l and GC l + 64) and leaves the answer in sCnt[204..207] / sKeep for everybody
    if (t < 64) {
        const uint32_t c0 = sCnt[t], c1 = t + 64 < NGC ? sCnt[t + 64] : 0u, o0 = sCnt[NGC + t], o1 = t + 64 < NGC ? sCnt[NGC + t + 64] : 0u;
        const unsigned long long totalA = wave_reduce_add_u64((unsigned long long)c0 + c1);
        const int averageCountPerGC = max(minBinsPerGc, (int)((double)totalA / NGC));
        const int threshold = min(100, averageCountPerGC);
        const bool k0 = (int)c0 >= threshold, k1 = t + 64 < NGC && (int)c1 >= threshold;
        const unsigned long long kept = wave_reduce_add_u64((k0 ? (unsigned long long)c0 + o0 : 0ull) + (k1 ? (unsigned long long)c1 + o1 : 0ull));
        const unsigned long long keptAuto = wave_reduce_add_u64((k0 ? (unsigned long long)c0 : 0ull) + (k1 ? (unsigned long long)c1 : 0ull));
        const bool active = kept > 0;                              // nothing survives: "proceed without GC correction" (CanvasClean.cs:500-505) — no strip, no normalisation
        sKeep[t] = (!active || k0) ? 1 : 0;
        if (t + 64 < NGC) sKeep[t + 64] = (!active || k1) ? 1 : 0;
        if (t == 0) { sDec[0] = kept; sDec[1] = keptAuto; sDec[2] = (unsigned long long)(unsigned)threshold; }
    }
    __syncthreads();
    CgDecision d;
    d.kept = sDec[0]; d.keptAuto = sDec[1]; d.threshold = (int)sDec[2];
    d.active = d.kept > 0;
    d.fail = sCnt[2 * NGC] | ((d.active && d.threshold < 100) ? CG_FAIL_THRESHOLD : 0u);
    return d;
}

__global__ void __launch_bounds__(CG_T) k_cg_medians(const CgPack P) {
    const CgArgs& A = P.a[blockIdx.y];
    const int role = (int)blockIdx.x, t = (int)threadIdx.x;
    CgDev* D = A.D;
    __shared__ uint32_t sRed[CG_NW][208];
    __shared__ uint32_t sCnt[208];
    __shared__ uint8_t sKeep[NGC + 3];
    __shared__ uint32_t sPart[CG_NG][CG_W];
    __shared__ unsigned long long sWave[CG_NW];
    __shared__ int sPick[2];
    __shared__ uint32_t sTot[CG_MAXG];
    // a bucket workgroup asks for its row of every slab (and the counts under the window) BEFORE it knows whether the bucket stays: the answer to that is one round trip
    // away as well, and the two travel together
    uint32_t aLo = 0, aHi = 0, bCnt = 0;
    uint32_t wd[CG_LB]; uint32_t b0w = 0;
    const int rowJ = t & (CG_ROWW - 1), rowGrp = t / CG_ROWW;
    const gptr<const uint32_t> slabRow = as_global((const uint32_t*)A.slab) + (size_t)(role < NGC ? role : 0) * CG_ROWW + rowJ;
    const gptr<const uint32_t> tabG = as_global((const uint32_t*)A.tab);
    {   // the first 256 chunks: these loads stay in flight across the decision below (the wait for ITS loads is the wait for these: loads return in order)
#pragma unroll
        for (int u = 0; u < CG_LB; u++) wd[u] = slabRow[(size_t)min(rowGrp + u * CG_NG, A.G - 1) * CG_SLAB];
        b0w = tabG[(size_t)min(t, A.G - 1) * CG_TABW + CG_TAB_BELOW + (role < NGC ? role : 0)];
    }
    __shared__ unsigned long long sDec[3];
    const CgDecision dec = cg_decide(A, P.minBinsPerGc, sRed, sCnt, sKeep, sDec);
    if (P.cut == 4) { if (dec.kept == 12345) D->fail = 1; return; }
    const bool live = !dec.fail && dec.active;
    const int lo = (int)A.tab[CG_TAB_LO];                          // (the same in every chunk's totals)
    if (role == NGC) {
        // ---- the sample's decisions for the host and k_cg_apply, and the chunks' output offsets: bins of the kept buckets per chunk (a wave per chunk row), exclusive sums
        if (t < NGC) D->keep[t] = sKeep[t];
        if (t == 0) {
            D->fail = dec.fail; D->active = dec.active ? 1u : 0u; D->threshold = (uint32_t)dec.threshold; D->lo = lo;
            D->nFinal = dec.active ? dec.kept : (unsigned long long)A.n; D->nAuto = dec.keptAuto; D->globalMedian = 0.0; D->failApply = 0u; D->nDeferred = 0u;
        }
        if (!live) return;
        const int wv = t >> 6, l = t & 63;
        const bool k0 = sKeep[l] != 0, k1 = l + 64 < NGC && sKeep[(l + 64) % NGC] != 0;
        const gptr<const uint32_t> tab = as_global((const uint32_t*)A.tab);
        for (int s0 = wv; s0 < A.G; s0 += 16 * CG_NW) {
            uint32_t v[16], r[16][2];
#pragma unroll
            for (int u = 0; u < 16; u++) { const gptr<const uint32_t> row = tab + (size_t)min(s0 + u * CG_NW, A.G - 1) * CG_TABW; r[u][0] = row[l]; r[u][1] = row[l + 64]; }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = (k0 ? (r[u][0] & 0xFFFFu) + (r[u][0] >> 16) : 0u) + (k1 ? (r[u][1] & 0xFFFFu) + (r[u][1] >> 16) : 0u);
#pragma unroll
            for (int u = 0; u < 16; u++) { const int s = s0 + u * CG_NW; const uint32_t tot = wave_reduce_add_u32(v[u]); if (l == 0 && s < A.G) sTot[s] = tot; }
        }
        __syncthreads();
        const uint32_t v0 = 2 * t < A.G ? sTot[2 * t] : 0u, v1 = 2 * t + 1 < A.G ? sTot[2 * t + 1] : 0u;
        const uint32_t inc = wave_inclusive_scan_u32(v0 + v1);
        if ((t & 63) == 63) sWave[t >> 6] = inc;
        __syncthreads();
        uint32_t base = 0;
        for (int k = 0; k < (t >> 6); k++) base += (uint32_t)sWave[k];
        const uint32_t ex = base + inc - (v0 + v1);
        if (2 * t < A.G) D->off[2 * t] = ex;
        if (2 * t + 1 < A.G) D->off[2 * t + 1] = ex + v0;
        if (2 * t + 1 == A.G) D->off[A.G] = ex + v0; else if (2 * t + 2 == A.G) D->off[A.G] = ex + v0 + v1;
        return;
    }
    const int g = role;
    if (!live || !sKeep[g]) { if (t == 0) D->medians[g] = 0.0; return; }
    // (the compiler must not pull the sums below in front of the decision's loads to shorten the life of wd[]: the loaded values are pinned to this point)
#pragma unroll
    for (int u = 0; u < CG_LB; u += 8) asm volatile("" : "+v"(wd[u]), "+v"(wd[u + 1]), "+v"(wd[u + 2]), "+v"(wd[u + 3]), "+v"(wd[u + 4]), "+v"(wd[u + 5]), "+v"(wd[u + 6]), "+v"(wd[u + 7]));
    asm volatile("" : "+v"(b0w));
#pragma unroll
    for (int u = 0; u < CG_LB; u++) { const uint32_t x = rowGrp + u * CG_NG < A.G ? wd[u] : 0u; aLo += x & 0xFFFFu; aHi += x >> 16; }
    if (t < A.G) bCnt = b0w;
    for (int s0 = rowGrp + CG_LB * CG_NG; s0 < A.G; s0 += CG_LB * CG_NG) {       // (samples of more than 256 chunks)
#pragma unroll
        for (int u = 0; u < CG_LB; u++) wd[u] = slabRow[(size_t)min(s0 + u * CG_NG, A.G - 1) * CG_SLAB];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int u = 0; u < CG_LB; u++) { const uint32_t x = s0 + u * CG_NG < A.G ? wd[u] : 0u; aLo += x & 0xFFFFu; aHi += x >> 16; }
    }
    for (int s = t + CG_T; s < A.G; s += CG_T) bCnt += tabG[(size_t)s * CG_TABW + CG_TAB_BELOW + g];
    sPart[rowGrp][2 * rowJ] = aLo; sPart[rowGrp][2 * rowJ + 1] = aHi;
    const uint32_t b = bCnt;
    const unsigned long long bw = wave_reduce_add_u64((unsigned long long)b);
    __syncthreads();
    if (P.cut == 5) { if (sPart[0][t & 255] == 0xFFFFFFFFu) D->fail = 1; return; }
    uint32_t h = 0;
    if (t < CG_W) { for (int k = 0; k < CG_NG; k++) h += sPart[k][t]; }
    if ((t & 63) == 0) sWave[t >> 6] = bw;
    if (t == 0) { sPick[0] = -1; sPick[1] = -1; }
    __syncthreads();
    unsigned long long below = 0;
    for (int k = 0; k < CG_NW; k++) below += sWave[k];
    unsigned long long excl, inc;
    cg_scan_row(h, sWave, excl, inc);
    const unsigned long long n = sCnt[g];
    if (t < CG_W) {
        // Utilities.Median of the bucket's counts: the middle one, or the mean of the two middle ones
        cg_pick((n - 1) / 2, below, excl, inc, t, lo, &sPick[0]);
        cg_pick(n / 2, below, excl, inc, t, lo, &sPick[1]);
        // the genome row
        if (h) __hip_atomic_fetch_add(&A.S->gh[t], h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (t == 0 && below) __hip_atomic_fetch_add(&A.S->ghBelow, below, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (t == 0) {
        if (sPick[0] < 0 || sPick[1] < 0) { D->medians[g] = 0.0; __hip_atomic_fetch_or(&D->failWin, CG_FAIL_WINDOW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        else {
            const float m = (n & 1ull) ? (float)sPick[1] : ((float)sPick[0] + (float)sPick[1]) / 2.0f;      // SortedList<float>.Median: float arithmetic
            D->medians[g] = (double)m;
        }
    }
}

__global__ void __launch_bounds__(CG_T) k_cg_apply(const CgPack P) {
    const CgArgs& A = P.a[blockIdx.y];
    const int t = (int)threadIdx.x;
    if ((int)blockIdx.x >= A.G) return;
    CgDev* D = A.D;
    if (D->fail || D->failWin || !D->active) return;               // nothing is touched: the general chain takes the sample / there is nothing to do
    __shared__ int sW;
    __shared__ double sMed[NGC]; __shared__ uint8_t sKeep[NGC + 3];
    __shared__ uint32_t sWc[CG_BPT * CG_NW], sBase[CG_BPT * CG_NW], sPart[4];
    __shared__ unsigned long long sWave[CG_NW];
    __shared__ int sPick[2]; __shared__ int sStall;
    // the chunk: the workgroup's index, or — a grid with more workgroups than the device has places for them — a ticket, i.e. the order in which the workgroups start (see the header)
    if (t == 0) {
        uint32_t id = blockIdx.x;
        if (P.ticket) {
            id = __hip_atomic_fetch_add(&A.S->tick[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((int)id == A.G - 1) __hip_atomic_store(&A.S->tick[2], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // every chunk is handed out: ready for the next call
        }
        sW = (int)id;
        sPick[0] = -1; sPick[1] = -1;
    }
    if (t < NGC) { sMed[t] = D->medians[t]; sKeep[t] = D->keep[t]; }
    const uint32_t hg = t < CG_W ? A.S->gh[t] : 0u;
    const unsigned long long gBelow = A.S->ghBelow, nAuto = D->nAuto;
    const int lo = D->lo;
    const gptr<int32_t> chr = as_global(A.chr), start = as_global(A.start), stop = as_global(A.stop), gc = as_global(A.gc); const gptr<float> cnt = as_global(A.count);
    // (asking for chunk blockIdx.x before the ticket is back and loading again on a mismatch was tried: workgroups do NOT take their tickets in index order — they start
    // round-robin over eight XCDs — and the second load cost more than the ticket's round trip: 23 -> 30 us)
    __syncthreads();
    const int w = sW;
    const int64_t b0 = (int64_t)w * A.chunk;
    const int len = (int)min((int64_t)A.chunk, A.n - b0);
    int cg[CG_BPT], s[CG_BPT], e[CG_BPT]; float x[CG_BPT];       // cg: chromosome | GC << 16 (both were range-checked by k_cg_count; nchr <= 256)
#pragma unroll
    for (int j = 0; j < CG_BPT; j++) {
        cg[j] = 0; s[j] = 0; e[j] = 0; x[j] = 0.0f;
        if (j < A.bpt) { const int i = j * CG_T + t; if (i < len) { cg[j] = chr[b0 + i] | (gc[b0 + i] << 16); s[j] = start[b0 + i]; e[j] = stop[b0 + i]; x[j] = cnt[b0 + i]; } }
    }
    if (P.cut == 7) { if (cg[0] + s[1] + e[2] + cg[3] + (int)x[4] + cg[9] + s[9] + e[9] + (int)x[9] == -12345) A.S->tick[3] = 1; return; }
    // while the chunk is on its way: the median of all autosomal counts that survive (globalMedian, CanvasClean.cs:171) off the genome row — every workgroup for itself
    {
        unsigned long long ex, in;
        cg_scan_row(hg, sWave, ex, in);
        if (t < CG_W) { cg_pick((nAuto - 1) / 2, gBelow, ex, in, t, lo, &sPick[0]); cg_pick(nAuto / 2, gBelow, ex, in, t, lo, &sPick[1]); }
    }
    // everything of this chunk is in registers: say so (the flag orders nothing but this workgroup's loads, which have completed)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) __hip_atomic_store(&A.S->ready[w], A.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (nAuto == 0 || sPick[0] < 0 || sPick[1] < 0) {              // (the same in every workgroup: nobody stores anything)
        if (w == 0 && t == 0) D->failApply = CG_FAIL_WINDOW;
        return;
    }
    const double gm = (double)((nAuto & 1ull) ? (float)sPick[1] : ((float)sPick[0] + (float)sPick[1]) / 2.0f);
    if (w == 0 && t == 0) D->globalMedian = gm;
    // position of every surviving bin inside the chunk's output: bins in index order (slot j, thread t) -> j * CG_T + t
    const int wv = t >> 6, l = t & 63;
#pragma unroll
    for (int j = 0; j < CG_BPT; j++) {
        if (j < A.bpt) {
            const unsigned long long m = __ballot(j * CG_T + t < len && sKeep[cg[j] >> 16] != 0);
            if (l == 0) sWc[j * CG_NW + wv] = (uint32_t)__popcll(m);
        }
    }
    __syncthreads();
    const int nslots = A.bpt * CG_NW;                              // <= 256
    uint32_t v = t < nslots ? sWc[t] : 0u, inc = 0;
    if (t < 256) { inc = wave_inclusive_scan_u32(v); if (l == 63) sPart[wv] = inc; }
    __syncthreads();
    if (t < 256) { uint32_t base = 0; for (int k = 0; k < wv; k++) base += sPart[k]; if (t < nslots) sBase[t] = base + inc - v; }
    __syncthreads();
    if (P.cut == 8) { if (sBase[t & 255] == 0xFFFFFFFFu) A.S->tick[3] = 1; return; }
    // the output range [off, off + kept) reaches back into the chunks in front of this one by as many bins as were stripped there: they must have been read
    const uint32_t off = D->off[w];
    if (t == 0) {
        const int first = (int)(off / (uint32_t)A.chunk);
        bool stalled = false;
        for (int q = first; q < w && !stalled; q++) {
            uint32_t spins = 0;
            while (__hip_atomic_load(&A.S->ready[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != A.epoch) {
                if (spins++ >= P.spin) { stalled = true; break; }      // (one budget for all the chunks in front)
                __builtin_amdgcn_s_sleep(1);
            }
        }
        sStall = stalled ? 1 : 0;
    }
    __syncthreads();
    // never write behind a chunk that may not have been read: a workgroup whose wait ran out puts its survivors aside (from b0 on in the spill columns, in output order) and
    // leaves — its CU is free for whoever it was waiting for — and k_cg_fixup moves them when the launch is over
    const bool deferred = sStall != 0;
    const gptr<int32_t> oChr = deferred ? as_global(A.spChr) : chr, oStart = deferred ? as_global(A.spStart) : start, oStop = deferred ? as_global(A.spStop) : stop, oGc = deferred ? as_global(A.spGc) : gc;
    const gptr<float> oCnt = deferred ? as_global(A.spCount) : cnt;
    const int64_t oBase = deferred ? b0 : (int64_t)off;
    if (deferred && t == 0) {
        uint32_t kept = 0; for (int k = 0; k < nslots; k++) kept += sWc[k];
        A.defer[w] = 1u + kept;
        __hip_atomic_fetch_add(&D->nDeferred, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int j = 0; j < CG_BPT; j++) {
        if (j >= A.bpt) continue;
        const int gj = cg[j] >> 16;
        const bool kp = j * CG_T + t < len && sKeep[gj] != 0;
        const unsigned long long m = __ballot(kp);
        if (kp) {
            const int64_t o = oBase + sBase[j * CG_NW + wv] + (uint32_t)__popcll(m & ((1ull << l) - 1ull));
            const double median = sMed[gj];
            const float y = median > 0.0 ? (float)(gm * (double)x[j] / median) : x[j];          // CanvasClean.cs:189-195
            oChr[o] = cg[j] & 0xFFFF; oStart[o] = s[j]; oStop[o] = e[j]; oGc[o] = gj; oCnt[o] = y;
        }
    }
}

// the chunks k_cg_apply put aside, moved to their place (launched only when there are any; every chunk of the sample has been read by now)
__global__ void __launch_bounds__(CG_T) k_cg_fixup(const CgPack P) {
    const CgArgs& A = P.a[blockIdx.y];
    const int w = (int)blockIdx.x, t = (int)threadIdx.x;
    if (w >= A.G) return;
    const uint32_t d = A.defer[w];
    if (d == 0) return;
    const int kept = (int)(d - 1u);
    const int64_t b0 = (int64_t)w * A.chunk, o0 = (int64_t)A.D->off[w];
    const gptr<const int32_t> sc = as_global((const int32_t*)A.spChr), ss = as_global((const int32_t*)A.spStart), se = as_global((const int32_t*)A.spStop), sg = as_global((const int32_t*)A.spGc);
    const gptr<const float> sv = as_global((const float*)A.spCount);
    const gptr<int32_t> chr = as_global(A.chr), start = as_global(A.start), stop = as_global(A.stop), gc = as_global(A.gc); const gptr<float> cnt = as_global(A.count);
    for (int i = t; i < kept; i += CG_T) { chr[o0 + i] = sc[b0 + i]; start[o0 + i] = ss[b0 + i]; stop[o0 + i] = se[b0 + i]; gc[o0 + i] = sg[b0 + i]; cnt[o0 + i] = sv[b0 + i]; }
}

// ---------------------------------------------------------------- host side
// B samples with flags == CANVAS_CLEAN_GCNORM.  handled[s] = 1: done in place, h_n_out / h_info filled; 0: the arrays are untouched and the general chain has to take the sample.
static int32_t clean_gc_only(canvas_ctx* ctx, int B, const int64_t* h_n, int32_t* const* d_chr, int32_t* const* d_start, int32_t* const* d_stop, float* const* d_count, int32_t* const* d_gc,
                             int32_t nchr, const uint8_t* h_chr_is_autosome, int32_t min_bins_per_gc, int64_t* h_n_out, int32_t* h_info, char* handled) {
    for (int s = 0; s < B; s++) handled[s] = 0;
    if (B > CG_MAXB || nchr > 256) return CANVAS_OK;
    for (int s = 0; s < B; s++) if (h_n[s] <= 0 || (h_n[s] + CG_CHUNK_MAX - 1) / CG_CHUNK_MAX > CG_MAXG) return CANVAS_OK;
    if (!ctx->cg_state) {
        CANVAS_HIP_TRY(ctx, hipMalloc(&ctx->cg_state, sizeof(CgState) * CG_MAXB));
        CANVAS_HIP_TRY(ctx, hipMemsetAsync(ctx->cg_state, 0, sizeof(CgState) * CG_MAXB, ctx->stream));
    }
    CgPack pack; memset(&pack, 0, sizeof pack);
    memcpy(pack.isAuto, h_chr_is_autosome, (size_t)nchr); pack.minBinsPerGc = min_bins_per_gc;
    { const char* cut = cvx_hook("CANVAS_CG_CUT"); pack.cut = cut ? atoi(cut) : 0; }
    WsSizer sz; sz.take<CgDev>(B);
    int Gmax = 1;
    for (int s = 0; s < B; s++) {
        const int64_t n = h_n[s];
        int64_t chunk = std::min<int64_t>(CG_CHUNK_MAX, std::max<int64_t>(CG_T, (n + 255) / 256));
        chunk = (chunk + 63) & ~63ll;
        const int G = (int)((n + chunk - 1) / chunk);
        CgArgs& a = pack.a[s];
        a.n = n; a.nchr = nchr; a.G = G; a.chunk = (int32_t)chunk; a.bpt = (int32_t)((chunk + CG_T - 1) / CG_T);
        a.chr = d_chr[s]; a.start = d_start[s]; a.stop = d_stop[s]; a.gc = d_gc[s]; a.count = d_count[s];
        sz.take<uint32_t>((size_t)G * CG_SLAB); sz.take<uint32_t>((size_t)G * CG_TABW);
        for (int k = 0; k < 5; k++) sz.take<uint32_t>((size_t)n); sz.take<uint32_t>((size_t)G);      // the spill columns and the deferral flags (touched only when a workgroup defers)
        Gmax = std::max(Gmax, G);
    }
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess) cus = 0;
    pack.ticket = ((long long)Gmax * B > (long long)cus || cvx_hook("CANVAS_CG_TICKET")) ? 1 : 0;       // (the hook forces the ticket for the tests)
    pack.spin = cvx_hook("CANVAS_CG_SPIN_LIMIT") ? (uint32_t)atoi(cvx_hook("CANVAS_CG_SPIN_LIMIT")) : CG_SPIN_LIMIT;      // (test hook: 0 = every workgroup that finds a chunk in front unread defers at once)
    int32_t rc = canvas_ws_reserve(ctx, sz.off + 4096); if (rc) return rc;
    WsCarver ws(ctx->ws);
    CgDev* dD = ws.take<CgDev>(B);
    unsigned epoch = ++ctx->cg_epoch; if (epoch == 0) epoch = ++ctx->cg_epoch;
    for (int s = 0; s < B; s++) {
        CgArgs& a = pack.a[s];
        a.slab = ws.take<uint32_t>((size_t)a.G * CG_SLAB); a.tab = ws.take<uint32_t>((size_t)a.G * CG_TABW);
        a.spChr = (int32_t*)ws.take<uint32_t>((size_t)a.n); a.spStart = (int32_t*)ws.take<uint32_t>((size_t)a.n); a.spStop = (int32_t*)ws.take<uint32_t>((size_t)a.n); a.spGc = (int32_t*)ws.take<uint32_t>((size_t)a.n);
        a.spCount = (float*)ws.take<uint32_t>((size_t)a.n); a.defer = ws.take<uint32_t>((size_t)a.G);
        a.D = dD + s; a.S = (CgState*)ctx->cg_state + s; a.epoch = epoch;
    }
    const size_t head = sizeof(CgDev);
    rc = canvas_pin_reserve(ctx, (size_t)B * head); if (rc) return rc;
    {
        ProfScope psTotal(ctx, "clean_total");
        hipLaunchKernelGGL(k_cg_count, dim3((unsigned)Gmax, (unsigned)B), dim3(CG_T), 0, ctx->stream, pack);
        if (pack.cut == 0 || pack.cut >= 4) hipLaunchKernelGGL(k_cg_medians, dim3(NGC + 1, (unsigned)B), dim3(CG_T), 0, ctx->stream, pack);
        if (pack.cut == 0 || pack.cut >= 7) hipLaunchKernelGGL(k_cg_apply, dim3((unsigned)Gmax, (unsigned)B), dim3(CG_T), 0, ctx->stream, pack);
    }
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(ctx->pin, dD, (size_t)B * head, hipMemcpyDeviceToHost, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    CANVAS_HIP_TRY(ctx, hipGetLastError());
    for (int s = 0; s < B; s++) {
        const CgDev& H = *(const CgDev*)((const char*)ctx->pin + (size_t)s * head);
        if (H.fail & CG_FAIL_INDEX) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_clean: a bin has gc outside 0..100 or a chromosome index outside [0, nchr) (the reference throws IndexOutOfRangeException)");
    }
    {   // chunks that were put aside: moved now
        bool any = false; long long nd = 0;
        for (int s = 0; s < B; s++) { const CgDev& H = *(const CgDev*)((const char*)ctx->pin + (size_t)s * head); if (!(H.fail || H.failWin || H.failApply) && H.nDeferred) { any = true; nd += H.nDeferred; } }
        if (any) {
            hipLaunchKernelGGL(k_cg_fixup, dim3((unsigned)Gmax, (unsigned)B), dim3(CG_T), 0, ctx->stream, pack);
            CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            CANVAS_HIP_TRY(ctx, hipGetLastError());
        }
        ctx->cg_deferred += nd;
    }
    for (int s = 0; s < B; s++) {
        const CgDev& H = *(const CgDev*)((const char*)ctx->pin + (size_t)s * head);
        if (H.fail || H.failWin || H.failApply) continue;
        handled[s] = 1; h_n_out[s] = (int64_t)H.nFinal;
        if (h_info) { int32_t info[8] = {0}; info[0] = (int32_t)h_n[s]; info[1] = (int32_t)h_n[s]; info[2] = (int32_t)H.nFinal; info[3] = (int32_t)H.nFinal; info[6] = 1; info[7] = (int32_t)H.nDeferred; memcpy(h_info + 8 * s, info, sizeof info); }      // ([7]: chunks whose wait ran out and that k_cg_fixup moved)
    }
    return CANVAS_OK;
}
