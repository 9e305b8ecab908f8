// CanvasClean -g alone (BASELINE configs[1]): RemoveBinsWithExtremeGC + NormalizeByGC (CanvasClean/CanvasClean.cs:163-196,207-237,497-505) on the whole-number counts of a
// .binned file, in THREE launches and one synchronisation.  (The general chain of clean_fast.hpp serves this configuration in seven launches — two compactions through a
// scratch copy, grouped keys, a counting sweep — at 0.11 ms for 2.5 M bins; a first three-launch attempt in round 4 counted per (GC, count) with device atomics and lost.)
//
//   k_cg_count    every workgroup holds its chunk of the bins (<= 16 384) in registers and counts the autosomal ones per (GC, count) in LDS — 101 rows x 256 counts, 16-bit
//                 halves, no atomic leaves the CU — and writes the table as a slab of its own (52 KB, coalesced); per-GC totals beside it.  The last workgroup to arrive
//                 sums the totals and takes the RemoveBinsWithExtremeGC decision (:207-237).
//   k_cg_medians  one workgroup per GC bucket sums its row over the slabs and reads the bucket's median off the counters; the rows of the kept buckets are added into one
//                 genome row, from which the last workgroup reads the global median.  One more workgroup turns the per-chunk totals into the chunks' output offsets meanwhile.
//   k_cg_apply    IN PLACE: every workgroup loads its whole chunk into registers, says so (a flag per chunk), waits for the chunks its output range reaches back into, and
//                 stores the surviving bins — normalised (:189-195) — at their final position.  Chunks are handed out by a ticket, so a workgroup only ever waits for
//                 workgroups that are already running.
// Nothing is written to the caller's arrays unless every decision could be taken exactly: counts that are not whole numbers, an order statistic outside the counter
// window, a bucket threshold below 100 (the neighbour-weighted medians) raise `fail`, the arrays stay as they were and the general chain takes the sample.
// Batch-native like clean_fast.hpp: blockIdx.y = sample; the argument blocks travel as a kernel argument.
#pragma once

#define CG_T 1024                      // threads of a workgroup
#define CG_W 256                       // counts a row of the counters covers: [lo, lo + CG_W)
#define CG_ROWW (CG_W / 2)             // ... as words of two 16-bit halves
#define CG_SLAB (NGC * CG_ROWW)        // words of a workgroup's table
#define CG_BPT 16                      // bins a thread holds at most
#define CG_CHUNK_MAX (CG_T * CG_BPT)
#define CG_MAXB 8                      // samples of a batch
#define CG_MAXG 2048                   // chunks of a sample
#define CG_TABROWS 5                   // per chunk and GC: autosomal bins, bins of the other chromosomes, autosomal counts below / above the window; row 4 word 0: flags

#define CG_FAIL_INDEX 1u               // a gc outside 0..100 or a chromosome index outside the table (an error, not a fallback)
#define CG_FAIL_VALUE 2u               // a count that is not a whole number in [0, 2^30)
#define CG_FAIL_THRESHOLD 4u           // buckets with fewer than 100 autosomal bins survive: their medians are neighbour-weighted quantiles (CanvasClean.cs:178-187)
#define CG_FAIL_WINDOW 8u              // a median lies outside the counter window

struct CgDev {
    double medians[NGC]; double globalMedian;
    unsigned long long nFinal;         // bins that survive (= n when nothing is stripped)
    uint32_t fail, active, threshold; int32_t lo;
    uint32_t cnt[NGC], all[NGC];       // autosomal bins / bins of any chromosome per GC
    uint8_t keep[NGC + 3];
    uint32_t off[CG_MAXG + 1];         // output offset of every chunk
};
struct CgState {                       // zero between calls (whoever uses a word last resets it)
    uint32_t tick[4];                  // arrival tickets of k_cg_count / k_cg_medians, chunk ticket of k_cg_apply
    unsigned long long ghBelow;        // autosomal counts of the kept buckets below the window
    uint32_t gh[CG_W];                 // ... inside it, per count
    uint32_t ready[CG_MAXG];           // k_cg_apply: epoch of the last call in which chunk w was read completely (never reset: epochs only grow)
};
struct CgArgs {
    int64_t n; int32_t nchr, G, chunk, bpt; uint32_t epoch, padA;
    int32_t *chr, *start, *stop, *gc; float* count;
    uint32_t* slab;                    // [G][CG_SLAB]
    uint32_t* tab;                     // [G][CG_TABROWS][NGC]
    CgDev* D; CgState* S;
};
struct CgPack { CgArgs a[CG_MAXB]; uint8_t isAuto[256]; int32_t minBinsPerGc, pad; };

__global__ void __launch_bounds__(CG_T) k_cg_count(const CgPack P) {
    const CgArgs& A = P.a[blockIdx.y];
    const int w = (int)blockIdx.x, t = (int)threadIdx.x;
    if (w >= A.G) return;
    __shared__ uint32_t sH[CG_SLAB];
    __shared__ uint32_t sOther[NGC], sBelow[NGC], sAbove[NGC], sBad;
    __shared__ uint8_t sAuto[256];
    __shared__ float sSamp[33];
    __shared__ int sLo, sLast;
    const gptr<const int32_t> chr = as_global((const int32_t*)A.chr), gc = as_global((const int32_t*)A.gc); const gptr<const float> cnt = as_global((const float*)A.count);
    const int64_t b0 = (int64_t)w * A.chunk;
    const int len = (int)min((int64_t)A.chunk, A.n - b0);
    // the chunk, all loads in flight together
    int c[CG_BPT], g[CG_BPT]; float x[CG_BPT];
#pragma unroll
    for (int j = 0; j < CG_BPT; j++) {
        c[j] = 0; g[j] = 0; x[j] = 0.0f;
        if (j < A.bpt) { const int i = j * CG_T + t; if (i < len) { c[j] = chr[b0 + i]; g[j] = gc[b0 + i]; x[j] = cnt[b0 + i]; } }
    }
    for (int i = t; i < CG_SLAB; i += CG_T) sH[i] = 0u;
    if (t < NGC) { sOther[t] = 0u; sBelow[t] = 0u; sAbove[t] = 0u; }
    if (t < 256) sAuto[t] = t < A.nchr ? P.isAuto[t] : (uint8_t)0;
    if (t == 0) sBad = 0u;
    // the level of the sample: the median of 33 strided counts (the same 33 in every workgroup); the window starts 128 below it
    if (t < 33) sSamp[t] = cnt[(int64_t)t * A.n / 33];
    __syncthreads();
    if (t < 64) {
        const float xs = t < 33 ? sSamp[t] : 0.0f;
        int rank = -1;
        if (t < 33) { rank = 0; for (int j = 0; j < 33; j++) { const float y = sSamp[j]; rank += (y < xs || (y == xs && j < t)) ? 1 : 0; } }
        const unsigned long long m = __ballot(rank == 16);
        if (t == 0) { int lo = 0; if (m) { const float lv = sSamp[__builtin_ctzll(m)]; if (lv >= 128.0f && lv < 1.0e9f) lo = (int)lv - 128; } sLo = lo; }
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
                    else atomicAdd(&sH[g[j] * CG_ROWW + (d >> 1)], 1u << (16 * (d & 1)));                    // (a chunk holds <= 16 384 bins: a half cannot overflow)
                }
            }
        }
    }
    if (bad) atomicOr(&sBad, bad);
    __syncthreads();
    // the table as this workgroup's slab; the row totals (+ the counts outside the window) = autosomal bins per GC
    uint32_t* slab = A.slab + (size_t)w * CG_SLAB;
    for (int i = t; i < CG_SLAB; i += CG_T) slab[i] = sH[i];
    uint32_t* tab = A.tab + (size_t)w * CG_TABROWS * NGC;
    if (t < NGC * 8) {
        const int gg = t >> 3, part = t & 7;
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < CG_ROWW / 8; k++) { const uint32_t wd = sH[gg * CG_ROWW + part * (CG_ROWW / 8) + k]; s += (wd & 0xFFFFu) + (wd >> 16); }
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        if (part == 0) {
            cf_st(&tab[0 * NGC + gg], s + sBelow[gg] + sAbove[gg]); cf_st(&tab[1 * NGC + gg], sOther[gg]); cf_st(&tab[2 * NGC + gg], sBelow[gg]); cf_st(&tab[3 * NGC + gg], sAbove[gg]);
        }
    }
    if (t == 0) cf_st(&tab[4 * NGC], sBad);
    if (!cf_arrive_last(&A.S->tick[0], (uint32_t)A.G, &sLast)) return;
    // ---- the last workgroup: totals per GC (a wave per chunk row: coalesced), the RemoveBinsWithExtremeGC decision
    if (t == 0) cf_st(&A.S->tick[0], 0u);
    uint32_t (*sRed)[2 * NGC + 2] = reinterpret_cast<uint32_t (*)[2 * NGC + 2]>(sH);       // (the table has left for its slab)
    uint32_t* sCnt = sH + 16 * (2 * NGC + 2);
    {
        const int wv = t >> 6, l = t & 63;
        uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, fl = 0;
        for (int s = wv; s < A.G; s += CG_T / 64) {
            const uint32_t* row = A.tab + (size_t)s * CG_TABROWS * NGC;
            a0 += row[l]; a1 += row[l + 64]; a2 += row[l + 128]; if (l + 192 < 2 * NGC) a3 += row[l + 192];
            if (l == 0) fl |= row[4 * NGC];
        }
        sRed[wv][l] = a0; sRed[wv][l + 64] = a1; sRed[wv][l + 128] = a2; if (l + 192 < 2 * NGC) sRed[wv][l + 192] = a3;
        if (l == 0) sRed[wv][2 * NGC] = fl;
    }
    __syncthreads();
    if (t < 2 * NGC + 1) { uint32_t s = 0; if (t < 2 * NGC) { for (int k = 0; k < CG_T / 64; k++) s += sRed[k][t]; } else { for (int k = 0; k < CG_T / 64; k++) s |= sRed[k][2 * NGC]; } sCnt[t] = s; }
    __syncthreads();
    if (t < 64) {
        // counts[] and totalCount of CanvasClean.cs:214-222 (integers: exact in any order); threshold = Math.Min(100, Math.Max(-w, (int)(totalCount / 101)))
        unsigned long long totalA = 0;
        for (int k = 0; k < NGC; k++) totalA += sCnt[k];
        const int averageCountPerGC = max(P.minBinsPerGc, (int)((double)totalA / NGC));
        const int threshold = min(100, averageCountPerGC);
        bool kp0 = (int)sCnt[t] >= threshold, kp1 = t + 64 < NGC ? (int)sCnt[t + 64] >= threshold : false;
        unsigned long long kept = (kp0 ? (unsigned long long)sCnt[t] + sCnt[NGC + t] : 0ull) + (kp1 ? (unsigned long long)sCnt[t + 64] + sCnt[NGC + t + 64] : 0ull);
        kept = wave_reduce_add_u64(kept);
        const bool active = kept > 0;                              // nothing survives: "proceed without GC correction" (CanvasClean.cs:500-505) — no strip, no normalisation
        if (!active) { kp0 = true; kp1 = true; }
        CgDev* D = A.D;
        D->keep[t] = kp0 ? 1 : 0; D->cnt[t] = sCnt[t]; D->all[t] = sCnt[t] + sCnt[NGC + t]; D->medians[t] = 0.0;
        if (t + 64 < NGC) { D->keep[t + 64] = kp1 ? 1 : 0; D->cnt[t + 64] = sCnt[t + 64]; D->all[t + 64] = sCnt[t + 64] + sCnt[NGC + t + 64]; D->medians[t + 64] = 0.0; }
        if (t == 0) {
            D->fail = sCnt[2 * NGC] | ((active && threshold < 100) ? CG_FAIL_THRESHOLD : 0u);
            D->active = active ? 1u : 0u; D->threshold = (uint32_t)threshold; D->lo = lo; D->globalMedian = 0.0; D->nFinal = active ? kept : (unsigned long long)A.n;
        }
    }
}

// value of 0-based rank r in the sorted counts described by `below` counts under the window and the inclusive prefix sums `inc` of the window's counters (thread v < CG_W
// holds counter v's): the thread whose counter covers r stores it
__device__ __forceinline__ void cg_pick(unsigned long long r, unsigned long long below, unsigned long long excl, unsigned long long inc, int v, int lo, int* out) {
    if (r >= below + excl && r < below + inc) *out = lo + v;
}

__global__ void __launch_bounds__(CG_T) k_cg_medians(const CgPack P) {
    const CgArgs& A = P.a[blockIdx.y];
    const int role = (int)blockIdx.x, t = (int)threadIdx.x;
    CgDev* D = A.D;
    __shared__ uint32_t sPart[8][CG_W];
    __shared__ unsigned long long sWave[16];
    __shared__ int sPick[2], sLast;
    __shared__ uint32_t sTot[CG_MAXG];
    const uint32_t failIn = D->fail;                               // (written by the previous launch)
    const bool live = !failIn && D->active;
    if (role == NGC) {
        // ---- the chunks' output offsets: bins of the kept buckets per chunk (a wave per chunk row), exclusive sums
        if (!live) return;
        const int wv = t >> 6, l = t & 63;
        const bool k0 = D->keep[l] != 0, k1 = D->keep[(l + 64) % NGC] != 0, k2 = D->keep[(l + 128) % NGC] != 0, k3 = l + 192 < 2 * NGC && D->keep[(l + 192) % NGC] != 0;
        for (int s = wv; s < A.G; s += CG_T / 64) {
            const uint32_t* row = A.tab + (size_t)s * CG_TABROWS * NGC;
            uint32_t v = (k0 ? row[l] : 0u) + (k1 ? row[l + 64] : 0u) + (k2 ? row[l + 128] : 0u) + (k3 ? row[l + 192] : 0u);
            v = wave_reduce_add_u32(v);
            if (l == 0) sTot[s] = v;
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
    const bool mine = live && D->keep[g] != 0;
    if (mine) {
        // the bucket's row over all slabs
        const int j = t & (CG_ROWW - 1), grp = t >> 7;
        uint32_t aLo = 0, aHi = 0;
        const gptr<const uint32_t> slab = as_global((const uint32_t*)A.slab);
        for (int s = grp; s < A.G; s += CG_T / CG_ROWW) { const uint32_t wd = slab[(size_t)s * CG_SLAB + g * CG_ROWW + j]; aLo += wd & 0xFFFFu; aHi += wd >> 16; }
        sPart[grp][2 * j] = aLo; sPart[grp][2 * j + 1] = aHi;
    }
    __syncthreads();
    uint32_t h = 0; unsigned long long below = 0;
    if (mine) {
        if (t < CG_W) { for (int k = 0; k < 8; k++) h += sPart[k][t]; }
        // counts under the window (row 2 of the chunk tables)
        uint32_t b = 0;
        for (int s = t; s < A.G; s += CG_T) b += A.tab[(size_t)s * CG_TABROWS * NGC + 2 * NGC + g];
        const unsigned long long bw = wave_reduce_add_u64((unsigned long long)b);
        if ((t & 63) == 0) sWave[t >> 6] = bw;
    }
    __syncthreads();
    unsigned long long inc = 0, excl = 0;
    if (mine) {
        for (int k = 0; k < 16; k++) below += sWave[k];
        inc = wave_inclusive_scan_u32(h);                          // (t < CG_W: four waves)
    }
    __syncthreads();
    if (mine && t < CG_W && (t & 63) == 63) sWave[t >> 6] = inc;
    if (t == 0) { sPick[0] = -1; sPick[1] = -1; }
    __syncthreads();
    if (mine) {
        const unsigned long long n = D->cnt[g];
        if (t < CG_W) {
            unsigned long long base = 0;
            for (int k = 0; k < (t >> 6); k++) base += sWave[k];
            excl = base + inc - h; inc = base + inc;
            // Utilities.Median of the bucket's counts: the middle one, or the mean of the two middle ones
            cg_pick((n - 1) / 2, below, excl, inc, t, D->lo, &sPick[0]);
            cg_pick(n / 2, below, excl, inc, t, D->lo, &sPick[1]);
            // the genome row
            if (h) __hip_atomic_fetch_add(&A.S->gh[t], h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (t == 0 && below) __hip_atomic_fetch_add(&A.S->ghBelow, below, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (mine && t == 0) {
        if (sPick[0] < 0 || sPick[1] < 0) __hip_atomic_fetch_or(&D->fail, CG_FAIL_WINDOW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else {
            const float m = (D->cnt[g] & 1u) ? (float)sPick[1] : ((float)sPick[0] + (float)sPick[1]) / 2.0f;      // SortedList<float>.Median: float arithmetic
            cf_st_f64(&D->medians[g], (double)m);
        }
    }
    if (!cf_arrive_last(&A.S->tick[1], (uint32_t)NGC, &sLast)) return;
    // ---- the last bucket workgroup: the median of all autosomal counts that survive (globalMedian, CanvasClean.cs:171), and the genome row back to zero
    if (t == 0) cf_st(&A.S->tick[1], 0u);
    const uint32_t hg = t < CG_W ? A.S->gh[t] : 0u;
    const unsigned long long gBelow = A.S->ghBelow;
    __syncthreads();
    if (t < CG_W) A.S->gh[t] = 0u;                                 // (whatever happens below: the next call finds zeros)
    if (t == 0) A.S->ghBelow = 0ull;
    if (!live) return;
    unsigned long long n = 0;
    for (int k = 0; k < NGC; k++) if (D->keep[k]) n += D->cnt[k];
    unsigned long long incg = wave_inclusive_scan_u32(hg);
    if (t < CG_W && (t & 63) == 63) sWave[t >> 6] = incg;
    if (t == 0) { sPick[0] = -1; sPick[1] = -1; }
    __syncthreads();
    if (t < CG_W) {
        unsigned long long base = 0;
        for (int k = 0; k < (t >> 6); k++) base += sWave[k];
        const unsigned long long ex = base + incg - hg; incg = base + incg;
        cg_pick((n - 1) / 2, gBelow, ex, incg, t, D->lo, &sPick[0]);
        cg_pick(n / 2, gBelow, ex, incg, t, D->lo, &sPick[1]);
    }
    __syncthreads();
    if (t == 0) {
        if (n == 0 || sPick[0] < 0 || sPick[1] < 0) __hip_atomic_fetch_or(&D->fail, CG_FAIL_WINDOW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else { const float m = (n & 1ull) ? (float)sPick[1] : ((float)sPick[0] + (float)sPick[1]) / 2.0f; D->globalMedian = (double)m; }
    }
}

__global__ void __launch_bounds__(CG_T) k_cg_apply(const CgPack P) {
    const CgArgs& A = P.a[blockIdx.y];
    const int t = (int)threadIdx.x;
    if ((int)blockIdx.x >= A.G) return;
    const CgDev* D = A.D;
    if (D->fail || !D->active) return;                             // nothing is touched: the general chain takes the sample / there is nothing to do
    __shared__ int sW;
    __shared__ double sMed[NGC]; __shared__ uint8_t sKeep[NGC + 3];
    __shared__ uint32_t sWc[CG_BPT * (CG_T / 64)], sBase[CG_BPT * (CG_T / 64)], sPart[4];
    // chunks in the order the workgroups start: a workgroup only ever waits for chunks that are already being worked on
    if (t == 0) {
        const uint32_t id = __hip_atomic_fetch_add(&A.S->tick[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)id == A.G - 1) __hip_atomic_store(&A.S->tick[2], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // every chunk is handed out: ready for the next call
        sW = (int)id;
    }
    if (t < NGC) { sMed[t] = D->medians[t]; sKeep[t] = D->keep[t]; }
    __syncthreads();
    const int w = sW;
    const double gm = D->globalMedian;
    const gptr<int32_t> chr = as_global(A.chr), start = as_global(A.start), stop = as_global(A.stop), gc = as_global(A.gc); const gptr<float> cnt = as_global(A.count);
    const int64_t b0 = (int64_t)w * A.chunk;
    const int len = (int)min((int64_t)A.chunk, A.n - b0);
    int c[CG_BPT], s[CG_BPT], e[CG_BPT], g[CG_BPT]; float x[CG_BPT];
#pragma unroll
    for (int j = 0; j < CG_BPT; j++) {
        c[j] = 0; s[j] = 0; e[j] = 0; g[j] = 0; x[j] = 0.0f;
        if (j < A.bpt) { const int i = j * CG_T + t; if (i < len) { c[j] = chr[b0 + i]; s[j] = start[b0 + i]; e[j] = stop[b0 + i]; g[j] = gc[b0 + i]; x[j] = cnt[b0 + i]; } }
    }
    // everything of this chunk is in registers: say so
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) __hip_atomic_store(&A.S->ready[w], A.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    // position of every surviving bin inside the chunk's output: bins in index order (slot j, thread t) -> j * CG_T + t
    const int wv = t >> 6, l = t & 63;
#pragma unroll
    for (int j = 0; j < CG_BPT; j++) {
        if (j < A.bpt) {
            const unsigned long long m = __ballot(j * CG_T + t < len && sKeep[g[j]] != 0);
            if (l == 0) sWc[j * (CG_T / 64) + wv] = (uint32_t)__popcll(m);
        }
    }
    __syncthreads();
    const int nslots = A.bpt * (CG_T / 64);                        // <= 256
    uint32_t v = t < nslots ? sWc[t] : 0u, inc = 0;
    if (t < 256) { inc = wave_inclusive_scan_u32(v); if (l == 63) sPart[wv] = inc; }
    __syncthreads();
    if (t < 256) { uint32_t base = 0; for (int k = 0; k < wv; k++) base += sPart[k]; if (t < nslots) sBase[t] = base + inc - v; }
    __syncthreads();
    // the output range [off, off + kept) reaches back into the chunks in front of this one by as many bins as were stripped there: they must have been read
    const uint32_t off = D->off[w];
    if (t == 0) {
        const int first = (int)(off / (uint32_t)A.chunk);
        for (int q = first; q < w; q++) while (__hip_atomic_load(&A.S->ready[q], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != A.epoch) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < CG_BPT; j++) {
        if (j >= A.bpt) continue;
        const bool kp = j * CG_T + t < len && sKeep[g[j]] != 0;
        const unsigned long long m = __ballot(kp);
        if (kp) {
            const int64_t o = (int64_t)off + sBase[j * (CG_T / 64) + wv] + (uint32_t)__popcll(m & ((1ull << l) - 1ull));
            const double median = sMed[g[j]];
            const float y = median > 0.0 ? (float)(gm * (double)x[j] / median) : x[j];          // CanvasClean.cs:189-195
            chr[o] = c[j]; start[o] = s[j]; stop[o] = e[j]; gc[o] = g[j]; cnt[o] = y;
        }
    }
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
        sz.take<uint32_t>((size_t)G * CG_SLAB); sz.take<uint32_t>((size_t)G * CG_TABROWS * NGC);
        Gmax = std::max(Gmax, G);
    }
    int32_t rc = canvas_ws_reserve(ctx, sz.off + 4096); if (rc) return rc;
    WsCarver ws(ctx->ws);
    CgDev* dD = ws.take<CgDev>(B);
    unsigned epoch = ++ctx->cg_epoch; if (epoch == 0) epoch = ++ctx->cg_epoch;
    for (int s = 0; s < B; s++) {
        CgArgs& a = pack.a[s];
        a.slab = ws.take<uint32_t>((size_t)a.G * CG_SLAB); a.tab = ws.take<uint32_t>((size_t)a.G * CG_TABROWS * NGC);
        a.D = dD + s; a.S = (CgState*)ctx->cg_state + s; a.epoch = epoch;
    }
    const size_t head = sizeof(CgDev);
    rc = canvas_pin_reserve(ctx, (size_t)B * head); if (rc) return rc;
    {
        ProfScope psTotal(ctx, "clean_total");
        hipLaunchKernelGGL(k_cg_count, dim3((unsigned)Gmax, (unsigned)B), dim3(CG_T), 0, ctx->stream, pack);
        hipLaunchKernelGGL(k_cg_medians, dim3(NGC + 1, (unsigned)B), dim3(CG_T), 0, ctx->stream, pack);
        hipLaunchKernelGGL(k_cg_apply, dim3((unsigned)Gmax, (unsigned)B), dim3(CG_T), 0, ctx->stream, pack);
    }
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(ctx->pin, dD, (size_t)B * head, hipMemcpyDeviceToHost, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    CANVAS_HIP_TRY(ctx, hipGetLastError());
    for (int s = 0; s < B; s++) {
        const CgDev& H = *(const CgDev*)((const char*)ctx->pin + (size_t)s * head);
        if (H.fail & CG_FAIL_INDEX) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_clean: a bin has gc outside 0..100 or a chromosome index outside [0, nchr) (the reference throws IndexOutOfRangeException)");
    }
    for (int s = 0; s < B; s++) {
        const CgDev& H = *(const CgDev*)((const char*)ctx->pin + (size_t)s * head);
        if (H.fail) continue;
        handled[s] = 1; h_n_out[s] = (int64_t)H.nFinal;
        if (h_info) { int32_t info[8] = {0}; info[0] = (int32_t)h_n[s]; info[1] = (int32_t)h_n[s]; info[2] = (int32_t)H.nFinal; info[3] = (int32_t)H.nFinal; info[6] = 1; memcpy(h_info + 8 * s, info, sizeof info); }
    }
    return CANVAS_OK;
}
