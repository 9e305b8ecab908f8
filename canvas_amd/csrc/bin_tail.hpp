// The single-read path of CanvasBin after the sweep (included by bin.hip), without a host round trip and without one-workgroup-per-chromosome scans:
//   k_tscan_reduce   per chunk of 8192 tiles the totals of the four per-tile quantities (possible, observed, masked hits, G/C) + per chromosome the possible positions
//                    in front of pos0; its last workgroup turns the chunk totals into exclusive prefixes
//   k_tscan_apply    ONE scan over all tiles of the genome (uint32 wrap-around: only differences inside a chromosome are ever used, and those are < 2^31); its last
//                    workgroup takes the decisions the reference takes once per sample: per-chromosome totals and rates (CanvasBin.cs:30-71), the median rate and the bin
//                    size (CanvasBin.cs:73-83: all IEEE double operations, identical on the device), the bins per chromosome and their offsets
//   k_bin_close2     k_bin_close reading the bin size from device memory (the host has not seen it yet)
//   k_bin_resolve_fin  boundary resolution + prefix differences in one kernel: a workgroup resolves the bin in front of its own bins once more instead of
//                    writing (stop, sums) out and reading them back in k_bin_finalize
// The host enqueues all of it, records an event behind the small D2H copy of the decisions and waits for THAT while close / resolve run: the device is never idle
// for the bin size (round 3: 24 + 12 us of gaps, 25 + 27 + 8 us of one-workgroup kernels, 39 us k_bin_finalize).
#pragma once

struct BinDev {                      // decisions of the sample, in device memory
    int32_t binSize;                 // 0: none (no autosome, non-finite rate, not positive): nothing downstream runs
    int32_t run;                     // 1: k_bin_close2 / k_bin_resolve_fin may run
    unsigned long long binMagic;     // floor(2^64 / binSize) + 1 (0 for binSize 1): rr / binSize without a division in k_bin_close2
    long long total;                 // bins of the sample
    int32_t flags;                   // BD_*: why binSize is 0 / run is 0
    int32_t nAuto;
};
#define BD_NO_AUTOSOME 1
#define BD_BAD_RATE 2                // an autosome without a possible position (rate inf / NaN): the host decides as the reference's sort would
#define BD_TOO_MANY 4                // more autosomes than the device sort holds
#define BD_CAPACITY 8                // total > cap
struct ChromDev { uint32_t rank0, c0, g0, pad; };     // global prefix of possible positions at the chromosome's first tile + popBefore; of the masked hits; of G/C
struct TsPart { uint32_t pop, obs, c, g; };

#define TS_T 1024
#define TS_ITEMS 8
#define TS_CHUNK (TS_T * TS_ITEMS)
#define BD_MAX_AUTO 2048

template <class T> __device__ __forceinline__ void ts_publish(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }      // write-through: read by another XCD's workgroup of the same launch
// arrival ticket: true in the workgroup that arrives last (the idiom of clean_fast.hpp: every wave drains its own stores first, the last workgroup issues one agent-scope
// acquire and then reads what the others published with plain loads).  The ticket is left at zero for the next launch.
__device__ __forceinline__ bool ts_arrive_last(uint32_t* tick, uint32_t expected, int* sFlag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int last = (__hip_atomic_fetch_add(tick, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == expected) ? 1 : 0;
        if (last) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); __hip_atomic_store(tick, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        *sFlag = last;
    }
    __syncthreads();
    return *sFlag != 0;
}
__device__ __forceinline__ TsPart ts_block_sum(TsPart v, TsPart* sh16) {
    v.pop = wave_reduce_add_u32(v.pop); v.obs = wave_reduce_add_u32(v.obs); v.c = wave_reduce_add_u32(v.c); v.g = wave_reduce_add_u32(v.g);
    __syncthreads();
    if (lane_id() == 0) sh16[threadIdx.x >> 6] = v;
    __syncthreads();
    TsPart t = {0, 0, 0, 0};
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) { t.pop += sh16[w].pop; t.obs += sh16[w].obs; t.c += sh16[w].c; t.g += sh16[w].g; }
    return t;
}
// exclusive scan of four uint32 per thread over the 1024-thread workgroup (two calls of the two-value scan, alternating buffers: one barrier each)
__device__ __forceinline__ TsPart ts_block_excl4(TsPart v, U2 (*sh)[16], TsPart& total) {
    U2 a; a.a = v.pop; a.b = v.obs; U2 b; b.a = v.c; b.b = v.g; U2 ta, tb;
    const U2 ea = block_exclusive_scan2_1024(a, sh[0], ta);
    const U2 eb = block_exclusive_scan2_1024(b, sh[1], tb);
    __syncthreads();           // the next call may overwrite sh[0] / sh[1]
    total.pop = ta.a; total.obs = ta.b; total.c = tb.a; total.g = tb.b;
    TsPart r; r.pop = ea.a; r.obs = ea.b; r.c = eb.a; r.g = eb.b;
    return r;
}

__global__ void __launch_bounds__(TS_T) k_tscan_reduce(const BinChrom* __restrict__ ch, int nchr, const unsigned long long* __restrict__ pos0, int packed, int64_t ntiles, int nchunks,
                                                       const uint32_t* __restrict__ tilePop, const uint32_t* __restrict__ tileObs, const uint32_t* __restrict__ tileTotC,
                                                       const uint32_t* __restrict__ tileTotG, TsPart* __restrict__ part, TsPart* __restrict__ partEx,
                                                       unsigned long long* __restrict__ popBefore, uint32_t* __restrict__ tick) {
    __shared__ TsPart shw[16];
    __shared__ U2 sh2[2][16];
    __shared__ unsigned long long shq[16];
    __shared__ int sLast;
    const int b = blockIdx.x, tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    if (b < nchunks) {
        const int64_t v0 = (int64_t)b * TS_CHUNK + (int64_t)tid * TS_ITEMS;
        uint32_t v[TS_ITEMS]; TsPart s = {0, 0, 0, 0};
        load8_u32(tilePop, v0, 0, ntiles, v);
#pragma unroll
        for (int i = 0; i < TS_ITEMS; i++) s.pop += v[i];
        if (tileObs) { load8_u32(tileObs, v0, 0, ntiles, v);
#pragma unroll
            for (int i = 0; i < TS_ITEMS; i++) s.obs += v[i]; }
        load8_u32(tileTotC, v0, 0, ntiles, v);
#pragma unroll
        for (int i = 0; i < TS_ITEMS; i++) s.c += v[i];
        load8_u32(tileTotG, v0, 0, ntiles, v);
#pragma unroll
        for (int i = 0; i < TS_ITEMS; i++) s.g += v[i];
        const TsPart t = ts_block_sum(s, shw);
        if (tid == 0) { ts_publish(&part[b].pop, t.pop); ts_publish(&part[b].obs, t.obs); ts_publish(&part[b].c, t.c); ts_publish(&part[b].g, t.g); }
    } else {
        // possible positions in front of pos0 (they count for the rates, not for the bins): the full tiles before the tile pos0 falls into + the head of that tile
        const int c = b - nchunks;
        const BinChrom C = ch[c];
        const int64_t p0 = (int64_t)pos0[c] < C.len ? (int64_t)pos0[c] : C.len;
        const int64_t t0 = p0 >> TILE_SHIFT;
        unsigned long long before = 0;
        for (int64_t t = tid; t < t0 && t < C.ntiles; t += TS_T) before += tilePop[C.tileBase + t];
        if (w == 0) {
            const int64_t wstart = (t0 << TILE_SHIFT) + (int64_t)l * 64;
            if (wstart < p0) {
                uint64_t mw = packed ? reinterpret_cast<const ulonglong2*>(C.bases)[wstart >> 6].x : C.mask[wstart >> 6];
                const int64_t valid = p0 - wstart;
                if (valid < 64) mw &= (~0ull) >> (64 - valid);
                before += (unsigned long long)__popcll(mw);
            }
        }
        before = wave_reduce_add_u64(before);
        if (l == 0) shq[w] = before;
        __syncthreads();
        if (tid == 0) { unsigned long long s = 0; for (int i = 0; i < 16; i++) s += shq[i]; ts_publish(&popBefore[c], s); }
    }
    if (ts_arrive_last(tick, gridDim.x, &sLast)) {
        TsPart carry = {0, 0, 0, 0};
        for (int base = 0; base < nchunks; base += TS_T) {
            const int i = base + tid;
            TsPart v = {0, 0, 0, 0};
            if (i < nchunks) v = part[i];
            TsPart tot;
            const TsPart ex = ts_block_excl4(v, sh2, tot);
            if (i < nchunks) { TsPart o; o.pop = carry.pop + ex.pop; o.obs = carry.obs + ex.obs; o.c = carry.c + ex.c; o.g = carry.g + ex.g; partEx[i] = o; }
            carry.pop += tot.pop; carry.obs += tot.obs; carry.c += tot.c; carry.g += tot.g;
        }
        if (tid == 0) partEx[nchunks] = carry;       // the genome's totals
    }
}

// ---- the decisions (last workgroup of k_tscan_apply, or k_bin_plan when the bin size comes from the host)
__device__ __forceinline__ unsigned long long ts_block_excl_u64(unsigned long long v, unsigned long long* sh16, unsigned long long& total) {
    unsigned long long inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long o = __shfl_up(inc, d, 64); if (lane_id() >= d) inc += o; }
    __syncthreads();
    if (lane_id() == 63) sh16[threadIdx.x >> 6] = inc;
    __syncthreads();
    unsigned long long off = 0, tot = 0;
    for (int w = 0; w < 16; w++) { if (w < (int)(threadIdx.x >> 6)) off += sh16[w]; tot += sh16[w]; }
    total = tot;
    return inc - v + off;
}
// bins per chromosome and their offsets for a given bin size (the last lines of SampleHitArrays / BinCounts: every chromosome yields floor(possible after pos0 / binSize) bins)
// hostOut / hostBd (optional): the same results written straight into pinned host memory (visible to the host once the kernel has completed: no D2H copy behind the kernel)
__device__ void plan_offsets(int nchr, const ChromOut* __restrict__ totals /* pop, popBefore valid */, ChromOut* __restrict__ dOut, long long* __restrict__ binOffset, BinDev* __restrict__ bd,
                             int binSize, long long cap, int flags, int nAuto, unsigned long long* sh16, ChromOut* __restrict__ hostOut = nullptr, BinDev* __restrict__ hostBd = nullptr,
                             unsigned* __restrict__ hostSeq = nullptr, unsigned seq = 0) {
    const int tid = threadIdx.x;
    const int per = (nchr + TS_T - 1) / TS_T;
    const int cA = tid * per < nchr ? tid * per : nchr, cB = cA + per < nchr ? cA + per : nchr;
    unsigned long long mine = 0;
    if (binSize > 0) for (int c = cA; c < cB; c++) mine += (unsigned long long)((totals[c].pop - totals[c].popBefore) / binSize);
    unsigned long long total;
    unsigned long long run = ts_block_excl_u64(mine, sh16, total);
    for (int c = cA; c < cB; c++) {
        const long long nb = binSize > 0 ? (totals[c].pop - totals[c].popBefore) / binSize : 0;
        binOffset[c] = (long long)run; dOut[c].nbins = nb; run += (unsigned long long)nb;
        if (hostOut) { ChromOut o = totals[c]; o.nbins = nb; hostOut[c] = o; }
    }
    if (hostOut) { __threadfence_system(); __syncthreads(); }      // (every thread's rows are in host memory before thread 0 stamps the mailbox: cvx_mail_publish, common.hpp)
    if (tid == 0) {
        binOffset[nchr] = (long long)total;
        if (binSize > 0 && (long long)total > cap) flags |= BD_CAPACITY;
        BinDev o; o.binSize = binSize; o.run = (binSize > 0 && !(flags & BD_CAPACITY)) ? 1 : 0;
        o.binMagic = binSize > 1 ? ~0ull / (unsigned long long)binSize + 1ull : 0ull; o.total = (long long)total; o.flags = flags; o.nAuto = nAuto;
        *bd = o;
        if (hostBd) { *hostBd = o; cvx_mail_publish(hostSeq, seq); }
    }
}
__global__ void __launch_bounds__(TS_T) k_bin_plan(int nchr, ChromOut* __restrict__ dOut, long long* __restrict__ binOffset, BinDev* __restrict__ bd, int binSize, long long cap) {
    __shared__ unsigned long long sh16[16];
    plan_offsets(nchr, dOut, dOut, binOffset, bd, binSize, cap, 0, 0, sh16);
}

// binSizeArg > 0: given; 0: derived here from the autosomes' rates (counts_per_bin / median rate); < 0: the host decides (chromosome-sharded pipeline: the rates of
// all ranks are exchanged first) and calls k_bin_plan
__global__ void __launch_bounds__(TS_T) k_tscan_apply(const BinChrom* __restrict__ ch, int nchr, int64_t ntiles, int nchunks, const uint32_t* __restrict__ tilePop,
                                                      const uint32_t* __restrict__ tileObs, uint32_t* __restrict__ tileTotC, uint32_t* __restrict__ tileTotG, uint32_t* __restrict__ rankRaw,
                                                      const TsPart* __restrict__ partEx, const unsigned long long* __restrict__ popBefore, TsPart* __restrict__ chrPre,
                                                      const uint8_t* __restrict__ isAuto, int countsPerBin, int binSizeArg, long long cap,
                                                      ChromOut* __restrict__ dOut, ChromDev* __restrict__ chrDev, long long* __restrict__ binOffset, BinDev* __restrict__ bd,
                                                      uint32_t* __restrict__ tick, ChromOut* __restrict__ hostOut, BinDev* __restrict__ hostBd, unsigned* __restrict__ hostSeq, unsigned seq) {
    __shared__ U2 sh2[2][16];
    __shared__ int sLast, sCA;
    __shared__ unsigned long long sh16[16];
    __shared__ double sRate[BD_MAX_AUTO], sSorted[BD_MAX_AUTO];
    __shared__ int sN, sFlags, sBinSize;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t t0 = (int64_t)b * TS_CHUNK, t1 = t0 + TS_CHUNK < ntiles ? t0 + TS_CHUNK : ntiles;
    const int64_t v0 = t0 + (int64_t)tid * TS_ITEMS;
    uint32_t vp[TS_ITEMS], vo[TS_ITEMS], vc[TS_ITEMS], vg[TS_ITEMS];
    TsPart s = {0, 0, 0, 0};
    load8_u32(tilePop, v0, 0, ntiles, vp);
    if (tileObs) load8_u32(tileObs, v0, 0, ntiles, vo);
    else {
#pragma unroll
        for (int i = 0; i < TS_ITEMS; i++) vo[i] = 0;
    }
    load8_u32(tileTotC, v0, 0, ntiles, vc); load8_u32(tileTotG, v0, 0, ntiles, vg);
#pragma unroll
    for (int i = 0; i < TS_ITEMS; i++) { s.pop += vp[i]; s.obs += vo[i]; s.c += vc[i]; s.g += vg[i]; }
    if (tid == 0) {       // first chromosome that starts at or behind this chunk's first tile
        int lo = 0, hi = nchr;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (ch[mid].tileBase < t0) lo = mid + 1; else hi = mid; }
        sCA = lo;
    }
    TsPart tot;
    const TsPart ex = ts_block_excl4(s, sh2, tot);       // (its barriers also publish sCA)
    const TsPart base = partEx[b];
    TsPart run; run.pop = base.pop + ex.pop; run.obs = base.obs + ex.obs; run.c = base.c + ex.c; run.g = base.g + ex.g;
    uint32_t rp[TS_ITEMS], ro[TS_ITEMS], rc[TS_ITEMS], rg[TS_ITEMS];
#pragma unroll
    for (int i = 0; i < TS_ITEMS; i++) { rp[i] = run.pop; ro[i] = run.obs; rc[i] = run.c; rg[i] = run.g; run.pop += vp[i]; run.obs += vo[i]; run.c += vc[i]; run.g += vg[i]; }
    store8_u32(rankRaw, v0, 0, ntiles, rp);
    store8_u32(tileTotC, v0, 0, ntiles, rc);
    store8_u32(tileTotG, v0, 0, ntiles, rg);
    // the prefixes at the first tile of every chromosome that starts in this chunk: published for the last workgroup
    for (int c = sCA; c < nchr; c++) {
        const int64_t tb = ch[c].tileBase;
        if (tb >= t1) break;
        const int64_t rel = tb - t0;
        if ((int)(rel / TS_ITEMS) == tid) {
            const int item = (int)(rel % TS_ITEMS);
            TsPart o = {0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < TS_ITEMS; i++) if (i == item) { o.pop = rp[i]; o.obs = ro[i]; o.c = rc[i]; o.g = rg[i]; }
            ts_publish(&chrPre[c].pop, o.pop); ts_publish(&chrPre[c].obs, o.obs); ts_publish(&chrPre[c].c, o.c); ts_publish(&chrPre[c].g, o.g);
        }
    }
    if (!ts_arrive_last(tick, gridDim.x, &sLast)) return;
    // ---- last workgroup: totals per chromosome (SampleHitArrays.GetRates' inputs), the bin size, the offsets
    const TsPart grand = partEx[nchunks];
    if (tid == 0) { sN = 0; sFlags = 0; sBinSize = binSizeArg > 0 ? binSizeArg : 0; }
    __syncthreads();
    {   // (a thread takes the same chromosomes here and in plan_offsets: it reads back only what it wrote itself)
        const int per = (nchr + TS_T - 1) / TS_T;
        const int cA = tid * per < nchr ? tid * per : nchr, cB = cA + per < nchr ? cA + per : nchr;
        for (int c = cA; c < cB; c++) {
            const TsPart a = chrPre[c], e = c + 1 < nchr ? chrPre[c + 1] : grand;
            const unsigned long long pb = popBefore[c];
            ChromOut o; o.pop = (long long)(uint32_t)(e.pop - a.pop); o.obs = (long long)(uint32_t)(e.obs - a.obs); o.nbins = 0; o.popBefore = (long long)pb;
            dOut[c] = o;
            ChromDev d; d.rank0 = a.pop + (uint32_t)pb; d.c0 = a.c; d.g0 = a.g; d.pad = 0;
            chrDev[c] = d;
            if (binSizeArg == 0 && isAuto[c]) {
                if (o.pop <= 0) atomicOr(&sFlags, BD_BAD_RATE);
                const int slot = atomicAdd(&sN, 1);
                if (slot < BD_MAX_AUTO) sRate[slot] = (double)(int)o.obs / (double)(int)o.pop;          // int / (double)int, CanvasBin.cs:60
            }
        }
    }
    __syncthreads();
    if (binSizeArg == 0) {
        const int n = sN;
        if (n == 0) { if (tid == 0) sFlags |= BD_NO_AUTOSOME; }
        else if (n > BD_MAX_AUTO) { if (tid == 0) sFlags |= BD_TOO_MANY; }
        else if (!(sFlags & BD_BAD_RATE)) {
            // SortedList<double>.Median(): the order of equal values does not matter, so every rate takes the rank "values below it + equal values in front of it"
            for (int i = tid; i < n; i += TS_T) {
                const double r = sRate[i]; int rank = 0;
                for (int j = 0; j < n; j++) { const double q = sRate[j]; rank += (q < r || (q == r && j < i)) ? 1 : 0; }
                sSorted[rank] = r;
            }
            __syncthreads();
            if (tid == 0) {
                const double med = (n % 2) ? sSorted[n / 2] : (sSorted[n / 2 - 1] + sSorted[n / 2]) / 2;
                const double q = (double)countsPerBin / med;                                          // CanvasBin.cs:82
                sBinSize = (q >= 1.0 && q < 2147483648.0) ? (int)q : 0;
            }
        }
        __syncthreads();
    }
    if (binSizeArg < 0) return;        // k_bin_plan follows
    __syncthreads();
    plan_offsets(nchr, dOut, dOut, binOffset, bd, sBinSize, cap, sFlags, sN, sh16, hostOut, hostBd, hostSeq, seq);
}

// ---- k_bin_close with the decisions read on the device.  The record of a bin: where it closes (word start | rank inside the word - 1) and the masked-hit / G/C sums of the
// GENOME in front of that word (global wrap-around prefixes: k_bin_resolve_fin only forms differences)
__global__ void __launch_bounds__(256) k_bin_close2(const BinChrom* __restrict__ ch, const ChromDev* __restrict__ chrDev, int nchr, int64_t ntilesTotal, const uint32_t* __restrict__ S,
                                                    const uint32_t* __restrict__ rankRaw, const uint32_t* __restrict__ tileExC, const uint32_t* __restrict__ tileExG,
                                                    const long long* __restrict__ binOffset, const BinDev* __restrict__ bd,
                                                    uint4* __restrict__ binRec, int32_t* __restrict__ oChr) {
    const int64_t gtile0 = ((int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) * CLOSE_TILES;   // uniform: scalar lookups
    if (gtile0 >= ntilesTotal) return;
    if (!bd->run) return;
    const int binSize = bd->binSize; const unsigned long long binMagic = bd->binMagic;
    const int l = lane_id();
    uint32_t sv[CLOSE_TILES], rk[CLOSE_TILES], tc[CLOSE_TILES], tg[CLOSE_TILES];
#pragma unroll
    for (int t = 0; t < CLOSE_TILES; t++) {
        const bool in = gtile0 + t < ntilesTotal;
        sv[t] = in ? S[(gtile0 + t) * 64 + l] : 0u;
        rk[t] = in ? rankRaw[gtile0 + t] : 0u;
        tc[t] = in ? tileExC[gtile0 + t] : 0u;
        tg[t] = in ? tileExG[gtile0 + t] : 0u;
    }
    int c = find_chrom(ch, nchr, gtile0);
    int64_t tileBase = ch[c].tileBase, tileEnd = tileBase + ch[c].ntiles;
    long long boff = binOffset[c];
    uint32_t rank0 = chrDev[c].rank0;
#pragma unroll
    for (int t = 0; t < CLOSE_TILES; t++) {
        const int64_t gtile = gtile0 + t;
        if (gtile >= ntilesTotal) break;
        while (gtile >= tileEnd) { c++; tileBase = ch[c].tileBase; tileEnd = tileBase + ch[c].ntiles; boff = binOffset[c]; rank0 = chrDev[c].rank0; }
        const uint32_t s = sv[t];
        const uint32_t pop = SUM_POP(s);
        const uint32_t pg = pop | (SUM_GC(s) << 16), cl = SUM_HITS(s);
        const uint32_t pgInc = wave_inclusive_scan_u32(pg), cInc = wave_inclusive_scan_u32(cl);
        const int32_t r = (int32_t)(rk[t] - rank0) + (int32_t)((pgInc - pg) & 0xFFFFu);        // rank before this lane's word (negative in front of pos0)
        if (pop > 0 && r + (int32_t)pop >= binSize) {
            const uint32_t gEx = tg[t] + ((pgInc - pg) >> 16), cEx = tc[t] + (cInc - cl);
            const int64_t wstart = ((gtile - tileBase) << TILE_SHIFT) + (int64_t)l * 64;
            const int32_t rr = r < 0 ? 0 : r;
            uint32_t q = binMagic ? (uint32_t)__umul64hi((unsigned long long)(uint32_t)rr, binMagic) : (uint32_t)rr;      // rr / binSize (see k_bin_close)
            uint32_t nextB = (q + 1u) * (uint32_t)binSize;
            for (; (int64_t)nextB - r <= (int64_t)pop && (int64_t)nextB - r >= 1; q++) {
                const long long bin = boff + (long long)q;
                binRec[bin] = make_uint4((uint32_t)(wstart + ((int64_t)nextB - r - 1)), cEx, gEx, (uint32_t)c);      // one 16-byte store per bin (four 4-byte arrays before: 87 -> .. us)
                oChr[bin] = c;
                nextB += (uint32_t)binSize;
            }
        }
    }
}

// ---- boundary resolution + the bin's five fields.  A workgroup owns RF_NEW consecutive bins and resolves the bin in front of them once more (slot 0): count and GC content are
// differences of the sums at two consecutive stops (CanvasBin.cs:609-640: the running BinState restarts at every stop), start = the previous stop.  Byte arrays: four lanes per
// bin (one 16-byte slice of the word's bases / hits each); packed planes: one lane per bin.  The per-base lines are touched once: non-temporal loads (tools/line_probe.hip:
// a bin costs three 128-byte lines whatever the kernel reads of them, 6.2-6.4 TB/s of line fills = the HBM bound of this kernel; 7 % less with the nt hint).
template <class T> __device__ __forceinline__ uint4 gload_uint4_nt(gptr<T> p) { const canvas_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<gptr<const canvas_u32x4>>(p)); return make_uint4(v.x, v.y, v.z, v.w); }
template <class T> __device__ __forceinline__ ulonglong2 gload_ulonglong2_nt(gptr<T> p) { const canvas_u64x2 v = __builtin_nontemporal_load(reinterpret_cast<gptr<const canvas_u64x2>>(p)); return make_ulonglong2(v.x, v.y); }

// position of the kk-th (1-based) set bit of m
__device__ __forceinline__ uint32_t kth_set_bit(uint64_t m, uint32_t kk) {
    uint32_t pos = 0, cnt;
    cnt = __popc((uint32_t)m);           if (kk > cnt) { kk -= cnt; pos += 32; m >>= 32; }
    cnt = __popc((uint32_t)m & 0xFFFFu); if (kk > cnt) { kk -= cnt; pos += 16; m >>= 16; }
    cnt = __popc((uint32_t)m & 0xFFu);   if (kk > cnt) { kk -= cnt; pos += 8; m >>= 8; }
    cnt = __popc((uint32_t)m & 0xFu);    if (kk > cnt) { kk -= cnt; pos += 4; m >>= 4; }
    cnt = __popc((uint32_t)m & 0x3u);    if (kk > cnt) { kk -= cnt; pos += 2; m >>= 2; }
    cnt = (uint32_t)m & 1u;              if (kk > cnt) { pos += 1; }
    return pos;
}
template <bool PACKED>
__global__ void __launch_bounds__(256) k_bin_resolve_fin(const BinChrom* __restrict__ ch, const ChromDev* __restrict__ chrDev, const long long* __restrict__ binOffset,
                                                         const unsigned long long* __restrict__ pos0, const BinDev* __restrict__ bd, int clampHits,
                                                         const uint4* __restrict__ binRec,
                                                         int32_t* __restrict__ oStart, int32_t* __restrict__ oStop, int32_t* __restrict__ oGc, float* __restrict__ oCount) {
    constexpr int LANES = PACKED ? 1 : 4, SLOTS = 256 / LANES, RF_NEW = SLOTS - 1;
    __shared__ int32_t sStop[SLOTS], sChr[SLOTS]; __shared__ uint32_t sC[SLOTS], sG[SLOTS];
    if (!bd->run) return;
    const long long total = bd->total;
    const long long ngroups = (total + RF_NEW - 1) / RF_NEW;
    const int slot = threadIdx.x / LANES, sub = threadIdx.x % LANES;
    for (long long g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const long long i = g * RF_NEW - 1 + slot;
        const bool live = i >= 0 && i < total;
        int32_t stop = 0; uint32_t aC = 0, aG = 0, hc = 0, hg = 0; int c = 0;
        if (live) {
            const uint4 R = binRec[i];
            const int32_t r = (int32_t)R.x;
            c = (int)R.w;
            aC = R.y; aG = R.z;
            const int64_t p0c = (int64_t)pos0[c];
            const int64_t wstart = (int64_t)(r & ~63);
            uint32_t kk = (uint32_t)(r & 63) + 1u;
            uint64_t mw, valid = ~0ull;
            if (PACKED) {
                const int64_t w = wstart >> 6;
                const gptr<const ulonglong2> hp = reinterpret_cast<gptr<const ulonglong2>>(as_global(ch[c].hits));
                const ulonglong2 rr = gload_ulonglong2_nt(reinterpret_cast<gptr<const ulonglong2>>(as_global(ch[c].bases)) + w);
                const ulonglong2 ha = gload_ulonglong2_nt(hp + 2 * w), hb = gload_ulonglong2_nt(hp + 2 * w + 1);
                valid = pk_valid(wstart, p0c);
                mw = rr.x;
                const uint32_t pos = kth_set_bit(mw, kk);
                const unsigned long long head = valid & ((2ull << pos) - 1ull);          // valid positions <= pos (pos = 63: all of them)
                unsigned long long b0 = ha.x, b1 = ha.y, b2 = hb.x; const unsigned long long b3 = hb.y;
                if (clampHits) pk_clamp10(b0, b1, b2, b3);
                hc = pk_masked_sum(b0, b1, b2, b3, mw & head);
                hg = (uint32_t)__popcll(rr.y & head);
                stop = (int32_t)(wstart + pos + 1);
            } else {
                const gptr<const uint8_t> bases = as_global(ch[c].bases), hits = as_global(ch[c].hits);
                const int64_t len = ch[c].len;
                const int64_t cstart = wstart + 16 * sub;
                mw = __builtin_nontemporal_load(as_global(ch[c].mask) + (wstart >> 6));
                uint32_t wb[4] = {0, 0, 0, 0}, wh[4] = {0, 0, 0, 0};
                if (cstart + 16 <= len) {
                    const uint4 bb = gload_uint4_nt(bases + cstart), hh = gload_uint4_nt(hits + cstart);
                    wb[0] = bb.x; wb[1] = bb.y; wb[2] = bb.z; wb[3] = bb.w; wh[0] = hh.x; wh[1] = hh.y; wh[2] = hh.z; wh[3] = hh.w;
                } else {
                    for (int j = 0; j < 16; j++) { const int64_t pi = cstart + j; if (pi < len) { wb[j >> 2] |= (uint32_t)bases[pi] << (8 * (j & 3)); wh[j >> 2] |= (uint32_t)hits[pi] << (8 * (j & 3)); } }
                }
                if (len - wstart < 64) { valid = (~0ull) >> (64 - (len - wstart)); mw &= valid; }       // positions that carry bin data: pos0 <= p < len
                if (wstart < p0c) valid = (p0c - wstart >= 64) ? 0ull : (valid & ((~0ull) << (p0c - wstart)));
                const uint32_t pos = kth_set_bit(mw, kk);
                const uint64_t head = valid & ((2ull << pos) - 1ull);
                const uint32_t head16 = (uint32_t)(head >> (16 * sub)) & 0xFFFFu, mVal16 = (uint32_t)(mw >> (16 * sub)) & head16;
                uint32_t gc16 = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    gc16 |= marks_to_bits4(gc_marks4(wb[q])) << (4 * q);
                    const uint32_t h = clampHits ? clamp10_bytes(wh[q]) : wh[q];
                    hc = sum_bytes(h & expand4(mVal16 >> (4 * q)), hc);
                }
                hg = __popc(gc16 & head16);
                stop = (int32_t)(wstart + pos + 1);
            }
        }
        if (!PACKED) {       // a quad holds one bin: the head sums of its four slices
            hc += __shfl_xor(hc, 1, 64); hg += __shfl_xor(hg, 1, 64);
            hc += __shfl_xor(hc, 2, 64); hg += __shfl_xor(hg, 2, 64);
        }
        if (sub == 0) { sStop[slot] = stop; sC[slot] = aC + hc; sG[slot] = aG + hg; sChr[slot] = c; }
        __syncthreads();
        {   // slot s > 0 of the group: its five fields from its own stop / sums and those of slot s - 1 (coalesced stores: thread = slot)
            const int s2 = threadIdx.x;
            const long long i2 = g * RF_NEW - 1 + s2;
            if (s2 > 0 && s2 < SLOTS && i2 < total) {
                const int c2 = sChr[s2];
                const bool first = i2 == binOffset[c2];              // first bin of its chromosome: starts at pos0 with empty sums (CanvasBin.cs:582-589)
                const int32_t stop2 = sStop[s2];
                const int32_t start = first ? (int32_t)pos0[c2] : sStop[s2 - 1];
                const uint32_t pC = first ? chrDev[c2].c0 : sC[s2 - 1], pG = first ? chrDev[c2].g0 : sG[s2 - 1];
                const uint32_t count = sC[s2] - pC, gcCount = sG[s2] - pG;
                float gcf = 100.0f * (float)(int32_t)gcCount;         // (int)(100f * GCCount / NucleotideCount), CanvasBin.cs:638
                gcf = gcf / (float)(stop2 - start);
                oStart[i2] = start; oStop[i2] = stop2; oGc[i2] = (int32_t)gcf; oCount[i2] = (float)(int32_t)count;
            }
        }
        __syncthreads();
    }
}
