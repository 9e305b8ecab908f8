// CanvasBin -m GCContentWeighted (mode 5, the Somatic-WGS default; included by bin.hip): CanvasBin.cs:416-506 (read-GC profile), :330-405 (observed vs expected),
// :626-636 (weighted count).  Round 3 ran this chain at 55 ms per 80x genome (~0.06 of its roofline): a 4 B/base GC prefix array materialised in HBM and read twice,
// fragment lengths streamed with 2-byte loads, and the weighted count of a bin added term by term through readlane.  This version:
//   k_nonzero_mean_all Utilities.NonZeroMean of the fragment lengths with 16-byte loads, one launch for all chromosomes            [2 B/base]
//   k_read_gc3 / _gc2 the read-GC profile of a tile (k_read_gc3: mean fragment > 100, one launch; k_read_gc2: the rest, one launch per chromosome) from a GC prefix that only ever exists in LDS: per 64 positions one bit word + one running count, for the tile and
//                     a halo of 3 x meanFragment positions behind it; the count of the window [pos, pos + cur) is a difference of two prefix values.  Histograms of
//                     ComputeObservedVsExpectedGC in LDS, flushed once per (persistent) workgroup into 16 replicas                 [~(1 + halo) + 2 + 1 read, 1 written B/base]
//   k_gcw_words_all + k_bin_weighted3   per bin: the terms min(10, hit / weight[readGC]) are floats, so their sum in double is exact in any order; the reference adds them in float32, in
//                     position order — its result lies within n * 2^-24 * sum of the exact sum (every partial sum is at most the final one: the terms are not negative).
//                     When that interval does not contain a value that rounds differently, (int)Math.Round is decided; the other bins (a few per thousand) replay the
//                     reference's additions one by one                                                                            [1/8 + 1 + 1 B/base]
#pragma once

// ---- gcContent[pos] (CanvasBin.cs:466-492) + the two 101-bin histograms of ComputeObservedVsExpectedGC (:349-356)
// A tile of RG_T positions per workgroup round:
//   1. the GC prefix of [tile, tile + 3 meanFragment): one bit word + one running count per 64 positions, in LDS only;
//   2. per half tile: (a) every position as if it started no fragment (length 0 -> window [pos, pos + meanFragment)): the count slides by one position, i.e. follows from the
//      previous one with the two bits at the window's ends, one prefix lookup per 16 positions; the positions that DO carry a fragment length are collected in an LDS list;
//      (b) the list is worked off by all lanes (two prefix lookups and a division each) and patches the bytes of step (a); (c) histograms and the 16-byte stores.
// History (80x genome): prefix array in HBM + two kernels 24 ms (round 3); LDS prefix, two lookups and a division per position 7 ms (VALU-bound); sliding counts, but the
// positions with a length handled by their own lane in a count-trailing-zeros loop 7.6 ms (3.6 ms of it that loop: nine rounds per wave for the lane with the most of them).
#define RG_T 8192                      // positions per tile
#define RG_HALF 4096
#define RG_LREP 4                      // histogram replicas in LDS
#define RG_REP 16                      // histogram replicas in global memory (workgroup % RG_REP): a single set of 202 counters would be a serial chain of atomics
template <class PB>
__device__ __forceinline__ uint64_t gc_bits64(PB bases, int64_t p, int64_t len) {
    uint64_t g = 0;
    if (p + 64 <= len) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint4 v = gload_uint4(bases + p + 16 * q);
            const uint64_t b16 = (uint64_t)(marks_to_bits4(gc_marks4(v.x)) | (marks_to_bits4(gc_marks4(v.y)) << 4) | (marks_to_bits4(gc_marks4(v.z)) << 8) | (marks_to_bits4(gc_marks4(v.w)) << 12));
            g |= b16 << (16 * q);
        }
    } else {
        for (int i = 0; i < 64 && p + i < len; i++) { const uint8_t b = bases[p + i] | 0x20; if (b == 'c' || b == 'g') g |= 1ull << i; }
    }
    return g;
}
// dynamic LDS: uint64 sBits[nWmax], then uint32 sCum[nWmax]   (nWmax = (RG_T + 3 meanFragment) / 64 + 2: a kilobyte or two for real fragment sizes, 20 KB for the largest Int16)
__global__ void __launch_bounds__(256) k_read_gc2(const uint8_t* __restrict__ bases, const int16_t* __restrict__ fl, const uint8_t* __restrict__ hits, int64_t len, int meanFrag,
                                                  unsigned long long mean40 /* ceil(2^40 / meanFrag): x * mean40 >> 40 == x / meanFrag for x < 2^22 */, int nWmax,
                                                  uint8_t* __restrict__ readGc, unsigned long long* __restrict__ histRep) {
    extern __shared__ uint64_t sDyn[];
    uint64_t* __restrict__ sBits = sDyn; uint32_t* __restrict__ sCum = reinterpret_cast<uint32_t*>(sDyn + nWmax);
    __shared__ unsigned int le[RG_LREP][101], lo[RG_LREP][101];
    __shared__ uint32_t sWave[4];
    __shared__ __attribute__((aligned(16))) uint8_t sG[RG_HALF];         // gcContent of the half tile
    __shared__ uint32_t sList[RG_HALF];                                 // positions with a fragment length: (offset in the tile) << 16 | length
    __shared__ uint32_t sListN;
    const int tid = threadIdx.x;
    for (int i = tid; i < RG_LREP * 101; i += 256) { (&le[0][0])[i] = 0; (&lo[0][0])[i] = 0; }
    unsigned int* __restrict__ myE = le[tid % RG_LREP]; unsigned int* __restrict__ myO = lo[tid % RG_LREP];
    const int H = 3 * meanFrag;
    const int64_t lim = len - (int64_t)H - 1;                      // positions from `lim` on keep gcContent 0 (the loop of CanvasBin.cs:466 stops there)
    const int64_t ntile = (len + RG_T - 1) / RG_T;
    for (int64_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int64_t t0 = tile * RG_T;
        const int64_t tEnd = t0 + RG_T < len ? t0 + RG_T : len;
        // ---- 1. the GC prefix of [t0, tEnd + H)
        const int64_t last = tEnd + H < len ? tEnd + H : len;
        const int nW = (int)((last - t0 + 63) >> 6) + 1;           // (+1: a window may end exactly on the word behind the last one)
        __syncthreads();                                           // the previous tile's readers are done with sBits / sCum
        uint32_t carry = 0;
        for (int base = 0; base < nW; base += 256) {
            const int w = base + tid;
            uint64_t g = 0;
            if (w < nW) { const int64_t p = t0 + ((int64_t)w << 6); if (p < len) g = gc_bits64(as_global(bases), p, len); sBits[w] = g; }
            const uint32_t cnt = (uint32_t)__popcll(g);
            const uint32_t inc = wave_inclusive_scan_u32(cnt);
            if ((tid & 63) == 63) sWave[tid >> 6] = inc;
            __syncthreads();
            uint32_t off = carry, tot = 0;
            for (int k = 0; k < 4; k++) { if (k < (tid >> 6)) off += sWave[k]; tot += sWave[k]; }
            if (w < nW) sCum[w] = off + inc - cnt;
            carry += tot;
            __syncthreads();
        }
        // ---- 2. the two halves of the tile, one 16-position group per thread
        for (int half = 0; half < RG_T / RG_HALF; half++) {
            const int64_t p = t0 + (int64_t)half * RG_HALF + 16 * (int64_t)tid;
            if (tid == 0) sListN = 0;
            __syncthreads();
            uint32_t hw[4] = {0, 0, 0, 0};
            const bool in = p < tEnd;
            if (in) {
                uint32_t fw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                if (p + 16 <= len) {
                    const uint4 h = *reinterpret_cast<const uint4*>(hits + p), f0 = *reinterpret_cast<const uint4*>(fl + p), f1 = *reinterpret_cast<const uint4*>(fl + p + 8);
                    hw[0] = h.x; hw[1] = h.y; hw[2] = h.z; hw[3] = h.w;
                    fw[0] = f0.x; fw[1] = f0.y; fw[2] = f0.z; fw[3] = f0.w; fw[4] = f1.x; fw[5] = f1.y; fw[6] = f1.z; fw[7] = f1.w;
                } else {
                    for (int j = 0; j < 16 && p + j < len; j++) { hw[j >> 2] |= (uint32_t)hits[p + j] << (8 * (j & 3)); fw[j >> 1] |= (uint32_t)(uint16_t)fl[p + j] << (16 * (j & 1)); }
                }
                const int a0 = (int)(p - t0);                                    // multiple of 16: the 16 bits at a0 lie inside one word
                const int nlim = lim - p >= 16 ? 16 : (lim - p > 0 ? (int)(lim - p) : 0);      // positions of this group in front of `lim`
                uint32_t out[4] = {0, 0, 0, 0};
                if (nlim > 0) {
                    const int b0 = a0 + meanFrag;
                    const uint32_t bitsA = (uint32_t)(sBits[a0 >> 6] >> (a0 & 63)) & 0xFFFFu;
                    const uint64_t wlo = sBits[b0 >> 6], whi = sBits[(b0 >> 6) + 1];
                    const uint32_t bitsB = (uint32_t)(((b0 & 63) ? (wlo >> (b0 & 63)) | (whi << (64 - (b0 & 63))) : wlo) & 0xFFFFull);
                    uint32_t cnt = (sCum[b0 >> 6] + (uint32_t)__popcll(wlo & ((1ull << (b0 & 63)) - 1ull))) - (sCum[a0 >> 6] + (uint32_t)__popcll(sBits[a0 >> 6] & ((1ull << (a0 & 63)) - 1ull)));
#pragma unroll
                    for (int j = 0; j < 16; j++) {
                        const uint32_t v = (uint32_t)(((unsigned long long)(100u * cnt) * mean40) >> 40);      // 100 * gcCounter / meanFragmentSize (exact: 100 * cnt < 2^22, see the host)
                        const uint32_t g = j < nlim ? (v < 101u ? v : 101u) : 0u;
                        out[j >> 2] |= g << (8 * (j & 3));
                        cnt += ((bitsB >> j) & 1u) - ((bitsA >> j) & 1u);
                    }
                    // the positions in front of `lim` that carry a fragment length go on the list
                    uint32_t nzf = 0;
#pragma unroll
                    for (int q = 0; q < 8; q++) nzf |= ((fw[q] & 0xFFFFu) ? 1u : 0u) << (2 * q) | ((fw[q] >> 16) ? 1u : 0u) << (2 * q + 1);
                    nzf &= nlim >= 16 ? 0xFFFFu : ((1u << nlim) - 1u);
                    if (nzf) {
                        uint32_t at = atomicAdd(&sListN, (uint32_t)__popc(nzf));
#pragma unroll
                        for (int j = 0; j < 16; j++) if ((nzf >> j) & 1u) sList[at++] = ((uint32_t)(a0 + j) << 16) | ((fw[j >> 1] >> (16 * (j & 1))) & 0xFFFFu);
                    }
                }
                *reinterpret_cast<uint4*>(sG + 16 * tid) = make_uint4(out[0], out[1], out[2], out[3]);
            }
            __syncthreads();
            // (b) the listed positions, all lanes busy: window [pos, pos + min(length, 3 meanFragment))
            const int nl = (int)sListN;
            for (int k = tid; k < nl; k += 256) {
                const uint32_t ent = sList[k];
                const int a = (int)(ent >> 16), f = (int)(int16_t)(ent & 0xFFFFu);
                const int cur = f < H ? f : H;
                uint32_t g = 0;
                if (cur > 0) {                                                   // (a negative length: an empty window in the reference, gcContent 0)
                    const int b = a + cur;
                    const uint32_t pa = sCum[a >> 6] + (uint32_t)__popcll(sBits[a >> 6] & ((1ull << (a & 63)) - 1ull));
                    const uint32_t pb = sCum[b >> 6] + (uint32_t)__popcll(sBits[b >> 6] & ((1ull << (b & 63)) - 1ull));
                    const uint32_t v = 100u * (pb - pa) / (uint32_t)cur;         // 100 * gcCounter / currentFragment
                    g = v < 101u ? v : 101u;
                }
                sG[a - half * RG_HALF] = (uint8_t)g;
            }
            __syncthreads();
            // (c) histograms (runs of equal gcContent inside the 16 positions share one pair of updates) and the store
            if (in) {
                const uint4 o4 = *reinterpret_cast<const uint4*>(sG + 16 * tid);
                const uint32_t out[4] = {o4.x, o4.y, o4.z, o4.w};
                uint32_t runG = 0xFFFFFFFFu, runN = 0, runObs = 0;
                const int nval = len - p >= 16 ? 16 : (int)(len - p);
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    if (j < nval) {
                        const uint32_t g = (out[j >> 2] >> (8 * (j & 3))) & 0xFFu;
                        const uint32_t h = (hw[j >> 2] >> (8 * (j & 3))) & 0xFFu;
                        if (g != runG) {
                            if (runN && runG < 101u) { atomicAdd(&myE[runG], runN); if (runObs) atomicAdd(&myO[runG], runObs); }
                            runG = g; runN = 0; runObs = 0;
                        }
                        runN++; runObs += h;
                    }
                }
                if (runN && runG < 101u) { atomicAdd(&myE[runG], runN); if (runObs) atomicAdd(&myO[runG], runObs); }
                if (p + 16 <= len) *reinterpret_cast<uint4*>(readGc + p) = o4;
                else for (int j = 0; j < 16 && p + j < len; j++) readGc[p + j] = (uint8_t)((out[j >> 2] >> (8 * (j & 3))) & 0xFFu);
            }
        }
    }
    __syncthreads();
    unsigned long long* rep = histRep + (size_t)(blockIdx.x % RG_REP) * 202;
    if (tid < 101) {
        unsigned long long e = 0, o = 0;
        for (int r = 0; r < RG_LREP; r++) { e += le[r][tid]; o += lo[r][tid]; }
        if (e) atomicAdd(&rep[tid], e);
        if (o) atomicAdd(&rep[101 + tid], o);
    }
}

// ---- Utilities.NonZeroMean(Int16[]) (CanvasCommon/Utilities.cs:135-151): sum and number of the positive lengths, per chromosome (k_nonzero_mean_all), and the read-GC
// profile (k_read_gc3) in one launch each for the whole genome: the per-chromosome launches of round 3 paid 37 us each for the 2 048 workgroups' atomics on the
// chromosome's two counters (same-line atomics serialise at ~12 ns) and a ramp / tail per launch.  A tile is RG_T positions of one chromosome; RgChrom::tile0 numbers them.
struct RgChrom { const uint8_t* bases; const int16_t* fl; const uint8_t* hits; uint8_t* readGc; int64_t len; int64_t tile0; };
#define NZ_REP 32                      // replicas of a chromosome's {sum, count} (workgroup % NZ_REP)
#define RG_REP_ALL 64                  // histogram replicas of k_read_gc3 (the host adds up RG_REP_ALL of them; k_read_gc2 uses the first RG_REP)
__device__ __forceinline__ int rg_find_chrom(const RgChrom* __restrict__ ch, int nchr, int64_t tile) {      // the chromosome whose tiles contain `tile`
    int lo = 0, hi = nchr - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (ch[mid].tile0 <= tile) lo = mid; else hi = mid - 1; }
    return lo;
}
__global__ void __launch_bounds__(256) k_nonzero_mean_all(const RgChrom* __restrict__ ch, int nchr, int64_t ntileAll, unsigned long long* __restrict__ sumCnt /* [nchr][NZ_REP][2] */) {
    __shared__ unsigned long long sh[2][4];
    unsigned long long s = 0, c = 0;
    int cur = -1;
    auto flush = [&]() {
        if (cur < 0) return;                                        // (uniform over the workgroup)
        s = wave_reduce_add_u64(s); c = wave_reduce_add_u64(c);
        __syncthreads();
        if (lane_id() == 0) { sh[0][threadIdx.x >> 6] = s; sh[1][threadIdx.x >> 6] = c; }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long* dst = sumCnt + ((size_t)cur * NZ_REP + blockIdx.x % NZ_REP) * 2;
            atomicAdd(&dst[0], sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]); atomicAdd(&dst[1], sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3]);
        }
        s = 0; c = 0;
    };
    for (int64_t tile = blockIdx.x; tile < ntileAll; tile += gridDim.x) {
        const int ci = rg_find_chrom(ch, nchr, tile);
        if (ci != cur) { flush(); cur = ci; }
        const int16_t* __restrict__ fl = ch[ci].fl; const int64_t len = ch[ci].len;
        const int64_t p0 = (tile - ch[ci].tile0) * RG_T;
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {                                // RG_T positions = 4 x 256 x 8 lengths
            const int64_t p = p0 + ((int64_t)(u * 256 + (int)threadIdx.x) << 3);
            if (p + 8 <= len) v[u] = *reinterpret_cast<const uint4*>(fl + p);
            else {
                uint32_t w[4] = {0, 0, 0, 0};
                for (int j = 0; j < 8 && p + j < len; j++) w[j >> 1] |= (uint32_t)(uint16_t)fl[p + j] << (16 * (j & 1));
                v[u] = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
            uint32_t s32 = 0, c32 = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int a = (int)(int16_t)(w[q] & 0xFFFFu), b = (int)(int16_t)(w[q] >> 16);
                if (a > 0) { s32 += (uint32_t)a; c32++; }
                if (b > 0) { s32 += (uint32_t)b; c32++; }
            }
            s += s32; c += c32;
        }
    }
    flush();
}

// ---- k_read_gc3: the same result as k_read_gc2 for meanFragment > 100.  k_read_gc2 took 6.5 ms per 80x genome (2.5x its HBM time) and a first rewrite with a third of its
// instructions took exactly as long: the kernel is a chain of phases — load, barrier, compute, barrier, list, barrier, store — in which a workgroup waits for its own loads, and
// four workgroups per CU do not cover that.  Here the workgroup only builds the tile's GC prefix together; then every WAVE works through its own quarter of the tile in rounds of
// 512 positions (8 per lane) with no workgroup barrier, the next round's hits and fragment lengths already in flight:
//   (a) every position as if it carried no fragment length: with x = 100 * count, v = x / meanFragment and r = x % meanFragment the count moves by d in {-1, 0, 1} from one
//       position to the next (round 5: the eight counts of a lane come from one multiplication and v from a float multiply-add; the remainder chain r += 100 d of
//       round 4 is kept for mean fragments beyond RG3_FAST_M).  32 histogram replicas laid out [gcContent][lane % 32]: a lane owns its LDS bank, the atomics of a
//       wave never meet; every position adds 1 to the expected histogram, the observed one is fed by step (b) alone;
//   (b) positions WITH a fragment length are patches of that default: they go on a wave-private list as (offset, length, hits) in one word (wave prefix sum of the lanes'
//       counts, no atomics), all lanes then work the list off: two lookups in the {32 GC bits, running count} table, a float reciprocal with an exact correction
//       (quotient <= 100, operands < 2^24), the byte replaced, the expected histogram corrected by -default +actual (32-bit counters in modular arithmetic, flushed
//       before they can wrap) and the hits added to the observed one.  Positions with a hit but no length are on the list too (window = the default);
//   (c) 8-byte stores.
#define RG3_HE 16                      // histogram replicas in LDS, [gcContent][lane % 16]: measured against 32 / 32 (four workgroups per CU instead of five: 5.5 vs 5.0 ms) and
#define RG3_HO 16                      // 32 / 8 (same time within the spread between boxes)
#define RG3_FLUSH_TILES 512            // 512 x 8192 positions x 255 hits < 2^32
#define RG3_WR 512                     // positions of a wave round
#define RG3_FAST_M 16384               // largest mean fragment whose default-window values are computed in float (see step (a)); the others carry a remainder along
__device__ __forceinline__ uint32_t rg3_spread8(uint32_t x) {          // bit j of the low byte -> bit 0 of nibble j
    uint32_t s = (x | (x << 12)) & 0x000F000Fu;
    s = (s | (s << 6)) & 0x03030303u;
    return (s | (s << 3)) & 0x11111111u;
}
__device__ __forceinline__ uint32_t rg3_prefix(const uint2* __restrict__ sPre, int a) {      // GC positions in [tile start, tile start + a)
    const uint2 e = sPre[a >> 5];
    return e.y + (uint32_t)__popc(e.x & ((1u << (a & 31)) - 1u));
}
__device__ __forceinline__ uint32_t rg3_div100(uint32_t c, uint32_t cur) {                  // 100 * c / cur for c <= cur < 2^17 (100 c < 2^24: exact as a float)
    const uint32_t x = 100u * c;
    uint32_t q = (uint32_t)((float)x * __builtin_amdgcn_rcpf((float)cur));                  // within 1 of the quotient (which is <= 100)
    const int32_t r = (int32_t)x - (int32_t)(q * cur);
    if (r < 0) q--; else if ((uint32_t)r >= cur) q++;
    return q;
}
template <class PH, class PF>
__device__ __forceinline__ void rg3_load8(PH hits, PF fl, int64_t p, int64_t len, uint32_t (&hw)[2], uint32_t (&fw)[4]) {
    hw[0] = hw[1] = 0; fw[0] = fw[1] = fw[2] = fw[3] = 0;
    if (p + 8 <= len) {
        const uint2 h = gload_uint2(hits + p); const uint4 f = gload_uint4(fl + p);
        hw[0] = h.x; hw[1] = h.y; fw[0] = f.x; fw[1] = f.y; fw[2] = f.z; fw[3] = f.w;
    } else {
        for (int j = 0; j < 8 && p + j < len; j++) { hw[j >> 2] |= (uint32_t)hits[p + j] << (8 * (j & 3)); fw[j >> 1] |= (uint32_t)(uint16_t)fl[p + j] << (16 * (j & 1)); }
    }
}
// dynamic LDS: uint2 sPre[2 * nWmax]  ({GC bits of 32 positions, GC positions of the tile in front of them})
template <int HE, int HO>          // replicas of the expected / observed histogram in LDS (RG3_HE, RG3_HO)
__global__ void __launch_bounds__(256) k_read_gc3(const RgChrom* __restrict__ ch, int nchr, int64_t ntileAll, int meanFrag, unsigned long long mean40, int nWmax,
                                                  unsigned long long* __restrict__ histRep) {
    extern __shared__ __attribute__((aligned(16))) uint2 sPre[];
    __shared__ unsigned int sHist[101 * HE], sHistO[101 * HO];       // expected [101][HE] (eight atomics per lane and round + the patches), observed [101][HO] (fed by the list only)
    __shared__ uint32_t sWave[4];
    __shared__ __attribute__((aligned(16))) uint8_t sGW[4][RG3_WR];      // gcContent of the wave's round
    __shared__ uint2 sListW[4][RG3_WR / 2];                             // the wave's positions with a fragment length or a hit, half a round at a time: {length | hits << 16, offset in the round}
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t laneR = (uint32_t)tid & (HE - 1u), laneO = (uint32_t)tid & (HO - 1u);
    for (int i = tid; i < 101 * HE; i += 256) sHist[i] = 0;
    for (int i = tid; i < 101 * HO; i += 256) sHistO[i] = 0;
    const int M = meanFrag, H = 3 * meanFrag;
    const bool fastF = M <= RG3_FAST_M;
    const float rMean = 1.0f / (float)M, c100 = (float)(100.0 / (double)M);        // (both correctly rounded: IEEE division)
    unsigned long long* rep = histRep + (size_t)(blockIdx.x % RG_REP_ALL) * 202;
    uint8_t* __restrict__ sG = sGW[wv]; uint2* __restrict__ sList = sListW[wv];
    int sinceFlush = 0;
    for (int64_t tile = blockIdx.x; tile < ntileAll; tile += gridDim.x) {
        const int ci = rg_find_chrom(ch, nchr, tile);
        const gptr<const uint8_t> bases = as_global(ch[ci].bases); const gptr<const int16_t> fl = as_global(ch[ci].fl); const gptr<const uint8_t> hits = as_global(ch[ci].hits); const gptr<uint8_t> readGc = as_global(ch[ci].readGc);
        const int64_t len = ch[ci].len;
        const int64_t lim = len - (int64_t)H - 1;                  // positions from `lim` on keep gcContent 0 (the loop of CanvasBin.cs:466 stops there)
        const int64_t t0 = (tile - ch[ci].tile0) * RG_T;
        const int64_t tEnd = t0 + RG_T < len ? t0 + RG_T : len;
        const int64_t wBase = t0 + (int64_t)wv * (RG_T / 4);       // this wave's quarter of the tile
        // the first round's hits and lengths are requested before the prefix is built
        uint32_t hwN[2], fwN[4];
        rg3_load8(hits, fl, wBase + 8 * lane, wBase + 8 * lane < tEnd ? len : 0, hwN, fwN);
        // ---- 1. the GC prefix of [t0, tEnd + H)
        const int64_t last = tEnd + H < len ? tEnd + H : len;
        const int nW = (int)((last - t0 + 63) >> 6) + 1;           // 64-position words (+1: a window may end exactly on the word behind the last one)
        __syncthreads();                                           // the previous tile's readers are done with sPre (and the zeroing of sHist is visible)
        if (sinceFlush == RG3_FLUSH_TILES) {
            if (tid < 101) {
                unsigned int e = 0, o = 0;
                for (int r = 0; r < HE; r++) { e += sHist[tid * HE + r]; sHist[tid * HE + r] = 0; }
                for (int r = 0; r < HO; r++) { o += sHistO[tid * HO + r]; sHistO[tid * HO + r] = 0; }
                if (e) atomicAdd(&rep[tid], (unsigned long long)e);
                if (o) atomicAdd(&rep[101 + tid], (unsigned long long)o);
            }
            sinceFlush = 0;
            __syncthreads();
        }
        sinceFlush++;
        uint32_t carry = 0;
        for (int base = 0; base < nW; base += 256) {
            const int w = base + tid;
            uint64_t g = 0;
            if (w < nW) { const int64_t p = t0 + ((int64_t)w << 6); if (p < len) g = gc_bits64(bases, p, len); }
            const uint32_t cnt = (uint32_t)__popcll(g);
            const uint32_t inc = wave_inclusive_scan_u32(cnt);
            if ((tid & 63) == 63) sWave[tid >> 6] = inc;
            __syncthreads();
            uint32_t off = carry, tot = 0;
            for (int k = 0; k < 4; k++) { if (k < (tid >> 6)) off += sWave[k]; tot += sWave[k]; }
            if (w < nW) {
                const uint32_t c0 = off + inc - cnt, glo = (uint32_t)g;
                *reinterpret_cast<uint4*>(&sPre[2 * w]) = make_uint4(glo, c0, (uint32_t)(g >> 32), c0 + (uint32_t)__popc(glo));
            }
            carry += tot;
            __syncthreads();
        }
        // ---- 2. the wave's rounds
        for (int rd = 0; rd < RG_T / 4 / RG3_WR; rd++) {
            const int64_t rBase = wBase + (int64_t)rd * RG3_WR;
            if (rBase >= tEnd) break;                                // (uniform over the wave)
            const int64_t p = rBase + 8 * lane;
            const bool in = p < tEnd;
            uint32_t hw[2] = {hwN[0], hwN[1]}, fw[4] = {fwN[0], fwN[1], fwN[2], fwN[3]};
            if (rd + 1 < RG_T / 4 / RG3_WR) rg3_load8(hits, fl, p + RG3_WR, p + RG3_WR < tEnd ? len : 0, hwN, fwN);      // the next round's, in flight during this one
            const int a0 = (int)(p - t0);                                        // multiple of 8: the 8 bits at a0 lie inside one 32-bit word
            uint32_t out[2] = {0, 0};
            if (in) {
                int nlim = 8;
                if (p + 8 <= lim) {
                    // (a) window [pos, pos + meanFragment), followed from position to position
                    const int b0 = a0 + M;
                    const uint32_t bitsA = (sPre[a0 >> 5].x >> (a0 & 31)) & 0xFFu;
                    const uint2 eB = sPre[b0 >> 5];
                    const uint32_t bitsB = __builtin_amdgcn_alignbit(sPre[(b0 >> 5) + 1].x, eB.x, (uint32_t)(b0 & 31)) & 0xFFu;
                    const uint32_t cnt = (eB.y + (uint32_t)__popc(eB.x & ((1u << (b0 & 31)) - 1u))) - rg3_prefix(sPre, a0);
                    if (fastF) {
                        // the eight counts at once: nibble j of `dex` = 8 + (GC positions entering - leaving the window over the j positions in front of position j), from
                        // the two ends' bits spread to one per nibble, subtracted and summed up by ONE multiplication (the nibbles stay in [1, 15]: no borrow between them);
                        // v = floor((100 count + 0.5) / meanFragment) in float: the operand is exact (< 2^23), the quotient is off by < 1.8e-5 and 0.5 / meanFragment away
                        // from the next integer (meanFragment <= RG3_FAST_M; tests/test_gcw_formula.py walks every boundary) — two instructions per position, no carry of
                        // a remainder from one position to the next
                        const uint32_t dex = (((rg3_spread8(bitsB) - rg3_spread8(bitsA)) * 0x11111111u + 0x88888888u) << 4) | 8u;
                        const uint32_t de = dex & 0x0F0F0F0Fu, dod = (dex >> 4) & 0x0F0F0F0Fu;
                        const float bf = ((float)(int32_t)(100u * cnt) - 799.5f) * rMean;
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            const float df = (float)((((j & 1) ? dod : de) >> (8 * (j >> 1))) & 0xFFu);
                            const uint32_t v = (uint32_t)__builtin_fmaf(df, c100, bf);
                            out[j >> 2] |= v << (8 * (j & 3));
                            atomicAdd(&sHist[v * HE + laneR], 1u);
                        }
                    } else {
                        const uint32_t x = 100u * cnt;
                        uint32_t v = (uint32_t)(((unsigned long long)x * mean40) >> 40);      // 100 * gcCounter / meanFragmentSize (exact: x < 2^22, see the host)
                        int32_t r = (int32_t)(x - v * (uint32_t)M);
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            out[j >> 2] |= v << (8 * (j & 3));
                            atomicAdd(&sHist[v * HE + laneR], 1u);
                            r = __mul24(100, (int32_t)((bitsB >> j) & 1u) - (int32_t)((bitsA >> j) & 1u)) + r;
                            const int32_t adj = (r >> 31) - ((M - 1 - r) >> 31);       // +1: r >= M, -1: r < 0
                            r -= __mul24(adj, M);
                            v += (uint32_t)adj;
                        }
                    }
                } else {
                    // the last groups of the chromosome: positions behind `lim` keep 0, positions behind `len` do not exist
                    nlim = lim - p > 0 ? (int)(lim - p) : 0;
                    const int nval = len - p >= 8 ? 8 : (int)(len - p);
                    for (int j = 0; j < nval; j++) {
                        uint32_t g = 0;
                        if (j < nlim) g = (uint32_t)(((unsigned long long)(100u * (rg3_prefix(sPre, a0 + j + M) - rg3_prefix(sPre, a0 + j))) * mean40) >> 40);
                        out[j >> 2] |= g << (8 * (j & 3));
                        atomicAdd(&sHist[g * HE + laneR], 1u);
                        if (j >= nlim) atomicAdd(&sHistO[g * HO + laneO], (hw[j >> 2] >> (8 * (j & 3))) & 0xFFu);      // (in front of `lim` a hit is a list entry)
                    }
                }
                if (nlim < 8) {                                                   // positions from `lim` on are no list entries: their lengths and hits are dropped here
#pragma unroll
                    for (int j = 0; j < 8; j++) if (j >= nlim) { fw[j >> 1] &= ~(0xFFFFu << (16 * (j & 1))); hw[j >> 2] &= ~(0xFFu << (8 * (j & 3))); }
                }
            }
            *reinterpret_cast<uint2*>(sG + 8 * lane) = make_uint2(out[0], out[1]);
            // the positions that carry a fragment length or a hit, position j of every lane at a time: the lanes' ranks come from the comparison's lane mask (no per-lane
            // bit set, no scan), an entry is {length and hits as they were loaded (one v_perm), offset in the round} — six instructions per position.  The observed histogram
            // is fed from this list alone (step (b)): a position without a hit adds nothing to it, and four positions in five have neither — step (a) is left with one
            // atomic per position.  A hit without a fragment length is an entry whose window is the default one.  Two half rounds (positions 0-3 and 4-7 of every lane),
            // so that the list is 2 KB per wave.
            const int aR = (int)(rBase - t0);
#pragma unroll
            for (int hb = 0; hb < 2; hb++) {
                uint32_t nl = 0;
#pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                    const int j = 4 * hb + jj;
                    const uint32_t lo = __builtin_amdgcn_perm(hw[j >> 2], fw[j >> 1], 0x0C000000u | ((4u + (j & 3)) << 16) | ((2u * (j & 1) + 1u) << 8) | (2u * (j & 1)));
                    const unsigned long long has = __ballot(lo != 0u);
                    if (lo != 0u) {
                        const uint32_t at = nl + __builtin_amdgcn_mbcnt_hi((uint32_t)(has >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)has, 0u));
                        sList[at] = make_uint2(lo, (uint32_t)(8 * lane + j));
                    }
                    nl += (uint32_t)__popcll(has);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                // (b) the listed positions, all lanes busy: window [pos, pos + min(length, 3 meanFragment)); the default value is replaced, the histograms follow
                for (int k = lane; k < (int)nl; k += 64) {
                    const uint2 ent = sList[k];
                    const int o = (int)ent.y, f = (int)(int16_t)(ent.x & 0xFFFFu);
                    const uint32_t h = ent.x >> 16;
                    const int a = aR + o;
                    const int cur = f > 0 ? (f < H ? f : H) : (f < 0 ? 0 : M);          // (a negative length: an empty window in the reference, gcContent 0)
                    uint32_t g = 0;
                    if (cur > 0) g = rg3_div100(rg3_prefix(sPre, a + cur) - rg3_prefix(sPre, a), (uint32_t)cur);
                    const uint32_t gd = sG[o];
                    if (g != gd) {
                        sG[o] = (uint8_t)g;
                        atomicAdd(&sHist[gd * HE + laneR], 0xFFFFFFFFu);
                        atomicAdd(&sHist[g * HE + laneR], 1u);
                    }
                    if (h) atomicAdd(&sHistO[g * HO + laneO], h);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            }
            // (c) the store
            if (in) {
                const uint2 o2 = *reinterpret_cast<const uint2*>(sG + 8 * lane);
                if (p + 8 <= len) gstore_uint2(readGc + p, o2);
                else { const uint32_t o[2] = {o2.x, o2.y}; for (int j = 0; j < 8 && p + j < len; j++) readGc[p + j] = (uint8_t)((o[j >> 2] >> (8 * (j & 3))) & 0xFFu); }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    if (tid < 101) {
        unsigned int e = 0, o = 0;
        for (int r = 0; r < HE; r++) e += sHist[tid * HE + r];
        for (int r = 0; r < HO; r++) o += sHistO[tid * HO + r];
        if (e) atomicAdd(&rep[tid], (unsigned long long)e);
        if (o) atomicAdd(&rep[101 + tid], (unsigned long long)o);
    }
}

// ---- weighted count (CanvasBin.cs:626-636): tmp += Math.Min(10, (float)hit / weight[readGC]) over the possible positions in position order (float32), then (int)Math.Round
#define GCW_REP 32
// the reference's own order, 64 positions (one mask word) per step: the non-zero terms added one by one (a position without a hit adds +0.0f, which leaves the sum as it
// is).  A bin can span megabases (a centromere, an assembly gap: 411 possible positions may lie far apart), so the wave first looks at 64 mask words at once and only
// visits the words that hold a possible position.
__device__ __forceinline__ float weighted_serial(const BinChrom& C, const uint8_t* __restrict__ rg, const float* __restrict__ w, int64_t s, int64_t e) {
    const int l = lane_id();
    float tmp = 0.0f;
    const int64_t wS = s >> 6, wE = (e - 1) >> 6;
    for (int64_t wb = wS; wb <= wE; wb += 64) {
        const int64_t wq = wb + l;
        uint64_t mw = 0;
        if (wq <= wE) {
            mw = C.mask[wq];
            if (wq == wS) mw &= (~0ull) << (s & 63);
            if (wq == wE && (e & 63)) mw &= (~0ull) >> (64 - (e & 63));
        }
        unsigned long long words = __ballot(mw != 0ull);
        while (words) {
            const int wl = __builtin_ctzll(words); words &= words - 1ull;
            const uint64_t m = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mw >> 32), wl) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mw, wl);
            const int64_t p = ((wb + wl) << 6) + l;
            float term = 0.0f;
            if ((m >> l) & 1ull) { const int h = C.hits[p]; if (h) term = fminf(10.0f, (float)h / w[rg[p]]); }
            unsigned long long todo = __ballot(term != 0.0f);
            while (todo) {
                const int src = __builtin_ctzll(todo);
                tmp += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(term), src));
                todo &= todo - 1ull;
            }
        }
    }
    return tmp;
}
// The weighted count.  History (80x genome, 6.2 M bins): one wave per bin adding the terms through readlane 19.5 ms (round 3); exact sum in double, one bin per
// wave 5.6 ms, 16 lanes per bin 5.2 ms, terms from a table + only the positions with a hit visited 6.8 ms — the SQ counters showed 1.9 ms of VALU time per SIMD (a wave
// instruction occupies its SIMD for four cycles) on top of a chain of four dependent loads per bin; then per-word sums from a streaming sweep + the two end words of a bin
// opened again (1.3 terms per position; removed in round 5).  Now k_bin_weighted3 alone: the terms come branch-free from a table in LDS indexed by (hit, readGC) — row 0 is
// 0.0f, so a position without a hit or outside the mask costs the same five instructions as any other; hits above GCW_HMAX (a handful per genome) take a division.
#define GCW_HMAX 20
#define GCW_LONG 64          // whole words between the two end words of a bin beyond which the entire wave sums them
#define GCW_TAB 64           // chromosomes whose pointers k_bin_weighted3 keeps in LDS (the others are read from the tables in memory)
__global__ void __launch_bounds__(256) k_gcw_terms(const float* __restrict__ w, float* __restrict__ lut /* [GCW_HMAX + 1][101] */) {
    for (int i = threadIdx.x; i < (GCW_HMAX + 1) * 101; i += 256) { const int h = i / 101, gc = i - h * 101; lut[i] = h ? fminf(10.0f, (float)h / w[gc]) : 0.0f; }
}
// the terms of 16 consecutive positions (hw: their hits, already zero where the position does not count; gw: their read-GC values) -> exact sum in double, number of non-zero terms
__device__ __forceinline__ void gcw_terms16(const uint32_t (&hw)[4], const uint32_t (&gw)[4], const double* __restrict__ sT, const float* __restrict__ sW, double& sum, uint32_t& n) {      // sT: the term table as doubles (the same values: no conversion per term)
    uint32_t big = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) big |= ((hw[q] & 0x7F7F7F7Fu) + 0x6B6B6B6Bu) | hw[q];       // bit 7 of a byte set <=> the byte is > GCW_HMAX (20 = 0x14: 0x14 + 0x6B = 0x7F)
    big &= 0x80808080u;
    double t[16];
    if (!big) {
        // every hit count is inside the table: the 16 indices hit * 101 + min(readGC, 100) two at a time in packed 16-bit arithmetic (byte pairs spread by v_perm_b32, one packed
        // minimum, one packed multiply-add) instead of seven scalar operations each
        typedef unsigned short gcw_u16x2 __attribute__((ext_vector_type(2)));
        const gcw_u16x2 k101 = {101, 101}, k100 = {100, 100};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t hl = __builtin_amdgcn_perm(0u, hw[q], 0x0c010c00u), hh = __builtin_amdgcn_perm(0u, hw[q], 0x0c030c02u);      // {byte0, byte1} / {byte2, byte3} as 16-bit lanes
            const uint32_t gl = __builtin_amdgcn_perm(0u, gw[q], 0x0c010c00u), gh = __builtin_amdgcn_perm(0u, gw[q], 0x0c030c02u);
            gcw_u16x2 vhl, vhh, vgl, vgh;
            __builtin_memcpy(&vhl, &hl, 4); __builtin_memcpy(&vhh, &hh, 4); __builtin_memcpy(&vgl, &gl, 4); __builtin_memcpy(&vgh, &gh, 4);
            vgl = vgl < k100 ? vgl : k100; vgh = vgh < k100 ? vgh : k100;
            const gcw_u16x2 il = vhl * k101 + vgl, ih = vhh * k101 + vgh;
            t[4 * q] = sT[il.x]; t[4 * q + 1] = sT[il.y]; t[4 * q + 2] = sT[ih.x]; t[4 * q + 3] = sT[ih.y];
        }
    } else {          // (rare) a hit count beyond the table: the division itself for those positions
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t h = (hw[j >> 2] >> (8 * (j & 3))) & 0xFFu, gc0 = (gw[j >> 2] >> (8 * (j & 3))) & 0xFFu;
            const uint32_t gc = gc0 < 101u ? gc0 : 100u;
            t[j] = h <= (uint32_t)GCW_HMAX ? sT[h * 101u + gc] : (double)fminf(10.0f, (float)(int)h / sW[gc]);
        }
    }
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
    for (int j = 0; j < 16; j += 4) { s0 += t[j]; s1 += t[j + 1]; s2 += t[j + 2]; s3 += t[j + 3]; }
    sum = (s0 + s1) + (s2 + s3);
    uint32_t nz = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) { const uint32_t x = hw[q]; nz += (uint32_t)__popc((((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u); }
    n = nz;
}
struct GcwChrom { const uint8_t* readGc; };
// one launch over every chromosome's tiles (BinChrom::tileBase numbers them)
__device__ __forceinline__ int gcw_find_chrom(const BinChrom* __restrict__ ch, int nchr, int64_t tile) {
    int lo = 0, hi = nchr - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (ch[mid].tileBase <= tile) lo = mid; else hi = mid - 1; }
    return lo;
}
// 16 lanes per bin: lanes 0-3 open the word the bin starts in, lanes 4-7 the word it ends in (when that is another one), lanes 8-15 add the sums of the words in between
// The bins partition the positions, so the terms of every position are computed ONCE, by the bin that owns it — no per-word sums in memory.  16 lanes walk their bin 256 positions
// at a time; a bin that spans more than GCW4_SPAN positions (centromeres, assembly gaps) is scanned by the whole wave, mask words first.
#define GCW4_SPAN 4096
__global__ void __launch_bounds__(256) k_bin_weighted3(const BinChrom* __restrict__ ch, const GcwChrom* __restrict__ gch, long long nbins, const int32_t* __restrict__ oChr,
                                                       const int32_t* __restrict__ oStart, const int32_t* __restrict__ oStop, const float* __restrict__ w, const float* __restrict__ lut,
                                                       float* __restrict__ oCount, unsigned long long* __restrict__ replayed /* [GCW_REP] replicas: bins that replayed the reference's additions */,
                                                       int serialOnly, int nchr) {
    __shared__ float sW[101];
    __shared__ double sT[(GCW_HMAX + 1) * 101];
    // A round was a chain of three dependent trips to memory — the bin's (chromosome, start, stop), that chromosome's pointers, the data under them — and a wave makes some
    // two hundred rounds: 74 % of its cycles waiting (SQ counters), 1.57 ms for 6.2 M bins.  The pointer tables now sit in LDS and the next round's bin is requested while this one works.
    struct Tab { const uint64_t* mask; const uint8_t* hits; const uint8_t* rg; int64_t len; };
    __shared__ Tab sTab[GCW_TAB];
    if (threadIdx.x < 101) sW[threadIdx.x] = w[threadIdx.x];
    for (int i = threadIdx.x; i < (GCW_HMAX + 1) * 101; i += 256) sT[i] = (double)lut[i];
    for (int i = threadIdx.x; i < nchr && i < GCW_TAB; i += 256) sTab[i] = Tab{ch[i].mask, ch[i].hits, gch[i].readGc, ch[i].len};
    __syncthreads();
    auto tab = [&](int c) -> Tab { return c < GCW_TAB ? sTab[c] : Tab{ch[c].mask, ch[c].hits, gch[c].readGc, ch[c].len}; };
    const int l = lane_id(), grp = l >> 4, sub = l & 15;
    const long long waveId = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nwaves = (long long)gridDim.x * 4;
    int cN = 0; int64_t sN = 0, eN = 0;
    { const long long i = waveId * 4 + grp; if (i < nbins) { cN = oChr[i]; sN = oStart[i]; eN = oStop[i]; } }
    for (long long i0 = waveId * 4; i0 < nbins; i0 += nwaves * 4) {      // four bins per wave and round
        const long long i = i0 + grp;
        const bool live = i < nbins;
        const int c = cN;
        const int64_t s = sN, e = eN;
        { const long long in = i + nwaves * 4; cN = 0; sN = 0; eN = 0; if (in < nbins) { cN = oChr[in]; sN = oStart[in]; eN = oStop[in]; } }
        double sum = 0.0; uint32_t nterms = 0;
        {
            if (!serialOnly && live && e > s && e - s <= GCW4_SPAN) {
                const Tab T = tab(c);
                const gptr<const uint64_t> mask = as_global(T.mask); const gptr<const uint8_t> hits = as_global(T.hits); const gptr<const uint8_t> rg = as_global(T.rg);
                // 16-aligned slices (the 16 mask bits of a slice lie inside one word), two slices per lane and step with all their loads in flight together: a bin of the
                // usual size (a few hundred positions) is one step
                auto load_slice = [&](int64_t p, uint32_t& m16, uint32_t (&hw)[4], uint32_t (&gw)[4]) {
                    m16 = 0; hw[0] = hw[1] = hw[2] = hw[3] = 0; gw[0] = gw[1] = gw[2] = gw[3] = 0;
                    const int64_t lo = s > p ? s : p, hi = e < p + 16 ? e : p + 16;
                    if (lo >= hi) return;
                    m16 = (uint32_t)((mask[p >> 6] >> (p & 63)) & 0xFFFFull) & (0xFFFFu << (lo - p)) & (0xFFFFu >> (p + 16 - hi));
                    if (p + 16 <= T.len) {
                        const uint4 h = gload_uint4(hits + p), gq = gload_uint4(rg + p);
                        hw[0] = h.x; hw[1] = h.y; hw[2] = h.z; hw[3] = h.w; gw[0] = gq.x; gw[1] = gq.y; gw[2] = gq.z; gw[3] = gq.w;
                    } else {
                        for (int j = 0; j < 16 && p + j < T.len; j++) { hw[j >> 2] |= (uint32_t)hits[p + j] << (8 * (j & 3)); gw[j >> 2] |= (uint32_t)rg[p + j] << (8 * (j & 3)); }
                    }
                };
                for (int64_t p = (s & ~15ll) + 16 * sub; p < e; p += 512) {
                    uint32_t mA, mB, hA[4], gA[4], hB[4], gB[4];
                    load_slice(p, mA, hA, gA);
                    load_slice(p + 256, mB, hB, gB);
#pragma unroll
                    for (int q = 0; q < 4; q++) { hA[q] &= expand4(mA >> (4 * q)); hB[q] &= expand4(mB >> (4 * q)); }
                    double ps; uint32_t pn;
                    if (mA) { gcw_terms16(hA, gA, sT, sW, ps, pn); sum += ps; nterms += pn; }
                    if (mB) { gcw_terms16(hB, gB, sT, sW, ps, pn); sum += ps; nterms += pn; }
                }
            }
            // the long bins: the whole wave, 64 mask words per step, only words that hold a possible position are opened
            unsigned long long longBins = __ballot(!serialOnly && live && e > s && sub == 0 && e - s > GCW4_SPAN);
            while (longBins) {
                const int src = __builtin_ctzll(longBins); longBins &= longBins - 1ull;
                const long long ib = i0 + (src >> 4);
                const int cb = oChr[ib];
                const int64_t sB = oStart[ib], eB = oStop[ib];
                const Tab T = tab(cb);
                const gptr<const uint64_t> mask = as_global(T.mask); const gptr<const uint8_t> hits = as_global(T.hits); const gptr<const uint8_t> rg = as_global(T.rg);
                const int64_t wS = sB >> 6, wE = (eB - 1) >> 6;
                double ls = 0.0; uint32_t ln = 0;
                for (int64_t wb = wS; wb <= wE; wb += 64) {
                    const int64_t wq = wb + l;
                    uint64_t mw = 0;
                    if (wq <= wE) {
                        mw = mask[wq];
                        if (wq == wS) mw &= (~0ull) << (sB & 63);
                        if (wq == wE && (eB & 63)) mw &= (~0ull) >> (64 - (eB & 63));
                    }
                    if (mw) {
                        for (int q4 = 0; q4 < 4; q4++) {
                            const uint32_t m16 = (uint32_t)(mw >> (16 * q4)) & 0xFFFFu;
                            if (!m16) continue;
                            const int64_t p = (wq << 6) + 16 * q4;
                            uint32_t hw[4] = {0, 0, 0, 0}, gw[4] = {0, 0, 0, 0};
                            if (p + 16 <= T.len) {
                                const uint4 h = gload_uint4(hits + p), gq = gload_uint4(rg + p);
                                hw[0] = h.x; hw[1] = h.y; hw[2] = h.z; hw[3] = h.w; gw[0] = gq.x; gw[1] = gq.y; gw[2] = gq.z; gw[3] = gq.w;
                            } else {
                                for (int j = 0; j < 16 && p + j < T.len; j++) { hw[j >> 2] |= (uint32_t)hits[p + j] << (8 * (j & 3)); gw[j >> 2] |= (uint32_t)rg[p + j] << (8 * (j & 3)); }
                            }
#pragma unroll
                            for (int q = 0; q < 4; q++) hw[q] &= expand4(m16 >> (4 * q));
                            double ps; uint32_t pn;
                            gcw_terms16(hw, gw, sT, sW, ps, pn);
                            ls += ps; ln += pn;
                        }
                    }
                }
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) { ls += __shfl_xor(ls, d, 64); ln += __shfl_xor(ln, d, 64); }
                if (l == src) { sum += ls; nterms += ln; }
            }
        }
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) { sum += __shfl_xor(sum, d, 64); nterms += __shfl_xor(nterms, d, 64); }      // within the bin's 16 lanes
        // the float32 running sum of the reference differs from the exact sum by at most nterms roundings of half an ulp of a partial sum <= the final sum (+ its own last ulp)
        // (E <= eps (S + E) with eps = nterms 2^-24  =>  E <= eps S / (1 - eps); a few more roundings' worth for the additions in double; 2 % margin)
        const double eps = ((double)nterms + 4.0) * 5.9604644775390625e-8;
        const double bound = eps * sum * 1.02 + 1e-30;
        const double rl = rint(sum - bound), rh = rint(sum + bound);
        // decided when both ends of the interval round (half to even) to the same integer: rounding is monotone, so every value in between does too
        const bool decided = !serialOnly && eps < 0.01 && rl == rh;
        if (live && decided && sub == 0) oCount[i] = (float)(int)rl;
        // the bins that are not decided replay the reference's own order of additions, the whole wave on one bin at a time
        unsigned long long todo = __ballot(live && !decided && sub == 0);
        while (todo) {
            const int src = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            const long long ib = i0 + (src >> 4);
            const int cb = oChr[ib];
            const BinChrom CB = ch[cb];
            const float r = weighted_serial(CB, gch[cb].readGc, sW, (int64_t)oStart[ib], (int64_t)oStop[ib]);
            if (l == 0) { oCount[ib] = (float)(int)rint((double)r); if (replayed) atomicAdd(&replayed[blockIdx.x % GCW_REP], 1ull); }
        }
    }
}
