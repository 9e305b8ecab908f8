// CanvasBin merge step on MI355X: rates (CanvasBin.cs:30-83) and BinCountsForChromosome (CanvasBin.cs:568-661).
//
// Data layout in HBM (per chromosome, caller-owned): bases u8[L] (FASTA chars), hits u8[L] (HitArray.Data),
// mask u64[ceil(L/64)] (BitArray layout: bit i -> word i>>6, bit i&63).
//
// Geometry: a *tile* is 4096 consecutive positions and is owned by one 64-lane wave (4 iterations x 1024 positions,
// 16 positions = one 16-byte load per lane per array).  Tiles never span chromosomes.
//
// Kernels (all integer arithmetic => bit-identical to the reference by construction):
//   k_find_pos0     first position whose base is not 'n' (CanvasBin.cs:582), multi-workgroup early-exit scan
//   k_tile_stats    per tile: popcount(mask) and #(hit>0)                       [1.125 B/base; 0.125 when rates not needed]
//   k_scan_tiles    per chromosome: exclusive scan of tile popcounts -> rank base per tile, #bins, chromosome totals
//   k_bin_pass      THE hot kernel: per tile, SWAR over 16-byte groups, wave prefix scans of (possible, GC, clamped hits);
//                   every lane that holds the binSize-th possible position of a bin writes that bin's stop and the
//                   tile-local prefix sums at the boundary                       [2.125 B/base read, 12 B/bin written]
//   k_scan_totals   per chromosome: exclusive scan of the per-tile totals (uint32 wrap-around arithmetic)
//   k_bin_finalize  per bin: count = prefix(stop_k) - prefix(stop_{k-1}); gc = (int)(100f*GC/len) in float32 (Q4)
// No atomics, no floating reduction: the prefix-difference formulation is deterministic.
#include "common.hpp"
#include <algorithm>
#include <thread>
#include <emmintrin.h>

#define TILE_SHIFT 12
#define TILE 4096

struct BinChrom {            // one per chromosome, in device memory
    const uint8_t* bases;
    const uint8_t* hits;
    const uint64_t* mask;
    int64_t len;
    int64_t tileBase;         // first global tile index
    int64_t ntiles;
};

__device__ __forceinline__ int find_chrom(const BinChrom* __restrict__ ch, int nchr, int64_t gtile) {
    int lo = 0, hi = nchr - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (ch[mid].tileBase <= gtile) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// ---------------------------------------------------------------------------------------------- k_find_pos0
__global__ void __launch_bounds__(256) k_find_pos0(const BinChrom* __restrict__ ch, unsigned long long* __restrict__ pos0, int c0) {
    const int c = blockIdx.y + c0;
    const BinChrom C = ch[c];
    const int64_t CHUNK = 256 * 16;
    __shared__ unsigned long long scur;
    for (int64_t chunk = blockIdx.x; chunk * CHUNK < C.len; chunk += gridDim.x) {
        int64_t start = chunk * CHUNK;
        // ONE read of the running minimum per workgroup: if every thread read it for itself, waves of one workgroup could disagree (another workgroup's atomicMin
        // lands in between) and leave the loop at different iterations — the ones that stay would then take the minimum over LDS slots nobody wrote
        if (threadIdx.x == 0) scur = __hip_atomic_load(&pos0[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned long long cur = scur;
        if ((unsigned long long)start >= cur) break;      // someone already found an earlier non-'n' (uniform over the workgroup)
        int64_t p = start + (int64_t)threadIdx.x * 16;
        int found = 16;
        if (p + 16 <= C.len) {
            uint4 v = *reinterpret_cast<const uint4*>(C.bases + p);
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 3; q >= 0; q--)
#pragma unroll
                for (int b = 3; b >= 0; b--) if (((w[q] >> (8 * b)) & 0xFF) != 'n') found = q * 4 + b;
        } else {
            for (int i = 15; i >= 0; i--) if (p + i < C.len && C.bases[p + i] != 'n') found = i;
        }
        unsigned long long cand = found < 16 ? (unsigned long long)(p + found) : ~0ull;
        // workgroup min
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { unsigned long long o = __shfl_xor(cand, d, 64); cand = o < cand ? o : cand; }
        __shared__ unsigned long long smin[4];
        if (lane_id() == 0) smin[threadIdx.x >> 6] = cand;
        __syncthreads();
        unsigned long long m = smin[0];
        for (int i = 1; i < 4; i++) m = smin[i] < m ? smin[i] : m;
        __syncthreads();
        if (m != ~0ull) { if (threadIdx.x == 0) atomicMin(&pos0[c], m); break; }
    }
}

// ---------------------------------------------------------------------------------------------- k_tile_stats
__device__ __forceinline__ uint32_t nonzero_bytes4(uint32_t w) {
    uint32_t z = ~(((w & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | w) & 0x80808080u;  // 0x80 where byte == 0
    return 4u - __popc(z);
}

__global__ void __launch_bounds__(256) k_tile_stats(const BinChrom* __restrict__ ch, int nchr, int64_t ntilesTotal, int wantObs,
                                                    uint32_t* __restrict__ tilePop, uint32_t* __restrict__ tileObs) {
    const int64_t gtile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gtile >= ntilesTotal) return;
    const int c = find_chrom(ch, nchr, gtile);
    const BinChrom C = ch[c];
    const int64_t tileStart = (gtile - C.tileBase) << TILE_SHIFT;
    const int l = lane_id();
    // mask: one 64-bit word per lane
    int64_t wstart = tileStart + (int64_t)l * 64;
    uint32_t pop = 0;
    if (wstart < C.len) {
        uint64_t w = C.mask[wstart >> 6];
        int64_t valid = C.len - wstart;
        if (valid < 64) w &= (~0ull) >> (64 - valid);
        pop = __popcll(w);
    }
    pop = wave_reduce_add_u32(pop);
    uint32_t obs = 0;
    if (wantObs) {
        if (tileStart + TILE <= C.len) {
#pragma unroll
            for (int it = 0; it < 4; it++) {
                uint4 v = gload_uint4(as_global(C.hits) + tileStart + it * 1024 + l * 16);
                obs += nonzero_bytes4(v.x) + nonzero_bytes4(v.y) + nonzero_bytes4(v.z) + nonzero_bytes4(v.w);
            }
        } else {
            for (int64_t p = tileStart + l; p < C.len; p += 64) obs += C.hits[p] > 0;
        }
        obs = wave_reduce_add_u32(obs);
    }
    if (l == 0) { tilePop[gtile] = pop; if (wantObs) tileObs[gtile] = obs; }
}

// ---------------------------------------------------------------------------------------------- k_scan_tiles
// one workgroup (1024 threads) per chromosome
struct ChromOut { long long pop, obs, nbins, popBefore; };

// exclusive scan of one or two uint32 values per thread over a 1024-thread workgroup (uint32 wrap-around); ONE barrier per call:
// the caller alternates `sh` between two buffers so that the next call cannot overwrite totals that are still being read
struct U2 { uint32_t a, b; };
__device__ __forceinline__ U2 block_exclusive_scan2_1024(U2 v, U2* sh /*16*/, U2& total) {
    const uint32_t ia = wave_inclusive_scan_u32(v.a), ib = wave_inclusive_scan_u32(v.b);
    const int w = threadIdx.x >> 6, l = lane_id();
    if (l == 63) { sh[w].a = ia; sh[w].b = ib; }
    __syncthreads();
    uint32_t ba = 0, bb = 0, ta = 0, tb = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) { const U2 t = sh[i]; if (i < w) { ba += t.a; bb += t.b; } ta += t.a; tb += t.b; }
    total.a = ta; total.b = tb;
    U2 r; r.a = ia - v.a + ba; r.b = ib - v.b + bb;
    return r;
}
#define ST_ITEMS 8
// Eight consecutive uint32 per thread as two 16-byte vectors.  `arr` is 16-byte aligned and `v0` (the virtual index of the first
// element, a multiple of 8) addresses arr + v0; valid elements are lo <= index < hi, the others read as 0 / are not written
// (they belong to the neighbouring chromosomes, which other workgroups own).  A lane-contiguous 32 B per thread keeps the
// accesses coalesced: with scalar accesses at a 32 B lane stride a single CU becomes request-rate bound.
__device__ __forceinline__ void load8_u32(const uint32_t* __restrict__ arr, int64_t v0, int64_t lo, int64_t hi, uint32_t (&v)[ST_ITEMS]) {
    if (v0 >= lo && v0 + ST_ITEMS <= hi) {
        const uint4 a = *reinterpret_cast<const uint4*>(arr + v0), b = *reinterpret_cast<const uint4*>(arr + v0 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int i = 0; i < ST_ITEMS; i++) v[i] = (v0 + i >= lo && v0 + i < hi) ? arr[v0 + i] : 0u;
    }
}
__device__ __forceinline__ void store8_u32(uint32_t* __restrict__ arr, int64_t v0, int64_t lo, int64_t hi, const uint32_t (&v)[ST_ITEMS]) {
    if (v0 >= lo && v0 + ST_ITEMS <= hi) {
        *reinterpret_cast<uint4*>(arr + v0) = make_uint4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<uint4*>(arr + v0 + 4) = make_uint4(v[4], v[5], v[6], v[7]);
    } else {
#pragma unroll
        for (int i = 0; i < ST_ITEMS; i++) if (v0 + i >= lo && v0 + i < hi) arr[v0 + i] = v[i];
    }
}
__global__ void __launch_bounds__(1024) k_scan_tiles(const BinChrom* __restrict__ ch, const unsigned long long* __restrict__ pos0,
                                                     const uint32_t* __restrict__ tilePop, const uint32_t* __restrict__ tileObs, int wantObs,
                                                     int binSize, int32_t* __restrict__ rankBase, ChromOut* __restrict__ out, int packed = 0) {
    __shared__ U2 sh[2][16];
    __shared__ unsigned long long sRed[2][16];
    const int c = blockIdx.x, tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const BinChrom C = ch[c];
    const int64_t p0 = (int64_t)pos0[c] < C.len ? (int64_t)pos0[c] : C.len;
    const int64_t t0 = p0 >> TILE_SHIFT;   // tile containing pos0
    const uint32_t* __restrict__ pop = tilePop + C.tileBase;
    // possible positions before pos0: full tiles before t0, plus the part of tile t0 in front of pos0 (64 mask words, wave 0)
    unsigned long long before = 0;
    for (int64_t t = tid; t < t0 && t < C.ntiles; t += 1024) before += pop[t];
    if (w == 0) {
        int64_t wstart = (t0 << TILE_SHIFT) + (int64_t)l * 64;
        if (wstart < p0) {
            uint64_t mw = packed ? reinterpret_cast<const ulonglong2*>(C.bases)[wstart >> 6].x : C.mask[wstart >> 6];    // bin_packed.hpp: {possible, gc} pairs
            int64_t valid = p0 - wstart;
            if (valid < 64) mw &= (~0ull) >> (64 - valid);
            before += (unsigned long long)__popcll(mw);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) before += __shfl_xor(before, d);
    if (l == 0) sRed[0][w] = before;
    __syncthreads();
    unsigned long long popBefore = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) popBefore += sRed[0][i];
    // exclusive scan of the tile popcounts, ST_ITEMS consecutive tiles per thread (absolute tile indices, 8-aligned chunks)
    long long carry = 0; unsigned long long obsAcc = 0;
    int buf = 0;
    const int64_t lo = C.tileBase, hi = C.tileBase + C.ntiles;
    for (int64_t base = lo & ~(int64_t)7; base < hi; base += 1024 * ST_ITEMS, buf ^= 1) {
        const int64_t v0 = base + (int64_t)tid * ST_ITEMS;
        uint32_t v[ST_ITEMS]; uint32_t sum = 0;
        load8_u32(tilePop, v0, lo, hi, v);
#pragma unroll
        for (int i = 0; i < ST_ITEMS; i++) sum += v[i];
        if (wantObs) {
            uint32_t o[ST_ITEMS];
            load8_u32(tileObs, v0, lo, hi, o);
#pragma unroll
            for (int i = 0; i < ST_ITEMS; i++) obsAcc += o[i];
        }
        U2 in; in.a = sum; in.b = 0; U2 tot;
        const U2 ex = block_exclusive_scan2_1024(in, sh[buf], tot);
        long long run = carry + (long long)ex.a - (long long)popBefore;
        uint32_t r[ST_ITEMS];
#pragma unroll
        for (int i = 0; i < ST_ITEMS; i++) { r[i] = (uint32_t)(int32_t)run; run += v[i]; }
        store8_u32(reinterpret_cast<uint32_t*>(rankBase), v0, lo, hi, r);
        carry += tot.a;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) obsAcc += __shfl_xor(obsAcc, d);
    if (l == 0) sRed[1][w] = obsAcc;
    __syncthreads();
    if (tid == 0) {
        unsigned long long obs = 0;
        for (int i = 0; i < 16; i++) obs += sRed[1][i];
        ChromOut o; o.pop = carry; o.obs = (long long)obs; o.popBefore = (long long)popBefore;
        o.nbins = binSize > 0 ? (carry - (long long)popBefore) / binSize : 0;
        out[c] = o;
    }
}

// ---------------------------------------------------------------------------------------------- k_bin_pass
__device__ __forceinline__ uint32_t expand4(uint32_t mb) {           // 4 mask bits -> 4 byte masks (0x00 / 0xFF)
    uint32_t x = __umul24(mb & 0xFu, 0x204081u) & 0x01010101u;
    return (x << 8) - x;
}
__device__ __forceinline__ uint32_t clamp10_bytes(uint32_t w) {      // per byte min(10, b)
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    union { uint32_t u; us2 v; } lo, hi, ten;
    ten.u = 0x000A000Au;
    lo.u = w & 0x00FF00FFu; hi.u = (w >> 8) & 0x00FF00FFu;
    lo.v = __builtin_elementwise_min(lo.v, ten.v);
    hi.v = __builtin_elementwise_min(hi.v, ten.v);
    return lo.u | (hi.u << 8);
}
__device__ __forceinline__ uint32_t gc_bits4(uint32_t w) {           // 4 bits: base is C/c/G/g
    uint32_t y = ((w | 0x20202020u) ^ 0x63636363u) & 0xFBFBFBFBu;     // 0 for 'c'(0x63) and 'g'(0x67)
    uint32_t z = ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y) & 0x80808080u;
    return (((z >> 7) * 0x01020408u) >> 24) & 0xFu;
}
__device__ __forceinline__ uint32_t sum_bytes(uint32_t w, uint32_t acc) { return __builtin_amdgcn_sad_u8(w, 0u, acc); }
__device__ __forceinline__ uint32_t gc_marks4(uint32_t w) {          // 0x80 in every byte that is C/c/G/g (popcount = GC count of the 4 bases)
    uint32_t y = ((w | 0x20202020u) ^ 0x63636363u) & 0xFBFBFBFBu;
    return ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y) & 0x80808080u;
}
__device__ __forceinline__ uint32_t marks_to_bits4(uint32_t z) { return (((z >> 7) * 0x01020408u) >> 24) & 0xFu; }
__device__ __forceinline__ uint32_t any_byte_gt10(uint32_t w) { return (((w & 0x7F7F7F7Fu) + 0x75757575u) | w) & 0x80808080u; }

// One tile (4096 positions) per wave.  FULL = the tile lies entirely inside [pos0, len): vector loads, no bounds logic.
template <bool FULL>
__device__ __forceinline__ void bin_tile(const BinChrom& C, int64_t gtile, int64_t tileStart, int64_t p0c, int32_t rank0, long long boff, int binSize, int clampHits,
                                         int32_t* __restrict__ stopOut, uint32_t* __restrict__ locC, uint32_t* __restrict__ locG,
                                         uint32_t* __restrict__ tileTotC, uint32_t* __restrict__ tileTotG) {
    const int l = lane_id();
    uint32_t carryC = 0, carryG = 0;
    // issue all loads of the tile up front (memory-level parallelism): 4 x (16 B bases, 16 B hits, 2 B mask)
    uint4 vb[4], vh[4];
    uint32_t m16[4];
    if (FULL) {
#pragma unroll
        for (int it = 0; it < 4; it++) {
            int64_t p = tileStart + it * 1024 + l * 16;
            vb[it] = gload_uint4(as_global(C.bases) + p);
            vh[it] = gload_uint4(as_global(C.hits) + p);
            m16[it] = reinterpret_cast<gptr<const uint16_t>>(as_global(C.mask))[p >> 4];
        }
    }
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const int64_t p = tileStart + it * 1024 + l * 16;
        uint32_t wb[4], wh[4], mRank, vmask = 0xFFFFu;
        if (FULL) {
            wb[0] = vb[it].x; wb[1] = vb[it].y; wb[2] = vb[it].z; wb[3] = vb[it].w;
            wh[0] = vh[it].x; wh[1] = vh[it].y; wh[2] = vh[it].z; wh[3] = vh[it].w;
            mRank = m16[it];
        } else {   // chromosome head (pos0) or tail tile: bounds-checked byte loads
            mRank = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) { wb[q] = 0; wh[q] = 0; }
            vmask = 0;
            for (int i = 0; i < 16; i++) {
                int64_t pi = p + i;
                if (pi < C.len) {
                    uint32_t mbit = (uint32_t)((C.mask[pi >> 6] >> (pi & 63)) & 1ull);
                    mRank |= mbit << i;
                    wb[i >> 2] |= (uint32_t)C.bases[pi] << (8 * (i & 3));
                    wh[i >> 2] |= (uint32_t)C.hits[pi] << (8 * (i & 3));
                    if (pi >= p0c) vmask |= 1u << i;
                }
            }
        }
        const uint32_t mVal = mRank & vmask;
        // per-lane values.  GC: only the COUNT is needed outside the (rare) boundary lanes, so the per-position bits are built lazily.
        uint32_t gz[4] = {gc_marks4(wb[0]), gc_marks4(wb[1]), gc_marks4(wb[2]), gc_marks4(wb[3])};
        uint32_t g, gcb = 0;
        if (FULL) g = __popc(gz[0]) + __popc(gz[1]) + __popc(gz[2]) + __popc(gz[3]);
        else { gcb = (marks_to_bits4(gz[0]) | (marks_to_bits4(gz[1]) << 4) | (marks_to_bits4(gz[2]) << 8) | (marks_to_bits4(gz[3]) << 12)) & vmask; g = __popc(gcb); }
        // hits: min(10, h) only matters when some byte exceeds 10 (pile-ups): wave-uniform fast path without the clamp
        const bool needClamp = clampHits && __any((int)((any_byte_gt10(wh[0]) | any_byte_gt10(wh[1]) | any_byte_gt10(wh[2]) | any_byte_gt10(wh[3])) != 0u));
        uint32_t cw[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t h = needClamp ? clamp10_bytes(wh[q]) : wh[q];
            cw[q] = h & expand4(mVal >> (4 * q));
        }
        uint32_t cl = sum_bytes(cw[0], sum_bytes(cw[1], sum_bytes(cw[2], sum_bytes(cw[3], 0u))));
        uint32_t pop = __popc(mRank);
        // wave scans: (pop | g<<16) packed, c separate
        uint32_t pg = pop | (g << 16);
        uint32_t pgInc = wave_inclusive_scan_u32(pg);
        uint32_t cInc = wave_inclusive_scan_u32(cl);
        uint32_t pgEx = pgInc - pg, cEx = cInc - cl;
        uint32_t popEx = pgEx & 0xFFFFu, gEx = pgEx >> 16;
        // boundaries inside this lane's 16 positions
        int32_t r = rank0 + (int32_t)popEx;            // rank before this lane
        if (pop > 0 && r + (int32_t)pop >= binSize) {  // cheap reject: a boundary needs rank >= binSize
            int32_t rr = r < 0 ? 0 : r;                // ranks <= 0 can never close a bin
            uint32_t nextB = ((uint32_t)rr / (uint32_t)binSize + 1u) * (uint32_t)binSize;   // next boundary rank > rr
            if (FULL) gcb = marks_to_bits4(gz[0]) | (marks_to_bits4(gz[1]) << 4) | (marks_to_bits4(gz[2]) << 8) | (marks_to_bits4(gz[3]) << 12);
            while ((int64_t)nextB - r <= (int64_t)pop && (int64_t)nextB - r >= 1) {
                uint32_t k = (uint32_t)((int64_t)nextB - r);       // k-th set bit of mRank closes the bin
                uint32_t pos = 0, m = mRank, kk = k, cnt;
                cnt = __popc(m & 0xFFu); if (kk > cnt) { kk -= cnt; pos += 8; m >>= 8; }
                cnt = __popc(m & 0xFu);  if (kk > cnt) { kk -= cnt; pos += 4; m >>= 4; }
                cnt = __popc(m & 0x3u);  if (kk > cnt) { kk -= cnt; pos += 2; m >>= 2; }
                cnt = m & 1u;            if (kk > cnt) { pos += 1; }
                // head sums: positions <= pos
                uint32_t hc = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    int rel = (int)pos - 4 * q;
                    uint32_t bm = rel >= 3 ? 0xFFFFFFFFu : (rel < 0 ? 0u : (0xFFFFFFFFu >> (8 * (3 - rel))));
                    hc = sum_bytes(cw[q] & bm, hc);
                }
                uint32_t hg = __popc(gcb & ((2u << pos) - 1u));
                long long bin = boff + (long long)(nextB / (uint32_t)binSize) - 1;
                stopOut[bin] = (int32_t)(p + pos + 1);
                locC[bin] = carryC + cEx + hc;
                locG[bin] = carryG + gEx + hg;
                nextB += (uint32_t)binSize;
            }
        }
        // carry to next iteration (wave totals from lane 63)
        uint32_t pgTot = __shfl(pgInc, 63, 64), cTot = __shfl(cInc, 63, 64);
        rank0 += (int32_t)(pgTot & 0xFFFFu);
        carryG += pgTot >> 16;
        carryC += cTot;
    }
    if (l == 0) { tileTotC[gtile] = carryC; tileTotG[gtile] = carryG; }
}

// the hot kernel: all tiles that lie completely inside [pos0, len) of their chromosome
__global__ void __launch_bounds__(256) k_bin_pass(const BinChrom* __restrict__ ch, int nchr, int64_t ntilesTotal,
                                                  const unsigned long long* __restrict__ pos0, const int32_t* __restrict__ rankBase,
                                                  const long long* __restrict__ binOffset, int binSize, int clampHits,
                                                  int32_t* __restrict__ stopOut, uint32_t* __restrict__ locC, uint32_t* __restrict__ locG,
                                                  uint32_t* __restrict__ tileTotC, uint32_t* __restrict__ tileTotG) {
    const int64_t gtile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gtile >= ntilesTotal) return;
    const int c = find_chrom(ch, nchr, gtile);
    const BinChrom C = ch[c];
    const int64_t tileStart = (gtile - C.tileBase) << TILE_SHIFT;
    const int64_t p0c = (int64_t)pos0[c];
    if (tileStart + TILE > C.len || tileStart < p0c) {
        // edge tiles (the one holding pos0, the partial tail) belong to k_bin_pass_edges; tiles entirely before pos0 hold no bin data
        if (tileStart + TILE <= p0c && lane_id() == 0) { tileTotC[gtile] = 0; tileTotG[gtile] = 0; }
        return;
    }
    bin_tile<true>(C, gtile, tileStart, p0c, rankBase[gtile], binOffset[c], binSize, clampHits, stopOut, locC, locG, tileTotC, tileTotG);
}
// at most two edge tiles per chromosome: wave 2c -> the tile holding pos0, wave 2c+1 -> the partial tail tile
__global__ void __launch_bounds__(64) k_bin_pass_edges(const BinChrom* __restrict__ ch, int nchr, const unsigned long long* __restrict__ pos0,
                                                       const int32_t* __restrict__ rankBase, const long long* __restrict__ binOffset, int binSize, int clampHits,
                                                       int32_t* __restrict__ stopOut, uint32_t* __restrict__ locC, uint32_t* __restrict__ locG,
                                                       uint32_t* __restrict__ tileTotC, uint32_t* __restrict__ tileTotG) {
    const int c = blockIdx.x >> 1, which = blockIdx.x & 1;
    if (c >= nchr) return;
    const BinChrom C = ch[c];
    const int64_t p0c = (int64_t)pos0[c];
    const int64_t headTile = (p0c < C.len ? p0c : C.len - 1) >> TILE_SHIFT, tailTile = C.ntiles - 1;
    int64_t t;
    if (which == 0) t = headTile;
    else { if (tailTile == headTile) return; t = tailTile; }
    const int64_t tileStart = t << TILE_SHIFT;
    const bool isFull = (tileStart + TILE <= C.len) && (tileStart >= p0c);
    if (isFull) return;                                   // handled by k_bin_pass
    if (tileStart + TILE <= p0c) return;                  // entirely before pos0 (all-'n' chromosome): zero totals written by k_bin_pass
    const int64_t gtile = C.tileBase + t;
    bin_tile<false>(C, gtile, tileStart, p0c, rankBase[gtile], binOffset[c], binSize, clampHits, stopOut, locC, locG, tileTotC, tileTotG);
}

// ---------------------------------------------------------------------------------------------- single-read path (bin size not known yet)
// When the bin size has to be derived from the sample's own hit rates (CanvasBin.cs:30-83) the per-base arrays would be streamed twice:
// once for the rates, once for the binning.  Instead the first pass reads bases, hits and mask ONCE and leaves, next to the rate
// inputs, a 4-byte summary of every 64 positions (one mask word): possible count (7 bits), G/C count (7 bits) and the summed
// (masked, mode-clamped) hits (14 bits).  None of them depends on the bin size.  After the host has fixed the bin size, k_bin_close
// walks the summaries (1/34 of the per-base bytes) and opens the per-base arrays only for the 64 positions that hold a bin boundary.
#define SUM_POP(s) ((s) & 127u)
#define SUM_GC(s) (((s) >> 7) & 127u)
#define SUM_HITS(s) ((s) >> 14)

// One tile per wave.  FULL: the tile lies inside [0, len) and not across pos0; `vm` = 0xFFFF (tile at/after pos0) or 0 (tile before pos0:
// positions in front of the first non-'n' base count as possible/observed for the rates but carry no bin data).
template <bool FULL>
__device__ __forceinline__ void summarize_tile(const BinChrom& C, int64_t gtile, int64_t tileStart, int64_t p0c, uint32_t vm, int clampHits, int wantObs,
                                               uint32_t* __restrict__ S, uint32_t* __restrict__ tilePop, uint32_t* __restrict__ tileObs,
                                               uint32_t* __restrict__ tileTotC, uint32_t* __restrict__ tileTotG) {
    const int l = lane_id();
    uint4 vb[4], vh[4];
    uint32_t m16[4];
    if (FULL) {
#pragma unroll
        for (int it = 0; it < 4; it++) {
            int64_t p = tileStart + it * 1024 + l * 16;
            vb[it] = gload_uint4(as_global(C.bases) + p);           // global_load, not flat_load (common.hpp)
            vh[it] = gload_uint4(as_global(C.hits) + p);
            m16[it] = reinterpret_cast<gptr<const uint16_t>>(as_global(C.mask))[p >> 4];
        }
    }
    uint32_t keep = 0, obs = 0;
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const int64_t p = tileStart + it * 1024 + l * 16;
        uint32_t wb[4], wh[4], mRank, vmask = vm;
        if (FULL) {
            wb[0] = vb[it].x; wb[1] = vb[it].y; wb[2] = vb[it].z; wb[3] = vb[it].w;
            wh[0] = vh[it].x; wh[1] = vh[it].y; wh[2] = vh[it].z; wh[3] = vh[it].w;
            mRank = m16[it];
        } else {
            mRank = 0; vmask = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) { wb[q] = 0; wh[q] = 0; }
            for (int i = 0; i < 16; i++) {
                int64_t pi = p + i;
                if (pi < C.len) {
                    uint32_t mbit = (uint32_t)((C.mask[pi >> 6] >> (pi & 63)) & 1ull);
                    mRank |= mbit << i;
                    wb[i >> 2] |= (uint32_t)C.bases[pi] << (8 * (i & 3));
                    wh[i >> 2] |= (uint32_t)C.hits[pi] << (8 * (i & 3));
                    if (pi >= p0c) vmask |= 1u << i;
                }
            }
        }
        const uint32_t mVal = mRank & vmask;
        uint32_t g;
        if (FULL) g = vmask ? __popc(gc_marks4(wb[0])) + __popc(gc_marks4(wb[1])) + __popc(gc_marks4(wb[2])) + __popc(gc_marks4(wb[3])) : 0u;
        else g = __popc((marks_to_bits4(gc_marks4(wb[0])) | (marks_to_bits4(gc_marks4(wb[1])) << 4) | (marks_to_bits4(gc_marks4(wb[2])) << 8) | (marks_to_bits4(gc_marks4(wb[3])) << 12)) & vmask);
        const bool needClamp = clampHits && __any((int)((any_byte_gt10(wh[0]) | any_byte_gt10(wh[1]) | any_byte_gt10(wh[2]) | any_byte_gt10(wh[3])) != 0u));
        uint32_t cl = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t h = needClamp ? clamp10_bytes(wh[q]) : wh[q];
            cl = sum_bytes(h & expand4(mVal >> (4 * q)), cl);
        }
        if (wantObs) obs += nonzero_bytes4(wh[0]) + nonzero_bytes4(wh[1]) + nonzero_bytes4(wh[2]) + nonzero_bytes4(wh[3]);
        // the four lanes of a quad hold one mask word (64 positions): the packed fields cannot carry into each other (64, 64, 16320)
        uint32_t q4 = (uint32_t)__popc(mRank) | (g << 7) | (cl << 14);
        q4 += __shfl_xor(q4, 1, 64);
        q4 += __shfl_xor(q4, 2, 64);
        if ((l & 3) == it) keep = q4;
    }
    // lane 4j+i keeps word 16*i + j of the tile
    S[gtile * 64 + (l & 3) * 16 + (l >> 2)] = keep;
    const uint32_t pg = wave_reduce_add_u32(SUM_POP(keep) | (SUM_GC(keep) << 16));
    const uint32_t ct = wave_reduce_add_u32(SUM_HITS(keep));
    if (wantObs) obs = wave_reduce_add_u32(obs);
    if (l == 0) { tilePop[gtile] = pg & 0xFFFFu; tileTotG[gtile] = pg >> 16; tileTotC[gtile] = ct; if (wantObs) tileObs[gtile] = obs; }
}
__global__ void __launch_bounds__(256) k_tile_summary(const BinChrom* __restrict__ ch, int nchr, int64_t ntilesTotal, const unsigned long long* __restrict__ pos0,
                                                      int clampHits, int wantObs, uint32_t* __restrict__ S, uint32_t* __restrict__ tilePop, uint32_t* __restrict__ tileObs,
                                                      uint32_t* __restrict__ tileTotC, uint32_t* __restrict__ tileTotG, int64_t tile0) {
    // the wave index is uniform: telling the compiler so keeps the chromosome lookup on the scalar unit.  tile0 / ntilesTotal = the range of global tiles
    // this launch covers (the whole genome, or one chromosome when the launch follows that chromosome's upload)
    const int64_t gtile = tile0 + (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (gtile >= ntilesTotal) return;
    const int c = find_chrom(ch, nchr, gtile);
    const BinChrom C = ch[c];
    const int64_t tileStart = (gtile - C.tileBase) << TILE_SHIFT;
    const int64_t p0c = (int64_t)pos0[c];
    if (tileStart + TILE > C.len) return;                                   // partial tail: k_tile_summary_edges
    if (tileStart < p0c && tileStart + TILE > p0c) return;                  // the tile pos0 falls into: k_tile_summary_edges
    summarize_tile<true>(C, gtile, tileStart, p0c, tileStart >= p0c ? 0xFFFFu : 0u, clampHits, wantObs, S, tilePop, tileObs, tileTotC, tileTotG);
}
// at most two edge tiles per chromosome: block 2c -> the tile pos0 falls into (if it is not tile-aligned), block 2c+1 -> the partial tail tile
__global__ void __launch_bounds__(64) k_tile_summary_edges(const BinChrom* __restrict__ ch, int nchr, const unsigned long long* __restrict__ pos0,
                                                           int clampHits, int wantObs, uint32_t* __restrict__ S, uint32_t* __restrict__ tilePop, uint32_t* __restrict__ tileObs,
                                                           uint32_t* __restrict__ tileTotC, uint32_t* __restrict__ tileTotG, int c0) {
    const int c = (blockIdx.x >> 1) + c0, which = blockIdx.x & 1;
    if (c >= nchr) return;
    const BinChrom C = ch[c];
    const int64_t p0c = (int64_t)pos0[c];
    const int64_t tailTile = C.ntiles - 1;
    const bool tailPartial = (tailTile << TILE_SHIFT) + TILE > C.len;
    int64_t t;
    if (which == 1) { if (!tailPartial) return; t = tailTile; }
    else {
        if (p0c >= C.len) return;                                           // no non-'n' base: every tile is "before pos0"
        t = p0c >> TILE_SHIFT;
        if ((t << TILE_SHIFT) == p0c) return;                               // tile-aligned pos0: nothing straddles
        if (t == tailTile && tailPartial) return;                           // the tail block takes it
    }
    summarize_tile<false>(C, C.tileBase + t, t << TILE_SHIFT, p0c, 0u, clampHits, wantObs, S, tilePop, tileObs, tileTotC, tileTotG);
}

// Second pass of the single-read path, step 1: one tile per wave, one 64-position summary per lane.  The scans are the ones of bin_tile at
// a coarser grain.  A lane whose word holds a boundary only records WHERE the bin closes (word start | rank inside the word - 1) and
// the sums in front of the word; k_bin_resolve opens the per-base arrays with one thread per bin (all lanes busy, no dependent-load chain
// inside a mostly idle wave).
#define CLOSE_TILES 4       // tiles per wave: their summary loads are issued together (the kernel is latency-bound, not bandwidth-bound)
__global__ void __launch_bounds__(256) k_bin_close(const BinChrom* __restrict__ ch, int nchr, int64_t ntilesTotal, const uint32_t* __restrict__ S,
                                                   const int32_t* __restrict__ rankBase, const long long* __restrict__ binOffset, int binSize, unsigned long long binMagic,
                                                   int32_t* __restrict__ stopOut, uint32_t* __restrict__ locC, uint32_t* __restrict__ locG, int32_t* __restrict__ oChr) {
    const int64_t gtile0 = ((int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) * CLOSE_TILES;   // uniform: scalar lookups
    if (gtile0 >= ntilesTotal) return;
    const int l = lane_id();
    uint32_t sv[CLOSE_TILES]; int32_t rk[CLOSE_TILES];
#pragma unroll
    for (int t = 0; t < CLOSE_TILES; t++) {
        const bool in = gtile0 + t < ntilesTotal;
        sv[t] = in ? S[(gtile0 + t) * 64 + l] : 0u;
        rk[t] = in ? rankBase[gtile0 + t] : 0;
    }
    int c = find_chrom(ch, nchr, gtile0);
    int64_t tileBase = ch[c].tileBase, tileEnd = tileBase + ch[c].ntiles;
    long long boff = binOffset[c];
#pragma unroll
    for (int t = 0; t < CLOSE_TILES; t++) {
        const int64_t gtile = gtile0 + t;
        if (gtile >= ntilesTotal) break;
        while (gtile >= tileEnd) { c++; tileBase = ch[c].tileBase; tileEnd = tileBase + ch[c].ntiles; boff = binOffset[c]; }
        const uint32_t s = sv[t];
        const uint32_t pop = SUM_POP(s);
        const uint32_t pg = pop | (SUM_GC(s) << 16), cl = SUM_HITS(s);
        const uint32_t pgInc = wave_inclusive_scan_u32(pg), cInc = wave_inclusive_scan_u32(cl);
        const int32_t r = rk[t] + (int32_t)((pgInc - pg) & 0xFFFFu);        // rank before this lane's word
        if (pop > 0 && r + (int32_t)pop >= binSize) {
            const uint32_t gEx = (pgInc - pg) >> 16, cEx = cInc - cl;
            const int64_t wstart = ((gtile - tileBase) << TILE_SHIFT) + (int64_t)l * 64;
            const int32_t rr = r < 0 ? 0 : r;
            // rr / binSize without a division (the kernel is VALU-bound and an integer division is ~30 instructions): binMagic = floor(2^64 / binSize) + 1, and
            // mulhi64(x, binMagic) == x / binSize for every x < 2^32 (the error term x * (binMagic * binSize - 2^64) / 2^64 / binSize stays below 1 / binSize)
            uint32_t q = binMagic ? (uint32_t)__umul64hi((unsigned long long)(uint32_t)rr, binMagic) : (uint32_t)rr;      // (binMagic == 0: binSize 1)
            uint32_t nextB = (q + 1u) * (uint32_t)binSize;
            for (; (int64_t)nextB - r <= (int64_t)pop && (int64_t)nextB - r >= 1; q++) {
                const long long bin = boff + (long long)q;
                stopOut[bin] = (int32_t)(wstart + ((int64_t)nextB - r - 1));    // the (nextB - r)-th possible position of the word closes the bin
                locC[bin] = cEx;
                locG[bin] = gEx;
                oChr[bin] = c;
                nextB += (uint32_t)binSize;
            }
        }
    }
}
// step 2: four lanes per bin (one 16-byte slice of the word's bases / hits each, so a bin's loads are in flight together and coalesce to 64 B).
// Turns the record of k_bin_close into the stop position and adds the sums of the word's head.
__global__ void __launch_bounds__(256) k_bin_resolve(const BinChrom* __restrict__ ch, int nchr, const long long* __restrict__ binOffset,
                                                     const unsigned long long* __restrict__ pos0, int clampHits,
                                                     const int32_t* __restrict__ oChr, int32_t* __restrict__ stopIO, uint32_t* __restrict__ locC, uint32_t* __restrict__ locG) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long i = t >> 2;
    const int sub = (int)(t & 3);
    if (i >= binOffset[nchr]) return;                                        // whole quads leave together
    const int32_t rec = stopIO[i];
    const int c = oChr[i];                                                   // left by k_bin_close
    const gptr<const uint8_t> bases = as_global(ch[c].bases), hits = as_global(ch[c].hits);
    const int64_t len = ch[c].len;
    const int64_t p0c = (int64_t)pos0[c];
    const int64_t wstart = (int64_t)(rec & ~63);
    uint32_t kk = (uint32_t)(rec & 63) + 1u;
    const int64_t cstart = wstart + 16 * sub;                                // this lane's slice
    uint64_t mw = as_global(ch[c].mask)[wstart >> 6];
    uint32_t wb[4] = {0, 0, 0, 0}, wh[4] = {0, 0, 0, 0};
    if (cstart + 16 <= len) {
        const uint4 b = gload_uint4(bases + cstart), h = gload_uint4(hits + cstart);
        wb[0] = b.x; wb[1] = b.y; wb[2] = b.z; wb[3] = b.w; wh[0] = h.x; wh[1] = h.y; wh[2] = h.z; wh[3] = h.w;
    } else {
        for (int j = 0; j < 16; j++) { const int64_t pi = cstart + j; if (pi < len) { wb[j >> 2] |= (uint32_t)bases[pi] << (8 * (j & 3)); wh[j >> 2] |= (uint32_t)hits[pi] << (8 * (j & 3)); } }
    }
    uint64_t valid = ~0ull;                                                  // positions that carry bin data: pos0 <= p < len
    if (len - wstart < 64) { valid = (~0ull) >> (64 - (len - wstart)); mw &= valid; }
    if (wstart < p0c) valid = (p0c - wstart >= 64) ? 0ull : (valid & ((~0ull) << (p0c - wstart)));
    uint32_t pos = 0, cnt;                                                   // the kk-th set bit of mw closes the bin
    uint64_t m = mw;
    cnt = __popc((uint32_t)m);           if (kk > cnt) { kk -= cnt; pos += 32; m >>= 32; }
    cnt = __popc((uint32_t)m & 0xFFFFu); if (kk > cnt) { kk -= cnt; pos += 16; m >>= 16; }
    cnt = __popc((uint32_t)m & 0xFFu);   if (kk > cnt) { kk -= cnt; pos += 8; m >>= 8; }
    cnt = __popc((uint32_t)m & 0xFu);    if (kk > cnt) { kk -= cnt; pos += 4; m >>= 4; }
    cnt = __popc((uint32_t)m & 0x3u);    if (kk > cnt) { kk -= cnt; pos += 2; m >>= 2; }
    cnt = (uint32_t)m & 1u;              if (kk > cnt) { pos += 1; }
    const uint64_t head = valid & ((2ull << pos) - 1ull);                    // valid positions <= pos (pos = 63: all of them)
    const uint32_t head16 = (uint32_t)(head >> (16 * sub)) & 0xFFFFu, mVal16 = (uint32_t)(mw >> (16 * sub)) & head16;
    uint32_t gc16 = 0, hc = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        gc16 |= marks_to_bits4(gc_marks4(wb[q])) << (4 * q);
        const uint32_t h = clampHits ? clamp10_bytes(wh[q]) : wh[q];
        hc = sum_bytes(h & expand4(mVal16 >> (4 * q)), hc);
    }
    uint32_t hg = __popc(gc16 & head16);
    hc += __shfl_xor(hc, 1, 64); hg += __shfl_xor(hg, 1, 64);
    hc += __shfl_xor(hc, 2, 64); hg += __shfl_xor(hg, 2, 64);
    if (sub == 0) {
        stopIO[i] = (int32_t)(wstart + pos + 1);
        locC[i] += hc;
        locG[i] += hg;
    }
}

// ---------------------------------------------------------------------------------------------- k_scan_totals
__global__ void __launch_bounds__(1024) k_scan_totals(const BinChrom* __restrict__ ch, uint32_t* __restrict__ tileTotC, uint32_t* __restrict__ tileTotG) {
    __shared__ U2 sh[2][16];
    const BinChrom C = ch[blockIdx.x];
    const int64_t lo = C.tileBase, hi = C.tileBase + C.ntiles;
    uint32_t carryC = 0, carryG = 0;
    int buf = 0;
    for (int64_t base = lo & ~(int64_t)7; base < hi; base += 1024 * ST_ITEMS, buf ^= 1) {
        const int64_t v0 = base + (int64_t)threadIdx.x * ST_ITEMS;
        uint32_t vc[ST_ITEMS], vg[ST_ITEMS]; U2 in; in.a = 0; in.b = 0;
        load8_u32(tileTotC, v0, lo, hi, vc); load8_u32(tileTotG, v0, lo, hi, vg);
#pragma unroll
        for (int i = 0; i < ST_ITEMS; i++) { in.a += vc[i]; in.b += vg[i]; }
        U2 tot;
        const U2 ex = block_exclusive_scan2_1024(in, sh[buf], tot);
        uint32_t rc = carryC + ex.a, rg = carryG + ex.b;
#pragma unroll
        for (int i = 0; i < ST_ITEMS; i++) { const uint32_t c0 = vc[i], g0 = vg[i]; vc[i] = rc; vg[i] = rg; rc += c0; rg += g0; }
        store8_u32(tileTotC, v0, lo, hi, vc); store8_u32(tileTotG, v0, lo, hi, vg);
        carryC += tot.a; carryG += tot.b;
    }
}

// ---------------------------------------------------------------------------------------------- k_bin_finalize
__global__ void __launch_bounds__(256) k_bin_finalize(const BinChrom* __restrict__ ch, int nchr, const long long* __restrict__ binOffset,
                                                      const unsigned long long* __restrict__ pos0, const int32_t* __restrict__ stopIn,
                                                      const uint32_t* __restrict__ locC, const uint32_t* __restrict__ locG,
                                                      const uint32_t* __restrict__ tileExC, const uint32_t* __restrict__ tileExG,
                                                      int32_t* __restrict__ oChr, int32_t* __restrict__ oStart, int32_t* __restrict__ oStop,
                                                      int32_t* __restrict__ oGc, float* __restrict__ oCount, int haveChr) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= binOffset[nchr]) return;
    // the per-bin records of this bin and of the one in front of it do not depend on the chromosome: requested first, all at once
    const long long ip = i > 0 ? i - 1 : 0;
    const int32_t stopHere = stopIn[i], pstop = stopIn[ip];
    const uint32_t lc = locC[i], lg = locG[i], plc = locC[ip], plg = locG[ip];
    int c;
    if (haveChr) c = oChr[i];                             // k_bin_close left the chromosome of every bin: one load instead of a bisection (five dependent ones)
    else {
        int lo = 0, hi = nchr - 1;
        while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (binOffset[mid] <= i) lo = mid; else hi = mid - 1; }
        // skip chromosomes without bins that share the same offset
        while (lo < nchr - 1 && binOffset[lo + 1] <= i) lo++;
        c = lo;
    }
    const BinChrom C = ch[c];
    const long long k = i - binOffset[c];
    const int32_t stop = stopHere;
    const int64_t tile = C.tileBase + ((int64_t)(stop - 1) >> TILE_SHIFT);
    const int64_t ptile = C.tileBase + ((int64_t)((k > 0 ? pstop : 1) - 1) >> TILE_SHIFT);      // (k == 0: any valid tile, the value is not used)
    const uint32_t exC = tileExC[tile], exG = tileExG[tile], pexC = tileExC[ptile], pexG = tileExG[ptile];      // four loads in flight
    uint32_t gC = exC + lc, gG = exG + lg;
    uint32_t pC = 0, pG = 0;
    int32_t start = (int32_t)pos0[c];
    if (k > 0) { pC = pexC + plc; pG = pexG + plg; start = pstop; }
    const uint32_t count = gC - pC, gcCount = gG - pG;
    const int32_t nuc = stop - start;
    float gcf = 100.0f * (float)(int32_t)gcCount;     // (int)(100f * GCCount / NucleotideCount), CanvasBin.cs:638
    gcf = gcf / (float)nuc;
    oChr[i] = c; oStart[i] = start; oStop[i] = stop; oGc[i] = (int32_t)gcf; oCount[i] = (float)(int32_t)count;
}

// exclusive scan of per-chromosome bin counts (tiny)
__global__ void k_bin_offsets(ChromOut* __restrict__ co, int nchr, int binSize, long long* __restrict__ binOffset) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        long long s = 0;
        for (int c = 0; c < nchr; c++) { const long long nb = (co[c].pop - co[c].popBefore) / binSize; co[c].nbins = nb; binOffset[c] = s; s += nb; }
        binOffset[nchr] = s;
    }
}
// The chromosome table arrives BY VALUE as a kernel argument (<= BIN_BYVAL chromosomes: the argument block is copied at launch, no H2D copy and no blit kernel in front of
// the pass): published to the table the other kernels read, together with the is-autosome bytes and the initial pos0 (the chromosome's length, or the packer's pos0).
#define BIN_BYVAL 64
struct BinChromPack { BinChrom c[BIN_BYVAL]; unsigned long long pos0[BIN_BYVAL]; uint8_t isAuto[BIN_BYVAL]; };
__global__ void __launch_bounds__(256) k_bin_begin(BinChrom* __restrict__ ch, uint8_t* __restrict__ isAuto, const BinChromPack pack, int nchr, unsigned long long* __restrict__ pos0, int pos0Given) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&pack.c[0]); uint32_t* dst = reinterpret_cast<uint32_t*>(ch);
    const int words = nchr * (int)(sizeof(BinChrom) / 4);
    for (int i = threadIdx.x; i < words; i += 256) dst[i] = src[i];
    for (int c = threadIdx.x; c < nchr; c += 256) { isAuto[c] = pack.isAuto[c]; pos0[c] = pos0Given ? pack.pos0[c] : (unsigned long long)pack.c[c].len; }
}
__global__ void k_init_pos0(const BinChrom* __restrict__ ch, int nchr, unsigned long long* __restrict__ pos0) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < nchr) pos0[c] = (unsigned long long)ch[c].len;
}


#include "bin_gcw.hpp"

#include "bin_packed.hpp"
#include "bin_tail.hpp"

// ---------------------------------------------------------------------------------------------- host side
// workgroups of `fn` that are resident on the device at the same time (grid of a persistent kernel)
static unsigned resident_grid(const void* fn, int threads, int device) {
    int perCu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, fn, threads, 0) != hipSuccess || perCu <= 0) perCu = 4;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
    return (unsigned)(perCu * cus);
}
struct BinPlan {
    std::vector<BinChrom> chroms;
    int64_t ntiles = 0;
};
static BinPlan make_plan(int nchr, const uint8_t* const* bases, const uint64_t* const* mask, const uint8_t* const* hits, const int64_t* len) {
    BinPlan p;
    p.chroms.resize(nchr);
    for (int c = 0; c < nchr; c++) {
        BinChrom& C = p.chroms[c];
        C.bases = bases ? bases[c] : nullptr; C.hits = hits ? hits[c] : nullptr; C.mask = mask[c]; C.len = len[c];
        C.tileBase = p.ntiles; C.ntiles = (len[c] + TILE - 1) / TILE;
        p.ntiles += C.ntiles;
    }
    return p;
}

int32_t cvx_bin_sample_hooked(canvas_ctx* ctx, int32_t nchr, const uint8_t* const* d_bases, const uint64_t* const* d_mask, const uint8_t* const* d_hits, const int64_t* h_len,
                              int32_t mode, cvx_bin_size_hook hook, void* user, int32_t* d_chr, int32_t* d_start, int32_t* d_stop, int32_t* d_gc, float* d_count, int64_t cap,
                              int64_t* h_nbins_per_chr, int64_t* h_nbins_total, const int64_t* h_pos0_packed, const int16_t* const* d_fraglen);

extern "C" {

int32_t canvas_bin_size_from_rates(const double* h_rates, int32_t n, int32_t counts_per_bin) {
    if (!h_rates || n <= 0) return CANVAS_ERR_INVALID;
    std::vector<double> r(h_rates, h_rates + n);
    std::sort(r.begin(), r.end());
    double med = (n % 2) ? r[n / 2] : (r[n / 2 - 1] + r[n / 2]) / 2;   // SortedList<double>.Median()
    return (int32_t)(counts_per_bin / med);                             // CanvasBin.cs:82
}

int64_t canvas_bin_count_upper_bound(int32_t nchr, const int64_t* h_len, int32_t bin_size) {
    if (bin_size <= 0 || !h_len) return CANVAS_ERR_INVALID;
    int64_t s = 0;
    for (int c = 0; c < nchr; c++) s += h_len[c] / bin_size;
    return s;
}

int32_t canvas_bin_rates(canvas_ctx* ctx, int32_t nchr, const uint8_t* const* d_hits, const uint64_t* const* d_mask,
                         const int64_t* h_len, int64_t* h_observed, int64_t* h_possible, double* h_rate) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (nchr <= 0 || !d_hits || !d_mask || !h_len) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_bin_rates: bad arguments");
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    { int32_t rcf = canvas_upload_fence(ctx); if (rcf) return rcf; }
    BinPlan plan = make_plan(nchr, nullptr, d_mask, d_hits, h_len);
    WsSizer sz; sz.take<BinChrom>(nchr); sz.take<unsigned long long>(nchr); sz.take<uint32_t>(plan.ntiles); sz.take<uint32_t>(plan.ntiles);
    sz.take<int32_t>(plan.ntiles); sz.take<ChromOut>(nchr);
    int32_t rc = canvas_ws_reserve(ctx, sz.off + 4096); if (rc) return rc;
    rc = canvas_pin_reserve(ctx, nchr * (sizeof(BinChrom) + sizeof(ChromOut))); if (rc) return rc;
    WsCarver ws(ctx->ws);
    BinChrom* dCh = ws.take<BinChrom>(nchr); unsigned long long* dPos0 = ws.take<unsigned long long>(nchr);
    uint32_t* tilePop = ws.take<uint32_t>(plan.ntiles); uint32_t* tileObs = ws.take<uint32_t>(plan.ntiles);
    int32_t* rankBase = ws.take<int32_t>(plan.ntiles); ChromOut* dOut = ws.take<ChromOut>(nchr);
    memcpy(ctx->pin, plan.chroms.data(), nchr * sizeof(BinChrom));
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dCh, ctx->pin, nchr * sizeof(BinChrom), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_init_pos0, dim3((nchr + 63) / 64), dim3(64), 0, ctx->stream, dCh, nchr, dPos0);   // pos0 = len: popBefore irrelevant here
    { ProfScope ps(ctx, "bin_tile_stats", true);
      hipLaunchKernelGGL(k_tile_stats, dim3((unsigned)((plan.ntiles + 3) / 4)), dim3(256), 0, ctx->stream, dCh, nchr, plan.ntiles, 1, tilePop, tileObs); }
    hipLaunchKernelGGL(k_scan_tiles, dim3(nchr), dim3(1024), 0, ctx->stream, dCh, dPos0, tilePop, tileObs, 1, 0, rankBase, dOut);
    ChromOut* hOut = (ChromOut*)((char*)ctx->pin + nchr * sizeof(BinChrom));
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(hOut, dOut, nchr * sizeof(ChromOut), hipMemcpyDeviceToHost, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    CANVAS_HIP_TRY(ctx, hipGetLastError());
    for (int c = 0; c < nchr; c++) {
        if (h_observed) h_observed[c] = hOut[c].obs;
        if (h_possible) h_possible[c] = hOut[c].pop;
        if (h_rate) h_rate[c] = (int)hOut[c].obs / (double)(int)hOut[c].pop;   // int / (double)int, CanvasBin.cs:60
    }
    return CANVAS_OK;
}

// ---- mode 5 pre-pass (BinCounts, CanvasBin.cs:427-505): mean fragment size, read-GC profile of EVERY chromosome that has fragment lengths, observed / expected weights.  Shared by
// the binning call (bin_genome_impl) and the predefined bins (canvas_bin_predefined_gcweighted).  The arena (1 B/base read-GC profile + tables) lives in the context and only grows.
struct GcwPre {
    canvas_ctx* ctx = nullptr; int nchr = 0;
    std::vector<GcwChrom> hGch; float* dW = nullptr; GcwChrom* dGch = nullptr; unsigned long long* dGcStats = nullptr; float* dLut = nullptr;
    bool overlap = false, pending = false;
    ~GcwPre() { if (pending && ctx && ctx->side) (void)hipStreamSynchronize(ctx->side); }      // an early return leaves nothing running on the arena
};
// the second half of the pre-pass: histograms of the read-GC profile -> observed vs expected weights (CanvasBin.cs:372-391) -> device.  Staging in the side stream's pinned
// buffer ({replicas of the histograms, the 101 weights, the GcwChrom table}: pinned copies are asynchronous whatever other streams are doing)
static int32_t gcw_prepass_finish(canvas_ctx* ctx, GcwPre& P) {
    CANVAS_HIP_TRY(ctx, hipEventSynchronize(ctx->side_ev2));
    P.pending = false;
    const unsigned long long* hr = (const unsigned long long*)ctx->side_pin;
    unsigned long long hh[202];
    for (int b = 0; b < 202; b++) { hh[b] = 0; for (int r = 0; r < RG_REP_ALL; r++) hh[b] += hr[(size_t)r * 202 + b]; }
    if (ctx->gcw_reduce) { int32_t rcr = ctx->gcw_reduce(ctx->gcw_reduce_user, hh, 202); if (rcr) return rcr; }      // the read-GC profile is the whole genome's (CanvasBin.cs:372-391)
    long long sumObserved = 0, sumExpected = 0;
    for (int b = 0; b < 101; b++) { sumExpected += (long long)hh[b]; sumObserved += (long long)hh[101 + b]; }
    float* w = (float*)((char*)ctx->side_pin + (size_t)RG_REP_ALL * 202 * 8);
    for (int b = 0; b < 101; b++) {
        long long e = (long long)hh[b], o = (long long)hh[101 + b];
        if (e == 0) e = 1;
        if (o == 0) o = 1;
        w[b] = ((float)o / (float)e) * ((float)sumExpected / (float)sumObserved);
    }
    GcwChrom* hg = (GcwChrom*)((char*)ctx->side_pin + (size_t)RG_REP_ALL * 202 * 8 + 512);
    for (int c = 0; c < P.nchr; c++) hg[c] = P.hGch[c];
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(P.dW, w, 101 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(P.dGch, hg, P.nchr * sizeof(GcwChrom), hipMemcpyHostToDevice, ctx->stream));
    return CANVAS_OK;
}
static int32_t gcw_prepass_begin(canvas_ctx* ctx, int32_t nchr, const uint8_t* const* d_bases, const uint8_t* const* d_hits, const int16_t* const* d_fraglen, const int64_t* h_len, bool mayOverlap, GcwPre& P) {
    P.ctx = ctx; P.nchr = nchr;
    int64_t totLen = 0;
    for (int c = 0; c < nchr; c++) totLen += (h_len[c] + 255) & ~255ll;
    size_t bytes = (size_t)totLen + 4096 + (size_t)nchr * (16 * NZ_REP + sizeof(GcwChrom) + sizeof(RgChrom)) + (size_t)RG_REP_ALL * 202 * 8 + 101 * 4 + (GCW_HMAX + 1) * 101 * 4 + 8192;
    if (bytes > ctx->gc_arena_bytes) {
        if (ctx->gc_arena) { CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); CANVAS_HIP_TRY(ctx, hipFree(ctx->gc_arena)); ctx->gc_arena = nullptr; ctx->gc_arena_bytes = 0; }
        CANVAS_HIP_TRY(ctx, hipMalloc(&ctx->gc_arena, bytes)); ctx->gc_arena_bytes = bytes;
    }
    uint8_t* gcArena = (uint8_t*)ctx->gc_arena;
    uint8_t* p = gcArena;
    P.hGch.assign(nchr, GcwChrom{nullptr}); std::vector<GcwChrom>& gch = P.hGch;
    for (int c = 0; c < nchr; c++) { gch[c].readGc = p; p += (h_len[c] + 255) & ~255ll; }
    p = (uint8_t*)(((uintptr_t)p + 255) & ~(uintptr_t)255);
    unsigned long long* sumCnt = (unsigned long long*)p; p += ((size_t)nchr * NZ_REP * 16 + 255) & ~size_t(255);   // NZ_REP replicas of {sum, count} per chromosome
    unsigned long long* hist = (unsigned long long*)p; p += ((size_t)RG_REP_ALL * 202 * 8 + 255) & ~size_t(255);    // replicas of {expected[101], observed[101]}
    P.dGcStats = (unsigned long long*)p; p += 256;       // GCW_REP counters of replayed bins
    P.dW = (float*)p; p += 512;
    P.dLut = (float*)p; p += (((GCW_HMAX + 1) * 101 * 4 + 255) & ~255);
    P.dGch = (GcwChrom*)p; p += ((size_t)nchr * sizeof(GcwChrom) + 255) & ~size_t(255);
    RgChrom* dRg = (RgChrom*)p;
    // Single GPU: the pre-pass runs on the side stream and the binning sweep below (k_tile_summary ... the bin boundaries: HBM-bound, no LDS, independent of the weights) runs
    // beside k_read_gc3, which is bound by its VALU / LDS work and leaves half of the memory bandwidth idle; the two meet in gcw_finish() in front of the weighted counts.
    // Sharded (ctx->gcw_reduce: the reductions have their place in the ranks' exchange order) and CANVAS_GCW_NO_OVERLAP=1: everything on ctx->stream, one after the other.
    int32_t rc0 = canvas_side_init(ctx); if (rc0) return rc0;
    const size_t sidePinBytes = (65536 + 65544) * sizeof(double);
    const size_t offW = (size_t)RG_REP_ALL * 202 * 8, offG = offW + 512;
    if (offG + (size_t)nchr * sizeof(GcwChrom) > sidePinBytes) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "GCContentWeighted mode: too many chromosomes for the pinned staging buffer");
    P.overlap = mayOverlap && !ctx->gcw_reduce && !cvx_hook("CANVAS_GCW_NO_OVERLAP");
    hipStream_t sp = ctx->stream;
    if (P.overlap) {
        CANVAS_HIP_TRY(ctx, hipEventRecord(ctx->side_ev, ctx->stream)); CANVAS_HIP_TRY(ctx, hipStreamWaitEvent(ctx->side, ctx->side_ev, 0));      // the inputs are ready with respect to ctx->stream
        sp = ctx->side;
    }
    CANVAS_HIP_TRY(ctx, hipMemsetAsync(sumCnt, 0, (size_t)((char*)P.dW - (char*)sumCnt), sp));       // fragment sums, histogram replicas, decision counters: one fill
    // the chromosome table of the two genome-wide launches: tiles of RG_T positions, numbered through the chromosomes
    rc0 = canvas_pin_reserve(ctx, (size_t)nchr * (sizeof(RgChrom) + NZ_REP * 16) + 128); if (rc0) return rc0;
    int64_t ntileAll = 0;
    {
        RgChrom* hRg = (RgChrom*)ctx->pin;
        for (int c = 0; c < nchr; c++) { hRg[c] = RgChrom{d_bases[c], d_fraglen[c], d_hits[c], (uint8_t*)gch[c].readGc, h_len[c], ntileAll}; ntileAll += (h_len[c] + RG_T - 1) / RG_T; }
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dRg, hRg, (size_t)nchr * sizeof(RgChrom), hipMemcpyHostToDevice, sp));
    }
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || cus <= 0) cus = 256;
    hipLaunchKernelGGL(k_nonzero_mean_all, dim3((unsigned)std::min<int64_t>((int64_t)cus * 8, ntileAll)), dim3(256), 0, sp, dRg, nchr, ntileAll, sumCnt);
    unsigned long long* hs = (unsigned long long*)((char*)ctx->pin + (((size_t)nchr * sizeof(RgChrom) + 63) & ~size_t(63)));      // (behind the table: its upload may still be reading it)
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(hs, sumCnt, (size_t)nchr * NZ_REP * 16, hipMemcpyDeviceToHost, sp));
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(sp));
    // MeanFragmentSize (CanvasBin.cs:164-174): NonZeroMean of the per-chromosome NonZeroMeans, all in Int16 with integer division
    long long s2 = 0, c2 = 0;
    for (int c = 0; c < nchr; c++) {
        unsigned long long sc = 0, cc = 0;
        for (int r = 0; r < NZ_REP; r++) { sc += hs[((size_t)c * NZ_REP + r) * 2]; cc += hs[((size_t)c * NZ_REP + r) * 2 + 1]; }
        int16_t m = cc ? (int16_t)(sc / cc) : 0; if (m > 0) { s2 += m; c2++; }
    }
    if (ctx->gcw_reduce) { unsigned long long v[2] = {(unsigned long long)s2, (unsigned long long)c2}; int32_t rcr = ctx->gcw_reduce(ctx->gcw_reduce_user, v, 2); if (rcr) return rcr; s2 = (long long)v[0]; c2 = (long long)v[1]; }      // chromosomes of the other ranks (canvas_bin_sample_sharded)
    const int meanFrag = c2 ? (int)(int16_t)(s2 / c2) : 0;
    if (meanFrag <= 0) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "CNV input error - unable to determine fragment size (CanvasBin.cs:431-434)");
    {
        const int nWmax = (RG_T + 3 * meanFrag) / 64 + 2;
        // k_read_gc3 follows the default window from position to position, which needs |100 d| < meanFragment; shorter fragments (and CANVAS_GCW_READ_GC2=1, the A/B and test
        // hook) take k_read_gc2, one launch per chromosome
        const bool rg3 = meanFrag > 100 && !cvx_hook("CANVAS_GCW_READ_GC2");
        const size_t ldsRg = (size_t)nWmax * (rg3 ? 16 : 12);
        int perCu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, rg3 ? (const void*)k_read_gc3<RG3_HE, RG3_HO> : (const void*)k_read_gc2, 256, ldsRg) != hipSuccess || perCu <= 0) perCu = 2;
        const unsigned gridRg = (unsigned)(perCu * cus);
        // x / meanFrag as (x * ceil(2^40 / meanFrag)) >> 40: exact while x < 2^22 and meanFrag < 2^15 (x = 100 * count <= 100 * 32767)
        const unsigned long long mean40 = ((1ull << 40) + (unsigned long long)meanFrag - 1ull) / (unsigned long long)meanFrag;
        ProfScope ps(ctx, "gcw_read_gc", false, sp);
        if (rg3) hipLaunchKernelGGL((k_read_gc3<RG3_HE, RG3_HO>), dim3((unsigned)std::min<int64_t>(gridRg, ntileAll)), dim3(256), ldsRg, sp, dRg, nchr, ntileAll, meanFrag, mean40, nWmax, hist);
        else for (int c = 0; c < nchr; c++) {
            const int64_t ntile = (h_len[c] + RG_T - 1) / RG_T;
            hipLaunchKernelGGL(k_read_gc2, dim3((unsigned)std::min<int64_t>(gridRg, ntile)), dim3(256), ldsRg, sp, d_bases[c], d_fraglen[c], d_hits[c], h_len[c], meanFrag, mean40, nWmax,
                               (uint8_t*)gch[c].readGc, hist);
        }
    }
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(ctx->side_pin, hist, (size_t)RG_REP_ALL * 202 * 8, hipMemcpyDeviceToHost, sp));
    CANVAS_HIP_TRY(ctx, hipEventRecord(ctx->side_ev2, sp));
    P.pending = true;
    if (!P.overlap) { rc0 = gcw_prepass_finish(ctx, P); if (rc0) return rc0; }
    return CANVAS_OK;
}

static int32_t bin_genome_impl(canvas_ctx* ctx, int32_t nchr, const uint8_t* const* d_bases, const uint64_t* const* d_mask,
                               const uint8_t* const* d_hits, const int16_t* const* d_fraglen, const int64_t* h_len, const uint8_t* h_is_auto, int32_t counts_per_bin, int32_t bin_size, int32_t mode,
                               int32_t* d_chr, int32_t* d_start, int32_t* d_stop, int32_t* d_gc, float* d_count, int64_t cap,
                               int32_t* h_bin_size_out, int64_t* h_nbins_per_chr, int64_t* h_nbins_total, cvx_bin_size_hook hook = nullptr, void* hookUser = nullptr,
                               const int64_t* h_pos0_packed = nullptr) {
    if (!ctx) return CANVAS_ERR_INVALID;
    const bool needRates = bin_size <= 0;
    const bool packed = h_pos0_packed != nullptr;       // bin_packed.hpp: d_bases = the reference planes, d_hits = the hit planes, d_mask unused
    if (nchr <= 0 || !d_bases || !d_mask || !d_hits || !h_len || (needRates && !hook && (!h_is_auto || counts_per_bin <= 0))) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_bin_genome: bad arguments");
    const bool gcw = mode == CANVAS_MODE_GC_CONTENT_WEIGHTED;
    if (gcw && !d_fraglen) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "GCContentWeighted mode needs the fragment-length arrays (canvas_bin_sample_gcweighted)");
    if (gcw && packed) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "the packed planes serve Binary and TruncatedDynamicRange; GCContentWeighted reads the per-base arrays");
    if (mode != CANVAS_MODE_BINARY && mode != CANVAS_MODE_TRUNCATED_DYNAMIC_RANGE && !gcw) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "unknown coverage mode");
    for (int c = 0; c < nchr; c++) if (h_len[c] <= 0 || h_len[c] > 0x7FFFFFFFll) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "chromosome length must be in [1, 2^31)");
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    BinPlan plan = make_plan(nchr, d_bases, d_mask, d_hits, h_len);
    GcwPre gp;
    if (gcw) { ctx->gcw_stats_dev = nullptr; ctx->gcw_total = 0; }       // (set again behind the weighted kernels: a call that fails or returns early in between leaves no pointer into an arena that may have been freed and regrown)
    if (gcw && ctx->up_active) {
        // the pre-pass below reads the per-base arrays on ctx->stream: a pending canvas_upload_genome_begin of them (copy stream) has to have landed first — mode 5 has no
        // per-chromosome overlap (the fence further down, which the other modes rely on, comes after these kernels)
        CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->copy)); ctx->up_active = false;
    }
    if (gcw) {
        for (int c = 0; c < nchr; c++) if (((uintptr_t)d_fraglen[c] | (uintptr_t)d_bases[c] | (uintptr_t)d_hits[c]) & 15) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "GCContentWeighted mode: bases, hits and fragment lengths must be 16-byte aligned");
        int32_t rc0 = gcw_prepass_begin(ctx, nchr, d_bases, d_hits, d_fraglen, h_len, true, gp); if (rc0) return rc0;
    }
    // the bin arrays are sized by the caller's capacity (the bin size may not be known yet)
    const int64_t ub = cap;
    WsSizer sz;
    const size_t tabSlots = (size_t)nchr + ((size_t)nchr + sizeof(BinChrom) - 1) / sizeof(BinChrom);      // the chromosome table + one byte per chromosome (is autosome) behind it: one upload
    sz.take<BinChrom>(tabSlots); sz.take<unsigned long long>(nchr); sz.take<uint32_t>(plan.ntiles); sz.take<uint32_t>(plan.ntiles); sz.take<int32_t>(plan.ntiles);
    sz.take<ChromOut>(nchr); sz.take<long long>(nchr + 1); sz.take<uint32_t>(plan.ntiles); sz.take<uint32_t>(plan.ntiles);
    sz.take<int32_t>(ub + 1); sz.take<uint32_t>(ub + 1); sz.take<uint32_t>(ub + 1);
    // bases/hits/mask streamed once (see k_tile_summary) when the rates are needed too; CANVAS_BIN_SINGLE_READ=1 takes that path for a given bin
    // size as well and CANVAS_BIN_TWO_PASS=1 never takes it (both are test hooks: the two paths must agree bit for bit)
    // a pending upload (canvas_upload_genome_begin) of exactly these arrays: every chromosome is swept as soon as it has arrived (single-read path only)
    bool streamed = ctx->up_active && (int)ctx->up_bases.size() == nchr && !gcw;
    for (int c = 0; streamed && c < nchr; c++) streamed = ctx->up_bases[c] == d_bases[c] && ctx->up_mask[c] == d_mask[c] && ctx->up_hits[c] == d_hits[c];
    if (ctx->up_active && !streamed) { CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->copy)); }    // other arrays (or mode 5): plain dependency on the whole upload
    ctx->up_active = false;
    const bool singleRead = packed || streamed || (needRates && !cvx_hook("CANVAS_BIN_TWO_PASS")) || cvx_hook("CANVAS_BIN_SINGLE_READ");
    const int nchunks = (int)((plan.ntiles + TS_CHUNK - 1) / TS_CHUNK);
    if (singleRead) { sz.take<uint32_t>(plan.ntiles * 64); sz.take<uint32_t>(plan.ntiles + 8); sz.take<TsPart>(nchunks + 1); sz.take<TsPart>(nchunks + 1); sz.take<unsigned long long>(nchr);
                      sz.take<TsPart>(nchr + 1); sz.take<ChromDev>(nchr); sz.take<uint4>(ub + 1); }
    int32_t rc = canvas_ws_reserve(ctx, sz.off + 4096); if (rc) return rc;
    // pinned staging: [chromosome table | is-autosome bytes] (one H2D), then what comes back: per-chromosome totals, (packed: pos0 going up), the decisions
    const size_t oAuto = (size_t)nchr * sizeof(BinChrom), oOut = (oAuto + (size_t)nchr + 15) & ~size_t(15), oP0 = oOut + (size_t)nchr * sizeof(ChromOut), oBd = oP0 + (size_t)nchr * 8;
    rc = canvas_pin_reserve(ctx, oBd + sizeof(BinDev) + 64); if (rc) return rc;
    WsCarver ws(ctx->ws);
    BinChrom* dCh = ws.take<BinChrom>(tabSlots); unsigned long long* dPos0 = ws.take<unsigned long long>(nchr);
    uint32_t* tilePop = ws.take<uint32_t>(plan.ntiles); uint32_t* tileObs = ws.take<uint32_t>(plan.ntiles); int32_t* rankBase = ws.take<int32_t>(plan.ntiles);
    ChromOut* dOut = ws.take<ChromOut>(nchr); long long* binOffset = ws.take<long long>(nchr + 1);
    uint32_t* tileTotC = ws.take<uint32_t>(plan.ntiles); uint32_t* tileTotG = ws.take<uint32_t>(plan.ntiles);
    int32_t* stopTmp = ws.take<int32_t>(ub + 1); uint32_t* locC = ws.take<uint32_t>(ub + 1); uint32_t* locG = ws.take<uint32_t>(ub + 1);
    uint32_t* wordSum = singleRead ? ws.take<uint32_t>(plan.ntiles * 64) : nullptr;
    uint32_t* rankRaw = nullptr; TsPart* tsPart = nullptr; TsPart* tsPartEx = nullptr; unsigned long long* dPopBefore = nullptr; TsPart* chrPre = nullptr; ChromDev* chrDev = nullptr; uint4* binRec = nullptr; uint8_t* dIsAuto = (uint8_t*)(dCh + nchr);
    if (singleRead) { rankRaw = ws.take<uint32_t>(plan.ntiles + 8); tsPart = ws.take<TsPart>(nchunks + 1); tsPartEx = ws.take<TsPart>(nchunks + 1); dPopBefore = ws.take<unsigned long long>(nchr);
                      chrPre = ws.take<TsPart>(nchr + 1); chrDev = ws.take<ChromDev>(nchr); binRec = ws.take<uint4>(ub + 1); }
    if (singleRead && !ctx->bin_dev) {
        CANVAS_HIP_TRY(ctx, hipMalloc(&ctx->bin_dev, 256)); CANVAS_HIP_TRY(ctx, hipMemsetAsync(ctx->bin_dev, 0, 256, ctx->stream));      // the tickets clean up after themselves
        CANVAS_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->bin_ev, hipEventDisableTiming));
    }
    uint32_t* dTick = (uint32_t*)ctx->bin_dev; BinDev* dBd = (BinDev*)((char*)ctx->bin_dev + 64);
    const int clampHits = mode == CANVAS_MODE_TRUNCATED_DYNAMIC_RANGE ? 1 : 0;
    const bool byVal = nchr <= BIN_BYVAL;
    if (byVal) {
        BinChromPack pack; memset(&pack, 0, sizeof pack);
        memcpy(pack.c, plan.chroms.data(), nchr * sizeof(BinChrom));
        for (int c = 0; c < nchr; c++) {
            pack.isAuto[c] = h_is_auto ? (h_is_auto[c] ? 1 : 0) : 0;
            if (packed) { if (h_pos0_packed[c] < 0 || h_pos0_packed[c] > h_len[c]) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_bin_sample_packed: pos0 outside [0, len]"); pack.pos0[c] = (unsigned long long)h_pos0_packed[c]; }
        }
        hipLaunchKernelGGL(k_bin_begin, dim3(1), dim3(256), 0, ctx->stream, dCh, dIsAuto, pack, nchr, dPos0, packed ? 1 : 0);
    } else {
        memcpy(ctx->pin, plan.chroms.data(), nchr * sizeof(BinChrom));
        for (int c = 0; c < nchr; c++) ((uint8_t*)ctx->pin)[oAuto + c] = h_is_auto ? (h_is_auto[c] ? 1 : 0) : 0;
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dCh, ctx->pin, oAuto + (size_t)nchr, hipMemcpyHostToDevice, ctx->stream));
    }
    ChromOut* hOut = (ChromOut*)((char*)ctx->pin + oOut);
    const unsigned pkGrid = (unsigned)((plan.ntiles + 4 * PK_TILES - 1) / (4 * PK_TILES));
    if (packed) {
        // pos0 comes with the planes (the packer found it)
        if (!byVal) {
            unsigned long long* hp0 = (unsigned long long*)((char*)ctx->pin + oP0);
            for (int c = 0; c < nchr; c++) {
                if (h_pos0_packed[c] < 0 || h_pos0_packed[c] > h_len[c]) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_bin_sample_packed: pos0 outside [0, len]");
                hp0[c] = (unsigned long long)h_pos0_packed[c];
            }
            CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dPos0, hp0, (size_t)nchr * 8, hipMemcpyHostToDevice, ctx->stream));
        }
        if (streamed) {
            for (int c = 0; c < nchr; c++) {
                const BinChrom& C = plan.chroms[c];
                CANVAS_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->up_ev[c], 0));
                ProfScope ps(ctx, "bin_summary_packed_streamed", true);
                hipLaunchKernelGGL(k_tile_summary_packed, dim3((unsigned)((C.ntiles + 4 * PK_TILES - 1) / (4 * PK_TILES))), dim3(256), 0, ctx->stream, dCh, nchr, C.tileBase + C.ntiles, dPos0,
                                   clampHits, needRates ? 1 : 0, wordSum, tilePop, tileObs, tileTotC, tileTotG, C.tileBase);
            }
        } else {
            ProfScope ps(ctx, "bin_summary_packed", true);
            hipLaunchKernelGGL(k_tile_summary_packed, dim3(pkGrid), dim3(256), 0, ctx->stream, dCh, nchr, plan.ntiles, dPos0, clampHits, needRates ? 1 : 0,
                               wordSum, tilePop, tileObs, tileTotC, tileTotG, (int64_t)0);
        }
    } else {
    if (!byVal) hipLaunchKernelGGL(k_init_pos0, dim3((nchr + 63) / 64), dim3(64), 0, ctx->stream, dCh, nchr, dPos0);
    if (streamed) {
        // copy / compute overlap: chromosome c's sweep waits for chromosome c's event only, chromosome c + 1 is on its way meanwhile
        for (int c = 0; c < nchr; c++) {
            const BinChrom& C = plan.chroms[c];
            CANVAS_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->up_ev[c], 0));
            hipLaunchKernelGGL(k_find_pos0, dim3(64, 1), dim3(256), 0, ctx->stream, dCh, dPos0, c);
            ProfScope ps(ctx, "bin_summary_streamed", true);
            hipLaunchKernelGGL(k_tile_summary, dim3((unsigned)((C.ntiles + 3) / 4)), dim3(256), 0, ctx->stream, dCh, nchr, C.tileBase + C.ntiles, dPos0, clampHits, needRates ? 1 : 0,
                               wordSum, tilePop, tileObs, tileTotC, tileTotG, C.tileBase);
            hipLaunchKernelGGL(k_tile_summary_edges, dim3(2), dim3(64), 0, ctx->stream, dCh, nchr, dPos0, clampHits, needRates ? 1 : 0, wordSum, tilePop, tileObs, tileTotC, tileTotG, c);
        }
    } else hipLaunchKernelGGL(k_find_pos0, dim3(64, nchr), dim3(256), 0, ctx->stream, dCh, dPos0, 0);
    if (streamed) {
    } else if (singleRead) {
        ProfScope ps(ctx, "bin_summary", true);
        hipLaunchKernelGGL(k_tile_summary, dim3((unsigned)((plan.ntiles + 3) / 4)), dim3(256), 0, ctx->stream, dCh, nchr, plan.ntiles, dPos0, clampHits, needRates ? 1 : 0,
                           wordSum, tilePop, tileObs, tileTotC, tileTotG, (int64_t)0);
    } else {
        ProfScope ps(ctx, "bin_tile_stats", true);
        hipLaunchKernelGGL(k_tile_stats, dim3((unsigned)((plan.ntiles + 3) / 4)), dim3(256), 0, ctx->stream, dCh, nchr, plan.ntiles, needRates ? 1 : 0, tilePop, tileObs);
    }
    if (singleRead && !streamed) hipLaunchKernelGGL(k_tile_summary_edges, dim3(2 * nchr), dim3(64), 0, ctx->stream, dCh, nchr, dPos0, clampHits, needRates ? 1 : 0, wordSum, tilePop, tileObs, tileTotC, tileTotG, 0);
    }   // byte arrays
    int64_t total = 0;
    if (singleRead) {
        // ---- the single-read path after the sweep: two scan launches, the sample's decisions taken by the last workgroup of the second (bin_tail.hpp), close + resolve
        // enqueued behind them; the host reads the decisions back while those run
        const int binSizeArg = !needRates ? bin_size : (hook ? -1 : 0);
        const uint32_t* obsArr = needRates ? tileObs : (const uint32_t*)nullptr;
        unsigned* hBdSeq = (unsigned*)((char*)ctx->pin + oBd + sizeof(BinDev));       // the decisions' mailbox stamp (cvx_mail_*, common.hpp)
        const unsigned bdSeq = hook ? 0u : cvx_mail_arm(ctx, hBdSeq);
        hipLaunchKernelGGL(k_tscan_reduce, dim3((unsigned)(nchunks + nchr)), dim3(TS_T), 0, ctx->stream, dCh, nchr, dPos0, packed ? 1 : 0, plan.ntiles, nchunks, tilePop, obsArr, tileTotC, tileTotG,
                           tsPart, tsPartEx, dPopBefore, dTick);
        hipLaunchKernelGGL(k_tscan_apply, dim3((unsigned)nchunks), dim3(TS_T), 0, ctx->stream, dCh, nchr, plan.ntiles, nchunks, tilePop, obsArr, tileTotC, tileTotG, rankRaw, tsPartEx, dPopBefore, chrPre,
                           dIsAuto, counts_per_bin, binSizeArg, (long long)cap, dOut, chrDev, binOffset, dBd, dTick + 1, hook ? (ChromOut*)nullptr : hOut, hook ? (BinDev*)nullptr : (BinDev*)((char*)ctx->pin + oBd),
                           hBdSeq, bdSeq);
        auto launch_close_resolve = [&]() {
            // (measured and dropped: close + resolve + finalize fused into one kernel — a wave collects its boundaries as tasks in LDS and resolves them densely — 376 us + a
            //  fix-up pass for each wave's first bin against 87 + 317 us for the two kernels below: the resolve is bound by the 128-byte line fills under the boundaries either way)
            { ProfScope ps(ctx, "bin_close");
              hipLaunchKernelGGL(k_bin_close2, dim3((unsigned)((plan.ntiles + 4 * CLOSE_TILES - 1) / (4 * CLOSE_TILES))), dim3(256), 0, ctx->stream, dCh, chrDev, nchr, plan.ntiles, wordSum, rankRaw,
                                 tileTotC, tileTotG, binOffset, dBd, binRec, d_chr); }
            ProfScope ps(ctx, "bin_resolve");
            // persistent workgroups (the number of bins is only known on the device): as many as are resident at once
            static const unsigned gridP = resident_grid((const void*)k_bin_resolve_fin<true>, 256, ctx->device), gridB = resident_grid((const void*)k_bin_resolve_fin<false>, 256, ctx->device);
            if (packed) hipLaunchKernelGGL((k_bin_resolve_fin<true>), dim3(gridP), dim3(256), 0, ctx->stream, dCh, chrDev, binOffset, dPos0, dBd, clampHits, binRec, d_start, d_stop, d_gc, d_count);
            else hipLaunchKernelGGL((k_bin_resolve_fin<false>), dim3(gridB), dim3(256), 0, ctx->stream, dCh, chrDev, binOffset, dPos0, dBd, clampHits, binRec, d_start, d_stop, d_gc, d_count);
        };
        auto host_totals = [&]() {
            total = 0;
            for (int c = 0; c < nchr; c++) { hOut[c].nbins = (hOut[c].pop - hOut[c].popBefore) / bin_size; if (h_nbins_per_chr) h_nbins_per_chr[c] = hOut[c].nbins; total += hOut[c].nbins; }
        };
        if (hook) {
            // chromosome-sharded pipeline: the rate pairs of every rank's chromosomes are exchanged inside the hook, every rank derives the same size
            CANVAS_HIP_TRY(ctx, hipMemcpyAsync(hOut, dOut, nchr * sizeof(ChromOut), hipMemcpyDeviceToHost, ctx->stream));
            CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            std::vector<long long> o(nchr), p(nchr), pb(nchr);
            for (int c = 0; c < nchr; c++) { o[c] = hOut[c].obs; p[c] = hOut[c].pop; pb[c] = hOut[c].popBefore; }
            int32_t rch = hook(hookUser, nchr, o.data(), p.data(), pb.data(), &bin_size); if (rch) return rch;
            if (bin_size <= 0) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "derived bin size is not positive");
            if (h_bin_size_out) *h_bin_size_out = bin_size;
            host_totals();
            if (h_nbins_total) *h_nbins_total = total;
            if (total > cap) CANVAS_FAIL(ctx, CANVAS_ERR_CAPACITY, "canvas_bin_genome: output capacity too small (see canvas_bin_count_upper_bound)");
            if (total == 0) return CANVAS_OK;
            hipLaunchKernelGGL(k_bin_plan, dim3(1), dim3(TS_T), 0, ctx->stream, nchr, dOut, binOffset, dBd, bin_size, (long long)cap);
            launch_close_resolve();
        } else {
            BinDev* hBd = (BinDev*)((char*)ctx->pin + oBd);       // (k_tscan_apply's last workgroup wrote the totals and the decisions into the pinned buffer itself)
            CANVAS_HIP_TRY(ctx, hipEventRecord(ctx->bin_ev, ctx->stream));
            launch_close_resolve();
            CANVAS_HIP_TRY(ctx, hipEventSynchronize(ctx->bin_ev));       // the decisions are on the host; close / resolve are still running
            { int32_t rcm = cvx_mail_await(ctx, hBdSeq, bdSeq, "canvas_bin: bin size and totals"); if (rcm) return rcm; }
            if (needRates && (hBd->flags & (BD_BAD_RATE | BD_TOO_MANY))) {
                // an autosome without possible positions, or more autosomes than the device sort holds: decided here as before (the kernels above did nothing)
                std::vector<double> rates;
                for (int c = 0; c < nchr; c++) if (h_is_auto[c]) rates.push_back((int)hOut[c].obs / (double)(int)hOut[c].pop);
                bin_size = canvas_bin_size_from_rates(rates.data(), (int32_t)rates.size(), counts_per_bin);
                if (bin_size <= 0) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "derived bin size is not positive");
                host_totals();
                if (total <= cap && total > 0) { hipLaunchKernelGGL(k_bin_plan, dim3(1), dim3(TS_T), 0, ctx->stream, nchr, dOut, binOffset, dBd, bin_size, (long long)cap); launch_close_resolve(); }
            } else {
                if (needRates && (hBd->flags & BD_NO_AUTOSOME)) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "no autosome to derive the bin size from");
                bin_size = hBd->binSize;
                if (bin_size <= 0) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "derived bin size is not positive");
                total = hBd->total;
                for (int c = 0; c < nchr; c++) if (h_nbins_per_chr) h_nbins_per_chr[c] = hOut[c].nbins;
            }
            if (h_bin_size_out) *h_bin_size_out = bin_size;
            if (h_nbins_total) *h_nbins_total = total;
            if (total > cap) CANVAS_FAIL(ctx, CANVAS_ERR_CAPACITY, "canvas_bin_genome: output capacity too small (see canvas_bin_count_upper_bound)");
            if (total == 0) return CANVAS_OK;
        }
    } else {
    if (needRates) {
        // two-pass path (CANVAS_BIN_TWO_PASS=1): rates (CanvasBin.cs:30-83) from the tile statistics, then the bin size on the host
        hipLaunchKernelGGL(k_scan_tiles, dim3(nchr), dim3(1024), 0, ctx->stream, dCh, dPos0, tilePop, tileObs, 1, 0, rankBase, dOut, packed ? 1 : 0);
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(hOut, dOut, nchr * sizeof(ChromOut), hipMemcpyDeviceToHost, ctx->stream));
        CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (hook) {
            std::vector<long long> o(nchr), p(nchr), pb(nchr);
            for (int c = 0; c < nchr; c++) { o[c] = hOut[c].obs; p[c] = hOut[c].pop; pb[c] = hOut[c].popBefore; }
            int32_t rch = hook(hookUser, nchr, o.data(), p.data(), pb.data(), &bin_size); if (rch) return rch;
        } else {
            std::vector<double> rates;
            for (int c = 0; c < nchr; c++) if (h_is_auto[c]) rates.push_back((int)hOut[c].obs / (double)(int)hOut[c].pop);
            if (rates.empty()) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "no autosome to derive the bin size from");
            bin_size = canvas_bin_size_from_rates(rates.data(), (int32_t)rates.size(), counts_per_bin);
        }
        if (bin_size <= 0) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "derived bin size is not positive");
    }
    if (h_bin_size_out) *h_bin_size_out = bin_size;
    if (!needRates) {
        hipLaunchKernelGGL(k_scan_tiles, dim3(nchr), dim3(1024), 0, ctx->stream, dCh, dPos0, tilePop, (const uint32_t*)nullptr, 0, bin_size, rankBase, dOut, packed ? 1 : 0);
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(hOut, dOut, nchr * sizeof(ChromOut), hipMemcpyDeviceToHost, ctx->stream));
        CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        CANVAS_HIP_TRY(ctx, hipGetLastError());
    }   // else: the scan that produced the rates already left the rank bases (they do not depend on the bin size)
    hipLaunchKernelGGL(k_bin_offsets, dim3(1), dim3(64), 0, ctx->stream, dOut, nchr, bin_size, binOffset);
    for (int c = 0; c < nchr; c++) { hOut[c].nbins = (hOut[c].pop - hOut[c].popBefore) / bin_size; if (h_nbins_per_chr) h_nbins_per_chr[c] = hOut[c].nbins; total += hOut[c].nbins; }
    if (h_nbins_total) *h_nbins_total = total;
    if (total > cap) CANVAS_FAIL(ctx, CANVAS_ERR_CAPACITY, "canvas_bin_genome: output capacity too small (see canvas_bin_count_upper_bound)");
    if (total == 0) return CANVAS_OK;
    { ProfScope ps(ctx, "bin_pass", true);
      hipLaunchKernelGGL(k_bin_pass, dim3((unsigned)((plan.ntiles + 3) / 4)), dim3(256), 0, ctx->stream, dCh, nchr, plan.ntiles, dPos0, rankBase,
                         binOffset, bin_size, clampHits, stopTmp, locC, locG, tileTotC, tileTotG); }
    hipLaunchKernelGGL(k_bin_pass_edges, dim3(2 * nchr), dim3(64), 0, ctx->stream, dCh, nchr, dPos0, rankBase, binOffset, bin_size,
                       clampHits, stopTmp, locC, locG, tileTotC, tileTotG);
    hipLaunchKernelGGL(k_scan_totals, dim3(nchr), dim3(1024), 0, ctx->stream, dCh, tileTotC, tileTotG);
    hipLaunchKernelGGL(k_bin_finalize, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, dCh, nchr, binOffset, dPos0, stopTmp, locC, locG,
                       tileTotC, tileTotG, d_chr, d_start, d_stop, d_gc, d_count, 0);
    }
    if (gcw) {
        if (gp.pending) { int32_t rcf = gcw_prepass_finish(ctx, gp); if (rcf) return rcf; }       // (single GPU) k_read_gc3 ran beside the kernels above
        ProfScope ps(ctx, "gcw_weighted");
        // one kernel that computes every position's term once, inside the bin that owns it
        const int serialOnly = cvx_hook("CANVAS_GCW_SERIAL") ? 1 : 0;      // (test hook: every bin through the reference's own order of additions)
        static const unsigned gridF = resident_grid((const void*)k_bin_weighted3, 256, ctx->device);
        hipLaunchKernelGGL(k_gcw_terms, dim3(1), dim3(256), 0, ctx->stream, gp.dW, gp.dLut);
        hipLaunchKernelGGL(k_bin_weighted3, dim3((unsigned)std::min<int64_t>(gridF, (total + 15) / 16)), dim3(256), 0, ctx->stream, dCh, gp.dGch, (long long)total, d_chr, d_start, d_stop, gp.dW, gp.dLut, d_count, gp.dGcStats, serialOnly, nchr);
        ctx->gcw_stats_dev = gp.dGcStats; ctx->gcw_total = (long long)total;       // how many bins the interval decided / how many replayed the reference's additions: canvas_bin_gcw_stats
    }
    CANVAS_HIP_TRY(ctx, hipGetLastError());
    return CANVAS_OK;
}

int32_t canvas_bin_gcw_stats(canvas_ctx* ctx, int64_t* h_out2) {
    if (!ctx || !h_out2) return CANVAS_ERR_INVALID;
    h_out2[0] = h_out2[1] = 0;
    if (!ctx->gcw_stats_dev) return CANVAS_OK;
    unsigned long long v[GCW_REP];
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    CANVAS_HIP_TRY(ctx, hipMemcpy(v, ctx->gcw_stats_dev, sizeof v, hipMemcpyDeviceToHost));
    long long rep = 0; for (int r = 0; r < GCW_REP; r++) rep += (long long)v[r];
    h_out2[0] = ctx->gcw_total - rep; h_out2[1] = rep;
    return CANVAS_OK;
}

int32_t canvas_bin_genome(canvas_ctx* ctx, int32_t nchr, const uint8_t* const* d_bases, const uint64_t* const* d_mask,
                          const uint8_t* const* d_hits, const int64_t* h_len, int32_t bin_size, int32_t mode,
                          int32_t* d_chr, int32_t* d_start, int32_t* d_stop, int32_t* d_gc, float* d_count, int64_t cap,
                          int64_t* h_nbins_per_chr, int64_t* h_nbins_total) {
    if (ctx && bin_size <= 0) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_bin_genome: bin size must be positive");
    return bin_genome_impl(ctx, nchr, d_bases, d_mask, d_hits, nullptr, h_len, nullptr, 0, bin_size, mode, d_chr, d_start, d_stop, d_gc, d_count, cap, nullptr, h_nbins_per_chr, h_nbins_total);
}

int32_t canvas_bin_sample(canvas_ctx* ctx, int32_t nchr, const uint8_t* const* d_bases, const uint64_t* const* d_mask,
                          const uint8_t* const* d_hits, const int64_t* h_len, const uint8_t* h_chr_is_autosome,
                          int32_t counts_per_bin, int32_t bin_size_in, int32_t mode,
                          int32_t* d_chr, int32_t* d_start, int32_t* d_stop, int32_t* d_gc, float* d_count, int64_t cap,
                          int32_t* h_bin_size_out, int64_t* h_nbins_per_chr, int64_t* h_nbins_total) {
    return bin_genome_impl(ctx, nchr, d_bases, d_mask, d_hits, nullptr, h_len, h_chr_is_autosome, counts_per_bin, bin_size_in, mode, d_chr, d_start, d_stop, d_gc, d_count, cap,
                           h_bin_size_out, h_nbins_per_chr, h_nbins_total);
}

int32_t canvas_bin_sample_gcweighted(canvas_ctx* ctx, int32_t nchr, const uint8_t* const* d_bases, const uint64_t* const* d_mask,
                                     const uint8_t* const* d_hits, const int16_t* const* d_fraglen, const int64_t* h_len, const uint8_t* h_chr_is_autosome,
                                     int32_t counts_per_bin, int32_t bin_size_in,
                                     int32_t* d_chr, int32_t* d_start, int32_t* d_stop, int32_t* d_gc, float* d_count, int64_t cap,
                                     int32_t* h_bin_size_out, int64_t* h_nbins_per_chr, int64_t* h_nbins_total) {
    return bin_genome_impl(ctx, nchr, d_bases, d_mask, d_hits, d_fraglen, h_len, h_chr_is_autosome, counts_per_bin, bin_size_in, CANVAS_MODE_GC_CONTENT_WEIGHTED,
                           d_chr, d_start, d_stop, d_gc, d_count, cap, h_bin_size_out, h_nbins_per_chr, h_nbins_total);
}

// ---- packed planes (bin_packed.hpp)
int32_t canvas_packed_plane_bytes(int64_t len, int64_t* ref_bytes, int64_t* hit_bytes) {
    if (len <= 0) return CANVAS_ERR_INVALID;
    const int64_t words = ((len + TILE - 1) / TILE) * 64;               // padded to whole tiles
    if (ref_bytes) *ref_bytes = words * 16;
    if (hit_bytes) *hit_bytes = words * 32;
    return CANVAS_OK;
}

int32_t canvas_bin_sample_packed(canvas_ctx* ctx, int32_t nchr, const uint64_t* const* d_ref, const uint64_t* const* d_hit_planes, const int64_t* h_len, const int64_t* h_pos0,
                                 const uint8_t* h_chr_is_autosome, int32_t counts_per_bin, int32_t bin_size_in, int32_t mode,
                                 int32_t* d_chr, int32_t* d_start, int32_t* d_stop, int32_t* d_gc, float* d_count, int64_t cap,
                                 int32_t* h_bin_size_out, int64_t* h_nbins_per_chr, int64_t* h_nbins_total) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (!h_pos0 || !d_ref || !d_hit_planes) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_bin_sample_packed: bad arguments");
    return bin_genome_impl(ctx, nchr, (const uint8_t* const*)d_ref, d_ref, (const uint8_t* const*)d_hit_planes, nullptr, h_len, h_chr_is_autosome, counts_per_bin, bin_size_in, mode,
                           d_chr, d_start, d_stop, d_gc, d_count, cap, h_bin_size_out, h_nbins_per_chr, h_nbins_total, nullptr, nullptr, h_pos0);
}

// per-base arrays already in HBM -> planes (either half may be omitted: a second sample over a packed reference passes d_bases = d_mask = NULL)
int32_t canvas_pack_genome_device(canvas_ctx* ctx, int32_t nchr, const uint8_t* const* d_bases, const uint64_t* const* d_mask, const uint8_t* const* d_hits, const int64_t* h_len,
                                  uint64_t* const* d_ref_out, uint64_t* const* d_hit_planes_out, int64_t* h_pos0_out, int64_t* h_saturated_out) {
    if (!ctx) return CANVAS_ERR_INVALID;
    const bool doRef = d_bases && d_mask && d_ref_out, doHits = d_hits && d_hit_planes_out;
    if (nchr <= 0 || !h_len || (!doRef && !doHits) || (doRef && !h_pos0_out)) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_pack_genome_device: bad arguments");
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    { int32_t rcf = canvas_upload_fence(ctx); if (rcf) return rcf; }
    for (int c = 0; c < nchr; c++) if (h_len[c] <= 0 || h_len[c] > 0x7FFFFFFFll) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "chromosome length must be in [1, 2^31)");
    WsSizer sz; sz.take<BinChrom>(nchr); sz.take<unsigned long long>(nchr + 1);
    int32_t rc = canvas_ws_reserve(ctx, sz.off + 4096); if (rc) return rc;
    rc = canvas_pin_reserve(ctx, nchr * (sizeof(BinChrom) + 8) + 8); if (rc) return rc;
    WsCarver ws(ctx->ws);
    BinChrom* dCh = ws.take<BinChrom>(nchr); unsigned long long* dPos0 = ws.take<unsigned long long>(nchr + 1);   // [nchr] = saturated positions
    if (doRef) {
        BinPlan plan = make_plan(nchr, d_bases, d_mask, d_hits, h_len);
        memcpy(ctx->pin, plan.chroms.data(), nchr * sizeof(BinChrom));
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dCh, ctx->pin, nchr * sizeof(BinChrom), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_init_pos0, dim3((nchr + 63) / 64), dim3(64), 0, ctx->stream, dCh, nchr, dPos0);
        hipLaunchKernelGGL(k_find_pos0, dim3(64, nchr), dim3(256), 0, ctx->stream, dCh, dPos0, 0);
    }
    CANVAS_HIP_TRY(ctx, hipMemsetAsync(dPos0 + nchr, 0, 8, ctx->stream));
    for (int c = 0; c < nchr; c++) {
        const int64_t words = ((h_len[c] + TILE - 1) / TILE) * 64;
        if (doRef) hipLaunchKernelGGL(k_pack_ref, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, ctx->stream, d_bases[c], d_mask[c], h_len[c], words, (ulonglong2*)d_ref_out[c]);
        if (doHits) hipLaunchKernelGGL(k_pack_hits, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, ctx->stream, d_hits[c], h_len[c], words, (ulonglong2*)d_hit_planes_out[c], dPos0 + nchr);
    }
    unsigned long long* hres = (unsigned long long*)((char*)ctx->pin + nchr * sizeof(BinChrom));
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(hres, dPos0, (size_t)(nchr + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    CANVAS_HIP_TRY(ctx, hipGetLastError());
    if (doRef) for (int c = 0; c < nchr; c++) h_pos0_out[c] = (int64_t)std::min<unsigned long long>(hres[c], (unsigned long long)h_len[c]);
    if (h_saturated_out) *h_saturated_out = (int64_t)hres[nchr];
    return CANVAS_OK;
}

// ---- the same packing on the host (no device involved): what a host that holds byte arrays runs before the upload; a host that fills the planes while it parses
// the FASTA / the BAM needs neither.  `threads` <= 0: one per hardware thread (at most 32).
static int pack_threads(int threads, int64_t words) {
    int t = threads > 0 ? threads : (int)std::min<unsigned>(32u, std::max(1u, std::thread::hardware_concurrency()));
    return (int)std::max<int64_t>(1, std::min<int64_t>(t, words / 4096 + 1));
}
int32_t canvas_pack_reference_host(const uint8_t* bases, const uint64_t* mask, int64_t len, uint64_t* ref_out, int64_t* pos0_out, int32_t threads) {
    if (!bases || !mask || !ref_out || len <= 0) return CANVAS_ERR_INVALID;
    const int64_t words = ((len + TILE - 1) / TILE) * 64;
    const int nt = pack_threads(threads, words);
    std::vector<int64_t> first((size_t)nt, len);
    auto work = [&](int t) {
        const int64_t w0 = words * t / nt, w1 = words * (t + 1) / nt;
        int64_t firstNonN = len;
        for (int64_t w = w0; w < w1; w++) {
            const int64_t p = w << 6;
            uint64_t m = 0, g = 0;
            if (p + 64 <= len) {
                // 16 bases at a time: case folded, compared with 'c' and 'g', one movemask; the first base that is not 'n' from the same registers
                m = mask[w];
                const __m128i fold = _mm_set1_epi8(0x20), cc = _mm_set1_epi8('c'), gg = _mm_set1_epi8('g'), nn = _mm_set1_epi8('n');
                for (int q = 0; q < 4; q++) {
                    const __m128i v = _mm_loadu_si128((const __m128i*)(bases + p + 16 * q));
                    const __m128i lb = _mm_or_si128(v, fold);
                    g |= (uint64_t)(unsigned)_mm_movemask_epi8(_mm_or_si128(_mm_cmpeq_epi8(lb, cc), _mm_cmpeq_epi8(lb, gg))) << (16 * q);
                    if (firstNonN == len) { const unsigned notN = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(v, nn)) ^ 0xFFFFu; if (notN) firstNonN = p + 16 * q + __builtin_ctz(notN); }
                }
            } else if (p < len) {
                const int n = (int)std::min<int64_t>(64, len - p);
                m = mask[w]; if (n < 64) m &= (~0ull) >> (64 - n);
                for (int i = 0; i < n; i++) {
                    const uint8_t b = bases[p + i];
                    const uint8_t lb = b | 0x20;
                    g |= (uint64_t)(lb == 'c' || lb == 'g') << i;
                    if (b != 'n' && firstNonN == len) firstNonN = p + i;
                }
            }
            ref_out[2 * w] = m; ref_out[2 * w + 1] = g;
        }
        first[t] = firstNonN;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    if (pos0_out) { int64_t p0 = len; for (int t = 0; t < nt; t++) p0 = std::min(p0, first[t]); *pos0_out = p0; }
    return CANVAS_OK;
}
int32_t canvas_pack_hits_host(const uint8_t* hits, int64_t len, uint64_t* planes_out, int64_t* saturated_out, int32_t threads) {
    if (!hits || !planes_out || len <= 0) return CANVAS_ERR_INVALID;
    const int64_t words = ((len + TILE - 1) / TILE) * 64;
    const int nt = pack_threads(threads, words);
    std::vector<int64_t> sat((size_t)nt, 0);
    auto work = [&](int t) {
        const int64_t w0 = words * t / nt, w1 = words * (t + 1) / nt;
        int64_t nsat = 0;
        const __m128i fifteen = _mm_set1_epi8(15);
        for (int64_t w = w0; w < w1; w++) {
            const int64_t p = w << 6;
            uint64_t b[4] = {0, 0, 0, 0};
            if (p + 64 <= len) {
                // 16 positions at a time: the k-th bit of every byte moved to the byte's top bit, then one movemask
                for (int q = 0; q < 4; q++) {
                    const __m128i v = _mm_loadu_si128((const __m128i*)(hits + p + 16 * q));
                    const __m128i c = _mm_min_epu8(v, fifteen);
                    nsat += __builtin_popcount((unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(c, v)) ^ 0xFFFFu);
                    b[0] |= (uint64_t)(unsigned)_mm_movemask_epi8(_mm_slli_epi16(c, 7)) << (16 * q);
                    b[1] |= (uint64_t)(unsigned)_mm_movemask_epi8(_mm_slli_epi16(c, 6)) << (16 * q);
                    b[2] |= (uint64_t)(unsigned)_mm_movemask_epi8(_mm_slli_epi16(c, 5)) << (16 * q);
                    b[3] |= (uint64_t)(unsigned)_mm_movemask_epi8(_mm_slli_epi16(c, 4)) << (16 * q);
                }
            } else if (p < len) {
                for (int i = 0; p + i < len; i++) {
                    unsigned h = hits[p + i];
                    if (h > 15u) { h = 15u; nsat++; }
                    for (int k = 0; k < 4; k++) b[k] |= (uint64_t)((h >> k) & 1u) << i;
                }
            }
            planes_out[4 * w] = b[0]; planes_out[4 * w + 1] = b[1]; planes_out[4 * w + 2] = b[2]; planes_out[4 * w + 3] = b[3];
        }
        sat[t] = nsat;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    if (saturated_out) { int64_t s2 = 0; for (int t = 0; t < nt; t++) s2 += sat[t]; *saturated_out = s2; }
    return CANVAS_OK;
}

// ---- two-bit wire format of the hit planes (bin_packed.hpp): host packer
int32_t canvas_pack_hits2_host(const uint8_t* hits, int64_t len, uint64_t* lo_out, uint64_t* hdr_out, uint64_t* extras_out, int64_t extras_cap_words, int64_t* n_extras_out,
                               int64_t* saturated_out, int32_t threads) {
    if (!hits || !lo_out || !hdr_out || !extras_out || !n_extras_out || len <= 0 || extras_cap_words < 0) return CANVAS_ERR_INVALID;
    const int64_t words = ((len + TILE - 1) / TILE) * 64, tiles = words / 64;
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(pack_threads(threads, words), tiles));
    std::vector<std::vector<uint64_t>> ext((size_t)nt);          // {b2, b3} of this thread's words that have one, in word order
    std::vector<int64_t> sat((size_t)nt, 0);
    auto work = [&](int t) {
        const int64_t t0 = tiles * t / nt, t1 = tiles * (t + 1) / nt;    // whole tiles per thread: a tile's header entry has one writer
        int64_t nsat = 0;
        const __m128i fifteen = _mm_set1_epi8(15);
        std::vector<uint64_t>& E = ext[t];
        for (int64_t tile = t0; tile < t1; tile++) {
            uint64_t xmask = 0; const uint64_t firstLocal = E.size() / 2;
            for (int wi = 0; wi < 64; wi++) {
                const int64_t w = tile * 64 + wi, p = w << 6;
                uint64_t b[4] = {0, 0, 0, 0};
                if (p + 64 <= len) {
                    for (int q = 0; q < 4; q++) {
                        const __m128i v = _mm_loadu_si128((const __m128i*)(hits + p + 16 * q));
                        const __m128i c = _mm_min_epu8(v, fifteen);
                        nsat += __builtin_popcount((unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(c, v)) ^ 0xFFFFu);
                        b[0] |= (uint64_t)(unsigned)_mm_movemask_epi8(_mm_slli_epi16(c, 7)) << (16 * q);
                        b[1] |= (uint64_t)(unsigned)_mm_movemask_epi8(_mm_slli_epi16(c, 6)) << (16 * q);
                        b[2] |= (uint64_t)(unsigned)_mm_movemask_epi8(_mm_slli_epi16(c, 5)) << (16 * q);
                        b[3] |= (uint64_t)(unsigned)_mm_movemask_epi8(_mm_slli_epi16(c, 4)) << (16 * q);
                    }
                } else if (p < len) {
                    for (int i = 0; p + i < len; i++) {
                        unsigned h = hits[p + i];
                        if (h > 15u) { h = 15u; nsat++; }
                        for (int k = 0; k < 4; k++) b[k] |= (uint64_t)((h >> k) & 1u) << i;
                    }
                }
                lo_out[2 * w] = b[0]; lo_out[2 * w + 1] = b[1];
                if (b[2] | b[3]) { xmask |= 1ull << wi; E.push_back(b[2]); E.push_back(b[3]); }
            }
            hdr_out[2 * tile] = xmask; hdr_out[2 * tile + 1] = firstLocal;       // local index: the thread's base is added below
        }
        sat[t] = nsat;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    int64_t total = 0; std::vector<int64_t> base((size_t)nt, 0);
    for (int t = 0; t < nt; t++) { base[t] = total; total += (int64_t)ext[t].size() / 2; }
    *n_extras_out = total;
    if (saturated_out) { int64_t s2 = 0; for (int t = 0; t < nt; t++) s2 += sat[t]; *saturated_out = s2; }
    if (total > extras_cap_words) return CANVAS_ERR_CAPACITY;
    auto fix = [&](int t) {
        const int64_t t0 = tiles * t / nt, t1 = tiles * (t + 1) / nt;
        for (int64_t tile = t0; tile < t1; tile++) hdr_out[2 * tile + 1] += (uint64_t)base[t];
        if (!ext[t].empty()) memcpy(extras_out + 2 * base[t], ext[t].data(), ext[t].size() * 8);
    };
    std::vector<std::thread> th2;
    for (int t = 1; t < nt; t++) th2.emplace_back(fix, t);
    fix(0);
    for (auto& x : th2) x.join();
    return CANVAS_OK;
}

// canvas_upload_packed_begin with the hit planes in their two-bit wire form: per chromosome the three pieces travel to a staging area of the context and are expanded
// into d_hit_planes[c] on the copy stream, right behind their transfer; the chromosome's event is recorded after the expansion
int32_t canvas_upload_packed2_begin(canvas_ctx* ctx, int32_t nchr, const int64_t* h_len, const uint64_t* const* h_ref, uint64_t* const* d_ref,
                                    const uint64_t* const* h_lo, const uint64_t* const* h_hdr, const uint64_t* const* h_extras, const int64_t* h_n_extras, uint64_t* const* d_hit_planes) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (nchr <= 0 || !h_len || !d_ref || !h_lo || !h_hdr || !h_extras || !h_n_extras || !d_hit_planes) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_upload_packed2_begin: bad arguments");
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->copy) CANVAS_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy, hipStreamNonBlocking));
    while ((int)ctx->up_ev.size() < nchr) { hipEvent_t e; CANVAS_HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming)); ctx->up_ev.push_back(e); }
    if (!ctx->up_fence) CANVAS_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->up_fence, hipEventDisableTiming));
    auto al = [](size_t v) { return (v + 255) & ~size_t(255); };
    size_t need = 0;
    for (int c = 0; c < nchr; c++) {
        if (h_len[c] <= 0 || h_n_extras[c] < 0) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_upload_packed2_begin: bad chromosome");
        const size_t words = (size_t)((h_len[c] + TILE - 1) / TILE) * 64;
        need += al(words * 16) + al(words / 64 * 16) + al((size_t)h_n_extras[c] * 16 + 16);
    }
    if (need > ctx->up2_stage_bytes) {
        if (ctx->up2_stage) { CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->copy)); CANVAS_HIP_TRY(ctx, hipFree(ctx->up2_stage)); ctx->up2_stage = nullptr; ctx->up2_stage_bytes = 0; }
        CANVAS_HIP_TRY(ctx, hipMalloc(&ctx->up2_stage, need + need / 8)); ctx->up2_stage_bytes = need + need / 8;
    }
    // the staging area and the destinations may still be read by work queued on the compute stream (the previous pass): the copies start after it
    CANVAS_HIP_TRY(ctx, hipEventRecord(ctx->up_fence, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipStreamWaitEvent(ctx->copy, ctx->up_fence, 0));
    ctx->up_bases.assign(nchr, nullptr); ctx->up_mask.assign(nchr, nullptr); ctx->up_hits.assign(nchr, nullptr);
    char* st = (char*)ctx->up2_stage;
    for (int c = 0; c < nchr; c++) {
        const size_t words = (size_t)((h_len[c] + TILE - 1) / TILE) * 64;
        char* dLo = st; st += al(words * 16); char* dHdr = st; st += al(words / 64 * 16); char* dEx = st; st += al((size_t)h_n_extras[c] * 16 + 16);
        if (h_ref && h_ref[c]) CANVAS_HIP_TRY(ctx, hipMemcpyAsync(d_ref[c], h_ref[c], words * 16, hipMemcpyHostToDevice, ctx->copy));
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dLo, h_lo[c], words * 16, hipMemcpyHostToDevice, ctx->copy));
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dHdr, h_hdr[c], words / 64 * 16, hipMemcpyHostToDevice, ctx->copy));
        if (h_n_extras[c] > 0) CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dEx, h_extras[c], (size_t)h_n_extras[c] * 16, hipMemcpyHostToDevice, ctx->copy));
        hipLaunchKernelGGL(k_expand_hits2, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, ctx->copy, (const ulonglong2*)dLo, (const ulonglong2*)dHdr, (const ulonglong2*)dEx, (int64_t)words,
                           (ulonglong2*)d_hit_planes[c]);
        CANVAS_HIP_TRY(ctx, hipEventRecord(ctx->up_ev[c], ctx->copy));
        ctx->up_bases[c] = d_ref[c]; ctx->up_mask[c] = d_ref[c]; ctx->up_hits[c] = d_hit_planes[c];
    }
    CANVAS_HIP_TRY(ctx, hipGetLastError());
    ctx->up_active = true;
    return CANVAS_OK;
}

}  // extern "C"

int32_t cvx_bin_sample_hooked(canvas_ctx* ctx, int32_t nchr, const uint8_t* const* d_bases, const uint64_t* const* d_mask, const uint8_t* const* d_hits, const int64_t* h_len,
                              int32_t mode, cvx_bin_size_hook hook, void* user, int32_t* d_chr, int32_t* d_start, int32_t* d_stop, int32_t* d_gc, float* d_count, int64_t cap,
                              int64_t* h_nbins_per_chr, int64_t* h_nbins_total, const int64_t* h_pos0_packed, const int16_t* const* d_fraglen) {
    // h_pos0_packed given: d_bases / d_hits are the packed reference / hit planes of these chromosomes (canvas_bin_sample_packed); d_fraglen: GCContentWeighted
    return bin_genome_impl(ctx, nchr, d_bases, d_mask, d_hits, d_fraglen, h_len, nullptr, 0, -1, mode, d_chr, d_start, d_stop, d_gc, d_count, cap, nullptr, h_nbins_per_chr, h_nbins_total, hook, user, h_pos0_packed);
}

// ---------------------------------------------------------------------------------------------- predefined bins (CanvasBin -n)
// BinCountsForChromosome with usePredefinedBins (CanvasBin.cs:575-655): the cursor jumps from interval to interval, so every bin is the range [Start, Stop) of its own —
// except the chromosome's FIRST bin, whose leading 'n' bases are skipped before anything is counted (:582-584).  One wave per bin, 64 positions per step.
struct PreChrom { const uint8_t* bases; const uint8_t* hits; const uint64_t* mask; long long len; long long binBegin, binEnd; };
__global__ void __launch_bounds__(256) k_bin_predefined(const PreChrom* __restrict__ ch, int nchr, long long nbins, const int32_t* __restrict__ bStart, const int32_t* __restrict__ bStop, int clampHits,
                                                        int32_t* __restrict__ oGc, float* __restrict__ oCount, int* __restrict__ err, int32_t* __restrict__ oChrIdx, int32_t* __restrict__ oStartUsed) {
    const long long b = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= nbins) return;
    int lo = 0, hi = nchr - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (ch[mid].binBegin <= b) lo = mid; else hi = mid - 1; }
    const PreChrom C = ch[lo];
    const int l = lane_id();
    long long s = bStart[b]; const long long e = bStop[b];
    if (b == C.binBegin) {                                   // "Skip past leading Ns" from the first bin's start
        for (;;) {
            const long long p = s + l;
            const bool isn = p < C.len && C.bases[p] == 'n';
            const unsigned long long m = __ballot(!isn);
            if (m) { s += __builtin_ctzll(m); break; }
            s += 64;
            if (s >= C.len) break;
        }
        if (s > e - 1) { if (l == 0) atomicExch(err, 1); return; }      // the reference's cursor never meets Stop - 1 again: no bin of this chromosome would ever be closed
    }
    uint32_t gcn = 0, obs = 0;
    for (long long p0 = s; p0 < e; p0 += 64) {
        const long long p = p0 + l;
        if (p < e) {
            const uint8_t c = C.bases[p];
            gcn += (c == 'C' || c == 'c' || c == 'G' || c == 'g') ? 1u : 0u;
            if ((C.mask[p >> 6] >> (p & 63)) & 1ull) { const uint32_t h = C.hits[p]; obs += clampHits ? min(10u, h) : h; }
        }
    }
    gcn = wave_reduce_add_u32(gcn); obs = wave_reduce_add_u32(obs);
    if (l == 0) {
        const int nucleotideCount = (int)(e - s);            // every position counts: "!Bases[pos].Equals("n")" compares a char with a string and is always true (:594)
        oGc[b] = (int32_t)(100.0f * (float)(int)gcn / (float)nucleotideCount);
        oCount[b] = (float)(int)obs;
        if (oChrIdx) { oChrIdx[b] = lo; oStartUsed[b] = (int32_t)s; }      // GCContentWeighted: the range the weighted count runs over (k_bin_weighted3)
    }
}

static int32_t bin_predefined_impl(canvas_ctx* ctx, int32_t nchr, const uint8_t* const* d_bases, const uint64_t* const* d_mask, const uint8_t* const* d_hits, const int16_t* const* d_fraglen,
                                   const int64_t* h_len, int32_t mode, const int64_t* h_bin_offset, const int32_t* h_bin_start, const int32_t* h_bin_stop, const int32_t* d_bin_start,
                                   const int32_t* d_bin_stop, int32_t* d_gc, float* d_count) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (nchr <= 0 || !d_bases || !d_mask || !d_hits || !h_len || !h_bin_offset || !h_bin_start || !h_bin_stop || !d_bin_start || !d_bin_stop || !d_gc || !d_count)
        CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_bin_predefined: bad arguments");
    const bool gcw = mode == CANVAS_MODE_GC_CONTENT_WEIGHTED;
    if (gcw && !d_fraglen) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "GCContentWeighted mode needs the fragment-length arrays (canvas_bin_predefined_gcweighted)");
    if (mode != CANVAS_MODE_BINARY && mode != CANVAS_MODE_TRUNCATED_DYNAMIC_RANGE && !gcw) CANVAS_FAIL(ctx, CANVAS_ERR_UNSUPPORTED, "canvas_bin_predefined: coverage modes 0, 3 and 5");
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    { int32_t rcf = canvas_upload_fence(ctx); if (rcf) return rcf; }
    const long long nbins = h_bin_offset[nchr];
    std::vector<PreChrom> pc(nchr);
    for (int c = 0; c < nchr; c++) {
        if (h_len[c] <= 0 || h_len[c] > 0x7FFFFFFFll) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "chromosome length must be in [1, 2^31)");
        pc[c] = PreChrom{d_bases[c], d_hits[c], d_mask[c], (long long)h_len[c], (long long)h_bin_offset[c], (long long)h_bin_offset[c + 1]};
        for (long long b = h_bin_offset[c]; b < h_bin_offset[c + 1]; b++) {
            if (h_bin_start[b] < 0 || h_bin_start[b] >= h_bin_stop[b]) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "predefined bin with Start < 0 or Start >= Stop (Utilities.LoadBedFile throws)");
            if (h_bin_stop[b] > h_len[c]) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "predefined bin beyond the end of its chromosome (the reference's cursor never closes it)");
        }
    }
    // GCContentWeighted: the mean fragment size, the read-GC profile and the weights come from EVERY chromosome handed in (BinCounts computes them before it looks at the
    // predefined bins, CanvasBin.cs:427-505), also when there is no bin to fill — a sample without usable fragment lengths fails as the reference does (:431-434)
    GcwPre gp;
    if (gcw) {
        ctx->gcw_stats_dev = nullptr; ctx->gcw_total = 0;
        for (int c = 0; c < nchr; c++) if (((uintptr_t)d_fraglen[c] | (uintptr_t)d_bases[c] | (uintptr_t)d_hits[c]) & 15) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "GCContentWeighted mode: bases, hits and fragment lengths must be 16-byte aligned");
        int32_t rcp = gcw_prepass_begin(ctx, nchr, d_bases, d_hits, d_fraglen, h_len, false, gp); if (rcp) return rcp;
    }
    if (nbins == 0) {
        // the pre-pass has enqueued copies out of the pinned staging area (weights, chromosome table): they must have left it before the next call refills it
        if (gcw) CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return CANVAS_OK;
    }
    WsSizer sz;
    sz.take<PreChrom>(nchr); sz.take<int>(64); sz.take<BinChrom>(nchr); sz.take<int32_t>(nbins); sz.take<int32_t>(nbins);
    int32_t rc = canvas_ws_reserve(ctx, sz.off + 4096); if (rc) return rc;
    WsCarver cv(ctx->ws);
    PreChrom* dCh = cv.take<PreChrom>(nchr); int* dErr = cv.take<int>(64); BinChrom* dBc = cv.take<BinChrom>(nchr); int32_t* dChrIdx = cv.take<int32_t>(nbins); int32_t* dStartUsed = cv.take<int32_t>(nbins);
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dCh, pc.data(), (size_t)nchr * sizeof(PreChrom), hipMemcpyHostToDevice, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipMemsetAsync(dErr, 0, 4, ctx->stream));
    BinPlan plan;
    if (gcw) { plan = make_plan(nchr, d_bases, d_mask, d_hits, h_len); CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dBc, plan.chroms.data(), (size_t)nchr * sizeof(BinChrom), hipMemcpyHostToDevice, ctx->stream)); }
    hipLaunchKernelGGL(k_bin_predefined, dim3((unsigned)((nbins + 3) / 4)), dim3(256), 0, ctx->stream, dCh, nchr, nbins, d_bin_start, d_bin_stop, mode == CANVAS_MODE_TRUNCATED_DYNAMIC_RANGE ? 1 : 0, d_gc, d_count, dErr,
                       gcw ? dChrIdx : nullptr, gcw ? dStartUsed : nullptr);
    int err = 0;
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(&err, dErr, 4, hipMemcpyDeviceToHost, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    CANVAS_HIP_TRY(ctx, hipGetLastError());
    if (err) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "the first predefined bin of a chromosome lies entirely in leading 'n' bases (the reference then fills no bin of that chromosome)");
    if (gcw) {
        // the weighted count of a bin (CanvasBin.cs:626-636) over [start after the 'n' skip, Stop): the kernel of the binning call, on the predefined ranges
        const int serialOnly = cvx_hook("CANVAS_GCW_SERIAL") ? 1 : 0;
        static const unsigned gridF = resident_grid((const void*)k_bin_weighted3, 256, ctx->device);
        hipLaunchKernelGGL(k_gcw_terms, dim3(1), dim3(256), 0, ctx->stream, gp.dW, gp.dLut);
        hipLaunchKernelGGL(k_bin_weighted3, dim3((unsigned)std::min<int64_t>(gridF, (nbins + 15) / 16)), dim3(256), 0, ctx->stream, dBc, gp.dGch, nbins, dChrIdx, dStartUsed, d_bin_stop, gp.dW, gp.dLut, d_count,
                           gp.dGcStats, serialOnly, nchr);
        ctx->gcw_stats_dev = gp.dGcStats; ctx->gcw_total = nbins;
        CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));       // (dChrIdx / dStartUsed live in the workspace)
        CANVAS_HIP_TRY(ctx, hipGetLastError());
    }
    return CANVAS_OK;
}
extern "C" int32_t canvas_bin_predefined(canvas_ctx* ctx, int32_t nchr, const uint8_t* const* d_bases, const uint64_t* const* d_mask, const uint8_t* const* d_hits, const int64_t* h_len,
                                         int32_t mode, const int64_t* h_bin_offset, const int32_t* h_bin_start, const int32_t* h_bin_stop, const int32_t* d_bin_start, const int32_t* d_bin_stop,
                                         int32_t* d_gc, float* d_count) {
    if (mode == CANVAS_MODE_GC_CONTENT_WEIGHTED) { if (!ctx) return CANVAS_ERR_INVALID; CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_bin_predefined: GCContentWeighted needs the fragment lengths (canvas_bin_predefined_gcweighted)"); }
    return bin_predefined_impl(ctx, nchr, d_bases, d_mask, d_hits, nullptr, h_len, mode, h_bin_offset, h_bin_start, h_bin_stop, d_bin_start, d_bin_stop, d_gc, d_count);
}
extern "C" int32_t canvas_bin_predefined_gcweighted(canvas_ctx* ctx, int32_t nchr, const uint8_t* const* d_bases, const uint64_t* const* d_mask, const uint8_t* const* d_hits,
                                                    const int16_t* const* d_fraglen, const int64_t* h_len, const int64_t* h_bin_offset, const int32_t* h_bin_start, const int32_t* h_bin_stop,
                                                    const int32_t* d_bin_start, const int32_t* d_bin_stop, int32_t* d_gc, float* d_count) {
    return bin_predefined_impl(ctx, nchr, d_bases, d_mask, d_hits, d_fraglen, h_len, CANVAS_MODE_GC_CONTENT_WEIGHTED, h_bin_offset, h_bin_start, h_bin_stop, d_bin_start, d_bin_stop, d_gc, d_count);
}

