// Packed per-base inputs (included by bin.hip): the same BinCountsForChromosome (CanvasBin.cs:568-661) and rates (CanvasBin.cs:30-83) over a representation that
// carries only what those loops read, 0.75 B/base instead of 2.125 B/base — less to push over PCIe (the transfer bounds a whole pass: 57 GB/s against 8 TB/s of HBM)
// and less for the one full sweep of the path to read.
//
//   reference plane  per 64 positions {u64 possible, u64 gc}: bit i of `possible` = the BitArray bit of CanvasBin.cs:593 (unique k-mer start), bit i of `gc` =
//                    the base is C/c/G/g (the switch of CanvasBin.cs:599-606).  Depends on the reference genome only: packed once, shared by every sample.
//   hit planes       per 64 positions {u64 b0, b1, b2, b3}: bit-sliced 4-bit counters, bit i of b_k = bit k of min(15, hits[i]).  TruncatedDynamicRange reads
//                    min(10, hits) (CanvasBin.cs:618-619) and Binary mode holds 0 / 1 (HitArray increments saturate at 1 there), so saturating at 15 loses nothing
//                    in the modes this path serves; the packers report how many positions saturated.
//   pos0             first position whose base is not 'n' (CanvasBin.cs:582-584), found by the packer.
// Both planes are zero beyond `len` and padded to whole tiles (4096 positions), so no kernel carries bounds logic: the chromosome tail and the positions in front
// of pos0 are plain 64-bit masks.  With bit-sliced counters every per-word quantity is a popcount:
//   sum of masked hits = popc(b0 & m) + 2 popc(b1 & m) + 4 popc(b2 & m) + 8 popc(b3 & m),   observed = popc(b0 | b1 | b2 | b3),   min(10, h): h > 10 <=> b3 & (b2 | b1 & b0).
// The 4-byte summaries are the ones of k_tile_summary, so k_bin_close, the scans and k_bin_finalize are shared with the byte path; only the sweep and the boundary
// resolution (two cache lines per bin instead of three) differ.  Results are bit-identical to the byte path (tests/test_bin_packed_gpu.py).
#pragma once

#define PK_TILES 4          // tiles per wave in the packed sweep: 4 x (16 + 32) B per lane in flight


__device__ __forceinline__ void pk_clamp10(unsigned long long& b0, unsigned long long& b1, unsigned long long& b2, unsigned long long b3) {
    const unsigned long long over = b3 & (b2 | (b1 & b0));          // 11..15 -> 10 = 1010b
    b0 &= ~over; b1 |= over; b2 &= ~over;
}
__device__ __forceinline__ uint32_t pk_masked_sum(unsigned long long b0, unsigned long long b1, unsigned long long b2, unsigned long long b3, unsigned long long m) {
    return (uint32_t)__popcll(b0 & m) + 2u * (uint32_t)__popcll(b1 & m) + 4u * (uint32_t)__popcll(b2 & m) + 8u * (uint32_t)__popcll(b3 & m);
}
// positions of the word starting at wstart that carry bin data (>= pos0; the planes are zero beyond len)
__device__ __forceinline__ unsigned long long pk_valid(int64_t wstart, int64_t p0c) {
    if (wstart >= p0c) return ~0ull;
    return (p0c - wstart >= 64) ? 0ull : ((~0ull) << (p0c - wstart));
}

// the sweep: one 64-position word per lane, PK_TILES tiles per wave (all loads of the wave issued before the first popcount)
__global__ void __launch_bounds__(256) k_tile_summary_packed(const BinChrom* __restrict__ ch, int nchr, int64_t ntilesTotal, const unsigned long long* __restrict__ pos0,
                                                             int clampHits, int wantObs, uint32_t* __restrict__ S, uint32_t* __restrict__ tilePop, uint32_t* __restrict__ tileObs,
                                                             uint32_t* __restrict__ tileTotC, uint32_t* __restrict__ tileTotG, int64_t tile0) {
    const int64_t g0 = tile0 + ((int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) * PK_TILES;
    if (g0 >= ntilesTotal) return;
    const int l = lane_id();
    int c = find_chrom(ch, nchr, g0);
    int64_t tileBase = ch[c].tileBase, tileEnd = tileBase + ch[c].ntiles;
    gptr<const ulonglong2> ref = reinterpret_cast<gptr<const ulonglong2>>(as_global(ch[c].bases));           // global_load, not flat_load (common.hpp)
    gptr<const ulonglong2> hp = reinterpret_cast<gptr<const ulonglong2>>(as_global(ch[c].hits));
    int64_t p0c = (int64_t)pos0[c];
    ulonglong2 r[PK_TILES], ha[PK_TILES], hb[PK_TILES];
    unsigned long long valid[PK_TILES];
#pragma unroll
    for (int t = 0; t < PK_TILES; t++) {
        const int64_t gtile = g0 + t;
        if (gtile < ntilesTotal) {
            while (gtile >= tileEnd) {
                c++; tileBase = ch[c].tileBase; tileEnd = tileBase + ch[c].ntiles; p0c = (int64_t)pos0[c];
                ref = reinterpret_cast<gptr<const ulonglong2>>(as_global(ch[c].bases)); hp = reinterpret_cast<gptr<const ulonglong2>>(as_global(ch[c].hits));
            }
            const int64_t w = ((gtile - tileBase) << 6) + l;
            r[t] = gload_ulonglong2(ref + w); ha[t] = gload_ulonglong2(hp + 2 * w); hb[t] = gload_ulonglong2(hp + 2 * w + 1);
            valid[t] = pk_valid(w << 6, p0c);
        } else { r[t] = make_ulonglong2(0, 0); ha[t] = r[t]; hb[t] = r[t]; valid[t] = 0; }
    }
#pragma unroll
    for (int t = 0; t < PK_TILES; t++) {
        const int64_t gtile = g0 + t;
        if (gtile >= ntilesTotal) break;
        unsigned long long b0 = ha[t].x, b1 = ha[t].y, b2 = hb[t].x; const unsigned long long b3 = hb[t].y;
        const uint32_t obs = (uint32_t)__popcll(b0 | b1 | b2 | b3);
        if (clampHits) pk_clamp10(b0, b1, b2, b3);
        const unsigned long long mv = r[t].x & valid[t];
        const uint32_t keep = (uint32_t)__popcll(r[t].x) | ((uint32_t)__popcll(r[t].y & valid[t]) << 7) | (pk_masked_sum(b0, b1, b2, b3, mv) << 14);
        S[gtile * 64 + l] = keep;
        const uint32_t pg = wave_reduce_add_u32(SUM_POP(keep) | (SUM_GC(keep) << 16));
        const uint32_t ct = wave_reduce_add_u32(SUM_HITS(keep));
        const uint32_t ob = wantObs ? wave_reduce_add_u32(obs) : 0u;
        if (l == 0) { tilePop[gtile] = pg & 0xFFFFu; tileTotG[gtile] = pg >> 16; tileTotC[gtile] = ct; if (wantObs) tileObs[gtile] = ob; }
    }
}

// boundary resolution: one thread per bin — the reference pair (16 B) and the hit planes (32 B) of the word k_bin_close recorded
__global__ void __launch_bounds__(256) k_bin_resolve_packed(const BinChrom* __restrict__ ch, int nchr, const long long* __restrict__ binOffset,
                                                            const unsigned long long* __restrict__ pos0, int clampHits,
                                                            const int32_t* __restrict__ oChr, int32_t* __restrict__ stopIO, uint32_t* __restrict__ locC, uint32_t* __restrict__ locG) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= binOffset[nchr]) return;
    const int32_t rec = stopIO[i];
    const int c = oChr[i];
    const int64_t wstart = (int64_t)(rec & ~63), w = wstart >> 6;
    uint32_t kk = (uint32_t)(rec & 63) + 1u;
    const gptr<const ulonglong2> hp = reinterpret_cast<gptr<const ulonglong2>>(as_global(ch[c].hits));
    const ulonglong2 r = gload_ulonglong2(reinterpret_cast<gptr<const ulonglong2>>(as_global(ch[c].bases)) + w);
    const ulonglong2 ha = gload_ulonglong2(hp + 2 * w), hb = gload_ulonglong2(hp + 2 * w + 1);
    const unsigned long long valid = pk_valid(wstart, (int64_t)pos0[c]);
    uint32_t pos = 0, cnt;                                                   // the kk-th set bit of the possible word closes the bin
    uint64_t m = r.x;
    cnt = __popc((uint32_t)m);           if (kk > cnt) { kk -= cnt; pos += 32; m >>= 32; }
    cnt = __popc((uint32_t)m & 0xFFFFu); if (kk > cnt) { kk -= cnt; pos += 16; m >>= 16; }
    cnt = __popc((uint32_t)m & 0xFFu);   if (kk > cnt) { kk -= cnt; pos += 8; m >>= 8; }
    cnt = __popc((uint32_t)m & 0xFu);    if (kk > cnt) { kk -= cnt; pos += 4; m >>= 4; }
    cnt = __popc((uint32_t)m & 0x3u);    if (kk > cnt) { kk -= cnt; pos += 2; m >>= 2; }
    cnt = (uint32_t)m & 1u;              if (kk > cnt) { pos += 1; }
    const unsigned long long head = valid & ((2ull << pos) - 1ull);          // valid positions <= pos (pos = 63: all of them)
    unsigned long long b0 = ha.x, b1 = ha.y, b2 = hb.x; const unsigned long long b3 = hb.y;
    if (clampHits) pk_clamp10(b0, b1, b2, b3);
    stopIO[i] = (int32_t)(wstart + pos + 1);
    locC[i] += pk_masked_sum(b0, b1, b2, b3, r.x & head);
    locG[i] += (uint32_t)__popcll(r.y & head);
}

// ---------------------------------------------------------------------------------------------- packers on the device (arrays already in HBM: tests, bench, a resident reference)
// one thread per 64-position word; words beyond the chromosome (tile padding) are written as zero
__global__ void __launch_bounds__(256) k_pack_ref(const uint8_t* __restrict__ bases, const uint64_t* __restrict__ mask, int64_t len, int64_t nwordsPadded, ulonglong2* __restrict__ out) {
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= nwordsPadded) return;
    const int64_t p = w << 6;
    unsigned long long m = 0, g = 0;
    if (p < len) {
        m = mask[w];
        if (p + 64 <= len) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint4 v = *reinterpret_cast<const uint4*>(bases + p + 16 * q);
                const unsigned long long b16 = (unsigned long long)(marks_to_bits4(gc_marks4(v.x)) | (marks_to_bits4(gc_marks4(v.y)) << 4) | (marks_to_bits4(gc_marks4(v.z)) << 8) |
                                                                     (marks_to_bits4(gc_marks4(v.w)) << 12));
                g |= b16 << (16 * q);
            }
        } else {
            m &= (~0ull) >> (64 - (len - p));
            for (int i = 0; p + i < len; i++) { const uint8_t b = bases[p + i] | 0x20; if (b == 'c' || b == 'g') g |= 1ull << i; }
        }
    }
    out[w] = make_ulonglong2(m, g);
}
__global__ void __launch_bounds__(256) k_pack_hits(const uint8_t* __restrict__ hits, int64_t len, int64_t nwordsPadded, ulonglong2* __restrict__ out, unsigned long long* __restrict__ saturated) {
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= nwordsPadded) return;
    const int64_t p = w << 6;
    unsigned long long b[4] = {0, 0, 0, 0};
    uint32_t sat = 0;
    for (int i = 0; i < 64 && p + i < len; i += 16) {
        uint32_t hw[4] = {0, 0, 0, 0};
        if (p + i + 16 <= len) { const uint4 v = *reinterpret_cast<const uint4*>(hits + p + i); hw[0] = v.x; hw[1] = v.y; hw[2] = v.z; hw[3] = v.w; }
        else for (int j = 0; p + i + j < len; j++) hw[j >> 2] |= (uint32_t)hits[p + i + j] << (8 * (j & 3));
#pragma unroll
        for (int j = 0; j < 16; j++) {
            uint32_t h = (hw[j >> 2] >> (8 * (j & 3))) & 0xFFu;
            if (h > 15u) { h = 15u; sat++; }
#pragma unroll
            for (int k = 0; k < 4; k++) b[k] |= (unsigned long long)((h >> k) & 1u) << (i + j);
        }
    }
    out[2 * w] = make_ulonglong2(b[0], b[1]); out[2 * w + 1] = make_ulonglong2(b[2], b[3]);
    if (sat) atomicAdd(saturated, (unsigned long long)sat);
}

// ---------------------------------------------------------------------------------------------- two-bit wire format of the hit planes
// At WGS depth a position with four hits or more is rare (6.6e-5 at 60x), so b2 and b3 are almost entirely zero.  Over PCIe the hit planes travel as
//   lo        per 64 positions {u64 b0, b1}                                   16 B
//   header    per tile (64 words) {u64 xmask, u64 xoff}: bit w of xmask = word w of the tile has a non-zero b2 or b3; xoff = index of the tile's first entry in `extras`
//   extras    {u64 b2, b3} of the words that have one, in word order
// = 0.25 B/base + a few MB instead of 0.5 B/base, and are expanded into the four planes on the device (one lane per word) before the sweep reads them: the expansion
// runs on the copy stream right behind each chromosome's transfer, so everything downstream sees the planes of canvas_bin_sample_packed.
__global__ void __launch_bounds__(256) k_expand_hits2(const ulonglong2* __restrict__ lo, const ulonglong2* __restrict__ hdr, const ulonglong2* __restrict__ extras, int64_t nwords, ulonglong2* __restrict__ planes) {
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= nwords) return;
    const ulonglong2 h = hdr[w >> 6];
    const int b = (int)(w & 63);
    ulonglong2 hi = make_ulonglong2(0, 0);
    if ((h.x >> b) & 1ull) hi = extras[h.y + (unsigned long long)__popcll(h.x & ((1ull << b) - 1ull))];
    planes[2 * w] = lo[w];
    planes[2 * w + 1] = hi;
}
