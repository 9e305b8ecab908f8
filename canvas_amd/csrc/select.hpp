// Exact order statistics on the GPU: multi-segment, multi-query MSB radix select (8-bit digits).
// Keys are order-preserving unsigned images of float / double / int32 values, so the selected key IS the k-th smallest
// value bit-for-bit (no floating reduction anywhere => matches a CPU sort exactly).
//
//   keys      contiguous array, partitioned in segments; a tile (<= 4096 keys) never crosses a segment
//   query q   (segLo..segHi, k): k-th smallest (0-based) of the union of segments segLo..segHi
//             (per-GC medians use segLo == segHi == gc; the genome-wide median uses 0..100 over the same array)
//   pass p    histogram of digit p among the keys that match the query's prefix so far (LDS-privatised, then
//             flushed with global atomics), then k_select_pick narrows every query by one digit.
// After KEYBITS/8 passes qprefix[q] is the answer.
#pragma once
#include "common.hpp"

#define SEL_TILE 4096      // keys per workgroup: LDS rows are zeroed and flushed once per tile, the keys go through in chunks of SEL_CHUNK
                           // (measured on the WGS CanvasClean stage: 2048 -> 0.79 ms, 4096 -> 0.78 ms, 8192 -> 0.82 ms, 16384 -> 1.00 ms)
#define SEL_CHUNK 4096
#define SEL_MAXQ 16   // max queries that can apply to one tile
#define SEL_REP 16    // replicas of the global histogram rows: tiles flush into replica (tile % SEL_REP), the pick sums them

struct SelTile { int32_t seg; int64_t begin; int64_t end; };
struct SelSegQ { int32_t nq; int32_t q[SEL_MAXQ]; };

__device__ __forceinline__ uint32_t key_of_float(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float float_of_key(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(u);
}
__device__ __forceinline__ unsigned long long key_of_double(double d) {
    unsigned long long u = (unsigned long long)__double_as_longlong(d);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
static inline float host_float_of_key(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    float f; memcpy(&f, &u, 4); return f;
}
static inline double host_double_of_key(unsigned long long k) {
    unsigned long long u = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    double d; memcpy(&d, &u, 8); return d;
}

// LDS contention control: (i) queries of one tile that still share a prefix (all of them in the first pass; the k / k+1 pair of
// a median until the last digits) share ONE LDS histogram row that is replicated at flush time; (ii) digits are concentrated
// (exponent bytes of similar counts, zero low bytes of integer-valued floats), so a wave first aggregates up to four distinct
// digits with ballots — one LDS atomic per distinct digit — and only the lanes left over issue their own atomics.
template <typename K>
__device__ __forceinline__ void select_hist_body(const K* __restrict__ keys, const SelTile* __restrict__ tiles, const SelSegQ* __restrict__ segq,
                                                 const unsigned long long* __restrict__ qprefix, int shift, int firstPass,
                                                 uint32_t* __restrict__ hist /* [SEL_REP][nq][256] */, int nq, const uint32_t* __restrict__ hdr) {
    __shared__ uint32_t lh[SEL_MAXQ * 256];
    __shared__ unsigned long long lpre[SEL_MAXQ];
    __shared__ int srep[SEL_MAXQ], suniq[SEL_MAXQ], snu;
    if (hdr && blockIdx.x >= hdr[0]) return;              // device-built problem (clean_fast.hpp): hdr = {tiles, queries}; the grid is an upper bound
    const SelTile T = tiles[blockIdx.x];
    const SelSegQ Q = segq[T.seg];
    if (Q.nq == 0) return;
    for (int i = threadIdx.x; i < Q.nq * 256; i += 256) lh[i] = 0;
    if (threadIdx.x < Q.nq) lpre[threadIdx.x] = firstPass ? 0ull : qprefix[Q.q[threadIdx.x]];
    __syncthreads();
    if (threadIdx.x == 0) {
        int nu = 0;
        for (int q = 0; q < Q.nq; q++) {
            int rep = q;
            for (int u = 0; u < nu; u++) if (lpre[suniq[u]] == lpre[q]) { rep = suniq[u]; break; }
            srep[q] = rep;
            if (rep == q) suniq[nu++] = q;
        }
        snu = nu;
    }
    __syncthreads();
    const int nu = snu;
    const int lane = threadIdx.x & 63;
    const int sh2 = firstPass ? 0 : shift + 8;
    bool agg = true;                                      // wave-uniform: is ballot aggregation paying off in this tile?
    for (int64_t cbeg = T.begin; cbeg < T.end; cbeg += SEL_CHUNK) {
    // all 16 keys of this thread are loaded up front (independent loads in flight), then binned
    K kreg[SEL_CHUNK / 256];
#pragma unroll
    for (int r = 0; r < SEL_CHUNK / 256; r++) { const int64_t i = cbeg + threadIdx.x + (int64_t)r * 256; kreg[r] = i < T.end ? keys[i] : (K)0; }
    if (nu == 1) {
        // one histogram row for the whole tile (every first pass, every single-query select): prefix and row index in registers, no LDS
        // reads and no loop over rows inside the key loop
        const int q = suniq[0];
        const unsigned long long pre = lpre[q];
#pragma unroll
        for (int r = 0; r < SEL_CHUNK / 256; r++) {
            const bool in = cbeg + threadIdx.x + (int64_t)r * 256 < T.end;
            const K key = kreg[r];
            const uint32_t d = (uint32_t)(key >> shift) & 255u;
            const bool m = in && (firstPass || (unsigned long long)(key >> sh2) == pre);
            if (agg) {
                unsigned long long todo = __ballot(m);
                for (int it = 0; it < 4 && todo; it++) {
                    const int leader = __builtin_ctzll(todo);
                    const uint32_t dl = (uint32_t)__builtin_amdgcn_readlane((int)d, leader);
                    const unsigned long long same = __ballot(m && d == dl) & todo;
                    if (lane == leader) atomicAdd(&lh[q * 256 + dl], (uint32_t)__builtin_popcountll(same));
                    todo &= ~same;
                }
                if ((todo >> lane) & 1ull) atomicAdd(&lh[q * 256 + d], 1u);
                if (__builtin_popcountll(todo) > 24) agg = false;
            } else if (m) atomicAdd(&lh[q * 256 + d], 1u);
        }
    } else
#pragma unroll
    for (int r = 0; r < SEL_CHUNK / 256; r++) {
        // several rows (never in the first pass): the unique prefixes are pairwise different, so a key matches at most ONE of them — find that row, then one
        // aggregation round on (row, digit) instead of one round per row
        const bool in = cbeg + threadIdx.x + (int64_t)r * 256 < T.end;
        const K key = kreg[r];
        const unsigned long long hi = (unsigned long long)(key >> sh2);
        int row = -1;
        for (int u = 0; u < nu; u++) { const int q = suniq[u]; if (hi == lpre[q]) row = q; }
        const bool m = in && row >= 0;
        const uint32_t cd = (uint32_t)(row < 0 ? 0 : row) * 256u + ((uint32_t)(key >> shift) & 255u);
        if (agg) {
            unsigned long long todo = __ballot(m);
            for (int it = 0; it < 4 && todo; it++) {
                const int leader = __builtin_ctzll(todo);
                const uint32_t cl = (uint32_t)__builtin_amdgcn_readlane((int)cd, leader);
                const unsigned long long same = __ballot(m && cd == cl) & todo;
                if (lane == leader) atomicAdd(&lh[cl], (uint32_t)__builtin_popcountll(same));
                todo &= ~same;
            }
            if ((todo >> lane) & 1ull) atomicAdd(&lh[cd], 1u);
            if (__builtin_popcountll(todo) > 24) agg = false;       // digits are spread (low mantissa bytes): plain atomics from here on
        } else if (m) atomicAdd(&lh[cd], 1u);
    }
    }   // chunks of the tile
    __syncthreads();
    for (int i = threadIdx.x; i < Q.nq * 256; i += 256) {
        uint32_t v = lh[srep[i >> 8] * 256 + (i & 255)];
        if (v) atomicAdd(&hist[((size_t)(blockIdx.x % SEL_REP) * nq + Q.q[i >> 8]) * 256 + (i & 255)], v);
    }
}
template <typename K>
__global__ void __launch_bounds__(256) k_select_hist(const K* __restrict__ keys, const SelTile* __restrict__ tiles, const SelSegQ* __restrict__ segq,
                                                     const unsigned long long* __restrict__ qprefix, int shift, int firstPass,
                                                     uint32_t* __restrict__ hist /* [SEL_REP][nq][256] */, int nq, const uint32_t* __restrict__ hdr = nullptr) {
    select_hist_body<K>(keys, tiles, segq, qprefix, shift, firstPass, hist, nq, hdr);
}

// one wave per query: locate the digit holding rank k, narrow (prefix, k), clear the histogram row
__device__ __forceinline__ void select_pick_body(uint32_t* __restrict__ hist, unsigned long long* __restrict__ qprefix,
                                                 unsigned long long* __restrict__ qk, int nq, int firstPass, const uint32_t* __restrict__ hdr) {
    const int q = blockIdx.x;
    if (q >= nq || (hdr && (uint32_t)q >= hdr[1])) return;
    const int l = threadIdx.x;
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
#pragma unroll
    for (int r = 0; r < SEL_REP; r++) {
        uint4* h = reinterpret_cast<uint4*>(hist + ((size_t)r * nq + q) * 256) + l;
        const uint4 v = *h;
        c0 += v.x; c1 += v.y; c2 += v.z; c3 += v.w;
        *h = make_uint4(0u, 0u, 0u, 0u);                  // cleared for the next pass
    }
    uint32_t s = c0 + c1 + c2 + c3;
    uint32_t inc = wave_inclusive_scan_u32(s);
    uint32_t ex = inc - s;
    unsigned long long k = qk[q];
    // the lane whose range [ex, inc) contains k
    if (k >= ex && k < inc) {
        uint32_t r = (uint32_t)(k - ex), d;
        if (r < c0) { d = 0; }
        else if (r < c0 + c1) { d = 1; r -= c0; }
        else if (r < c0 + c1 + c2) { d = 2; r -= c0 + c1; }
        else { d = 3; r -= c0 + c1 + c2; }
        qprefix[q] = ((firstPass ? 0ull : qprefix[q]) << 8) | (unsigned long long)(4 * l + d);   // (the prefix array is not cleared between calls)
        qk[q] = r;
    }
}
static __global__ void __launch_bounds__(64) k_select_pick(uint32_t* __restrict__ hist, unsigned long long* __restrict__ qprefix,
                                                    unsigned long long* __restrict__ qk, int nq, int firstPass, const uint32_t* __restrict__ hdr = nullptr) {
    select_pick_body(hist, qprefix, qk, nq, firstPass, hdr);
}

// Exact order statistics inside ONE workgroup: 8 MSB-radix passes over keyOf(i), i in [lo, hi), for two ranks at once
// (the k / k+1 pair of an even-length median).  Used where the segments are small (per-chromosome window SDs).
template <typename F>
__device__ __forceinline__ void wg_select2(F keyOf, int64_t lo, int64_t hi, unsigned long long rank0, unsigned long long rank1, uint32_t (*sH)[256],
                                           unsigned long long* sPre, unsigned long long* sK) {
    const int tid = threadIdx.x, nt = blockDim.x;
    if (tid == 0) { sPre[0] = 0; sPre[1] = 0; sK[0] = rank0; sK[1] = rank1; }
    for (int shift = 56; shift >= 0; shift -= 8) {
        for (int i = tid; i < 512; i += nt) sH[i >> 8][i & 255] = 0;
        __syncthreads();
        const unsigned long long p0 = sPre[0], p1 = sPre[1];
        const bool same = p0 == p1;
        for (int64_t i = lo + tid; i < hi; i += nt) {
            const unsigned long long key = keyOf(i);
            const uint32_t d = (uint32_t)(key >> shift) & 255u;
            const unsigned long long hiPart = shift == 56 ? 0ull : key >> (shift + 8);
            if (hiPart == p0) atomicAdd(&sH[0][d], 1u);
            if (!same && hiPart == p1) atomicAdd(&sH[1][d], 1u);
        }
        __syncthreads();
        const int w = tid >> 6, l = tid & 63;
        if (w < 2) {
            const uint32_t* h = sH[same ? 0 : w];
            uint32_t c0 = h[4 * l], c1 = h[4 * l + 1], c2 = h[4 * l + 2], c3 = h[4 * l + 3];
            uint32_t sum = c0 + c1 + c2 + c3;
            uint32_t inc = wave_inclusive_scan_u32(sum), ex = inc - sum;
            const unsigned long long k = sK[w];
            if (k >= ex && k < inc) {
                uint32_t r = (uint32_t)(k - ex), d;
                if (r < c0) { d = 0; } else if (r < c0 + c1) { d = 1; r -= c0; } else if (r < c0 + c1 + c2) { d = 2; r -= c0 + c1; } else { d = 3; r -= c0 + c1 + c2; }
                sPre[w] = (sPre[w] << 8) | (unsigned long long)(4 * l + d);
                sK[w] = r;
            }
        }
        __syncthreads();
    }
}
__device__ __forceinline__ double double_of_key(unsigned long long k) {
    unsigned long long u = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double((long long)u);
}

// ---- host driver ----------------------------------------------------------------------------------------------
struct SelQuery { int32_t segLo, segHi; int64_t k; };

// Runs all queries; results (keys as u64) are returned in `out`.  segOff has nseg+1 entries (host).  One sync at the end.
template <typename K>
static int32_t radix_select(canvas_ctx* ctx, const K* d_keys, int nseg, const std::vector<int64_t>& segOff,
                            const std::vector<SelQuery>& queries, std::vector<unsigned long long>& out, const unsigned long long** d_results = nullptr) {
    const int nq = (int)queries.size();
    out.assign(nq, 0);
    if (nq == 0) return CANVAS_OK;
    std::vector<SelTile> tiles;
    for (int s = 0; s < nseg; s++)
        for (int64_t b = segOff[s]; b < segOff[s + 1]; b += SEL_TILE) tiles.push_back({s, b, std::min<int64_t>(b + SEL_TILE, segOff[s + 1])});
    std::vector<SelSegQ> segq(nseg);
    for (int s = 0; s < nseg; s++) segq[s].nq = 0;
    for (int q = 0; q < nq; q++)
        for (int s = queries[q].segLo; s <= queries[q].segHi; s++) {
            if (segq[s].nq >= SEL_MAXQ) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "radix_select: too many queries per segment");
            segq[s].q[segq[s].nq++] = q;
        }
    if (tiles.empty()) return CANVAS_OK;
    // persistent scratch of the context: one pinned blob [tiles | segq | ranks | results] <-> one device blob [tiles | segq | ranks | prefix | hist]
    // (one H2D copy and one D2H copy per call; every call ends with a synchronisation, so the blobs are reused from offset 0)
    auto al = [](size_t v) { return (v + 255) & ~size_t(255); };
    const size_t oTiles = 0, oSegq = al(tiles.size() * sizeof(SelTile)), oK = oSegq + al(segq.size() * sizeof(SelSegQ)), oPrefix = oK + al((size_t)nq * 8),
                 devBytes = oPrefix + al((size_t)nq * 8), histBytes = (size_t)nq * 1024 * SEL_REP, pinBytes = oPrefix + al((size_t)nq * 8);
    if (devBytes > ctx->sel_ws_bytes) {
        if (ctx->sel_ws) { CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); CANVAS_HIP_TRY(ctx, hipFree(ctx->sel_ws)); ctx->sel_ws = nullptr; ctx->sel_ws_bytes = 0; }
        CANVAS_HIP_TRY(ctx, hipMalloc(&ctx->sel_ws, devBytes * 2)); ctx->sel_ws_bytes = devBytes * 2;
    }
    if (histBytes > ctx->sel_hist_bytes) {
        // the histograms live in their own buffer and are zero at the start of every call: cleared here once, and k_select_pick clears every
        // row it has read (all replicas, after the last pass too), so there is no per-call memset
        if (ctx->sel_hist) { CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); CANVAS_HIP_TRY(ctx, hipFree(ctx->sel_hist)); ctx->sel_hist = nullptr; ctx->sel_hist_bytes = 0; }
        CANVAS_HIP_TRY(ctx, hipMalloc(&ctx->sel_hist, histBytes * 2)); ctx->sel_hist_bytes = histBytes * 2;
        CANVAS_HIP_TRY(ctx, hipMemsetAsync(ctx->sel_hist, 0, ctx->sel_hist_bytes, ctx->stream));
    }
    if (pinBytes > ctx->sel_pin_bytes) {
        if (ctx->sel_pin) { CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); CANVAS_HIP_TRY(ctx, hipHostFree(ctx->sel_pin)); ctx->sel_pin = nullptr; ctx->sel_pin_bytes = 0; }
        CANVAS_HIP_TRY(ctx, hipHostMalloc(&ctx->sel_pin, pinBytes * 2, hipHostMallocDefault)); ctx->sel_pin_bytes = pinBytes * 2;
    }
    char* d = (char*)ctx->sel_ws; char* h = (char*)ctx->sel_pin;
    SelTile* dTiles = (SelTile*)(d + oTiles); SelSegQ* dSegq = (SelSegQ*)(d + oSegq);
    unsigned long long* dK = (unsigned long long*)(d + oK); unsigned long long* dPrefix = (unsigned long long*)(d + oPrefix);
    uint32_t* dHist = (uint32_t*)ctx->sel_hist;
    memcpy(h + oTiles, tiles.data(), tiles.size() * sizeof(SelTile));
    memcpy(h + oSegq, segq.data(), segq.size() * sizeof(SelSegQ));
    unsigned long long* hk = (unsigned long long*)(h + oK);
    for (int q = 0; q < nq; q++) hk[q] = (unsigned long long)queries[q].k;
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(d, h, oPrefix, hipMemcpyHostToDevice, ctx->stream));
    const int bits = (int)sizeof(K) * 8;
    for (int shift = bits - 8; shift >= 0; shift -= 8) {
        hipLaunchKernelGGL((k_select_hist<K>), dim3((unsigned)tiles.size()), dim3(256), 0, ctx->stream, d_keys, dTiles, dSegq, dPrefix, shift,
                           shift == bits - 8 ? 1 : 0, dHist, nq);
        hipLaunchKernelGGL(k_select_pick, dim3(nq), dim3(64), 0, ctx->stream, dHist, dPrefix, dK, nq, shift == bits - 8 ? 1 : 0);
    }
    if (d_results) {        // the consumer is a kernel: results stay on the device, no synchronisation.  The caller must synchronise the stream
        *d_results = dPrefix;   // before the next radix_select (the pinned staging blob is reused from offset 0)
        return CANVAS_OK;
    }
    unsigned long long* hres = (unsigned long long*)(h + oPrefix);
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(hres, dPrefix, (size_t)nq * 8, hipMemcpyDeviceToHost, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    CANVAS_HIP_TRY(ctx, hipGetLastError());
    for (int q = 0; q < nq; q++) out[q] = hres[q];
    return CANVAS_OK;
}
