// Synthetic per-base inputs (bases / possible-alignment mask / hit counts) generated directly in HBM for bench.py and
// the full-size property tests: a 60x whole genome is 6.6 GB and cannot be shipped to the GPU box.
// Pure counter-based integer hashing, mirrored bit-for-bit by canvas_amd/synth.py (numpy) so that the CPU oracle
// can be run on exactly the same bytes.  Not part of the product ABI (separate libcanvas_synth.so).
#include <hip/hip_runtime.h>
#include <cstdint>

__host__ __device__ inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__host__ __device__ inline uint32_t H(uint32_t seed, uint32_t chr, uint32_t stream, uint32_t p) {
    return mix32(mix32(p + 0x9e3779b9u * (stream + 1u)) ^ (seed * 0x85ebca6bu + chr * 0xc2b2ae35u + stream));
}

struct SynthParams {
    uint32_t seed, chr;
    int64_t len;
    int64_t gap0_end;            // [0, gap0_end) is 'n'
    int64_t gap1_start, gap1_end;  // centromere-like 'n' block
    uint32_t baseCN;             // 2 for autosomes
    uint32_t hitSeed;            // seed of the copy-number segments and the hit draws (streams 6..8): == seed for a single sample; a tumour / normal pair
                                 // shares `seed` (same reference bases and mask) and differs in hitSeed
};

// thr: [5 CN levels][16 gc levels][8] cumulative Poisson thresholds as uint32 (hits = #thr <= u)
// bases / mask may be NULL (second sample over the same reference); fraglen (may be NULL) = Int16 fragment length at positions with a hit, 0 elsewhere:
// 143 + 60 * (sum of four hash bytes) / 148, i.e. roughly N(350, 60) clipped to [143, 556] (CanvasBin's GCContentWeighted input)
__global__ void __launch_bounds__(256) k_synth(SynthParams P, const uint32_t* __restrict__ thr, uint8_t* __restrict__ bases,
                                               uint8_t* __restrict__ hits, uint64_t* __restrict__ mask, int16_t* __restrict__ fraglen) {
    // one thread per 64 positions (one mask word)
    int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t p0 = w * 64;
    if (p0 >= P.len) return;
    uint64_t mw = 0;
    for (int i = 0; i < 64; i++) {
        int64_t p = p0 + i;
        if (p >= P.len) break;
        uint32_t pp = (uint32_t)p;
        bool gap = p < P.gap0_end || (p >= P.gap1_start && p < P.gap1_end);
        uint8_t b, h = 0;
        bool m = false;
        if (gap) b = 'n';
        else {
            uint32_t cell1k = pp >> 10, off = pp & 1023u;
            uint32_t hc = H(P.seed, P.chr, 1, cell1k);
            m = true;
            if ((hc % 100u) < 22u) {
                uint32_t a = (hc >> 8) & 511u, ln = 300u + ((hc >> 17) % 724u);
                if (off >= a && off < a + ln) m = false;
            }
            uint32_t hp = H(P.seed, P.chr, 4, pp);
            if ((hp % 100u) < 3u) m = false;
            // GC level per 4 kb cell: slow component (64 kb) + wobble
            uint32_t lvl = ((H(P.seed, P.chr, 2, pp >> 16) & 7u) + (H(P.seed, P.chr, 5, pp >> 12) & 7u) + 1u) & 15u;
            uint32_t gcfrac = 26u + 2u * lvl;   // percent
            uint32_t ub = H(P.seed, P.chr, 3, pp);
            bool isgc = ((ub & 0xFFFFu) * 100u) < (gcfrac << 16);
            bool which = (ub >> 16) & 1u;
            b = isgc ? (which ? 'G' : 'C') : (which ? 'A' : 'T');
            if (!m) b |= 0x20;  // lowercase = not a unique 35-mer start
            if (m) {
                uint32_t hcn = H(P.hitSeed, P.chr, 6, pp >> 20) % 1000u;
                uint32_t cn = P.baseCN;
                if (hcn < 15u) cn = P.baseCN - 1u; else if (hcn < 30u) cn = P.baseCN + 1u; else if (hcn < 33u) cn = 0u; else if (hcn < 36u) cn = P.baseCN + 2u;
                if (cn > 4u) cn = 4u;
                uint32_t u = H(P.hitSeed, P.chr, 7, pp);
                const uint32_t* t = thr + ((cn * 16u + lvl) << 3);
                uint32_t k = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) k += (u >= t[j]) ? 1u : 0u;
                h = (uint8_t)k;
            }
        }
        if (bases) bases[p] = b;
        hits[p] = h;
        if (fraglen) {
            int16_t fl = 0;
            if (h) { const uint32_t uf = H(P.hitSeed, P.chr, 8, pp); const uint32_t ssum = (uf & 255u) + ((uf >> 8) & 255u) + ((uf >> 16) & 255u) + (uf >> 24); fl = (int16_t)(143u + ssum * 60u / 148u); }
            fraglen[p] = fl;
        }
        if (m) mw |= 1ull << i;
    }
    if (mask) mask[w] = mw;
}

#ifndef CANVAS_SRC_HASH
#define CANVAS_SRC_HASH "unhashed-build-0000000000000000"
#endif
__attribute__((used)) static const char src_hash_marker[] = "CANVAS_SRC_HASH=" CANVAS_SRC_HASH;
extern "C" int synth_generate(uint32_t seed, uint32_t chr, int64_t len, int64_t gap0_end, int64_t gap1_start, int64_t gap1_end, uint32_t baseCN,
                              const uint32_t* d_thr, uint8_t* d_bases, uint8_t* d_hits, uint64_t* d_mask, void* stream) {
    SynthParams P{seed, chr, len, gap0_end, gap1_start, gap1_end, baseCN, seed};
    int64_t words = (len + 63) / 64;
    hipLaunchKernelGGL(k_synth, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P, d_thr, d_bases, d_hits, d_mask, (int16_t*)nullptr);
    return (int)hipGetLastError();
}
// a further sample over the same reference (tumour / normal pairs, BASELINE configs[4]): hits (and optionally fragment lengths) from hit_seed
extern "C" int synth_generate_sample(uint32_t seed, uint32_t hit_seed, uint32_t chr, int64_t len, int64_t gap0_end, int64_t gap1_start, int64_t gap1_end, uint32_t baseCN,
                                     const uint32_t* d_thr, uint8_t* d_bases, uint8_t* d_hits, uint64_t* d_mask, int16_t* d_fraglen, void* stream) {
    SynthParams P{seed, chr, len, gap0_end, gap1_start, gap1_end, baseCN, hit_seed};
    int64_t words = (len + 63) / 64;
    hipLaunchKernelGGL(k_synth, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P, d_thr, d_bases, d_hits, d_mask, d_fraglen);
    return (int)hipGetLastError();
}
