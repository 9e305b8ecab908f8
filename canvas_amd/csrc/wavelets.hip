// CanvasPartition -m Wavelets (the reference's default method): unbalanced Haar segmentation.
//   reference: CanvasPartition/WaveletSegmentation.cs:19-428 (GetInnerProdIter, GetInnerProdMax, FindBestUnbalancedHaarDecomposition, HardThresh,
//              GetReconstructedVector, GetSegments, GetBreakpointsAfterHealingBadSplits, RefineSegments, HaarWavelets),
//              WaveletsRunner.cs:52-150 (Run / LaunchWavelets), Segmentation.cs:297-429 (coverage variability inputs).
//
// Where the time goes in the reference: every node of the top-down tree runs GetInnerProdIter over its stretch of the chromosome,
// a pair of first-order recurrences
//     I+[m] = I+[m-1] * f_m + x_m * g_m          I-[m] = I-[m-1] / f_m - x_m / h_m          (f, g, h: square roots of ratios of n and m)
// behind a sequential sum of the stretch, and takes the first arg-max of |I+ - I-|.  A level of the tree costs one sweep of the
// chromosome; a whole-genome sample needs ~35 sweeps.  The recurrences are rounded at every step, so they cannot be re-associated:
// the order of operations below is exactly the reference's.
//
// Device mapping (one launch set per tree LEVEL, all chromosomes and all nodes of the level at once):
//   * k_wv_coeff: the coefficients f, g*x, x/h (6 divisions + 3 square roots per element) carry no dependence and are computed for every
//     element of the long nodes in parallel, together with the refined reciprocal of f (see below);
//   * k_wv_chain_long: one wave per long node and nothing but the recurrences: the operands of 7 steps at a time are prefetched through
//     LDS, every lane runs the same chain on broadcast operands (6 FP64 operations per step, issued back to back: the chain IS the
//     critical path of the method), and the state in front of every chunk of 56 steps is written out;
//   * the division I-[m-1] / f_m inside the chain is evaluated as the hardware's own division sequence with its operand-independent
//     half hoisted out (r = refined reciprocal of f; t = a*r; q = fma(fma(-f, t, a), r, t)): 3 dependent operations instead of ~15;
//   * k_wv_chunks: one LANE per chunk recomputes the chunk from its checkpoint with plain IEEE divisions and must arrive at the next
//     checkpoint bit for bit.  Checkpoint 0 is exact by construction, so by induction every checkpoint and every value is exact — or the
//     node is flagged and recomputed with IEEE divisions in the chain; results never depend on the shortcut.  The same lanes take the
//     first arg-max of their steps, k_wv_reduce combines them per node;
//   * k_wv_subtree: a node of at most WV_LONG elements leaves the level loop together with its whole subtree: one lane walks it depth-first
//     (millions of small nodes in the deep levels never reach the host), on a second stream next to the long chains.
// The host keeps the tree (start / breakpoint / end, coefficient per node), applies HardThresh, rebuilds the breakpoints and runs the
// two median-based clean-up passes (GetBreakpointsAfterHealingBadSplits, RefineSegments), which are sequential decisions over a
// few hundred breakpoints; the coverage-variability inputs (windowed MAD / median, factor-of-three CMADs) are order statistics of
// small windows and run on a few host threads.
#include "common.hpp"
#include <algorithm>
#include <cmath>
#include <functional>
#include <limits>
#include <map>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#define WV_LONG_MAX 256   // (the subtree walker packs positions relative to its root into 9 bits)
#define WV_LONG_DEFAULT 64 // nodes longer than this stay in the level loop; shorter ones leave it with their whole subtree, one lane each (CANVAS_WV_LONG = 8 .. 256 overrides).
                          // A lane walks its subtree node by node with the reference's sequential loop, and an unbalanced tree over L positions costs up to L^2 / 2 steps:
                          // with 256 the slowest lanes of a WGS sample kept k_wv_subtree busy for 20 ms next to (and on the same SIMDs as) the exact chains
#define WV_CS 56          // steps per chunk of a long node (a checkpoint of the chain state is kept per chunk)
#define WV_PB 7           // steps per operand-prefetch block: 2 LDS reads per step and at most 15 outstanding for s_waitcnt to tell apart

struct WvNode { int32_t start; int32_t len; };            // start = index into the concatenated coverage, len >= 2
struct WvOut { double coef; int32_t ind; int32_t flag; };  // ind = GetInnerProdMax (1-based inside the node); flag: shortcut mismatch
struct WvOps { double f, r, c, d; };                       // per step: factor, its refined reciprocal, x*g, x/h
struct WvCk { double p, q; };                              // chain state in front of a chunk
struct WvBest { double val; int32_t idx; int32_t pad; };   // first arg-max of |I+ - I-| inside a chunk
struct WvHead { double mean, ip0; };                       // per long node: mean of the stretch, I+[0] - I-[0]

__device__ __forceinline__ double wv_factor(long long n, long long m) { return sqrt((double)(n - m - 1) * (double)m / (double)(m + 1) / (double)(n - m)); }
__device__ __forceinline__ double wv_g(long long n, long long m) { return sqrt(1.0 / (double)(m + 1) - 1.0 / (double)n); }
__device__ __forceinline__ double wv_h(long long n, long long m) { return sqrt(((double)n * (double)n / (double)(m + 1)) - (double)n); }
// the operand-independent half of the FP64 division sequence: v_rcp_f64 + two Newton steps
__device__ __forceinline__ double wv_rcp_refined(double f) {
    double r = __builtin_amdgcn_rcp(f);
    double e = __builtin_fma(-f, r, 1.0); r = __builtin_fma(r, e, r);
    e = __builtin_fma(-f, r, 1.0); r = __builtin_fma(r, e, r);
    return r;
}
// chunk g of the level -> index into the long-node list (base[k] <= g < base[k+1])
__device__ __forceinline__ int wv_find(const int32_t* __restrict__ base, int nLong, int g) {
    int lo = 0, hi = nLong - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (base[mid] <= g) lo = mid; else hi = mid - 1; }
    return lo;
}

// one block (64 lanes) per chunk, one lane per step
__global__ void __launch_bounds__(64) k_wv_coeff(const WvNode* __restrict__ nodes, const int32_t* __restrict__ list, const int32_t* __restrict__ base, int nLong,
                                                 const double* __restrict__ X, WvOps* __restrict__ ops, const long long* __restrict__ opsOff, const int32_t* __restrict__ lim) {
    const int g = blockIdx.x;
    const int k = wv_find(base, nLong, g);
    const WvNode nd = nodes[list[k]];
    const long long n = nd.len, m = 1 + (long long)(g - base[k]) * WV_CS + threadIdx.x;
    if (threadIdx.x >= WV_CS || m > (lim ? min<long long>(n - 2, lim[k]) : n - 2)) return;      // (lim: the chain of node k stops after step lim[k] — its arg-max is known)
    const size_t p = (size_t)nd.start + (size_t)m;
    const double x = X[p];
    WvOps o;
    o.f = wv_factor(n, m); o.r = wv_rcp_refined(o.f);
    o.c = x * wv_g(n, m);
    o.d = x / wv_h(n, m);
    ops[(opsOff ? (size_t)opsOff[k] : (size_t)nd.start) + (size_t)m] = o;      // (opsOff: a slice of its own per node, for lists whose nodes overlap in position)
}

// One step of both recurrences on uniform operands (every lane computes the same values).
template <bool FAST>
__device__ __forceinline__ void wv_step(double& p, double& q, double f, double r, double c, double d) {
    const double pn = p * f + c;
    double qd;
    if (FAST) { const double t = q * r; qd = __builtin_fma(__builtin_fma(-f, t, q), r, t); }
    else qd = q / f;
    q = qd - d;
    p = pn;
}

// The critical path of the method: one wave per long node, nothing but the two recurrences (6 dependent-issue FP64 operations per
// step) behind the sequential sum.  The state in front of every chunk of WV_CS steps is written out; k_wv_chunks recomputes every
// chunk from its checkpoint with IEEE divisions in parallel, which both proves the checkpoints (by induction over the chunks) and
// yields the arg-max, so neither the shortcut division nor the arg-max bookkeeping sits on the chain.
template <bool FAST>
__global__ void __launch_bounds__(64) k_wv_chain_long(const WvNode* __restrict__ nodes, const int32_t* __restrict__ list, const int32_t* __restrict__ base,
                                                      const double* __restrict__ X, const WvOps* __restrict__ ops, WvCk* __restrict__ ck, WvHead* __restrict__ head, const long long* __restrict__ opsOff, const int32_t* __restrict__ lim) {
    __shared__ double4 sO[2][64];              // [buf][step] = (f, r, c, d); the sum pass uses sO[buf][s].x
    __builtin_amdgcn_s_setprio(3);             // the chain is a string of dependent FP64 operations (9-10 cycles each, tools/dp_latency.hip): whatever shares the SIMD issues behind it
    const int k = blockIdx.x;
    const WvNode nd = nodes[list[k]];
    const long long n = nd.len;
    const int l = threadIdx.x;
    const double* __restrict__ x = X + nd.start;
    // ---- sumX = x[1] + ... + x[n-1], left to right (WaveletSegmentation.cs:26-30); LDS operations of one wave execute in order
    double sum = 0.0;
    {
        int buf = 0;
        double nxt = (1 + l < n) ? x[1 + l] : 0.0;
        for (long long i0 = 1; i0 < n; i0 += 64, buf ^= 1) {
            sO[buf][l].x = nxt;
            __builtin_amdgcn_wave_barrier();
            { const long long t = i0 + 64 + l; nxt = t < n ? x[t] : 0.0; }
            const long long cnt = n - i0;
            if (cnt >= 64) {
                double v[64];
#pragma unroll
                for (int s = 0; s < 64; s++) v[s] = sO[buf][s].x;
#pragma unroll
                for (int s = 0; s < 64; s++) sum = sum + v[s];
            } else {
                for (int s = 0; s < (int)cnt; s++) sum = sum + sO[buf][s].x;
            }
        }
    }
    const double x0 = x[0];
    double p = sqrt(1 - 1.0 / (double)n) * x0;
    double q = (1.0 / sqrt((double)(n * (n - 1)))) * sum;
    if (l == 0) { WvHead h; h.mean = (x0 + sum) / (double)n; h.ip0 = p - q; head[k] = h; }
    {
        int buf = 0;
        const long long last = lim ? min<long long>(n - 2, lim[k]) : n - 2;   // steps m = 1 .. n-2
        const WvOps* __restrict__ o = ops + (opsOff ? (size_t)opsOff[k] : (size_t)nd.start);
        WvCk* __restrict__ ckn = ck + base[k];
        // the operands of a chunk are requested TWO chunks ahead (with the exact chains of a whole tree side by side the operand array no longer fits the Infinity Cache:
        // one chunk of 56 steps = 1.5 us of lead was less than a loaded HBM round trip, and the chain ran at half its speed)
        double4 nv = make_double4(0, 0, 0, 0), nv2 = make_double4(0, 0, 0, 0);
        if (l < WV_CS && 1 + l <= last) { const WvOps t = o[1 + l]; nv = make_double4(t.f, t.r, t.c, t.d); }
        if (l < WV_CS && 1 + WV_CS + l <= last) { const WvOps t = o[1 + WV_CS + l]; nv2 = make_double4(t.f, t.r, t.c, t.d); }
        long long c = 0;
        for (long long m0 = 1; m0 <= last; m0 += WV_CS, buf ^= 1, c++) {
            if (l == 0) { WvCk t; t.p = p; t.q = q; ckn[c] = t; }
            sO[buf][l] = nv;
            __builtin_amdgcn_wave_barrier();
            nv = nv2;
            { const long long t = m0 + 2 * WV_CS + l; if (l < WV_CS && t <= last) { const WvOps u = o[t]; nv2 = make_double4(u.f, u.r, u.c, u.d); } }
            const long long cnt = last - m0 + 1;
            if (cnt >= WV_CS) {
                double4 cur[WV_PB], nx[WV_PB];
#pragma unroll
                for (int j = 0; j < WV_PB; j++) cur[j] = sO[buf][j];
#pragma unroll
                for (int blk = 0; blk < WV_CS / WV_PB; blk++) {
                    if (blk < WV_CS / WV_PB - 1) {
#pragma unroll
                        for (int j = 0; j < WV_PB; j++) nx[j] = sO[buf][(blk + 1) * WV_PB + j];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < WV_PB; j++) wv_step<FAST>(p, q, cur[j].x, cur[j].y, cur[j].z, cur[j].w);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < WV_PB; j++) cur[j] = nx[j];
                }
            } else {
                for (int s = 0; s < (int)cnt; s++) { const double4 t = sO[buf][s]; wv_step<FAST>(p, q, t.x, t.y, t.z, t.w); }
            }
        }
    }
}

// One lane per chunk: the chunk again, from its checkpoint, with plain IEEE divisions.  Its end state must be the next checkpoint bit
// for bit (checkpoint 0 is exact by construction, so every checkpoint and therefore every value seen here is exact); the lane keeps the
// first arg-max of |I+ - I-| of its steps.
__global__ void __launch_bounds__(64) k_wv_chunks(const WvNode* __restrict__ nodes, const int32_t* __restrict__ list, const int32_t* __restrict__ base, int nLong,
                                                  int nChunks, const WvOps* __restrict__ ops, const WvCk* __restrict__ ck, WvBest* __restrict__ best, int32_t* __restrict__ flag, const long long* __restrict__ opsOff, const int32_t* __restrict__ lim) {
    const int g = blockIdx.x * 64 + threadIdx.x;
    if (g >= nChunks) return;
    const int k = wv_find(base, nLong, g);
    const WvNode nd = nodes[list[k]];
    const long long n = nd.len, last = lim ? min<long long>(n - 2, lim[k]) : n - 2;
    const long long m0 = 1 + (long long)(g - base[k]) * WV_CS;
    const long long cnt = last - m0 + 1 < WV_CS ? last - m0 + 1 : WV_CS;
    const WvOps* __restrict__ o = ops + (opsOff ? (size_t)opsOff[k] : (size_t)nd.start) + m0;
    double p = ck[g].p, q = ck[g].q;
    double bestVal = 0.0, bestAbs = -1.0; int32_t bestIdx = 0;
    for (int s = 0; s < (int)cnt; s++) {
        const WvOps t = o[s];
        p = p * t.f + t.c;
        q = q / t.f - t.d;
        const double ip = p - q, a = fabs(ip);
        if (a > bestAbs) { bestAbs = a; bestVal = ip; bestIdx = (int32_t)(m0 + s); }
    }
    WvBest b; b.val = bestVal; b.idx = bestIdx; b.pad = 0;
    best[g] = b;
    if (g + 1 < base[k + 1]) {
        const WvCk nxt = ck[g + 1];
        if (__double_as_longlong(nxt.p) != __double_as_longlong(p) || __double_as_longlong(nxt.q) != __double_as_longlong(q)) flag[k] = 1;
    }
}
// one wave per long node: first index of the maximum of |I+ - I-| (WaveletSegmentation.cs:54-68) over m = 0 and the chunk results
__global__ void __launch_bounds__(64) k_wv_reduce(const int32_t* __restrict__ list, const int32_t* __restrict__ base, const WvHead* __restrict__ head,
                                                  const WvBest* __restrict__ best, const int32_t* __restrict__ flag, WvOut* __restrict__ out) {
    const int k = blockIdx.x, l = threadIdx.x;
    const WvHead h = head[k];
    double bestVal = h.ip0, bestAbs = l == 0 ? fabs(h.ip0) : -1.0;
    int32_t bestIdx = 0;
    for (int g = base[k] + l; g < base[k + 1]; g += 64) {
        const WvBest b = best[g];
        const double a = fabs(b.val);
        if (a > bestAbs) { bestAbs = a; bestVal = b.val; bestIdx = b.idx; }      // chunks arrive in increasing index order within a lane
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const double oa = __shfl_xor(bestAbs, d, 64), ov = __shfl_xor(bestVal, d, 64);
        const int32_t oi = __shfl_xor(bestIdx, d, 64);
        if (oa > bestAbs || (oa == bestAbs && oi < bestIdx)) { bestAbs = oa; bestVal = ov; bestIdx = oi; }
    }
    if (l == 0) { WvOut o; o.coef = bestVal / fmax(0.5, h.mean / 200.0); o.ind = bestIdx + 1; o.flag = flag[k]; out[list[k]] = o; }
}

// A node of at most WV_LONG elements with its whole SUBTREE: one lane per such node.  The lane walks the subtree depth-first (the pending
// right children sit on a stack kept in the node's own slice of a per-position array, so no allocation is needed: a subtree of len
// positions never has more than len - 1 pending nodes), runs the reference's loop as it stands on every node, counts the nodes per
// (chromosome, level) for HardThresh's level weights and appends the coefficients that can survive the threshold to a global list
// (the host sorts that list back into the reference's order: level by level, left to right).  These kernels run on a second stream
// next to the long chains, so the host only ever handles the few thousand long nodes.
struct WvRoot { int32_t start; int32_t len; int32_t chrom; int32_t level; int32_t s1; int32_t cbase; };   // s1: 1-based start inside the chromosome; cbase: first bin of the chromosome
struct WvCand { int32_t chrom, level, s, b, e, pad; double coef; };
// The step coefficients depend on (n, m) only, and a short node has n <= WV_LONG: a table of all of them (3 square roots and 6 divisions per entry, the same expressions as
// above, so the same bits) turns a step of the walker from ~250 FP64 instructions into two divisions and a handful of operations — its launches were 1-4 ms of one lane's latency.
struct WvFgh { double f, g, h; };
__global__ void __launch_bounds__(256) k_wv_fgh_table(WvFgh* __restrict__ tab, int L) {
    const long long n = blockIdx.x;
    for (long long m = threadIdx.x; m <= L; m += 256) {
        WvFgh t; t.f = 0.0; t.g = 0.0; t.h = 1.0;
        if (n >= 3 && m >= 1 && m <= n - 2) { t.f = wv_factor(n, m); t.g = wv_g(n, m); t.h = wv_h(n, m); }
        tab[(size_t)n * (size_t)(L + 1) + (size_t)m] = t;
    }
}
#define WV_REP 16          // replicas of the level loop's counters (WvCn below)
struct WvRootSegs { int n; int start[WV_REP + 1], count[WV_REP + 1]; };      // stretches of the root array of one launch: the host's own and one per replica of the level loop
__global__ void __launch_bounds__(64) k_wv_subtree(const WvRoot* __restrict__ roots, const WvRootSegs segs, const double* __restrict__ X, const double* __restrict__ keepAbove,
                                                   uint32_t* __restrict__ stack, int32_t* __restrict__ counts, WvCand* __restrict__ cands,
                                                   unsigned long long* __restrict__ ncand, unsigned long long capCand, int32_t* __restrict__ overflow, const WvFgh* __restrict__ fgh, int fghLen) {
    int i = blockIdx.x * 64 + threadIdx.x, sg = 0;
    while (sg < segs.n && i >= segs.count[sg]) { i -= segs.count[sg]; sg++; }
    if (sg >= segs.n) return;
    const WvRoot R = roots[(size_t)segs.start[sg] + i];
    const double* __restrict__ xr = X + R.start;
    uint32_t* __restrict__ st = stack + R.start;
    const double keep = keepAbove[R.chrom];
    int sp = 0;
    uint32_t cur = 0u | ((uint32_t)(R.len - 1) << 9) | ((uint32_t)0 << 18);   // (first, last) relative to the root, 9 bits each; level offset above
    bool have = true;
    while (have) {
        const int a = (int)(cur & 511u), e = (int)((cur >> 9) & 511u), lv = (int)(cur >> 18);
        const long long n = e - a + 1;
        const double* __restrict__ x = xr + a;
        double sum = 0.0;
        for (long long k = 1; k < n; k++) sum = sum + x[k];
        const double x0 = x[0];
        double p = sqrt(1 - 1.0 / (double)n) * x0;
        double q = (1.0 / sqrt((double)(n * (n - 1)))) * sum;
        const double mean = (x0 + sum) / (double)n;
        double bestVal = p - q, bestAbs = fabs(bestVal);
        long long bestIdx = 0;
        if (n <= fghLen) {
            const WvFgh* __restrict__ row = fgh + (size_t)n * (size_t)(fghLen + 1);
            long long m = 1;
            for (; m + 3 < n - 1; m += 4) {                                      // the operands of four steps are requested together: nothing about them depends on the chain
                const WvFgh t0 = row[m], t1 = row[m + 1], t2 = row[m + 2], t3 = row[m + 3];
                const double x0m = x[m], x1m = x[m + 1], x2m = x[m + 2], x3m = x[m + 3];
                { p = p * t0.f + x0m * t0.g; q = q / t0.f - x0m / t0.h; const double ip = p - q, ab = fabs(ip); if (ab > bestAbs) { bestAbs = ab; bestVal = ip; bestIdx = m; } }
                { p = p * t1.f + x1m * t1.g; q = q / t1.f - x1m / t1.h; const double ip = p - q, ab = fabs(ip); if (ab > bestAbs) { bestAbs = ab; bestVal = ip; bestIdx = m + 1; } }
                { p = p * t2.f + x2m * t2.g; q = q / t2.f - x2m / t2.h; const double ip = p - q, ab = fabs(ip); if (ab > bestAbs) { bestAbs = ab; bestVal = ip; bestIdx = m + 2; } }
                { p = p * t3.f + x3m * t3.g; q = q / t3.f - x3m / t3.h; const double ip = p - q, ab = fabs(ip); if (ab > bestAbs) { bestAbs = ab; bestVal = ip; bestIdx = m + 3; } }
            }
            for (; m < n - 1; m++) {
                const WvFgh t = row[m]; const double xm = x[m];
                p = p * t.f + xm * t.g; q = q / t.f - xm / t.h;
                const double ip = p - q, ab = fabs(ip);
                if (ab > bestAbs) { bestAbs = ab; bestVal = ip; bestIdx = m; }
            }
        } else
        for (long long m = 1; m < n - 1; m++) {
            const double f = wv_factor(n, m), xm = x[m];
            p = p * f + xm * wv_g(n, m);
            q = q / f - xm / wv_h(n, m);
            const double ip = p - q, ab = fabs(ip);
            if (ab > bestAbs) { bestAbs = ab; bestVal = ip; bestIdx = m; }
        }
        const double coef = bestVal / fmax(0.5, mean / 200.0);
        const int b = a + (int)bestIdx;                                        // last position before the breakpoint, relative to the root
        const int level = R.level + lv;
        atomicAdd(&counts[(size_t)R.cbase + level], 1);                          // a tree over L bins has fewer than L levels: the chromosome's own slice
        if (fabs(coef) > keep) {
            const unsigned long long slot = atomicAdd(ncand, 1ull);
            if (slot < capCand) { WvCand c; c.chrom = R.chrom; c.level = level; c.s = R.s1 + a; c.b = R.s1 + b; c.e = R.s1 + e; c.pad = 0; c.coef = coef; cands[slot] = c; }
            else *overflow = 1;
        }
        // children (WaveletSegmentation.cs:297, 323): left [a, b] if it has at least two positions, right [b + 1, e] likewise
        const bool left = b - a >= 1, right = e - b >= 2;
        const uint32_t lvn = (uint32_t)(lv + 1) << 18;
        if (left && right) { st[sp++] = (uint32_t)(b + 1) | ((uint32_t)e << 9) | lvn; cur = (uint32_t)a | ((uint32_t)b << 9) | lvn; }
        else if (left) cur = (uint32_t)a | ((uint32_t)b << 9) | lvn;
        else if (right) cur = (uint32_t)(b + 1) | ((uint32_t)e << 9) | lvn;
        else if (sp > 0) cur = st[--sp];
        else have = false;
    }
}

// ------------------------------------------------------------------------------------------------ closed-form decisions (the default path of the long nodes)
// What a node contributes to the segmentation is (a) the FIRST arg-max of |I+ - I-| (it places the children) and (b) its coefficient — but only if that
// survives HardThresh.  Both are decisions about the reference's rounded values, and those values lie within a computable distance of the exact ones:
//     I+*[m] = sqrt((n-m-1) / (n (m+1))) S_{m+1}        I-*[m] = sqrt((m+1) / (n (n-m-1))) (S_n - S_{m+1})        S_j = x_0 + ... + x_{j-1}
// (the recurrences of GetInnerProdIter are these products, advanced one element at a time).  CanvasPartition reads its coverage from F2 text, so 100 x is a non-negative
// integer and every S_j is EXACT in 64-bit integers: two prefix arrays per chromosome (the sums and the sums of the sums), built once, give T[m] = I+*[m] - I-*[m] for
// every node of every level with no dependence between elements.  The reference's value differs from T[m] by the rounding errors of its own evaluation; carried through
// the two recurrences (I+ contracts by f_m per step, I- expands by 1 / f_m, and the products of the f telescope) they are bounded by
//     B[m] = u { A_m [4.02 SS_m + (3 + n/(n-m-1)) S_{m+1}]  +  C_m [(n + 1.01) R_1 + 4.03 RR_m + (3.5 + n/(2(n-m-1))) S_{m+1}]  +  9 (A_m S_{m+1} + C_m R_{m+1}) } (1 + 1e-2)
// with u = 2^-53, A_m / C_m the two square roots above, R_j = S_n - S_j, SS_m = S_0 + ... + S_m, RR_m = R_1 + ... + R_m (derivation: DESIGN.md, Wavelets; the terms are the
// per-step rounding of factor, of the two coefficients — whose subtractions 1/(m+1) - 1/n and n^2/(m+1) - n cancel near the end of the node, hence n/(n-m-1) — of the
// products and the sums, and of the left-to-right sum of the stretch, (n-2) u S_n, which enters I-[0] and is carried along).  A node is DECIDED when one index's lower bound
// |T| - B lies above every other index's upper bound |T| + B: that index is the reference's first arg-max whatever the rounding did.  Its coefficient is then bracketed the
// same way; a node whose bracket lies at or below the threshold every level weight implies is zeroed by HardThresh, and nothing else about it matters.  Undecided nodes
// (exact ties: flat or periodic input) and nodes whose coefficient may survive go through the exact chain above — the latter all at once at the end, because only the
// arg-max is needed to go on.  The whole tree is built on the device: a level is ONE launch over the node list the previous level left, the host looks at the
// counters once per batch of levels.  (Round 2 ran every long node through the chain: 14 M dependent steps and ~300 host round trips for a WGS sample, 0.5 s.)
#define WV_LB 128         // levels enqueued per batch (even; a launch over an empty list is not free: 512 per batch were 9 ms for a 312-level tree)
struct WvDNode { int32_t start, len, chrom, level; };      // start: index into the concatenated coverage
struct LongBufs { WvNode* nodes; WvOut* out; int32_t *list, *base, *flag; WvHead* head; WvCk* ck; WvBest* best; };      // what one pass of the chain kernels works on
#define WV_EB 1024        // launches of early exact chains between two harvests (each takes a base entry more than it has nodes)
struct WvDev { unsigned int cnt[WV_LB + 2], nch[WV_LB + 2]; unsigned int nRoots, nExact, nUndec, overflow, exactReported, seq, pad[2]; unsigned int nRootsR[WV_REP]; };      // seq: number of the batch whose end wrote this report (written last)
// The lists of a level are appended to by whichever workgroup finishes a node, and every append used to be two device atomics on the level's two counters (and one on the
// root counter for a short child): a few thousand per level on ONE line each, performed one after the other at the memory side — and every workgroup's first load (the
// chunk count of its own level) sat in the same queue.  One more atomic per append cost the level loop 6 of its 13 ms (measured with a dummy).  So the counters are
// REPLICATED: WV_REP lists per level, each with its own counter line (nodes | chunks << 32: one atomic per append) and its own stretch of the slot / chunk / root arrays;
// a workgroup appends to replica blockIdx % WV_REP, and a level walks the replicas' chunks as one index space.  cnt[] / nch[] / nRootsR[] of WvDev are the REPORT
// k_wv_batch_end sums up for the host.
struct WvCn { unsigned long long cn; unsigned long long pad[15]; };      // one counter on a 128-byte line of its own
__device__ __forceinline__ long long wv_block_scan_i64(long long v, long long* sh /* [17] */, long long* total) {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    long long inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const long long o = __shfl_up(inc, d, 64); if (l >= d) inc += o; }
    __syncthreads();
    if (l == 63) sh[w] = inc;
    __syncthreads();
    long long offw = 0, tot = 0;
    for (int k = 0; k < 16; k++) { if (k < w) offw += sh[k]; tot += sh[k]; }
    *total = tot;
    return inc + offw;
}
// highest level of every chromosome that has a node (blockIdx.y = chromosome, the workgroups of a row share the chromosome's slice of the counters; top[] starts at -1)
__global__ void __launch_bounds__(256) k_wv_tops(const int32_t* __restrict__ counts, const long long* __restrict__ off, int32_t* __restrict__ top) {
    const long long lo = off[blockIdx.y], hi = off[blockIdx.y + 1];
    int t = -1;
    for (long long i = lo + (long long)blockIdx.x * 256 + threadIdx.x; i < hi; i += (long long)gridDim.x * 256) if (counts[i]) t = (int)(i - lo);      // (ascending per thread: the last hit is the thread's highest)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(t, d, 64); t = o > t ? o : t; }
    if ((threadIdx.x & 63) == 0 && t >= 0) atomicMax(&top[blockIdx.y], t);
}
// k = 100 x as integers (checked bit for bit), P1[i] = k_0 + ... + k_i, P2[i] = P1[0] + ... + P1[i] (both restart at every chromosome).  TWO launches over tiles of 1 024 bins
// (a tile belongs to one chromosome; tile0[c] numbers them): k_wv_prefix_tiles leaves the integers (32 bits: the range check keeps them below 2^31) and every tile's
// {sum of k, sum of its tile-local inclusive sums}; k_wv_prefix_apply takes its carries from the tiles in front of it (all of them full) and writes P1 / P2.
// (Round 2-5: one workgroup per chromosome walking its tiles one after the other — 1.14 ms for chr1 of a WGS sample in front of everything else of the call.)
#define WV_PT 1024
__device__ __forceinline__ int wv_range_of_tile(const int32_t* __restrict__ tile0, int nr, int tile) {      // the range whose tiles contain `tile`: tile0[r] <= tile < tile0[r + 1]
    int lo = 0, hi = nr - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tile0[mid] <= tile) lo = mid; else hi = mid - 1; }
    return lo;
}
__global__ void __launch_bounds__(WV_PT) k_wv_prefix_tiles(const double* __restrict__ X, const long long* __restrict__ off, const int32_t* __restrict__ tile0, int nchr,
                                                           uint32_t* __restrict__ K32, long long* __restrict__ agg /* [tiles][2] */, int* __restrict__ bad) {
    __shared__ long long sh[17];
    const int c = wv_range_of_tile(tile0, nchr, (int)blockIdx.x);
    const long long hi = off[c + 1], i = off[c] + (long long)((int)blockIdx.x - tile0[c]) * WV_PT + threadIdx.x;
    long long k = 0; int myBad = 0;
    if (i < hi) {
        const double x = X[i];
        if (!(x >= 0.0 && x < 2.0e7)) { myBad = 1; if (!(fabs(x) <= 1.7976931348623157e308)) myBad = 3; }      // (NaN or an infinity: the call refuses the coverage)
        else { k = llrint(x * 100.0); if ((double)k / 100.0 != x) myBad = 1; }
        K32[i] = (uint32_t)k;
    }
    long long t1, t2;
    const long long l1 = wv_block_scan_i64(k, sh, &t1);
    (void)wv_block_scan_i64(i < hi ? l1 : 0, sh, &t2);
    if (threadIdx.x == 0) { agg[2 * (size_t)blockIdx.x] = t1; agg[2 * (size_t)blockIdx.x + 1] = t2; }
    if (myBad) bad[0] = 1;
    if (myBad == 3) bad[1] = 1;
}
__global__ void __launch_bounds__(WV_PT) k_wv_prefix_apply(const uint32_t* __restrict__ K32, const long long* __restrict__ off, const int32_t* __restrict__ tile0, int nchr,
                                                           const long long* __restrict__ agg, long long* __restrict__ P1, long long* __restrict__ P2, int* __restrict__ bad) {
    __shared__ long long sh[17];
    const int c = wv_range_of_tile(tile0, nchr, (int)blockIdx.x);
    const int first = tile0[c], j = (int)blockIdx.x - first;                   // the tile's number inside its chromosome
    const long long lo = off[c], hi = off[c + 1], base = lo + (long long)j * WV_PT, i = base + threadIdx.x;
    // carries: carry1 = sum of the k in front of the tile, carry2 = sum of the P1 in front of it = sum over the tiles T in front of (1 024 x carry1(T) + their local sums)
    long long carry1 = 0, carry2 = 0;
    for (int b0 = 0; b0 < j; b0 += WV_PT) {
        const int T = b0 + (int)threadIdx.x;
        const long long a1 = T < j ? agg[2 * (size_t)(first + T)] : 0, a2 = T < j ? agg[2 * (size_t)(first + T) + 1] : 0;
        long long tot1, tot2;
        const long long inc = wv_block_scan_i64(a1, sh, &tot1);
        const long long term = T < j ? (long long)WV_PT * (carry1 + inc - a1) + a2 : 0;
        (void)wv_block_scan_i64(term, sh, &tot2);
        carry1 += tot1; carry2 += tot2;
    }
    const long long k = i < hi ? (long long)K32[i] : 0;
    long long t1, t2;
    const long long l1 = wv_block_scan_i64(k, sh, &t1);
    const long long l2 = wv_block_scan_i64(i < hi ? l1 : 0, sh, &t2);
    if (i < hi) { P1[i] = carry1 + l1; P2[i] = carry2 + (long long)(threadIdx.x + 1) * carry1 + l2; }
    const long long cntTile = min<long long>(WV_PT, hi - base);
    const long long c2 = carry2 + cntTile * carry1 + t2, c1 = carry1 + t1;
    if (threadIdx.x == 0 && (c2 > (1ll << 61) || (hi - lo) * c1 > (1ll << 61))) *bad = 1;       // the bound's integer terms stay inside 64 bits
}

// ------------------------------------------------------------------------------------------------ exact medians of many stretches of the coverage at once
// Median (Utilities.cs:428-443) of every stretch [start, start + len) of a list: the chromosomes (threshold = mad_factor x median x variability, WaveletSegmentation.cs:394-405)
// or the stretches between preliminary breakpoints (healing step).  The coverage of this path is x = k / 100 with the integers k of k_wv_prefix_tiles, and k -> x is
// increasing: the median's two middle elements are order statistics of the k.  Three radix passes over the 31 bits (11 + 11 + 9), each ONE launch over tiles of 4 096
// elements: a workgroup counts its tile's digits in LDS (a wave adds equal digits up with ballots first: in the upper passes every element has the same one), adds
// them to the stretch's counters, and the LAST workgroup of a stretch (arrival ticket per stretch) picks the digit of both ranks and clears the counters for the next pass.
// (One workgroup per stretch with 8 passes over doubles: 0.95 ms for the chromosomes and 0.63 ms for the healing step of a WGS sample, chr1 deciding both.)
#define WV_MT 4096                     // elements of a median tile
#define WV_MD 2048                     // counters per rank and pass
#define WV_MED_MAXR 1024               // stretches per call of the multi-stretch median (more: one workgroup per stretch as before)
struct WvMedState { uint32_t prefix[2]; uint32_t pad[2]; unsigned long long rank[2]; };
__device__ __forceinline__ void wv_lds_count(uint32_t* h, uint32_t d, bool on) {
    unsigned long long todo = __ballot(on);
    const int lane = (int)(threadIdx.x & 63);
    for (int r = 0; r < 3 && todo; r++) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t d0 = (uint32_t)__shfl((int)d, leader, 64);
        const unsigned long long same = __ballot(on && d == d0);
        if (lane == leader) atomicAdd(&h[d0], (uint32_t)__popcll(same));
        todo &= ~same;
    }
    if ((todo >> lane) & 1ull) atomicAdd(&h[d], 1u);
}
__global__ void __launch_bounds__(1024) k_wv_rmed_pass(const uint32_t* __restrict__ K32, const long long* __restrict__ rStart, const long long* __restrict__ rLen, const int32_t* __restrict__ tile0, int nr,
                                                       int pass, WvMedState* __restrict__ st, uint32_t* __restrict__ hist /* [nr][2][WV_MD], zero */, uint32_t* __restrict__ tick /* [nr], zero */,
                                                       double* __restrict__ out) {
    __shared__ uint32_t lh[2][WV_MD];
    __shared__ long long sh[17];
    __shared__ int sLast;
    const int r = wv_range_of_tile(tile0, nr, (int)blockIdx.x), t = (int)threadIdx.x;
    const long long n = rLen[r], a = rStart[r] + (long long)((int)blockIdx.x - tile0[r]) * WV_MT, e = min<long long>(rStart[r] + n, a + WV_MT);
    const int shift = pass == 0 ? 20 : (pass == 1 ? 9 : 0), bits = pass == 2 ? 9 : 11;
    const uint32_t hiMask = pass == 0 ? 0u : (~0u << (shift + bits)), dMask = (1u << bits) - 1u;
    uint32_t p0 = 0, p1 = 0;
    if (pass > 0) { p0 = st[r].prefix[0]; p1 = st[r].prefix[1]; }
    uint32_t kk[WV_MT / 1024]; bool in[WV_MT / 1024];
#pragma unroll
    for (int u = 0; u < WV_MT / 1024; u++) { const long long i = a + u * 1024 + t; in[u] = i < e; kk[u] = in[u] ? K32[i] : 0u; }
    for (int i = t; i < 2 * WV_MD; i += 1024) (&lh[0][0])[i] = 0u;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < WV_MT / 1024; u++) {
        const uint32_t d = (kk[u] >> shift) & dMask;
        wv_lds_count(lh[0], d, in[u] && (kk[u] & hiMask) == p0);
        if (p1 != p0) wv_lds_count(lh[1], d, in[u] && (kk[u] & hiMask) == p1);
    }
    __syncthreads();
    uint32_t* __restrict__ gh = hist + (size_t)r * 2 * WV_MD;
    for (int i = t; i < 2 * WV_MD; i += 1024) { const uint32_t v = (&lh[0][0])[i]; if (v) atomicAdd(&gh[i], v); }
    // arrival ticket of the stretch (the idiom of clean_fast.hpp: own atomics drained, one agent-scope acquire in the last workgroup)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
        const uint32_t ntiles = (uint32_t)(tile0[r + 1] - tile0[r]);
        const int last = (__hip_atomic_fetch_add(&tick[r], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == ntiles) ? 1 : 0;
        if (last) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); __hip_atomic_store(&tick[r], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        sLast = last;
    }
    __syncthreads();
    if (!sLast) return;
    // ---- the digit of both ranks: two counters per thread, exclusive sums over the workgroup
    unsigned long long rk[2];
    if (pass == 0) { rk[0] = (unsigned long long)((n & 1) ? n / 2 : n / 2 - 1); rk[1] = (unsigned long long)(n / 2); }
    else { rk[0] = st[r].rank[0]; rk[1] = st[r].rank[1]; }
    __shared__ uint32_t sNp[2];
    for (int q = 0; q < 2; q++) {
        const uint32_t* __restrict__ h = gh + ((q == 1 && p0 != p1) ? WV_MD : 0);
        const uint32_t c0 = h[2 * t], c1 = h[2 * t + 1];
        long long tot;
        const long long inc = wv_block_scan_i64((long long)c0 + (long long)c1, sh, &tot);
        const unsigned long long before = (unsigned long long)(inc - (long long)c0 - (long long)c1), want = rk[q];
        if (want >= before && want < before + c0 + c1) {
            const int d = want < before + c0 ? 2 * t : 2 * t + 1;
            const uint32_t np = (q == 0 ? p0 : p1) | ((uint32_t)d << shift);
            st[r].prefix[q] = np; sNp[q] = np;
            st[r].rank[q] = want - before - (d == 2 * t ? 0ull : (unsigned long long)c0);
        }
        __syncthreads();
    }
    for (int i = t; i < 2 * WV_MD; i += 1024) gh[i] = 0u;                     // (plain stores: nobody else touches the stretch's counters before the next launch)
    if (pass == 2 && t == 0) {
        const double lo = (double)sNp[0] / 100.0, hi = (double)sNp[1] / 100.0;      // x = k / 100 bit for bit (k_wv_prefix_tiles checked it)
        out[r] = (n & 1) ? hi : (lo + hi) / 2;
    }
}
// A node is evaluated in chunks of WV_CH elements, one workgroup per chunk; the last chunk of a node to finish combines the partial results and takes the node's decision
// (arrival counter per node; partial results are published with write-through stores and read after one agent-scope acquire).  A 380 000-bin chromosome is 186 chunks: its
// level costs the same few microseconds as a level of short nodes.
#define WV_CH 2048
struct WvPart { double lw, t, b, u1, u2; int32_t idx, i1; };
struct WvSlot { WvDNode nd; int32_t chunkBase, arrived; };      // a node of the current level + where its chunks start + how many of them are done
__device__ __forceinline__ void wv_st_f64(double* p, double v) { __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wv_st_i32(int32_t* p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// appends a long node to replica r of a level's list: slot, chunk range, chunk -> slot map (global numbers: replica x capacity + local)
__device__ __forceinline__ void wv_push_long(WvSlot* __restrict__ list, int32_t* __restrict__ chunkNode, WvDev* __restrict__ dev, WvCn* __restrict__ cnLevel /* [WV_REP] of the level */, unsigned r,
                                             unsigned listCap, unsigned chCap, const WvDNode& nd) {
    const int nch = (int)((nd.len - 1 + WV_CH - 1) / WV_CH);                    // m = 0 .. len - 2
    const unsigned long long old = atomicAdd(&cnLevel[r].cn, ((unsigned long long)(unsigned)nch << 32) | 1ull);
    const unsigned slot = (unsigned)(old & 0xFFFFFFFFull), cb = (unsigned)(old >> 32);
    if (slot >= listCap || cb + (unsigned)nch > chCap) { dev->overflow = 1u; return; }
    WvSlot sl; sl.nd = nd; sl.chunkBase = (int32_t)(r * chCap + cb); sl.arrived = 0;
    list[(size_t)r * listCap + slot] = sl;
    for (int c = 0; c < nch; c++) chunkNode[(size_t)r * chCap + cb + c] = (int32_t)(r * listCap + slot);
}
// the host's node list (level 0, or the children of nodes the chain decided) becomes the list of iteration 0
__global__ void __launch_bounds__(256) k_wv_list_init(const WvDNode* __restrict__ in, int n, WvSlot* __restrict__ list, int32_t* __restrict__ chunkNode, WvDev* __restrict__ dev, WvCn* __restrict__ cn, unsigned listCap, unsigned chCap) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) wv_push_long(list, chunkNode, dev, cn, (unsigned)i % WV_REP, listCap, chCap, in[i]);
}
// One level of the tree for the long nodes in `cur` (dev->cnt[it] nodes, dev->nch[it] chunks): a workgroup per chunk (grid-stride).
__global__ void __launch_bounds__(256) k_wv_level(WvSlot* __restrict__ cur, const int32_t* __restrict__ curChunkNode, WvSlot* __restrict__ nxt, int32_t* __restrict__ nxtChunkNode, WvPart* __restrict__ parts,
                                                  WvDev* __restrict__ dev, WvCn* __restrict__ cn /* [levels][WV_REP] */, WvCn* __restrict__ rootCn /* [WV_REP] */, int it, unsigned maxList /* per replica */, unsigned maxChunks /* per replica */, unsigned maxList2,
                                                  const long long* __restrict__ P1, const long long* __restrict__ P2, const long long* __restrict__ off, const double* __restrict__ keepAbove,
                                                  WvRoot* __restrict__ roots, unsigned maxRoots, WvDNode* __restrict__ exactList, int32_t* __restrict__ exactInd, WvDNode* __restrict__ undecList, int32_t* __restrict__ counts, int WV_LONG) {
    __shared__ WvPart sP[4];
    __shared__ int sLast;
    // the replicas' chunks as one index space: replica r holds [pre[r], pre[r + 1])
    __shared__ unsigned sPre[WV_REP + 1];
    if (threadIdx.x < WV_REP) sPre[threadIdx.x + 1] = min((unsigned)(cn[(size_t)it * WV_REP + threadIdx.x].cn >> 32), maxChunks);
    __syncthreads();
    if (threadIdx.x == 0) { unsigned a = 0; sPre[0] = 0; for (int r = 0; r < WV_REP; r++) { a += sPre[r + 1]; sPre[r + 1] = a; } }
    __syncthreads();
    const unsigned nChunks = sPre[WV_REP], myRep = blockIdx.x % WV_REP;
    WvCn* __restrict__ cnNext = cn + (size_t)(it + 1) * WV_REP;
    const double u = 1.1102230246251565e-16;
    for (unsigned v = blockIdx.x; v < nChunks; v += gridDim.x) {
        unsigned rr = 0;
        while (v >= sPre[rr + 1]) rr++;
        const unsigned ch = rr * maxChunks + (v - sPre[rr]);                  // the chunk's number in the arrays
        const int slot = curChunkNode[ch];
        const WvDNode nd = cur[slot].nd;
        const int cbIdx = cur[slot].chunkBase;
        const long long n = nd.len, s = nd.start, cbase = off[nd.chrom];
        const long long pm1 = s > cbase ? P1[s - 1] : 0, qm1 = s > cbase ? P2[s - 1] : 0;       // sums in front of the node
        const long long Ktot = P1[s + n - 1] - pm1, KR1 = Ktot - (P1[s] - pm1);
        const double dn = (double)n, invN = 1.0 / dn;
        // ---- the chunk's m: T, B; the thread keeps its best lower bound and its two largest upper bounds
        double bLw = -1.0e300, bT = 0.0, bB = 0.0; int32_t bIdx = 0x7FFFFFFF;
        double u1 = -1.0e300, u2 = -1.0e300; int32_t i1 = -1;
        const long long m0 = (long long)(ch - (unsigned)cbIdx) * WV_CH, m1 = min<long long>(m0 + WV_CH, n - 1);
        for (long long m = m0 + threadIdx.x; m < m1; m += 256) {
            const long long Kp = P1[s + m] - pm1, Kr = Ktot - Kp;
            const long long KS = m > 0 ? (P2[s + m - 1] - qm1) - m * pm1 : 0;          // 100 SS_m
            const long long KRR = m * Ktot - KS;                                       // 100 RR_m
            const double sp = (double)Kp, sr = (double)Kr, dm1 = (double)(m + 1), dr = (double)(n - m - 1);
            // A_m = sqrt((n-m-1) / ((m+1) n)), C_m = 1 / (n A_m): one division, one square root and one more division per element (2.6 u and 4.6 u off the exact
            // square roots: with the products and the subtraction below, T is within 6.6 u (I+* + I-*) of its exact value — the 9 u of the bound); n / (n-m-1) only
            // enters the bound and is taken from the single-precision reciprocal, rounded up
            const double a = sqrt((dr / dm1) * invN), c = 1.0 / (dn * a);
            const double ip = a * sp, im = c * sr;
            const double T = ip - im;
            const double ndr = dn * (double)(__frcp_rn((float)dr) * 1.000001f) * 1.000001;
            const double B = u * (a * (4.02 * (double)KS + (3.0 + ndr) * sp) + c * ((dn + 1.01) * (double)KR1 + 4.03 * (double)KRR + (3.5 + 0.5 * ndr) * sp) + 9.0 * (ip + im)) * 1.01;
            const double at = fabs(T), lw = at - B, up = at + B;
            if (lw > bLw) { bLw = lw; bT = T; bB = B; bIdx = (int32_t)m; }
            if (up > u1) { u2 = u1; u1 = up; i1 = (int32_t)m; } else if (up > u2) u2 = up;
        }
        // ---- workgroup: best lower bound; the two largest upper bounds (every thread's candidate -> `best` of thread 0; two barriers)
        WvPart best;
        auto wg_reduce = [&]() {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const double oLw = __shfl_xor(bLw, d, 64), oT = __shfl_xor(bT, d, 64), oB = __shfl_xor(bB, d, 64); const int32_t oI = __shfl_xor(bIdx, d, 64);
                if (oLw > bLw || (oLw == bLw && oI < bIdx)) { bLw = oLw; bT = oT; bB = oB; bIdx = oI; }
                const double o1 = __shfl_xor(u1, d, 64), o2 = __shfl_xor(u2, d, 64); const int32_t oi1 = __shfl_xor(i1, d, 64);
                if (o1 > u1) { u2 = fmax(u1, o2); u1 = o1; i1 = oi1; } else u2 = fmax(u2, o1);
            }
            __syncthreads();
            if ((threadIdx.x & 63) == 0) { WvPart pk; pk.lw = bLw; pk.t = bT; pk.b = bB; pk.u1 = u1; pk.u2 = u2; pk.idx = bIdx; pk.i1 = i1; sP[threadIdx.x >> 6] = pk; }
            __syncthreads();
            if (threadIdx.x == 0) {
                best = sP[0];
                for (int w = 1; w < 4; w++) {
                    const WvPart o = sP[w];
                    if (o.u1 > best.u1) { best.u2 = fmax(best.u1, o.u2); best.u1 = o.u1; best.i1 = o.i1; } else best.u2 = fmax(best.u2, o.u1);
                    if (o.lw > best.lw || (o.lw == best.lw && o.idx < best.idx)) { best.lw = o.lw; best.t = o.t; best.b = o.b; best.idx = o.idx; }
                }
            }
        };
        wg_reduce();
        const int nch = (int)((n - 1 + WV_CH - 1) / WV_CH);
        bool lastOfNode = true;
        if (nch > 1) {
            // publish the chunk's result (write-through), then the arrival ticket; the last chunk of the node acquires and combines — with the whole workgroup: one
            // thread walking the 186 parts of a 380 000-bin chromosome was 55 us, the floor of every level that still held such a node (the top levels, and every level
            // of a stretch that keeps shedding short pieces)
            if (threadIdx.x == 0) {
                WvPart* pp = parts + ch;
                wv_st_f64(&pp->lw, best.lw); wv_st_f64(&pp->t, best.t); wv_st_f64(&pp->b, best.b); wv_st_f64(&pp->u1, best.u1); wv_st_f64(&pp->u2, best.u2); wv_st_i32(&pp->idx, best.idx); wv_st_i32(&pp->i1, best.i1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                sLast = (__hip_atomic_fetch_add(&cur[slot].arrived, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == nch) ? 1 : 0;
            }
            __syncthreads();
            lastOfNode = sLast != 0;
            if (lastOfNode) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                bLw = -1.0e300; bT = 0.0; bB = 0.0; bIdx = 0x7FFFFFFF; u1 = -1.0e300; u2 = -1.0e300; i1 = -1;
                for (int c = (int)threadIdx.x; c < nch; c += 256) {
                    const WvPart o = parts[cbIdx + c];
                    if (o.u1 > u1) { u2 = fmax(u1, o.u2); u1 = o.u1; i1 = o.i1; } else u2 = fmax(u2, o.u1);
                    if (o.lw > bLw || (o.lw == bLw && o.idx < bIdx)) { bLw = o.lw; bT = o.t; bB = o.b; bIdx = o.idx; }
                }
                wg_reduce();
            }
        }
        if (threadIdx.x == 0) {
            const int last = lastOfNode ? 1 : 0;
            if (last) {
                const double others = best.i1 == best.idx ? best.u2 : best.u1;   // the largest upper bound among the OTHER indices
                const bool decided = best.lw > others;
                atomicAdd(&counts[(size_t)cbase + nd.level], 1);                // node count per (chromosome, level): HardThresh's level weights
                if (!decided) {
                    const unsigned k = atomicAdd(&dev->nUndec, 1u);
                    if (k < maxList2) undecList[k] = nd; else dev->overflow = 1u;
                } else {
                    // children (WaveletSegmentation.cs:297, 323): left [s, b] if it has at least two positions, right [b + 1, e] likewise
                    const long long b = s + best.idx, e = s + n - 1;
                    auto place = [&](long long cs, long long ce) {
                        const int32_t len = (int32_t)(ce - cs + 1);
                        if (len > WV_LONG) wv_push_long(nxt, nxtChunkNode, dev, cnNext, myRep, maxList, maxChunks, WvDNode{(int32_t)cs, len, nd.chrom, nd.level + 1});
                        else { const unsigned k = (unsigned)atomicAdd(&rootCn[myRep].cn, 1ull); if (k < maxRoots) roots[(size_t)myRep * maxRoots + k] = WvRoot{(int32_t)cs, len, nd.chrom, nd.level + 1, (int32_t)(cs - cbase + 1), (int32_t)cbase}; else dev->overflow = 1u; }
                    };
                    if (b - s >= 1) place(s, b);
                    if (e - b >= 2) place(b + 1, e);
                    // the coefficient ipi[ind - 1] / max(0.5, mean / 200) (cs:282, 314, 340), bracketed: mean = fl(fl(x0 + sumX) / n) lies within (n + 2) u of S_n / n
                    const double meanX = (double)Ktot * 0.01 / dn, mLo = meanX * (1.0 - (dn + 4.0) * u), denLo = fmax(0.5, mLo / 200.0) * (1.0 - 4.0 * u);
                    const double coefHi = 0.01 * (fabs(best.t) + best.b) / denLo * (1.0 + 8.0 * u);
                    if (!(coefHi <= keepAbove[nd.chrom])) {                        // may survive HardThresh (or the threshold is NaN): the exact value is needed
                        const unsigned k = atomicAdd(&dev->nExact, 1u);
                        if (k < maxList2) { exactList[k] = nd; exactInd[k] = best.idx + 1; } else dev->overflow = 1u;
                    }
                }
            }
        }
        __syncthreads();
    }
}

// End of a batch of `lb` levels, on the device: the counters as they stand and the entries the exact list has gained go straight into pinned host memory (the host waits
// for an event and reads them: no copy on the stream — one behind a batch of levels stood still until the chain kernel on the OTHER stream had finished, 9 ms), then the pending
// level becomes level 0 of the next batch.  The next batch is enqueued before this one has been looked at; if nothing is pending its launches find empty lists.
__global__ void __launch_bounds__(256) k_wv_batch_end(WvDev* __restrict__ dev, WvCn* __restrict__ cn, const WvCn* __restrict__ rootCn, int lb, const WvDNode* __restrict__ exactList, const int32_t* __restrict__ exactInd,
                                                      WvDev* __restrict__ pinDev, WvDNode* __restrict__ pinExact, int32_t* __restrict__ pinExactInd, unsigned maxCopy, unsigned seq) {
    const unsigned from = dev->exactReported, nE = dev->nExact;
    const unsigned k = nE > from ? min(nE - from, maxCopy) : 0u;
    // the report: the device's block with the replicas of every level summed up (nodes, chunks) and the replicas' root counts
    { const unsigned* src = (const unsigned*)dev; unsigned* dst = (unsigned*)pinDev; const unsigned w0 = offsetof(WvDev, nRoots) / 4, w1 = offsetof(WvDev, nRootsR) / 4;
      for (unsigned i = w0 + threadIdx.x; i < w1; i += 256) if (i != offsetof(WvDev, seq) / 4) dst[i] = src[i]; }
    for (int lv = threadIdx.x; lv < WV_LB + 2; lv += 256) {             // (levels behind lb + 1 have not been touched since they were cleared)
        unsigned c = 0, n = 0;
        if (lv <= lb + 1) for (int r = 0; r < WV_REP; r++) { const unsigned long long q = cn[(size_t)lv * WV_REP + r].cn; c += (unsigned)(q & 0xFFFFFFFFull); n += (unsigned)(q >> 32); }
        pinDev->cnt[lv] = c; pinDev->nch[lv] = n;
    }
    if (threadIdx.x < WV_REP) pinDev->nRootsR[threadIdx.x] = (unsigned)rootCn[threadIdx.x].cn;
    for (unsigned i = threadIdx.x; i < k; i += 256) { pinExact[i] = exactList[from + i]; pinExactInd[i] = exactInd[from + i]; }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(&pinDev->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);      // the host polls this word: everything above is in host memory before it
        dev->exactReported = from + k;
    }
    // the pending level becomes level 0 of the next batch, replica by replica (a replica's list is where it was: list A / B by the parity of lb, which is even)
    for (int i = threadIdx.x; i < (lb + 2) * WV_REP; i += 256) {
        const int lv = i / WV_REP, r = i % WV_REP;
        if (lv == 0) continue;
        if (lv == lb) cn[r].cn = cn[i].cn;
        cn[i].cn = 0ull;
    }
}

// ------------------------------------------------------------------------------------------------ factor-of-three coverage variabilities on the device
// SegmentationInput.FactorOfThreeCoverageVariabilities (Segmentation.cs:366-429): per exponent, every chromosome is cut into triplets, each triplet leaves its median (the next
// exponent's series) and (max - min) / 2 / median, and the MEDIAN of all those ratios is the exponent's value.  The triplets are independent; the median of up to N / 3 doubles is
// a radix selection over order-preserving 64-bit keys (8 passes of 8 bits: a histogram kernel over all keys that still match the prefix, and one workgroup that picks the digit).
// NaN sorts in front of every number (SortedList<double> of .NET): its key is 0.  The host thread this replaces needed 30 ms for a WGS sample — more than the decomposition.
#define WV_F3_MAXCHR 64
struct WvF3Level { long long inOff[WV_F3_MAXCHR + 1], outOff[WV_F3_MAXCHR + 1]; int nchr; };      // chromosome c: series [inOff[c], inOff[c+1]) -> triplets [outOff[c], outOff[c+1])
struct WvF3Sel { unsigned long long prefix[2]; unsigned long long rank[2]; unsigned int hist[2][256]; };
__device__ __forceinline__ unsigned long long wv_f3_key(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    if (v != v) return 0ull;
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__global__ void __launch_bounds__(256) k_f3_triplets(const double* __restrict__ in, double* __restrict__ med, unsigned long long* __restrict__ key, const WvF3Level lv) {
    const long long total = lv.outOff[lv.nchr];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        int c = 0; while (c + 1 < lv.nchr && i >= lv.outOff[c + 1]) c++;
        const long long j = lv.inOff[c] + 3 * (i - lv.outOff[c]);
        double a = in[j], b = in[j + 1], e = in[j + 2];
        if (a > b) { const double t = a; a = b; b = t; }
        if (a > e) { const double t = a; a = e; e = t; }
        if (b > e) { const double t = b; b = e; e = t; }
        med[i] = b;
        key[i] = wv_f3_key((e - a) / 2.0 / b);
    }
}
// ranks (0-based, in .NET order) of the two middle elements; hist cleared
__global__ void k_f3_sel_init(WvF3Sel* __restrict__ S, unsigned long long rank0, unsigned long long rank1) {
    const int t = threadIdx.x;
    if (t < 2) { S->prefix[t] = 0ull; S->rank[t] = t == 0 ? rank0 : rank1; }
    for (int i = t; i < 512; i += blockDim.x) (&S->hist[0][0])[i] = 0u;
}
__global__ void __launch_bounds__(256) k_f3_hist(const unsigned long long* __restrict__ key, long long n, int shift, WvF3Sel* __restrict__ S) {
    __shared__ unsigned int sh[2][256];
    sh[0][threadIdx.x] = 0u; sh[1][threadIdx.x] = 0u;
    __syncthreads();
    const unsigned long long p0 = S->prefix[0], p1 = S->prefix[1];
    const unsigned long long hiMask = shift >= 56 ? 0ull : ~0ull << (shift + 8);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const unsigned long long k = key[i];
        const unsigned d = (unsigned)((k >> shift) & 255ull);
        if ((k & hiMask) == p0) atomicAdd(&sh[0][d], 1u);
        if (p1 != p0 && (k & hiMask) == p1) atomicAdd(&sh[1][d], 1u);
    }
    __syncthreads();
    if (sh[0][threadIdx.x]) atomicAdd(&S->hist[0][threadIdx.x], sh[0][threadIdx.x]);
    if (sh[1][threadIdx.x]) atomicAdd(&S->hist[1][threadIdx.x], sh[1][threadIdx.x]);
}
// one workgroup: the digit that holds each rank, the rank inside it; after the last pass the two keys go to `out` (pinned host memory)
__global__ void __launch_bounds__(64) k_f3_pick(WvF3Sel* __restrict__ S, int shift, unsigned long long* __restrict__ out, unsigned* __restrict__ outSeq, unsigned seq) {
    const int w = threadIdx.x;
    unsigned long long np = 0, nr = 0;
    if (w < 2) {
        const unsigned long long p0 = S->prefix[0], p1 = S->prefix[1];
        const unsigned int* h = S->hist[(w == 1 && p0 != p1) ? 1 : 0];
        const unsigned long long r = S->rank[w]; unsigned long long cum = 0; int d = 0;
        for (; d < 255; d++) { if (r < cum + h[d]) break; cum += h[d]; }
        nr = r - cum; np = (w == 0 ? p0 : p1) | ((unsigned long long)d << shift);
    }
    __syncthreads();
    if (w < 2) { S->prefix[w] = np; S->rank[w] = nr; if (shift == 0) { out[w] = np; __threadfence_system(); } }
    if (shift == 0) { __syncthreads(); if (w == 0) cvx_mail_publish(outSeq, seq); }      // (pinned host memory: the two keys, then the mailbox stamp — common.hpp)
    for (int i = w; i < 512; i += 64) (&S->hist[0][0])[i] = 0u;
}

// ------------------------------------------------------------------------------------------------ coverage variability per window on the device
// SegmentationInput.reportVariabilityByWindow (Segmentation.cs:334-349): MAD / median of every window of `window` bins, as float.  One workgroup per window: the median is a
// radix selection over order-preserving keys made on the fly from the coverage (both middle ranks at once for an even window), the MAD a second selection over
// |x - median|.  The coverage is finite (checked by the caller), so there are no NaN keys; the window's 8 x 2 passes read it from the cache.  (470 windows of 10 000 bins,
// four order statistics each, took the host's 16 threads 6 ms per WGS sample.)
__device__ __forceinline__ unsigned long long wv_key_finite(double v) { const unsigned long long b = (unsigned long long)__double_as_longlong(v); return (b >> 63) ? ~b : (b | 0x8000000000000000ull); }
__device__ __forceinline__ double wv_unkey_finite(unsigned long long k) { return __longlong_as_double((long long)((k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k)); }
template <bool DEV>
__device__ __forceinline__ double wv_window_median(const double* __restrict__ x, int n, double center, unsigned int (*hist)[256], unsigned long long* sel /* [4]: prefix0, prefix1, rank0, rank1 */) {
    const int t = threadIdx.x;
    if (t < 2) { sel[t] = 0ull; sel[2 + t] = (unsigned long long)(t == 0 ? ((n & 1) ? n / 2 : n / 2 - 1) : n / 2); }
    for (int shift = 56; shift >= 0; shift -= 8) {
        for (int i = t; i < 512; i += blockDim.x) (&hist[0][0])[i] = 0u;
        __syncthreads();
        const unsigned long long p0 = sel[0], p1 = sel[1];
        const unsigned long long hiMask = shift >= 56 ? 0ull : ~0ull << (shift + 8);
        for (int i = t; i < n; i += blockDim.x) {
            const double v = DEV ? fabs(x[i] - center) : x[i];
            const unsigned long long k = wv_key_finite(v);
            const unsigned d = (unsigned)((k >> shift) & 255ull);
            if ((k & hiMask) == p0) atomicAdd(&hist[0][d], 1u);
            if (p1 != p0 && (k & hiMask) == p1) atomicAdd(&hist[1][d], 1u);
        }
        __syncthreads();
        unsigned long long np = 0, nr = 0;
        if (t < 2) {
            const unsigned int* h = hist[(t == 1 && p0 != p1) ? 1 : 0];
            const unsigned long long r = sel[2 + t]; unsigned long long cum = 0; int d = 0;
            for (; d < 255; d++) { if (r < cum + h[d]) break; cum += h[d]; }
            nr = r - cum; np = (t == 0 ? p0 : p1) | ((unsigned long long)d << shift);
        }
        __syncthreads();
        if (t < 2) { sel[t] = np; sel[2 + t] = nr; }
        __syncthreads();
    }
    const double lo = wv_unkey_finite(sel[0]), hi = wv_unkey_finite(sel[1]);
    __syncthreads();
    return (n & 1) ? hi : (lo + hi) / 2;                        // Utilities.Median (Utilities.cs:428-443)
}
__global__ void __launch_bounds__(1024) k_wv_variability(const double* __restrict__ X, const long long* __restrict__ start, int window, float* __restrict__ out) {
    __shared__ unsigned int hist[2][256];
    __shared__ unsigned long long sel[4];
    const double* __restrict__ x = X + start[blockIdx.x];
    const double median = wv_window_median<false>(x, window, 0.0, hist, sel);
    const double mad = wv_window_median<true>(x, window, median, hist, sel);      // Utilities.Mad (Utilities.cs:451-462)
    if (threadIdx.x == 0) out[blockIdx.x] = (float)(mad / median);
}

// median of every root chromosome (the threshold of a chromosome is mad_factor x median x coverage variability, WaveletSegmentation.cs:394-405): one workgroup per chromosome,
// the same radix selection (the host's nth_element over 380 000 doubles kept the level loop waiting for 3 ms)
__global__ void __launch_bounds__(1024) k_wv_chrom_median(const double* __restrict__ X, const long long* __restrict__ off, const uint8_t* __restrict__ isRoot, double* __restrict__ out) {
    __shared__ unsigned int hist[2][256];
    __shared__ unsigned long long sel[4];
    if (!isRoot[blockIdx.x]) { if (threadIdx.x == 0) out[blockIdx.x] = 0.0; return; }
    const long long lo = off[blockIdx.x], n = off[blockIdx.x + 1] - lo;
    const double m = wv_window_median<false>(X + lo, (int)n, 0.0, hist, sel);
    if (threadIdx.x == 0) out[blockIdx.x] = m;
}

// medians of the stretches between preliminary breakpoints (GetBreakpointsAfterHealingBadSplits compares the medians left and right of every breakpoint): seg = start << 32 | length
__global__ void __launch_bounds__(1024) k_wv_segment_median(const double* __restrict__ X, const unsigned long long* __restrict__ seg, double* __restrict__ out) {
    __shared__ unsigned int hist[2][256];
    __shared__ unsigned long long sel[4];
    const unsigned long long s = seg[blockIdx.x];
    const double m = wv_window_median<false>(X + (long long)(s >> 32), (int)(s & 0xFFFFFFFFull), 0.0, hist, sel);
    if (threadIdx.x == 0) out[blockIdx.x] = m;
}

// ------------------------------------------------------------------------------------------------ host side
namespace wv {
// SortedList<T>.Median() / List<T>.Sort(): NaN sorts in front of every number
template <class T>
static void dotnet_sort(std::vector<T>& v) { auto mid = std::partition(v.begin(), v.end(), [](T a) { return a != a; }); std::sort(mid, v.end()); }
template <class T>
static T median_sorted(const std::vector<T>& v) { const size_t n = v.size(); if (!n) return T(0); return (n & 1) ? v[n / 2] : (T)((v[n / 2 - 1] + v[n / 2]) / (T)2); }
// SortedList<double>.Median() by selection: NaN (which .NET sorts in front of every number) counted as the smallest values
static double dotnet_median_select(std::vector<double>& v) {
    const size_t n = v.size();
    if (!n) return 0.0;
    auto mid = std::partition(v.begin(), v.end(), [](double a) { return a != a; });
    const size_t nan = (size_t)(mid - v.begin());
    auto kth = [&](size_t k) -> double {                  // k-th smallest in .NET order
        if (k < nan) return std::numeric_limits<double>::quiet_NaN();
        std::nth_element(mid, v.begin() + k, v.end());
        return v[k];
    };
    if (n & 1) return kth(n / 2);
    const double hi = kth(n / 2), lo = kth(n / 2 - 1);
    return (lo + hi) / 2;
}
// Utilities.Median(x, start, end) (Utilities.cs:428-443) on finite data: two order statistics instead of a full sort
static double median_range(const double* x, int64_t a, int64_t b) {
    std::vector<double> v(x + a, x + b);
    const size_t n = v.size();
    if (!n) return 0.0;
    std::nth_element(v.begin(), v.begin() + n / 2, v.end());
    const double hi = v[n / 2];
    if (n & 1) return hi;
    const double lo = *std::max_element(v.begin(), v.begin() + n / 2);
    return (lo + hi) / 2;
}
static double mad_range(const double* x, int64_t a, int64_t b) {                 // Utilities.Mad (Utilities.cs:451-462)
    const double med = median_range(x, a, b);
    std::vector<double> d((size_t)(b - a));
    for (int64_t i = a; i < b; i++) d[(size_t)(i - a)] = std::fabs(x[i] - med);
    return median_range(d.data(), 0, (int64_t)d.size());
}
static void quartiles(std::vector<float> s, float& q1, float& q2, float& q3) {    // Utilities.Quartiles (Utilities.cs:361-419)
    dotnet_sort(s);
    const int iSize = (int)s.size(), iMid = iSize / 2;
    q1 = q2 = q3 = 0;
    if (iSize == 0) return;
    if (iSize % 2 == 0) {
        q2 = (s[iMid - 1] + s[iMid]) / 2;
        const int mm = iMid / 2;
        if (iMid % 2 == 0) { q1 = (s[mm - 1] + s[mm]) / 2; q3 = (s[iMid + mm - 1] + s[iMid + mm]) / 2; }
        else { q1 = s[mm]; q3 = s[mm + iMid]; }
    } else {
        q2 = s[iMid];
        if ((iSize - 1) % 4 == 0) { const int n = (iSize - 1) / 4; q1 = (s[n - 1] * 0.25f) + (s[n] * 0.75f); q3 = (s[3 * n] * 0.75f) + (s[3 * n + 1] * 0.25f); }
        else if ((iSize - 3) % 4 == 0) { const int n = (iSize - 3) / 4; q1 = (s[n] * 0.75f) + (s[n + 1] * 0.25f); q3 = (s[3 * n + 1] * 0.25f) + (s[3 * n + 2] * 0.75f); }
    }
}
// the windows / chromosomes of the variability statistics are independent: a few host threads share them
template <class Fn>
static void host_parallel_for(int64_t n, const Fn& fn) {
    const unsigned hw = std::thread::hardware_concurrency();
    const int nt = (int)std::min<int64_t>(n, std::max(1u, std::min(hw ? hw : 1u, 16u)));
    if (nt <= 1) { for (int64_t i = 0; i < n; i++) fn(i); return; }
    std::atomic<int64_t> next(0);
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++) th.emplace_back([&]() { for (int64_t i; (i = next.fetch_add(1)) < n;) fn(i); });
    for (auto& t : th) t.join();
}
// SegmentationInput.reportVariabilityByWindow (Segmentation.cs:334-349): MAD / median per window, as float (the order of the values does
// not matter: the callers only take order statistics of them)
static std::vector<float> variability_by_window(int window, int nchr, const double* cov, const int64_t* off) {
    std::vector<std::pair<int, int64_t>> wins;
    for (int c = 0; c < nchr; c++) { const int64_t L = off[c + 1] - off[c]; for (int64_t i = 0; i < L - window; i += window) wins.push_back({c, i}); }
    std::vector<float> out(wins.size());
    host_parallel_for((int64_t)wins.size(), [&](int64_t w) {
        const double* x = cov + off[wins[w].first]; const int64_t i = wins[w].second;
        out[w] = (float)(mad_range(x, i, i + window) / median_range(x, i, i + window));
    });
    return out;
}
// SegmentationInput.GetCoverageVariability (Segmentation.cs:308-328)
static bool coverage_variability(int window, int nchr, const double* cov, const int64_t* off, double& cv) {
    if (off[nchr] - off[0] < 10 * (int64_t)window) return false;
    if (window > 10000) {
        std::vector<float> rv = variability_by_window(10000, nchr, cov, off);
        float q1, q2, q3; quartiles(rv, q1, q2, q3);
        if ((q3 - q1) / q2 > 0.015) { cv = q1; return true; }
    }
    std::vector<float> rv = variability_by_window(window, nchr, cov, off);
    dotnet_sort(rv);
    cv = (double)median_sorted(rv);
    return true;
}
// SegmentationInput.FactorOfThreeCoverageVariabilities (Segmentation.cs:366-429)
static std::vector<double> factor_of_three(int nchr, const double* cov, const int64_t* off) {
    const int maxExponent = 8;
    std::vector<double> f3{0.0};
    std::vector<std::vector<double>> cur(nchr);
    for (int c = 0; c < nchr; c++) cur[c].assign(cov + off[c], cov + off[c + 1]);
    for (int exponent = 1; exponent <= maxExponent; ++exponent) {
        std::vector<std::vector<double>> part(nchr);
        host_parallel_for(nchr, [&](int64_t c) {
            const std::vector<double>& d = cur[c];
            const size_t n = d.size() / 3;
            std::vector<double> med(n); part[c].resize(n);
            for (size_t i = 0; i < n; i++) {
                double a = d[3 * i], b = d[3 * i + 1], e = d[3 * i + 2];
                if (a > b) std::swap(a, b);
                if (a > e) std::swap(a, e);
                if (b > e) std::swap(b, e);
                med[i] = b;
                part[c][i] = (e - a) / 2.0 / b;
            }
            cur[c].swap(med);
        });
        std::vector<double> cmads;
        for (int c = 0; c < nchr; c++) cmads.insert(cmads.end(), part[c].begin(), part[c].end());
        if (cmads.size() < 50) { const double last = f3.back(); while ((int)f3.size() < maxExponent + 1) f3.push_back(last); break; }
        f3.push_back(dotnet_median_select(cmads));
    }
    return f3;
}

// Array.Sort<int>(indices, comparison) of .NET Core 2.0 (coreclr ArraySortHelper<T>.IntrospectiveSort): unstable, so the tie order of
// the level counts in HardThresh follows this exact procedure (parity unpinned, like Q11)
typedef std::function<int(int, int)> Cmp;
static void sig(int* k, const Cmp& c, int a, int b) { if (a != b && c(k[a], k[b]) > 0) std::swap(k[a], k[b]); }
static void ins(int* k, int lo, int hi, const Cmp& c) { for (int i = lo; i < hi; i++) { int j = i, t = k[i + 1]; while (j >= lo && c(t, k[j]) < 0) { k[j + 1] = k[j]; j--; } k[j + 1] = t; } }
static void down(int* k, int i, int n, int lo, const Cmp& c) {
    const int d = k[lo + i - 1];
    while (i <= n / 2) {
        int ch = 2 * i;
        if (ch < n && c(k[lo + ch - 1], k[lo + ch]) < 0) ch++;
        if (!(c(d, k[lo + ch - 1]) < 0)) break;
        k[lo + i - 1] = k[lo + ch - 1]; i = ch;
    }
    k[lo + i - 1] = d;
}
static void heap(int* k, int lo, int hi, const Cmp& c) {
    const int n = hi - lo + 1;
    for (int i = n / 2; i >= 1; i--) down(k, i, n, lo, c);
    for (int i = n; i > 1; i--) { std::swap(k[lo], k[lo + i - 1]); down(k, 1, i - 1, lo, c); }
}
static int part(int* k, int lo, int hi, const Cmp& c) {
    const int mid = lo + (hi - lo) / 2;
    sig(k, c, lo, mid); sig(k, c, lo, hi); sig(k, c, mid, hi);
    const int pivot = k[mid];
    std::swap(k[mid], k[hi - 1]);
    int left = lo, right = hi - 1;
    while (left < right) {
        while (c(k[++left], pivot) < 0) ;
        while (c(pivot, k[--right]) < 0) ;
        if (left >= right) break;
        std::swap(k[left], k[right]);
    }
    std::swap(k[left], k[hi - 1]);
    return left;
}
static void intro(int* k, int lo, int hi, int depth, const Cmp& c) {
    while (hi > lo) {
        const int sz = hi - lo + 1;
        if (sz <= 16) {
            if (sz == 1) return;
            if (sz == 2) { sig(k, c, lo, hi); return; }
            if (sz == 3) { sig(k, c, lo, hi - 1); sig(k, c, lo, hi); sig(k, c, hi - 1, hi); return; }
            ins(k, lo, hi, c); return;
        }
        if (depth == 0) { heap(k, lo, hi, c); return; }
        depth--;
        const int p = part(k, lo, hi, c);
        intro(k, p + 1, hi, depth, c);
        hi = p - 1;
    }
}
static void dotnet_sort_ints(std::vector<int>& k, const Cmp& c) {
    const int n = (int)k.size();
    if (n < 2) return;
    int fl = 0; for (int v = n; v >= 1; v /= 2) fl++;
    intro(k.data(), 0, n - 1, 2 * fl, c);
}

struct HNode { int32_t chrom, s, e; };                       // 1-based inclusive positions inside the chromosome
struct Cand { int32_t level; int32_t s, b, e; double coef; };
// host staging of the level loop in pinned memory: the per-level copies (a few KB each, 4 per level, ~300 levels) then run as real asynchronous transfers
template <typename T> struct PinVec {
    T* p = nullptr; size_t n = 0, cap = 0;
    PinVec() = default; PinVec(const PinVec&) = delete; PinVec& operator=(const PinVec&) = delete;
    ~PinVec() { if (p && !ext) (void)hipHostFree(p); }
    bool ext = false;                                         // storage handed in by the caller (a slice of the context's pinned arena): fixed capacity, not freed here
    void attach(void* q, size_t c) { p = (T*)q; cap = c; n = 0; ext = true; }
    bool reserve(size_t c) { if (c <= cap) return true; if (ext) return false; T* q = nullptr; if (hipHostMalloc((void**)&q, c * sizeof(T), hipHostMallocDefault) != hipSuccess) return false; if (p) { memcpy(q, p, n * sizeof(T)); (void)hipHostFree(p); } p = q; cap = c; return true; }
    bool resize(size_t m) { if (m > cap && !reserve(std::max(m, cap * 2))) return false; n = m; return true; }
    bool push_back(const T& v) { if (n == cap && !reserve(std::max<size_t>(64, cap * 2))) return false; p[n++] = v; return true; }
    void clear() { n = 0; }
    bool empty() const { return n == 0; }
    size_t size() const { return n; }
    T* data() { return p; } const T* data() const { return p; }
    T& operator[](size_t i) { return p[i]; } const T& operator[](size_t i) const { return p[i]; }
    T& back() { return p[n - 1]; }
    T* begin() { return p; } T* end() { return p + n; } const T* begin() const { return p; } const T* end() const { return p + n; }
};
struct ChromTree { std::vector<int> counts; std::vector<Cand> cands; double sigma = 0, keepAbove = 0; };
}  // namespace wv

extern "C" int32_t canvas_wavelets(canvas_ctx* ctx, int32_t nchr, const double* d_cov, const int64_t* h_chr_offset, int32_t is_germline,
                                   double threshold_lower, double threshold_upper, double mad_factor, int32_t variability_window, int32_t min_size,
                                   int32_t* h_breakpoints, int64_t cap, int64_t* h_bp_offset) {
    return cvx_wavelets_masked(ctx, nchr, d_cov, h_chr_offset, is_germline, threshold_lower, threshold_upper, mad_factor, variability_window, min_size, nullptr, h_breakpoints, cap, h_bp_offset);
}
// h_mask (optional): the chromosomes to decompose (canvas_wavelets_sharded: the ones this rank owns; the others get no breakpoints).  The coverage variability is genome-wide
// (Segmentation.cs:297-330) and is computed from the whole coverage whatever the mask says.
int32_t cvx_wavelets_masked(canvas_ctx* ctx, int32_t nchr, const double* d_cov, const int64_t* h_chr_offset, int32_t is_germline,
                            double threshold_lower, double threshold_upper, double mad_factor, int32_t variability_window, int32_t min_size,
                            const uint8_t* h_mask, int32_t* h_breakpoints, int64_t cap, int64_t* h_bp_offset) {
    using namespace wv;
    if (!ctx) return CANVAS_ERR_INVALID;
    const double tEntry = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    if (nchr <= 0 || !d_cov || !h_chr_offset || !h_breakpoints || !h_bp_offset || variability_window <= 0) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_wavelets: bad arguments");
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    int32_t rc0 = CANVAS_OK;
    const int64_t N = h_chr_offset[nchr] - h_chr_offset[0];
    if (N <= 0 || N > 0x7FFFFFF0ll) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_wavelets: bin count out of range");
    const int64_t base = h_chr_offset[0];
    static const int WV_LONG = [] { const char* e = cvx_hook("CANVAS_WV_LONG"); const int v = e ? atoi(e) : WV_LONG_DEFAULT; return v >= 8 && v <= WV_LONG_MAX ? v : WV_LONG_DEFAULT; }();
    std::vector<int64_t> off(nchr + 1);
    for (int c = 0; c <= nchr; c++) off[c] = h_chr_offset[c] - base;
    const double* dX = d_cov + base;
    // the coverage is needed on both sides: the decomposition runs on the device, the median-based decisions on the host
    // (pinned, kept by the context: a pageable destination is staged by the runtime — 37 MB took 8 ms — and five hipHostMalloc per call were 3 ms)
    const size_t maxLongPin = (size_t)N / WV_LONG + (size_t)nchr + 16;
    const size_t pinBytes = (((size_t)N * sizeof(double) + 255) & ~size_t(255)) + (maxLongPin + 8) * (sizeof(WvNode) + sizeof(WvOut) + 4 * sizeof(int32_t)) + 4096
                          + (maxLongPin + 8) * (sizeof(WvNode) + 3 * sizeof(int32_t) + sizeof(long long)) + WV_EB * sizeof(int32_t) + 3 * sizeof(WvDev) + 2 * (maxLongPin + 8) * (sizeof(WvDNode) + sizeof(int32_t)) + 17 * 256;      // + the staging of the early exact chains
    if (pinBytes > ctx->wv_pin_bytes) {
        if (ctx->wv_pin) { (void)hipHostFree(ctx->wv_pin); ctx->wv_pin = nullptr; ctx->wv_pin_bytes = 0; }
        CANVAS_HIP_TRY(ctx, hipHostMalloc(&ctx->wv_pin, pinBytes + pinBytes / 4, hipHostMallocDefault)); ctx->wv_pin_bytes = pinBytes + pinBytes / 4;
    }
    struct XView { double* p; double* data() const { return p; } double operator[](size_t i) const { return p[i]; } } X{(double*)ctx->wv_pin};
    char* pinCursor = (char*)ctx->wv_pin + (((size_t)N * sizeof(double) + 255) & ~size_t(255));
    unsigned long long* hF3Keys = (unsigned long long*)pinCursor; unsigned* hF3Seq = (unsigned*)(pinCursor + 128); unsigned f3Seq[8] = {0, 0, 0, 0, 0, 0, 0, 0}; pinCursor += 256;      // the two middle keys of every exponent (factor-of-three statistics on the device)
    // The copy travels on a stream of its own behind whatever produced the coverage (37 MB: 1.2 ms) while the first kernels of the call run; the host waits for it
    // where it first reads X (need_X) — with the default switches that is the reconstruction at the end.  The finiteness check rides on k_wv_prefix_tiles.
    if (!ctx->wv_copy) CANVAS_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->wv_copy, hipStreamNonBlocking));
    if (!ctx->wv_ev_in) CANVAS_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->wv_ev_in, hipEventDisableTiming));
    if (!ctx->wv_ev_x) CANVAS_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->wv_ev_x, hipEventDisableTiming));
    CANVAS_HIP_TRY(ctx, hipEventRecord(ctx->wv_ev_in, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipStreamWaitEvent(ctx->wv_copy, ctx->wv_ev_in, 0));
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(X.data(), dX, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, ctx->wv_copy));
    CANVAS_HIP_TRY(ctx, hipEventRecord(ctx->wv_ev_x, ctx->wv_copy));
    struct CopyDrain { canvas_ctx* c; ~CopyDrain() { (void)hipStreamSynchronize(c->wv_copy); } } copyDrain{ctx};      // (no way out of the call leaves the copy running)
    bool xHere = false;
    auto need_X = [&]() -> int32_t { if (!xHere) { CANVAS_HIP_TRY(ctx, hipEventSynchronize(ctx->wv_ev_x)); xHere = true; } return CANVAS_OK; };
    auto host_finite_check = [&]() -> int32_t {
        int32_t rcx = need_X(); if (rcx) return rcx;
        std::atomic<int> nonFinite{0};
        host_parallel_for((N + 262143) / 262144, [&](int64_t blk) { const int64_t a = blk * 262144, b = std::min<int64_t>(N, a + 262144); bool ok = true; for (int64_t i = a; i < b; i++) ok &= std::isfinite(X[(size_t)i]); if (!ok) nonFinite = 1; });
        if (nonFinite) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_wavelets: coverage must be finite");
        return CANVAS_OK;
    };
    // ---- from here on the call runs on two streams of its own, confined to disjoint sets of compute units: the exact chains are single waves of dependent FP64 operations
    // (s_setprio 3), and whatever shares a SIMD with one of them runs at a third of its speed — with a common pool the level kernels paid for the chains started next to
    // them (levels 12.6 -> 19.2 ms).  CANVAS_WV_CHAIN_CUS = compute units set aside for the chains (an experiment that measured no gain: default 0 = the context's own streams, no masks).
    if (!ctx->wv_streams_tried) {
        ctx->wv_streams_tried = 1;
        const char* e = cvx_hook("CANVAS_WV_CHAIN_CUS"); int k = e ? atoi(e) : 0;
        hipDeviceProp_t prop;
        if (k > 0 && hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount >= 4 * k) {
            const int ncu = prop.multiProcessorCount; std::vector<uint32_t> m((size_t)(ncu + 31) / 32, 0u), inv((size_t)(ncu + 31) / 32, 0u);
            for (int i = 0; i < ncu; i++) (i < k ? m : inv)[(size_t)i / 32] |= 1u << (i % 32);
            hipStream_t a = nullptr, b = nullptr;
            if (hipExtStreamCreateWithCUMask(&a, (uint32_t)m.size(), m.data()) == hipSuccess && hipExtStreamCreateWithCUMask(&b, (uint32_t)inv.size(), inv.data()) == hipSuccess) { ctx->wv_chain = a; ctx->wv_main = b; }
            else { if (a) (void)hipStreamDestroy(a); if (b) (void)hipStreamDestroy(b); (void)hipGetLastError(); }
        }
    }
    rc0 = canvas_side_init(ctx); if (rc0) return rc0;
    if (!ctx->wv_sub2) CANVAS_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->wv_sub2, hipStreamNonBlocking));
    if (ctx->wv_fgh_len != WV_LONG) {
        if (ctx->wv_fgh) { (void)hipFree(ctx->wv_fgh); ctx->wv_fgh = nullptr; ctx->wv_fgh_len = 0; }
        CANVAS_HIP_TRY(ctx, hipMalloc(&ctx->wv_fgh, (size_t)(WV_LONG + 1) * (size_t)(WV_LONG + 1) * sizeof(WvFgh)));
        hipLaunchKernelGGL(k_wv_fgh_table, dim3((unsigned)(WV_LONG + 1)), dim3(256), 0, ctx->stream, (WvFgh*)ctx->wv_fgh, WV_LONG);
        CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        ctx->wv_fgh_len = WV_LONG;
    }
    const WvFgh* dFgh = cvx_hook("CANVAS_WV_NO_TABLE") ? nullptr : (const WvFgh*)ctx->wv_fgh; const int fghLen = dFgh ? WV_LONG : 0;
    if (!ctx->wv_sub) CANVAS_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->wv_sub, hipStreamNonBlocking));      // the subtree walkers of the roots a batch of levels leaves: next to the following levels and to the chains
    if (ctx->wv_main && ctx->wv_chain) CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // (the masked streams start from a finished producer)
    struct StreamSwap { canvas_ctx* c; hipStream_t s0, s1; StreamSwap(canvas_ctx* x) : c(x), s0(x->stream), s1(x->side) { if (x->wv_main && x->wv_chain) { x->stream = x->wv_main; x->side = x->wv_chain; } }
                        ~StreamSwap() { if (c->stream != s0) { (void)hipStreamSynchronize(c->stream); (void)hipStreamSynchronize(c->side); } (void)hipStreamSynchronize(c->wv_sub); (void)hipStreamSynchronize(c->wv_sub2); c->stream = s0; c->side = s1; } } streamSwap(ctx);      // (both are drained on every way out)
    const bool timing = cvx_hook("CANVAS_WV_TIMING") != nullptr, trace = cvx_hook("CANVAS_WV_TRACE") != nullptr;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    double cv = 0;
    // the factor-of-three CMADs are needed by the healing step only: they are computed on a host thread while the device decomposes
    std::vector<double> f3;
    double f3Seconds = 0;
    const bool f3OnDevice = nchr <= WV_F3_MAXCHR && !cvx_hook("CANVAS_WV_F3_HOST"), f3Check = cvx_hook("CANVAS_WV_F3_CHECK") != nullptr;
    std::thread f3Thread([&]() { if (f3OnDevice && !f3Check) return; (void)hipEventSynchronize(ctx->wv_ev_x); const double a = now(); f3 = factor_of_three(nchr, X.data(), off.data()); f3Seconds = now() - a; });
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } f3Join{f3Thread};      // joined on every exit path
    // ---- device buffers
    const size_t maxLong = (size_t)N / WV_LONG + (size_t)nchr + 16, maxChunks = 3 * (size_t)N / WV_CS + maxLong + 16, maxRoots = (size_t)N / 2 + (size_t)nchr + 16;
    const size_t maxCh = (size_t)N / WV_CH + maxLong + 16;
    // the level loop's lists are WV_REP replicas (see WvCn): a replica takes an eighth of the worst case of the whole level (twice its even share; whoever appends is picked
    // by the chunk's number, which deals the appends out evenly) — but never less than one node can need
    const size_t listCap = maxLong / 8 + 1024, chCap = std::max(maxCh / 8, (size_t)N / WV_CH + 16) + 4096, rootCap = maxRoots / 8 + 4096, nCn = (size_t)(WV_LB + 3) * WV_REP;
    if (maxRoots + WV_REP * rootCap > 0x7FFFFFF0ull || WV_REP * chCap > 0x7FFFFFF0ull) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_wavelets: bin count out of range");      // (the replicas' stretches are addressed with 32-bit numbers)
    const unsigned long long capCand = (unsigned long long)N + 16;
    WsSizer sz;
    const size_t opsCap = 5 * (size_t)N;                                  // the exact chains of nodes of several levels (which overlap in position) run side by side: [N, 5 N); the first N belong to the level loop's own passes
    sz.take<WvOps>(opsCap); sz.take<WvNode>(maxLong); sz.take<WvOut>(maxLong); sz.take<int32_t>(maxLong);
    sz.take<int32_t>(maxLong + 1); sz.take<int32_t>(maxLong); sz.take<WvHead>(maxLong); sz.take<WvCk>(maxChunks); sz.take<WvBest>(maxChunks);
    sz.take<WvRoot>(maxRoots + WV_REP * rootCap); sz.take<uint32_t>((size_t)N); sz.take<int32_t>((size_t)N); sz.take<WvCand>((size_t)capCand); sz.take<double>(nchr); sz.take<unsigned long long>(2);
    const size_t maxList2 = (size_t)N / 8 + 1024;                          // nodes of ALL levels that wait for the exact chain
    sz.take<long long>((size_t)N); sz.take<long long>((size_t)N); sz.take<long long>(nchr + 1); sz.take<int>(4); sz.take<WvSlot>(WV_REP * listCap); sz.take<WvSlot>(WV_REP * listCap); sz.take<WvCn>(nCn); sz.take<WvDNode>(maxList2); sz.take<int32_t>(maxList2);
    sz.take<int32_t>(WV_REP * chCap); sz.take<int32_t>(WV_REP * chCap); sz.take<WvPart>(WV_REP * chCap); sz.take<WvDNode>(maxLong);
    sz.take<WvDNode>(maxList2); sz.take<WvDev>(1); sz.take<long long>(maxLong); sz.take<int32_t>(maxLong);
    sz.take<WvNode>(maxLong); sz.take<WvOut>(maxLong); sz.take<int32_t>(maxLong); sz.take<int32_t>(maxLong + WV_EB); sz.take<int32_t>(maxLong); sz.take<WvHead>(maxLong); sz.take<WvCk>(maxChunks); sz.take<WvBest>(maxChunks);
    sz.take<long long>(maxLong); sz.take<int32_t>(maxLong);
    const size_t f3Cap = (size_t)N / 3 + 16;
    const size_t varCap = (size_t)N / (size_t)std::max(1, std::min(variability_window, 10000)) + (size_t)nchr + 16;
    sz.take<long long>(varCap); sz.take<float>(varCap); sz.take<double>(nchr + 1); sz.take<uint8_t>(nchr + 1); sz.take<double>(varCap);
    sz.take<double>(f3Cap); sz.take<double>(f3Cap); sz.take<unsigned long long>(f3Cap); sz.take<WvF3Sel>(1);
    // the integers of the coverage, the prefix tiles' sums, and the state of the multi-stretch medians (up to WV_MED_MAXR stretches per launch set)
    const size_t prefTiles = (size_t)N / WV_PT + (size_t)nchr + 1, medR = WV_MED_MAXR;
    sz.take<uint32_t>((size_t)N); sz.take<long long>(2 * prefTiles); sz.take<int32_t>(nchr + 2); sz.take<long long>(medR); sz.take<long long>(medR); sz.take<int32_t>(medR + 2); sz.take<WvMedState>(medR);
    sz.take<uint32_t>(medR * 2 * WV_MD); sz.take<uint32_t>(medR);
    int32_t rc = canvas_ws_reserve(ctx, sz.off + 4096); if (rc) return rc;
    rc = canvas_side_init(ctx); if (rc) return rc;
    WsCarver ws(ctx->ws);
    WvOps* dOps = ws.take<WvOps>(opsCap); WvNode* dNodes = ws.take<WvNode>(maxLong); WvOut* dOut = ws.take<WvOut>(maxLong);
    int32_t* dLong = ws.take<int32_t>(maxLong); int32_t* dBase = ws.take<int32_t>(maxLong + 1);
    int32_t* dFlag = ws.take<int32_t>(maxLong); WvHead* dHead = ws.take<WvHead>(maxLong); WvCk* dCk = ws.take<WvCk>(maxChunks); WvBest* dBest = ws.take<WvBest>(maxChunks);
    WvRoot* dRoots = ws.take<WvRoot>(maxRoots + WV_REP * rootCap); uint32_t* dStack = ws.take<uint32_t>((size_t)N); int32_t* dCounts = ws.take<int32_t>((size_t)N);
    WvCand* dCands = ws.take<WvCand>((size_t)capCand); double* dKeep = ws.take<double>(nchr); unsigned long long* dNcand = ws.take<unsigned long long>(2);
    int32_t* dOverflow = (int32_t*)(dNcand + 1);
    long long* dP1 = ws.take<long long>((size_t)N); long long* dP2 = ws.take<long long>((size_t)N); long long* dOff = ws.take<long long>(nchr + 1); int* dBad = ws.take<int>(4);
    WvSlot* dListA = ws.take<WvSlot>(WV_REP * listCap); WvSlot* dListB = ws.take<WvSlot>(WV_REP * listCap); WvCn* dCn = ws.take<WvCn>(nCn); WvCn* dRootCn = dCn + (size_t)(WV_LB + 2) * WV_REP; WvDNode* dExact = ws.take<WvDNode>(maxList2); int32_t* dExactInd = ws.take<int32_t>(maxList2);
    int32_t* dChA = ws.take<int32_t>(WV_REP * chCap); int32_t* dChB = ws.take<int32_t>(WV_REP * chCap); WvPart* dParts = ws.take<WvPart>(WV_REP * chCap); WvDNode* dListIn = ws.take<WvDNode>(maxLong);
    WvDNode* dUndec = ws.take<WvDNode>(maxList2); WvDev* dDev = ws.take<WvDev>(1); long long* dOpsOff = ws.take<long long>(maxLong); int32_t* dLim = ws.take<int32_t>(maxLong);
    // a second set for the exact chains that start while the level loop is still running (side stream; slices handed out by running offsets, see exact_launch below)
    LongBufs E; E.nodes = ws.take<WvNode>(maxLong); E.out = ws.take<WvOut>(maxLong); E.list = ws.take<int32_t>(maxLong); E.base = ws.take<int32_t>(maxLong + WV_EB); E.flag = ws.take<int32_t>(maxLong);
    E.head = ws.take<WvHead>(maxLong); E.ck = ws.take<WvCk>(maxChunks); E.best = ws.take<WvBest>(maxChunks);
    long long* dOpsOffE = ws.take<long long>(maxLong); int32_t* dLimE = ws.take<int32_t>(maxLong);
    double* dF3Med[2] = {ws.take<double>(f3Cap), nullptr}; dF3Med[1] = ws.take<double>(f3Cap); unsigned long long* dF3Key = ws.take<unsigned long long>(f3Cap); WvF3Sel* dF3Sel = ws.take<WvF3Sel>(1);
    long long* dVarStart = ws.take<long long>(varCap); float* dVarOut = ws.take<float>(varCap); double* dChromMed = ws.take<double>(nchr + 1); uint8_t* dIsRoot = ws.take<uint8_t>(nchr + 1); double* dSegMed = ws.take<double>(varCap);
    uint32_t* dK32 = ws.take<uint32_t>((size_t)N); long long* dAgg = ws.take<long long>(2 * prefTiles); int32_t* dTile0P = ws.take<int32_t>(nchr + 2);
    long long* dMedStart = ws.take<long long>(medR); long long* dMedLen = ws.take<long long>(medR); int32_t* dMedTile0 = ws.take<int32_t>(medR + 2); WvMedState* dMedState = ws.take<WvMedState>(medR);
    uint32_t* dMedHist = ws.take<uint32_t>(medR * 2 * WV_MD); uint32_t* dMedTick = ws.take<uint32_t>(medR);
    // medians of up to WV_MED_MAXR stretches (start, length) of the coverage from its integers, three launches on the main stream; the results stay in `out` (device)
    auto medians_enqueue = [&](const std::vector<long long>& st, const std::vector<long long>& ln, double* out) -> int32_t {
        const size_t nr = st.size();
        std::vector<int32_t> t0(nr + 1, 0);
        for (size_t r = 0; r < nr; r++) t0[r + 1] = t0[r] + (int32_t)((ln[r] + WV_MT - 1) / WV_MT);
        if (t0[nr] == 0) return CANVAS_OK;
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dMedStart, st.data(), nr * sizeof(long long), hipMemcpyHostToDevice, ctx->stream));
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dMedLen, ln.data(), nr * sizeof(long long), hipMemcpyHostToDevice, ctx->stream));
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dMedTile0, t0.data(), (nr + 1) * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        CANVAS_HIP_TRY(ctx, hipMemsetAsync(dMedHist, 0, nr * 2 * WV_MD * sizeof(uint32_t), ctx->stream));
        CANVAS_HIP_TRY(ctx, hipMemsetAsync(dMedTick, 0, nr * sizeof(uint32_t), ctx->stream));
        CANVAS_HIP_TRY(ctx, hipMemsetAsync(out, 0, nr * sizeof(double), ctx->stream));      // (an empty stretch has no tile: its median is 0)
        for (int pass = 0; pass < 3; pass++)
            hipLaunchKernelGGL(k_wv_rmed_pass, dim3((unsigned)t0[nr]), dim3(1024), 0, ctx->stream, dK32, dMedStart, dMedLen, dMedTile0, (int)nr, pass, dMedState, dMedHist, dMedTick, out);
        return CANVAS_OK;
    };
    // what does not depend on the thresholds starts now, next to the host's order statistics: counters cleared, the exact prefix sums of the closed-form decisions
    CANVAS_HIP_TRY(ctx, hipMemsetAsync(dCounts, 0, (size_t)N * sizeof(int32_t), ctx->stream));
    CANVAS_HIP_TRY(ctx, hipMemsetAsync(dNcand, 0, 2 * sizeof(unsigned long long), ctx->stream));
    const bool tryClosedForm = !cvx_hook("CANVAS_WV_CHAIN_ONLY");
    int hBad[2] = {0, 0};
    if (tryClosedForm) {
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dOff, off.data(), (nchr + 1) * sizeof(long long), hipMemcpyHostToDevice, ctx->stream));
        CANVAS_HIP_TRY(ctx, hipMemsetAsync(dBad, 0, 4 * sizeof(int), ctx->stream));
        std::vector<int32_t> t0((size_t)nchr + 1, 0);
        for (int c = 0; c < nchr; c++) t0[(size_t)c + 1] = t0[(size_t)c] + (int32_t)((off[c + 1] - off[c] + WV_PT - 1) / WV_PT);
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dTile0P, t0.data(), ((size_t)nchr + 1) * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        if (t0[(size_t)nchr] > 0) {
            hipLaunchKernelGGL(k_wv_prefix_tiles, dim3((unsigned)t0[(size_t)nchr]), dim3(WV_PT), 0, ctx->stream, dX, dOff, dTile0P, nchr, dK32, dAgg, dBad);
            hipLaunchKernelGGL(k_wv_prefix_apply, dim3((unsigned)t0[(size_t)nchr]), dim3(WV_PT), 0, ctx->stream, dK32, dOff, dTile0P, nchr, dAgg, dP1, dP2, dBad);
        }
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(hBad, dBad, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));      // [0] not two-decimal values, [1] not finite: looked at behind the next synchronisation
    } else { rc = host_finite_check(); if (rc) return rc; }
    // ---- SegmentationInput.GetCoverageVariability (Segmentation.cs:308-328): the per-window statistics come from the device (CANVAS_WV_VAR_HOST=1: the host threads;
    // CANVAS_WV_VAR_CHECK=1: both, compared), the order statistics over the few hundred windows stay on the host
    const bool varOnDevice = !cvx_hook("CANVAS_WV_VAR_HOST"), varCheck = cvx_hook("CANVAS_WV_VAR_CHECK") != nullptr;
    std::vector<long long> varStart;
    auto var_enqueue = [&](int window) -> int32_t {          // one workgroup per window, results left in dVarOut
        varStart.clear();
        for (int c = 0; c < nchr; c++) { const int64_t L = off[c + 1] - off[c]; for (int64_t i = 0; i < L - window; i += window) varStart.push_back(off[c] + i); }
        if (varStart.size() > varCap) CANVAS_FAIL(ctx, CANVAS_ERR_HIP, "canvas_wavelets: variability staging");
        if (varStart.empty()) return CANVAS_OK;
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dVarStart, varStart.data(), varStart.size() * sizeof(long long), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_wv_variability, dim3((unsigned)varStart.size()), dim3(1024), 0, ctx->stream, dX, dVarStart, window, dVarOut);
        return CANVAS_OK;
    };
    auto var_collect = [&](int window, std::vector<float>& rv) -> int32_t {
        rv.assign(varStart.size(), 0.0f);
        if (!rv.empty()) { CANVAS_HIP_TRY(ctx, hipMemcpyAsync(rv.data(), dVarOut, rv.size() * sizeof(float), hipMemcpyDeviceToHost, ctx->stream)); CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); }
        if (varCheck) {
            int32_t rcx = need_X(); if (rcx) return rcx;
            const std::vector<float> hv = variability_by_window(window, nchr, X.data(), off.data());
            bool same = hv.size() == rv.size();
            for (size_t i = 0; same && i < hv.size(); i++) same = memcmp(&hv[i], &rv[i], 4) == 0 || (hv[i] != hv[i] && rv[i] != rv[i]);
            if (!same) CANVAS_FAIL(ctx, CANVAS_ERR_HIP, "canvas_wavelets: the device's window variabilities differ from the host's");
        }
        return CANVAS_OK;
    };
    const bool hasCV = N >= 10 * (int64_t)variability_window;
    const int firstWindow = variability_window > 10000 ? 10000 : variability_window;
    if (hasCV && varOnDevice) { rc = var_enqueue(firstWindow); if (rc) return rc; }
    // ---- roots: chromosomes longer than MinSize (WaveletsRunner.cs:117-126); their medians (order statistics of whole chromosomes: one host thread each) next to the device
    std::vector<char> isRoot((size_t)nchr, 0);
    for (int c = 0; c < nchr; c++) {
        const int64_t L = off[c + 1] - off[c];
        if (std::max<int64_t>(L, 1) <= min_size || (h_mask && !h_mask[c])) continue;
        if (L < 2) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_wavelets: a chromosome that passes MinSize needs at least two bins");
        isRoot[(size_t)c] = 1;
    }
    std::vector<double> chromMedian((size_t)nchr, 0.0), chromMad((size_t)nchr, 0.0);
    const bool medOnDevice = hasCV && varOnDevice && tryClosedForm && !cvx_hook("CANVAS_WV_MEDIAN_HOST");      // (tryClosedForm: dOff is on the device)
    const bool medFromIntegers = medOnDevice && (size_t)nchr <= WV_MED_MAXR && !cvx_hook("CANVAS_WV_MEDIAN_PER_WG");      // (needs the integers of k_wv_prefix_tiles: checked below)
    auto chrom_median_per_wg = [&]() -> int32_t {            // one workgroup per chromosome over the doubles (any finite coverage)
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dIsRoot, isRoot.data(), (size_t)nchr, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_wv_chrom_median, dim3((unsigned)nchr), dim3(1024), 0, ctx->stream, dX, dOff, dIsRoot, dChromMed);
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(chromMedian.data(), dChromMed, (size_t)nchr * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));      // (waited for with the window statistics below)
        return CANVAS_OK;
    };
    if (medFromIntegers) {
        std::vector<long long> st((size_t)nchr), ln((size_t)nchr);
        for (int c = 0; c < nchr; c++) { st[(size_t)c] = off[c]; ln[(size_t)c] = isRoot[(size_t)c] ? off[c + 1] - off[c] : 0; }
        rc = medians_enqueue(st, ln, dChromMed); if (rc) return rc;
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(chromMedian.data(), dChromMed, (size_t)nchr * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    } else if (medOnDevice) { rc = chrom_median_per_wg(); if (rc) return rc; }
    bool medResolved = !medFromIntegers;
    auto resolve_medians = [&]() -> int32_t {                // behind a synchronisation of the main stream: a coverage that is not made of two-decimal values has no integers
        if (medResolved) return CANVAS_OK;
        medResolved = true;
        if (!hBad[0]) return CANVAS_OK;
        int32_t rcm = chrom_median_per_wg(); if (rcm) return rcm;
        CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return CANVAS_OK;
    };
    if (!medOnDevice || varCheck) {
        rc = need_X(); if (rc) return rc;
        std::vector<double> hostMedian((size_t)nchr, 0.0);
        host_parallel_for(nchr, [&](int64_t c) {
            if (!isRoot[(size_t)c]) return;
            const int64_t L = off[c + 1] - off[c];
            const double* r = X.data() + off[c];
            hostMedian[(size_t)c] = median_range(r, 0, L);
            if (!hasCV) chromMad[(size_t)c] = mad_range(r, 0, L);
        });
        if (medOnDevice) {       // CANVAS_WV_VAR_CHECK: both, compared
            CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            rc = resolve_medians(); if (rc) return rc;
            for (int c = 0; c < nchr; c++) if (isRoot[(size_t)c] && memcmp(&hostMedian[(size_t)c], &chromMedian[(size_t)c], 8) != 0) CANVAS_FAIL(ctx, CANVAS_ERR_HIP, "canvas_wavelets: the device's chromosome medians differ from the host's");
        } else chromMedian.swap(hostMedian);
    }
    // ---- the factor-of-three statistics (needed by the healing step only): enqueued on a stream of their own, read at the end (144 launches: behind the enqueue of everything the thresholds wait for)
    int f3Levels = 0; long long f3Count[8] = {0, 0, 0, 0, 0, 0, 0, 0}; bool f3Enqueued = false;
    auto f3_enqueue = [&]() -> int32_t {                     // called where the host would otherwise wait: behind the first two batches of levels (or wherever the call leaves that path)
        if (f3Enqueued || !f3OnDevice) return CANVAS_OK;
        f3Enqueued = true;
        const int maxExponent = 8;
        WvF3Level lv; lv.nchr = nchr;
        std::vector<long long> len((size_t)nchr);
        for (int c = 0; c < nchr; c++) len[(size_t)c] = off[c + 1] - off[c];
        for (int c = 0; c <= nchr; c++) lv.inOff[c] = off[c];                   // exponent 1 reads the coverage itself
        for (int e = 1; e <= maxExponent; e++) {
            lv.outOff[0] = 0;
            for (int c = 0; c < nchr; c++) lv.outOff[c + 1] = lv.outOff[c] + len[(size_t)c] / 3;
            const long long M = lv.outOff[nchr];
            if (M < 50) break;                                                   // (Segmentation.cs:415-421: the remaining exponents repeat the last value)
            if ((size_t)M > f3Cap) CANVAS_FAIL(ctx, CANVAS_ERR_HIP, "canvas_wavelets: factor-of-three staging");
            const double* in = e == 1 ? dX : dF3Med[e & 1]; double* out = dF3Med[(e + 1) & 1];
            hipLaunchKernelGGL(k_f3_triplets, dim3((unsigned)std::min<long long>(2048, (M + 255) / 256)), dim3(256), 0, ctx->wv_sub2, in, out, dF3Key, lv);
            f3Seq[e - 1] = cvx_mail_arm(ctx, hF3Seq + (e - 1));
            hipLaunchKernelGGL(k_f3_sel_init, dim3(1), dim3(256), 0, ctx->wv_sub2, dF3Sel, (unsigned long long)((M & 1) ? M / 2 : M / 2 - 1), (unsigned long long)(M / 2));
            for (int shift = 56; shift >= 0; shift -= 8) {
                hipLaunchKernelGGL(k_f3_hist, dim3((unsigned)std::min<long long>(1024, (M + 1023) / 1024)), dim3(256), 0, ctx->wv_sub2, dF3Key, M, shift, dF3Sel);
                hipLaunchKernelGGL(k_f3_pick, dim3(1), dim3(64), 0, ctx->wv_sub2, dF3Sel, shift, hF3Keys + 2 * (e - 1), hF3Seq + (e - 1), f3Seq[e - 1]);
            }
            f3Count[e - 1] = M; f3Levels = e;
            for (int c = 0; c < nchr; c++) { len[(size_t)c] /= 3; lv.inOff[c] = lv.outOff[c]; }
            lv.inOff[nchr] = lv.outOff[nchr];
        }
        return CANVAS_OK;
    };
    if (hasCV && !varOnDevice) { rc = need_X(); if (rc) return rc; const bool got = coverage_variability(variability_window, nchr, X.data(), off.data(), cv); (void)got; }
    else if (hasCV) {
        std::vector<float> rv;
        rc = var_collect(firstWindow, rv); if (rc) return rc;
        bool done = false;
        if (variability_window > 10000) { float q1, q2, q3; quartiles(rv, q1, q2, q3); if ((q3 - q1) / q2 > 0.015) { cv = q1; done = true; } }
        if (!done) {
            if (variability_window > 10000) { rc = var_enqueue(variability_window); if (rc) return rc; rc = var_collect(variability_window, rv); if (rc) return rc; }
            dotnet_sort(rv);
            cv = (double)median_sorted(rv);
        }
    }

    if (!medResolved) { CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); rc = resolve_medians(); if (rc) return rc; }
    const double t1 = now();
    std::vector<ChromTree> trees(nchr);
    std::vector<HNode> cur, nxt;
    for (int c = 0; c < nchr; c++) {
        if (!isRoot[(size_t)c]) continue;
        const int64_t L = off[c + 1] - off[c];
        double threshold = mad_factor * (hasCV ? chromMedian[(size_t)c] * cv : chromMad[(size_t)c]);       // WaveletSegmentation.cs:394-405
        if (threshold < threshold_lower) threshold = threshold_lower;
        if (threshold > threshold_upper) threshold = threshold_upper;
        trees[c].sigma = threshold;
        // a coefficient at or below 2 sigma t sqrt(2 ln n) with the smallest possible level weight t is zeroed whatever the level
        // weights turn out to be: such nodes need not be kept (the margin keeps every borderline node for the exact test)
        trees[c].keepAbove = 2 * threshold * (is_germline ? 0.8 : 1.0) * std::sqrt(2 * std::log((double)L)) * (1.0 - 1e-9);
        if (!(trees[c].keepAbove == trees[c].keepAbove)) trees[c].keepAbove = -1.0;   // NaN threshold (a window with median 0): nothing is ever zeroed
    }
    for (int c = 0; c < nchr; c++) if (isRoot[(size_t)c]) cur.push_back({c, 1, (int32_t)(off[c + 1] - off[c])});
    if (trace) { fprintf(stderr, "canvas_wavelets: cv %.17g (%d)", cv, (int)hasCV); for (int c = 0; c < nchr && c < 4; c++) fprintf(stderr, "  chr%d sigma %.17g keep %.17g", c, trees[c].sigma, trees[c].keepAbove); fprintf(stderr, "\n"); }
    {
        std::vector<double> keep(nchr);
        for (int c = 0; c < nchr; c++) keep[c] = trees[c].keepAbove;
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dKeep, keep.data(), nchr * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));       // the other streams start from initialised buffers
        if (tryClosedForm && hBad[1]) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_wavelets: coverage must be finite");      // (k_wv_prefix_tiles looked at every bin)
    }
    PinVec<WvNode> hNodes; PinVec<WvOut> hOut; PinVec<int32_t> hLong, hBase, hRedo; std::vector<WvRoot> hRoots;
    WvNode* hNodesE = nullptr; int32_t *hLongE = nullptr, *hBaseE = nullptr, *hLimE = nullptr; long long* hOpsOffE = nullptr; WvDev *hdevIn = nullptr, *hdevRep[2] = {nullptr, nullptr}; WvDNode* hExactPin[2] = {nullptr, nullptr}; int32_t* hExactIndPin[2] = {nullptr, nullptr};
    {   // slices of the context's pinned arena
        auto slice = [&](size_t bytes) { void* q = pinCursor; pinCursor += (bytes + 255) & ~size_t(255); return q; };
        if (maxLong > maxLongPin) CANVAS_FAIL(ctx, CANVAS_ERR_HIP, "canvas_wavelets: staging layout");
        hNodes.attach(slice((maxLong + 1) * sizeof(WvNode)), maxLong + 1); hOut.attach(slice((maxLong + 1) * sizeof(WvOut)), maxLong + 1);
        hLong.attach(slice((maxLong + 1) * 4), maxLong + 1); hBase.attach(slice((maxLong + 2) * 4), maxLong + 2); hRedo.attach(slice((maxLong + 1) * 4), maxLong + 1);
        hNodesE = (WvNode*)slice((maxLong + 1) * sizeof(WvNode)); hLongE = (int32_t*)slice((maxLong + 1) * 4); hBaseE = (int32_t*)slice((maxLong + 1 + WV_EB) * 4); hLimE = (int32_t*)slice((maxLong + 1) * 4);
        hOpsOffE = (long long*)slice((maxLong + 1) * sizeof(long long)); hdevIn = (WvDev*)slice(sizeof(WvDev));
        for (int k = 0; k < 2; k++) { hdevRep[k] = (WvDev*)slice(sizeof(WvDev)); hExactPin[k] = (WvDNode*)slice((maxLong + 1) * sizeof(WvDNode)); hExactIndPin[k] = (int32_t*)slice((maxLong + 1) * 4); }
    }
    if (!hNodes.reserve(maxLong) || !hOut.reserve(maxLong) || !hLong.reserve(maxLong) || !hBase.reserve(maxLong + 1) || !hRedo.reserve(maxLong)) CANVAS_FAIL(ctx, CANVAS_ERR_HIP, "canvas_wavelets: cannot pin the staging buffers");
    std::vector<std::vector<WvRoot>> rootBatches;
    size_t rootsUsed = 0;
    long long levels = 0, redone = 0;
    // short nodes leave the level loop with their whole subtree: one lane each, on the side stream (nothing waits for them until the end)
    // (kernels on one stream run one after the other and a subtree lane is latency-bound, so the roots of many levels share a launch)
    auto flush_roots = [&](bool force) -> int32_t {
        if (hRoots.empty() || (!force && hRoots.size() < 4096)) return CANVAS_OK;
        if (rootsUsed + hRoots.size() > maxRoots) CANVAS_FAIL(ctx, CANVAS_ERR_HIP, "canvas_wavelets: root list overflow");
        rootBatches.emplace_back(std::move(hRoots)); hRoots.clear();         // the host copy stays alive until the side stream has been drained
        const std::vector<WvRoot>& B = rootBatches.back();
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dRoots + rootsUsed, B.data(), B.size() * sizeof(WvRoot), hipMemcpyHostToDevice, ctx->side));
        { WvRootSegs sg; memset(&sg, 0, sizeof sg); sg.n = 1; sg.start[0] = (int)rootsUsed; sg.count[0] = (int)B.size();
          hipLaunchKernelGGL(k_wv_subtree, dim3((unsigned)((B.size() + 63) / 64)), dim3(64), 0, ctx->side, dRoots, sg, dX, dKeep, dStack, dCounts, dCands, dNcand, capCand, dOverflow, dFgh, fghLen); }
        rootsUsed += B.size();
        return CANVAS_OK;
    };
    auto place = [&](int chrom, int32_t s1, int32_t e1, int level) {          // a node of the next level: long -> level loop, short -> subtree list
        const int32_t len = e1 - s1 + 1;
        if (len > WV_LONG) nxt.push_back({chrom, s1, e1});
        else hRoots.push_back({(int32_t)(off[chrom] + s1 - 1), len, chrom, level, s1, (int32_t)off[chrom]});
    };
    {   // level 0: chromosomes that are short themselves
        std::vector<HNode> longRoots;
        for (const HNode& h : cur) { nxt.clear(); place(h.chrom, h.s, h.e, 0); if (!nxt.empty()) longRoots.push_back(nxt[0]); }
        nxt.clear(); cur.swap(longRoots);
        rc = flush_roots(false); if (rc) return rc;
    }
    // the kernels of one level for the long nodes in dLong / dBase (chain = the shortcut or the IEEE division)
    auto long_pass_on = [&](hipStream_t st, const LongBufs& B, size_t nLong, int nChunks, bool fast, const long long* opsOff, const int32_t* lim) -> int32_t {
        CANVAS_HIP_TRY(ctx, hipMemsetAsync(B.flag, 0, nLong * sizeof(int32_t), st));
        if (nChunks > 0) hipLaunchKernelGGL(k_wv_coeff, dim3((unsigned)nChunks), dim3(64), 0, st, B.nodes, B.list, B.base, (int)nLong, dX, dOps, opsOff, lim);
        { ProfScope ps(ctx, "wavelet_chain", false, st);
          if (fast) hipLaunchKernelGGL((k_wv_chain_long<true>), dim3((unsigned)nLong), dim3(64), 0, st, B.nodes, B.list, B.base, dX, dOps, B.ck, B.head, opsOff, lim);
          else hipLaunchKernelGGL((k_wv_chain_long<false>), dim3((unsigned)nLong), dim3(64), 0, st, B.nodes, B.list, B.base, dX, dOps, B.ck, B.head, opsOff, lim); }
        if (nChunks > 0) hipLaunchKernelGGL(k_wv_chunks, dim3((unsigned)((nChunks + 63) / 64)), dim3(64), 0, st, B.nodes, B.list, B.base, (int)nLong, nChunks, dOps, B.ck, B.best, B.flag, opsOff, lim);
        hipLaunchKernelGGL(k_wv_reduce, dim3((unsigned)nLong), dim3(64), 0, st, B.list, B.base, B.head, B.best, B.flag, B.out);
        return CANVAS_OK;
    };
    LongBufs M; M.nodes = dNodes; M.out = dOut; M.list = dLong; M.base = dBase; M.flag = dFlag; M.head = dHead; M.ck = dCk; M.best = dBest;
    auto long_pass = [&](size_t nLong, int nChunks, bool fast, const long long* opsOff = nullptr, const int32_t* lim = nullptr) -> int32_t { return long_pass_on(ctx->stream, M, nLong, nChunks, fast, opsOff, lim); };
    auto upload_long = [&](const PinVec<int32_t>& list, const std::vector<int32_t>* lim = nullptr) -> int {                // returns the number of chunks
        hBase.clear(); hBase.push_back(0);
        size_t at = 0;
        for (int32_t i : list) { const int32_t last = lim ? std::min<int32_t>(hNodes[i].len - 2, (*lim)[at]) : hNodes[i].len - 2; at++; hBase.push_back(hBase.back() + (int32_t)((last + WV_CS - 1) / WV_CS)); }
        (void)hipMemcpyAsync(dLong, list.data(), list.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream);
        (void)hipMemcpyAsync(dBase, hBase.data(), hBase.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream);
        return hBase.back();
    };
    // ---- the exact prefix sums of the closed-form decisions; a coverage that is not made of non-negative two-decimal values takes the chains for every long node
    const double tSetup = now();
    if (timing) fprintf(stderr, "canvas_wavelets: set-up %.4f s\n", tSetup - t1);
    bool closedForm = false;
    long long nDecided = 0, nUndecided = 0, nExactChains = 0;
    if (tryClosedForm && !cur.empty()) {
        int bad = 1;
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(&bad, dBad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        closedForm = bad == 0;
    }
    if (!closedForm) { rc = f3_enqueue(); if (rc) return rc; }
    double tcA = now(), tcLevels = 0, tcUndec = 0, tcExact = 0;
    if (closedForm) {
        // ---- FindBestUnbalancedHaarDecomposition (WaveletSegmentation.cs:252-366) on the device: one launch per level, the host reads the counters once per batch of levels
        std::vector<WvDNode> hList;
        for (const HNode& h : cur) hList.push_back({(int32_t)(off[h.chrom] + h.s - 1), h.e - h.s + 1, h.chrom, 0});
        WvDev hdev; memset(&hdev, 0, sizeof hdev);
        // the short chromosomes placed above are the first roots of the device-side list
        std::vector<WvRoot> firstRoots(hRoots.begin(), hRoots.end()); hRoots.clear();          // (what flush_roots has launched already stays in front of them)
        size_t rootsLaunched = rootsUsed; unsigned subLaunches = 0;
        unsigned devRootsLaunched[WV_REP]; for (int r = 0; r < WV_REP; r++) devRootsLaunched[r] = 0;
        auto launch_subtrees = [&](unsigned upTo, const unsigned* repR) {   // the host's roots [rootsLaunched, upTo) and what the replicas of the level loop have gained: written by kernels that have completed
            WvRootSegs sg; memset(&sg, 0, sizeof sg); size_t nr = 0;
            if (upTo > rootsLaunched) { sg.start[sg.n] = (int)rootsLaunched; sg.count[sg.n] = (int)(upTo - rootsLaunched); nr += upTo - rootsLaunched; sg.n++; rootsLaunched = upTo; }
            for (int r = 0; r < WV_REP && repR; r++) {
                const unsigned have = std::min<unsigned>(repR[r], (unsigned)rootCap);
                if (have > devRootsLaunched[r]) { sg.start[sg.n] = (int)(maxRoots + (size_t)r * rootCap + devRootsLaunched[r]); sg.count[sg.n] = (int)(have - devRootsLaunched[r]); nr += have - devRootsLaunched[r]; sg.n++; devRootsLaunched[r] = have; }
            }
            if (nr == 0) return;
            hipLaunchKernelGGL(k_wv_subtree, dim3((unsigned)((nr + 63) / 64)), dim3(64), 0, (subLaunches++ & 1) ? ctx->wv_sub2 : ctx->wv_sub, dRoots, sg, dX, dKeep, dStack, dCounts, dCands, dNcand, capCand, dOverflow, dFgh, fghLen);      // (a launch lasts as long as its slowest lane: two streams take turns)
        };
        if (firstRoots.size() + rootsUsed > maxRoots) CANVAS_FAIL(ctx, CANVAS_ERR_HIP, "canvas_wavelets: root list overflow");
        if (!firstRoots.empty()) CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dRoots + rootsUsed, firstRoots.data(), firstRoots.size() * sizeof(WvRoot), hipMemcpyHostToDevice, ctx->stream));
        hdev.nRoots = (unsigned)(rootsUsed + firstRoots.size());
        CANVAS_HIP_TRY(ctx, hipMemsetAsync(dRootCn, 0, WV_REP * sizeof(WvCn), ctx->stream));          // (the replicas' root counters run over all lists of the call)
        std::vector<WvDNode> hUndec;
        // ---- the nodes whose coefficient may survive HardThresh need the exact chain — one wave of dependent FP64 operations per node, 10 ms for the longest of a WGS sample —
        // but nothing waits for its result: the chains of the nodes a batch of levels has found start on the side stream while the next levels run.  Every launch takes slices
        // of the E buffers and of the operand array (positions [N, 3 N): the first N belong to the undecided nodes' passes) by running offsets; harvest() drains the side
        // stream, checks the results, and hands the slices out again.
        struct EPending { WvDNode nd; int32_t ind; };
        std::vector<EPending> ePend;                                     // slot i of the E buffers belongs to ePend[i]
        size_t eNodes = 0, eChunks = 0, eBase = 0, exactSeen = 0; long long eOps = N;
        unsigned batchSeq = (unsigned)(ctx->wv_calls++ << 20);     // (never the number a previous call left in the pinned report slots)
        std::vector<WvOut> hOutE;
        long long chainSteps = 0, chainLongest = 0;
        auto harvest = [&]() -> int32_t {
            if (eNodes == 0) { eChunks = eBase = 0; eOps = N; return CANVAS_OK; }
            CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->side));
            CANVAS_HIP_TRY(ctx, hipGetLastError());
            hOutE.resize(eNodes);
            CANVAS_HIP_TRY(ctx, hipMemcpy(hOutE.data(), E.out, eNodes * sizeof(WvOut), hipMemcpyDeviceToHost));
            std::vector<int32_t> redo;
            for (size_t i = 0; i < eNodes; i++) if (hOutE[i].flag) redo.push_back((int32_t)i);
            if (!redo.empty()) {                                 // a checkpoint of the shortcut chain was not reproduced: IEEE divisions in the chain for those nodes (their own operand slices)
                redone += (long long)redo.size();
                std::vector<long long> ro; std::vector<int32_t> rl, rb{0};
                for (int32_t i : redo) { ro.push_back(hOpsOffE[(size_t)i]); rl.push_back(hLimE[(size_t)i]); const int32_t last = std::min<int32_t>(hNodesE[(size_t)i].len - 2, hLimE[(size_t)i]); rb.push_back(rb.back() + (int32_t)((last + WV_CS - 1) / WV_CS)); }
                CANVAS_HIP_TRY(ctx, hipMemcpy(E.list, redo.data(), redo.size() * sizeof(int32_t), hipMemcpyHostToDevice));
                CANVAS_HIP_TRY(ctx, hipMemcpy(E.base, rb.data(), rb.size() * sizeof(int32_t), hipMemcpyHostToDevice));
                CANVAS_HIP_TRY(ctx, hipMemcpy(dOpsOffE, ro.data(), ro.size() * sizeof(long long), hipMemcpyHostToDevice));
                CANVAS_HIP_TRY(ctx, hipMemcpy(dLimE, rl.data(), rl.size() * sizeof(int32_t), hipMemcpyHostToDevice));
                int32_t rcr = long_pass_on(ctx->side, E, redo.size(), rb.back(), false, dOpsOffE, dLimE); if (rcr) return rcr;
                CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->side));
                CANVAS_HIP_TRY(ctx, hipGetLastError());
                CANVAS_HIP_TRY(ctx, hipMemcpy(hOutE.data(), E.out, eNodes * sizeof(WvOut), hipMemcpyDeviceToHost));
                for (int32_t i : redo) if (hOutE[(size_t)i].flag) CANVAS_FAIL(ctx, CANVAS_ERR_HIP, "canvas_wavelets: the exact chain was not reproduced by its own check");
            }
            for (size_t i = 0; i < eNodes; i++) {
                const WvDNode& u = ePend[i].nd;
                if (hOutE[i].ind != ePend[i].ind) CANVAS_FAIL(ctx, CANVAS_ERR_HIP, "canvas_wavelets: a decided arg-max disagrees with the exact chain (error bound violated)");
                ChromTree& T = trees[u.chrom];
                const int32_t s1 = (int32_t)(u.start - off[u.chrom] + 1), e1 = s1 + u.len - 1, b = s1 + hOutE[i].ind - 1;
                if (std::fabs(hOutE[i].coef) > T.keepAbove) T.cands.push_back({u.level, s1, b, e1, hOutE[i].coef});
            }
            ePend.clear(); eNodes = eChunks = eBase = 0; eOps = N;
            return CANVAS_OK;
        };
        std::vector<WvDNode> hExactNew; std::vector<int32_t> hExactIndNew;
        std::vector<WvDNode> hDeferred; std::vector<int32_t> hDeferredInd;
        auto exact_launch = [&](bool mayWait) -> int32_t {       // the entries of hExactNew / hExactIndNew, longest chain first
            size_t ne = hExactNew.size();
            std::vector<size_t> order(ne);
            for (size_t i = 0; i < ne; i++) order[i] = i;
            std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return hExactIndNew[x] != hExactIndNew[y] ? hExactIndNew[x] > hExactIndNew[y] : x < y; });
            for (size_t a = 0; a < ne;) {
                size_t nn = 0; long long used = 0; size_t chunks = 0;
                while (a + nn < ne && eNodes + nn < maxLong && eBase + nn + 2 <= maxLong + WV_EB) {
                    const int32_t ind = hExactIndNew[order[a + nn]], len = hExactNew[order[a + nn]].len;
                    const int32_t last = std::min<int32_t>(len - 2, ind - 1);
                    const size_t ch = (size_t)((last + WV_CS - 1) / WV_CS);
                    if (eOps + used + ind + 8 > (long long)opsCap || eChunks + chunks + ch > maxChunks) break;
                    used += ind + 8; chunks += ch; nn++;
                }
                if (nn == 0) {
                    if (eNodes == 0) CANVAS_FAIL(ctx, CANVAS_ERR_HIP, "canvas_wavelets: exact-chain batching");
                    if (!mayWait) { for (; a < ne; a++) { hDeferred.push_back(hExactNew[order[a]]); hDeferredInd.push_back(hExactIndNew[order[a]]); } break; }      // (the level loop must not stand still behind the chains: these wait for its end)
                    int32_t rch = harvest(); if (rch) return rch;
                    continue;
                }
                long long o = eOps; int32_t cb = 0;
                hBaseE[eBase] = 0;
                for (size_t i = 0; i < nn; i++) {
                    const WvDNode& u = hExactNew[order[a + i]]; const int32_t ind = hExactIndNew[order[a + i]];
                    hNodesE[eNodes + i] = {u.start, u.len}; hLongE[eNodes + i] = (int32_t)i; hOpsOffE[eNodes + i] = o; hLimE[eNodes + i] = ind - 1;      // the chain stops at the decided arg-max: nothing behind it enters the coefficient
                    o += ind + 8; cb += (int32_t)((std::min<int32_t>(u.len - 2, ind - 1) + WV_CS - 1) / WV_CS); hBaseE[eBase + i + 1] = cb;
                    ePend.push_back({u, ind}); nExactChains++; chainSteps += ind; chainLongest = std::max<long long>(chainLongest, ind);
                }
                LongBufs B; B.nodes = E.nodes + eNodes; B.out = E.out + eNodes; B.list = E.list + eNodes; B.base = E.base + eBase; B.flag = E.flag + eNodes; B.head = E.head + eNodes; B.ck = E.ck + eChunks; B.best = E.best + eChunks;
                CANVAS_HIP_TRY(ctx, hipMemcpyAsync(B.nodes, hNodesE + eNodes, nn * sizeof(WvNode), hipMemcpyHostToDevice, ctx->side));
                CANVAS_HIP_TRY(ctx, hipMemcpyAsync(B.list, hLongE + eNodes, nn * sizeof(int32_t), hipMemcpyHostToDevice, ctx->side));
                CANVAS_HIP_TRY(ctx, hipMemcpyAsync(B.base, hBaseE + eBase, (nn + 1) * sizeof(int32_t), hipMemcpyHostToDevice, ctx->side));
                CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dOpsOffE + eNodes, hOpsOffE + eNodes, nn * sizeof(long long), hipMemcpyHostToDevice, ctx->side));
                CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dLimE + eNodes, hLimE + eNodes, nn * sizeof(int32_t), hipMemcpyHostToDevice, ctx->side));
                int32_t rcl = long_pass_on(ctx->side, B, nn, cb, true, dOpsOffE + eNodes, dLimE + eNodes); if (rcl) return rcl;
                eNodes += nn; eBase += nn + 1; eChunks += (size_t)cb; eOps = o;
                a += nn;
            }
            return CANVAS_OK;
        };
        // the new entries of the device's exact list (the main stream has just been drained: the copy costs a few microseconds of it)
        auto fetch_exact = [&](unsigned upTo) -> int32_t {       // at most maxLong entries per call (the pinned staging); the caller comes back for the rest
            hExactNew.clear(); hExactIndNew.clear();
            if (upTo <= exactSeen) return CANVAS_OK;
            const size_t k = std::min<size_t>(upTo - exactSeen, maxLong);
            CANVAS_HIP_TRY(ctx, hipMemcpyAsync(hExactPin[0], dExact + exactSeen, k * sizeof(WvDNode), hipMemcpyDeviceToHost, ctx->stream));
            CANVAS_HIP_TRY(ctx, hipMemcpyAsync(hExactIndPin[0], dExactInd + exactSeen, k * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
            CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            hExactNew.assign(hExactPin[0], hExactPin[0] + k); hExactIndNew.assign(hExactIndPin[0], hExactIndPin[0] + k);
            exactSeen += k;
            return CANVAS_OK;
        };
        static const unsigned levelGrid = [] { const char* e = cvx_hook("CANVAS_WV_LEVEL_GRID"); const int v = e ? atoi(e) : 2048; return (unsigned)(v >= 64 && v <= 8192 ? v : 2048); }();      // (2 048 workgroups of 256 = eight waves per SIMD: levels 13.6-14.6 -> 13.0-13.2 ms against 1 024; 4 096: 12.7)
        while (!hList.empty()) {
            if (hList.size() > maxLong) CANVAS_FAIL(ctx, CANVAS_ERR_HIP, "canvas_wavelets: long-node list overflow");
            const int nIn = (int)hList.size();
            CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dListIn, hList.data(), hList.size() * sizeof(WvDNode), hipMemcpyHostToDevice, ctx->stream));
            hList.clear();
            // batches of levels, short ones first (the longest chains hang off the first levels), one batch enqueued ahead of the one the host is waiting for
            for (int i = 0; i < WV_LB + 2; i++) { hdev.cnt[i] = 0; hdev.nch[i] = 0; }
            hdev.exactReported = (unsigned)exactSeen;
            *hdevIn = hdev;
            CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dDev, hdevIn, sizeof hdev, hipMemcpyHostToDevice, ctx->stream));
            CANVAS_HIP_TRY(ctx, hipMemsetAsync(dCn, 0, (size_t)(WV_LB + 2) * WV_REP * sizeof(WvCn), ctx->stream));
            hipLaunchKernelGGL(k_wv_list_init, dim3((unsigned)((nIn + 255) / 256)), dim3(256), 0, ctx->stream, dListIn, nIn, dListA, dChA, dDev, dCn, (unsigned)listCap, (unsigned)chCap);
            hipEvent_t ev[2] = {ctx->side_ev, ctx->side_ev2};
            static const int firstBatch = [] { const char* e = cvx_hook("CANVAS_WV_FIRST_BATCH"); const int v = e ? atoi(e) : 16; return v >= 2 && v <= 64 && !(v & 1) ? v : 16; }();
            int lbOf[2] = {0, 0}, lbNext = firstBatch; unsigned seqOf[2] = {0, 0};       // (16, 32, 64 ... levels.  Smaller first batches start the longest chain earlier but split the chains over several launches of ONE in-order stream, which then run one after the other: 2, 4, 8 ... was 9 ms slower)
            auto enqueue_batch = [&](int slot) -> int32_t {
                const int lb = lbNext; lbOf[slot] = lb; lbNext = std::min(64, lbNext * 2);
                for (int it = 0; it < lb; it++)
                    hipLaunchKernelGGL(k_wv_level, dim3(levelGrid), dim3(256), 0, ctx->stream, (it & 1) ? dListB : dListA, (it & 1) ? dChB : dChA, (it & 1) ? dListA : dListB, (it & 1) ? dChA : dChB, dParts,
                                       dDev, dCn, dRootCn, it, (unsigned)listCap, (unsigned)chCap, (unsigned)maxList2, dP1, dP2, dOff, dKeep, dRoots + maxRoots, (unsigned)rootCap, dExact, dExactInd, dUndec, dCounts, WV_LONG);
                hipLaunchKernelGGL(k_wv_batch_end, dim3(1), dim3(256), 0, ctx->stream, dDev, dCn, dRootCn, lb, dExact, dExactInd, hdevRep[slot], hExactPin[slot], hExactIndPin[slot], (unsigned)maxLong, ++batchSeq);
                seqOf[slot] = batchSeq;
                CANVAS_HIP_TRY(ctx, hipEventRecord(ev[slot], ctx->stream));
                return CANVAS_OK;
            };
            rc = enqueue_batch(0); if (rc) return rc;
            rc = enqueue_batch(1); if (rc) return rc;
            rc = f3_enqueue(); if (rc) return rc;                // (the host would wait for the first batch now)
            for (int k = 0;; k++) {
                const int slot = k & 1, lb = lbOf[slot];
                const double tw0 = now();
                CANVAS_HIP_TRY(ctx, hipEventSynchronize(ev[slot]));
                {   // the report carries the number of its batch in its last word
                    const volatile unsigned* sq = &hdevRep[slot]->seq;
                    if (*sq != seqOf[slot]) {
                        if (trace) fprintf(stderr, "canvas_wavelets: the event of batch %d was complete before its report (seq %u, expected %u)\n", k, *sq, seqOf[slot]);
                        const double tp = now();
                        while (*sq != seqOf[slot]) { if (now() - tp > 20.0) CANVAS_FAIL(ctx, CANVAS_ERR_HIP, "canvas_wavelets: the report of a batch of levels did not arrive"); __builtin_ia32_pause(); }
                    }
                    std::atomic_thread_fence(std::memory_order_acquire);
                }
                const double tw1 = now();
                hdev = *hdevRep[slot];
                if (hdev.overflow) CANVAS_FAIL(ctx, CANVAS_ERR_HIP, "canvas_wavelets: a node list of the device-side tree overflowed");
                for (int i = 0; i < lb; i++) nDecided += hdev.cnt[i];
                const bool pending = hdev.cnt[lb] != 0;
                hExactNew.clear(); hExactIndNew.clear();
                if (hdev.exactReported != exactSeen) CANVAS_FAIL(ctx, CANVAS_ERR_HIP, "canvas_wavelets: exact-list bookkeeping");
                { const size_t cnt = hdev.nExact > hdev.exactReported ? std::min<size_t>(hdev.nExact - hdev.exactReported, maxLong) : 0;
                  hExactNew.assign(hExactPin[slot], hExactPin[slot] + cnt); hExactIndNew.assign(hExactIndPin[slot], hExactIndPin[slot] + cnt); exactSeen += cnt; }
                if (pending) { rc = enqueue_batch(slot); if (rc) return rc; }                 // (batch k + 2; batch k + 1 is running)
                launch_subtrees(std::min<unsigned>(hdev.nRoots, (unsigned)maxRoots), hdev.nRootsR);         // (the host's first roots were uploaded on the main stream in front of batch 0: complete as well)
                const double tw2 = now(); const size_t nNew = hExactNew.size();
                if (!hExactNew.empty()) { rc = exact_launch(false); if (rc) return rc; }
                if (trace && k == 0) fprintf(stderr, "canvas_wavelets: first report: nodes per level %u %u %u %u %u %u, chunks %u %u %u %u, undecided %u, roots %u, exact %u\n", hdev.cnt[0], hdev.cnt[1], hdev.cnt[2], hdev.cnt[3], hdev.cnt[4], hdev.cnt[5], hdev.nch[0], hdev.nch[1], hdev.nch[2], hdev.nch[3], hdev.nUndec, hdev.nRoots, hdev.nExact);
                if (trace) fprintf(stderr, "canvas_wavelets: batch %d (%d levels): waited %.0f us from %.0f us, enqueue %.0f us, %zu chains launched in %.0f us, pending %u nodes\n", k, lb, (tw1 - tw0) * 1e6, (tw0 - tSetup) * 1e6, (tw2 - tw1) * 1e6, nNew, (now() - tw2) * 1e6, hdev.cnt[lb]);
                if (!pending) break;                                                        // (batch k + 1 finds empty lists)
            }
            CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            CANVAS_HIP_TRY(ctx, hipGetLastError());
            nDecided -= hdev.nUndec;
            tcLevels += now() - tcA; tcA = now();
            // ---- undecided nodes (exact ties between candidates): the chain decides them, their children go on as a new list
            if (hdev.nUndec) {
                const size_t nu = hdev.nUndec;
                nUndecided += (long long)nu;
                hUndec.resize(nu);
                CANVAS_HIP_TRY(ctx, hipMemcpyAsync(hUndec.data(), dUndec, nu * sizeof(WvDNode), hipMemcpyDeviceToHost, ctx->stream));
                CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                // (a node and its descendants are never undecided together — the descendants do not exist yet — so the stretches are disjoint: but the list may exceed
                // the per-level bound, so it goes through in slices)
                for (size_t a = 0; a < nu; a += maxLong) {
                    const size_t nn = std::min(maxLong, nu - a);
                    hNodes.resize(nn); hOut.resize(nn); hLong.resize(nn);
                    for (size_t i = 0; i < nn; i++) { hNodes[i] = {hUndec[a + i].start, hUndec[a + i].len}; hLong[i] = (int32_t)i; }
                    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dNodes, hNodes.data(), nn * sizeof(WvNode), hipMemcpyHostToDevice, ctx->stream));
                    { const int nChunks = upload_long(hLong); rc = long_pass(nn, nChunks, false); if (rc) return rc; }
                    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(hOut.data(), dOut, nn * sizeof(WvOut), hipMemcpyDeviceToHost, ctx->stream));
                    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                    CANVAS_HIP_TRY(ctx, hipGetLastError());
                    for (size_t i = 0; i < nn; i++) {
                        const WvDNode& u = hUndec[a + i];
                        if (hOut[i].flag) CANVAS_FAIL(ctx, CANVAS_ERR_HIP, "canvas_wavelets: the exact chain was not reproduced by its own check");
                        ChromTree& T = trees[u.chrom];
                        const int32_t s1 = (int32_t)(u.start - off[u.chrom] + 1), e1 = s1 + u.len - 1, b = s1 + hOut[i].ind - 1;
                        if (std::fabs(hOut[i].coef) > T.keepAbove || !(T.keepAbove == T.keepAbove)) T.cands.push_back({u.level, s1, b, e1, hOut[i].coef});
                        auto placeD = [&](int32_t cs, int32_t ce) {
                            const int32_t len = ce - cs + 1;
                            if (len > WV_LONG) hList.push_back({(int32_t)(off[u.chrom] + cs - 1), len, u.chrom, u.level + 1});
                            else hRoots.push_back({(int32_t)(off[u.chrom] + cs - 1), len, u.chrom, u.level + 1, cs, (int32_t)off[u.chrom]});
                        };
                        if (b - s1 >= 1) placeD(s1, b);
                        if (e1 - b >= 2) placeD(b + 1, e1);
                    }
                }
                if (!hRoots.empty()) {
                    if (hdev.nRoots + hRoots.size() > maxRoots) CANVAS_FAIL(ctx, CANVAS_ERR_HIP, "canvas_wavelets: root list overflow");
                    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dRoots + hdev.nRoots, hRoots.data(), hRoots.size() * sizeof(WvRoot), hipMemcpyHostToDevice, ctx->stream));
                    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                    hdev.nRoots += (unsigned)hRoots.size(); hRoots.clear();
                }
                hdev.nUndec = 0;
            }
            tcUndec += now() - tcA; tcA = now();
        }
        // ---- the chains that were not started early (more entries than one report holds), then the short nodes with their whole subtrees: one lane each,
        // on the main stream (the side stream belongs to the chains); then everything the chains have found
        while (exactSeen < hdev.nExact) { rc = fetch_exact(hdev.nExact); if (rc) return rc; rc = exact_launch(true); if (rc) return rc; }
        if (!hDeferred.empty()) { hExactNew.swap(hDeferred); hExactIndNew.swap(hDeferredInd); rc = exact_launch(true); if (rc) return rc; }
        launch_subtrees(hdev.nRoots, hdev.nRootsR);
        rc = harvest(); if (rc) return rc;
        CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->wv_sub));
        CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->wv_sub2));
        if (timing) fprintf(stderr, "canvas_wavelets: %lld exact chains, %lld steps in all, longest %lld (device list %u entries, %zu seen)\n", nExactChains, chainSteps, chainLongest, hdev.nExact, exactSeen);
        tcExact = now() - tcA;
        if (timing) fprintf(stderr, "canvas_wavelets: closed-form levels %.4f s, undecided nodes %.4f s, rest of the exact chains + subtrees %.4f s (%u roots)\n", tcLevels, tcUndec, tcExact, [&] { unsigned t = hdev.nRoots; for (int r = 0; r < WV_REP; r++) t += hdev.nRootsR[r]; return t; }());
        rootsUsed = hdev.nRoots;
        cur.clear();
    }
    ctx->wv_stats[0] = nDecided; ctx->wv_stats[1] = nUndecided; ctx->wv_stats[2] = nExactChains; ctx->wv_stats[3] = closedForm ? 1 : 0;
    // ---- the same with the chain for every long node (coverage that is not two-decimal text, or CANVAS_WV_CHAIN_ONLY=1), level by level for all chromosomes at once
    for (int level = 0; !cur.empty(); level++, levels++) {
        const size_t nn = cur.size();
        if (nn > maxLong) CANVAS_FAIL(ctx, CANVAS_ERR_HIP, "canvas_wavelets: long-node list overflow");
        hNodes.resize(nn); hOut.resize(nn); hLong.resize(nn);
        for (size_t i = 0; i < nn; i++) { const HNode& h = cur[i]; hNodes[i] = {(int32_t)(off[h.chrom] + h.s - 1), h.e - h.s + 1}; hLong[i] = (int32_t)i; }
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dNodes, hNodes.data(), nn * sizeof(WvNode), hipMemcpyHostToDevice, ctx->stream));
        { const int nChunks = upload_long(hLong); rc = long_pass(nn, nChunks, true); if (rc) return rc; }
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(hOut.data(), dOut, nn * sizeof(WvOut), hipMemcpyDeviceToHost, ctx->stream));
        CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        CANVAS_HIP_TRY(ctx, hipGetLastError());
        hRedo.clear();
        for (int32_t i : hLong) if (hOut[i].flag || cvx_hook("CANVAS_WV_TEST_EXACT")) hRedo.push_back(i);
        if (!hRedo.empty()) {                                // a checkpoint of the shortcut chain was not reproduced: IEEE divisions in the chain for those nodes
            if (!cvx_hook("CANVAS_WV_TEST_EXACT")) redone += (long long)hRedo.size();
            const int nChunks = upload_long(hRedo);
            rc = long_pass(hRedo.size(), nChunks, false); if (rc) return rc;
            CANVAS_HIP_TRY(ctx, hipMemcpyAsync(hOut.data(), dOut, nn * sizeof(WvOut), hipMemcpyDeviceToHost, ctx->stream));
            CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            CANVAS_HIP_TRY(ctx, hipGetLastError());
            for (int32_t i : hRedo) if (hOut[i].flag) CANVAS_FAIL(ctx, CANVAS_ERR_HIP, "canvas_wavelets: the exact chain was not reproduced by its own check");
        }
        nxt.clear();
        for (size_t i = 0; i < nn; i++) {
            const HNode& h = cur[i];
            ChromTree& T = trees[h.chrom];
            if ((int)T.counts.size() <= level) T.counts.resize(level + 1, 0);
            T.counts[level]++;
            const int32_t b = h.s + hOut[i].ind - 1;          // "last time point before the breakpoint" (cs:279, 311, 337)
            if (std::fabs(hOut[i].coef) > T.keepAbove) T.cands.push_back({level, h.s, b, h.e, hOut[i].coef});
            if (b - h.s >= 1) place(h.chrom, h.s, b, level + 1);
            if (h.e - b >= 2) place(h.chrom, b + 1, h.e, level + 1);
        }
        rc = flush_roots(false); if (rc) return rc;
        cur.swap(nxt);
    }
    rc = flush_roots(true); if (rc) return rc;
    // ---- the subtrees: node counts per level and the surviving coefficients come back in one piece
    {
        unsigned long long hN[2] = {0, 0};
        const double tSide0 = now();
        CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->side));
        if (timing) fprintf(stderr, "canvas_wavelets: waited %.4f s for the subtrees\n", now() - tSide0);
        CANVAS_HIP_TRY(ctx, hipGetLastError());
        CANVAS_HIP_TRY(ctx, hipMemcpy(hN, dNcand, sizeof hN, hipMemcpyDeviceToHost));
        if (hN[1] & 0xFFFFFFFFull) CANVAS_FAIL(ctx, CANVAS_ERR_HIP, "canvas_wavelets: candidate list overflow");
        // node counts per (chromosome, level): a chromosome's slice is as long as the chromosome, but only its first few hundred entries are used — the device finds the
        // highest used level per chromosome and only that much comes back (the whole 4 N bytes took several milliseconds)
        std::vector<std::vector<int32_t>> hCounts((size_t)nchr);
        {
            int32_t* dTop = (int32_t*)dStack;                    // (the subtree stacks are no longer needed)
            CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dOff, off.data(), (nchr + 1) * sizeof(long long), hipMemcpyHostToDevice, ctx->stream));      // (the chain-only path has not uploaded it)
            CANVAS_HIP_TRY(ctx, hipMemsetAsync(dTop, 0xFF, (size_t)nchr * 4, ctx->stream));
            hipLaunchKernelGGL(k_wv_tops, dim3(32, (unsigned)nchr), dim3(256), 0, ctx->stream, dCounts, dOff, dTop);
            std::vector<int32_t> top((size_t)nchr, -1);
            CANVAS_HIP_TRY(ctx, hipMemcpyAsync(top.data(), dTop, (size_t)nchr * 4, hipMemcpyDeviceToHost, ctx->stream));
            CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            for (int c = 0; c < nchr; c++) if (top[(size_t)c] >= 0) { hCounts[(size_t)c].resize((size_t)top[(size_t)c] + 1); CANVAS_HIP_TRY(ctx, hipMemcpyAsync(hCounts[(size_t)c].data(), dCounts + off[c], ((size_t)top[(size_t)c] + 1) * 4, hipMemcpyDeviceToHost, ctx->stream)); }
            CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        }
        std::vector<WvCand> hc((size_t)hN[0]);
        if (hN[0]) CANVAS_HIP_TRY(ctx, hipMemcpy(hc.data(), dCands, hc.size() * sizeof(WvCand), hipMemcpyDeviceToHost));
        for (int c = 0; c < nchr; c++) {
            ChromTree& T = trees[c];
            const std::vector<int32_t>& cc = hCounts[(size_t)c];
            const int64_t top = (int64_t)cc.size() - 1;          // (its last entry is the highest level that has a node)
            if (top >= (int64_t)T.counts.size()) T.counts.resize((size_t)top + 1, 0);
            for (int64_t lv = 0; lv <= top; lv++) T.counts[(size_t)lv] += cc[(size_t)lv];
            levels = std::max<long long>(levels, (long long)T.counts.size());
        }
        for (const WvCand& k : hc) trees[k.chrom].cands.push_back({k.level, k.s, k.b, k.e, k.coef});
        // the reference's order: level by level, inside a level from left to right
        for (int c = 0; c < nchr; c++)
            std::sort(trees[c].cands.begin(), trees[c].cands.end(), [](const Cand& a, const Cand& b) { return a.level != b.level ? a.level < b.level : a.s < b.s; });
    }
    ctx->wv_levels = levels; ctx->wv_redone = redone;
    const double t2 = now();
    if (timing) fprintf(stderr, "canvas_wavelets: prefix + roots %.4f s after variability\n", t2 - t1);
    // ---- per chromosome: HardThresh, reconstruction, healing, refinement (WaveletSegmentation.cs:73-250, 373-425); the chromosomes are independent (the reference runs
    // them under Parallel.ForEach, WaveletsRunner.cs:115-135): one task per chromosome on a few host threads, results concatenated in chromosome order
    if (f3Thread.joinable()) f3Thread.join();
    rc = f3_enqueue(); if (rc) return rc;                        // (a call that never entered the level loop)
    if (f3OnDevice) {
        CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->wv_sub2));
        for (int e = 1; e <= f3Levels; e++) { int32_t rcm = cvx_mail_await(ctx, hF3Seq + (e - 1), f3Seq[e - 1], "canvas_wavelets: factor-of-three medians"); if (rcm) return rcm; }
        auto dec = [](unsigned long long k) -> double { if (k == 0ull) return std::numeric_limits<double>::quiet_NaN(); const unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k; double v; memcpy(&v, &b, 8); return v; };
        std::vector<double> g{0.0};
        for (int e = 1; e <= f3Levels; e++) { const double lo = dec(hF3Keys[2 * (e - 1)]), hi = dec(hF3Keys[2 * (e - 1) + 1]); g.push_back((f3Count[e - 1] & 1) ? hi : (lo + hi) / 2); }
        { const double last = g.back(); while ((int)g.size() < 9) g.push_back(last); }
        if (f3Check) {
            bool same = g.size() == f3.size();
            for (size_t i = 0; same && i < g.size(); i++) same = (g[i] != g[i] && f3[i] != f3[i]) || g[i] == f3[i];
            if (!same) CANVAS_FAIL(ctx, CANVAS_ERR_HIP, "canvas_wavelets: the device's factor-of-three statistics differ from the host's");
        }
        f3.swap(g);
    }
    if (timing) fprintf(stderr, "canvas_wavelets: waited %.4f s for the factor-of-three statistics (%.4f s on their thread)\n", now() - t2, f3Seconds);
    std::vector<std::vector<int>> bpOf((size_t)nchr), prelimOf((size_t)nchr);
    std::vector<int64_t> segBase((size_t)nchr + 1, 0); std::vector<double> segMed; bool segOnDevice = false;      // medians of [prelim[i], prelim[i + 1]) per chromosome, from the device
    auto finishChrom = [&](int c, int phase) {               // phase 0: thresholds, reconstruction, preliminary breakpoints; phase 1: healing, refinement
        const ChromTree& T = trees[c];
        const int64_t L = off[c + 1] - off[c];
        if (T.counts.empty()) return;                         // not segmented: no breakpoints (WaveletsRunner.cs:113-131)
        const double* r = X.data() + off[c];
        if (phase == 1) {
            const std::vector<int>& prelim = prelimOf[(size_t)c];
            // GetBreakpointsAfterHealingBadSplits
            std::vector<int> bp{prelim[0]};
            const int Lp = (int)prelim.size();
            int cachedStart = -1; double cachedMedian = 0;         // the median of [cachedStart, prelim[i]): the right segment of the previous step is the left one of this step when its breakpoint was kept
            for (int i = 1; i < Lp; ++i) {
                const int leftStart = bp.back(), rightStart = prelim[i], rightEnd = (i < Lp - 1) ? prelim[i + 1] : (int)L;
                const int leftLength = rightStart - leftStart, rightLength = rightEnd - rightStart;
                const double* sm = segOnDevice ? segMed.data() + segBase[(size_t)c] : nullptr;
                const double leftMedian = leftStart == cachedStart ? cachedMedian : (sm && leftStart == prelim[i - 1] ? sm[i - 1] : median_range(r, leftStart, leftStart + leftLength));
                const double rightMedian = sm ? sm[i] : median_range(r, rightStart, rightStart + rightLength);
                cachedStart = rightStart; cachedMedian = rightMedian;
                const double weightedMedian = (leftLength * leftMedian + rightLength * rightMedian) / (rightEnd - leftStart);
                const int smaller = std::min(leftLength, rightLength);
                const int scale = std::min((int)f3.size() - 1, (int)std::ceil(std::log((double)smaller) / std::log(3.0)));
                if (std::fabs(leftMedian - rightMedian) > f3[scale] * 4 * std::max(weightedMedian, 50.0)) bp.push_back(prelim[i]);
            }
            if (is_germline) {                                    // RefineSegments
                const double totalMedian = median_range(r, 0, L);
                for (int i = 1; i < (int)bp.size() - 1; i++) {
                    const int li = std::min(5, (bp[i] - bp[i - 1]) / 2), ri = std::min(5, (bp[i + 1] - bp[i]) / 2);
                    double best = std::fabs(median_range(r, bp[i - 1], bp[i]) - totalMedian);
                    int bestBp = bp[i];
                    for (int j = bp[i] - li; j < bp[i] + ri; j++) {
                        const double t = std::fabs(median_range(r, bp[i - 1], j) - totalMedian);
                        if (t > best) { best = t; bestBp = j; }
                    }
                    bp[i] = bestBp;
                }
            }
            bpOf[(size_t)c] = std::move(bp);
            return;
        }
        const int treeSize = (int)T.counts.size();
        std::vector<double> thresholds((size_t)treeSize, 1.0);
        std::vector<int> indices((size_t)treeSize);
        for (int i = 0; i < treeSize; i++) indices[i] = i;
        if (is_germline) {
            dotnet_sort_ints(indices, [&](int a, int b) { return T.counts[b] < T.counts[a] ? -1 : (T.counts[b] > T.counts[a] ? 1 : 0); });
            for (int x = 1; x <= treeSize; x++) thresholds[x - 1] = ((double)x * (1.0 - 0.8)) / treeSize + 0.8;
        }
        const double n = (double)L;
        double smooth = 0;
        for (int64_t i = 0; i < L; i++) smooth += r[i];
        smooth = smooth / std::sqrt(n);
        // GetReconstructedVector + GetSegments (WaveletSegmentation.cs:166-204): rec starts as one constant and every surviving node adds one value to its left part and one
        // to its right part, so rec is piecewise constant with at most three new piece boundaries per survivor.  The pieces are kept instead of the L elements (a WGS chromosome
        // spent 5 ms adding constants to 380 000 doubles per survivor): every piece receives exactly the additions, in the same order, that each of its elements would.
        // nodes whose coefficient is zeroed add +-0 to rec and cannot create or remove a difference: only the survivors are applied, in the reference's order (level by level,
        // left to right)
        std::vector<int> prelim{0};
        {
            std::map<int64_t, double> piece;                  // start of a piece -> its value
            piece[0] = 1.0 / std::sqrt(n) * smooth;
            auto split = [&](int64_t p) { if (p <= 0 || p >= L) return; auto it = piece.upper_bound(p); --it; if (it->first != p) piece.emplace_hint(std::next(it), p, it->second); };
            bool finite = true;
            for (const Cand& k : T.cands) {
                if (std::fabs(k.coef) <= 2 * T.sigma * (thresholds[indices[k.level]]) * std::sqrt(2 * std::log(n))) continue;
                const double nn = (double)k.e - (double)k.s + 1, m = (double)k.b - (double)k.s + 1;
                const double val1 = std::sqrt(1 / m - 1 / nn), val2 = -1.0 / std::sqrt(nn * nn / m - nn);
                const int64_t a = k.s - 1, b = k.b, c = k.e;  // elements [a, b) take val1 * coef, [b, c) val2 * coef
                split(a); split(b); split(c);
                for (auto it = piece.lower_bound(a); it != piece.end() && it->first < c; ++it) { it->second = it->second + (it->first < b ? val1 : val2) * k.coef; if (!std::isfinite(it->second)) finite = false; }
            }
            if (finite) {
                double prev = 0; bool first = true;
                for (const auto& kv : piece) { if (!first && kv.second - prev != 0) prelim.push_back((int)kv.first); prev = kv.second; first = false; }
            } else {                                          // (a difference of infinities is NaN at EVERY element of a piece: the element-wise form decides)
                std::vector<double> rec((size_t)L, 1.0 / std::sqrt(n) * smooth);
                for (const Cand& k : T.cands) {
                    if (std::fabs(k.coef) <= 2 * T.sigma * (thresholds[indices[k.level]]) * std::sqrt(2 * std::log(n))) continue;
                    const double nn = (double)k.e - (double)k.s + 1, m = (double)k.b - (double)k.s + 1;
                    const double val1 = std::sqrt(1 / m - 1 / nn), val2 = -1.0 / std::sqrt(nn * nn / m - nn);
                    for (int32_t i = k.s - 1; i < k.e; i++) rec[i] = rec[i] + ((double)(i - (k.s - 1)) < m ? val1 : val2) * k.coef;
                }
                for (int64_t i = 1; i < L; i++) if (rec[i] - rec[i - 1] != 0) prelim.push_back((int)i);
            }
        }
        prelimOf[(size_t)c] = std::move(prelim);
    };
    auto run_phase = [&](int phase) {
        std::atomic<int> next{0};
        const int nth = (int)std::max(1u, std::min(16u, std::min((unsigned)nchr, std::thread::hardware_concurrency())));
        auto worker = [&]() { for (int c = next.fetch_add(1); c < nchr; c = next.fetch_add(1)) finishChrom(c, phase); };
        std::vector<std::thread> pool;
        for (int t = 1; t < nth; t++) pool.emplace_back(worker);
        worker();
        for (auto& t : pool) t.join();
    };
    rc = need_X(); if (rc) return rc;
    run_phase(0);
    {   // the medians between consecutive preliminary breakpoints, all chromosomes in one launch (one workgroup per stretch); CANVAS_WV_HEAL_HOST=1: on the host as before
        std::vector<unsigned long long> segs;
        for (int c = 0; c < nchr; c++) {
            segBase[(size_t)c] = (int64_t)segs.size();
            const std::vector<int>& pr = prelimOf[(size_t)c]; const int64_t L = off[c + 1] - off[c];
            for (size_t i = 0; i < pr.size(); i++) { const int64_t a = pr[i], b = i + 1 < pr.size() ? pr[i + 1] : L; segs.push_back(((unsigned long long)(off[c] + a) << 32) | (unsigned long long)(b - a)); }
        }
        segBase[(size_t)nchr] = (int64_t)segs.size();
        bool okLen = true; for (unsigned long long v : segs) if ((v & 0xFFFFFFFFull) == 0) okLen = false;
        if (!segs.empty() && okLen && segs.size() <= varCap && !cvx_hook("CANVAS_WV_HEAL_HOST")) {
            segMed.assign(segs.size(), 0.0);
            if (closedForm && segs.size() <= WV_MED_MAXR && !cvx_hook("CANVAS_WV_MEDIAN_PER_WG")) {      // (closedForm: every bin has its integer)
                std::vector<long long> st(segs.size()), ln(segs.size());
                for (size_t k = 0; k < segs.size(); k++) { st[k] = (long long)(segs[k] >> 32); ln[k] = (long long)(segs[k] & 0xFFFFFFFFull); }
                rc = medians_enqueue(st, ln, dSegMed); if (rc) return rc;
            } else {
                CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dVarStart, segs.data(), segs.size() * 8, hipMemcpyHostToDevice, ctx->stream));
                hipLaunchKernelGGL(k_wv_segment_median, dim3((unsigned)segs.size()), dim3(1024), 0, ctx->stream, dX, (const unsigned long long*)dVarStart, dSegMed);
            }
            CANVAS_HIP_TRY(ctx, hipMemcpyAsync(segMed.data(), dSegMed, segs.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
            CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            CANVAS_HIP_TRY(ctx, hipGetLastError());
            segOnDevice = true;
            if (cvx_hook("CANVAS_WV_VAR_CHECK")) for (size_t k = 0; k < segs.size(); k++) {
                const int64_t a = (int64_t)(segs[k] >> 32), n = (int64_t)(segs[k] & 0xFFFFFFFFull);
                const double h = median_range(X.data(), a, a + n);
                if (memcmp(&h, &segMed[k], 8) != 0) CANVAS_FAIL(ctx, CANVAS_ERR_HIP, "canvas_wavelets: the device's segment medians differ from the host's");
            }
        }
    }
    run_phase(1);
    int64_t total = 0;
    for (int c = 0; c < nchr; c++) {
        h_bp_offset[c] = total;
        if (total + (int64_t)bpOf[(size_t)c].size() > cap) CANVAS_FAIL(ctx, CANVAS_ERR_CAPACITY, "canvas_wavelets: breakpoint capacity too small");
        for (int v : bpOf[(size_t)c]) h_breakpoints[total++] = v;
    }
    h_bp_offset[nchr] = total;
    if (timing) fprintf(stderr, "canvas_wavelets: %.4f s from the entry to the host copy of the coverage\n", t0 - tEntry);
    if (timing) fprintf(stderr, "canvas_wavelets: variability %.3f s, decomposition %.3f s (%lld levels), thresholds/reconstruction/healing %.3f s\n", t1 - t0, t2 - t1, levels, now() - t2);
    return CANVAS_OK;
}

extern "C" int32_t canvas_wavelets_stats(canvas_ctx* ctx, int64_t* h_out2) {
    if (!ctx || !h_out2) return CANVAS_ERR_INVALID;
    h_out2[0] = ctx->wv_levels; h_out2[1] = ctx->wv_redone;
    return CANVAS_OK;
}
extern "C" int32_t canvas_wavelets_decisions(canvas_ctx* ctx, int64_t* h_out4) {
    if (!ctx || !h_out4) return CANVAS_ERR_INVALID;
    for (int i = 0; i < 4; i++) h_out4[i] = ctx->wv_stats[i];
    return CANVAS_OK;
}
