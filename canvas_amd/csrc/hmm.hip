// CanvasPartition PerSampleHMM on MI355X: HiddenMarkovModelsRunner.Run (HiddenMarkovModelsRunner.cs:23-109),
// NegativeBinomialWrapper (CanvasCommon/DistributionUtilities.cs:51-69), BestPathViterbi (HMM.cs:62-130) and the
// state-path -> segment-id step (Segmentation.cs:83-125, SegmentationResultsProcessor.cs:17-129).
//
// What runs where
//   device  per-sample genome-wide quartiles of (float)coverage: exact order statistics (select.hpp)
//   host    the 5 negative-binomial emission tables (<= a few thousand doubles, libm log/exp/pow/lgamma — the reference calls the
//           platform libm too, Q13) -> log tables uploaded once
//   device  k_hmm_index    coverage -> table index: Convert.ToInt32(min(x, 5*haploidMean)), round-half-even   [parallel]
//   device  k_viterbi      one wave per chromosome; lanes 0..4 own states j=0..4.  The recurrence is evaluated in the
//                          reference's sequential order and association: tmp = delta[i] + (logpmf_j(x_t) + logA[i][j]),
//                          strict '>' scan i=0..4 from -DBL_MAX.  A max-plus parallel scan would re-associate the double
//                          additions and can flip near-tie argmaxes, so the time loop stays sequential (latency-bound,
//                          not HBM-bound: 16 B/bin algorithmic traffic).  Emissions for 64 steps are staged in LDS by all
//                          64 lanes in parallel; back-pointers are written as bytes.
//   device  k_backtrack_*  back-pointer chasing as an exact function-composition scan over 256-step blocks [parallel]
//   device  k_seg_flags / k_seg_ids   break points + inter-bin-distance rule -> running segment id (prefix sum)
#include "common.hpp"
#include <type_traits>
#include "select.hpp"
#include "quantize.hpp"
#include <mutex>
#include <cmath>
#include <thread>
#include <cstring>
#include <limits>
#include <algorithm>

#define NSTATE 5

struct HmmParams {
    double logA[NSTATE][NSTATE];   // log(transition[i][j])
    double logPi[NSTATE];
    int32_t tableLen;              // entries per state in logPmf
    double maxThreshold;
};

__global__ void __launch_bounds__(256) k_keys_cov_f32(const double* __restrict__ cov, int64_t n, uint32_t* __restrict__ keys) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) keys[i] = key_of_float((float)cov[i]);     // (float)x, HiddenMarkovModelsRunner.cs:43
}

// ---- genome-wide quartiles of the coverage by counting (HiddenMarkovModelsRunner.cs:36-50 takes them from a sorted copy)
// The coverage CanvasPartition reads is the F2 text of CanvasClean (IO.cs:21,40): every value is k / 100 for an integer k, and (float)x is monotone in x, so the order
// statistics of (float)coverage are (float)(k / 100) for the order statistics k of the integers.  One sweep counts the k of a window of CQ_WIN values around the sample's
// level in LDS (the quartiles of a sample lie within a few units of its median) and the elements below the window; a one-workgroup pick reads the ranks off the counters.
// Every element is checked ((double)k / 100 == x, bit for bit); a value that is not of that form, or a rank outside the window, hands the call to the radix select.
#define CQ_WIN 8192
struct CovQ { unsigned long long rank[8]; unsigned long long below; uint32_t nq, bad, fail, pad; int32_t lo; int32_t resultK[8]; long long n; };
// Utilities.Quartiles index logic (the host's quartile_idx below, on the device: the number of bins is only known there when the quantisation is enqueued behind CanvasClean)
__device__ inline void quartile_idx_dev(long long n, unsigned long long* idx, uint32_t& cnt) {
    cnt = 0;
    const long long mid = n / 2;
    if (n % 2 == 0) {
        const long long mm = mid / 2;
        idx[cnt++] = mid - 1; idx[cnt++] = mid;
        if (mid % 2 == 0) { idx[cnt++] = mm - 1; idx[cnt++] = mm; idx[cnt++] = mid + mm - 1; idx[cnt++] = mid + mm; }
        else { idx[cnt++] = mm; idx[cnt++] = mm + mid; }
    } else {
        idx[cnt++] = mid;
        if ((n - 1) % 4 == 0) { const long long k = (n - 1) / 4; idx[cnt++] = k - 1; idx[cnt++] = k; idx[cnt++] = 3 * k; idx[cnt++] = 3 * k + 1; }
        else { const long long k = (n - 3) / 4; idx[cnt++] = k; idx[cnt++] = k + 1; idx[cnt++] = 3 * k + 1; idx[cnt++] = 3 * k + 2; }
    }
}
__device__ __forceinline__ bool covq_key(double x, long long& k) {
    k = llrint(x * 100.0);
    return x == x && k >= 0 && k < (1ll << 30) && (double)k / 100.0 == x && !(k == 0 && __double2hiint(x) < 0);     // (-0.0 has its own place in the sorted order)
}
__global__ void __launch_bounds__(1024) k_covq_hist(const double* __restrict__ cov, int64_t n, CovQ* __restrict__ Q, uint32_t* __restrict__ win) {
    __shared__ uint32_t lw[CQ_WIN];
    __shared__ long long sv[33];
    __shared__ int sLo;
    __shared__ uint32_t sBelow;
    // the sample's level: median of 33 strided elements (the same in every workgroup); 33 lanes fetch them, one sorts
    if (threadIdx.x < 33) { const int64_t i = (int64_t)((double)n * (threadIdx.x + 0.5) / 33.0); long long k = -1; if (i < n && !covq_key(cov[i], k)) k = -1; sv[threadIdx.x] = k; }
    for (int i = threadIdx.x; i < CQ_WIN; i += 1024) lw[i] = 0;
    if (threadIdx.x == 0) sBelow = 0;
    __syncthreads();
    if (threadIdx.x < 33) {                                  // every lane ranks its own sample among the valid ones; the one in the middle sets the window
        const long long mine = sv[threadIdx.x];
        int m = 0, rank = 0;
        for (int j = 0; j < 33; j++) { const long long o = sv[j]; if (o >= 0) { m++; if (o < mine || (o == mine && j < (int)threadIdx.x)) rank++; } }
        if (m == 0) { if (threadIdx.x == 0) sLo = 0; }
        else if (mine >= 0 && rank == m / 2) sLo = (int)(mine > CQ_WIN / 2 ? mine - CQ_WIN / 2 : 0);
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) Q->lo = sLo;
    const long long lo = sLo;
    uint32_t below = 0, bad = 0;
    const int64_t stride = (int64_t)gridDim.x * 1024;
    for (int64_t i0 = (int64_t)blockIdx.x * 1024 + threadIdx.x; i0 < n; i0 += 4 * stride) {
        double x[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int64_t i = i0 + u * stride; x[u] = i < n ? cov[i] : 0.0; }          // four loads in flight
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (i0 + u * stride >= n) break;
            long long k;
            if (!covq_key(x[u], k)) { bad = 1; continue; }
            if (k < lo) below++;
            else if (k - lo < CQ_WIN) atomicAdd(&lw[k - lo], 1u);
        }
    }
    below = wave_reduce_add_u32(below);                      // (summed over the workgroup first: 4 096 waves adding to ONE word are performed one after the other at the memory side)
    if ((threadIdx.x & 63) == 0 && below) atomicAdd(&sBelow, below);
    if (bad) Q->bad = 1u;
    __syncthreads();
    if (threadIdx.x == 0 && sBelow) atomicAdd(&Q->below, (unsigned long long)sBelow);
    for (int i = threadIdx.x; i < CQ_WIN; i += 1024) { const uint32_t v = lw[i]; if (v) atomicAdd(&win[i], v); }
}
// Qres == NULL: the ranks were written by the host, the result stays in Q.  Qres != NULL (pipeline): the ranks are derived here from the number of bins (n, or *nDev when the
// host does not know it yet), the result goes to Qres and Q / win are left all zero for the next call (the counters live in the context, nothing is uploaded or cleared per call).
__global__ void __launch_bounds__(1024) k_covq_pick(CovQ* __restrict__ Q, uint32_t* __restrict__ win, long long n = -1, const unsigned long long* __restrict__ nDev = nullptr, CovQ* __restrict__ Qres = nullptr, unsigned seq = 0) {
    __shared__ unsigned long long sTot[16];
    __shared__ unsigned long long sRank[8];
    __shared__ uint32_t sNq, sFail;
    __shared__ int32_t sRes[8];
    const int PER = CQ_WIN / 1024;
    if (Qres) {
        if (nDev) n = (long long)*nDev < n ? (long long)*nDev : n;
        if (threadIdx.x == 0) { sFail = 0; uint32_t c = 0; if (n >= 5) quartile_idx_dev(n, sRank, c); else sFail = 1; sNq = c; }
        if (threadIdx.x < 8) sRes[threadIdx.x] = 0;
    } else if (threadIdx.x == 0) { sNq = Q->nq; sFail = 0; for (uint32_t j = 0; j < Q->nq && j < 8; j++) sRank[j] = Q->rank[j]; }
    uint32_t c[PER]; unsigned long long mine = 0;
    for (int k = 0; k < PER; k++) { c[k] = win[threadIdx.x * PER + k]; mine += c[k]; }
    unsigned long long inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long o = __shfl_up(inc, d, 64); if ((int)(threadIdx.x & 63) >= d) inc += o; }
    if ((threadIdx.x & 63) == 63) sTot[threadIdx.x >> 6] = inc;
    __syncthreads();
    const unsigned long long below = Q->below; const int32_t lo = Q->lo;
    unsigned long long before = below, total = below;
    for (int w = 0; w < 16; w++) { if (w < (int)(threadIdx.x >> 6)) before += sTot[w]; total += sTot[w]; }
    before += inc - mine;
    const uint32_t nq = sNq;
    for (uint32_t j = 0; j < nq; j++) {
        const unsigned long long want = sRank[j];
        if (threadIdx.x == 0 && (want < below || want >= total)) { if (Qres) sFail = 1; else Q->fail = 1u; }          // a quartile outside the window
        if (want >= before && want < before + mine) {
            unsigned long long cum = before;
            for (int k = 0; k < PER; k++) { cum += c[k]; if (want < cum) { const int32_t r = lo + (int)threadIdx.x * PER + k; if (Qres) sRes[j] = r; else Q->resultK[j] = r; break; } }
        }
    }
    if (!Qres) return;
    __syncthreads();
    for (int k = 0; k < PER; k++) win[threadIdx.x * PER + k] = 0u;
    if (threadIdx.x == 0) {
        CovQ o; memset(&o, 0, sizeof o);
        o.nq = nq; o.bad = Q->bad; o.fail = sFail; o.lo = lo; o.below = below; o.n = n;
        for (uint32_t j = 0; j < nq; j++) { o.rank[j] = sRank[j]; o.resultK[j] = sRes[j]; }
        // (o.pad = 0: the mailbox stamp lives in the block's spare word; the host armed it with 0 and waits for the number the line below stores)
        *Qres = o; cvx_mail_publish(&Qres->pad, seq);      // (pinned host memory, read behind a synchronisation: common.hpp)
        CovQ z; memset(&z, 0, sizeof z); *Q = z;
    }
}

// The same count inside the sweep that produces the coverage (pipeline.hip): cov[i] = F2(count[i]) is written and counted in one pass — the value is N / 100 by construction,
// N comes out of the digit arithmetic — so PerSampleHMM starts without a sweep of its own (37 us for a WGS sample) and without its own round trip for the six ranks.
#define QC_U 8               // counts a thread of k_quant_covq has in flight per round
__global__ void __launch_bounds__(1024) k_quant_covq(const float* __restrict__ count, int64_t n, double* __restrict__ cov, CovQ* __restrict__ Q, uint32_t* __restrict__ win, const unsigned long long* __restrict__ nDev = nullptr) {
    __shared__ uint32_t lw[CQ_WIN];
    if (nDev) n = (int64_t)*nDev < n ? (int64_t)*nDev : n;        // (enqueued behind CanvasClean: its bin count is still on the device)
    __shared__ long long sv[33];
    __shared__ int sLo;
    __shared__ uint32_t sBelow;
    // the first round of counts is requested before the level of the sample is worked out (two barriers and a dependent load: the loads travel meanwhile)
    const int64_t stride = (int64_t)gridDim.x * 1024;
    int64_t i0 = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    float c[QC_U];
#pragma unroll
    for (int u = 0; u < QC_U; u++) { const int64_t i = i0 + u * stride; c[u] = i < n ? count[i] : 0.0f; }
    if (threadIdx.x < 33) { const int64_t i = (int64_t)((double)n * (threadIdx.x + 0.5) / 33.0); long long k = -1; if (i < n) (void)quantize_f2_one(count[i], &k); sv[threadIdx.x] = k; }
    for (int i = threadIdx.x; i < CQ_WIN; i += 1024) lw[i] = 0;
    if (threadIdx.x == 0) sBelow = 0;
    __syncthreads();
    if (threadIdx.x < 33) {                                  // every lane ranks its own sample among the valid ones; the one in the middle sets the window
        const long long mine = sv[threadIdx.x];
        int m = 0, rank = 0;
        for (int j = 0; j < 33; j++) { const long long o = sv[j]; if (o >= 0) { m++; if (o < mine || (o == mine && j < (int)threadIdx.x)) rank++; } }
        if (m == 0) { if (threadIdx.x == 0) sLo = 0; }
        else if (mine >= 0 && rank == m / 2) sLo = (int)(mine > CQ_WIN / 2 ? mine - CQ_WIN / 2 : 0);
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) Q->lo = sLo;
    const long long lo = sLo;
    uint32_t below = 0, bad = 0;
    while (i0 < n) {
        const int64_t nx = i0 + QC_U * stride;
        float cn[QC_U];
#pragma unroll
        for (int u = 0; u < QC_U; u++) { const int64_t i = nx + u * stride; cn[u] = i < n ? count[i] : 0.0f; }      // the next round travels while this one is worked on
#pragma unroll
        for (int u = 0; u < QC_U; u++) {
            const int64_t i = i0 + u * stride;
            if (i >= n) break;
            long long k;
            cov[i] = quantize_f2_one(c[u], &k);
            if (k < 0) { bad = 1; continue; }                // NaN, a negative count, a count beyond the table: the radix select decides (as for any coverage that is not F2 text)
            if (k < lo) below++;
            else if (k - lo < CQ_WIN) atomicAdd(&lw[k - lo], 1u);
        }
#pragma unroll
        for (int u = 0; u < QC_U; u++) c[u] = cn[u];
        i0 = nx;
    }
    below = wave_reduce_add_u32(below);                      // (summed over the workgroup first: 4 096 waves adding to ONE word are performed one after the other at the memory side)
    if ((threadIdx.x & 63) == 0 && below) atomicAdd(&sBelow, below);
    if (bad) Q->bad = 1u;
    __syncthreads();
    if (threadIdx.x == 0 && sBelow) atomicAdd(&Q->below, (unsigned long long)sBelow);
    for (int i = threadIdx.x; i < CQ_WIN; i += 1024) { const uint32_t v = lw[i]; if (v) atomicAdd(&win[i], v); }
}

// RemoveOutliers (HiddenMarkovModelsRunner.cs:154-162) + Convert.ToInt32 (Distributions.cs:271)
__global__ void __launch_bounds__(256) k_hmm_index(const double* __restrict__ cov, int64_t n, double maxThreshold, int32_t* __restrict__ idx) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double x = cov[i];
    x = x > maxThreshold ? maxThreshold : x;
    int32_t k = (int32_t)rint(x);  // round half to even
    idx[i] = k < 0 ? 0 : k;        // negative coverage would throw in the reference; never read out of bounds here
}

struct HmmChrom { int64_t begin; int64_t T; };

// broadcast lane `src`'s double to the whole wave through SGPRs (v_readlane_b32 x2: no LDS round trip)
__device__ __forceinline__ double readlane_f64(double v, int src) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// one wave per chromosome
__global__ void __launch_bounds__(64) k_viterbi(const HmmChrom* __restrict__ chroms, const int32_t* __restrict__ idx, const double* __restrict__ logPmf,
                                                HmmParams P, uint16_t* __restrict__ psi /* 5 x 3 bits per bin */, int32_t* __restrict__ lastState,
                                                const int32_t* __restrict__ chromList) {
    extern __shared__ double sTab[];                 // [5][tableLen] when it fits, else unused
    __shared__ double sE[2][64 * NSTATE];            // emissions of the current / next 64-step block
    __shared__ uint8_t sPsi[NSTATE][64];             // back-pointers of the current block, flushed coalesced
    const int cidx = chromList ? chromList[blockIdx.x] : (int)blockIdx.x;
    const HmmChrom C = chroms[cidx];
    const int l = threadIdx.x;
    const bool useLds = P.tableLen * NSTATE * 8 <= 48 * 1024;
    if (useLds) { for (int i = l; i < P.tableLen * NSTATE; i += 64) sTab[i] = logPmf[i]; }
    __syncthreads();
    if (C.T <= 10) { if (l == 0) lastState[cidx] = -1; return; }     // chromosome skipped (HiddenMarkovModelsRunner.cs:69)
    const double* tab = useLds ? sTab : logPmf;
    const int j = l < NSTATE ? l : 0;
    double la[NSTATE];
#pragma unroll
    for (int i = 0; i < NSTATE; i++) la[i] = P.logA[i][j];
    const double NEG = -1.7976931348623157e308;      // Double.MinValue
    double delta = 0;
    const int32_t* ix = idx + C.begin;
    // stage block 0; keep the table index of block 1 in a register (its global load is issued a whole block ahead)
    { int64_t t = l; if (t < C.T) { int k = ix[t];
#pragma unroll
        for (int s = 0; s < NSTATE; s++) sE[0][l * NSTATE + s] = tab[s * P.tableLen + k]; } }
    int kNext = (64 + l < C.T) ? ix[64 + l] : 0;
    __syncthreads();
    int buf = 0;
    for (int64_t t0 = 0; t0 < C.T; t0 += 64, buf ^= 1) {
        // all 64 lanes: table look-ups for the NEXT 64 steps from the index loaded during the previous block, and the
        // global load of the index for the block after that (both overlap with the sequential part of this block)
        { int64_t t = t0 + 64 + l; if (t < C.T) {
#pragma unroll
            for (int s = 0; s < NSTATE; s++) sE[buf ^ 1][l * NSTATE + s] = tab[s * P.tableLen + kNext]; }
          int64_t t2 = t0 + 128 + l; kNext = (t2 < C.T) ? ix[t2] : 0; }
        const int steps = (int)((C.T - t0) < 64 ? (C.T - t0) : 64);
        const double* E = sE[buf];
        int s0 = 0;
        if (t0 == 0) {
            // bestScore[0][j] = log(pi_j) + EstimateViterbiLikelihood(x0, j, transition[0]) - log(transition[0][j])   (HMM.cs:78)
            double lik = E[j] + P.logA[0][j];
            delta = P.logPi[j] + lik - P.logA[0][j];
            if (l < NSTATE) sPsi[j][0] = 0;
            s0 = 1;
        }
        double e = E[s0 * NSTATE + j];
        for (int s = s0; s < steps; s++) {
            double eNext = E[(s + 1 < 64 ? s + 1 : s) * NSTATE + j];        // prefetch (off the dependent chain)
            double v0 = e + la[0], v1 = e + la[1], v2 = e + la[2], v3 = e + la[3], v4 = e + la[4];   // Math.Log(pmf) + Math.Log(transition)
            double t_0 = readlane_f64(delta, 0) + v0;                      // bestScore[t-1][i] + vitLogL
            double t_1 = readlane_f64(delta, 1) + v1;
            double t_2 = readlane_f64(delta, 2) + v2;
            double t_3 = readlane_f64(delta, 3) + v3;
            double t_4 = readlane_f64(delta, 4) + v4;
            // strict '>' scan i = 0..4 from Double.MinValue == first index of the maximum (no NaNs can occur), as a tree
            double a = t_0; int ia = 0; if (t_1 > a) { a = t_1; ia = 1; }
            double b = t_2; int ib = 2; if (t_3 > b) { b = t_3; ib = 3; }
            if (b > a) { a = b; ia = ib; }
            if (t_4 > a) { a = t_4; ia = 4; }
            if (!(a > NEG)) { a = NEG; ia = 0; }
            delta = a;
            if (l < NSTATE) sPsi[j][s] = (uint8_t)ia;
            e = eNext;
        }
        __syncthreads();
        // flush the block's back-pointers, packed 3 bits per state: one coalesced 128-byte row
        if (l < steps)
            psi[C.begin + t0 + l] = (uint16_t)(sPsi[0][l] | (sPsi[1][l] << 3) | (sPsi[2][l] << 6) | (sPsi[3][l] << 9) | (sPsi[4][l] << 12));
        __syncthreads();
    }
    // best final state: strict '>' scan from Double.MinValue, bestState initialised to -1 (HMM.cs:100-111)
    double d[NSTATE];
#pragma unroll
    for (int i = 0; i < NSTATE; i++) d[i] = readlane_f64(delta, i);
    if (l == 0) {
        int best = -1; double m1 = NEG;
#pragma unroll
        for (int i = 0; i < NSTATE; i++) if (d[i] > m1) { best = i; m1 = d[i]; }
        lastState[cidx] = best;
    }
}


// =====================================================================================================================
// Speculative block-parallel Viterbi with exact verification.
//
// The recurrence must be evaluated in the reference's order to be bit-identical, which makes k_viterbi a 390 k-step chain for chr1.
// But the *decisions* (back-pointers) depend only on differences of delta, and those forget their history within a few dozen
// bins.  So:
//   A  k_vit_spec      every 128-step block re-runs the SAME recurrence code from a cold start 128 steps earlier and records the
//                      back-pointers it sees (a guess: its delta differs from the true one by a block constant + rounding noise),
//                      together with the block's back-pointer map (state at the block's last step -> state before its first);
//   A2 k_bt_*          compose the block maps backwards -> guessed state path s_t (the "backbone");
//   B1 k_vit_backbone* the exact delta along the backbone is a plain sequential sum D_t = D_{t-1} + (e_{s_t}(t) + logA[s_{t-1}][s_t])
//                      with the reference's association;
//   C  k_vit_verify    every block rebuilds the exact delta of all five states from D (off-backbone states are re-anchored to the
//                      backbone inside a 64-step lead-in), then re-does the exact strict-'>' arg-max at each step and checks it
//                      against the guess, and checks delta_t(s_t) == D_t bit for bit.
// If every check passes, induction from the exact t = 0 shows the guessed pointers ARE the reference's (and so is the path); any
// failed check (a near-tie resolved differently by the cold-start run, a lead-in that did not re-anchor) flags the chromosome,
// which is then recomputed by the sequential k_viterbi.  Results are therefore always exact; speculation only buys time.
//
// Work mapping of A and C: ONE LANE PER BLOCK.  A lane keeps all five delta values of its block in registers, so a step needs no
// cross-lane traffic at all and a wave advances 64 blocks at once (the earlier one-wave-per-block version used 5 lanes of 64 and
// was issue-bound).  The emission table sits in LDS transposed to [k][5] so that a lane reads its five log-pmf values from 40
// contiguous bytes; bin indices / states / increments are fetched through a small per-lane register queue PQ steps ahead.
// Back-pointers are packed 3 bits per state into one uint16 per bin.
#define VB 128       // block length
#define VW0 16       // cold-start lead-in of the speculative pass, first attempt
#define VW 128       // ... second attempt (a run-time argument; the later ones use 8x, 64x, 512x)
#define VW2 64       // lead-in of the verification pass, first attempt (a multiple of 64: carry[] holds D at multiples of 64)
#define PQ 8         // per-lane prefetch depth in steps
#define MAP_IDENT (0u | (1u << 3) | (2u << 6) | (3u << 9) | (4u << 12))
struct VitBlock { int32_t chrom; int32_t t0; };   // chromosome-relative start
struct __attribute__((aligned(4))) VecI4 { int v[4]; };      // 16-byte load at 4-byte alignment (the per-lane streams start anywhere)
struct __attribute__((aligned(8))) VecD2 { double v[2]; };
struct __attribute__((aligned(2))) VecH8 { uint32_t w[4]; };   // eight packed 16-bit values

// new delta and back-pointer of state J:  tmp_i = delta_i + (logpmf_J(x_t) + logA[i][J]),  strict '>' scan i = 0..4 from
// Double.MinValue == first index of the maximum (evaluated as a tree, no NaNs can occur)   (HMM.cs:84-97, Distributions.cs:322)
// TWO: the transition matrix has one value on the diagonal and one off it (the HMMs of CanvasPartition: 0.99 / 0.0025) — the five sums e + logA[i][J] are then two
// different additions, each with the same operands and therefore the same result as in the reference's expression
template <int J, bool TWO = false>
__device__ __forceinline__ void vit_state(const double (&d)[NSTATE], double e, const HmmParams& P, double& outDelta, uint32_t& outArg) {
    const double NEG = -1.7976931348623157e308;
    const double cOff = e + P.logA[J == 0 ? 1 : 0][J], cDiag = e + P.logA[J][J];
    const double t_0 = d[0] + (TWO ? (J == 0 ? cDiag : cOff) : (e + P.logA[0][J]));
    const double t_1 = d[1] + (TWO ? (J == 1 ? cDiag : cOff) : (e + P.logA[1][J]));
    const double t_2 = d[2] + (TWO ? (J == 2 ? cDiag : cOff) : (e + P.logA[2][J]));
    const double t_3 = d[3] + (TWO ? (J == 3 ? cDiag : cOff) : (e + P.logA[3][J]));
    const double t_4 = d[4] + (TWO ? (J == 4 ? cDiag : cOff) : (e + P.logA[4][J]));
    // maximum first (v_max_f64), then the first index that attains it: fewer selects than carrying (value, index) through the scan
    double a = __builtin_fmax(__builtin_fmax(__builtin_fmax(t_0, t_1), __builtin_fmax(t_2, t_3)), t_4);
    uint32_t ia = t_3 == a ? 3u : 4u;
    ia = t_2 == a ? 2u : ia; ia = t_1 == a ? 1u : ia; ia = t_0 == a ? 0u : ia;
    if (!(a > NEG)) { a = NEG; ia = 0; }
    outDelta = a; outArg = ia;
}
// one full step for all five states; returns the packed back-pointers
template <bool TWO = false>
__device__ __forceinline__ uint32_t vit_step5(double (&d)[NSTATE], const double (&e)[NSTATE], const HmmParams& P) {
    double n0, n1, n2, n3, n4; uint32_t a0, a1, a2, a3, a4;
    vit_state<0, TWO>(d, e[0], P, n0, a0); vit_state<1, TWO>(d, e[1], P, n1, a1); vit_state<2, TWO>(d, e[2], P, n2, a2);
    vit_state<3, TWO>(d, e[3], P, n3, a3); vit_state<4, TWO>(d, e[4], P, n4, a4);
    d[0] = n0; d[1] = n1; d[2] = n2; d[3] = n3; d[4] = n4;
    return a0 | (a1 << 3) | (a2 << 6) | (a3 << 9) | (a4 << 12);
}
// bestScore[0][j] = log(pi_j) + EstimateViterbiLikelihood(x0, j, transition[0]) - log(transition[0][j])   (HMM.cs:78)
__device__ __forceinline__ void vit_init5(double (&d)[NSTATE], const double (&e)[NSTATE], const HmmParams& P) {
#pragma unroll
    for (int j = 0; j < NSTATE; j++) { double lik = e[j] + P.logA[0][j]; d[j] = P.logPi[j] + lik - P.logA[0][j]; }
}
__device__ __forceinline__ int vit_best5(const double (&d)[NSTATE]) {     // HMM.cs:100-111
    int best = -1; double m1 = -1.7976931348623157e308;
#pragma unroll
    for (int i = 0; i < NSTATE; i++) if (d[i] > m1) { best = i; m1 = d[i]; }
    return best;
}
__device__ __forceinline__ uint32_t map_get(uint32_t m, uint32_t j) { return (m >> (3u * j)) & 7u; }
// (first `inner`, then `outer`):  r[j] = outer[inner[j]]
__device__ __forceinline__ uint32_t map_compose(uint32_t outer, uint32_t inner) {
    return map_get(outer, map_get(inner, 0)) | (map_get(outer, map_get(inner, 1)) << 3) | (map_get(outer, map_get(inner, 2)) << 6) |
           (map_get(outer, map_get(inner, 3)) << 9) | (map_get(outer, map_get(inner, 4)) << 12);
}
__device__ __forceinline__ double sel5(const double (&d)[NSTATE], uint32_t i) {
    double r = d[0];
    r = i == 1 ? d[1] : r; r = i == 2 ? d[2] : r; r = i == 3 ? d[3] : r; r = i == 4 ? d[4] : r;
    return r;
}
// emission row of table index k: LDS copy is transposed to [k][5], the global table is [5][tableLen]
__device__ __forceinline__ void vit_emissions(double (&e)[NSTATE], const double* __restrict__ sTab, const double* __restrict__ logPmf, bool useLds, int tableLen, int k) {
    if (useLds) {
#pragma unroll
        for (int j = 0; j < NSTATE; j++) e[j] = sTab[k * NSTATE + j];
    } else {
#pragma unroll
        for (int j = 0; j < NSTATE; j++) e[j] = logPmf[(size_t)j * tableLen + k];
    }
}
__device__ __forceinline__ void vit_stage_table(double* sTab, const double* __restrict__ logPmf, int tableLen, bool useLds) {
    if (useLds) for (int i = threadIdx.x; i < tableLen * NSTATE; i += blockDim.x) { int s = i / tableLen, k = i - s * tableLen; sTab[k * NSTATE + s] = logPmf[i]; }
    __syncthreads();
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { int o = __shfl_xor(v, d); v = o > v ? o : v; }
    return v;
}

// block / chunk lists: entry i of a list with `per` steps per entry, from the per-chromosome first indices (binary search)
template <typename T>
__global__ void __launch_bounds__(256) k_make_blocks(const int32_t* __restrict__ first, int nchr, int n, int per, T* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int lo = 0, hi = nchr - 1;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (first[mid] <= i) lo = mid; else hi = mid - 1; }
    while (lo < nchr - 1 && first[lo + 1] <= i) lo++;        // chromosomes without blocks share an index
    T b; b.chrom = lo; b.t0 = (i - first[lo]) * per;
    out[i] = b;
}

// The data-independent tables of the stage in ONE launch: the three block / chunk lists (k_make_blocks), the per-chromosome result slots, and — PerSampleHMM — the
// table index of every bin (k_hmm_index: it needs nothing but the threshold).  Round 3 issued them as seven launches / fills between two host round trips.
template <typename T> __device__ __forceinline__ void make_block_entry(const int32_t* __restrict__ first, int nchr, int i, int per, T* __restrict__ out) {
    int lo = 0, hi = nchr - 1;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (first[mid] <= i) lo = mid; else hi = mid - 1; }
    while (lo < nchr - 1 && first[lo + 1] <= i) lo++;        // chromosomes without blocks share an index
    T b; b.chrom = lo; b.t0 = (i - first[lo]) * per;
    out[i] = b;
}
// the per-chromosome descriptors of the stage, BY VALUE (<= HMM_BYVAL chromosomes: the argument block is copied at launch, no H2D copy in front of the stage)
#define HMM_BYVAL 64
struct HmmDescPack { HmmChrom chroms[HMM_BYVAL]; int64_t off[HMM_BYVAL + 1]; int32_t first[HMM_BYVAL + 1], firstChunk[HMM_BYVAL + 1], firstS[HMM_BYVAL + 1], firstGroup[HMM_BYVAL + 1]; };
struct HmmDescDev { HmmChrom* chroms; int64_t* off; int32_t* first; int32_t* firstChunk; int32_t* firstS; int32_t* firstGroup; };
template <typename VB_T, typename BB_T>
__global__ void __launch_bounds__(256) k_hmm_setup(const HmmDescPack pack, int usePack, HmmDescDev D, int nchr,
                                                   int nblocks, int nblocksS, int nchunks, int vb, int vbs, int bbChunk, VB_T* __restrict__ vBlocks, VB_T* __restrict__ sBlocks, BB_T* __restrict__ bChunks,
                                                   const double* __restrict__ cov, int64_t N, double maxThreshold, int32_t* __restrict__ idx, int32_t* __restrict__ dLast, int32_t* __restrict__ dFail) {
    const int32_t* first = usePack ? pack.first : D.first; const int32_t* firstS = usePack ? pack.firstS : D.firstS; const int32_t* firstChunk = usePack ? pack.firstChunk : D.firstChunk;
    if (usePack && blockIdx.x == 0) {        // publish the tables for the kernels that follow
        for (int c = threadIdx.x; c < nchr; c += 256) D.chroms[c] = pack.chroms[c];
        for (int c = threadIdx.x; c <= nchr; c += 256) { D.off[c] = pack.off[c]; D.first[c] = pack.first[c]; D.firstChunk[c] = pack.firstChunk[c]; D.firstS[c] = pack.firstS[c]; D.firstGroup[c] = pack.firstGroup[c]; }
    }
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nblocks) make_block_entry(first, nchr, (int)i, vb, vBlocks);
    if (i < nblocksS) make_block_entry(firstS, nchr, (int)i, vbs, sBlocks);
    if (i < nchunks) make_block_entry(firstChunk, nchr, (int)i, bbChunk, bChunks);
    if (i < nchr) { dLast[i] = -1; dFail[i] = 0; }            // -1: skipped chromosomes
    if (cov && i < N) {                                       // RemoveOutliers (HiddenMarkovModelsRunner.cs:154-162) + Convert.ToInt32 (Distributions.cs:271)
        double x = cov[i];
        x = x > maxThreshold ? maxThreshold : x;
        const int32_t k = (int32_t)rint(x);
        idx[i] = k < 0 ? 0 : k;
    }
}

// A: one lane per block
template <bool useLds, bool TWO>      // compile-time: with both emission sources in one kernel every step waited for ALL outstanding memory operations
__global__ void __launch_bounds__(64) k_vit_spec(const VitBlock* __restrict__ blocks, int nblocks, const HmmChrom* __restrict__ chroms, const int32_t* __restrict__ idx,
                                                 const double* __restrict__ logPmf, HmmParams P, uint16_t* __restrict__ psi, uint16_t* __restrict__ maps,
                                                 int32_t* __restrict__ lastGuess, int leadIn, const int32_t* __restrict__ todo, int vb) {
    extern __shared__ double sTab[];
    vit_stage_table(sTab, logPmf, P.tableLen, useLds);
    const int b = blockIdx.x * 64 + threadIdx.x;
    const bool actBlock = b < nblocks;
    const VitBlock B = blocks[actBlock ? b : nblocks - 1];
    const HmmChrom C = chroms[B.chrom];
    const int64_t tBeg = B.t0, tEnd = (B.t0 + vb < C.T) ? B.t0 + vb : C.T;       // block covers [tBeg, tEnd)
    const int64_t ts = tBeg > leadIn ? tBeg - leadIn : 0;                          // cold start
    const bool act = actBlock && (!todo || todo[B.chrom]);                         // a retry only recomputes the chromosomes that failed
    const int nsteps = act ? (int)(tEnd - ts) : 0;
    const int maxSteps = wave_max_i32(nsteps);
    const int32_t* __restrict__ ix = idx + C.begin + ts;
    uint16_t* __restrict__ pp = psi + C.begin;
    double d[NSTATE] = {0.0, 0.0, 0.0, 0.0, 0.0};
    uint32_t fm = MAP_IDENT;                                                       // state at t -> state at tBeg - 1
    const int lead = (int)(tBeg - ts);                                             // steps in front of the block
    // The loop body is straight-line code: per-lane conditions (a lane past its last step, a lane still in its lead-in) select results
    // instead of branching, the first step (no history) is peeled off, the bin indices of eight steps arrive with two 16-byte loads one group
    // ahead and the eight back-pointers leave together.  With a branch per condition the compiler could no longer tell which memory
    // operations were pending and waited for ALL of them at every step (SQ_WAIT_ANY: 40 % of the wave cycles).
    if (nsteps > 0) {
        double e[NSTATE];
        vit_emissions(e, sTab, logPmf, useLds, P.tableLen, ix[0]);
        if (ts == 0) vit_init5(d, e, P);
        else {
#pragma unroll
            for (int j = 0; j < NSTATE; j++) d[j] = e[j];                          // first step of a cold start: no history
        }
        if (lead == 0) pp[ts] = 0;                                                 // only the first bin of a chromosome: it has no back-pointer
    }
    // (the loads run up to 2 * PQ elements past a lane's own range: idx is padded for that, and steps past the range are discarded)
    VecI4 qa = *reinterpret_cast<const VecI4*>(ix + 1), qb = *reinterpret_cast<const VecI4*>(ix + 5);
    double eCur[NSTATE];                                                            // emission row of the step about to run: read one step ahead,
    vit_emissions(eCur, sTab, logPmf, useLds, P.tableLen, 1 < nsteps ? qa.v[0] : 0);   // so that the LDS latency hides behind the previous step's arithmetic
    // one group of PQ steps; `withMaps`: some lane of the wave is past its lead-in in this group, so back-pointers are stored and composed into the block map
    // (two thirds of a lane's steps are lead-in: the groups in front of the wave's shortest lead-in run without those ~25 integer instructions per step)
    auto group = [&](int s0, auto withMaps) {
        constexpr bool WM = decltype(withMaps)::value;
        const VecI4 na = *reinterpret_cast<const VecI4*>(ix + s0 + PQ), nb = *reinterpret_cast<const VecI4*>(ix + s0 + PQ + 4);
        uint32_t pks[PQ];
#pragma unroll
        for (int u = 0; u < PQ; u++) {
            const int s = s0 + u;
            const bool on = s < nsteps;
            const int kNext = (s + 1 < nsteps) ? (u + 1 < 4 ? qa.v[(u + 1) & 3] : (u + 1 < 8 ? qb.v[(u + 1) & 3] : na.v[0])) : 0;
            double eNext[NSTATE], dn[NSTATE];
            vit_emissions(eNext, sTab, logPmf, useLds, P.tableLen, kNext);
#pragma unroll
            for (int j = 0; j < NSTATE; j++) dn[j] = d[j];
            const uint32_t pk = vit_step5<TWO>(dn, eCur, P);
#pragma unroll
            for (int j = 0; j < NSTATE; j++) { d[j] = on ? dn[j] : d[j]; eCur[j] = eNext[j]; }
            if (WM) { fm = (on && s >= lead) ? map_compose(fm, pk) : fm; pks[u] = pk; }
        }
        if (WM) {
#pragma unroll
            for (int u = 0; u < PQ; u++) { const int s = s0 + u; if (s < nsteps && s >= lead) pp[ts + s] = (uint16_t)pks[u]; }
        }
        qa = na; qb = nb;
    };
    const int leadMin = -wave_max_i32(nsteps > 0 ? -lead : -0x7FFFFFFF);           // shortest lead-in among the wave's active lanes
    int s0 = 1;
    for (; s0 + PQ <= leadMin && s0 < maxSteps; s0 += PQ) group(s0, std::false_type{});
    for (; s0 < maxSteps; s0 += PQ) group(s0, std::true_type{});
    if (act) {
        maps[b] = (uint16_t)fm;
        if (tEnd == C.T) lastGuess[B.chrom] = vit_best5(d);      // guess of the best final state (HMM.cs:100-111 on the shifted delta)
    }
}

// chromosome of global bin g: the offsets (nchr + 1 of them) are staged in LDS when they fit, so that the per-thread binary search does not
// turn into five dependent global loads.  Every thread of the block must call stage_chr_offsets before the bounds check.
#define CHR_LDS_CAP 1024
__device__ __forceinline__ const int64_t* stage_chr_offsets(const int64_t* __restrict__ chrOff, int nchr, int64_t* sOff) {
    if (nchr + 1 > CHR_LDS_CAP) return chrOff;
    for (int k = threadIdx.x; k <= nchr; k += blockDim.x) sOff[k] = chrOff[k];
    __syncthreads();
    return sOff;
}
__device__ __forceinline__ int chrom_of_bin(const int64_t* off, int nchr, int64_t g) {
    int lo = 0, hi = nchr - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (off[mid] <= g) lo = mid; else hi = mid - 1; }
    return lo;
}
// The speculative pass is latency-bound at one wave per SIMD (a second wave per SIMD is nearly free), the verification is issue-bound:
// speculation therefore runs on blocks of VBS = VB / 2 steps (twice the waves, 3/4 of the steps per lane) and the maps of the two halves
// of a VB block are composed here for the backtrack (first the later half, then the earlier one).
// (VB / 1: 167 us.)  (VB / 4 was measured too: 161 us against 140 us for the WGS sample — with two waves per SIMD the pass is issue-bound and the extra lead-in steps cost more than they hide.)
#define VBS (VB / 2)
__global__ void __launch_bounds__(256) k_pair_maps(const VitBlock* __restrict__ blocks, int nblocks, const HmmChrom* __restrict__ chroms, const int32_t* __restrict__ firstS,
                                                   const uint16_t* __restrict__ mapsS, uint16_t* __restrict__ maps, const int32_t* __restrict__ todo) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nblocks) return;
    const VitBlock B = blocks[b];
    if (todo && !todo[B.chrom]) return;
    const int64_t T = chroms[B.chrom].T;
    const int s0 = firstS[B.chrom] + B.t0 / VBS;
    uint32_t m = mapsS[s0];
#pragma unroll
    for (int k = 1; k < VB / VBS; k++) if (B.t0 + (int64_t)k * VBS < T) m = map_compose(m, mapsS[s0 + k]);
    maps[b] = (uint16_t)m;
}

// B1a: per-step increments of the exact delta along the guessed path (parallel): v_0 from HMM.cs:78,
//      v_t = logpmf_{s_t}(x_t) + logA[s_{t-1}][s_t]  (= Math.Log(emission) + Math.Log(transition), Distributions.cs:322)
__global__ void __launch_bounds__(256) k_vit_increments(const HmmChrom* __restrict__ chroms, int nchr, const int64_t* __restrict__ chrOff, const int32_t* __restrict__ idx,
                                                        const double* __restrict__ logPmf, HmmParams P, const int32_t* __restrict__ state, int64_t N, double* __restrict__ D) {
    __shared__ int64_t sOff[CHR_LDS_CAP];
    const int64_t* off = stage_chr_offsets(chrOff, nchr, sOff);
    int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= N) return;
    const int s1 = state[g], sPrev = state[g > 0 ? g - 1 : 0], ix = idx[g];          // requested together: the table lookup below is the only dependent load
    const int lo = chrom_of_bin(off, nchr, g);
    const int64_t begin = chroms[lo].begin;
    if (s1 < 0) { D[g] = 0.0; return; }
    double e = logPmf[(size_t)s1 * P.tableLen + ix];
    if (g == begin) { double lik = e + P.logA[0][s1]; D[g] = P.logPi[s1] + lik - P.logA[0][s1]; }
    else { const int s0 = sPrev; D[g] = e + P.logA[s0 < 0 ? 0 : s0][s1]; }
}

// B1b: D_t = D_{t-1} + v_t in the reference's (sequential) association.  One wave per chromosome; the increments of a 64-step
// chunk sit one per lane and are broadcast through SGPRs (v_readlane), so the dependent chain is ONE FP64 add per step.  Only the
// running sum at every chunk start is stored (carry[first bin of chunk] = D_{t-1}); k_vit_verify rebuilds D_t inside its blocks.
__global__ void __launch_bounds__(64) k_vit_backbone(const HmmChrom* __restrict__ chroms, const double* __restrict__ V, double* __restrict__ carryOut, const int32_t* __restrict__ todo = nullptr) {
    __shared__ double sV[2][64];
    const HmmChrom C = chroms[blockIdx.x];
    if (C.T <= 10 || (todo && !todo[blockIdx.x])) return;
    const int l = threadIdx.x;
    const double* __restrict__ Vc = V + C.begin;
    double acc = 0.0;
    // lanes past the end of the chromosome hold +0.0: adding it leaves acc unchanged
    double vNext = l < C.T ? Vc[l] : 0.0;
    int buf = 0;
    for (int64_t c0 = 0; c0 < C.T; c0 += 64, buf ^= 1) {
        sV[buf][l] = vNext;                                            // LDS ops of one wave execute in order: no barrier needed
        { int64_t t2 = c0 + 64 + l; vNext = t2 < C.T ? Vc[t2] : 0.0; }  // global prefetch one chunk ahead
        if (l == 0) carryOut[C.begin + c0] = acc;
        double r[64];
#pragma unroll
        for (int s = 0; s < 64; s++) r[s] = sV[buf][s];                // uniform address: LDS broadcast reads, all issued up front
#pragma unroll
        for (int s = 0; s < 64; s++) acc = acc + r[s];                 // the dependent chain: one FP64 add per step
    }
}

// B1c: the same sequential sum, evaluated exactly by a parallel scan.  All increments are <= 0, so |D| never decreases and stays
// inside one binade [2^e, 2^(e+1)) for long stretches (about 21 binades per chromosome).  Inside a binade D = -k*u with u = 2^(e-52)
// and k an integer in [2^52, 2^53); IEEE round-to-nearest-even of k*u + a is k + A (+1 when the discarded part of a/u is above one
// half, or exactly one half and k + A is odd).  Each step is therefore a function of the PARITY of k alone, "add a0 if k is even,
// a1 if it is odd", and such functions compose associatively: a block scan gives every k exactly.  The step that leaves the
// binade is done with one real FP64 add and the scan restarts behind it with the new u.  One workgroup per chromosome, 8192 steps
// per iteration.  Anything outside the assumptions (NaN, infinities, positive increments) flags the chromosome for the
// sequential k_viterbi; k_vit_verify re-checks D bit for bit in any case.
#define BS_T 1024
#define BS_I 8
struct ParFn { unsigned long long a0, a1; };     // amount added to k when the incoming k is even / odd
__device__ __forceinline__ ParFn parfn_then(ParFn f, ParFn g) {   // f first, then g
    ParFn h;
    h.a0 = f.a0 + ((f.a0 & 1ull) ? g.a1 : g.a0);
    h.a1 = f.a1 + ((f.a1 & 1ull) ? g.a0 : g.a1);
    return h;
}
__device__ __forceinline__ ParFn parfn_shfl_up(ParFn f, int d) {
    ParFn o;
    o.a0 = __shfl_up(f.a0, d); o.a1 = __shfl_up(f.a1, d);
    return o;
}
__global__ void __launch_bounds__(BS_T) k_vit_backbone_scan(const HmmChrom* __restrict__ chroms, const double* __restrict__ V, double* __restrict__ carryOut, int32_t* __restrict__ fail) {
    __shared__ ParFn sAgg[BS_T / 64];
    __shared__ long long sCross, sP;
    __shared__ double sM;
    __shared__ int sBad;
    const HmmChrom C = chroms[blockIdx.x];
    if (C.T <= 10) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* __restrict__ Vc = V + C.begin;
    double* __restrict__ carry = carryOut + C.begin;
    const unsigned long long TWO53 = 1ull << 53, MANT = (1ull << 52) - 1ull;
    if (tid == 0) { carry[0] = 0.0; sM = 0.0; sP = 0; sBad = 0; sCross = 0x7fffffffffffffffll; }
    __syncthreads();
    for (;;) {
        const long long p = sP; const double M = sM; const int bad = sBad;     // M = |D_{p-1}|
        __syncthreads();
        if (bad || p >= C.T) break;
        if (!(M >= 2.2250738585072014e-308)) {               // zero (chromosome start) or subnormal: one plain step
            if (tid == 0) {
                const double v = Vc[p];
                const unsigned long long vb = (unsigned long long)__double_as_longlong(v);
                if (((vb >> 52) & 0x7ffull) == 0x7ffull || (!(vb >> 63) && (vb << 1) != 0ull)) { sBad = 1; fail[blockIdx.x] = 1; }
                else { const double acc = -M + v; if (((p + 1) & 63) == 0 && p + 1 < C.T) carry[p + 1] = acc; sM = -acc; sP = p + 1; }
            }
            __syncthreads();
            continue;
        }
        const unsigned long long mb = (unsigned long long)__double_as_longlong(M);
        const int eM = (int)((mb >> 52) & 0x7ffull);                               // biased exponent of the binade
        const unsigned long long k0 = (mb & MANT) | (1ull << 52);
        const long long base = p + (long long)tid * BS_I;
        ParFn f[BS_I]; bool myBad = false;
#pragma unroll
        for (int i = 0; i < BS_I; i++) {
            const long long t = base + i;
            unsigned long long A = 0; int r = 0;
            if (t < C.T) {
                const unsigned long long vb = (unsigned long long)__double_as_longlong(Vc[t]);
                const int ea = (int)((vb >> 52) & 0x7ffull);
                if (ea == 0x7ff || (!(vb >> 63) && (vb << 1) != 0ull)) myBad = true;
                const unsigned long long ma = ea ? ((vb & MANT) | (1ull << 52)) : (vb & MANT);   // |v| = ma * 2^(max(ea,1) - 1075)
                const int shift = eM - (ea ? ea : 1);
                if (shift <= 0) { if (ma) A = TWO53; }                                     // |v| >= 2^e: leaves the binade
                else if (shift < 64) {
                    A = ma >> shift;
                    const unsigned long long rem = ma & ((1ull << shift) - 1ull), half = 1ull << (shift - 1);
                    r = rem > half ? 1 : (rem == half ? 2 : 0);
                }
            }
            const unsigned long long up = A + (r == 1 ? 1ull : 0ull);
            f[i].a0 = up + ((r == 2 && (A & 1ull)) ? 1ull : 0ull);
            f[i].a1 = up + ((r == 2 && !(A & 1ull)) ? 1ull : 0ull);
        }
        ParFn g = f[0];
#pragma unroll
        for (int i = 1; i < BS_I; i++) g = parfn_then(g, f[i]);
        // inclusive scan over the workgroup (thread order = step order)
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { ParFn o = parfn_shfl_up(g, d); if (lane >= d) g = parfn_then(o, g); }
        if (lane == 63) sAgg[wave] = g;
        if (myBad) sBad = 1;
        __syncthreads();
        ParFn ex; ex.a0 = 0; ex.a1 = 0;                        // exclusive prefix of this thread
        for (int w = 0; w < wave; w++) ex = parfn_then(ex, sAgg[w]);
        { ParFn o = parfn_shfl_up(g, 1); if (lane > 0) ex = parfn_then(ex, o); }
        if (sBad) { if (tid == 0) fail[blockIdx.x] = 1; break; }
        unsigned long long k = k0 + ((k0 & 1ull) ? ex.a1 : ex.a0);
        unsigned long long kAt[BS_I];
        long long myCross = 0x7fffffffffffffffll; unsigned long long kBefore = 0;
#pragma unroll
        for (int i = 0; i < BS_I; i++) {
            const unsigned long long kp = k;
            k += (k & 1ull) ? f[i].a1 : f[i].a0;
            kAt[i] = k;
            if (k >= TWO53 && myCross == 0x7fffffffffffffffll && base + i < C.T) { myCross = base + i; kBefore = kp; }
        }
        if (myCross != 0x7fffffffffffffffll) atomicMin((unsigned long long*)&sCross, (unsigned long long)myCross);
        __syncthreads();
        const long long cross = sCross;
        const long long chunkEnd = (p + (long long)BS_T * BS_I < C.T) ? p + (long long)BS_T * BS_I : C.T;
        const long long validEnd = cross < chunkEnd ? cross : chunkEnd;            // steps [p, validEnd) are exact
        const unsigned long long expBits = (unsigned long long)eM << 52;
#pragma unroll
        for (int i = 0; i < BS_I; i++) {
            const long long t = base + i;
            if (t < validEnd) {
                const double Mt = __longlong_as_double((long long)(expBits | (kAt[i] & MANT)));
                if (((t + 1) & 63) == 0 && t + 1 < C.T) carry[t + 1] = -Mt;
                if (t + 1 == validEnd && cross >= chunkEnd) { sM = Mt; sP = validEnd; }
            }
        }
        __syncthreads();
        if (cross < chunkEnd && myCross == cross) {             // the owner of the leaving step does it with a real add
            const double Mprev = __longlong_as_double((long long)(expBits | (kBefore & MANT)));
            const double acc = -Mprev + Vc[cross];
            if (((cross + 1) & 63) == 0 && cross + 1 < C.T) carry[cross + 1] = acc;
            sM = -acc; sP = cross + 1; sCross = 0x7fffffffffffffffll;
        }
        __syncthreads();
    }
}

// B1d: the parity scan without the sequential chunk loop.  A plain (re-associated) FP64 prefix sum of |v| is accurate to ~1e-13
// relative, which is enough to PREDICT the binade of every running sum and the ~21 steps per chromosome that leave a binade
// ("crossings"); the exact computation then needs no search:
//   k_bb_sums / k_bb_bases   chunk sums of |v| (1024 steps per chunk) and their per-chromosome exclusive scan          [parallel]
//   k_bb_pieces              per chunk: predicted binade of every step, its parity function under that binade, a segmented
//                            composition scan that restarts behind every crossing -> the function of every piece between
//                            crossings, plus the function up to every 64th step                                            [parallel]
//   k_bb_walk                one wave per chromosome walks chunks and crossings in order: applies the piece functions to the exact
//                            (k, e), does the ~21 crossing steps with real FP64 adds, and CHECKS every prediction (binade of
//                            each piece, no piece reaching 2^53); a wrong prediction flags the chromosome for k_viterbi
//   k_bb_emit                carry[] at the multiples of 64 from the exact piece origins                                   [parallel]
#define BB_CHUNK 1024
#define BB_MAXC 16
struct BbChunk { int32_t chrom; int32_t t0; };
struct BbCross { ParFn before; double v; int32_t pos; int32_t e; };              // piece before the crossing, the crossing's increment, predicted binade after it
struct BbChunkOut { ParFn tail; int32_t nCross; int32_t headE; int64_t pad; };    // piece from the last crossing (or chunk start) to the chunk end
struct BbPost { unsigned long long bits; };                                       // |D| right after a crossing (raw double bits)
__device__ __forceinline__ bool bb_bad_increment(unsigned long long vb) { return ((vb >> 52) & 0x7ffull) == 0x7ffull || (!(vb >> 63) && (vb << 1) != 0ull); }
__device__ __forceinline__ ParFn parfn_of(unsigned long long vb, int eM) {       // step "add |v|" inside binade eM (biased, >= 1)
    const unsigned long long MANT = (1ull << 52) - 1ull;
    const int ea = (int)((vb >> 52) & 0x7ffull);
    const unsigned long long ma = ea ? ((vb & MANT) | (1ull << 52)) : (vb & MANT);
    const int shift = eM - (ea ? ea : 1);
    unsigned long long A = 0; int r = 0;
    if (shift <= 0) { if (ma) A = 1ull << 53; }
    else if (shift < 64) { A = ma >> shift; const unsigned long long rem = ma & ((1ull << shift) - 1ull), half = 1ull << (shift - 1); r = rem > half ? 1 : (rem == half ? 2 : 0); }
    const unsigned long long up = A + (r == 1 ? 1ull : 0ull);
    ParFn f; f.a0 = up + ((r == 2 && (A & 1ull)) ? 1ull : 0ull); f.a1 = up + ((r == 2 && !(A & 1ull)) ? 1ull : 0ull);
    return f;
}
__device__ __forceinline__ double shfl_up_f64(double v, int d) { return __hiloint2double(__shfl_up(__double2hiint(v), d), __shfl_up(__double2loint(v), d)); }

__global__ void __launch_bounds__(256) k_bb_sums(const BbChunk* __restrict__ chunks, const HmmChrom* __restrict__ chroms, const double* __restrict__ V, double* __restrict__ chunkSum,
                                                 int32_t* __restrict__ fail) {
    __shared__ double sw[4];
    const BbChunk K = chunks[blockIdx.x];
    const HmmChrom C = chroms[K.chrom];
    const double* __restrict__ Vc = V + C.begin;
    double s = 0.0; bool bad = false;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int64_t t = (int64_t)K.t0 + threadIdx.x * 4 + i;
        if (t < C.T) { const double v = Vc[t]; bad |= bb_bad_increment((unsigned long long)__double_as_longlong(v)); s += fabs(v); }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __hiloint2double(__shfl_xor(__double2hiint(s), d), __shfl_xor(__double2loint(s), d));
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
    if (bad) fail[K.chrom] = 1;
    __syncthreads();
    if (threadIdx.x == 0) chunkSum[blockIdx.x] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}
// one wave per chromosome: exclusive scan of the chunk sums
__global__ void __launch_bounds__(64) k_bb_bases(const int32_t* __restrict__ firstChunk, const double* __restrict__ chunkSum, double* __restrict__ chunkBase) {
    const int c = blockIdx.x, l = threadIdx.x;
    double carry = 0.0;
    for (int g = firstChunk[c]; g < firstChunk[c + 1]; g += 64) {
        const int i = g + l;
        const double v = i < firstChunk[c + 1] ? chunkSum[i] : 0.0;
        double inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { double o = shfl_up_f64(inc, d); if (l >= d) inc += o; }
        if (i < firstChunk[c + 1]) chunkBase[i] = carry + (inc - v);
        carry += __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(inc), 63), __builtin_amdgcn_readlane(__double2loint(inc), 63));
    }
}
struct SegFn { ParFn f; uint32_t head; uint32_t cnt; };       // segmented composition + crossing count
__device__ __forceinline__ SegFn segfn_then(SegFn a, SegFn b) {  // a first, then b
    SegFn r; r.head = a.head | b.head; r.cnt = a.cnt + b.cnt; r.f = b.head ? b.f : parfn_then(a.f, b.f);
    return r;
}
__global__ void __launch_bounds__(256) k_bb_pieces(const BbChunk* __restrict__ chunks, const HmmChrom* __restrict__ chroms, const double* __restrict__ V,
                                                   const double* __restrict__ chunkBase, BbChunkOut* __restrict__ outChunk, BbCross* __restrict__ outCross,
                                                   ParFn* __restrict__ at64Fn, uint8_t* __restrict__ at64Rank, int32_t* __restrict__ fail) {
    __shared__ double swSum[4];
    __shared__ SegFn swSeg[4];
    const BbChunk K = chunks[blockIdx.x];
    const HmmChrom C = chroms[K.chrom];
    const double* __restrict__ Vc = V + C.begin;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long vb[4]; double a[4]; bool in[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int64_t t = (int64_t)K.t0 + tid * 4 + i;
        in[i] = t < C.T;
        const double v = in[i] ? Vc[t] : 0.0;
        vb[i] = (unsigned long long)__double_as_longlong(v); a[i] = fabs(v);
    }
    // approximate running |D| before / after every step
    const double tsum = ((a[0] + a[1]) + a[2]) + a[3];
    double inc = tsum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { double o = shfl_up_f64(inc, d); if (lane >= d) inc += o; }
    if (lane == 63) swSum[wave] = inc;
    __syncthreads();
    double base = chunkBase[blockIdx.x];
    for (int w = 0; w < wave; w++) base += swSum[w];
    double Mprev = base + (inc - tsum);
    // per-step functions under the predicted binade; a crossing step restarts the composition behind it
    SegFn el[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const double Mcur = Mprev + a[i];
        const int ep = (int)(((unsigned long long)__double_as_longlong(Mprev) >> 52) & 0x7ffull), ec = (int)(((unsigned long long)__double_as_longlong(Mcur) >> 52) & 0x7ffull);
        const bool cross = in[i] && (ep == 0 || ec != ep);
        el[i].head = cross ? 1u : 0u; el[i].cnt = cross ? 1u : 0u;
        if (cross || !in[i]) { el[i].f.a0 = 0; el[i].f.a1 = 0; }
        else el[i].f = parfn_of(vb[i], ep);
        a[i] = Mcur;                                                 // keep: predicted |D| after the step
        Mprev = Mcur;
    }
    SegFn th = el[0];
#pragma unroll
    for (int i = 1; i < 4; i++) th = segfn_then(th, el[i]);
    SegFn sc = th;                                                   // inclusive scan over threads
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        SegFn o; o.f = parfn_shfl_up(sc.f, d); o.head = __shfl_up(sc.head, d); o.cnt = __shfl_up(sc.cnt, d);
        if (lane >= d) sc = segfn_then(o, sc);
    }
    if (lane == 63) swSeg[wave] = sc;
    __syncthreads();
    SegFn ex; ex.f.a0 = 0; ex.f.a1 = 0; ex.head = 0; ex.cnt = 0;     // exclusive prefix of this thread
    for (int w = 0; w < wave; w++) ex = segfn_then(ex, swSeg[w]);
    { SegFn o; o.f = parfn_shfl_up(sc.f, 1); o.head = __shfl_up(sc.head, 1); o.cnt = __shfl_up(sc.cnt, 1); if (lane > 0) ex = segfn_then(ex, o); }
    // walk the four steps again with the exclusive prefix
    SegFn run = ex;
    BbCross* __restrict__ myCross = outCross + (size_t)blockIdx.x * BB_MAXC;
    bool overflow = false;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int pos = tid * 4 + i;
        if (el[i].head) {
            if (run.cnt < BB_MAXC) {
                BbCross R; R.before = run.f; R.v = __longlong_as_double((long long)vb[i]);
                R.pos = pos; R.e = (int)(((unsigned long long)__double_as_longlong(a[i]) >> 52) & 0x7ffull);
                myCross[run.cnt] = R;
            } else overflow = true;
        }
        run = segfn_then(run, el[i]);
        if ((pos & 63) == 63) {
            at64Fn[(size_t)blockIdx.x * 16 + (pos >> 6)] = run.f;
            at64Rank[(size_t)blockIdx.x * 16 + (pos >> 6)] = (uint8_t)((run.cnt > 127u ? 127u : run.cnt) | (el[i].head ? 0x80u : 0u));
        }
    }
    if (overflow) fail[K.chrom] = 1;
    if (tid == 0) {   // binade the first piece was computed under (the first step's predicted "before" exponent)
        outChunk[blockIdx.x].headE = (int)(((unsigned long long)__double_as_longlong(base) >> 52) & 0x7ffull);
    }
    if (tid == 255) { outChunk[blockIdx.x].tail = run.f; outChunk[blockIdx.x].nCross = (int32_t)run.cnt; }
}
// one wave per chromosome.  Runs of chunks without a crossing are handled 64 at a time (a composition scan over the lanes gives
// every chunk its exact starting value); a chunk with crossings is walked record by record, redundantly in all lanes so that
// the running value stays wave-uniform.
__device__ __forceinline__ bool bb_apply(unsigned long long& bits, const ParFn& f, int eUsed) {    // false: a prediction was wrong
    const unsigned long long MANT = (1ull << 52) - 1ull;
    if (f.a0 == 0ull && f.a1 == 0ull) return true;        // empty piece
    const int e = (int)(bits >> 52);
    if (e != eUsed || e == 0) return false;
    const unsigned long long k = (bits & MANT) | (1ull << 52);
    const unsigned long long k2 = k + ((k & 1ull) ? f.a1 : f.a0);
    if (k2 >= (1ull << 53)) return false;
    bits = ((unsigned long long)e << 52) | (k2 & MANT);
    return true;
}
__global__ void __launch_bounds__(64) k_bb_walk(const int32_t* __restrict__ firstChunk, const BbChunkOut* __restrict__ chunkOut, const BbCross* __restrict__ cross,
                                                unsigned long long* __restrict__ chunkBits, BbPost* __restrict__ post, int32_t* __restrict__ fail) {
    __shared__ BbCross sCr[BB_MAXC];
    const int c = blockIdx.x, l = threadIdx.x;
    const unsigned long long MANT = (1ull << 52) - 1ull;
    if (fail[c]) return;
    unsigned long long bits = 0ull;       // |D| so far, wave-uniform
    bool failed = false;                  // wave-uniform
    for (int g = firstChunk[c]; g < firstChunk[c + 1] && !failed; g += 64) {
        const int n = firstChunk[c + 1] - g < 64 ? firstChunk[c + 1] - g : 64;
        BbChunkOut my; my.tail.a0 = 0; my.tail.a1 = 0; my.nCross = 0; my.headE = 0;
        if (l < n) my = chunkOut[g + l];
        unsigned long long crossMask = __ballot(l < n && my.nCross > 0);
        int cur = 0;
        while (cur < n && !failed) {
            const int f = crossMask ? (int)__builtin_ctzll(crossMask) : n;       // next chunk with crossings
            if (f > cur) {
                const bool inRun = l >= cur && l < f;
                ParFn inc; inc.a0 = inRun ? my.tail.a0 : 0ull; inc.a1 = inRun ? my.tail.a1 : 0ull;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { ParFn o = parfn_shfl_up(inc, d); if (l >= d) inc = parfn_then(o, inc); }
                ParFn ex = parfn_shfl_up(inc, 1); if (l == 0) { ex.a0 = 0; ex.a1 = 0; }
                const int e = (int)(bits >> 52);
                const unsigned long long k = (bits & MANT) | (e ? (1ull << 52) : 0ull);
                const unsigned long long ks = k + ((k & 1ull) ? ex.a1 : ex.a0), ke = k + ((k & 1ull) ? inc.a1 : inc.a0);
                const bool nonEmpty = my.tail.a0 != 0ull || my.tail.a1 != 0ull;
                const bool bad = inRun && nonEmpty && (my.headE != e || e == 0 || ke >= (1ull << 53));
                if (inRun) chunkBits[g + l] = ((unsigned long long)e << 52) | (ks & MANT);
                if (__ballot(bad)) failed = true;
                const unsigned long long totLo0 = __shfl(inc.a0, f - 1), totLo1 = __shfl(inc.a1, f - 1);
                const unsigned long long k2 = k + ((k & 1ull) ? totLo1 : totLo0);
                bits = ((unsigned long long)e << 52) | (k2 & MANT);
                cur = f;
            }
            if (cur < n && !failed) {      // chunk `cur` has crossings
                const int nc = __shfl(my.nCross, cur), headE = __shfl(my.headE, cur);
                ParFn tail; tail.a0 = __shfl(my.tail.a0, cur); tail.a1 = __shfl(my.tail.a1, cur);
                if (l < nc && l < BB_MAXC) sCr[l] = cross[(size_t)(g + cur) * BB_MAXC + l];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                if (l == 0) chunkBits[g + cur] = bits;
                int eCur = headE;
                for (int r = 0; r < nc && r < BB_MAXC && !failed; r++) {
                    const BbCross R = sCr[r];
                    if (!bb_apply(bits, R.before, eCur)) { failed = true; break; }
                    const double acc = -__longlong_as_double((long long)bits) + R.v;          // the leaving step: a real add
                    bits = (unsigned long long)__double_as_longlong(-acc) & ~(1ull << 63);
                    if ((int)(bits >> 52) != R.e) failed = true;
                    if (l == 0) post[(size_t)(g + cur) * BB_MAXC + r].bits = bits;
                    eCur = R.e;
                }
                if (!failed && !bb_apply(bits, tail, eCur)) failed = true;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                crossMask &= ~(1ull << cur);
                cur++;
            }
        }
    }
    if (failed && l == 0) fail[c] = 1;
}
__global__ void __launch_bounds__(256) k_bb_emit(const BbChunk* __restrict__ chunks, int nchunks, const HmmChrom* __restrict__ chroms, const unsigned long long* __restrict__ chunkBits,
                                                 const BbPost* __restrict__ post, const ParFn* __restrict__ at64Fn, const uint8_t* __restrict__ at64Rank, double* __restrict__ carryOut) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int c = i >> 4, q = i & 15;
    if (c >= nchunks) return;
    const BbChunk K = chunks[c];
    const HmmChrom C = chroms[K.chrom];
    if (q == 0 && K.t0 == 0) carryOut[C.begin] = 0.0;
    const int64_t t = (int64_t)K.t0 + q * 64 + 63;
    if (t + 1 >= C.T) return;
    const unsigned long long MANT = (1ull << 52) - 1ull;
    const uint32_t rk = at64Rank[i];
    const uint32_t rank = rk & 0x7fu;
    unsigned long long bits = rank == 0 ? chunkBits[c] : post[(size_t)c * BB_MAXC + (rank - 1)].bits;
    if (!(rk & 0x80u)) {
        const ParFn f = at64Fn[i];
        if (f.a0 != 0ull || f.a1 != 0ull) {
            const unsigned long long k = (bits & MANT) | (1ull << 52);
            const unsigned long long k2 = k + ((k & 1ull) ? f.a1 : f.a0);
            bits = (bits & ~MANT) | (k2 & MANT);
        }
    }
    carryOut[C.begin + t + 1] = -__longlong_as_double((long long)bits);
}

// the 25 transition terms as separate scalars held in registers: as an array (or read from the argument block) the select over the guessed
// predecessor turns into a table lookup in memory -- a load and a wait per state and step
struct LogA25 { double a00, a01, a02, a03, a04, a10, a11, a12, a13, a14, a20, a21, a22, a23, a24, a30, a31, a32, a33, a34, a40, a41, a42, a43, a44; };
#define LOGA_COL(A, j) A.a0##j, A.a1##j, A.a2##j, A.a3##j, A.a4##j
__device__ __forceinline__ double pick5(uint32_t p, double x0, double x1, double x2, double x3, double x4) {
    double r = x0;
    r = p == 1 ? x1 : r; r = p == 2 ? x2 : r; r = p == 3 ? x3 : r; r = p == 4 ? x4 : r;
    return r;
}
// ---- C: exact verification.  State of one lane:
struct VerState { double d[NSTATE]; uint32_t valid; bool bad; double Dprev; int sPrev; uint32_t why; };   // valid bit j: d[j] is the exact delta_t(j)
// lead-in step: follow the guessed pointers; a state is exact once its ancestry reaches the backbone (valid == 0 in front of the lead-in:
// nothing is known there).  `on` == false leaves everything as it is.
template <bool useLds>
__device__ __forceinline__ void ver_lead_step(VerState& S, const LogA25& A, const double* __restrict__ sTab, const double* __restrict__ logPmf, int tableLen,
                                              bool on, int k, int sCur, double v, uint32_t pk) {
    double e[NSTATE];
    vit_emissions(e, sTab, logPmf, useLds, tableLen, on ? k : 0);
    const double Dt = S.Dprev + v;                    // D_t = D_{t-1} + v_t (same association as the backbone)
    double nd[NSTATE]; uint32_t nv = 0;
    const double la5[NSTATE] = {pick5(map_get(pk, 0), LOGA_COL(A, 0)), pick5(map_get(pk, 1), LOGA_COL(A, 1)), pick5(map_get(pk, 2), LOGA_COL(A, 2)),
                                pick5(map_get(pk, 3), LOGA_COL(A, 3)), pick5(map_get(pk, 4), LOGA_COL(A, 4))};
#pragma unroll
    for (int j = 0; j < NSTATE; j++) {
        const uint32_t p = map_get(pk, j);
        const double dp = pick5(p, S.d[0], S.d[1], S.d[2], S.d[3], S.d[4]);
        const bool vp = ((S.valid >> p) & 1u) != 0;
        const double step = e[j] + la5[j];
        const bool isCur = j == sCur, fromPrev = (int)p == S.sPrev;
        const double viaPrev = S.Dprev + step, viaAnc = dp + step;                 // both computed: the selects below then carry no arithmetic (no branches)
        const double r0 = vp ? viaAnc : S.d[j], r1 = fromPrev ? viaPrev : r0, r = isCur ? Dt : r1;
        nd[j] = r; nv |= (isCur || fromPrev || vp) ? (1u << j) : 0u;
    }
#pragma unroll
    for (int j = 0; j < NSTATE; j++) S.d[j] = on ? nd[j] : S.d[j];
    S.valid = on ? nv : S.valid; S.Dprev = on ? Dt : S.Dprev; S.sPrev = on ? sCur : S.sPrev;
}
// block step: every state must be exact by now; redo the reference's arg-max and compare with the guess
template <bool useLds>
__device__ __forceinline__ void ver_block_step(VerState& S, const HmmParams& P, const double* __restrict__ sTab, const double* __restrict__ logPmf, bool on, int k, int sCur,
                                               double v, uint32_t pk) {
    double e[NSTATE], dn[NSTATE];
    vit_emissions(e, sTab, logPmf, useLds, P.tableLen, on ? k : 0);
    const double Dt = S.Dprev + v;
#pragma unroll
    for (int j = 0; j < NSTATE; j++) dn[j] = S.d[j];
    const uint32_t got = vit_step5(dn, e, P);       // (the two-constant form of the speculative pass was measured here too: no change, 151 us)
    S.bad = S.bad || (on && (S.valid != 31u || got != pk || sel5(dn, (uint32_t)(on ? sCur : 0)) != Dt));
    S.why |= on ? ((S.valid != 31u ? 2u : 0u) | (got != pk ? 4u : 0u) | (sel5(dn, (uint32_t)(on ? sCur : 0)) != Dt ? 8u : 0u)) : 0u;      // (which check: CANVAS_HMM_DEBUG_FAIL)
#pragma unroll
    for (int j = 0; j < NSTATE; j++) S.d[j] = on ? dn[j] : S.d[j];
    S.Dprev = on ? Dt : S.Dprev; S.sPrev = on ? sCur : S.sPrev;
}
// A phase = `cnt` consecutive steps of one kind (per-lane count; the wave runs max(cnt) rounds and the lanes that are done are switched off by
// predicate, so the step code is straight-line).  The operands of the next eight steps are fetched with 16-byte loads one group ahead; they stay
// inside the lane's own range (state[] is the caller's array), the last group is loaded guarded.  With one load per array and step and a
// branch per condition every step used to wait for ALL outstanding loads (SQ_WAIT_ANY 62 % of the wave cycles).
template <bool useLds, bool LEAD>
__device__ __forceinline__ void ver_phase(VerState& S, const LogA25& A, const HmmParams& P, const double* __restrict__ sTab, const double* __restrict__ logPmf,
                                          const int32_t* __restrict__ ixp, const int32_t* __restrict__ stp, const double* __restrict__ vp, const uint16_t* __restrict__ ppp, int cnt) {
    const int maxCnt = wave_max_i32(cnt);
    int kq[PQ], sq[PQ]; double vq[PQ]; uint32_t pq[PQ];
#pragma unroll
    for (int u = 0; u < PQ; u++) { const bool in = u < cnt; kq[u] = in ? ixp[u] : 0; sq[u] = in ? stp[u] : 0; vq[u] = in ? vp[u] : 0.0; pq[u] = in ? ppp[u] : 0u; }
    for (int j0 = 0; j0 < maxCnt; j0 += PQ) {
        int kn[PQ], sn[PQ]; double vn[PQ]; VecH8 pn;          // (the packed back-pointers are only taken apart at the end of the group: unpacking them
        if (j0 + 2 * PQ <= cnt) {                              //  here would make the group wait for its own prefetch)
            const int o = j0 + PQ;
            const VecI4 ka = *reinterpret_cast<const VecI4*>(ixp + o), kb = *reinterpret_cast<const VecI4*>(ixp + o + 4);
            const VecI4 sa = *reinterpret_cast<const VecI4*>(stp + o), sb = *reinterpret_cast<const VecI4*>(stp + o + 4);
            const VecD2 v0 = *reinterpret_cast<const VecD2*>(vp + o), v1 = *reinterpret_cast<const VecD2*>(vp + o + 2), v2 = *reinterpret_cast<const VecD2*>(vp + o + 4),
                        v3 = *reinterpret_cast<const VecD2*>(vp + o + 6);
            pn = *reinterpret_cast<const VecH8*>(ppp + o);
#pragma unroll
            for (int u = 0; u < 4; u++) { kn[u] = ka.v[u]; kn[4 + u] = kb.v[u]; sn[u] = sa.v[u]; sn[4 + u] = sb.v[u]; }
            vn[0] = v0.v[0]; vn[1] = v0.v[1]; vn[2] = v1.v[0]; vn[3] = v1.v[1]; vn[4] = v2.v[0]; vn[5] = v2.v[1]; vn[6] = v3.v[0]; vn[7] = v3.v[1];
        } else {
#pragma unroll
            for (int u = 0; u < PQ; u++) { const bool in = j0 + PQ + u < cnt; kn[u] = in ? ixp[j0 + PQ + u] : 0; sn[u] = in ? stp[j0 + PQ + u] : 0; vn[u] = in ? vp[j0 + PQ + u] : 0.0; }
#pragma unroll
            for (int u = 0; u < 4; u++) pn.w[u] = (uint32_t)(j0 + PQ + 2 * u < cnt ? ppp[j0 + PQ + 2 * u] : (uint16_t)0) | ((uint32_t)(j0 + PQ + 2 * u + 1 < cnt ? ppp[j0 + PQ + 2 * u + 1] : (uint16_t)0) << 16);
        }
#pragma unroll
        for (int u = 0; u < PQ; u++) {
            if (LEAD) ver_lead_step<useLds>(S, A, sTab, logPmf, P.tableLen, j0 + u < cnt, kq[u], sq[u], vq[u], pq[u]);
            else ver_block_step<useLds>(S, P, sTab, logPmf, j0 + u < cnt, kq[u], sq[u], vq[u], pq[u]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < PQ; u++) { kq[u] = kn[u]; sq[u] = sn[u]; vq[u] = vn[u]; pq[u] = (pn.w[u >> 1] >> (16 * (u & 1))) & 0xFFFFu; }
    }
}
template <bool useLds>
__global__ void __launch_bounds__(64) k_vit_verify(const VitBlock* __restrict__ blocks, int nblocks, const HmmChrom* __restrict__ chroms, const int32_t* __restrict__ idx,
                                                   const double* __restrict__ logPmf, HmmParams P, const uint16_t* __restrict__ psi,
                                                   const int32_t* __restrict__ state, const double* __restrict__ V, const double* __restrict__ carry, const int32_t* __restrict__ lastGuess,
                                                   int32_t* __restrict__ fail, int leadIn, const int32_t* __restrict__ todo) {
    extern __shared__ double sTab[];
    vit_stage_table(sTab, logPmf, P.tableLen, useLds);
    const int b = blockIdx.x * 64 + threadIdx.x;
    const bool actBlock = b < nblocks;
    const VitBlock B = blocks[actBlock ? b : nblocks - 1];
    const HmmChrom C = chroms[B.chrom];
    const int64_t tBeg = B.t0, tEnd = (B.t0 + VB < C.T) ? B.t0 + VB : C.T;
    const int64_t ts = tBeg > leadIn ? tBeg - leadIn : 0;                          // a multiple of 64: carry[] is defined there
    const bool act = actBlock && (!todo || todo[B.chrom]);
    const int nsteps = act ? (int)(tEnd - ts) : 0;
    const int maxSteps = wave_max_i32(nsteps);
    const int32_t* __restrict__ ix = idx + C.begin + ts;
    const int32_t* __restrict__ st = state + C.begin + ts;
    const double* __restrict__ Vc = V + C.begin + ts;
    const uint16_t* __restrict__ pp = psi + C.begin + ts;
    VerState S;
#pragma unroll
    for (int j = 0; j < NSTATE; j++) S.d[j] = 0.0;
    S.valid = 0; S.bad = false; S.why = 0;
    S.Dprev = carry[C.begin + ts];              // D_{ts-1} (0 at the start of the chromosome)
    S.sPrev = ts > 0 ? state[C.begin + ts - 1] : -1;
    // the transition terms as opaque register values: left as loads from the argument block, the select over the guessed predecessor below
    // becomes a select over ADDRESSES followed by a load (and a wait) per state and step
    LogA25 A;
#define LOGA_LOAD(i, j) A.a##i##j = P.logA[i][j]; asm volatile("" : "+v"(A.a##i##j));
    LOGA_LOAD(0, 0) LOGA_LOAD(0, 1) LOGA_LOAD(0, 2) LOGA_LOAD(0, 3) LOGA_LOAD(0, 4) LOGA_LOAD(1, 0) LOGA_LOAD(1, 1) LOGA_LOAD(1, 2) LOGA_LOAD(1, 3) LOGA_LOAD(1, 4)
    LOGA_LOAD(2, 0) LOGA_LOAD(2, 1) LOGA_LOAD(2, 2) LOGA_LOAD(2, 3) LOGA_LOAD(2, 4) LOGA_LOAD(3, 0) LOGA_LOAD(3, 1) LOGA_LOAD(3, 2) LOGA_LOAD(3, 3) LOGA_LOAD(3, 4)
    LOGA_LOAD(4, 0) LOGA_LOAD(4, 1) LOGA_LOAD(4, 2) LOGA_LOAD(4, 3) LOGA_LOAD(4, 4)
#undef LOGA_LOAD
    const int lead = (int)(tBeg - ts);
    // step 0 on its own: the first bin of a chromosome starts the recurrence (HMM.cs:78), any other first step is a lead-in step without history
    if (nsteps > 0) {
        if (ts == 0) {
            double e[NSTATE];
            vit_emissions(e, sTab, logPmf, useLds, P.tableLen, ix[0]);
            const double Dt = S.Dprev + Vc[0];
            const int sCur = st[0];
            vit_init5(S.d, e, P); S.valid = 31u;
            if (sel5(S.d, (uint32_t)sCur) != Dt) { S.bad = true; S.why |= 16u; }
            S.Dprev = Dt; S.sPrev = sCur;
        } else ver_lead_step<useLds>(S, A, sTab, logPmf, P.tableLen, true, ix[0], st[0], Vc[0], (uint32_t)pp[0]);
    }
    const int start2 = lead > 1 ? lead : 1;
    ver_phase<useLds, true>(S, A, P, sTab, logPmf, ix + 1, st + 1, Vc + 1, pp + 1, nsteps > 0 && lead > 1 ? lead - 1 : 0);
    ver_phase<useLds, false>(S, A, P, sTab, logPmf, ix + start2, st + start2, Vc + start2, pp + start2, nsteps > start2 ? nsteps - start2 : 0);
    double (&d)[NSTATE] = S.d; const uint32_t valid = S.valid; bool bad = S.bad;
    uint32_t why = S.why;
    if (act && tEnd == C.T) { if (valid != 31u || vit_best5(d) != lastGuess[B.chrom]) { bad = true; why |= 32u; } }
    if (act && bad) atomicOr(&fail[B.chrom], (int32_t)(1u | why));
}

// test hook (CANVAS_HMM_TEST_CORRUPT): flips one guessed back-pointer so that tests can prove k_vit_verify catches a wrong guess
__global__ void k_vit_corrupt(uint16_t* __restrict__ psi, int64_t at) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { uint32_t v = psi[at], p = (v >> 6) & 7u; psi[at] = (uint16_t)((v & ~(7u << 6)) | (((p + 1u) % 5u) << 6)); }
}

// ---- backtracking as function composition over the VB-step blocks: state[t-1] = psi_t[state[t]]
// map of a block = psi_{tBeg} o ... o psi_{tEnd-1}: state at the block's last step -> state just before its first step
__global__ void __launch_bounds__(256) k_bt_maps(const VitBlock* __restrict__ blocks, int nblocks, const HmmChrom* __restrict__ chroms, const uint16_t* __restrict__ psi,
                                                 uint16_t* __restrict__ maps) {
    int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nblocks) return;
    const VitBlock B = blocks[b];
    const HmmChrom C = chroms[B.chrom];
    const int64_t tBeg = B.t0 > 0 ? B.t0 : 1, tEnd = (B.t0 + VB < C.T) ? B.t0 + VB : C.T;
    uint32_t m = MAP_IDENT;
    for (int64_t t = tEnd - 1; t >= tBeg; t--) m = map_compose(psi[C.begin + t], m);
    maps[b] = (uint16_t)m;
}
// entry state of every block (the state at its last step), backwards from the chromosome's last state, in two parallel steps:
// k_bt_group: one wave per group of 64 blocks (counted from the END of the chromosome): composition scan of the block maps -> per block the map
//             from the group's entry state to the block's, per group the map to the next group's entry state;
// k_bt_gchain: one wave per chromosome: the same scan over the group maps (64 groups per iteration) -> entry state of every group.
// k_bt_states combines the two.  (One wave per chromosome used to walk all groups in sequence: 39 us for 48 dependent iterations on chr1.)
__global__ void __launch_bounds__(64) k_bt_group(const int32_t* __restrict__ firstBlock, const int32_t* __restrict__ firstGroup, int nchr, const uint16_t* __restrict__ maps,
                                                 uint16_t* __restrict__ pre, uint16_t* __restrict__ gmap) {
    const int g = blockIdx.x, l = threadIdx.x;
    int lo = 0, hi = nchr - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (firstGroup[mid] <= g) lo = mid; else hi = mid - 1; }
    while (lo < nchr - 1 && firstGroup[lo + 1] <= g) lo++;                     // chromosomes without blocks share an index
    const int first = firstBlock[lo], last = firstBlock[lo + 1] - 1;
    const int bi = last - ((g - firstGroup[lo]) * 64 + l);
    uint32_t f = bi >= first ? (uint32_t)maps[bi] : MAP_IDENT;
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) { const uint32_t o = __shfl_up(f, dlt); if (l >= dlt) f = map_compose(f, o); }   // first the later blocks (o), then this one
    const uint32_t ex = __shfl_up(f, 1);
    if (bi >= first) pre[bi] = (uint16_t)(l == 0 ? MAP_IDENT : ex);
    if (l == 63) gmap[g] = (uint16_t)f;
}
__global__ void __launch_bounds__(64) k_bt_gchain(const int32_t* __restrict__ firstGroup, const int32_t* __restrict__ lastState, const uint16_t* __restrict__ gmap,
                                                  int8_t* __restrict__ gentry) {
    const int c = blockIdx.x, l = threadIdx.x;
    const int g0 = firstGroup[c], ng = firstGroup[c + 1] - g0;
    int s = lastState[c];
    for (int base = 0; base < ng; base += 64) {
        uint32_t f = base + l < ng ? (uint32_t)gmap[g0 + base + l] : MAP_IDENT;
#pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1) { const uint32_t o = __shfl_up(f, dlt); if (l >= dlt) f = map_compose(f, o); }
        const uint32_t ex = __shfl_up(f, 1);
        if (base + l < ng) gentry[g0 + base + l] = (int8_t)(s < 0 ? s : (l == 0 ? s : (int)map_get(ex, (uint32_t)s)));
        const uint32_t all = (uint32_t)__builtin_amdgcn_readlane((int)f, 63);
        if (s >= 0) s = (int)map_get(all, (uint32_t)s);
    }
}
// One wave per block: the states of the block are the suffix compositions of its back-pointer maps applied to the state at the block's last
// step, state[tEnd-1-j] = (h_j o ... o h_1)(entry) with h_j = psi[tEnd-j].  Two steps per lane, a composition scan over the lanes, coalesced
// loads and stores (one lane per block walked its 128 steps behind 128 dependent loads and wrote with a 512-byte stride).
__global__ void __launch_bounds__(256) k_bt_states(const VitBlock* __restrict__ blocks, int nblocks, const HmmChrom* __restrict__ chroms, const uint16_t* __restrict__ psi,
                                                   const int32_t* __restrict__ firstBlock, const int32_t* __restrict__ firstGroup, const uint16_t* __restrict__ pre,
                                                   const int8_t* __restrict__ gentry, int32_t* __restrict__ state) {
    static_assert(VB <= 128, "two steps per lane");
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), l = lane_id();
    if (b >= nblocks) return;
    const VitBlock B = blocks[b];
    const HmmChrom C = chroms[B.chrom];
    const int64_t tBeg = B.t0, tEnd = (B.t0 + VB < C.T) ? B.t0 + VB : C.T;
    const int cnt = (int)(tEnd - tBeg);
    const int sg = gentry[firstGroup[B.chrom] + (firstBlock[B.chrom + 1] - 1 - b) / 64];           // entry state of the block's group ...
    const int s = sg < 0 ? sg : (int)map_get((uint32_t)pre[b], (uint32_t)sg);                      // ... pushed through the later blocks of the group
    const int j0 = 2 * l, j1 = 2 * l + 1;
    const uint32_t a0 = (j0 >= 1 && j0 < cnt) ? (uint32_t)psi[C.begin + tEnd - j0] : MAP_IDENT;     // h_0 is the identity (the entry state itself)
    const uint32_t a1 = (j1 < cnt) ? (uint32_t)psi[C.begin + tEnd - j1] : MAP_IDENT;
    const uint32_t x1 = map_compose(a1, a0);                                                         // first h_{2l}, then h_{2l+1}
    uint32_t v = x1;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(v, d); if (l >= d) v = map_compose(v, o); }   // first the earlier lanes (o), then this one
    uint32_t ex = __shfl_up(v, 1);
    if (l == 0) ex = MAP_IDENT;
    const uint32_t m0 = map_compose(a0, ex), m1 = map_compose(x1, ex);
    if (j0 < cnt) state[C.begin + tEnd - 1 - j0] = s < 0 ? s : (int)map_get(m0, (uint32_t)s);
    if (j1 < cnt) state[C.begin + tEnd - 1 - j1] = s < 0 ? s : (int)map_get(m1, (uint32_t)s);
}

__global__ void __launch_bounds__(256) k_fill_i32(int32_t* __restrict__ p, int64_t begin, int64_t end, int32_t v) {
    int64_t i = begin + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < end) p[i] = v;
}

// ---- segment ids
__global__ void __launch_bounds__(256) k_seg_flags(const int64_t* __restrict__ chrOff, int nchr, const int32_t* __restrict__ state, const int32_t* __restrict__ start,
                                                   const int32_t* __restrict__ stop, int64_t n, int32_t maxDist, const int64_t* __restrict__ exclOff,
                                                   const int32_t* __restrict__ exclStart, const int32_t* __restrict__ exclStop,
                                                   const int64_t* __restrict__ plOff, const int32_t* __restrict__ plStart, const int32_t* __restrict__ plEnd, const int32_t* __restrict__ plCn,
                                                   uint8_t* __restrict__ flags) {
    __shared__ int64_t sOff[CHR_LDS_CAP];
    const int64_t* off = stage_chr_offsets(chrOff, nchr, sOff);
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // everything the decisions below may look at is requested up front (four loads in flight instead of up to three round trips one after the other)
    const int64_t im = i > 0 ? i - 1 : 0;
    const int32_t st = state[i], stPrev = state[im], stopPrev = stop[im], stopHere = stop[i], startHere = start[i];
    const int lo = chrom_of_bin(off, nchr, i);
    const bool first = (i == off[lo]);
    bool newSeg = st >= 0 && (first || stPrev != st);                 // breakpoint -> segment start present in `starts`
    if (exclOff) {
        // forbidden intervals (SegmentationResultsProcessor.cs:88-111): excludeIndex = first interval whose Stop >= previousBinEnd
        // (the reference advances a cursor; with intervals sorted by Stop that is a lower bound), split when the interval's
        // midpoint lies in (previousBinEnd, end]
        const uint32_t prevEnd = first ? 0u : (uint32_t)stopPrev;
        int64_t a = exclOff[lo], b = exclOff[lo + 1];
        while (a < b) { int64_t mid = (a + b) >> 1; if ((int64_t)exclStop[mid] < (int64_t)prevEnd) a = mid + 1; else b = mid; }
        if (a < exclOff[lo + 1]) {
            const int forbiddenZoneMid = (exclStart[a] + exclStop[a]) / 2;
            if ((int64_t)prevEnd < forbiddenZoneMid && (int64_t)(uint32_t)stopHere >= forbiddenZoneMid) newSeg = true;
        }
    }
    if (!newSeg && !first) {
        uint32_t prevEnd = (uint32_t)stopPrev;
        if (prevEnd > 0 && maxDist >= 0 && (uint64_t)prevEnd + (uint64_t)maxDist < (uint64_t)(uint32_t)startHere) newSeg = true;   // SegmentationResultsProcessor.cs:112-116
    }
    if (!newSeg && plOff) {
        // reference ploidy changes between the end of the previous bin and the end of this one (SegmentationResultsProcessor.cs:117-128):
        // PloidyInfo.getPloidyCounts over the one-based interval [previousBinEnd > 0 ? previousBinEnd : 1, end] (PloidyInfo.cs:93-110); a chromosome
        // has a handful of records (PAR / non-PAR stretches of the sex chromosomes), so every bin walks its chromosome's list
        const uint32_t prevEnd = first ? 0u : (uint32_t)stopPrev;
        const int qs = prevEnd > 0 ? (int)prevEnd : 1, qe = (int)(uint32_t)stopHere;
        int bc0 = 0, bc1 = 0, bc2 = qe - qs + 1, bc3 = 0, bc4 = 0;
        for (int64_t k = plOff[lo]; k < plOff[lo + 1]; k++) {
            const int p = plCn[k];
            if (p == 2) continue;
            const int overlapStart = max(qs - 1, plStart[k] - 1);
            if (overlapStart > plEnd[k]) continue;
            const int overlapBases = min(qe, plEnd[k]) - overlapStart;
            if (overlapBases <= 0) continue;
            bc2 -= overlapBases;
            bc0 += p == 0 ? overlapBases : 0; bc1 += p == 1 ? overlapBases : 0; bc3 += p == 3 ? overlapBases : 0; bc4 += p == 4 ? overlapBases : 0;
        }
        if ((bc0 > 0) + (bc1 > 0) + (bc2 > 0) + (bc3 > 0) + (bc4 > 0) >= 2) newSeg = true;
    }
    flags[i] = newSeg;
}
__global__ void __launch_bounds__(256) k_count_blocks(const uint8_t* __restrict__ flags, int64_t n, uint32_t* __restrict__ blockCnt) {
    __shared__ uint32_t sh[4];
    int64_t base = (int64_t)blockIdx.x * 2048;
    uint32_t c = 0;
    for (int jj = 0; jj < 8; jj++) { int64_t i = base + jj * 256 + threadIdx.x; if (i < n) c += flags[i]; }
    c = wave_reduce_add_u32(c);
    if (lane_id() == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blockCnt[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ void __launch_bounds__(1024) k_scan_blocks2(uint32_t* __restrict__ blockCnt, int nblocks, unsigned long long* __restrict__ total, unsigned* __restrict__ totalSeq = nullptr, unsigned seq = 0) {
    __shared__ uint32_t sh[17];
    uint32_t carry = 0;
    for (int base = 0; base < nblocks; base += 1024) {
        int i = base + threadIdx.x;
        uint32_t v = i < nblocks ? blockCnt[i] : 0;
        uint32_t inc = wave_inclusive_scan_u32(v);
        int w = threadIdx.x >> 6;
        if (lane_id() == 63) sh[w] = inc;
        __syncthreads();
        if (threadIdx.x == 0) { uint32_t s = 0; for (int k = 0; k < 16; k++) { uint32_t t = sh[k]; sh[k] = s; s += t; } sh[16] = s; }
        __syncthreads();
        if (i < nblocks) blockCnt[i] = carry + sh[w] + inc - v;
        carry += sh[16];
        __syncthreads();
    }
    if (threadIdx.x == 0) { *total = carry; if (totalSeq) cvx_mail_publish(totalSeq, seq); }      // (total in pinned host memory: stamped, common.hpp)
}
__global__ void __launch_bounds__(256) k_seg_ids(const uint8_t* __restrict__ flags, const uint32_t* __restrict__ blockOff, int64_t n, int32_t* __restrict__ segId) {
    __shared__ uint32_t sh[4];
    int64_t base = (int64_t)blockIdx.x * 2048;
    uint32_t running = blockOff[blockIdx.x];
    for (int jj = 0; jj < 8; jj++) {
        int64_t i = base + jj * 256 + threadIdx.x;
        uint32_t f = (i < n) ? flags[i] : 0;
        uint32_t inc = wave_inclusive_scan_u32(f);
        if (lane_id() == 63) sh[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t woff = 0, tot = 0;
        for (int k = 0; k < 4; k++) { if (k < (int)(threadIdx.x >> 6)) woff += sh[k]; tot += sh[k]; }
        if (i < n) segId[i] = (int32_t)(running + woff + inc) - 1;     // counter starts at -1 and increments on every new segment
        running += tot;
        __syncthreads();
    }
}

// =====================================================================================================================
// Joint (multi-sample) mode: HiddenMarkovModelsRunner.Run(inputs, isPerSample: false) (HiddenMarkovModelsRunner.cs:23-152),
// NegativeBinomialMixture.EstimateViterbiLikelihood with useAllStates = false (Distributions.cs:257-323).
//
// Per chromosome and sample: haploid mean = max(1, median)/2 and Utilities.Variance of the uncapped coverage, tables of S x 5 negative
// binomials of length maxValues + 10.  The likelihood of a step is
//     log( max over genotype combinations of state j of  prod_s pmf*(x_s) )  +  log( transition term )
// and the transition term of Distributions.cs:299-320 is A[i][j]: the best combination only holds states j and 2, so coming from
// diploid (i == 2) the minimum over it is A[2][j], into diploid it is A[i][2], and otherwise the minimum over its non-diploid
// members is A[i][j] (0.0025 < 0.99).  The recurrence therefore has the per-sample form with a per-bin emission value, and the whole
// speculate / verify machinery is reused through a [5][N] "table" indexed by the bin itself.
// Math.Log of the per-bin maximum is evaluated by the HOST libm on all bins (threads), like the emission tables: the reference
// calls the platform libm there and bit-exactness against it is what keeps near-tie arg-maxes identical (Q13).
#define JOINT_MAXS 16          // samples
#define JOINT_MAXCOMBO 15      // multiset permutations of {j x (S'-d), 2 x d}, S' = min(S, 4)
struct JointCombos { int32_t n[NSTATE]; int8_t g[NSTATE][JOINT_MAXCOMBO][4]; int32_t sp; };
struct JointPtrs { const double* cov[JOINT_MAXS]; };

// Utilities.Median(IEnumerable<double>) per (chromosome, sample): SortedList<double>.Median()
__global__ void __launch_bounds__(1024) k_joint_median(JointPtrs cov, const HmmChrom* __restrict__ chroms, int nsamples, double* __restrict__ med) {
    __shared__ uint32_t sH[2][256];
    __shared__ unsigned long long sPre[2], sK[2];
    const int c = blockIdx.x, d = blockIdx.y;
    const HmmChrom C = chroms[c];
    if (C.T <= 10) return;
    const double* __restrict__ x = cov.cov[d];
    const int64_t cnt = C.T;
    const unsigned long long r1 = (unsigned long long)(cnt / 2), r0 = (cnt % 2) ? r1 : r1 - 1;
    wg_select2([&](int64_t i) { return key_of_double(x[i]); }, C.begin, C.begin + C.T, r0, r1, sH, sPre, sK);
    if (threadIdx.x == 0) med[c * nsamples + d] = (cnt % 2) ? double_of_key(sPre[1]) : (double_of_key(sPre[0]) + double_of_key(sPre[1])) / 2;
}
// sequential double sums in list order (Enumerable.Average and the loop of Utilities.Variance, Utilities.cs:290-301): one wave per
// (chromosome, sample), 64 operands staged per chunk and read back as LDS broadcasts so that the chain is one FP64 add per element.
// pass 0: sum of x;  pass 1: sum of (x - mu)^2 with mu = sum0 / T
__global__ void __launch_bounds__(64) k_joint_seqsum(JointPtrs cov, const HmmChrom* __restrict__ chroms, int nsamples, int pass, const double* __restrict__ sum0, double* __restrict__ out) {
    __shared__ double sV[2][64];
    const int c = blockIdx.x, d = blockIdx.y, l = threadIdx.x;
    const HmmChrom C = chroms[c];
    if (C.T <= 10) return;
    const double* __restrict__ x = cov.cov[d] + C.begin;
    const double mu = pass ? sum0[c * nsamples + d] / (double)C.T : 0.0;
    auto val = [&](int64_t t) -> double { if (t >= C.T) return 0.0; const double v = x[t]; if (!pass) return v; const double df = v - mu; return df * df; };
    double acc = 0.0;
    double vNext = val(l);
    int buf = 0;
    for (int64_t c0 = 0; c0 < C.T; c0 += 64, buf ^= 1) {
        sV[buf][l] = vNext;
        vNext = val(c0 + 64 + l);
        double r[64];
#pragma unroll
        for (int s = 0; s < 64; s++) r[s] = sV[buf][s];
#pragma unroll
        for (int s = 0; s < 64; s++) acc = acc + r[s];          // past the end: + 0.0 leaves acc unchanged (acc >= 0)
    }
    if (l == 0) out[c * nsamples + d] = acc;
}
// RemoveOutliers (HiddenMarkovModelsRunner.cs:154-162) + Convert.ToInt32; per-chromosome maximum of the indices
__global__ void __launch_bounds__(256) k_joint_index(JointPtrs cov, int nsamples, const int64_t* __restrict__ chrOff, int nchr, const double* __restrict__ thr, int64_t N,
                                                     int32_t* __restrict__ idxS, int32_t* __restrict__ maxIdx) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= N) return;
    int lo = 0, hi = nchr - 1;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (chrOff[mid] <= g) lo = mid; else hi = mid - 1; }
    const double t = thr[lo];
    int m = 0;
    for (int d = 0; d < nsamples; d++) {
        double x = cov.cov[d][g];
        x = x > t ? t : x;
        int k = (int)rint(x); k = k < 0 ? 0 : k;
        idxS[(size_t)d * N + g] = k;
        m = k > m ? k : m;
    }
    atomicMax(&maxIdx[lo], m);
}
// per bin and state: max over the genotype combinations of the product of the (grouped) pmf values (Distributions.cs:262-296)
__global__ void __launch_bounds__(256) k_joint_emission(const int32_t* __restrict__ idxS, int nsamples, const int64_t* __restrict__ chrOff, int nchr, const HmmChrom* __restrict__ chroms,
                                                        const double* __restrict__ pmf /* [chr][sample][state][stride] */, int stride, JointCombos K, int64_t N,
                                                        double* __restrict__ maxL /* [5][N] */) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= N) return;
    int lo = 0, hi = nchr - 1;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (chrOff[mid] <= g) lo = mid; else hi = mid - 1; }
    if (chroms[lo].T <= 10) { for (int j = 0; j < NSTATE; j++) maxL[(size_t)j * N + g] = 1.0; return; }
    // grouped probabilities per sample: states 0/1 and 3/4 share the larger of the two (useAllStates == false)
    double pg[4][NSTATE];
    for (int s = 0; s < K.sp; s++) {
        const double* row = pmf + ((size_t)(lo * nsamples + s) * NSTATE) * stride + idxS[(size_t)s * N + g];
        const double p0 = row[0], p1 = row[(size_t)stride], p2 = row[(size_t)2 * stride], p3 = row[(size_t)3 * stride], p4 = row[(size_t)4 * stride];
        const double lo01 = p0 > p1 ? p0 : p1, hi34 = p3 > p4 ? p3 : p4;      // Math.Max
        pg[s][0] = lo01; pg[s][1] = lo01; pg[s][2] = p2; pg[s][3] = hi34; pg[s][4] = hi34;
    }
    for (int j = 0; j < NSTATE; j++) {
        double best = -1.7976931348623157e308;
        for (int k = 0; k < K.n[j]; k++) {
            double em = 1.0;
            for (int s = 0; s < K.sp; s++) em *= pg[s][K.g[j][k][s]];
            if (em != em || em == INFINITY || em == -INFINITY) em = 0.0;
            if (best < em) best = em;
        }
        maxL[(size_t)j * N + g] = best;
    }
}
__global__ void __launch_bounds__(256) k_iota_i32(int32_t* __restrict__ p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = (int32_t)i;
}

// ---- host: emission tables (DistributionUtilities.cs:51-69).  GammaLn/FactorialLn (MathNet) -> lgamma; Math.Pow(x,2) := x*x.
// lgamma(x + 1.0) for integer x does not depend on the sample: filled once per process (same call, same argument, same value)
static const double* lgamma1_table(int n) {
    static std::mutex mu;
    static std::vector<double> tab(70016);
    static int filled = 0;
    std::lock_guard<std::mutex> lk(mu);
    if (n > (int)tab.size()) return nullptr;
    for (; filled < n; filled++) tab[filled] = std::lgamma((double)filled + 1.0);
    return tab.data();
}
static void negative_binomial_log_table(double mean, double variance, int maxValue, double* out) {
    const double* lg1 = lgamma1_table(maxValue);
    double m = std::max(mean, 0.1);
    double r = (m * m) / (std::max(variance, mean * 1.2) - mean);
    r = std::max(2.0, r);
    // the first and the last term do not depend on x: evaluated once, same calls and arguments, so the values are unchanged
    const double term0 = std::log(std::pow(1 + mean / r, -r)), lgr = std::lgamma(r), ratio = mean / (mean + r);
    for (int x = 0; x < maxValue; x++) {
        double dens = std::exp(term0 + std::log(std::pow(ratio, (double)x)) + std::lgamma(r + x) -
                               (lg1 ? lg1[x] : std::lgamma((double)x + 1.0)) - lgr);
        if (std::isnan(dens) || std::isinf(dens)) dens = 0;
        out[x] = std::log(dens);   // EstimateViterbiLikelihood takes Math.Log of the table value (Distributions.cs:322)
    }
}

// The five emission tables of a sample are independent: helper threads of the context fill four of them while the calling thread builds the descriptors of the stage and
// fills the fifth (same calls, same arguments, same values; 1400 entries of pow / log / lgamma / exp / log are 50-90 us on one core — on the critical path of every pass,
// with the device idle: the quartiles they depend on have only just arrived)
#include <condition_variable>
struct NbJob { double mean, variance; int len; double* out; };
class NbPool {
    std::vector<std::thread> th; std::mutex mu; std::condition_variable cv, cvDone; std::vector<NbJob> jobs; size_t next = 0; int pending = 0; bool stop = false;
    void loop() {
        for (;;) {
            NbJob j;
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return stop || next < jobs.size(); }); if (stop) return; j = jobs[next++]; }
            negative_binomial_log_table(j.mean, j.variance, j.len, j.out);
            { std::lock_guard<std::mutex> lk(mu); if (--pending == 0) cvDone.notify_all(); }
        }
    }
public:
    explicit NbPool(int n) { for (int i = 0; i < n; i++) th.emplace_back([this] { loop(); }); }
    ~NbPool() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); for (auto& t : th) t.join(); }
    void submit(const NbJob* j, int n) { { std::lock_guard<std::mutex> lk(mu); jobs.assign(j, j + n); next = 0; pending = n; } cv.notify_all(); }
    void finish() {                         // the caller takes what is still waiting, then waits for the helpers
        for (;;) {
            NbJob j;
            { std::lock_guard<std::mutex> lk(mu); if (next >= jobs.size()) break; j = jobs[next++]; }
            negative_binomial_log_table(j.mean, j.variance, j.len, j.out);
            { std::lock_guard<std::mutex> lk(mu); --pending; }
        }
        std::unique_lock<std::mutex> lk(mu); cvDone.wait(lk, [&] { return pending == 0; });
    }
};

// Utilities.Quartiles index logic (shared with clean.hip semantics)
static void quartile_idx(int64_t n, int64_t* idx, int& cnt) {
    cnt = 0;
    int64_t mid = n / 2;
    if (n % 2 == 0) {
        int64_t mm = mid / 2;
        idx[cnt++] = mid - 1; idx[cnt++] = mid;
        if (mid % 2 == 0) { idx[cnt++] = mm - 1; idx[cnt++] = mm; idx[cnt++] = mid + mm - 1; idx[cnt++] = mid + mm; }
        else { idx[cnt++] = mm; idx[cnt++] = mm + mid; }
    } else {
        idx[cnt++] = mid;
        if ((n - 1) % 4 == 0) { int64_t k = (n - 1) / 4; idx[cnt++] = k - 1; idx[cnt++] = k; idx[cnt++] = 3 * k; idx[cnt++] = 3 * k + 1; }
        else { int64_t k = (n - 3) / 4; idx[cnt++] = k; idx[cnt++] = k + 1; idx[cnt++] = 3 * k + 1; idx[cnt++] = 3 * k + 2; }
    }
}
static void quartile_val(int64_t n, const float* v, float& q1, float& q2, float& q3) {
    int64_t mid = n / 2;
    if (n % 2 == 0) {
        q2 = (v[0] + v[1]) / 2;
        if (mid % 2 == 0) { q1 = (v[2] + v[3]) / 2; q3 = (v[4] + v[5]) / 2; } else { q1 = v[2]; q3 = v[3]; }
    } else {
        q2 = v[0];
        if ((n - 1) % 4 == 0) { q1 = (v[1] * 0.25f) + (v[2] * 0.75f); q3 = (v[3] * 0.75f) + (v[4] * 0.25f); }
        else { q1 = (v[1] * 0.75f) + (v[2] * 0.25f); q3 = (v[3] * 0.25f) + (v[4] * 0.75f); }
    }
}

static inline unsigned nblk2(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

// Shared Viterbi driver of the per-sample and the joint mode.  `prepare` fills the emission source after the common workspace has been
// carved: idx[N] (table index per bin), the log-emission table dTab ([5][P.tableLen]) and P.  Everything after that — speculation,
// backtrack, exact backbone, verification, sequential fallback — is mode independent: the likelihood of a step is
// log-emission_j(t) + logA[i][j] in both modes (for the joint mode see canvas_hmm_joint).
struct HmmEmis { int32_t* idx = nullptr; double* dTab = nullptr;
                 const double* indexCov = nullptr; };   // != NULL (PerSampleHMM): idx[i] = Convert.ToInt32(min(indexCov[i], P.maxThreshold)) is filled by the set-up kernel
// the running segment ids of the bins (canvas_segment_ids without forbidden intervals / reference ploidy) enqueued right behind the verification: one synchronisation
// serves both; when a chromosome takes another attempt the caller derives the ids once the states are final (*valid stays false)
struct SegPost { const int32_t* d_start; const int32_t* d_stop; int32_t maxDist; int32_t* d_segment_id; int64_t nseg = 0; bool valid = false; };
static void enqueue_segment_ids(canvas_ctx* ctx, const int64_t* dOff, int nchr, const int32_t* d_state, const int32_t* d_start, const int32_t* d_stop, int64_t N, int32_t maxDist,
                                uint8_t* flags, uint32_t* blockCnt, unsigned long long* dTot, int32_t* d_segment_id, unsigned* totSeq = nullptr, unsigned seq = 0);
template <class PrepareA, class PrepareB>
static int32_t hmm_pipeline(canvas_ctx* ctx, int32_t nchr, const int64_t* h_chr_offset, size_t extraBytes, PrepareA prepareA, PrepareB prepareB, int32_t* d_state, SegPost* seg = nullptr,
                            bool descByValue = false /* prepareA launches nothing that reads the descriptor tables: they may arrive with the set-up kernel */) {
    const int64_t N = h_chr_offset[nchr];
    // VB-step blocks (speculation, verification and backtrack share them)
    std::vector<HmmChrom> chroms(nchr);
    std::vector<int32_t> firstBlock(nchr + 1);
    int nblocks = 0;
    for (int c = 0; c < nchr; c++) {
        chroms[c].begin = h_chr_offset[c]; chroms[c].T = h_chr_offset[c + 1] - h_chr_offset[c];
        firstBlock[c] = nblocks;
        if (chroms[c].T > 10) nblocks += (int)((chroms[c].T + VB - 1) / VB);
    }
    firstBlock[nchr] = nblocks;
    std::vector<int32_t> firstGroup(nchr + 1);                  // groups of 64 blocks for the backtrack (counted from the end of each chromosome)
    int ngroups = 0;
    for (int c = 0; c < nchr; c++) { firstGroup[c] = ngroups; ngroups += (firstBlock[c + 1] - firstBlock[c] + 63) / 64; }
    firstGroup[nchr] = ngroups;
    std::vector<int32_t> firstS(nchr + 1);                      // blocks of the speculative pass (VBS steps)
    int nblocksS = 0;
    for (int c = 0; c < nchr; c++) { firstS[c] = nblocksS; if (chroms[c].T > 10) nblocksS += (int)((chroms[c].T + VBS - 1) / VBS); }
    firstS[nchr] = nblocksS;
    std::vector<int32_t> firstChunk(nchr + 1);
    int nchunks = 0;
    for (int c = 0; c < nchr; c++) {
        firstChunk[c] = nchunks;
        if (chroms[c].T > 10) nchunks += (int)((chroms[c].T + BB_CHUNK - 1) / BB_CHUNK);
    }
    firstChunk[nchr] = nchunks;
    const int nbSeg = (int)nblk2(N, 2048);
    WsSizer sz;
    sz.take<BbChunk>(nchunks + 1); sz.take<int32_t>(nchr + 1); sz.take<double>(nchunks + 1); sz.take<double>(nchunks + 1); sz.take<BbChunkOut>(nchunks + 1);
    sz.take<BbCross>((size_t)nchunks * BB_MAXC + 1); sz.take<ParFn>((size_t)nchunks * 16 + 1); sz.take<uint8_t>((size_t)nchunks * 16 + 8);
    sz.take<unsigned long long>(nchunks + 1); sz.take<BbPost>((size_t)nchunks * BB_MAXC + 1);
    sz.take<uint16_t>(N + 8); sz.take<HmmChrom>(nchr); sz.take<int32_t>(nchr);
    sz.take<int32_t>(nchr + 1); sz.take<uint16_t>(nblocks + 8); sz.take<int8_t>(nblocks + 8);
    sz.take<int64_t>(nchr + 1); sz.take<VitBlock>(nblocks + 1); sz.take<double>(N); sz.take<double>(N + 64); sz.take<int32_t>(nchr); sz.take<int32_t>(nchr);
    sz.take<VitBlock>(nblocksS + 1); sz.take<uint16_t>(nblocksS + 8); sz.take<int32_t>(nchr + 1);
    sz.take<int32_t>(nchr + 1); sz.take<uint16_t>(nblocks + 8); sz.take<uint16_t>(ngroups + 8); sz.take<int8_t>(ngroups + 8);
    sz.take<char>(8 * 256 + (size_t)nchr * 64 + 64);
    if (seg) { sz.take<uint8_t>(N + 16); sz.take<uint32_t>(nbSeg + 1); sz.take<unsigned long long>(1); }
    int32_t rc = canvas_ws_reserve(ctx, sz.off + extraBytes + 65536); if (rc) return rc;
    rc = canvas_pin_reserve(ctx, (size_t)nchr * 4 + 64); if (rc) return rc;         // verification flags (+ the segment count and its mailbox stamp) come back here
    WsCarver ws(ctx->ws);
    uint16_t* psi = ws.take<uint16_t>(N + 8);
    HmmChrom* dChroms = ws.take<HmmChrom>(nchr); int32_t* dLast = ws.take<int32_t>(nchr);
    int32_t* dFirst = ws.take<int32_t>(nchr + 1);
    uint16_t* dMaps = ws.take<uint16_t>(nblocks + 8); (void)ws.take<int8_t>(nblocks + 8);
    int64_t* dOffDev = ws.take<int64_t>(nchr + 1);
    VitBlock* dVBlocks = ws.take<VitBlock>(nblocks + 1); double* dD = ws.take<double>(N); double* dCarry = ws.take<double>(N + 64); int32_t* dFail = ws.take<int32_t>(nchr); int32_t* dRedo = ws.take<int32_t>(nchr);
    int32_t* dFirstGroup = ws.take<int32_t>(nchr + 1); uint16_t* dPre = ws.take<uint16_t>(nblocks + 8); uint16_t* dGmap = ws.take<uint16_t>(ngroups + 8);
    int8_t* dGentry = ws.take<int8_t>(ngroups + 8);
    VitBlock* dSBlocks = ws.take<VitBlock>(nblocksS + 1); uint16_t* dMapsS = ws.take<uint16_t>(nblocksS + 8); int32_t* dFirstS = ws.take<int32_t>(nchr + 1);
    BbChunk* dBChunks = ws.take<BbChunk>(nchunks + 1); int32_t* dFirstChunk = ws.take<int32_t>(nchr + 1);
    double* dChunkSum = ws.take<double>(nchunks + 1); double* dChunkBase = ws.take<double>(nchunks + 1); BbChunkOut* dChunkOut = ws.take<BbChunkOut>(nchunks + 1);
    BbCross* dCross = ws.take<BbCross>((size_t)nchunks * BB_MAXC + 1); ParFn* dAt64Fn = ws.take<ParFn>((size_t)nchunks * 16 + 1); uint8_t* dAt64Rank = ws.take<uint8_t>((size_t)nchunks * 16 + 8);
    unsigned long long* dChunkBits = ws.take<unsigned long long>(nchunks + 1); BbPost* dPost = ws.take<BbPost>((size_t)nchunks * BB_MAXC + 1);
    uint8_t* segFlags = nullptr; uint32_t* segBlockCnt = nullptr; unsigned long long* segTot = nullptr;
    if (seg) { segFlags = ws.take<uint8_t>(N + 16); segBlockCnt = ws.take<uint32_t>(nbSeg + 1); segTot = ws.take<unsigned long long>(1); seg->valid = false; }

    // static descriptors: six small tables, by value with the set-up kernel (<= HMM_BYVAL chromosomes) or in ONE packed upload
    const bool byVal = descByValue && nchr <= HMM_BYVAL;
    HmmDescPack pack;
    {
        auto al = [](size_t v) { return (v + 255) & ~size_t(255); };
        const size_t oCh = 0, oFirst = al((size_t)nchr * sizeof(HmmChrom)), oOff = oFirst + al((size_t)(nchr + 1) * 4), oChunk = oOff + al((size_t)(nchr + 1) * 8),
                     oS = oChunk + al((size_t)(nchr + 1) * 4), oG = oS + al((size_t)(nchr + 1) * 4), total = oG + al((size_t)(nchr + 1) * 4);
        char* dBlob = ws.take<char>(total);
        if (byVal) {
            memset(&pack, 0, sizeof pack);
            memcpy(pack.chroms, chroms.data(), (size_t)nchr * sizeof(HmmChrom)); memcpy(pack.off, h_chr_offset, (size_t)(nchr + 1) * 8); memcpy(pack.first, firstBlock.data(), (size_t)(nchr + 1) * 4);
            memcpy(pack.firstChunk, firstChunk.data(), (size_t)(nchr + 1) * 4); memcpy(pack.firstS, firstS.data(), (size_t)(nchr + 1) * 4); memcpy(pack.firstGroup, firstGroup.data(), (size_t)(nchr + 1) * 4);
        } else {
            std::vector<char> blob(total, 0);
            memcpy(blob.data() + oCh, chroms.data(), (size_t)nchr * sizeof(HmmChrom)); memcpy(blob.data() + oFirst, firstBlock.data(), (size_t)(nchr + 1) * 4);
            memcpy(blob.data() + oOff, h_chr_offset, (size_t)(nchr + 1) * 8); memcpy(blob.data() + oChunk, firstChunk.data(), (size_t)(nchr + 1) * 4);
            memcpy(blob.data() + oS, firstS.data(), (size_t)(nchr + 1) * 4); memcpy(blob.data() + oG, firstGroup.data(), (size_t)(nchr + 1) * 4);
            rc = canvas_h2d_small(ctx, dBlob, blob.data(), total); if (rc) return rc;
        }
        dChroms = (HmmChrom*)(dBlob + oCh); dFirst = (int32_t*)(dBlob + oFirst); dOffDev = (int64_t*)(dBlob + oOff); dFirstChunk = (int32_t*)(dBlob + oChunk);
        dFirstS = (int32_t*)(dBlob + oS); dFirstGroup = (int32_t*)(dBlob + oG);
    }
    // phase A of the mode: everything the set-up kernel needs (PerSampleHMM: the threshold); phase B: the emission tables, computed on the host while the set-up kernel runs
    HmmParams P; HmmEmis E;
    ctx->hmm_dispersion = 1e9;          // (prepareA of the per-sample model sets it; the joint model keeps the long first lead-in)
    rc = prepareA(ws, P, E, (const HmmChrom*)dChroms, (const int64_t*)dOffDev); if (rc) return rc;
    {
        const int64_t most = std::max<int64_t>(std::max<int64_t>(E.indexCov ? N : 0, nblocks), std::max<int64_t>(std::max(nblocksS, nchunks), nchr));
        HmmDescDev DD{dChroms, dOffDev, dFirst, dFirstChunk, dFirstS, dFirstGroup};
        hipLaunchKernelGGL((k_hmm_setup<VitBlock, BbChunk>), dim3(nblk2(most, 256)), dim3(256), 0, ctx->stream, pack, byVal ? 1 : 0, DD, nchr, nblocks, nblocksS, nchunks, VB, VBS, BB_CHUNK,
                           dVBlocks, dSBlocks, dBChunks, E.indexCov, N, P.maxThreshold, E.idx, dLast, dFail);
    }
    rc = prepareB(ws, P, E); if (rc) return rc;
    int32_t* idx = E.idx; double* dTab = E.dTab;
    size_t tabBytes = (size_t)NSTATE * P.tableLen * 8;
    size_t lds = tabBytes <= 48 * 1024 ? tabBytes : 0;
    for (int c = 0; c < nchr; c++)
        if (chroms[c].T <= 10 && chroms[c].T > 0)
            hipLaunchKernelGGL(k_fill_i32, dim3(nblk2(chroms[c].T, 256)), dim3(256), 0, ctx->stream, d_state, chroms[c].begin, chroms[c].begin + chroms[c].T, -1);
    const unsigned laneGrid = (unsigned)((nblocks + 63) / 64);          // one lane per block
    auto backtrack = [&](bool haveMaps) {
        if (nblocks > 0) {
            if (!haveMaps) hipLaunchKernelGGL(k_bt_maps, dim3(nblk2(nblocks, 256)), dim3(256), 0, ctx->stream, dVBlocks, nblocks, dChroms, psi, dMaps);
            hipLaunchKernelGGL(k_bt_group, dim3(ngroups), dim3(64), 0, ctx->stream, dFirst, dFirstGroup, nchr, dMaps, dPre, dGmap);
            hipLaunchKernelGGL(k_bt_gchain, dim3(nchr), dim3(64), 0, ctx->stream, dFirstGroup, dLast, dGmap, dGentry);
            hipLaunchKernelGGL(k_bt_states, dim3(nblk2(nblocks, 4)), dim3(256), 0, ctx->stream, dVBlocks, nblocks, dChroms, psi, dFirst, dFirstGroup, dPre, dGentry, d_state);
        }
    };
    // one value on the diagonal of the transition matrix and one off it (bit patterns compared): the speculative pass then adds two constants per state instead of five
    bool twoValuedAll = true;
    for (int i = 0; i < NSTATE; i++) for (int j = 0; j < NSTATE; j++) {
        const double ref = i == j ? P.logA[0][0] : P.logA[0][1];
        if (memcmp(&P.logA[i][j], &ref, sizeof(double)) != 0) twoValuedAll = false;
    }
    const bool speculative = cvx_hook("CANVAS_HMM_SEQUENTIAL") == nullptr;
    std::vector<int32_t> redo;
    int32_t* hFail = (int32_t*)ctx->pin; unsigned long long* hSegTot = (unsigned long long*)((char*)ctx->pin + (((size_t)nchr * 4 + 15) & ~size_t(15)));
    bool segEnqueued = false; unsigned* hSegSeq = (unsigned*)(hSegTot + 1); unsigned segSeq = 0;
    if (speculative && nblocks > 0) {
        ProfScope ps(ctx, "viterbi");
        // attempt 0: lead-ins of 128 / 64 steps.  Noisy samples (states that overlap heavily) forget their history more slowly: chromosomes
        // whose verification fails are tried again with 8x and then 64x longer lead-ins before the sequential kernel takes them (57 ms for a chr1-size chromosome: a noisy
        // sample used to fall off that cliff 2-15 times per soak run; the 64x attempt costs about 0.6 ms for the same chromosome).
        const int32_t* dTodo = nullptr;
        ctx->hmm_retry = 0;
        // (a fourth attempt with 512x: 65 536 / 32 768 steps reach the start of every chromosome of up to ~65 000 bins, where both passes then start from the exact
        // initial state — the last fallbacks of the noise-40 soak were such chromosomes, whose off-backbone states did not re-anchor within 4 096 steps; for a chr1-size
        // chromosome the attempt costs ~7 ms, an eighth of the sequential kernel)
        // (round 5: two thirds of a speculative lane's steps were lead-in.  At the coverage CanvasPartition sees, the five states' paths merge within a handful of bins, so the
        // first attempt now starts VW0 = 16 steps in front of its block — k_vit_spec 122 -> 60 us on the WGS sample — and a chromosome that fails its verification first gets
        // the 128-step lead-in of the earlier rounds, with the same cheap backbone, before the long lead-ins with the plain chain)
        // How fast the paths merge goes with how well the states are separated, i.e. with the sample's relative dispersion r = IQR / median of the coverage (known on the host:
        // the emission model is built from the same quartiles).  Share of randomised PerSampleHMM calls that needed the second attempt (tools/soak.py, SOAK_DISPERSION=1):
        // r < 0.20: 0 % at 16 steps; 0.20-0.30: 8-34 % at 16, 4-7 % at 32, 0.2 % at 64; r >= 0.30: most at 16 / 32, 5 % (r < 0.4) to 25-90 % at 64 — those keep 128
        static const int vwEnv = cvx_hook("CANVAS_HMM_LEAD") ? atoi(cvx_hook("CANVAS_HMM_LEAD")) : 0;      // (test hook: the first attempt's cold-start lead-in)
        const int vw0 = vwEnv > 0 ? vwEnv : (ctx->hmm_dispersion < 0.20 ? VW0 : (ctx->hmm_dispersion < 0.30 ? 64 : VW));
        const int nAttempts = 5;
        for (int attempt = 0; attempt < nAttempts; attempt++) {
            const int mult = attempt <= 1 ? 1 : (attempt == 2 ? 8 : (attempt == 3 ? 64 : 512));
            const int leadSpec = attempt == 0 ? vw0 : mult * VW, leadVer = mult * VW2;
            if (attempt > 0) CANVAS_HIP_TRY(ctx, hipMemsetAsync(dFail, 0, nchr * 4, ctx->stream));      // (attempt 0: cleared by the set-up kernel)
            const dim3 gs((unsigned)((nblocksS + 63) / 64));
            // (the last two attempts run the recurrence in the reference's own form: the two-constant form of the transition term rounds differently, and a near-tie that it
            // resolves the other way fails the verification whatever the lead-in — together with a lead-in that reaches the chromosome's start the guess is then exact)
            const bool twoValued = twoValuedAll && attempt < 3;
            if (lds && twoValued) hipLaunchKernelGGL((k_vit_spec<true, true>), gs, dim3(64), lds, ctx->stream, dSBlocks, nblocksS, dChroms, idx, dTab, P, psi, dMapsS, dLast, leadSpec, dTodo, VBS);
            else if (lds) hipLaunchKernelGGL((k_vit_spec<true, false>), gs, dim3(64), lds, ctx->stream, dSBlocks, nblocksS, dChroms, idx, dTab, P, psi, dMapsS, dLast, leadSpec, dTodo, VBS);
            else if (twoValued) hipLaunchKernelGGL((k_vit_spec<false, true>), gs, dim3(64), 0, ctx->stream, dSBlocks, nblocksS, dChroms, idx, dTab, P, psi, dMapsS, dLast, leadSpec, dTodo, VBS);
            else hipLaunchKernelGGL((k_vit_spec<false, false>), gs, dim3(64), 0, ctx->stream, dSBlocks, nblocksS, dChroms, idx, dTab, P, psi, dMapsS, dLast, leadSpec, dTodo, VBS);
            hipLaunchKernelGGL(k_pair_maps, dim3(nblk2(nblocks, 256)), dim3(256), 0, ctx->stream, dVBlocks, nblocks, dChroms, dFirstS, dMapsS, dMaps, dTodo);
            if (attempt == 0 && cvx_hook("CANVAS_HMM_TEST_CORRUPT")) hipLaunchKernelGGL(k_vit_corrupt, dim3(1), dim3(64), 0, ctx->stream, psi, chroms[0].begin + chroms[0].T / 2);
            backtrack(true);
            hipLaunchKernelGGL(k_vit_increments, dim3(nblk2(N, 256)), dim3(256), 0, ctx->stream, dChroms, nchr, dOffDev, idx, dTab, P, d_state, N, dD);
            const char* bbMode = cvx_hook("CANVAS_HMM_BACKBONE");      // "chain" | "scan" | default: predicted pieces
            // (a retry takes the plain chain for its backbone: the predicted pieces below give up on increments they cannot bracket — more binade crossings in a chunk than
            // they keep, exponents that do not behave — and no longer lead-in changes that; the chain is one FP64 add per step, 1.5 ms for a chr1-size chromosome)
            if ((bbMode && !strcmp(bbMode, "chain")) || (attempt > 1 && !bbMode)) hipLaunchKernelGGL(k_vit_backbone, dim3(nchr), dim3(64), 0, ctx->stream, dChroms, dD, dCarry, dTodo);
            else if (bbMode && !strcmp(bbMode, "scan")) hipLaunchKernelGGL(k_vit_backbone_scan, dim3(nchr), dim3(BS_T), 0, ctx->stream, dChroms, dD, dCarry, dFail);
            else {
                hipLaunchKernelGGL(k_bb_sums, dim3(nchunks), dim3(256), 0, ctx->stream, dBChunks, dChroms, dD, dChunkSum, dFail);
                hipLaunchKernelGGL(k_bb_bases, dim3(nchr), dim3(64), 0, ctx->stream, dFirstChunk, dChunkSum, dChunkBase);
                hipLaunchKernelGGL(k_bb_pieces, dim3(nchunks), dim3(256), 0, ctx->stream, dBChunks, dChroms, dD, dChunkBase, dChunkOut, dCross, dAt64Fn, dAt64Rank, dFail);
                hipLaunchKernelGGL(k_bb_walk, dim3(nchr), dim3(64), 0, ctx->stream, dFirstChunk, dChunkOut, dCross, dChunkBits, dPost, dFail);
                hipLaunchKernelGGL(k_bb_emit, dim3(nblk2((int64_t)nchunks * 16, 256)), dim3(256), 0, ctx->stream, dBChunks, nchunks, dChroms, dChunkBits, dPost, dAt64Fn, dAt64Rank, dCarry);
            }
            if (lds) hipLaunchKernelGGL((k_vit_verify<true>), dim3(laneGrid), dim3(64), lds, ctx->stream, dVBlocks, nblocks, dChroms, idx, dTab, P, psi, d_state, dD, dCarry, dLast, dFail, leadVer, dTodo);
            else hipLaunchKernelGGL((k_vit_verify<false>), dim3(laneGrid), dim3(64), 0, ctx->stream, dVBlocks, nblocks, dChroms, idx, dTab, P, psi, d_state, dD, dCarry, dLast, dFail, leadVer, dTodo);
            CANVAS_HIP_TRY(ctx, hipMemcpyAsync(hFail, dFail, nchr * 4, hipMemcpyDeviceToHost, ctx->stream));
            if (seg && attempt == 0 && !cvx_hook("CANVAS_HMM_TEST_CORRUPT")) {
                // the states are final unless a chromosome fails its verification (rare): the segment ids are derived now, under the same synchronisation
                (void)segTot;
                segSeq = cvx_mail_arm(ctx, hSegSeq);
                enqueue_segment_ids(ctx, dOffDev, nchr, d_state, seg->d_start, seg->d_stop, N, seg->maxDist, segFlags, segBlockCnt, hSegTot /* pinned: the count is written straight to the host */, seg->d_segment_id, hSegSeq, segSeq);
                segEnqueued = true;
            }
            CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            redo.clear();
            for (int c = 0; c < nchr; c++) if (hFail[c] && chroms[c].T > 10) redo.push_back(c);
            if (!redo.empty() && cvx_hook("CANVAS_HMM_DEBUG_FAIL")) { fprintf(stderr, "viterbi attempt %d (lead-in x%d):", attempt, mult); for (int c : redo) fprintf(stderr, " chr%d(T=%lld, why=0x%x)", c, (long long)chroms[c].T, (unsigned)hFail[c]); fprintf(stderr, "\n"); }
            if (redo.empty() || cvx_hook("CANVAS_HMM_TEST_CORRUPT") || cvx_hook("CANVAS_HMM_NO_RETRY")) break;
            if (attempt < nAttempts - 1) {       // the failed chromosomes become the to-do mask of the next attempt (dRedo doubles as the mask: nchr entries)
                if (attempt == 0) { ctx->hmm_retry = (int)redo.size(); { ProfScope pr(ctx, "viterbi_retry"); } }     // counted for the tests / bench
                int32_t rcq = canvas_h2d_small(ctx, dRedo, hFail, nchr * 4); if (rcq) return rcq;
                dTodo = dRedo;
            }
        }
        if (seg && segEnqueued && redo.empty() && ctx->hmm_retry == 0) { int32_t rcm = cvx_mail_await(ctx, hSegSeq, segSeq, "PerSampleHMM: segment count"); if (rcm) return rcm; seg->valid = true; seg->nseg = (int64_t)*hSegTot; }
    } else {
        for (int c = 0; c < nchr; c++) redo.push_back(c);
    }
    if (!redo.empty()) {
        // exact sequential evaluation (all chromosomes when speculation is disabled, otherwise only those whose verification failed)
        // ("viterbi_sequential" counts FALLBACKS — chromosomes the speculative attempts gave up on; a call without any block to speculate on, every chromosome at most ten bins
        // long or empty, runs the same kernel for its few steps under another name: 27 of the 27 'fallbacks' of a noise-40 soak were such calls)
        const bool isFallback = speculative && nblocks > 0;
        ProfScope ps(ctx, isFallback ? "viterbi_sequential" : "viterbi_small");
        if (cvx_hook("CANVAS_HMM_DEBUG_FAIL")) { fprintf(stderr, "viterbi sequential kernel for %zu chromosomes (speculative %d, blocks %d):", redo.size(), (int)speculative, nblocks); for (size_t i = 0; i < redo.size() && i < 6; i++) fprintf(stderr, " chr%d(T=%lld)", redo[i], (long long)chroms[redo[i]].T); fprintf(stderr, "\n"); }
        rc = canvas_h2d_small(ctx, dRedo, redo.data(), redo.size() * 4); if (rc) return rc;
        hipLaunchKernelGGL(k_viterbi, dim3((unsigned)redo.size()), dim3(64), lds, ctx->stream, dChroms, idx, dTab, P, psi, dLast, dRedo);
        backtrack(false);
    }
    ctx->hmm_redo = (int)redo.size();
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // host vectors feed async copies
    CANVAS_HIP_TRY(ctx, hipGetLastError());
    return CANVAS_OK;
}

extern "C" {

static int32_t hmm_per_sample_impl(canvas_ctx* ctx, int32_t nchr, const double* d_cov, const int64_t* h_chr_offset, const double* d_cov_all, int64_t nAll, int32_t* d_state, const CovQ* preQ = nullptr, SegPost* seg = nullptr) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (nchr <= 0 || !d_cov || !h_chr_offset || !d_state || !d_cov_all) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_hmm_per_sample: bad arguments");
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t N = h_chr_offset[nchr];
    // nAll / d_cov_all: the genome the quartiles are taken over (the whole sample; == the chromosomes handled here unless the sample is sharded)
    if (nAll < 5) CANVAS_FAIL(ctx, CANVAS_ERR_UNSUPPORTED, "HMM: fewer than 5 bins genome-wide (Quartiles would throw in the reference)");
    WsSizer ex; ex.take<uint32_t>(nAll); ex.take<int32_t>(N + 256); ex.take<double>(NSTATE * 70000); ex.take<CovQ>(1); ex.take<uint32_t>(CQ_WIN);
    double haploidMean = 0, pseudoVariance = 0;
    std::vector<double> tab;
    struct Joiner { NbPool* p = nullptr; ~Joiner() { if (p) p->finish(); } } joiner;      // (an early return must not leave helpers writing into `tab`)
    auto prepareA = [&](WsCarver& ws, HmmParams& P, HmmEmis& E, const HmmChrom*, const int64_t*) -> int32_t {
        int32_t rc;
        uint32_t* keys = ws.take<uint32_t>(nAll); int32_t* idx = ws.take<int32_t>(N + 256); double* dTab = ws.take<double>(NSTATE * 70000);   // idx: padded for the group loads of k_vit_spec
    // 1. genome-wide quartiles of (float)coverage (HiddenMarkovModelsRunner.cs:36-50): by counting when the coverage is F2 text (k_covq_hist), by the radix select otherwise
    //    (CANVAS_HMM_RADIX_SELECT=1 forces the latter: test hook, both must agree)
    int64_t qidx[6]; int nq;
    quartile_idx(nAll, qidx, nq);
    float v[6], q1, q2, q3;
    bool radix = cvx_hook("CANVAS_HMM_RADIX_SELECT") != nullptr;
    if (!radix && preQ) { rc = cvx_mail_await(ctx, &preQ->pad, ctx->covq_seq, "PerSampleHMM: coverage quartiles"); if (rc) return rc; }      // (preQ is always the context's pinned block: cvx_quant_covq_enqueue)
    if (!radix && preQ && preQ->nq == (uint32_t)nq && preQ->n == nAll) {        // counted while the coverage was quantised (cvx_quantize_f2_covq): the ranks are already on the host
        bool same = true; for (int k = 0; k < nq; k++) same = same && preQ->rank[k] == (unsigned long long)qidx[k];
        if (same && !preQ->bad && !preQ->fail) for (int k = 0; k < nq; k++) v[k] = (float)((double)preQ->resultK[k] / 100.0);
        else radix = true;
    } else if (!radix) {
        CovQ* dQ = ws.take<CovQ>(1); uint32_t* dWin = ws.take<uint32_t>(CQ_WIN);
        CovQ hQ; memset(&hQ, 0, sizeof hQ); hQ.nq = (uint32_t)nq;
        for (int k = 0; k < nq; k++) hQ.rank[k] = (unsigned long long)qidx[k];
        rc = canvas_h2d_small(ctx, dQ, &hQ, sizeof hQ); if (rc) return rc;
        CANVAS_HIP_TRY(ctx, hipMemsetAsync(dWin, 0, CQ_WIN * 4, ctx->stream));
        hipLaunchKernelGGL(k_covq_hist, dim3(256), dim3(1024), 0, ctx->stream, d_cov_all, nAll, dQ, dWin);
        hipLaunchKernelGGL(k_covq_pick, dim3(1), dim3(1024), 0, ctx->stream, dQ, dWin);
        rc = canvas_pin_reserve(ctx, sizeof(CovQ)); if (rc) return rc;
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(ctx->pin, dQ, sizeof(CovQ), hipMemcpyDeviceToHost, ctx->stream));
        CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        CANVAS_HIP_TRY(ctx, hipGetLastError());
        const CovQ& R = *(const CovQ*)ctx->pin;
        if (R.bad || R.fail) radix = true;                 // coverage that is not F2 text, or a sample whose quartiles lie more than 40 units from its level
        else for (int k = 0; k < nq; k++) v[k] = (float)((double)R.resultK[k] / 100.0);
    }
    if (radix) {
        hipLaunchKernelGGL(k_keys_cov_f32, dim3(nblk2(nAll, 256)), dim3(256), 0, ctx->stream, d_cov_all, nAll, keys);
        std::vector<SelQuery> qs;
        for (int k = 0; k < nq; k++) qs.push_back({0, 0, qidx[k]});
        std::vector<unsigned long long> res;
        rc = radix_select<uint32_t>(ctx, keys, 1, std::vector<int64_t>{0, nAll}, qs, res); if (rc) return rc;
        for (int k = 0; k < nq; k++) v[k] = host_float_of_key((uint32_t)res[k]);
    }
    quartile_val(nAll, v, q1, q2, q3);
    const double median = (double)q2;
    const float iqr = q3 - q1;
    ctx->hmm_dispersion = q2 > 0 ? (double)iqr / (double)q2 : 1e9;      // (chooses the first attempt's lead-in: hmm_pipeline)
    pseudoVariance = (double)(iqr * iqr);
    // 2. emission tables (HiddenMarkovModelsRunner.cs:111-152)
    haploidMean = median / 2.0;
    P.maxThreshold = haploidMean * NSTATE;
    if (!(P.maxThreshold >= 0) || P.maxThreshold > 60000) CANVAS_FAIL(ctx, CANVAS_ERR_UNSUPPORTED, "HMM: coverage scale outside the supported table size");
    P.tableLen = (int32_t)std::nearbyint(P.maxThreshold) + 10 + 1;      // >= max over chromosomes of (maxValues + 10)
        E.idx = idx; E.dTab = dTab; E.indexCov = d_cov;                 // the table indices need the threshold only: the set-up kernel fills them while the host fills the tables below
        // the emission tables start now, on the context's helper threads (the calling thread joins in once the set-up kernel is enqueued)
        if (!ctx->hmm_pool) ctx->hmm_pool = std::make_shared<NbPool>(NSTATE - 1);
        (void)lgamma1_table(P.tableLen);                                 // (filled once per process, before the helpers read it)
        tab.assign((size_t)NSTATE * P.tableLen, 0.0);
        NbJob jobs[NSTATE];
        for (int CN = 0; CN < NSTATE; CN++) jobs[CN] = NbJob{std::max((double)CN, 0.1) * haploidMean, pseudoVariance, P.tableLen, &tab[(size_t)CN * P.tableLen]};
        joiner.p = static_cast<NbPool*>(ctx->hmm_pool.get());
        joiner.p->submit(jobs, NSTATE);
        return CANVAS_OK;
    };
    auto prepareB = [&](WsCarver&, HmmParams& P, HmmEmis& E) -> int32_t {
    std::static_pointer_cast<NbPool>(ctx->hmm_pool)->finish();
    const double selfTransition = 0.99;
    for (int i = 0; i < NSTATE; i++) {
        for (int j = 0; j < NSTATE; j++) P.logA[i][j] = std::log(i == j ? selfTransition : (1.0 - selfTransition) / (NSTATE - 1));
        P.logPi[i] = std::log((double)(1.0f / NSTATE));      // 1f / nStates widened (HMM.cs:41)
    }
    // 3. Viterbi and backtrack: hmm_pipeline
        return canvas_h2d_small(ctx, E.dTab, tab.data(), tab.size() * 8);          // (through the pinned staging area: `tab` is a local)
    };
    return hmm_pipeline(ctx, nchr, h_chr_offset, ex.off, prepareA, prepareB, d_state, seg, true);
}
int32_t canvas_hmm_per_sample(canvas_ctx* ctx, int32_t nchr, const double* d_cov, const int64_t* h_chr_offset, int32_t* d_state) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (nchr <= 0 || !h_chr_offset) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_hmm_per_sample: bad arguments");
    return hmm_per_sample_impl(ctx, nchr, d_cov, h_chr_offset, d_cov, h_chr_offset[nchr], d_state);
}

}  // extern "C"

int32_t cvx_hmm_per_sample_preq(canvas_ctx* ctx, int32_t nchr, const double* d_cov, const int64_t* h_chr_offset, int32_t* d_state, const void* h_covq) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (nchr <= 0 || !h_chr_offset) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_hmm_per_sample: bad arguments");
    return hmm_per_sample_impl(ctx, nchr, d_cov, h_chr_offset, d_cov, h_chr_offset[nchr], d_state, (const CovQ*)h_covq);
}
// PerSampleHMM + the running segment ids (canvas_segment_ids) with the ids enqueued behind the verification of the speculative Viterbi pass: one synchronisation for both
int32_t cvx_hmm_per_sample_segments(canvas_ctx* ctx, int32_t nchr, const double* d_cov, const int64_t* h_chr_offset, int32_t* d_state, const void* h_covq,
                                    const int32_t* d_start, const int32_t* d_stop, int32_t max_inter_bin_dist, int32_t* d_segment_id, int64_t* h_nsegments) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (nchr <= 0 || !h_chr_offset || !d_start || !d_stop || !d_segment_id) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_sample_pipeline: bad arguments");
    SegPost sp; sp.d_start = d_start; sp.d_stop = d_stop; sp.maxDist = max_inter_bin_dist; sp.d_segment_id = d_segment_id;
    const int64_t N = h_chr_offset[nchr];
    int32_t rc = hmm_per_sample_impl(ctx, nchr, d_cov, h_chr_offset, d_cov, N, d_state, (const CovQ*)h_covq, N > 0 ? &sp : nullptr); if (rc) return rc;
    if (sp.valid) { if (h_nsegments) *h_nsegments = sp.nseg; return CANVAS_OK; }
    return canvas_segment_ids(ctx, nchr, h_chr_offset, d_state, d_start, d_stop, max_inter_bin_dist, d_segment_id, h_nsegments);
}
// d_n != NULL: the number of bins is min(n, *d_n), read on the device (the call is enqueued behind the CanvasClean that produces it)
int32_t cvx_quant_covq_enqueue(canvas_ctx* ctx, const float* d_count, int64_t n, const unsigned long long* d_n, double* d_cov, const void** h_covq_out) {
    if (!ctx) return CANVAS_ERR_INVALID;
    *h_covq_out = nullptr;
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->covq_dev) { CANVAS_HIP_TRY(ctx, hipMalloc(&ctx->covq_dev, 512 + CQ_WIN * 4)); CANVAS_HIP_TRY(ctx, hipMemsetAsync(ctx->covq_dev, 0, 512 + CQ_WIN * 4, ctx->stream)); }      // k_covq_pick leaves the counters zero
    if (!ctx->covq_pin) CANVAS_HIP_TRY(ctx, hipHostMalloc(&ctx->covq_pin, 256, hipHostMallocDefault));
    CovQ* dQ = (CovQ*)ctx->covq_dev; CovQ* dQres = (CovQ*)ctx->covq_pin; uint32_t* dWin = (uint32_t*)((char*)ctx->covq_dev + 512);       // (the result goes straight into pinned host memory)
    static_assert(sizeof(CovQ) <= 256, "CovQ");
    hipLaunchKernelGGL(k_quant_covq, dim3(256), dim3(1024), 0, ctx->stream, d_count, n, d_cov, dQ, dWin, d_n);      // (128 / 192 / 512 / 1 024 workgroups: 50 / 39 / 46 / 60 us against 35 — the loop shrinks and the flush of the counters grows with the grid)
    ctx->covq_seq = cvx_mail_arm(ctx, &dQres->pad);
    hipLaunchKernelGGL(k_covq_pick, dim3(1), dim3(1024), 0, ctx->stream, dQ, dWin, (long long)n, d_n, dQres, ctx->covq_seq);
    *h_covq_out = ctx->covq_pin;
    return CANVAS_OK;
}
int32_t cvx_quantize_f2_covq(canvas_ctx* ctx, const float* d_count, int64_t n, double* d_cov, const void** h_covq_out) {
    if (!ctx) return CANVAS_ERR_INVALID;
    *h_covq_out = nullptr;
    if (n < 5 || cvx_hook("CANVAS_HMM_RADIX_SELECT")) return canvas_quantize_f2(ctx, d_count, n, d_cov);     // (fewer than 5 bins: PerSampleHMM refuses the sample anyway)
    return cvx_quant_covq_enqueue(ctx, d_count, n, nullptr, d_cov, h_covq_out);
}

int32_t cvx_hmm_per_sample_subset(canvas_ctx* ctx, int32_t nchr, const double* d_cov, const int64_t* h_chr_offset, const double* d_cov_all, int64_t n_all, int32_t* d_state, const void* h_covq) {
    return hmm_per_sample_impl(ctx, nchr, d_cov, h_chr_offset, d_cov_all, n_all, d_state, (const CovQ*)h_covq);
}

// NegativeBinomialWrapper density table (DistributionUtilities.cs:51-69), the same calls as negative_binomial_log_table without the final log
static void negative_binomial_table(double mean, double variance, int maxValue, double* out) {
    double m = std::max(mean, 0.1);
    double r = (m * m) / (std::max(variance, mean * 1.2) - mean);
    r = std::max(2.0, r);
    const double term0 = std::log(std::pow(1 + mean / r, -r)), lgr = std::lgamma(r), ratio = mean / (mean + r);
    for (int x = 0; x < maxValue; x++) {
        double dens = std::exp(term0 + std::log(std::pow(ratio, (double)x)) + std::lgamma(r + x) - std::lgamma((double)x + 1.0) - lgr);
        if (std::isnan(dens) || std::isinf(dens)) dens = 0;
        out[x] = dens;
    }
}
// DistributionUtilities.GetGenotypeCombinations (DistributionUtilities.cs:11-40): for every state the distinct arrangements of
// {state x (S'-d), diploid x d}, d = 0 .. S'-1, each multiset sorted and then enumerated in lexicographic order (Permutations.cs:399-433)
static void genotype_combinations(int nsamples, JointCombos& K) {
    const int sp = std::min(nsamples, 4);
    K.sp = sp;
    for (int j = 0; j < NSTATE; j++) {
        int n = 0;
        if (j == 2) { for (int s = 0; s < 4; s++) K.g[j][0][s] = 2; n = 1; }
        else for (int nd = 0; nd < sp; nd++) {
            std::vector<int> st(sp - nd, j); st.insert(st.end(), nd, 2);
            std::sort(st.begin(), st.end());
            do { for (int s = 0; s < 4; s++) K.g[j][n][s] = (int8_t)(s < sp ? st[s] : 2); n++; } while (std::next_permutation(st.begin(), st.end()));
        }
        K.n[j] = n;
    }
}
template <class F> static void parallel_for(int64_t n, int64_t grain, F f) {
    unsigned hw = std::thread::hardware_concurrency(); if (hw == 0) hw = 4;
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(hw, (n + grain - 1) / grain));
    if (nt == 1) { f(0, n); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++) th.emplace_back([=]() { f(n * t / nt, n * (t + 1) / nt); });
    for (auto& t : th) t.join();
}

extern "C" int32_t canvas_hmm_joint(canvas_ctx* ctx, int32_t nsamples, int32_t nchr, const double* const* d_cov, const int64_t* h_chr_offset, int32_t* d_state) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (nsamples <= 0 || nsamples > JOINT_MAXS || nchr <= 0 || !d_cov || !h_chr_offset || !d_state) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_hmm_joint: bad arguments (1..16 samples)");
    for (int d = 0; d < nsamples; d++) if (!d_cov[d]) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_hmm_joint: null coverage pointer");
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t N = h_chr_offset[nchr];
    if (N <= 0 || N > 0x7FFFFFFFll) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_hmm_joint: bin count out of range");
    const int S = nsamples;
    JointPtrs ptrs; for (int d = 0; d < JOINT_MAXS; d++) ptrs.cov[d] = d < S ? d_cov[d] : nullptr;
    std::vector<HmmChrom> chroms(nchr);
    for (int c = 0; c < nchr; c++) { chroms[c].begin = h_chr_offset[c]; chroms[c].T = h_chr_offset[c + 1] - h_chr_offset[c]; }
    // ---- phase 1: per chromosome and sample median, mean and variance of the uncapped coverage (HiddenMarkovModelsRunner.cs:117-131)
    const size_t nCS = (size_t)nchr * S;
    char* small = nullptr;
    const size_t smallBytes = ((nchr * sizeof(HmmChrom) + 255) & ~size_t(255)) + 3 * ((nCS * 8 + 255) & ~size_t(255));
    CANVAS_HIP_TRY(ctx, hipMalloc((void**)&small, smallBytes));
    struct Free { canvas_ctx* c; char* p; ~Free() { (void)hipStreamSynchronize(c->stream); (void)hipFree(p); } } fr{ctx, small};
    HmmChrom* dCh = (HmmChrom*)small;
    double* dMed = (double*)(small + ((nchr * sizeof(HmmChrom) + 255) & ~size_t(255)));
    double* dSum0 = dMed + ((nCS * 8 + 255) & ~size_t(255)) / 8; double* dSum1 = dSum0 + ((nCS * 8 + 255) & ~size_t(255)) / 8;
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dCh, chroms.data(), nchr * sizeof(HmmChrom), hipMemcpyHostToDevice, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipMemsetAsync(dMed, 0, 3 * ((nCS * 8 + 255) & ~size_t(255)), ctx->stream));
    hipLaunchKernelGGL(k_joint_median, dim3(nchr, S), dim3(1024), 0, ctx->stream, ptrs, dCh, S, dMed);
    hipLaunchKernelGGL(k_joint_seqsum, dim3(nchr, S), dim3(64), 0, ctx->stream, ptrs, dCh, S, 0, (const double*)nullptr, dSum0);
    hipLaunchKernelGGL(k_joint_seqsum, dim3(nchr, S), dim3(64), 0, ctx->stream, ptrs, dCh, S, 1, (const double*)dSum0, dSum1);
    std::vector<double> hMed(nCS), hS2(nCS);
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(hMed.data(), dMed, nCS * 8, hipMemcpyDeviceToHost, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(hS2.data(), dSum1, nCS * 8, hipMemcpyDeviceToHost, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    CANVAS_HIP_TRY(ctx, hipGetLastError());
    std::vector<double> haploid(nCS, 0.0), variance(nCS, 0.0), thr(nchr, 0.0);
    int strideUb = 16;
    for (int c = 0; c < nchr; c++) {
        if (chroms[c].T <= 10) continue;
        double mx = 0;
        for (int d = 0; d < S; d++) {
            haploid[c * S + d] = std::max(1.0, hMed[c * S + d]) / 2.0;
            variance[c * S + d] = hS2[c * S + d] / (double)(chroms[c].T - 1);
            mx = d == 0 ? haploid[c * S + d] : std::max(mx, haploid[c * S + d]);
        }
        thr[c] = mx * NSTATE;
        if (!(thr[c] >= 0) || thr[c] > 60000) CANVAS_FAIL(ctx, CANVAS_ERR_UNSUPPORTED, "HMM: coverage scale outside the supported table size");
        strideUb = std::max(strideUb, (int)std::nearbyint(thr[c]) + 11);
    }
    JointCombos K; genotype_combinations(S, K);
    // ---- phase 2 inside the shared pipeline: indices, tables, per-bin maxima, host log, then Viterbi
    WsSizer ex; ex.take<int32_t>(N + 256); ex.take<int32_t>((size_t)S * N); ex.take<double>((size_t)NSTATE * N); ex.take<double>(nCS * NSTATE * strideUb); ex.take<double>(nchr); ex.take<int32_t>(nchr);
    auto prepare = [&](WsCarver& ws, HmmParams& P, HmmEmis& E, const HmmChrom* dChroms, const int64_t* dOffDev) -> int32_t {
        int32_t* idx = ws.take<int32_t>(N + 256); int32_t* idxS = ws.take<int32_t>((size_t)S * N); double* dL = ws.take<double>((size_t)NSTATE * N);
        double* dPmf = ws.take<double>(nCS * NSTATE * strideUb); double* dThr = ws.take<double>(nchr); int32_t* dMaxIdx = ws.take<int32_t>(nchr);
        int32_t rc = canvas_h2d_small(ctx, dThr, thr.data(), nchr * 8); if (rc) return rc;
        CANVAS_HIP_TRY(ctx, hipMemsetAsync(dMaxIdx, 0, nchr * 4, ctx->stream));
        hipLaunchKernelGGL(k_joint_index, dim3(nblk2(N, 256)), dim3(256), 0, ctx->stream, ptrs, S, dOffDev, nchr, dThr, N, idxS, dMaxIdx);
        std::vector<int32_t> hMax(nchr);
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(hMax.data(), dMaxIdx, nchr * 4, hipMemcpyDeviceToHost, ctx->stream));
        CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        int stride = 16;
        for (int c = 0; c < nchr; c++) if (chroms[c].T > 10) stride = std::max(stride, hMax[c] + 10);
        if (stride > strideUb) CANVAS_FAIL(ctx, CANVAS_ERR_HIP, "canvas_hmm_joint: internal table bound exceeded");
        std::vector<double> tab(nCS * NSTATE * (size_t)stride, 0.0);
        parallel_for((int64_t)nCS * NSTATE, 1, [&](int64_t a, int64_t b) {
            for (int64_t i = a; i < b; i++) {
                const int c = (int)(i / (S * NSTATE)), d = (int)((i / NSTATE) % S), CN = (int)(i % NSTATE);
                if (chroms[c].T <= 10) continue;
                negative_binomial_table(std::max((double)CN, 0.1) * haploid[c * S + d], variance[c * S + d], hMax[c] + 10, &tab[(size_t)i * stride]);   // HiddenMarkovModelsRunner.cs:136-147
            }
        });
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dPmf, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_joint_emission, dim3(nblk2(N, 256)), dim3(256), 0, ctx->stream, idxS, S, dOffDev, nchr, dChroms, dPmf, stride, K, N, dL);
        std::vector<double> hL((size_t)NSTATE * N);
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(hL.data(), dL, hL.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
        CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        CANVAS_HIP_TRY(ctx, hipGetLastError());
        parallel_for((int64_t)hL.size(), 1 << 16, [&](int64_t a, int64_t b) { for (int64_t i = a; i < b; i++) hL[i] = std::log(hL[i]); });   // Math.Log(maxLikelyhood), Distributions.cs:322
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dL, hL.data(), hL.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_iota_i32, dim3(nblk2(N, 256)), dim3(256), 0, ctx->stream, idx, N);
        CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));       // hL / tab are local
        P.tableLen = (int32_t)N; P.maxThreshold = 0.0;
        const double selfTransition = 0.99;
        for (int i = 0; i < NSTATE; i++) {
            for (int j = 0; j < NSTATE; j++) P.logA[i][j] = std::log(i == j ? selfTransition : (1.0 - selfTransition) / (NSTATE - 1));
            P.logPi[i] = std::log((double)(1.0f / NSTATE));
        }
        E.idx = idx; E.dTab = dL;
        return CANVAS_OK;
    };
    return hmm_pipeline(ctx, nchr, h_chr_offset, ex.off, prepare, [](WsCarver&, HmmParams&, HmmEmis&) -> int32_t { return CANVAS_OK; }, d_state);
}

extern "C" {

}  // extern "C"
static void enqueue_segment_ids(canvas_ctx* ctx, const int64_t* dOff, int nchr, const int32_t* d_state, const int32_t* d_start, const int32_t* d_stop, int64_t N, int32_t maxDist,
                                uint8_t* flags, uint32_t* blockCnt, unsigned long long* dTot, int32_t* d_segment_id, unsigned* totSeq, unsigned seq) {
    const int nb = (int)nblk2(N, 2048);
    hipLaunchKernelGGL(k_seg_flags, dim3(nblk2(N, 256)), dim3(256), 0, ctx->stream, dOff, nchr, d_state, d_start, d_stop, N, maxDist, (const int64_t*)nullptr, (const int32_t*)nullptr, (const int32_t*)nullptr,
                       (const int64_t*)nullptr, (const int32_t*)nullptr, (const int32_t*)nullptr, (const int32_t*)nullptr, flags);
    hipLaunchKernelGGL(k_count_blocks, dim3(nb), dim3(256), 0, ctx->stream, flags, N, blockCnt);
    hipLaunchKernelGGL(k_scan_blocks2, dim3(1), dim3(1024), 0, ctx->stream, blockCnt, nb, dTot, totSeq, seq);
    hipLaunchKernelGGL(k_seg_ids, dim3(nb), dim3(256), 0, ctx->stream, flags, blockCnt, N, d_segment_id);
}
extern "C" {
int32_t canvas_segment_ids_ploidy(canvas_ctx* ctx, int32_t nchr, const int64_t* h_chr_offset, const int32_t* d_state, const int32_t* d_start,
                                  const int32_t* d_stop, int32_t max_inter_bin_dist, const int64_t* h_excl_offset, const int32_t* h_excl_start,
                                  const int32_t* h_excl_stop, const int64_t* h_ploidy_offset, const int32_t* h_ploidy_start, const int32_t* h_ploidy_end,
                                  const int32_t* h_ploidy_cn, int32_t* d_segment_id, int64_t* h_nsegments) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (nchr <= 0 || !h_chr_offset || !d_state || !d_start || !d_stop || !d_segment_id) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_segment_ids: bad arguments");
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t N = h_chr_offset[nchr];
    if (N == 0) { if (h_nsegments) *h_nsegments = 0; return CANVAS_OK; }
    const int nb = (int)nblk2(N, 2048);
    const int64_t nex = h_excl_offset ? h_excl_offset[nchr] : 0;
    if (h_excl_offset && nex > 0 && (!h_excl_start || !h_excl_stop)) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_segment_ids_filtered: missing interval arrays");
    if (h_excl_offset) for (int c = 0; c < nchr; c++) for (int64_t k = h_excl_offset[c] + 1; k < h_excl_offset[c + 1]; k++)
        if (h_excl_stop[k] < h_excl_stop[k - 1]) CANVAS_FAIL(ctx, CANVAS_ERR_UNSUPPORTED, "forbidden intervals must be sorted by end within a chromosome");
    const int64_t npl = h_ploidy_offset ? h_ploidy_offset[nchr] : 0;
    if (h_ploidy_offset && npl > 0 && (!h_ploidy_start || !h_ploidy_end || !h_ploidy_cn)) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_segment_ids_ploidy: missing ploidy arrays");
    for (int64_t k = 0; k < npl; k++) if (h_ploidy_cn[k] < 0 || h_ploidy_cn[k] > 4)
        CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "reference ploidy outside 0..4 (PloidyInfo.getPloidyCounts indexes a 5-element array: the reference throws)");
    WsSizer sz; sz.take<int64_t>(nchr + 1); sz.take<uint8_t>(N); sz.take<uint32_t>(nb + 1); sz.take<unsigned long long>(1);
    sz.take<int64_t>(nchr + 1); sz.take<int32_t>(nex + 1); sz.take<int32_t>(nex + 1);
    sz.take<int64_t>(nchr + 1); sz.take<int32_t>(npl + 1); sz.take<int32_t>(npl + 1); sz.take<int32_t>(npl + 1);
    int32_t rc = canvas_ws_reserve(ctx, sz.off + 4096); if (rc) return rc;
    WsCarver ws(ctx->ws);
    int64_t* dOff = ws.take<int64_t>(nchr + 1); uint8_t* flags = ws.take<uint8_t>(N); uint32_t* blockCnt = ws.take<uint32_t>(nb + 1); unsigned long long* dTot = ws.take<unsigned long long>(1);
    int64_t* dExOff = ws.take<int64_t>(nchr + 1); int32_t* dExStart = ws.take<int32_t>(nex + 1); int32_t* dExStop = ws.take<int32_t>(nex + 1);
    int64_t* dPlOff = ws.take<int64_t>(nchr + 1); int32_t* dPlStart = ws.take<int32_t>(npl + 1); int32_t* dPlEnd = ws.take<int32_t>(npl + 1); int32_t* dPlCn = ws.take<int32_t>(npl + 1);
    if (h_ploidy_offset) {
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dPlOff, h_ploidy_offset, (nchr + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        if (npl > 0) { CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dPlStart, h_ploidy_start, npl * 4, hipMemcpyHostToDevice, ctx->stream));
                       CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dPlEnd, h_ploidy_end, npl * 4, hipMemcpyHostToDevice, ctx->stream));
                       CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dPlCn, h_ploidy_cn, npl * 4, hipMemcpyHostToDevice, ctx->stream)); }
    }
    if (h_excl_offset) {
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dExOff, h_excl_offset, (nchr + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        if (nex > 0) { CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dExStart, h_excl_start, nex * 4, hipMemcpyHostToDevice, ctx->stream));
                       CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dExStop, h_excl_stop, nex * 4, hipMemcpyHostToDevice, ctx->stream)); }
    }
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dOff, h_chr_offset, (nchr + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_seg_flags, dim3(nblk2(N, 256)), dim3(256), 0, ctx->stream, dOff, nchr, d_state, d_start, d_stop, N, max_inter_bin_dist, h_excl_offset ? dExOff : (const int64_t*)nullptr, dExStart, dExStop,
                       h_ploidy_offset ? dPlOff : (const int64_t*)nullptr, dPlStart, dPlEnd, dPlCn, flags);
    hipLaunchKernelGGL(k_count_blocks, dim3(nb), dim3(256), 0, ctx->stream, flags, N, blockCnt);
    hipLaunchKernelGGL(k_scan_blocks2, dim3(1), dim3(1024), 0, ctx->stream, blockCnt, nb, dTot);
    hipLaunchKernelGGL(k_seg_ids, dim3(nb), dim3(256), 0, ctx->stream, flags, blockCnt, N, d_segment_id);
    unsigned long long tot = 0;
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(&tot, dTot, 8, hipMemcpyDeviceToHost, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    CANVAS_HIP_TRY(ctx, hipGetLastError());
    if (h_nsegments) *h_nsegments = (int64_t)tot;
    return CANVAS_OK;
}

int32_t canvas_segment_ids_filtered(canvas_ctx* ctx, int32_t nchr, const int64_t* h_chr_offset, const int32_t* d_state, const int32_t* d_start,
                                    const int32_t* d_stop, int32_t max_inter_bin_dist, const int64_t* h_excl_offset, const int32_t* h_excl_start,
                                    const int32_t* h_excl_stop, int32_t* d_segment_id, int64_t* h_nsegments) {
    return canvas_segment_ids_ploidy(ctx, nchr, h_chr_offset, d_state, d_start, d_stop, max_inter_bin_dist, h_excl_offset, h_excl_start, h_excl_stop,
                                     nullptr, nullptr, nullptr, nullptr, d_segment_id, h_nsegments);
}

int32_t canvas_segment_ids(canvas_ctx* ctx, int32_t nchr, const int64_t* h_chr_offset, const int32_t* d_state, const int32_t* d_start,
                           const int32_t* d_stop, int32_t max_inter_bin_dist, int32_t* d_segment_id, int64_t* h_nsegments) {
    return canvas_segment_ids_filtered(ctx, nchr, h_chr_offset, d_state, d_start, d_stop, max_inter_bin_dist, nullptr, nullptr, nullptr, d_segment_id, h_nsegments);
}

// GenomeSegmentationResults.SplitOverlappingSegments (GenomeSegmentationResults.cs:18-55), one chromosome, host scalar code
// (a few hundred segments per sample): union of all samples' starts/ends -> minimal non-overlapping partition.
int32_t canvas_split_overlapping(int32_t nsamples, const uint32_t* const* h_start, const uint32_t* const* h_end, const int32_t* h_nseg,
                                 uint32_t* h_out_start, uint32_t* h_out_end, int32_t cap, int32_t* h_nout) {
    if (nsamples <= 0 || !h_start || !h_end || !h_nseg || !h_nout) return CANVAS_ERR_INVALID;
    if (nsamples == 1) {   // a single sample is returned as is (:20)
        if (h_nseg[0] > cap) return CANVAS_ERR_CAPACITY;
        for (int i = 0; i < h_nseg[0]; i++) { h_out_start[i] = h_start[0][i]; h_out_end[i] = h_end[0][i]; }
        *h_nout = h_nseg[0];
        return CANVAS_OK;
    }
    std::vector<std::pair<uint32_t, int>> ev;
    for (int s = 0; s < nsamples; s++) for (int i = 0; i < h_nseg[s]; i++) { ev.push_back({h_start[s][i], 0}); ev.push_back({h_end[s][i], 1}); }
    std::stable_sort(ev.begin(), ev.end(), [](const std::pair<uint32_t, int>& a, const std::pair<uint32_t, int>& b) { return a.first < b.first; });
    int overlapping = 0, n = 0; uint32_t cur = 0;
    for (auto& e : ev) {
        if (overlapping > 0 && cur != e.first) { if (n < cap) { h_out_start[n] = cur; h_out_end[n] = e.first; } n++; }
        cur = e.first; overlapping += e.second == 0 ? 1 : -1;
    }
    *h_nout = n;
    return n > cap ? CANVAS_ERR_CAPACITY : CANVAS_OK;
}

}  // extern "C"
