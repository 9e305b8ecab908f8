// CanvasPartition CBS on MI355X: CBSRunner.Run (CBSRunner.cs:40-151), ChangePoint.ChangePoints / FindChangePoints
// (ChangePoint.cs:44-153,291-400), CBSTStatistic.TMaxO / HTMaxP / TMaxP / TPermP (CBSTStatistic.cs), GetBoundary, TailProbability.
//
// Division of labour (DESIGN.md "CBS"):
//   * The recursion is a depth-first stack per chromosome that shares ONE Mersenne-Twister stream, every floating sum that feeds
//     a decision is a sequential left-to-right double accumulation (mean, TSS, prefix sums sx; CBSTStatistic.cs:74-81), and
//     TPermP (CBSTStatistic.cs:947-1024) is a single chain of nPerm x m1 dependent random swaps.  Those are latency-bound scalar
//     chains; they run on the host exactly as in the reference (one std::thread per chromosome, CBSRunner.cs:115-147).
//   * The arithmetic bulk of TMaxO — the maximum of n/(k(n-k)) (S_j - S_i)^2 over all O(n^2) circular arcs — runs on the GPU as an
//     exhaustive, pruning-free search (k_arc_search: one lane per arc length, LDS-staged prefix sums, 2 FP64 ops per arc).
//     The reference's sqrt(n)-block pruning is lossless, so the exhaustive maximum is the same double; the arg-max is
//     the reference's as long as the maximising arc is unique, which the host verifies per call — on an exact floating-point tie it
//     replays the reference's block order on the host (rare).
//   * Permutation reference distribution of the hybrid test (XPerm + HTMaxP, segments of >= 1024 bins): device engine below.
//     - the chromosome's MT19937 stream is continued on the device: a sequential history of 19937*128 outputs per batch, the rest
//       with the 134-term recurrence of the characteristic polynomial at stride 128 (data parallel, phi(x)^128 = phi(x^128));
//     - the Fisher-Yates swaps are evaluated without replaying them (counting sort of the swap targets + pointer doubling);
//     - every statistic comes back as an interval from a worst-case rounding bound; a permutation is re-evaluated in the
//       reference's exact order only when the observed statistic lies inside the interval (never, in the tests and the WGS run);
//     - the sequential stopping rule and the generator hand-over (state rebuilt from the last 624 outputs) run on the host;
//     - launcher threads batch the requests of all chromosome threads, so kernels of different chromosomes overlap.
//     TMaxP (n <= 200), short hybrid segments, TPermP and TailP (its series of normal-CDF evaluations runs on a host thread pool)
//     stay on the host.  WGS-size run (4.7 M bins, 104 k permutations over 1.27e9 elements): 4.1 s host-only -> 1.5 s.
#include "common.hpp"
#include "cbs_boundary_default.hpp"
#include "cbs_mt_jump.hpp"
#include "../../include/canvas_mathnet.h"
#include <algorithm>
#include <atomic>
#include <cmath>
#include <limits>
#include <mutex>
#include <thread>
#include <chrono>
#include <functional>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>

// ================================================================================================ device: Nu(x) of the analytic tail probability
// TailProbability.Nu (TailProbability.cs:52-85): l1 = log 2 - 2 log x - sum_dk 2 Phi(-x sqrt(dk) / 2) / dk, summed in blocks of 2, 2, 4, 8, ... terms until the relative change of
// a block drops below tol; up to ~10^6 normal-CDF evaluations for the small arguments of long segments, 100 arguments per TailP call — 10 host thread-seconds per WGS sample.
// One workgroup per argument evaluates the blocks in parallel.  The sums are re-associated and erfc / log / exp are the device's, so the value is an APPROXIMATION (relative error
// far below 1e-9); it only feeds two decisions of FindChangePoints (p1 > cutoff; nrejc = (int)((cutoff - p1) nPerm), ChangePoint.cs:318-323), and the host accepts it only when both
// come out the same for every p1 within 1e-8 relative — otherwise, and whenever a stopping comparison of the series itself is within 1e-6 relative of tol (flag), the call is
// redone with the host libm in the reference's order.
// The 100 arguments of a call travel as the kernel's argument block and the results are written straight into the engine's pinned mailbox (values, flags, then — by the last
// workgroup to finish — the call's sequence number with release semantics, common.hpp cvx_mail_*): a call is ONE launch and one synchronisation where it used to be an upload,
// the launch and two downloads (three of every five copies of a CBS call were these).
struct TailArgs { double x[100]; };
__global__ void __launch_bounds__(256) k_tail_nu(const TailArgs A, int n, double tol, double* __restrict__ nus /* pinned host */, int* __restrict__ flags /* pinned host */,
                                                 unsigned* __restrict__ doneCnt /* device, zero between calls */, unsigned* __restrict__ seqWord /* pinned host */, unsigned seq) {
    __shared__ double sh[4];
    const int g = blockIdx.x; if (g >= n) return;
    const double x = A.x[g];
    const int t = threadIdx.x;
    auto finish = [&](double nu, int flag) {      // thread 0 of the workgroup
        nus[g] = nu; flags[g] = flag;
        __threadfence_system();
        const unsigned old = __hip_atomic_fetch_add(doneCnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned)n - 1u) { __hip_atomic_store(doneCnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); cvx_mail_publish(seqWord, seq); }
    };
    if (!(x > 0.01)) { if (t == 0) finish(exp(-0.583 * x), 0); return; }
    double l1 = log(2.0) - 2.0 * log(x), l0 = l1;
    long long dk = 0; long long k = 2; int flag = 0;
    // A block of the series is sum_{d = dk + 1}^{dk + cnt} f(d), f(d) = 2 Phi(-x sqrt(d) / 2) / d = erfc(a sqrt(d)) / d with a = x / (2 sqrt 2); from the second block on dk = cnt
    // = D, i.e. the block is d in (D, 2 D].  Up to D = 256 every term is evaluated (one per thread).  From D = 512 on the block is taken from the Euler-Maclaurin formula in its
    // midpoint form,  sum = int_{D + 1/2}^{2 D + 1/2} f(t) dt - (f1(2 D + 1/2) - f1(D + 1/2)) / 24 + O(7 / 5760 * 6 / D^4) with f1 the first derivative:  the integral is
    // 2 int erfc(u) / u du over u = a sqrt(t), i.e. int 2 erfc(e^v) dv over an interval of width ln(2) / 2 — sixteen 16-point Gauss-Legendre panels, one node per thread —
    // and f1(t) = -erfc(a sqrt t) / t^2 - a e^{-a^2 t} / (sqrt(pi) t^{3/2}).  Against the term-by-term sum the block is off by < 1e-13 at D = 512 and 16 x less with every
    // doubling (tools/tail_em_check.py; the series itself is ~ 10, the value only feeds decisions the host accepts when they hold for every p1 within 1e-8 relative).  The
    // long segments of a WGS sample ask for x down to 0.012, i.e. blocks up to D = 2^20: ~2 M erfc evaluations per argument became ~5 000 (k_tail_nu: 9-12 % of the kernel
    // time of a CBS call, 165-470 us per launch; now tens of us).
    const double a = x / 2.0 / 1.4142135623730951;
    auto block_sum = [&](double acc) -> double {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
        __syncthreads();
        if ((t & 63) == 0) sh[t >> 6] = acc;
        __syncthreads();
        return (sh[0] + sh[1]) + (sh[2] + sh[3]);
    };
    auto block = [&](long long cnt) -> double {           // sum_{i=1..cnt} 2 Phi(-x sqrt(dk + i) / 2) / (dk + i), all threads return the same value
        if (cnt >= 512 && dk == cnt) {
            constexpr double GX[16] = {-9.89400934991649939e-01, -9.44575023073232600e-01, -8.65631202387831755e-01, -7.55404408355002999e-01, -6.17876244402643771e-01, -4.58016777657227370e-01, -2.81603550779258915e-01, -9.50125098376374544e-02,
                                       9.50125098376374544e-02, 2.81603550779258915e-01, 4.58016777657227370e-01, 6.17876244402643771e-01, 7.55404408355002999e-01, 8.65631202387831755e-01, 9.44575023073232600e-01, 9.89400934991649939e-01};
            constexpr double GW[16] = {2.71524594117540374e-02, 6.22535239386477063e-02, 9.51585116824925914e-02, 1.24628971255534030e-01, 1.49595988816576764e-01, 1.69156519395002619e-01, 1.82603415044923612e-01, 1.89450610455068585e-01,
                                       1.89450610455068585e-01, 1.82603415044923612e-01, 1.69156519395002619e-01, 1.49595988816576764e-01, 1.24628971255534030e-01, 9.51585116824925914e-02, 6.22535239386477063e-02, 2.71524594117540374e-02};
            const double t1 = (double)cnt + 0.5, t2 = 2.0 * (double)cnt + 0.5;
            const double v1 = log(a * sqrt(t1)), v2 = log(a * sqrt(t2)), h = (v2 - v1) / 16.0;
            const int panel = t >> 4, node = t & 15;
            double gx = 0.0, gw = 0.0;
#pragma unroll
            for (int q = 0; q < 16; q++) if (q == node) { gx = GX[q]; gw = GW[q]; }      // (a select chain: a constant array indexed by a lane-dependent value would live in scratch memory)
            const double v = v1 + ((double)panel + 0.5) * h + 0.5 * h * gx;
            const double integral = block_sum(0.5 * h * gw * 2.0 * erfc(exp(v)));
            auto fprime = [&](double tt) { return -erfc(a * sqrt(tt)) / (tt * tt) - a * exp(-a * a * tt) / (1.7724538509055160273 * tt * sqrt(tt)); };
            return integral - (fprime(t2) - fprime(t1)) / 24.0;
        }
        double acc = 0.0;
        for (long long i = t + 1; i <= cnt; i += 256) { const double d = (double)(dk + i); acc += erfc(a * sqrt(d)) / d; }     // 2 * (0.5 erfc(-xk / sqrt 2)), xk = -x sqrt(d) / 2
        return block_sum(acc);
    };
    l1 = l1 - block(k); dk += k;
    for (;;) {
        const double rel = fabs((l1 - l0) / l1);
        if (fabs(rel - tol) <= tol * 1e-6) flag = 1;        // the reference's comparison could go either way: redo on the host
        if (!(rel > tol)) break;
        if (k > (1ll << 40)) { flag = 1; break; }
        l0 = l1;
        l1 = l1 - block(k); dk += k;
        k *= 2;
    }
    if (t == 0) finish(exp(l1), flag);
}

// ================================================================================================ device: exhaustive arc search
// For every arc length L in [1, n-1] (arc = pair i < j = i + L of 0-based prefix-sum indices): dmax[L] = max_i |sx[i+L] - sx[i]|,
// firstI[L] = smallest such i.  Thread t of the grid owns lengths L = t+1 and n-1-t (balanced: n iterations per thread).
#define ARC_THREADS 256
#define ARC_CHUNK 512
struct ArcReq { const double* sx; int n; double* dmax; int32_t* firstI; };
__global__ void __launch_bounds__(ARC_THREADS) k_arc_search(const ArcReq* __restrict__ reqs) {
    const ArcReq Rq = reqs[blockIdx.y];
    const double* __restrict__ sx = Rq.sx; const int n = Rq.n; double* __restrict__ dmax = Rq.dmax; int32_t* __restrict__ firstI = Rq.firstI;
    if ((int)blockIdx.x * ARC_THREADS >= n / 2 + 1) return;          // this request needs fewer workgroups than the largest one of the launch
    __shared__ double sA[ARC_CHUNK];                         // sx[i0 .. i0+CHUNK)
    __shared__ double sB[ARC_CHUNK + ARC_THREADS];           // sx[i0+Lbase .. ) window for the block's lengths
    const int half = (n - 1 + 1) / 2;                        // number of threads needed: lengths 1..n-1 paired (L, n-L)
    const int t = blockIdx.x * ARC_THREADS + threadIdx.x;
    for (int pass = 0; pass < 2; pass++) {
        // pass 0: L = t + 1 (ascending with thread id); pass 1: L = n - 1 - t (descending with thread id)
        const int L = pass == 0 ? t + 1 : n - 1 - t;
        const bool active = t < half && L >= 1 && L <= n - 1 && !(pass == 1 && L <= half);   // do not do the middle length twice
        const int Lblock0 = pass == 0 ? blockIdx.x * ARC_THREADS + 1 : n - 1 - (blockIdx.x * ARC_THREADS + ARC_THREADS - 1);   // smallest L of the block
        const int off = L - Lblock0;                         // 0..ARC_THREADS-1 (may be out of range when inactive)
        double best = -1.0; int bestI = 0;
        // longest arc count in this block: lengths >= Lblock0 (clamped to >= 1) -> i ranges over [0, n - Lmin)
        const int Lmin = Lblock0 < 1 ? 1 : Lblock0;
        const int iEnd = n - Lmin;
        for (int i0 = 0; i0 < iEnd; i0 += ARC_CHUNK) {
            __syncthreads();
            for (int k = threadIdx.x; k < ARC_CHUNK; k += ARC_THREADS) { int idx = i0 + k; sA[k] = idx < n ? sx[idx] : 0.0; }
            for (int k = threadIdx.x; k < ARC_CHUNK + ARC_THREADS; k += ARC_THREADS) { long idx = (long)i0 + Lblock0 + k; sB[k] = (idx >= 0 && idx < n) ? sx[idx] : 0.0; }
            __syncthreads();
            if (active) {
                int lim = n - L - i0; if (lim > ARC_CHUNK) lim = ARC_CHUNK;      // i < n - L
                for (int k = 0; k < lim; k++) {
                    double d = fabs(sB[k + off] - sA[k]);
                    if (d > best) { best = d; bestI = i0 + k; }
                }
            }
        }
        if (active) { dmax[L] = best; firstI[L] = bestI; }
    }
}

// ================================================================================================ device: pruned arc search
// The statistic of an arc (i, j) is c(L) * (sx[j] - sx[i])^2 with L = j - i and c(L) = n / (L (n - L)).  For blocks A <= B of 1024
// prefix sums every arc from A to B satisfies  stat <= max(c(Lmin), c(Lmax)) * max(max_B - min_A, max_A - min_B)^2  (IEEE operations
// are monotone, c is evaluated from an exact integer product, so the bound holds for the rounded values too).  Only block pairs whose
// bound reaches the incumbent (the reference's own starting point: the arc between the global extremes of the prefix sums,
// CBSTStatistic.cs:112-137) are evaluated arc by arc — typically a few dozen of the (n/1024)^2/2 pairs.  The result is the exact
// maximum, the number of arcs attaining it and the smallest (L, i) among them: what the host needs to decide between "unique
// maximiser" and "replay the reference's block order".  If more pairs survive than the list holds, the exhaustive kernel is used.
#define AP_BK 1024
#define AP_PAIRCAP 8192
struct ArcPReq { const double* sx; int n; int al0; double tau; double* bmin; double* bmax; int* pairs; unsigned long long* out /* [0] max key, [1] count, [2] min packed arc, [3] npairs, [4] overflow, [5] best block-extreme arc (bits) */; double* pairMax;
                 int* bpos /* [2 nb]: position of every block's first minimum / first maximum */;
                 unsigned long long* hOut /* pinned host: out[0..4] of the finished search */; unsigned* hSeq /* pinned host: the mailbox's sequence word */; unsigned seq; };
__global__ void __launch_bounds__(256) k_arcp_blocks(const ArcPReq* __restrict__ reqs) {
    const ArcPReq R = reqs[blockIdx.y];
    const int nb = (R.n + AP_BK - 1) / AP_BK;
    if ((int)blockIdx.x >= nb) return;
    if (blockIdx.x == 0 && threadIdx.x < 6) R.out[threadIdx.x] = threadIdx.x == 2 ? ~0ull : 0ull;      // the result words of this search (they used to be two memsets per request in front of the launch)
    __shared__ double smn[4], smx[4]; __shared__ int pmn[4], pmx[4];
    double mn = 1.7976931348623157e308, mx = -1.7976931348623157e308; int imn = 0x7fffffff, imx = 0x7fffffff;
    for (int k = threadIdx.x; k < AP_BK; k += 256) { const int i = blockIdx.x * AP_BK + k; if (i < R.n) { const double v = R.sx[i]; if (v < mn) { mn = v; imn = i; } if (v > mx) { mx = v; imx = i; } } }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const double a = __hiloint2double(__shfl_xor(__double2hiint(mn), d), __shfl_xor(__double2loint(mn), d)), b = __hiloint2double(__shfl_xor(__double2hiint(mx), d), __shfl_xor(__double2loint(mx), d));
        const int ia = __shfl_xor(imn, d), ib = __shfl_xor(imx, d);
        if (a < mn || (a == mn && ia < imn)) { mn = a; imn = ia; }
        if (b > mx || (b == mx && ib < imx)) { mx = b; imx = ib; }
    }
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; pmn[threadIdx.x >> 6] = imn; pmx[threadIdx.x >> 6] = imx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) { if (smn[w] < mn || (smn[w] == mn && pmn[w] < imn)) { mn = smn[w]; imn = pmn[w]; } if (smx[w] > mx || (smx[w] == mx && pmx[w] < imx)) { mx = smx[w]; imx = pmx[w]; } }
        R.bmin[blockIdx.x] = mn; R.bmax[blockIdx.x] = mx; R.bpos[2 * blockIdx.x] = imn; R.bpos[2 * blockIdx.x + 1] = imx;
    }
}
// the finished search's result words into the request's pinned mailbox (one workgroup per request, behind the last k_arcp_eval on the same stream)
__global__ void __launch_bounds__(64) k_arcp_mail(const ArcPReq* __restrict__ reqs) {
    const ArcPReq R = reqs[blockIdx.x];
    if (threadIdx.x < 5) { R.hOut[threadIdx.x] = R.out[threadIdx.x]; __threadfence_system(); }
    __syncthreads();
    if (threadIdx.x == 0) cvx_mail_publish(R.hSeq, R.seq);
}
__device__ __forceinline__ double arc_c(double rn, int L) { const double rj = (double)L; return rn / (rj * (rn - rj)); }
// pass 0: a better incumbent than the reference's starting arc — for every block pair the two arcs between the blocks' extremes (real arcs, evaluated exactly as k_arcp_eval
// evaluates them, so the true maximum is at least that large); pass 1: the pairs whose bound reaches the larger of the two incumbents.  (With the starting arc alone thousands of
// pairs survived on a segment without a strong change — the bound of a pair near the diagonal is loose because c(L) is large there — and k_arcp_eval was 40 % of the kernel time
// of a WGS-size CBS call; every arc that attains the maximum still lies in a surviving pair: its pair's bound is at least the maximum, which is at least the incumbent.)
__global__ void __launch_bounds__(256) k_arcp_bounds(const ArcPReq* __restrict__ reqs, int pass) {
    const ArcPReq R = reqs[blockIdx.y];
    const int nb = (R.n + AP_BK - 1) / AP_BK;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool live = idx < (long long)nb * nb;
    const int A = live ? (int)(idx / nb) : 0, B = live ? (int)(idx % nb) : 0;
    if (pass == 0) {
        __shared__ double sbest[4];
        double best = 0.0;
        if (live && B >= A) {
            const double rn = (double)R.n;
            auto arc = [&](int p, int q, double vp, double vq) {           // positions p, q and their prefix sums
                const int L = p < q ? q - p : p - q;
                if (L >= R.al0 && L <= R.n - R.al0) { const double d = fabs(vq - vp), v = arc_c(rn, L) * (d * d); best = v > best ? v : best; }
            };
            arc(R.bpos[2 * A], R.bpos[2 * B + 1], R.bmin[A], R.bmax[B]);
            arc(R.bpos[2 * A + 1], R.bpos[2 * B], R.bmax[A], R.bmin[B]);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const double o = __hiloint2double(__shfl_xor(__double2hiint(best), d), __shfl_xor(__double2loint(best), d)); best = o > best ? o : best; }
        if ((threadIdx.x & 63) == 0) sbest[threadIdx.x >> 6] = best;
        __syncthreads();
        if (threadIdx.x == 0) { for (int w = 1; w < 4; w++) best = sbest[w] > best ? sbest[w] : best; if (best > 0.0) atomicMax(&R.out[5], (unsigned long long)__double_as_longlong(best)); }
        return;
    }
    if (!live || B < A) return;
    const double tau2 = __longlong_as_double((long long)R.out[5]), tau = tau2 > R.tau ? tau2 : R.tau;
    double D = R.bmax[B] - R.bmin[A]; const double D2 = R.bmax[A] - R.bmin[B]; D = D2 > D ? D2 : D;
    if (!(D > 0.0)) return;
    const int lmin = B == A ? 1 : (B - A - 1) * AP_BK + 1;
    int lmax = (B - A + 1) * AP_BK - 1; lmax = lmax > R.n - 1 ? R.n - 1 : lmax;
    const int llo = lmin > R.al0 ? lmin : R.al0, lhi = lmax < R.n - R.al0 ? lmax : R.n - R.al0;
    if (llo > lhi) return;
    const double rn = (double)R.n, c1 = arc_c(rn, llo), c2 = arc_c(rn, lhi), c = c1 > c2 ? c1 : c2;
    if (c * (D * D) >= tau) {
        const unsigned long long slot = atomicAdd(&R.out[3], 1ull);
        if (slot < AP_PAIRCAP) R.pairs[slot] = A * 65536 + B; else R.out[4] = 1ull;
    }
}
// pass 0: maximum over the arcs of a surviving block pair; pass 1: count of the arcs that attain the global maximum + the smallest (L, i)
// (a grid-stride loop over the surviving pairs: a launch of AP_PAIRCAP workgroups per request, nearly all of which returned at once, cost 0.1-1 ms in dispatch alone)
#define AP_EVAL_GRID 512
// Inside a surviving pair the same bound is applied once more to the 16 x 16 pairs of 64-element sub-blocks (their extremes are taken from the LDS tiles): on a segment
// without a strong change thousands of pairs survive the first level — c(L) spans three orders of magnitude inside a pair near the diagonal, and the extremes of 1024
// prefix sums are far apart — but few of their sub-pairs do.  An arc that attains the maximum lies in a sub-pair whose bound is at least the maximum, so nothing is lost.
#define AP_SUB 64
#define AP_NSUB (AP_BK / AP_SUB)
__global__ void __launch_bounds__(256) k_arcp_eval(const ArcPReq* __restrict__ reqs, int pass) {
    const ArcPReq R = reqs[blockIdx.y];
    unsigned long long np = R.out[3]; if (np > AP_PAIRCAP) np = AP_PAIRCAP;
    if (R.out[4]) return;
    __shared__ double sA[AP_BK], sB[AP_BK], sC[2 * AP_BK];
    __shared__ double sMn[2][AP_NSUB], sMx[2][AP_NSUB];
    __shared__ double sred[4];
    const int n = R.n;
    const double rn = (double)n;
    const double target = __longlong_as_double((long long)R.out[0]);
    const double tau2 = __longlong_as_double((long long)R.out[5]);
    const double thr = pass == 1 ? target : (tau2 > R.tau ? tau2 : R.tau);      // pass 0: the incumbent (a lower bound of the maximum); pass 1: the maximum itself
    for (unsigned long long pi = blockIdx.x; pi < np; pi += gridDim.x) {
        if (pass == 1 && (unsigned long long)__double_as_longlong(R.pairMax[pi]) != R.out[0]) continue;
        const int A = R.pairs[pi] >> 16, B = R.pairs[pi] & 65535;
        const int baseL = (B - A) * AP_BK - (AP_BK - 1);          // L = baseL + (jj - ii + AP_BK - 1)
        __syncthreads();                                           // (the previous pair's LDS tiles are no longer read)
        for (int k = threadIdx.x; k < AP_BK; k += 256) { const int i = A * AP_BK + k, j = B * AP_BK + k; sA[k] = i < n ? R.sx[i] : 0.0; sB[k] = j < n ? R.sx[j] : 0.0; }
        for (int k = threadIdx.x; k < 2 * AP_BK - 1; k += 256) { const int L = baseL + k; sC[k] = (L >= R.al0 && L <= n - R.al0) ? arc_c(rn, L) : -1.0; }   // -1: arc length not allowed
        __syncthreads();
        if (threadIdx.x < 2 * AP_NSUB) {                           // extremes of the sub-blocks (positions beyond n do not take part)
            const int which = threadIdx.x / AP_NSUB, sb = threadIdx.x % AP_NSUB;
            const double* src = which ? sB : sA; const int g0 = (which ? B : A) * AP_BK + sb * AP_SUB;
            double mn = 1.7976931348623157e308, mx = -1.7976931348623157e308;
            for (int k = 0; k < AP_SUB; k++) if (g0 + k < n) { const double v = src[sb * AP_SUB + k]; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
            sMn[which][sb] = mn; sMx[which][sb] = mx;
        }
        __syncthreads();
        double best = -1.0; unsigned long long cnt = 0, arcMin = ~0ull;
        const int li = threadIdx.x & (AP_SUB - 1), lj0 = (threadIdx.x >> 6) * (AP_SUB / 4);      // thread: one row of the sub-pair, a quarter of its columns
        for (int sa = 0; sa < AP_NSUB; sa++) {
            if (A * AP_BK + sa * AP_SUB >= n) break;
            for (int sb = (A == B ? sa : 0); sb < AP_NSUB; sb++) {
                if (B * AP_BK + sb * AP_SUB >= n) break;
                // bound of the sub-pair (the same for every thread: no divergence)
                double D = sMx[1][sb] - sMn[0][sa]; const double D2 = sMx[0][sa] - sMn[1][sb]; D = D2 > D ? D2 : D;
                if (!(D > 0.0)) continue;
                int lmin = (B - A) * AP_BK + (sb - sa) * AP_SUB - (AP_SUB - 1), lmax = (B - A) * AP_BK + (sb - sa) * AP_SUB + (AP_SUB - 1);
                lmin = lmin < 1 ? 1 : lmin; lmax = lmax > n - 1 ? n - 1 : lmax;
                const int llo = lmin > R.al0 ? lmin : R.al0, lhi = lmax < n - R.al0 ? lmax : n - R.al0;
                if (llo > lhi) continue;
                const double c1 = arc_c(rn, llo), c2 = arc_c(rn, lhi), cb = c1 > c2 ? c1 : c2;
                if (cb * (D * D) < thr) continue;
                const int ii = sa * AP_SUB + li, i = A * AP_BK + ii;
                if (i >= n) continue;
                const double a = sA[ii];
                for (int q = 0; q < AP_SUB / 4; q++) {
                    const int jj = sb * AP_SUB + lj0 + q;
                    if (B * AP_BK + jj >= n || (A == B && jj <= ii)) continue;
                    const double c = sC[jj - ii + AP_BK - 1];
                    if (c < 0.0) continue;
                    const double d = fabs(sB[jj] - a), v = c * (d * d);
                    if (pass == 0) best = v > best ? v : best;
                    else if (v == target) { cnt++; const unsigned long long key = ((unsigned long long)(unsigned)(baseL + jj - ii + AP_BK - 1) << 32) | (unsigned)i; arcMin = key < arcMin ? key : arcMin; }
                }
            }
        }
        if (pass == 0) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { const double o = __hiloint2double(__shfl_xor(__double2hiint(best), d), __shfl_xor(__double2loint(best), d)); best = o > best ? o : best; }
            if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = best;
            __syncthreads();
            if (threadIdx.x == 0) {
                for (int w = 1; w < 4; w++) best = sred[w] > best ? sred[w] : best;
                R.pairMax[pi] = best;
                if (best >= 0.0) atomicMax(&R.out[0], (unsigned long long)__double_as_longlong(best));      // non-negative doubles order like their bit patterns
            }
        } else if (cnt) { atomicAdd(&R.out[1], cnt); atomicMin(&R.out[2], arcMin); }
    }
}

// ================================================================================================ device: permutation reference distribution
// XPerm (ChangePoint.cs:407-421) + HTMaxP (CBSTStatistic.cs:354-586) for a batch of B permutations of one segment.
//
//  k_mt_draws    MT19937 continues the chromosome's generator on the device: one workgroup regenerates the 624-word state in three
//                data-parallel phases per twist and emits B*n tempered draws in order; the generator state after every permutation is
//                snapshotted so that the host can resume exactly where the sequential stopping rule ends.
//  k_perm_stat   one workgroup per permutation.  The Fisher-Yates chain "for i = n-1..0: swap(px[i], px[j_i])" is evaluated without
//                replaying it: position i is final after step i and receives what position j_i held just before; a position q is
//                only modified by the steps that target it, so with list[q] = steps with j = q (ascending),
//                    final[s_k] = R(s_{k+1})  (x[q] for the largest step of the list),   R(i) = R(g(i)),  g(i) = min{s in list[i] : s > i}
//                (R(i) = x[i] when no such step exists).  Lists are built with a counting sort, g-chains are resolved by pointer
//                doubling: exact integer logic, the permutation IS the reference's.
//                The statistic is the maximum over arcs of length al0..k (and their complements) of c_j * (sx[b]-sx[a])^2; the
//                reference's block pruning is lossless, so the exhaustive maximum is the same set.  Prefix sums are
//                re-associated here, so every statistic is returned as an interval [lo, hi] from a worst-case rounding bound; the
//                host re-evaluates a permutation in the reference's exact order only when the observed statistic falls inside it.
#define PG_T 512
#define PG_MAXK 32
struct PermBuf { uint32_t* draws; int32_t* j; int32_t* off; int32_t* cur; int32_t* items; int32_t* g; int32_t* succ; double* px; double* sx; };
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) { y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18; return y; }
// one request = one batch of permutations of one segment; a launch serves the requests of all chromosome threads that are waiting
struct PermReq {
    uint32_t state[625];       // generator state at the start of the batch: mt[624], mti
    long long total; int n; int nb; uint32_t* snaps; const double* x; int hk, al0; double tss, errBound; PermBuf P; double* pstat; int blockBase;
    int cont;                  // this batch continues the previous one of its loop: `hist` = that batch's last MT_HISTORY outputs, no sequential part needed
    const uint32_t* hist;      // (cont) the MT_HISTORY outputs in front of position 0 of this batch — where they were written: in the OTHER of the loop's two draw buffers (they used to be
                               // copied in front of P.draws, 10 MB device to device per batch: 7 % of the device time of the tumour / normal flow's CBS went into those copies)
    int fy;                    // 0: k_perm_stat, 1: k_perm_fy, 2: k_perm_small, 3: k_perm_rp evaluates this request's permutations
    double* mailStat = nullptr; unsigned* mailSeq = nullptr; unsigned seq = 0;      // (pinned host) where k_perm_mail puts the batch's nb intervals, its sequence word and this batch's number; nullptr: pstat is copied by the launcher
    int cached = 0;            // the draws lie in the chromosome's stream cache (MtStreamCache): P.draws points INTO it, nothing is generated or snapshotted for this request
    // k_perm_rp: rpWGs persistent workgroups (blocks rpBase .. rpBase + rpWGs of its launch), each with its own scratch of rp.stride words behind rpScratch
    int rpBase, rpWGs; uint32_t* rpScratch; long long* rpClk;      // rpClk (probe only): cycles of workgroup 0 per phase
    struct RpPlan { int K, nT; uint32_t inOff[32], inCap[32]; uint32_t oEndsIn, oEndsOwn, oInbox, oOutIn, oOutOwn, stride; } rp;
};
// MT19937 is linear over GF(2): every bit of its output stream obeys the recurrence of the characteristic polynomial phi (degree 19937,
// 135 terms), i.e. out[k] = XOR_i out[k - MT_LAG[i]]; and because phi(x)^(2^m) = phi(x^(2^m)) over GF(2) the same holds with every
// lag multiplied by 2^m.  With stride MT_STRIDE the smallest lag is 623 * MT_STRIDE, so that many outputs are independent of each
// other: the generator becomes data parallel after a sequentially generated history of 19937 * MT_STRIDE outputs.
// (MT_LAG was obtained with Berlekamp-Massey from the output stream and is checked against the sequential generator by the tests.)
#define MT_STRIDE 128
#define MT_NLAG 134
#define MT_HISTORY (19937LL * MT_STRIDE)
#define MT_BOOT_MIN 600000LL      // requests with fewer draws than this generate their (whole) stream in the sequential kernel: the seven doubling launches cost ~0.5 ms, the sequential kernel ~1.6 us per 1000 draws
#define MT_WIDTH (623 * MT_STRIDE)
__constant__ int MT_LAG[MT_NLAG] = {623, 850, 1077, 1246, 1304, 1531, 1700, 1758, 1869, 1985, 2096, 2154, 2212, 2439, 2492, 2608, 2666, 2777, 2893, 3004, 3062, 3115, 3120, 3342, 3347, 3400,
    3516, 3569, 3574, 3685, 3796, 3801, 3912, 3970, 4028, 4255, 4308, 4361, 4424, 4482, 4588, 4593, 4709, 4820, 4878, 4931, 4936, 4984, 5158, 5163, 5216, 5332, 5385, 5390, 5501, 5612, 5617,
    5728, 5786, 5844, 6071, 6124, 6177, 6240, 6298, 6404, 6409, 6525, 6636, 6694, 6747, 6752, 6800, 6974, 6979, 7032, 7148, 7201, 7206, 7264, 7317, 7428, 7433, 7544, 7602, 7660, 7940, 7993,
    8056, 8099, 8220, 8225, 8326, 8452, 8553, 8563, 8616, 8722, 8780, 8790, 8848, 9017, 9176, 9244, 9809, 9968, 10036, 10432, 11731, 11958, 12185, 12354, 12412, 12460, 12808, 13368, 13600,
    14276, 15184, 15575, 15802, 16029, 16256, 16483, 16710, 16937, 17164, 17444, 18067, 18294, 18352, 18521, 18748, 19937};
// the same table as compile-time constants: k_mt_classes unrolls its 134 terms completely, so every lag is an immediate operand (with the lags fetched by scalar loads the
// wave waited on lgkmcnt — the counter the LDS reads share — at every group: 5.5 us per 623-value iteration instead of 1.5)
static constexpr int MT_LAG_C[MT_NLAG] = {623, 850, 1077, 1246, 1304, 1531, 1700, 1758, 1869, 1985, 2096, 2154, 2212, 2439, 2492, 2608, 2666, 2777, 2893, 3004, 3062, 3115, 3120, 3342, 3347, 3400,
    3516, 3569, 3574, 3685, 3796, 3801, 3912, 3970, 4028, 4255, 4308, 4361, 4424, 4482, 4588, 4593, 4709, 4820, 4878, 4931, 4936, 4984, 5158, 5163, 5216, 5332, 5385, 5390, 5501, 5612, 5617,
    5728, 5786, 5844, 6071, 6124, 6177, 6240, 6298, 6404, 6409, 6525, 6636, 6694, 6747, 6752, 6800, 6974, 6979, 7032, 7148, 7201, 7206, 7264, 7317, 7428, 7433, 7544, 7602, 7660, 7940, 7993,
    8056, 8099, 8220, 8225, 8326, 8452, 8553, 8563, 8616, 8722, 8780, 8790, 8848, 9017, 9176, 9244, 9809, 9968, 10036, 10432, 11731, 11958, 12185, 12354, 12412, 12460, 12808, 13368, 13600,
    14276, 15184, 15575, 15802, 16029, 16256, 16483, 16710, 16937, 17164, 17444, 18067, 18294, 18352, 18521, 18748, 19937};
// sequential part: the first min(total, MT_HISTORY) draws of every request
__global__ void __launch_bounds__(256) k_mt_draws(const PermReq* __restrict__ reqs, int bootstrap) {
    __shared__ uint32_t mtA[624], mtB[624];
    const PermReq& R = reqs[blockIdx.x];
    if (R.cont || R.fy == 2 || R.cached) return;
    uint32_t* __restrict__ draws = R.P.draws;
    const long long seqMax = bootstrap && R.total >= MT_BOOT_MIN ? 19937LL : MT_HISTORY;     // bootstrap: only the first 19937 outputs come from here (see k_mt_classes)
    const long long total = R.total < seqMax ? R.total : seqMax;
    const int tid = threadIdx.x;
    uint32_t* cur = mtA; uint32_t* nxt = mtB;
    for (int i = tid; i < 624; i += 256) cur[i] = R.state[i];
    __syncthreads();
    int mti = (int)R.state[624];
    long long produced = 0;
    while (produced < total) {
        if (mti >= 624) {
            // three data-parallel phases into the second buffer: one barrier per phase
            auto mix = [&](uint32_t a, uint32_t b2, uint32_t src) { const uint32_t y = (a & 0x80000000u) | (b2 & 0x7fffffffu); return src ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u); };
            if (tid < 227) nxt[tid] = mix(cur[tid], cur[tid + 1], cur[tid + 397]);
            __syncthreads();
            if (tid < 227) nxt[227 + tid] = mix(cur[227 + tid], cur[228 + tid], nxt[tid]);
            __syncthreads();
            if (tid < 169) nxt[454 + tid] = mix(cur[454 + tid], cur[455 + tid], nxt[227 + tid]);
            __syncthreads();
            if (tid == 0) nxt[623] = mix(cur[623], nxt[0], nxt[396]);
            __syncthreads();
            uint32_t* t = cur; cur = nxt; nxt = t;
            mti = 0;
        }
        const long long left = total - produced;
        const int take = (int)(left < 624 - mti ? left : 624 - mti);
        for (int t = tid; t < take; t += 256) draws[produced + t] = mt_temper(cur[mti + t]);
        produced += take; mti += take;
    }
}
// data-parallel part, one launch for the whole batch.  With every lag a multiple of MT_STRIDE the stream falls apart into MT_STRIDE interleaved sequences (position mod
// MT_STRIDE) that never read each other: y_r[t] = out[r + MT_STRIDE t] obeys the ORIGINAL 134-term recurrence, y[t] = XOR_i y[t - MT_LAG[i]].  One workgroup owns one
// sequence and keeps its last 19937 values in a ring in LDS (80 KB), so the 134 operands of a value are LDS reads instead of 134 loads from a 10 MB window in the
// L2 / Infinity Cache; the smallest lag is 623, so 623 values are computed per barrier.  (k_mt_stride, one launch per 623 * MT_STRIDE outputs with the history in
// global memory, was 37 us per step — 1.2 TB/s of cache traffic for 0.3 MB of output — and 8 of the 14 ms of a 256-permutation batch of a 67 k-bin segment.)
#define MTC_T 640
#define MTC_BUF 40000             // words of LDS: the 19937-value history + 32 iterations of 623 new values; then the last 19937 move to the front
// stride / boot: the recurrence holds at every power-of-two stride, so the sequentially generated history itself is grown the same way — 19937 outputs from the sequential
// kernel, then stride 1 doubles them (one workgroup), stride 2 doubles again (two) ... stride 64 reaches the 19937 * 128 outputs the main launch (stride 128, boot = 0)
// starts from: seven short launches instead of 2.5 M outputs from one workgroup (4.3 ms per permutation loop of a long segment, 16 % of the somatic flow's kernel time).
__global__ void __launch_bounds__(MTC_T) k_mt_classes(const PermReq* __restrict__ reqs, int stride, int boot) {
    __shared__ uint32_t ring[MTC_BUF];
    const PermReq& R = reqs[blockIdx.y];
    // XCD-aware placement: workgroups go round-robin over the 8 XCDs, and the 4-byte stores of 16 neighbouring sequences make up one 64-byte line — with sequence = workgroup
    // index every line was written by 8 different L2s, 4 bytes at a time (PMC: 43 GB of WRITE_SIZE for 4.8 GB of draws).  Sequences r = x * (stride / 8) + k for the k-th
    // workgroup of XCD x: a line's 16 writers share an L2 and run side by side.
    const int bx = (int)blockIdx.x, r = stride >= 8 ? (bx & 7) * (stride >> 3) + (bx >> 3) : bx, tid = (int)threadIdx.x;
    if (R.fy == 2 || R.cached || (boot && (R.cont || R.total < MT_BOOT_MIN))) return;           // continued from the previous batch: the history is there already; short requests: generated sequentially
    const gptr<uint32_t> d = as_global(R.P.draws);               // (typed global: a flat store in the loop would tie every LDS wait to the store's completion, common.hpp)
    const long long start = boot ? 19937LL * stride : (R.cont ? 0 : MT_HISTORY);        // first position to generate; the 19937 * stride positions in front of it are there
    const long long end = boot ? (R.total < 19937LL * 2 * stride ? R.total : 19937LL * 2 * stride) : R.total;
    const long long left = end - start - r;
    if (left <= 0) return;
    const long long cnt = (left + stride - 1) / stride;              // values of this sequence to generate
    const gptr<const uint32_t> hist = (!boot && R.cont) ? as_global(R.hist) + r : (gptr<const uint32_t>)(d + (start - 19937LL * stride + r));
    for (int t = tid; t < 19937; t += MTC_T) ring[t] = hist[(long long)t * stride];
    __syncthreads();
    const gptr<uint32_t> out = d + (start + r);
    int head = 19937;                                                // slot of the next value; the buffer is linear, so every operand sits at a CONSTANT distance below the
                                                                     // value's own slot: 134 ds_read_b32 with immediate offsets, no address arithmetic, issued back to back
                                                                     // (with a ring and a wrap per operand the compiler waited for every single read: 4.5 us per iteration)
    for (long long u0 = 0; u0 < cnt; u0 += 623) {
        if (head + 623 > MTC_BUF) {                                  // out of room: the last 19937 values become the history at the front (through registers: the ranges overlap)
            uint32_t keep[(19937 + MTC_T - 1) / MTC_T];
#pragma unroll
            for (int k = 0; k < (19937 + MTC_T - 1) / MTC_T; k++) { const int t = tid + k * MTC_T; keep[k] = t < 19937 ? ring[head - 19937 + t] : 0u; }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < (19937 + MTC_T - 1) / MTC_T; k++) { const int t = tid + k * MTC_T; if (t < 19937) ring[t] = keep[k]; }
            head = 19937;
            __syncthreads();
        }
        const long long m = cnt - u0 < 623 ? cnt - u0 : 623;
        if (tid < m) {
            const uint32_t* __restrict__ w = ring + (head + tid - 19937);
            uint32_t v = 0;
#pragma unroll
            for (int i = 0; i < MT_NLAG; i++) v ^= w[19937 - MT_LAG_C[i]];
            ring[head + tid] = v;
            out[(u0 + tid) * stride] = v;
        }
        head += 623;
        __syncthreads();
    }
}
// generator state after every permutation of the batch, rebuilt from the outputs: the 624 words behind a position are the untempered
// last 624 outputs, with the read index at 624 (the next draw starts a new block) — what the host generator resumes from
__device__ __forceinline__ uint32_t mt_untemper(uint32_t y) {
    y ^= y >> 18;
    y ^= (y << 15) & 0xefc60000u;
    uint32_t t = y; t = y ^ ((t << 7) & 0x9d2c5680u); t = y ^ ((t << 7) & 0x9d2c5680u); t = y ^ ((t << 7) & 0x9d2c5680u); t = y ^ ((t << 7) & 0x9d2c5680u); y = t;
    t = y; t = y ^ (t >> 11); t = y ^ (t >> 11); y = t;
    return y;
}
__global__ void __launch_bounds__(256) k_mt_snapshots(const PermReq* __restrict__ reqs, int nreq) {
    int ri = 0;
    { int lo = 0, hi = nreq - 1; while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (reqs[mid].blockBase <= (int)blockIdx.x) lo = mid; else hi = mid - 1; } ri = lo; }
    const PermReq& R = reqs[ri];
    const int b = (int)blockIdx.x - R.blockBase;
    if (R.fy == 2 || R.cached) return;                     // (draws of short segments come from the host: perm_loop_small_gpu; cached draws: the host keeps a POSITION, not a state)
    const long long end = (long long)(b + 1) * R.n;       // (fewer than 624 outputs in front of it in a batch that continues nothing: the host advances the start state instead, perm_loop_gpu)
    uint32_t* s = R.snaps + (size_t)b * 625;
    for (int i = threadIdx.x; i < 624; i += 256) { const long long at = end - 624 + i; s[i] = mt_untemper(at >= 0 ? R.P.draws[at] : (R.cont ? R.hist[MT_HISTORY + at] : 0u)); }      // (at < 0: a continued batch of a short segment — the words lie in the previous batch; a batch that continues nothing: the snapshot is not used, see above)
    if (threadIdx.x == 0) s[624] = 624u;
}
// the intervals of a finished batch into its pinned mailbox: one workgroup per request behind the statistic kernels (instead of one device-to-host copy per request)
__global__ void __launch_bounds__(256) k_perm_mail(const PermReq* __restrict__ reqs) {
    const PermReq& R = reqs[blockIdx.x];
    if (!R.mailStat) return;
    for (int i = threadIdx.x; i < 2 * R.nb; i += 256) R.mailStat[i] = R.pstat[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) cvx_mail_publish(R.mailSeq, R.seq);
}
__device__ __forceinline__ int block_excl_scan_i32(int v, int* sh /*PG_T/64 + 1*/, int& total) {
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { int o = __shfl_up(inc, d); if ((threadIdx.x & 63) >= d) inc += o; }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) sh[w] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int i = 0; i < PG_T / 64; i++) { const int t = sh[i]; if (i < w) base += t; tot += t; }
    total = tot;
    __syncthreads();
    return base + inc - v;
}
// the statistic of one permuted sequence px[0, n) (shared by the two permutation kernels): prefix sums, arcs of length al0..hk and their complements, the interval
// Prefix sums and arcs go tile by tile through LDS (PT_TILE prefix values + a halo of the previous tile's last PT_HALO): the arcs of position a need sx[a + 2 .. a + hk], i.e.
// they are evaluated PT_HALO positions behind the prefix front, 24 LDS reads per position instead of 48 loads through the vector cache; the prefix array never goes to
// global memory — the complements (a < j) only need its first and last PT_HALO values, kept aside.
#define PT_TILE 2048
#define PT_HALO 32
#define PT_PER (PT_TILE / PG_T)
__device__ __forceinline__ void perm_stat_tail(const double* __restrict__ px, double* __restrict__ /*sx: not used any more*/, int n, int hk, int al0, double tss, double errBound, double* __restrict__ pstat, int b,
                                               double* shD /* [PG_T / 64 + 1] */, double (*shM)[PG_T / 64] /* [PG_MAXK + 1] */, double* sT /* [PT_TILE + PT_HALO] */, double* sEdge /* [2 * PT_HALO] */) {
    const int tid = threadIdx.x, w = tid >> 6;
    double m[PG_MAXK + 1];
#pragma unroll
    for (int j = 0; j <= PG_MAXK; j++) m[j] = 0.0;
    double dcarry = 0.0;
    double nx[PT_PER];                                          // the next tile's values are requested while this tile is worked on
#pragma unroll
    for (int r = 0; r < PT_PER; r++) { const int i = tid * PT_PER + r; nx[r] = i < n ? px[i] : 0.0; }
    for (int base = 0; base < n; base += PT_TILE) {
        // prefix sums of the tile (re-associated: PT_PER consecutive values per thread, wave scan of the thread totals, wave totals through LDS)
        double v[PT_PER]; double run = 0.0;
#pragma unroll
        for (int r = 0; r < PT_PER; r++) { run += nx[r]; v[r] = run; }
#pragma unroll
        for (int r = 0; r < PT_PER; r++) { const int i = base + PT_TILE + tid * PT_PER + r; nx[r] = i < n ? px[i] : 0.0; }
        double inc = run;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const double oo = __hiloint2double(__shfl_up(__double2hiint(inc), d), __shfl_up(__double2loint(inc), d)); if ((tid & 63) >= d) inc += oo; }
        if ((tid & 63) == 63) shD[w] = inc;
        __syncthreads();
        double wb = 0.0, tot = 0.0;
        for (int k = 0; k < PG_T / 64; k++) { const double t = shD[k]; if (k < w) wb += t; tot += t; }
        const double before = dcarry + wb + (inc - run);
#pragma unroll
        for (int r = 0; r < PT_PER; r++) sT[PT_HALO + tid * PT_PER + r] = before + v[r];
        dcarry += tot;
        __syncthreads();
        const int cnt = n - base < PT_TILE ? n - base : PT_TILE;          // prefix values of this tile: sT[PT_HALO, PT_HALO + cnt)
        if (base == 0 && tid < PT_HALO) sEdge[tid] = sT[PT_HALO + tid];   // sx[0 .. PT_HALO)  (n >= 2 PT_HALO: the device path starts at 1024 bins)
        if (base + PT_TILE >= n && tid < PT_HALO) sEdge[PT_HALO + tid] = sT[cnt + tid];      // sx[n - PT_HALO .. n)
        // arcs that start PT_HALO positions behind the front: LDS index u <-> position a = base - PT_HALO + u; the last tile also takes its own last PT_HALO positions
        const int uEnd = base + PT_TILE >= n ? cnt + PT_HALO : PT_TILE;
        for (int u = tid; u < uEnd; u += PG_T) {
            const int a = base - PT_HALO + u;
            if (a < 0) continue;
            const double s0 = sT[u];
#pragma unroll
            for (int j = 2; j <= PG_MAXK; j++)
                if (j >= al0 && j <= hk && a + j < n) { const double d = fabs(sT[u + j] - s0); m[j] = d > m[j] ? d : m[j]; }
        }
        __syncthreads();
        if (tid < PT_HALO) sT[tid] = sT[PT_TILE + tid];                   // the halo of the next tile
        __syncthreads();
    }
    // complements |sx[a + n - j] - sx[a]| for a < j <= hk <= PG_MAXK < PT_HALO
    if (tid < PT_HALO) {
        const int a = tid;
#pragma unroll
        for (int j = 2; j <= PG_MAXK; j++)
            if (j >= al0 && j <= hk && a < j) { const double d = fabs(sEdge[PT_HALO + (a + PT_HALO - j)] - sEdge[a]); m[j] = d > m[j] ? d : m[j]; }      // sx[a + n - j] = sEdge[PT_HALO + (a + n - j) - (n - PT_HALO)]
    }
#pragma unroll
    for (int j = 2; j <= PG_MAXK; j++) {
        double vv = m[j];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const double oo = __hiloint2double(__shfl_xor(__double2hiint(vv), d), __shfl_xor(__double2loint(vv), d)); vv = oo > vv ? oo : vv; }
        if ((tid & 63) == 0) shM[j][tid >> 6] = vv;
    }
    __syncthreads();
    if (tid == 0) {
        const double rn = (double)n;
        double hLo = 0.0, hHi = 0.0;
        for (int j = al0; j <= hk && j <= PG_MAXK; j++) {
            double vv = 0.0; for (int k = 0; k < PG_T / 64; k++) vv = shM[j][k] > vv ? shM[j][k] : vv;
            const double rj = (double)j, c = rn / (rj * (rn - rj));
            const double lo = vv - errBound > 0.0 ? vv - errBound : 0.0, hi = vv + errBound;
            const double a = c * (lo * lo) * (1.0 - 1e-15), bb = c * (hi * hi) * (1.0 + 1e-15);
            hLo = a > hLo ? a : hLo; hHi = bb > hHi ? bb : hHi;
        }
        auto norm = [&](double h) { double t = tss; if (t <= h + 0.0001) t = h + 1.0; return h / ((t - h) / (rn - 2.0)); };   // CBSTStatistic.cs:334-337
        if ((tss <= hLo + 0.0001) != (tss <= hHi + 0.0001)) { pstat[2 * b] = -INFINITY; pstat[2 * b + 1] = INFINITY; }          // the clamp is not monotone across its switch: let the host decide
        else { pstat[2 * b] = norm(hLo) * (1.0 - 1e-15); pstat[2 * b + 1] = norm(hHi) * (1.0 + 1e-15); }
    }
}
__global__ void __launch_bounds__(PG_T) k_perm_stat(const PermReq* __restrict__ reqs, int nreq) {
    __shared__ int shI[PG_T / 64 + 1];
    __shared__ double shD[PG_T / 64 + 1];
    __shared__ double shM[PG_MAXK + 1][PG_T / 64];
    __shared__ double sT[PT_TILE + PT_HALO], sEdge[2 * PT_HALO];
    int ri = 0;
    { int lo = 0, hi = nreq - 1; while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (reqs[mid].blockBase <= (int)blockIdx.x) lo = mid; else hi = mid - 1; } ri = lo; }
    const PermReq& R = reqs[ri];
    if (R.fy) return;
    const double* __restrict__ x = R.x; const int n = R.n, hk = R.hk, al0 = R.al0; const double tss = R.tss, errBound = R.errBound; const PermBuf P = R.P; double* __restrict__ pstat = R.pstat;
    const int b = (int)blockIdx.x - R.blockBase, tid = threadIdx.x;
    const size_t o = (size_t)b * n, o1 = (size_t)b * (n + 1);
    const uint32_t* __restrict__ draws = P.draws + o;
    int32_t* __restrict__ jj = P.j + o; int32_t* __restrict__ off = P.off + o1; int32_t* __restrict__ cur = P.cur + o1; int32_t* __restrict__ items = P.items + o;
    int32_t* __restrict__ g = P.g + o; int32_t* __restrict__ succ = P.succ + o; double* __restrict__ px = P.px + o; double* __restrict__ sx = P.sx + o;
    for (int i = tid; i <= n; i += PG_T) off[i] = 0;
    __syncthreads();
    // Fisher-Yates targets: draws are consumed in the order i = n-1, n-2, ..., 0 (ChangePoint.cs:411-419)
    for (int i = tid; i < n; i += PG_T) {
        const double cc = (double)draws[n - 1 - i] * (1.0 / 4294967296.0);
        int t = (int)(cc * (double)(i + 1)); t = t > i ? i : t;
        jj[i] = t; atomicAdd(&off[t], 1);
    }
    __syncthreads();
    // exclusive scan of the list sizes -> list offsets (off) and fill cursors (cur)
    int carry = 0;
    for (int base = 0; base <= n; base += PG_T) {
        const int i = base + tid; const int v = i <= n ? off[i] : 0; int tot;
        const int ex = block_excl_scan_i32(v, shI, tot);
        if (i <= n) { off[i] = carry + ex; cur[i] = carry + ex; }
        carry += tot;
    }
    __syncthreads();
    for (int i = tid; i < n; i += PG_T) { const int pos = atomicAdd(&cur[jj[i]], 1); items[pos] = i; }
    __syncthreads();
    // successor of every step inside its target's list; g(q) = first step > q that targets q (q itself when there is none)
    for (int i = tid; i < n; i += PG_T) {
        { const int q = jj[i]; int s = 0x7fffffff; for (int t = off[q]; t < off[q + 1]; t++) { const int v = items[t]; if (v > i && v < s) s = v; } succ[i] = s; }
        { int s = 0x7fffffff; for (int t = off[i]; t < off[i + 1]; t++) { const int v = items[t]; if (v > i && v < s) s = v; } g[i] = s == 0x7fffffff ? i : s; }
    }
    __syncthreads();
    // pointer doubling g <- g o g until stable (chains run towards larger indices and end in a fixed point); cur is the second buffer
    int32_t* ga = g; int32_t* gb = cur;
    for (int round = 0; round < 32; round++) {
        int changed = 0;
        for (int i = tid; i < n; i += PG_T) { const int a = ga[i]; const int t = ga[a]; gb[i] = t; changed |= (t != a); }
        int32_t* tmp = ga; ga = gb; gb = tmp;
        if (!__syncthreads_or(changed)) break;
    }
    // the permuted data
    for (int i = tid; i < n; i += PG_T) { const int s = succ[i]; px[i] = s == 0x7fffffff ? x[jj[i]] : x[ga[s]]; }
    __syncthreads();
    perm_stat_tail(px, sx, n, hk, al0, tss, errBound, pstat, b, shD, shM, sT, sEdge);
}

// ---- the same statistic with Fisher-Yates SIMULATED instead of resolved (k_perm_stat's counting sort + pointer doubling costs ~15 scattered 4-byte accesses per element into
// GBs of per-batch workspace: 64-byte HBM sectors moved for 4 bytes; this one costs one scattered read and one scattered write).  The swaps a[i] <-> a[t_i], i = n-1 .. 0,
// t_i <= i, are executed on an index array a (a[p] = p at the start) in BLOCKS of consecutive steps [I0, I1): the block's own positions sit in LDS; two steps of a block commute
// unless they share a position, and with targets uniform in [0, i] that is rare when the block is short against i (expected 1.5 Bk^2 / i of Bk steps; Bk = 8 sqrt(I1), <= 4096,
// keeps it near a hundred).  Steps that share nothing (the target lies below the block, no other step of the block has the same target — detected with two hashed bitmaps, false
// positives only move a step to the other class — and no step of the block targets the step's own position) run in parallel, one global read and one global write each.  The
// others are compacted in step order, the values under their targets are gathered into LDS (one slot per distinct position, found through a small hash map), ONE thread replays
// them in the reference's order on LDS, and the slots go back.  When a block has more than PF_CMAX such steps the permutation is given up: its statistic comes back as
// [-inf, inf] and the host evaluates it in the reference's order, like any other undecided one.
#define PF_BK 4096
#define PF_SPT (PF_BK / PG_T)
#define PF_HS (1 << 18)           // bits of the "some step of the block targets this position" table (hashed by the low bits of the position)
#define PF_HS2 (1 << 16)          // bits of the "two steps do" table: few are ever set, so a smaller table gives no more false positives
#define PF_CMAX 512
#define PF_MAP 1024               // (the kernel's LDS stays under half a CU's 160 KB: two workgroups per CU)
__global__ void __launch_bounds__(PG_T) k_perm_fy(const PermReq* __restrict__ reqs, int nreq) {
    __shared__ uint32_t sM1[PF_HS / 32], sM2[PF_HS2 / 32];
    __shared__ int32_t sA[PF_BK];
    __shared__ uint32_t sHit[PF_BK / 32];
    __shared__ int32_t sCI[PF_CMAX], sCT[PF_CMAX], sCV[PF_CMAX], sCanon[PF_CMAX];
    __shared__ uint32_t sMapKey[PF_MAP];
    __shared__ int32_t sMapVal[PF_MAP];
    __shared__ int shI[PG_T / 64 + 1];
    __shared__ double shD[PG_T / 64 + 1];
    __shared__ double shM[PG_MAXK + 1][PG_T / 64];
    __shared__ int sOver;
    __shared__ double sEdge[2 * PT_HALO];
    double* sT = reinterpret_cast<double*>(sM1);             // the statistic's tile buffer lives where the detection table was (PT_TILE + PT_HALO doubles < PF_HS / 8 bytes)
    static_assert((PT_TILE + PT_HALO) * 8 <= PF_HS / 8, "tile buffer does not fit the detection table");
    __shared__ int16_t sDep1[PF_CMAX], sDep2[PF_CMAX];
    __shared__ uint8_t sDone[PF_CMAX];
    int ri = 0;
    { int lo = 0, hi = nreq - 1; while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (reqs[mid].blockBase <= (int)blockIdx.x) lo = mid; else hi = mid - 1; } ri = lo; }
    const PermReq& R = reqs[ri];
    if (R.fy != 1) return;
    const double* __restrict__ x = R.x; const int n = R.n; const PermBuf P = R.P;
    const int b = (int)blockIdx.x - R.blockBase, tid = threadIdx.x;
    const size_t o = (size_t)b * n;
    const uint32_t* __restrict__ draws = P.draws + o;
    int32_t* __restrict__ a = P.j + o; double* __restrict__ px = P.px + o; double* __restrict__ sx = P.sx + o;
    for (int p = tid; p < n; p += PG_T) a[p] = p;
    if (tid == 0) sOver = 0;
    __syncthreads();
    int I1 = n;
    while (I1 > 0) {
        int Bk = (int)(8.0 * sqrt((double)I1)); Bk = Bk < 64 ? 64 : (Bk > PF_BK ? PF_BK : Bk);
        const int I0 = I1 - Bk > 0 ? I1 - Bk : 0; Bk = I1 - I0;
        // ---- the block's own positions, clean detection tables
        for (int p = tid; p < Bk; p += PG_T) sA[p] = a[I0 + p];
        { uint4* z1 = reinterpret_cast<uint4*>(sM1); uint4* z2 = reinterpret_cast<uint4*>(sM2); const uint4 z = make_uint4(0u, 0u, 0u, 0u);
          for (int w = tid; w < PF_HS / 128; w += PG_T) z1[w] = z;
          for (int w = tid; w < PF_HS2 / 128; w += PG_T) z2[w] = z; }
        for (int w = tid; w < PF_BK / 32; w += PG_T) sHit[w] = 0u;
        for (int w = tid; w < PF_MAP; w += PG_T) { sMapKey[w] = 0xFFFFFFFFu; sMapVal[w] = 0x7FFFFFFF; }
        __syncthreads();
        // ---- targets (ChangePoint.cs:411-419: the draw of step i is draws[n - 1 - i]); thread tid owns the steps I1 - 1 - (tid * PF_SPT + q): ascending (tid, q) = the reference's order
        int t[PF_SPT], pre[PF_SPT];                            // pre: what lies under a target below the block (valid for the steps that turn out to share nothing: the
                                                                //      previous block's writes are complete, and nobody else in this block touches that position)
#pragma unroll
        for (int q = 0; q < PF_SPT; q++) {
            const int k = tid * PF_SPT + q;
            t[q] = -1; pre[q] = 0;
            if (k < Bk) {
                const int i = I1 - 1 - k;
                const double cc = (double)draws[n - 1 - i] * (1.0 / 4294967296.0);
                int tt = (int)(cc * (double)(i + 1)); tt = tt > i ? i : tt;
                if (tt != i) {                                  // (a step that targets itself changes nothing)
                    t[q] = tt;
                    if (tt >= I0) atomicOr(&sHit[(tt - I0) >> 5], 1u << ((tt - I0) & 31));
                    else { pre[q] = a[tt]; const uint32_t h = (uint32_t)tt & (PF_HS - 1), bit = 1u << (h & 31); const uint32_t old = atomicOr(&sM1[h >> 5], bit); if (old & bit) { const uint32_t h2 = (uint32_t)tt & (PF_HS2 - 1); atomicOr(&sM2[h2 >> 5], 1u << (h2 & 31)); } }
                }
            }
        }
        __syncthreads();
        // ---- steps that share a position with another step of the block: compacted in step order; the others are executed here
        unsigned cm = 0; int cnt = 0;
#pragma unroll
        for (int q = 0; q < PF_SPT; q++) {
            if (t[q] < 0) continue;
            const int i = I1 - 1 - (tid * PF_SPT + q), tt = t[q];
            bool c = tt >= I0 || ((sHit[(i - I0) >> 5] >> ((i - I0) & 31)) & 1u);
            if (!c) { const uint32_t h2 = (uint32_t)tt & (PF_HS2 - 1); c = (sM2[h2 >> 5] >> (h2 & 31)) & 1u; }
            if (c) { cm |= 1u << q; cnt++; }
        }
        int nC;
        int base = block_excl_scan_i32(cnt, shI, nC);
        if (nC > PF_CMAX) { if (tid == 0) sOver = 1; __syncthreads(); break; }
#pragma unroll
        for (int q = 0; q < PF_SPT; q++) {
            if (t[q] < 0) continue;
            const int i = I1 - 1 - (tid * PF_SPT + q), tt = t[q];
            if ((cm >> q) & 1u) { sCI[base] = i; sCT[base] = tt; base++; }
            else { a[tt] = sA[i - I0]; sA[i - I0] = pre[q]; }
        }
        __syncthreads();
        // ---- one LDS slot per distinct target below the block: the first step (list order) that names it owns the slot
        for (int k = tid; k < nC; k += PG_T) {
            const int tt = sCT[k];
            if (tt < I0) {
                uint32_t slot = ((uint32_t)tt * 2654435761u) >> 22;
                for (;;) {
                    const uint32_t prev = atomicCAS(&sMapKey[slot], 0xFFFFFFFFu, (uint32_t)tt);
                    if (prev == 0xFFFFFFFFu || prev == (uint32_t)tt) { atomicMin(&sMapVal[slot], k); sCanon[k] = (int32_t)slot; break; }
                    slot = (slot + 1) & (PF_MAP - 1);
                }
            }
        }
        __syncthreads();
        for (int k = tid; k < nC; k += PG_T) {
            const int tt = sCT[k];
            if (tt < I0) { const int kc = sMapVal[sCanon[k]]; sCanon[k] = kc; if (kc == k) sCV[k] = a[tt]; }
        }
        __syncthreads();
        // ---- the replay, in the reference's order where the order matters: a step waits for the latest earlier step of the list that touches one of its two positions
        // (found by looking back through the list: ~a hundred entries), steps that wait for nothing swap in the same round.  Most entries are pairs of steps with a common
        // target: two or three rounds; the last blocks of a permutation (every target inside the block) degenerate into one step per round, on LDS.
        for (int k = tid >> 6; k < nC; k += PG_T / 64) {     // one wave per step, 64 earlier entries per look
            const int i = sCI[k], tt = sCT[k], lane = tid & 63;
            int d1 = -1, d2 = -1;                      // latest earlier step that touches position i / position tt
            for (int top = k - 1; top >= 0 && (d1 < 0 || d2 < 0); top -= 64) {
                const int e = top - lane;
                const int ie = e >= 0 ? sCI[e] : -1, te = e >= 0 ? sCT[e] : -2;
                const unsigned long long h1 = __ballot(te == i), h2 = __ballot(te == tt || ie == tt);      // (ie == i is impossible: one step per position)
                if (d1 < 0 && h1) d1 = top - (__ffsll((long long)h1) - 1);      // lane 0 holds the latest entry of the look
                if (d2 < 0 && h2) d2 = top - (__ffsll((long long)h2) - 1);
            }
            if (lane == 0) { sDep1[k] = (int16_t)d1; sDep2[k] = (int16_t)d2; sDone[k] = 0; }
        }
        __syncthreads();
        for (;;) {
            int left = 0;
            for (int k = tid; k < nC; k += PG_T) {
                if (sDone[k]) continue;
                const int d1 = sDep1[k], d2 = sDep2[k];
                if ((d1 < 0 || sDone[d1] == 1) && (d2 < 0 || sDone[d2] == 1)) {
                    const int i = sCI[k], tt = sCT[k];
                    const int vi = sA[i - I0];
                    int vt;
                    if (tt >= I0) { vt = sA[tt - I0]; sA[tt - I0] = vi; }
                    else { const int kc = sCanon[k]; vt = sCV[kc]; sCV[kc] = vi; }
                    sA[i - I0] = vt;
                    sDone[k] = 2;                      // done in this round: visible as 1 from the next round on
                } else left = 1;
            }
            const int more = __syncthreads_or(left);
            for (int k = tid; k < nC; k += PG_T) if (sDone[k] == 2) sDone[k] = 1;
            __syncthreads();
            if (!more) break;
        }
       
        for (int k = tid; k < nC; k += PG_T) { const int tt = sCT[k]; if (tt < I0 && sCanon[k] == k) a[tt] = sCV[k]; }
        // position i is final after step i: the permuted data of the block
        for (int p = tid; p < Bk; p += PG_T) px[I0 + p] = x[sA[p]];
        I1 = I0;
        __syncthreads();
    }
    if (sOver) { if (tid == 0) { R.pstat[2 * b] = -INFINITY; R.pstat[2 * b + 1] = INFINITY; } return; }
    __syncthreads();
    perm_stat_tail(px, sx, n, R.hk, R.al0, R.tss, R.errBound, R.pstat, b, shD, shM, sT, sEdge);
}

// ---- Fisher-Yates with every position in LDS (k_perm_rp; replaces k_perm_fy and, from 1024 bins on, k_perm_stat).  k_perm_fy keeps the index array of a permutation in global
// memory: every step reads and writes one scattered word of it, and a block of steps is a chain of dependent global round trips (read what lies under the targets, write, read
// the next block's own positions) — 21 G elements/s with the whole device busy, which made the permutation loops of a tumour / normal sample (10 G elements) device-bound.
// Here the positions are cut into RANGES of 16 384 that are processed from the top one down, each entirely in LDS:
//   * a step i swaps a[i] with a[t], t <= i.  If t lies in a LOWER range the step only needs the value under i (V) now; what lies under t is owed to position i by the range that
//     holds t.  The step appends the message (i, t, V) to that range's inbox and is done.  All steps of the ranges above a range precede its own steps in time, so the range first
//     applies its inbox (old = a[t]; a[t] = V; "position i receives old"), then runs its own steps; nothing ever flows upwards except those results.
//   * steps run in blocks that never straddle a multiple of 2048 (a "tile").  Two steps of a block commute unless they share a position; the targets inside the range are marked
//     in a bitmap of the range (exact, no hashing: the range is the bitmap's domain), and steps that share nothing run at once.  The others — and the messages of an inbox chunk
//     that name the same position — are PEELED in the reference's order: every position they touch gets a slot of a small hash table, every pending step posts its time stamp
//     (tagged with the round, so that nothing has to be reset) to the slots of its positions with atomicMax, and runs when it holds the maximum of all of them, i.e. when it is
//     the earliest pending step on each of its positions; steps that run in the same round share no position.  The last 512 steps of a permutation, where every step depends on
//     another, go through one wave 64 at a time (each lane waits for the latest earlier lane on its two positions).
//   * results ("position i holds value old") are appended to streams in tile order — the inbox results at the index of their message, the own results per tile — so the statistic
//     finds the values of a tile as a handful of contiguous segments: it scatters x[old] into an LDS tile and forms prefix sums and arcs as perm_stat_tail does.
// Per element: 4 bytes of draws, 8 + 8 bytes of message, 4 + 4 bytes of result, one gather from x (L2-resident) — streams instead of two scattered 128-byte line fills.  The
// permutation is the reference's (integer logic); the statistic is an interval as before.  An inbox that outgrows its (generous) capacity gives the permutation up: [-inf, inf],
// the host evaluates it in the reference's order — as does a block whose ordered steps touch more positions than the table has slots.  The phases are separate functions (not inlined) so that each gets its own register allocation: 128 VGPRs, two workgroups per CU.
#define RP_R 16384
#define RP_RSHIFT 14
#ifndef RP_T
#define RP_T 512                           // threads of a workgroup (two workgroups per CU by LDS: 128 VGPRs; -DRP_T=256 — four waves with 256 VGPRs, no spills — measured 27 against 34 G elements/s)
#endif
#define RP_SPT (2048 / RP_T)
#define RP_PER (PT_TILE / RP_T)            // consecutive positions of a thread in the statistic
#define RP_BK (RP_T * RP_SPT)             // 2048 = PT_TILE
#define RP_TPR (RP_R / RP_BK)
#define RP_MAXK 32                         // n <= 524 288
#define RP_MAXT (RP_MAXK * RP_TPR)
#define RP_HASH 1024                       // slots for the positions that several steps / messages of a block touch
#define RP_TAIL 512                        // the last steps: the wave routine alone
static_assert(RP_BK == PT_TILE, "a block of steps is a tile of the statistic");
// LDS of a workgroup (dynamic: the phase functions address it by constant offsets)
extern __shared__ __align__(16) unsigned char rp_lds[];
#define RPL_A 0                            // int32[RP_R]: the range's positions
#define RPL_TB (RP_R * 4)                  // uint32[512]: positions of the range that a step / message of the block targets
#define RPL_DB (RPL_TB + 2048)             // uint32[512]: ... that two do
#define RPL_HKEY (RPL_DB + 2048)           // uint32[RP_HASH]  (the fallback's lists sCI / sCT alias it)
#define RPL_HSTAMP (RPL_HKEY + RP_HASH * 4)
#define RPL_ROW (RPL_HSTAMP + RP_HASH * 4) // uint32[RP_MAXT + 1]: the inbox table's row of the range
#define RPL_CNT (RPL_ROW + (RP_MAXT + 2) * 4)
#define RPL_INOFF (RPL_CNT + RP_MAXK * 4)
#define RPL_INCAP (RPL_INOFF + RP_MAXK * 4)
#define RPL_MISC (RPL_INCAP + RP_MAXK * 4) // [0] own results so far, [1] give-up flag, [2..] scan scratch
#define RPL_TOTAL (RPL_MISC + 64)
static_assert(RPL_TOTAL <= 81920, "two workgroups per CU");
#define RP_LDS(T, off) (reinterpret_cast<T*>(rp_lds + (off)))
__device__ __forceinline__ int rp_target(gptr<const uint32_t> draws, int n, int i) {      // ChangePoint.cs:411-419: the draw of step i is draws[n - 1 - i]
    const double cc = (double)draws[n - 1 - i] * (1.0 / 4294967296.0);
    int tt = (int)(cc * (double)(i + 1)); return tt > i ? i : tt;
}
__device__ __forceinline__ bool rp_bit(const uint32_t* m, int p) { return (m[p >> 5] >> (p & 31)) & 1u; }
__device__ __forceinline__ uint32_t rp_slot(uint32_t* hKey, uint32_t key, int* over) {             // the hash slot of a position (inserted on first sight)
    uint32_t h = (key * 2654435761u) >> 22; int probes = 0;
    for (;;) { const uint32_t prev = atomicCAS(&hKey[h], 0xFFFFFFFFu, key); if (prev == 0xFFFFFFFFu || prev == key) return h; h = (h + 1) & (RP_HASH - 1); if (++probes > RP_HASH) { *over = 1; return h; } }
}
// exclusive scan of a small count over the wave + one LDS atomic for the wave's total: the first slot of this lane (all lanes arrive together)
__device__ __forceinline__ uint32_t rp_wave_reserve(uint32_t* cnt, int c) {
    const int lane = (int)(threadIdx.x & 63);
    const int inc = (int)wave_inclusive_scan_u32((uint32_t)c);      // (DPP: no LDS round trips)
    const int total = __builtin_amdgcn_readlane(inc, 63);
    uint32_t base = 0u;
    if (total) { if (lane == 63) base = atomicAdd(cnt, (uint32_t)total); base = __builtin_amdgcn_readlane(base, 63); }
    return base + (uint32_t)(inc - c);
}
// inclusive sum over the wave of a double, the lanes' values added in ascending lane order within rows and rows in ascending order (DPP moves: no LDS round trips; lanes that a
// shift does not reach receive +0.0)
template <int CTRL, int ROWMASK> __device__ __forceinline__ double rp_dpp_f64(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROWMASK, 0xF, true), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROWMASK, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rp_wave_scan_f64(double v) {
    v += rp_dpp_f64<0x111, 0xF>(v);      // row_shr:1
    v += rp_dpp_f64<0x112, 0xF>(v);      // row_shr:2
    v += rp_dpp_f64<0x114, 0xF>(v);      // row_shr:4
    v += rp_dpp_f64<0x118, 0xF>(v);      // row_shr:8
    v += rp_dpp_f64<0x142, 0xA>(v);      // row_bcast:15 into rows 1 and 3
    v += rp_dpp_f64<0x143, 0xC>(v);      // row_bcast:31 into rows 2 and 3
    return v;
}
// (typed global pointers: through generic ones every access is a flat instruction, and every LDS wait of these LDS-heavy phases would wait for the stores as well — common.hpp)
struct RpPtr { gptr<const uint32_t> draws; gptr<const double> x; gptr<uint32_t> scr; gptr<uint32_t> endsIn; gptr<uint32_t> endsOwn; gptr<uint2> inbox; gptr<uint32_t> outIn; gptr<uint32_t> outOwn; int n, K, nT; };
__device__ __forceinline__ RpPtr rp_ptr(const PermReq& Rq, int w, int b) {
    const gptr<const PermReq> R = as_global(&Rq);
    RpPtr P; P.n = R->n; P.K = R->rp.K; P.nT = R->rp.nT; P.draws = as_global((const uint32_t*)R->P.draws) + (size_t)b * R->n; P.x = as_global(R->x);
    P.scr = as_global(R->rpScratch) + (size_t)w * R->rp.stride; P.endsIn = P.scr + R->rp.oEndsIn; P.endsOwn = P.scr + R->rp.oEndsOwn; P.inbox = reinterpret_cast<gptr<uint2>>(P.scr + R->rp.oInbox);
    P.outIn = P.scr + R->rp.oOutIn; P.outOwn = P.scr + R->rp.oOutOwn;
    return P;
}
// one step, by itself (ordered replay / the last steps)
__device__ __forceinline__ void rp_exec1(const RpPtr& P, int lo, int i, int tt) {
    int32_t* sA = RP_LDS(int32_t, RPL_A); uint32_t* sCnt = RP_LDS(uint32_t, RPL_CNT); uint32_t* sMisc = RP_LDS(uint32_t, RPL_MISC);
    const int v = sA[i - lo];
    if (tt < lo) {
        const int d = tt >> RP_RSHIFT; const uint32_t sidx = atomicAdd(&sCnt[d], 1u);
        if (sidx < RP_LDS(uint32_t, RPL_INCAP)[d]) gstore_uint2(P.inbox + (RP_LDS(uint32_t, RPL_INOFF)[d] + sidx), make_uint2((uint32_t)v | ((uint32_t)(i & 2047) << 20), (uint32_t)(tt & 16383) | ((uint32_t)(i >> 11) << 14)));
        else sMisc[1] = 1u;
    } else {
        int old = v; if (tt != i) { old = sA[tt - lo]; sA[tt - lo] = v; }
        const uint32_t sidx = atomicAdd(&sMisc[0], 1u); P.outOwn[sidx] = (uint32_t)old | ((uint32_t)(i & 2047) << 20);
    }
}
// The last RP_TAIL steps of a permutation, 64 at a time in the reference's order on one wave (lane = step, smaller lane = earlier; every position < RP_TAIL): a lane runs when
// the latest earlier lane on each of its two positions has run.  Those lanes are found in a table of 64-bit lane masks per position (one ds_or + two reads per lane); the table
// (RP_TAIL x 8 bytes over the hash keys) is all zero before and after
__device__ __noinline__ void rp_tail_ordered(const RpPtr& P, int i, int tt, bool active) {
    const int lane = (int)(threadIdx.x & 63);
    unsigned long long* mask = RP_LDS(unsigned long long, RPL_HKEY);
    const bool isB = active && tt != i;
    if (isB) atomicOr(&mask[tt], 1ull << lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned long long m1 = active ? mask[i] & below : 0ull, m2 = isB ? mask[tt] & below : 0ull;
    const int d1 = m1 ? 63 - __clzll((long long)m1) : -1, d2 = m2 ? 63 - __clzll((long long)m2) : -1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (isB) mask[tt] = 0ull;
    unsigned long long done = ~__ballot(active);
    bool mine = !active;
    while (~done) {
        const bool ready = !mine && (d1 < 0 || ((done >> d1) & 1ull)) && (d2 < 0 || ((done >> d2) & 1ull));
        if (ready) { rp_exec1(P, 0, i, tt); mine = true; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        done |= __ballot(ready);
    }
}
// ---- the inbox of range k: everything the ranges above owe to / want from this range, in chunks of whole producer tiles (inside a chunk the time stamps order the messages)
__device__ __noinline__ void rp_range_inbox(const PermReq& R, int w, int b, int k) {
    const RpPtr P = rp_ptr(R, w, b);
    const int tid = threadIdx.x, nT = P.nT;
    int32_t* sA = RP_LDS(int32_t, RPL_A); uint32_t* sTb = RP_LDS(uint32_t, RPL_TB); uint32_t* sDb = RP_LDS(uint32_t, RPL_DB);
    uint32_t* hKey = RP_LDS(uint32_t, RPL_HKEY); uint32_t* hStamp = RP_LDS(uint32_t, RPL_HSTAMP); uint32_t* sRow = RP_LDS(uint32_t, RPL_ROW); int* sOver = RP_LDS(int, RPL_MISC) + 1;
    const uint32_t inOff = RP_LDS(uint32_t, RPL_INOFF)[k];
    const gptr<uint2> ib = P.inbox + inOff; const gptr<uint32_t> ob = P.outIn + inOff;
    const int tauLo = (k + 1) * RP_TPR;
    int tau = nT - 1; uint32_t c0 = 0u;
    // the chunk [c0, c1) = tiles tau .. t2 (uniform: every thread walks the row in LDS)
    auto next_chunk = [&](int tauIn, uint32_t cIn, int& t2o, uint32_t& c1o) { int t2 = tauIn; uint32_t c1 = sRow[t2]; while (t2 - 1 >= tauLo && sRow[t2 - 1] - cIn <= (uint32_t)RP_BK) { t2--; c1 = sRow[t2]; } t2o = t2; c1o = c1; };
    int t2 = tau; uint32_t c1 = 0u;
    uint2 m[RP_SPT]; bool valid[RP_SPT];
#pragma unroll
    for (int q = 0; q < RP_SPT; q++) { valid[q] = false; m[q] = make_uint2(0u, 0u); }
    if (tau >= tauLo) {
        next_chunk(tau, c0, t2, c1);
#pragma unroll
        for (int q = 0; q < RP_SPT; q++) { const uint32_t j = c0 + (uint32_t)(q * RP_T + tid); valid[q] = j < c1; if (valid[q]) m[q] = gload_uint2(ib + j); }
    }
    while (tau >= tauLo) {
        // the messages of this chunk are in registers; those of the next one are requested before this one is worked on
        const uint32_t cc0 = c0; const int tauN = t2 - 1; const uint32_t cN0 = c1;
        int t2N = tauN; uint32_t cN1 = cN0; uint2 mN[RP_SPT]; bool validN[RP_SPT];
#pragma unroll
        for (int q = 0; q < RP_SPT; q++) { validN[q] = false; mN[q] = make_uint2(0u, 0u); }
        if (tauN >= tauLo) {
            next_chunk(tauN, cN0, t2N, cN1);
#pragma unroll
            for (int q = 0; q < RP_SPT; q++) { const uint32_t j = cN0 + (uint32_t)(q * RP_T + tid); validN[q] = j < cN1; if (validN[q]) mN[q] = gload_uint2(ib + j); }
        }
        if (c1 > cc0) {
#pragma unroll
            for (int q = 0; q < RP_SPT; q++) if (valid[q]) { const int tl = (int)(m[q].y & 16383u); const uint32_t bt = 1u << (tl & 31); const uint32_t old = atomicOr(&sTb[tl >> 5], bt); if (old & bt) atomicOr(&sDb[tl >> 5], bt); }
            __syncthreads();
            unsigned pend = 0u; int slotq[RP_SPT];
#pragma unroll
            for (int q = 0; q < RP_SPT; q++) {
                slotq[q] = 0;
                if (valid[q]) {
                    const uint32_t j = cc0 + (uint32_t)(q * RP_T + tid); const int tl = (int)(m[q].y & 16383u); const uint32_t v = m[q].x & 0xFFFFFu, io = m[q].x >> 20;
                    if (!rp_bit(sDb, tl)) { const int old = sA[tl]; sA[tl] = (int)v; ob[j] = (uint32_t)old | (io << 20); }
                    else { slotq[q] = (int)rp_slot(hKey, (uint32_t)tl, sOver); pend |= 1u << q; }      // several messages name this position
                }
            }
            int more = __syncthreads_or((int)pend);
            const int dirtyChunk = more;
            for (uint32_t round = 1; more; round++) {
                // the earliest pending message of every position (largest i) applies; a later round's stamps outrank every earlier round's
#pragma unroll
                for (int q = 0; q < RP_SPT; q++) if ((pend >> q) & 1u) { const uint32_t iE = ((m[q].y >> 14) << 11) | (m[q].x >> 20); atomicMax(&hStamp[slotq[q]], (round << 20) | (iE + 1u)); }
                __syncthreads();
#pragma unroll
                for (int q = 0; q < RP_SPT; q++) if ((pend >> q) & 1u) {
                    const uint32_t iE = ((m[q].y >> 14) << 11) | (m[q].x >> 20);
                    if (hStamp[slotq[q]] == ((round << 20) | (iE + 1u))) {
                        const uint32_t j = cc0 + (uint32_t)(q * RP_T + tid); const int tl = (int)(m[q].y & 16383u);
                        const int old = sA[tl]; sA[tl] = (int)(m[q].x & 0xFFFFFu); ob[j] = (uint32_t)old | ((m[q].x >> 20) << 20);
                        pend &= ~(1u << q);
                    }
                }
                more = __syncthreads_or((int)pend);
                if (round > 4000u) { if (tid == 0) *sOver = 1; break; }
            }
            for (int z = tid; z < 512; z += RP_T) { sTb[z] = 0u; sDb[z] = 0u; }
            if (dirtyChunk) { for (int z = tid; z < RP_HASH; z += RP_T) { hKey[z] = 0xFFFFFFFFu; hStamp[z] = 0u; } }
            __syncthreads();
        }
        c0 = c1; tau = tauN; t2 = t2N; c1 = cN1;
#pragma unroll
        for (int q = 0; q < RP_SPT; q++) { m[q] = mN[q]; valid[q] = validN[q]; }
    }
}
// ---- the own steps of range k, hi - 1 down to lo (range 0: down to RP_TAIL, then the last steps through one wave)
__device__ __noinline__ void rp_range_own(const PermReq& R, int w, int b, int k, gptr<long long> clk) {
    const RpPtr P = rp_ptr(R, w, b);
    const int tid = threadIdx.x, lane = tid & 63, n = P.n, nT = P.nT;
    const int lo = k << RP_RSHIFT, hi = n < lo + RP_R ? n : lo + RP_R;
    int32_t* sA = RP_LDS(int32_t, RPL_A); uint32_t* sTb = RP_LDS(uint32_t, RPL_TB); uint32_t* sDb = RP_LDS(uint32_t, RPL_DB);
    uint32_t* hKey = RP_LDS(uint32_t, RPL_HKEY); uint32_t* hStamp = RP_LDS(uint32_t, RPL_HSTAMP);
    uint32_t* sCnt = RP_LDS(uint32_t, RPL_CNT); uint32_t* sInOff = RP_LDS(uint32_t, RPL_INOFF); uint32_t* sInCap = RP_LDS(uint32_t, RPL_INCAP);
    uint32_t* sMisc = RP_LDS(uint32_t, RPL_MISC); int* sOver = RP_LDS(int, RPL_MISC) + 1;
    long long tClk = clk ? clock64() : 0;
    auto lapc = [&](int slot) { if (clk) { const long long t = clock64(); clk[slot] += t - tClk; tClk = t; } };
    const int stop = k == 0 ? (hi < RP_TAIL ? hi : RP_TAIL) : lo;
    auto geometry = [&](int I1, int& I0o) { const int tileBase = ((I1 - 1) >> 11) << 11; int I0 = tileBase > stop ? tileBase : stop;
                                              if (k == 0) { int bk = (int)(8.0 * sqrt((double)I1)); bk = bk < 64 ? 64 : (bk > RP_BK ? RP_BK : bk); if (I1 - bk > I0) I0 = I1 - bk; } I0o = I0; };
    int I1 = hi, I0 = hi;
    int t[RP_SPT];
#pragma unroll
    for (int q = 0; q < RP_SPT; q++) t[q] = -1;
    if (I1 > stop) {
        geometry(I1, I0);
#pragma unroll
        for (int q = 0; q < RP_SPT; q++) { const int kk = tid * RP_SPT + q; if (kk < I1 - I0) t[q] = rp_target(P.draws, n, I1 - 1 - kk); }
    }
    while (I1 > stop) {
        const int Bk = I1 - I0, tileBase = ((I1 - 1) >> 11) << 11, tau = tileBase >> 11;
#pragma unroll
        for (int q = 0; q < RP_SPT; q++) {
            const int kk = tid * RP_SPT + q;
            if (kk < Bk) { const int i = I1 - 1 - kk, tt = t[q];
                if (tt >= lo && tt != i) { const int tl = tt - lo; const uint32_t bt = 1u << (tl & 31); const uint32_t old = atomicOr(&sTb[tl >> 5], bt); if (old & bt) atomicOr(&sDb[tl >> 5], bt); } }
        }
        // the next block's targets are requested now: their loads fly while this block is worked on
        const int I1n = I0; int I0n = I0;
        int tn[RP_SPT];
#pragma unroll
        for (int q = 0; q < RP_SPT; q++) tn[q] = -1;
        if (I1n > stop) {
            geometry(I1n, I0n);
#pragma unroll
            for (int q = 0; q < RP_SPT; q++) { const int kk = tid * RP_SPT + q; if (kk < I1n - I0n) tn[q] = rp_target(P.draws, n, I1n - 1 - kk); }
        }
        lapc(8);
        __syncthreads();
        lapc(9);
        // independent steps run here; their results and messages are appended with one reservation per wave (and destination)
        unsigned dm = 0u, ownm = 0u, msgm = 0u; int vv[RP_SPT], ov[RP_SPT];
#pragma unroll
        for (int q = 0; q < RP_SPT; q++) {
            const int kk = tid * RP_SPT + q; const bool in = kk < Bk;
            const int i = I1 - 1 - kk, tt = t[q];
            const bool dirty = in && (rp_bit(sTb, i - lo) || (tt >= lo && tt != i && rp_bit(sDb, tt - lo)));
            vv[q] = 0; ov[q] = 0;
            if (in && !dirty) { const int v = sA[i - lo]; vv[q] = v; ov[q] = v; if (tt < lo) msgm |= 1u << q; else { ownm |= 1u << q; if (tt != i) { ov[q] = sA[tt - lo]; sA[tt - lo] = v; } } }
            if (dirty) dm |= 1u << q;
        }
        lapc(10);
        {
            uint32_t so = rp_wave_reserve(&sMisc[0], __popc(ownm));
#pragma unroll
            for (int q = 0; q < RP_SPT; q++) if ((ownm >> q) & 1u) { const int i = I1 - 1 - (tid * RP_SPT + q); P.outOwn[so++] = (uint32_t)ov[q] | ((uint32_t)(i & 2047) << 20); }
        }
        for (int d = 0; d < k; d++) {
            int c = 0;
#pragma unroll
            for (int q = 0; q < RP_SPT; q++) if (((msgm >> q) & 1u) && (t[q] >> RP_RSHIFT) == d) c++;
            if (!__ballot(c > 0)) continue;
            uint32_t sm = rp_wave_reserve(&sCnt[d], c);
            const uint32_t cap = sInCap[d], off = sInOff[d];
#pragma unroll
            for (int q = 0; q < RP_SPT; q++) if (((msgm >> q) & 1u) && (t[q] >> RP_RSHIFT) == d) {
                const int i = I1 - 1 - (tid * RP_SPT + q);
                if (sm < cap) gstore_uint2(P.inbox + (off + sm), make_uint2((uint32_t)vv[q] | ((uint32_t)(i & 2047) << 20), (uint32_t)(t[q] & 16383) | ((uint32_t)tau << 14))); else *sOver = 1;
                sm++;
            }
        }
        lapc(11);
        // ---- peel the ordered steps: a slot per position they touch; the earliest pending step on all its positions runs (a later round's stamps outrank every earlier
        // round's).  The first round's stamps are posted right here, so a block without ordered steps costs one barrier more and a block with them two per round.
        int sI[RP_SPT], sT2[RP_SPT];
#pragma unroll
        for (int q = 0; q < RP_SPT; q++) { sI[q] = 0; sT2[q] = -1;
            if ((dm >> q) & 1u) { const int i = I1 - 1 - (tid * RP_SPT + q), tt = t[q]; sI[q] = (int)rp_slot(hKey, (uint32_t)(i - lo), sOver); if (tt >= lo && tt != i) sT2[q] = (int)rp_slot(hKey, (uint32_t)(tt - lo), sOver); } }
        lapc(12);
        unsigned pend = dm;
        int more = __syncthreads_or((int)dm);                      // (also: every independent step has run, nobody reads the bitmaps any more)
        const int anyOrdered = more;
        for (int z = tid; z < 512; z += RP_T) { sTb[z] = 0u; sDb[z] = 0u; }
        lapc(2);
        for (uint32_t round = 1; more; round++) {
#pragma unroll
            for (int q = 0; q < RP_SPT; q++) if ((pend >> q) & 1u) { const uint32_t key = (round << 20) | (uint32_t)(I1 - (tid * RP_SPT + q)); atomicMax(&hStamp[sI[q]], key); if (sT2[q] >= 0) atomicMax(&hStamp[sT2[q]], key); }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < RP_SPT; q++) if ((pend >> q) & 1u) {
                const uint32_t key = (round << 20) | (uint32_t)(I1 - (tid * RP_SPT + q));
                if (hStamp[sI[q]] == key && (sT2[q] < 0 || hStamp[sT2[q]] == key)) { rp_exec1(P, lo, I1 - 1 - (tid * RP_SPT + q), t[q]); pend &= ~(1u << q); }
            }
            more = __syncthreads_or((int)pend);
            if (round > 4000u) { if (tid == 0) *sOver = 1; break; }
        }
        if (anyOrdered) { for (int z = tid; z < RP_HASH; z += RP_T) { hKey[z] = 0xFFFFFFFFu; hStamp[z] = 0u; } }
        else __syncthreads();                                      // (the cleared bitmaps in front of the next block's marks; with ordered steps the rounds' barriers stand there)
        lapc(3);
        // (every step of the block has run behind the last barrier; the next block touches the counters only behind its own first barrier)
        if (I0 == tileBase) { if (tid < k) P.endsIn[(size_t)tid * (nT + 1) + tau] = sCnt[tid]; if (tid == 0) P.endsOwn[tau] = sMisc[0]; }
        I1 = I1n; I0 = I0n;
#pragma unroll
        for (int q = 0; q < RP_SPT; q++) t[q] = tn[q];
    }
    __syncthreads();
    if (k == 0) {
        static_assert(RP_TAIL * 8 <= RP_HASH * 4, "the lane-mask table of the last steps lies over the hash keys");
        for (int z = tid; z < RP_TAIL; z += RP_T) RP_LDS(unsigned long long, RPL_HKEY)[z] = 0ull;
        __syncthreads();
        if (tid < 64) for (int base = I1; base > 0; base -= 64) { const int i = base - 1 - lane; const bool act = i >= 0; rp_tail_ordered(P, act ? i : -1, act ? rp_target(P.draws, n, i) : -1, act); }
        __syncthreads();
        if (tid == 0) P.endsOwn[0] = sMisc[0];
        lapc(4);
    }
}
// ---- the statistic (as perm_stat_tail, for FindChangePoints' own parameters: arcs of 2 .. 25 bins).  The values of a tile are gathered from the result streams into LDS
// instead of read from a px array: the entries of tile t + 2 and the values of tile t + 1 are in flight while tile t is worked on.
#define RP_J0 2
#define RP_J1 25
#define RP_NJ (RP_J1 - RP_J0 + 1)
__device__ __noinline__ void rp_statistic(const PermReq& R, int w, int b, gptr<long long> clk) {
    const RpPtr P = rp_ptr(R, w, b);
    const int tid = threadIdx.x, n = P.n, K = P.K, nT = P.nT;
    long long tClk = clk ? clock64() : 0;
    auto lapc = [&](int slot) { if (clk) { const long long t = clock64(); clk[slot] += t - tClk; tClk = t; } };
    double* sPx = RP_LDS(double, 0); double* sT = sPx + PT_TILE;                                     // [PT_TILE], [PT_TILE + PT_HALO]
    uint32_t* sTab = reinterpret_cast<uint32_t*>(sT + PT_TILE + PT_HALO);                            // endsOwn[nT + 1], then endsIn[K - 1][nT + 1]
    double (*shM)[RP_T / 64] = reinterpret_cast<double (*)[RP_T / 64]>(sTab + (((size_t)K * (nT + 1) + 3) & ~size_t(1)));      // (an even number of words in front: 8-byte aligned)
    double* shD = reinterpret_cast<double*>(shM + RP_NJ); double* sEdge = shD + (RP_T / 64 + 1);
    int* sSrcOff = reinterpret_cast<int*>(sEdge + 2 * PT_HALO); uint32_t* sSrcBase = reinterpret_cast<uint32_t*>(sSrcOff + 4 * (RP_MAXK + 2));      // four tables each (tile & 3)
    const uint32_t* sInOff = RP_LDS(uint32_t, RPL_INOFF);
    const gptr<const PermReq> Rg = as_global(&R);                      // (the request table lies in global memory: read it through a typed pointer, once)
    const double tss = Rg->tss, errBound = Rg->errBound; const uint32_t oOutOwn = Rg->rp.oOutOwn, oOutIn = Rg->rp.oOutIn; const gptr<double> pstat = as_global(Rg->pstat);
    for (int t = tid; t <= nT; t += RP_T) sTab[t] = P.endsOwn[t];
    for (int d = 0; d < K - 1; d++) for (int t = tid; t <= nT; t += RP_T) sTab[(size_t)(d + 1) * (nT + 1) + t] = P.endsIn[(size_t)d * (nT + 1) + t];
    __syncthreads();
    // sources of a tile: the own stream, and the inbox results of every lower range
    auto fill_sources = [&](int tau) {
        int* so = sSrcOff + (tau & 3) * (RP_MAXK + 2); uint32_t* sb = sSrcBase + (tau & 3) * (RP_MAXK + 2);
        const int S = (tau >> 3) + 1; int run = 0;
        for (int sI = 0; sI < S; sI++) {
            const uint32_t beg = sTab[(size_t)sI * (nT + 1) + tau + 1], end = sTab[(size_t)sI * (nT + 1) + tau];
            so[sI] = run; sb[sI] = (sI == 0 ? oOutOwn : oOutIn + sInOff[sI - 1]) + beg; run += (int)(end - beg);
        }
        so[S] = run;
    };
    auto load_entries = [&](int tau, uint32_t* ent) -> int {      // the 4 entries of this thread for the tile; 0xFFFFFFFF = none; returns 1 when the tile's sources do not add up
        const int base = tau * PT_TILE, cnt = n - base < PT_TILE ? n - base : PT_TILE, S = (tau >> 3) + 1;
        const int* so = sSrcOff + (tau & 3) * (RP_MAXK + 2); const uint32_t* sb = sSrcBase + (tau & 3) * (RP_MAXK + 2);
        const bool okT = so[S] == cnt;
#pragma unroll
        for (int q = 0; q < RP_SPT; q++) {
            const int e = q * RP_T + tid; ent[q] = 0xFFFFFFFFu;
            if (okT && e < cnt) { int sI = 0; while (sI + 1 < S && so[sI + 1] <= e) sI++; ent[q] = P.scr[sb[sI] + (uint32_t)(e - so[sI])]; }
        }
        return okT ? 0 : 1;
    };
    if (tid == 0) { fill_sources(0); if (nT > 1) fill_sources(1); if (nT > 2) fill_sources(2); }
    __syncthreads();
    uint32_t ent[RP_SPT]; int io[RP_SPT]; double xv[RP_SPT];
    int bad = load_entries(0, ent);
#pragma unroll
    for (int q = 0; q < RP_SPT; q++) { io[q] = ent[q] == 0xFFFFFFFFu ? -1 : (int)(ent[q] >> 20); xv[q] = io[q] >= 0 ? P.x[ent[q] & 0xFFFFFu] : 0.0; }
    if (nT > 1) bad |= load_entries(1, ent);
    else {
#pragma unroll
        for (int q = 0; q < RP_SPT; q++) ent[q] = 0xFFFFFFFFu;
    }
    const int wv = tid >> 6;
    double mx[RP_NJ];
#pragma unroll
    for (int j = 0; j < RP_NJ; j++) mx[j] = 0.0;
    double dcarry = 0.0;
    for (int base = 0, tau = 0; base < n; base += PT_TILE, tau++) {
        const int cnt = n - base < PT_TILE ? n - base : PT_TILE;
#pragma unroll
        for (int q = 0; q < RP_SPT; q++) if (io[q] >= 0) sPx[io[q]] = xv[q];
        __syncthreads();
        lapc(6);
        // values of the next tile, entries of the one after it
#pragma unroll
        for (int q = 0; q < RP_SPT; q++) { io[q] = ent[q] == 0xFFFFFFFFu ? -1 : (int)(ent[q] >> 20); xv[q] = io[q] >= 0 ? P.x[ent[q] & 0xFFFFFu] : 0.0; }
        if (tau + 2 < nT) bad |= load_entries(tau + 2, ent);
        else {
#pragma unroll
            for (int q = 0; q < RP_SPT; q++) ent[q] = 0xFFFFFFFFu;
        }
        double v[RP_PER]; double run = 0.0;
#pragma unroll
        for (int r = 0; r < RP_PER; r++) { const int i = tid * RP_PER + r; run += i < cnt ? sPx[i] : 0.0; v[r] = run; }
        const double inc = rp_wave_scan_f64(run);
        if ((tid & 63) == 63) shD[wv] = inc;
        __syncthreads();
        if (tid == 0 && tau + 3 < nT) fill_sources(tau + 3);      // (slot (tau + 3) & 3 held tile tau - 1: last read two iterations ago)
        double wb = 0.0, tot = 0.0;
        for (int kq = 0; kq < RP_T / 64; kq++) { const double tq = shD[kq]; if (kq < wv) wb += tq; tot += tq; }
        const double before = dcarry + wb + (inc - run);
#pragma unroll
        for (int r = 0; r < RP_PER; r++) sT[PT_HALO + tid * RP_PER + r] = before + v[r];
        dcarry += tot;
        __syncthreads();
        if (base == 0 && tid < PT_HALO) sEdge[tid] = sT[PT_HALO + tid];
        if (base + PT_TILE >= n && tid < PT_HALO) sEdge[PT_HALO + tid] = sT[cnt + tid];
        if (base > 0 && base + PT_TILE < n) {
            // (fmax: one v_max_f64 with the |.| modifier per arc — the ternary compiled to a compare, two selects and an and; all operands are finite)
            // positions u = 4 tid .. 4 tid + 3: arcs of 2 .. 13 from sT[4 tid .. 4 tid + 16], arcs of 14 .. 25 from sT[4 tid + 14 .. 4 tid + 28] — two windows in registers, 96 arcs from 36 reads
            const double* wp = sT + tid * RP_PER;
            double w0[RP_PER];
#pragma unroll
            for (int r = 0; r < RP_PER; r++) w0[r] = wp[r];
            {
                double wa[RP_PER + 11];      // wp[2 .. RP_PER + 12]
#pragma unroll
                for (int e = 0; e < RP_PER + 11; e++) wa[e] = wp[2 + e];
#pragma unroll
                for (int r = 0; r < RP_PER; r++) {
#pragma unroll
                    for (int j = 2; j <= 13; j++) mx[j - RP_J0] = fmax(mx[j - RP_J0], fabs(wa[r + j - 2] - w0[r]));
                }
            }
            {
                double wb2[RP_PER + 11];     // wp[14 .. RP_PER + 24]
#pragma unroll
                for (int e = 0; e < RP_PER + 11; e++) wb2[e] = wp[14 + e];
#pragma unroll
                for (int r = 0; r < RP_PER; r++) {
#pragma unroll
                    for (int j = 14; j <= 25; j++) mx[j - RP_J0] = fmax(mx[j - RP_J0], fabs(wb2[r + j - 14] - w0[r]));
                }
            }
        } else {
            const int uEnd = base + PT_TILE >= n ? cnt + PT_HALO : PT_TILE;
            for (int u = tid; u < uEnd; u += RP_T) {
                const int a = base - PT_HALO + u;
                if (a < 0) continue;
                const double s0 = sT[u];
#pragma unroll
                for (int j = RP_J0; j <= RP_J1; j++)
                    if (a + j < n) mx[j - RP_J0] = fmax(mx[j - RP_J0], fabs(sT[u + j] - s0));
            }
        }
        __syncthreads();
        if (tid < PT_HALO) sT[tid] = sT[PT_TILE + tid];      // (the next tile's halo; read behind two more barriers)
        lapc(5);
    }
    if (tid < PT_HALO) {
        const int a = tid;
#pragma unroll
        for (int j = RP_J0; j <= RP_J1; j++)
            if (a < j) mx[j - RP_J0] = fmax(mx[j - RP_J0], fabs(sEdge[PT_HALO + (a + PT_HALO - j)] - sEdge[a]));
    }
#pragma unroll
    for (int j = 0; j < RP_NJ; j++) {
        double vv = mx[j];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const double oo = __hiloint2double(__shfl_xor(__double2hiint(vv), d), __shfl_xor(__double2loint(vv), d)); vv = oo > vv ? oo : vv; }
        if ((tid & 63) == 0) shM[j][tid >> 6] = vv;
    }
    lapc(7);
    const int anyBad = __syncthreads_or(bad);
    if (tid == 0) {
        const double rn = (double)n;
        double hLo = 0.0, hHi = 0.0;
        for (int j = RP_J0; j <= RP_J1; j++) {
            double vv = 0.0; for (int kq = 0; kq < RP_T / 64; kq++) vv = shM[j - RP_J0][kq] > vv ? shM[j - RP_J0][kq] : vv;
            const double rj = (double)j, c = rn / (rj * (rn - rj));
            const double lo2 = vv - errBound > 0.0 ? vv - errBound : 0.0, hi2 = vv + errBound;
            const double aa = c * (lo2 * lo2) * (1.0 - 1e-15), bb = c * (hi2 * hi2) * (1.0 + 1e-15);
            hLo = aa > hLo ? aa : hLo; hHi = bb > hHi ? bb : hHi;
        }
        auto norm = [&](double h) { double tq = tss; if (tq <= h + 0.0001) tq = h + 1.0; return h / ((tq - h) / (rn - 2.0)); };   // CBSTStatistic.cs:334-337
        if (anyBad || (tss <= hLo + 0.0001) != (tss <= hHi + 0.0001)) { pstat[2 * b] = -INFINITY; pstat[2 * b + 1] = INFINITY; }
        else { pstat[2 * b] = norm(hLo) * (1.0 - 1e-15); pstat[2 * b + 1] = norm(hHi) * (1.0 + 1e-15); }
    }
    __syncthreads();
}
__global__ void __launch_bounds__(RP_T) __attribute__((amdgpu_waves_per_eu(RP_T / 128, RP_T / 128))) k_perm_rp(const PermReq* __restrict__ reqs, int nreq) {
    int ri = 0;
    { int lo = 0, hi = nreq - 1; while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (reqs[mid].rpBase <= (int)blockIdx.x) lo = mid; else hi = mid - 1; } ri = lo; }
    const PermReq& R = reqs[ri];
    if (R.fy != 3) return;
    const int w = (int)blockIdx.x - R.rpBase, tid = threadIdx.x;
    if (w >= R.rpWGs) return;
    const int n = R.n, K = R.rp.K, nT = R.rp.nT;
    int32_t* sA = RP_LDS(int32_t, RPL_A); uint32_t* sTb = RP_LDS(uint32_t, RPL_TB); uint32_t* sDb = RP_LDS(uint32_t, RPL_DB);
    uint32_t* hKey = RP_LDS(uint32_t, RPL_HKEY); uint32_t* hStamp = RP_LDS(uint32_t, RPL_HSTAMP); uint32_t* sRow = RP_LDS(uint32_t, RPL_ROW);
    uint32_t* sCnt = RP_LDS(uint32_t, RPL_CNT); uint32_t* sMisc = RP_LDS(uint32_t, RPL_MISC);
    const gptr<long long> clk = (w == 0 && tid == 0) ? as_global(R.rpClk) : as_global((long long*)nullptr); long long tClk = clk ? clock64() : 0;
    auto lapc = [&](int slot) { if (clk) { const long long t = clock64(); clk[slot] += t - tClk; tClk = t; } };
    for (int b = w; b < R.nb; b += R.rpWGs) {
        const RpPtr P = rp_ptr(R, w, b);
        if (tid < RP_MAXK) { sCnt[tid] = 0u; RP_LDS(uint32_t, RPL_INOFF)[tid] = R.rp.inOff[tid]; RP_LDS(uint32_t, RPL_INCAP)[tid] = R.rp.inCap[tid]; }      // (the statistic of the previous permutation wrote over them)
        if (tid == 0) { sMisc[0] = 0u; sMisc[1] = 0u; sMisc[2] = 0u; P.endsOwn[nT] = 0u; }
        if (tid < K - 1) P.endsIn[(size_t)tid * (nT + 1) + nT] = 0u;
        __syncthreads();
        for (int k = K - 1; k >= 0; k--) {
            const int lo = k << RP_RSHIFT, hi = n < lo + RP_R ? n : lo + RP_R;
            for (int p = tid; p < hi - lo; p += RP_T) sA[p] = lo + p;
            for (int z = tid; z < 512; z += RP_T) { sTb[z] = 0u; sDb[z] = 0u; }
            for (int z = tid; z < RP_HASH; z += RP_T) { hKey[z] = 0xFFFFFFFFu; hStamp[z] = 0u; }
            if (k < K - 1) for (int t = tid; t <= nT; t += RP_T) sRow[t] = P.endsIn[(size_t)k * (nT + 1) + t];
            __syncthreads();
            lapc(0);
            if (k < K - 1) rp_range_inbox(R, w, b, k);
            lapc(1);
            rp_range_own(R, w, b, k, clk);
            if (clk) tClk = clock64();
            __syncthreads();
        }
        if (sMisc[1]) { __syncthreads(); if (tid == 0) { const gptr<double> pstat = as_global(R.pstat); pstat[2 * b] = -INFINITY; pstat[2 * b + 1] = INFINITY; } continue; }
        rp_statistic(R, w, b, clk);
        if (clk) tClk = clock64();
    }
}

// ---- segments of at most 200 bins (the non-hybrid test, TMaxP: CBSTStatistic.cs:599-934): one wave per permutation, and the value is EXACT.  The swaps and the prefix sums run
// on lane 0 in the reference's order (200 dependent steps on LDS) together with build_blocks' block extremes (CBSTStatistic.cs:44-110).  The arc maximum itself is order-free,
// but WHICH arcs the reference looks at is part of the result for blocks this small: for a block pair it scans the lengths alenlo .. min(alen, n - alen) and n - min(alen, n - alen)
// .. alenhi, alen being the distance between the pair's extremes (CBSTStatistic.cs:233-326) — when the extremes are closer than the minimum width nothing of the pair is scanned,
// and with a dozen elements per block that happens (a randomised soak run found a permutation whose exhaustive maximum is larger than the reference's).  Every lane therefore
// evaluates its arcs with the host's expression and keeps those whose length lies in the reference's ranges for the arc's block pair; pairs the reference skips because their
// bound is below the running maximum cannot hold the maximum of the scanned set, so the order of its visits does not matter.  The draws come with the request (host generator).
#define PS_MAXN 256
#define PS_MAXB 16
__global__ void __launch_bounds__(64) k_perm_small(const PermReq* __restrict__ reqs, int nreq) {
    __shared__ double px[PS_MAXN], sx[PS_MAXN], sC[PS_MAXN];
    __shared__ double sBmn[PS_MAXB], sBmx[PS_MAXB];
    __shared__ int sBB[PS_MAXB + 1], sImn[PS_MAXB], sImx[PS_MAXB], sBlkOf[PS_MAXN + 1];
    __shared__ int sR[PS_MAXB * PS_MAXB][4];
    __shared__ double sBss0;
    int ri = 0;
    { int lo = 0, hi = nreq - 1; while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (reqs[mid].blockBase <= (int)blockIdx.x) lo = mid; else hi = mid - 1; } ri = lo; }
    const PermReq& R = reqs[ri];
    if (R.fy != 2) return;
    const int n = R.n, al0 = R.al0, b = (int)blockIdx.x - R.blockBase, lane = threadIdx.x;
    const double rn = (double)n;
    const int nb = n >= 50 ? (int)rint(sqrt(rn)) : 1;            // dn_round(sqrt(n)): half to even (at most 14 for n <= 200)
    const uint32_t* __restrict__ draws = R.P.draws + (size_t)b * n;
    for (int i = lane; i < n; i += 64) px[i] = R.x[i];
    for (int L = lane; L < n; L += 64) { const double rj = (double)L; sC[L] = (L >= al0 && L <= n - al0) ? rn / (rj * (rn - rj)) : -1.0; }
    if (lane < nb) sBB[lane + 1] = (int)rint(rn * (((double)lane + 1.0) / (double)nb));      // bb[i] = round(rn * ((i + 1.0) / nb)), 1-based last position of block i
    if (lane == 0) sBB[0] = 0;
    __syncthreads();
    for (int p = 1 + lane; p <= n; p += 64) { int k = 0; while (k + 1 < nb && p > sBB[k + 1]) k++; sBlkOf[p] = k; }
    if (lane == 0) {
        for (int i = n - 1; i >= 0; i--) {                          // ChangePoint.cs:411-419
            const double cc = (double)draws[n - 1 - i] * (1.0 / 4294967296.0);
            int j = (int)(cc * (double)(i + 1)); j = j > i ? i : j;
            const double t = px[i]; px[i] = px[j]; px[j] = t;
        }
        // build_blocks: prefix sums; per block the first minimum / maximum (1-based positions); the global extremes start at (0, position n) and are replaced by strictly smaller / larger block extremes
        double run = 0.0, gmn = 0.0, gmx = 0.0; int igmn = n, igmx = n;
        for (int k = 0; k < nb; k++) {
            const int ilo = sBB[k] + 1, ihi = sBB[k + 1];
            run = run + px[ilo - 1]; sx[ilo - 1] = run;
            double mn = run, mx = run; int imn = ilo, imx = ilo;
            for (int i = ilo + 1; i <= ihi; i++) { run = run + px[i - 1]; sx[i - 1] = run; if (run < mn) { mn = run; imn = i; } if (run > mx) { mx = run; imx = i; } }
            sBmn[k] = mn; sBmx[k] = mx; sImn[k] = imn; sImx[k] = imx;
            if (mn < gmn) { gmn = mn; igmn = imn; }
            if (mx > gmx) { gmx = mx; igmx = imx; }
        }
        const double rj = (double)(igmx > igmn ? igmx - igmn : igmn - igmx), d = gmx - gmn;
        sBss0 = (rn / (rj * (rn - rj))) * (d * d);
    }
    __syncthreads();
    // the lengths the reference scans for every block pair (bi <= bj): [lo1, hi1] and [lo2, hi2]
    for (int t = lane; t < nb * nb; t += 64) {
        const int bi = t / nb, bj = t % nb;
        int lo1 = 1, hi1 = 0, lo2 = 1, hi2 = 0;
        if (bi <= bj) {
            const int ilo = sBB[bi] + 1, ihi = sBB[bi + 1], jlo = sBB[bj] + 1, jhi = sBB[bj + 1], nal0 = n - al0;
            int alenhi = jhi - ilo; if (alenhi > nal0) alenhi = nal0;
            int alenlo = bi == bj ? 1 : jlo - ihi; if (alenlo < al0) alenlo = al0;
            const double s1 = fabs(sBmx[bj] - sBmn[bi]), s2 = fabs(sBmx[bi] - sBmn[bj]);
            int alen = s1 > s2 ? sImx[bj] - sImn[bi] : sImn[bj] - sImx[bi]; alen = alen < 0 ? -alen : alen;
            int amax = alen > n - alen ? n - alen : alen;
            const double rnov2 = rn / 2;
            if ((double)alenlo <= rnov2 && alenlo <= amax) { lo1 = alenlo; hi1 = amax; }
            amax = n - amax;
            if ((double)alenhi >= rnov2 && alenhi >= amax) { lo2 = amax; hi2 = alenhi; }
        }
        sR[bi * PS_MAXB + bj][0] = lo1; sR[bi * PS_MAXB + bj][1] = hi1; sR[bi * PS_MAXB + bj][2] = lo2; sR[bi * PS_MAXB + bj][3] = hi2;
    }
    __syncthreads();
    double best = -1.0;
    for (int a = lane; a < n; a += 64) {                            // arc between the prefix sums a and a + L = 1-based positions p = a + 1, q = p + L
        const double s0 = sx[a];
        const int bi = sBlkOf[a + 1];
        for (int L = al0; a + L < n && L <= n - al0; L++) {
            const int* r = sR[bi * PS_MAXB + sBlkOf[a + 1 + L]];
            if (!((L >= r[0] && L <= r[1]) || (L >= r[2] && L <= r[3]))) continue;
            const double d = fabs(sx[a + L] - s0), v = sC[L] * (d * d); best = v > best ? v : best;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const double o = __hiloint2double(__shfl_xor(__double2hiint(best), d), __shfl_xor(__double2loint(best), d)); best = o > best ? o : best; }
    if (lane == 0) {
        double bss = sBss0; if (best > bss) bss = best;
        double t = R.tss; if (t <= bss + 0.0001) t = bss + 1.0;       // CBSTStatistic.cs:334-337
        const double v = bss / ((t - bss) / (rn - 2.0));
        R.pstat[2 * b] = v; R.pstat[2 * b + 1] = v;
    }
}

// ================================================================================================ host: scalar pieces of the reference
namespace cbs {

static inline double sq(double v) { return v * v; }          // Math.Pow(v, 2) := exact square (Q13)
static inline int dn_round(double v) { return (int)std::nearbyint(v); }   // Convert.ToInt32: half to even

// MathNet MersenneTwister as assumed in SURVEY §8c (parity unpinned): init_genrand, NextDouble = u32 * 2^-32
struct MT {
    uint32_t mt[624]; int mti;
    long long drawn = 0;       // outputs taken since the object was created / adopted a state (Rng: the chromosome's stream position is a base + this)
    explicit MT(uint32_t s) { mt[0] = s; for (mti = 1; mti < 624; mti++) mt[mti] = 1812433253u * (mt[mti - 1] ^ (mt[mti - 1] >> 30)) + (uint32_t)mti; }
    uint32_t u32() {
        drawn++;
        if (mti >= 624) {
            static const uint32_t mag[2] = {0u, 0x9908b0dfu};
            int k; uint32_t y;
            for (k = 0; k < 227; k++) { y = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu); mt[k] = mt[k + 397] ^ (y >> 1) ^ mag[y & 1u]; }
            for (; k < 623; k++) { y = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu); mt[k] = mt[k - 227] ^ (y >> 1) ^ mag[y & 1u]; }
            y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu); mt[623] = mt[396] ^ (y >> 1) ^ mag[y & 1u];
            mti = 0;
        }
        uint32_t y = mt[mti++];
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        return y;
    }
    double next_double() { return u32() * (1.0 / 4294967296.0); }
    void get_state(uint32_t* s625) const { memcpy(s625, mt, sizeof mt); s625[624] = (uint32_t)mti; }
    void set_state(const uint32_t* s625) { memcpy(mt, s625, sizeof mt); mti = (int)s625[624]; }
    int32_t next_full_range_int32() { uint32_t g[4]; for (int b = 0; b < 4; b++) g[b] = u32(); return canvas_mathnet_full_range_int32(g, canvas_mathnet_seed_variant()); }      // (include/canvas_mathnet.h: the switch the oracle shares)
};

// ---- R nmath subset used by GetBoundary (R.cs:8-160,528-548)
static double bd0(double x, double np) {
    if (std::fabs(x - np) < 0.1 * (x + np)) {
        double v = (x - np) / (x + np), s = (x - np) * v, ej = 2 * x * v;
        v = v * v;
        for (int j = 1;; j++) { ej *= v; double s1 = s + ej / ((j << 1) + 1); if (s1 == s) return s1; s = s1; }
    }
    return x * std::log(x / np) + np - x;
}
static double stirlerr(double n) {
    static const double S0 = 0.083333333333333333333, S1 = 0.00277777777777777777778, S2 = 0.00079365079365079365079365, S3 = 0.000595238095238095238095238, S4 = 0.0008417508417508417508417508;
    static const double h[31] = {0.0, 0.1534264097200273452913848, 0.0810614667953272582196702, 0.0548141210519176538961390, 0.0413406959554092940938221, 0.03316287351993628748511048,
        0.02767792568499833914878929, 0.02374616365629749597132920, 0.02079067210376509311152277, 0.01848845053267318523077934, 0.01664469118982119216319487, 0.01513497322191737887351255,
        0.01387612882307074799874573, 0.01281046524292022692424986, 0.01189670994589177009505572, 0.01110455975820691732662991, 0.010411265261972096497478567, 0.009799416126158803298389475,
        0.009255462182712732917728637, 0.008768700134139385462952823, 0.008330563433362871256469318, 0.007934114564314020547248100, 0.007573675487951840794972024, 0.007244554301320383179543912,
        0.006942840107209529865664152, 0.006665247032707682442354394, 0.006408994188004207068439631, 0.006171712263039457647532867, 0.005951370112758847735624416, 0.005746216513010115682023589,
        0.005554733551962801371038690};
    if (n <= 15.0) { double nn = n + n; if (nn == (int)nn) return h[(int)nn]; return std::lgamma(n + 1.0) - (n + 0.5) * std::log(n) + n - 0.918938533204672741780329736406; }
    double nn = n * n;
    if (n > 500) return (S0 - S1 / nn) / n;
    if (n > 80) return (S0 - (S1 - S2 / nn) / nn) / n;
    if (n > 35) return (S0 - (S1 - (S2 - S3 / nn) / nn) / nn) / n;
    return (S0 - (S1 - (S2 - (S3 - S4 / nn) / nn) / nn) / nn) / n;
}
static double dbinom_raw(double x, double n, double p, double q) {
    if (p == 0) return x == 0 ? 1.0 : 0.0;
    if (q == 0) return x == n ? 1.0 : 0.0;
    if (x == 0) { if (n == 0) return 1.0; double lc = (p < 0.1) ? (-bd0(n, n * q) - n * p) : (n * std::log(q)); return std::exp(lc); }
    if (x == n) { double lc = (q < 0.1) ? -bd0(n, n * p) - n * q : n * std::log(p); return std::exp(lc); }
    if (x < 0 || x > n) return 0.0;
    double lc = stirlerr(n) - stirlerr(x) - stirlerr(n - x) - bd0(x, n * p) - bd0(n - x, n * q);
    double lf = std::log(2 * M_PI) + std::log(x) + std::log1p(-x / n);
    return std::exp(lc - 0.5 * lf);
}
static double phyper_lower(double x, double NR, double NB, double n) {
    x = std::floor(x + 1e-7); NR = std::floor(NR + 0.5); NB = std::floor(NB + 0.5); n = std::floor(n + 0.5);
    bool lower = true;
    if (x * (NR + NB) > n * NR) { double o = NB; NB = NR; NR = o; x = n - x - 1; lower = !lower; }
    if (x < 0) return lower ? 0.0 : 1.0;
    if (x >= NR || x >= n) return lower ? 1.0 : 0.0;
    double d;
    if (n < x || NR < x || (n - x) > NB) d = 0.0;
    else if (n == 0) d = x == 0 ? 1.0 : 0.0;
    else { double p = n / (NR + NB), q = (NR + NB - n) / (NR + NB); d = dbinom_raw(x, NR, p, q) * dbinom_raw(n - x, NB, p, q) / dbinom_raw(n, NR + NB, p, q); }
    double sum = 0, term = 1, xx = x;
    while (xx > 0 && term >= 2.2204460492503131E-16 * sum) { term *= xx * (NB - n + xx) / (n + 1 - xx) / (NR + 1 - xx); sum += term; xx--; }
    double pv = d * (1 + sum);
    return lower ? pv : (0.5 - pv + 0.5);
}
static double binom_ln(int n, int k) { if (k < 0 || n < 0 || k > n) return -std::numeric_limits<double>::infinity(); return std::lgamma(n + 1.0) - std::lgamma(k + 1.0) - std::lgamma(n - k + 1.0); }

// GetBoundary.cs:19-157
static void eta_boundary(uint32_t nPerm, double eta0, uint32_t n1s, std::vector<uint32_t>& sb, uint32_t off) {
    double dn = (double)nPerm - (double)n1s; uint32_t k = 0;
    for (uint32_t i = 1; i <= nPerm; i++) if (phyper_lower((double)k, (double)n1s, dn, (double)i) <= eta0) { sb[off + k] = i; k++; }
}
static double p_exceed(uint32_t nPerm, uint32_t n1s, const std::vector<uint32_t>& sb, uint32_t off) {
    int n = (int)nPerm, k = (int)n1s, n1 = (int)(nPerm - sb[off]);
    double dl = binom_ln(n, k), p = std::exp(binom_ln(n1, k) - dl);
    if (n1s >= 2) { n1 = (int)sb[off]; n = (int)(nPerm - sb[off + 1]); k = (int)(n1s - 1); p += std::exp(std::log((double)n1) + binom_ln(n, k) - dl); }
    if (n1s >= 3) {
        n1 = (int)sb[off]; int n2 = (int)sb[off + 1]; n = (int)(nPerm - sb[off + 2]); k = (int)(n1s - 2);
        p += std::exp(std::log((double)n1) + std::log(n1 - 1.0) - std::log(2.0) + binom_ln(n, k) - dl) + std::exp(std::log((double)n1) + std::log((double)(n2 - n1)) + binom_ln(n, k) - dl);
    }
    if (n1s > 3) for (int i = 4; i <= (int)n1s; i++) {
        n1 = (int)sb[off + i - 4]; int k1 = i - 1, k2 = i - 2, k3 = i - 3, n2 = (int)sb[off + i - 3], n3 = (int)sb[off + i - 2];
        n = (int)(nPerm - sb[off + i - 1]); k = (int)(n1s - i + 1);
        p += std::exp(binom_ln(n1, k1) + binom_ln(n, k) - dl) + std::exp(binom_ln(n1, k2) + std::log((double)(n3 - n1)) + binom_ln(n, k) - dl) +
             std::exp(binom_ln(n1, k3) + std::log((double)(n2 - n1)) + std::log((double)(n3 - n2)) + binom_ln(n, k) - dl) +
             std::exp(binom_ln(n1, k3) + std::log((double)(n2 - n1)) - std::log(2.0) + std::log(n2 - n1 - 1.0) + binom_ln(n, k) - dl);
    }
    return p;
}
static void eta_boundary_fast(uint32_t nPerm, double eta0, uint32_t n1s, std::vector<uint32_t>& sb, uint32_t off);
static void compute_boundary_fresh(uint32_t nPerm, double alpha, double eta, std::vector<uint32_t>& sb);
// CanvasPartition's default parameters take the table that was computed once and compiled in (cbs_boundary_default.hpp); CANVAS_CBS_NO_EMBEDDED_BOUNDARY=1 computes it anew (test hook)
static void compute_boundary(uint32_t nPerm, double alpha, double eta, std::vector<uint32_t>& sb) {
    if (nPerm == CBS_DEFAULT_NPERM && alpha == CBS_DEFAULT_ALPHA && eta == 0.05 && sizeof(kCbsDefaultBoundary) > 4 && !cvx_hook("CANVAS_CBS_NO_EMBEDDED_BOUNDARY")) {
        sb.assign(kCbsDefaultBoundary, kCbsDefaultBoundary + sizeof(kCbsDefaultBoundary) / sizeof(uint32_t));
        return;
    }
    compute_boundary_fresh(nPerm, alpha, eta, sb);
}
static void compute_boundary_fresh(uint32_t nPerm, double alpha, double eta, std::vector<uint32_t>& sb) {
    uint32_t maxOnes = (uint32_t)(std::floor(nPerm * alpha) + 1);
    sb.assign((size_t)maxOnes * (maxOnes + 1) / 2, 0);
    uint32_t l = 0; sb[0] = nPerm - (uint32_t)(nPerm * eta);
    double eta0 = eta;
    for (uint32_t j = 2; j <= maxOnes; j++) {
        double hi = eta0 * 1.1; eta_boundary_fast(nPerm, hi, j, sb, l + 1); double pHi = p_exceed(nPerm, j, sb, l + 1);
        double lo = eta0 * 0.25; eta_boundary_fast(nPerm, lo, j, sb, l + 1); double pLo = p_exceed(nPerm, j, sb, l + 1);
        while ((hi - lo) / lo > 1E-2) {
            eta0 = lo + (hi - lo) * (eta - pLo) / (pHi - pLo);
            eta_boundary_fast(nPerm, eta0, j, sb, l + 1); double pe = p_exceed(nPerm, j, sb, l + 1);
            if (pe > eta) { hi = eta0; pHi = pe; } else { lo = eta0; pLo = pe; }
        }
        l += j;
    }
}

// TailProbability.cs:21-105 (Normal CDF of MathNet -> erfc, parity unpinned)
static double pnorm(double x) { return 0.5 * std::erfc(-x / M_SQRT2); }
static double nu(double x, double tol) {
    double l1;
    if (x > 0.01) {
        l1 = std::log(2.0) - 2 * std::log(x); double l0 = l1; int k = 2; double dk = 0;
        for (int i = 0; i < k; i++) { dk = dk + 1; double xk = -x * std::sqrt(dk) / 2.0; l1 = l1 - 2.0 * pnorm(xk) / dk; }
        while (std::fabs((l1 - l0) / l1) > tol) { l0 = l1; for (int i = 0; i < k; i++) { dk = dk + 1; double xk = -x * std::sqrt(dk) / 2.0; l1 = l1 - 2.0 * pnorm(xk) / dk; } k *= 2; }
    } else l1 = -0.583 * x;
    return std::exp(l1);
}
static double integral_inv(double x, double a) {
    double y = x + a - 0.5;
    double r = (8.0 * y) / (1.0 - 4.0 * sq(y)) + 2.0 * std::log((1.0 + 2.0 * y) / (1.0 - 2.0 * y));
    y = x - 0.5;
    return r - (8.0 * y) / (1.0 - 4.0 * sq(y)) - 2.0 * std::log((1.0 + 2.0 * y) / (1.0 - 2.0 * y));
}
// A process-wide pool for the one expensive scalar piece of the hybrid test: Nu(x) is a series of up to ~10^6 normal-CDF evaluations
// for the small arguments of long segments (TailProbability.cs:52-85), 100 of them per TailP call.  The grid points are independent,
// so they are evaluated by the pool (the chromosome threads mostly wait for the device) and then accumulated in the reference's order.
struct HostPool {
    std::vector<std::thread> th; std::mutex mu; std::condition_variable cv; std::vector<std::function<void()>> q; bool stop = false;
    HostPool() { unsigned n = std::thread::hardware_concurrency(); if (n == 0) n = 8; if (n > 64) n = 64; for (unsigned i = 0; i < n; i++) th.emplace_back([this]() { run(); }); }
    ~HostPool() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); for (auto& t : th) t.join(); }
    void run() { for (;;) { std::function<void()> f; { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return stop || !q.empty(); }); if (q.empty()) return; f = std::move(q.back()); q.pop_back(); } f(); } }
    void submit(std::function<void()> f) { { std::lock_guard<std::mutex> lk(mu); q.push_back(std::move(f)); } cv.notify_one(); }
    static HostPool& get() { static HostPool p; return p; }
};
// eta_boundary with the same evaluations and the same comparisons, on the pool.  The scan "for i = 1 .. nPerm: if phyper(k, ..., i) <= eta0 { sb[k] = i; k++ }" is sequential
// through k, but where it will stop for each k can be PREDICTED (phyper(k, ., i) falls with i: a bisection per k, all k at once), and a prediction can be CHECKED with exactly
// the evaluations the scan would make: for every i the k the scan would have there, the comparison must fail between two predicted stops and succeed at them.  If every check
// holds, the scan's result is the prediction; if one does not (the computed phyper is not monotone at the crossing), the sequential scan runs.  (10 000 evaluations in ~700 calls:
// 0.86 s of the first canvas_cbs call of a process — the table is cached per (nPerm, alpha) afterwards — and of every CanvasPartition -m CBS run.)
template <class F> static void pool_for(int ntasks, F f) {
    if (ntasks <= 1) { for (int t = 0; t < ntasks; t++) f(t); return; }
    std::mutex mu; std::condition_variable cv; int left = ntasks;
    for (int t = 0; t < ntasks; t++) HostPool::get().submit([&, t]() { f(t); std::lock_guard<std::mutex> lk(mu); if (--left == 0) cv.notify_one(); });
    std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return left == 0; });
}
static void eta_boundary_fast(uint32_t nPerm, double eta0, uint32_t n1s, std::vector<uint32_t>& sb, uint32_t off) {
    static const bool sequential = cvx_hook("CANVAS_CBS_SEQUENTIAL_BOUNDARY") != nullptr;
    if (sequential || nPerm < 2000) { eta_boundary(nPerm, eta0, n1s, sb, off); return; }
    const double dn = (double)nPerm - (double)n1s;
    const int K = (int)n1s;                                    // k = 0 .. n1s - 1 can stop the scan (phyper(n1s, n1s, ., i) = 1)
    std::vector<uint32_t> cross((size_t)K, nPerm + 1);
    const int T = 32;
    pool_for(std::min(T, K), [&](int t) {
        for (int k = t; k < K; k += T) {
            if (!(phyper_lower((double)k, (double)n1s, dn, (double)nPerm) <= eta0)) continue;          // never stops
            uint32_t lo = 1, hi = nPerm;                       // smallest i with phyper <= eta0, assuming it falls with i
            while (lo < hi) { const uint32_t mid = lo + (hi - lo) / 2; if (phyper_lower((double)k, (double)n1s, dn, (double)mid) <= eta0) hi = mid; else lo = mid + 1; }
            cross[(size_t)k] = lo;
        }
    });
    std::vector<uint32_t> pred; pred.reserve((size_t)K);
    { uint32_t prev = 0; for (int k = 0; k < K; k++) { const uint32_t p = std::max(cross[(size_t)k], prev + 1); if (p > nPerm) break; pred.push_back(p); prev = p; } }
    // the check: every evaluation of the sequential scan, in chunks of i
    std::atomic<int> bad{0};
    const uint32_t chunk = (nPerm + T - 1) / T;
    pool_for(T, [&](int t) {
        const uint32_t a = 1 + (uint32_t)t * chunk, b = std::min<uint32_t>(nPerm + 1, a + chunk);
        size_t k = (size_t)(std::lower_bound(pred.begin(), pred.end(), a) - pred.begin());      // stops in front of a = the scan's k at i = a
        for (uint32_t i = a; i < b && !bad.load(std::memory_order_relaxed); i++) {
            const bool hit = phyper_lower((double)k, (double)n1s, dn, (double)i) <= eta0;
            const bool expect = k < pred.size() && pred[k] == i;
            if (hit != expect) { bad = 1; break; }
            if (hit) k++;
        }
    });
    if (bad) { eta_boundary(nPerm, eta0, n1s, sb, off); return; }
    for (size_t k = 0; k < pred.size(); k++) sb[off + k] = pred[k];
}
static double tail_p(double b, double delta, int m, int nGrid, double tol) {
    double dincr = (0.5 - delta) / nGrid, bs = b / std::sqrt((double)m), tl = 0.5 - dincr, t = 0.5 - 0.5 * dincr, tp = 0.0;
    std::vector<double> xs(nGrid), tls(nGrid), nus(nGrid);
    for (int i = 0; i < nGrid; i++) { tl = tl + dincr; t = t + dincr; xs[i] = bs / std::sqrt(t * (1 - t)); tls[i] = tl; }
    {   // nus[i] = Nu(xs[i]) on the pool
        std::mutex mu; std::condition_variable cv; int left = nGrid;
        for (int i = 0; i < nGrid; i++) HostPool::get().submit([&, i]() { nus[i] = nu(xs[i], tol); std::lock_guard<std::mutex> lk(mu); if (--left == 0) cv.notify_one(); });
        std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return left == 0; });
    }
    for (int i = 0; i < nGrid; i++) tp = tp + sq(nus[i]) * integral_inv(tls[i], dincr);
    tp = 9.973557E-2 * (b * b * b) * std::exp(-sq(b) / 2) * tp;
    return 2.0 * tp;
}

// ---- Array.Sort<double,int> of .NET Core 2.0 (introsort; tie order parity-unpinned, Q11) — used by the host replay of TMaxO/TMaxP
struct KI { double* k; int* v; };
static inline void sw(KI a, int i, int j) { if (i != j) { std::swap(a.k[i], a.k[j]); std::swap(a.v[i], a.v[j]); } }
static inline void sig(KI a, int i, int j) { if (i != j && a.k[i] > a.k[j]) { std::swap(a.k[i], a.k[j]); std::swap(a.v[i], a.v[j]); } }
static void ins(KI a, int lo, int hi) { for (int i = lo; i < hi; i++) { int j = i; double t = a.k[i + 1]; int tv = a.v[i + 1]; while (j >= lo && t < a.k[j]) { a.k[j + 1] = a.k[j]; a.v[j + 1] = a.v[j]; j--; } a.k[j + 1] = t; a.v[j + 1] = tv; } }
static void dheap(KI a, int i, int n, int lo) { double d = a.k[lo + i - 1]; int dv = a.v[lo + i - 1]; while (i <= n / 2) { int c = 2 * i; if (c < n && a.k[lo + c - 1] < a.k[lo + c]) c++; if (a.k[lo + c - 1] < d) break; a.k[lo + i - 1] = a.k[lo + c - 1]; a.v[lo + i - 1] = a.v[lo + c - 1]; i = c; } a.k[lo + i - 1] = d; a.v[lo + i - 1] = dv; }
static void hsort(KI a, int lo, int hi) { int n = hi - lo + 1; for (int i = n / 2; i >= 1; i--) dheap(a, i, n, lo); for (int i = n; i > 1; i--) { sw(a, lo, lo + i - 1); dheap(a, 1, i - 1, lo); } }
static int part(KI a, int lo, int hi) { int mid = lo + (hi - lo) / 2; sig(a, lo, mid); sig(a, lo, hi); sig(a, mid, hi); double pv = a.k[mid]; sw(a, mid, hi - 1); int l = lo, r = hi - 1; while (l < r) { while (pv > a.k[++l]) {} while (pv < a.k[--r]) {} if (l >= r) break; sw(a, l, r); } sw(a, l, hi - 1); return l; }
static void intro(KI a, int lo, int hi, int depth) {
    while (hi > lo) {
        int ps = hi - lo + 1;
        if (ps <= 16) { if (ps == 1) return; if (ps == 2) { sig(a, lo, hi); return; } if (ps == 3) { sig(a, lo, hi - 1); sig(a, lo, hi); sig(a, hi - 1, hi); return; } ins(a, lo, hi); return; }
        if (depth == 0) { hsort(a, lo, hi); return; }
        depth--; int p = part(a, lo, hi); intro(a, p + 1, hi, depth); hi = p - 1;
    }
}
static void dotnet_sort(double* keys, int* items, int length, int arrayLength) { if (length < 2) return; int fl = 0, n = arrayLength; while (n >= 1) { fl++; n /= 2; } intro(KI{keys, items}, 0, length - 1, 2 * fl); }

// ---- CBSTStatistic.cs restated for the host (small segments, permutations, tie replay)
struct Blocks { int nb; std::vector<int> bb, ibmin, ibmax; std::vector<double> bpsmin, bpsmax; double psmin0, psmax0; int ipsmin0, ipsmax0; };
static void build_blocks(const double* x, int n, double* sx, Blocks& B) {     // CBSTStatistic.cs:44-110
    double rn = (double)n;
    B.nb = n >= 50 ? dn_round(std::sqrt((double)n)) : 1;
    int nb = B.nb;
    B.bb.resize(nb); B.ibmin.resize(nb); B.ibmax.resize(nb); B.bpsmin.resize(nb); B.bpsmax.resize(nb);
    for (int i = 0; i < nb; i++) B.bb[i] = dn_round(rn * ((i + 1.0) / nb));
    int ilo = 1; double psum = 0;
    B.psmin0 = 0; B.psmax0 = 0; B.ipsmin0 = n; B.ipsmax0 = n;
    for (int j = 0; j < nb; j++) {
        sx[ilo - 1] = psum + x[ilo - 1];
        double mn = sx[ilo - 1], mx = mn; int imn = ilo, imx = ilo;
        for (int i = ilo + 1; i <= B.bb[j]; i++) { sx[i - 1] = sx[i - 2] + x[i - 1]; if (sx[i - 1] < mn) { mn = sx[i - 1]; imn = i; } if (sx[i - 1] > mx) { mx = sx[i - 1]; imx = i; } }
        B.ibmin[j] = imn; B.ibmax[j] = imx; B.bpsmin[j] = mn; B.bpsmax[j] = mx;
        if (mn < B.psmin0) { B.psmin0 = mn; B.ipsmin0 = imn; }
        if (mx > B.psmax0) { B.psmax0 = mx; B.ipsmax0 = imx; }
        psum = sx[B.bb[j] - 1]; ilo = B.bb[j] + 1;
    }
}
static void block_search(const double* sx, int n, int al0, const Blocks& B, double& bssmax, int& tmaxi, int& tmaxj) {   // CBSTStatistic.cs:128-326
    double rn = (double)n; int nb = B.nb, nb2 = nb * (nb + 1) / 2;
    std::vector<double> bssbij(nb2), bssijmax(nb2); std::vector<int> bloci(nb2), blocj(nb2), loc(nb2), alen(nb2);
    double rnov2 = rn / 2; int l = 0, nal0 = n - al0; const std::vector<int>& bb = B.bb;
    for (int i = 1; i <= nb; i++) for (int j = i; j <= nb; j++) {
        int ilo = i == 1 ? 1 : bb[i - 2] + 1, ihi = bb[i - 1], jlo = j == 1 ? 1 : bb[j - 2] + 1, jhi = bb[j - 1];
        int alenhi = jhi - ilo; if (alenhi > nal0) alenhi = nal0;
        double rjhi = (double)alenhi;
        int alenlo = i == j ? 1 : jlo - ihi; if (alenlo < al0) alenlo = al0;
        double s1 = std::fabs(B.bpsmax[j - 1] - B.bpsmin[i - 1]), s2 = std::fabs(B.bpsmax[i - 1] - B.bpsmin[j - 1]), s0 = std::max(s1, s2);
        double rjlo = (double)alenlo;
        double rnj = rn / std::min(rjlo * (rn - rjlo), rjhi * (rn - rjhi));
        double lim = rnj * sq(s0);
        if (bssmax <= lim) {
            loc[l] = l + 1; bloci[l] = i; blocj[l] = j; bssijmax[l] = lim;
            if (s1 > s2) { alen[l] = std::abs(B.ibmax[j - 1] - B.ibmin[i - 1]); double rj = (double)alen[l]; bssbij[l] = (rn / (rj * (rn - rj))) * sq(s1); }
            else { alen[l] = std::abs(B.ibmin[j - 1] - B.ibmax[i - 1]); double rj = (double)alen[l]; bssbij[l] = (rn / (rj * (rn - rj))) * sq(s2); }
            l++;
        }
    }
    int nb1 = l;
    for (int k = 0; k < nb1; k++) loc[k] = k + 1;
    dotnet_sort(bssbij.data(), loc.data(), nb1, nb2);
    auto scan = [&](int i2j, int ilo, int ihi, int jlo, int jhi) {
        int ixlo = std::max(0, jlo - ilo - i2j), ixhi = std::max(0, ihi + i2j - jhi);
        double sxmx = 0; int sxmxi = ilo + ixlo - 1;
        for (int i = ilo + ixlo; i <= ihi - ixhi; i++) { double a = std::fabs(sx[i + i2j - 1] - sx[i - 1]); if (sxmx < a) { sxmx = a; sxmxi = i; } }
        double rj = (double)i2j; double v = (rn / (rj * (rn - rj))) * sq(sxmx);
        if (v > bssmax) { bssmax = v; tmaxi = sxmxi; tmaxj = sxmxi + i2j; }
    };
    for (l = nb1 - 1; l >= 0; l--) {
        int k = loc[l] - 1;
        if (bssmax <= bssijmax[k]) {
            int bi = bloci[k], bj = blocj[k], alenmax = alen[k];
            int ilo = bi == 1 ? 1 : bb[bi - 2] + 1, ihi = bb[bi - 1], jlo = bj == 1 ? 1 : bb[bj - 2] + 1, jhi = bb[bj - 1];
            int alenhi = jhi - ilo; if (alenhi > nal0) alenhi = nal0;
            double rjhi = (double)alenhi;
            int alenlo = bi == bj ? 1 : jlo - ihi; if (alenlo < al0) alenlo = al0;
            double rjlo = (double)alenlo;
            if (alenmax > n - alenmax) alenmax = n - alenmax;
            if (rjlo <= rnov2 && alenlo <= alenmax) for (int i2j = alenlo; i2j <= alenmax; i2j++) scan(i2j, ilo, ihi, jlo, jhi);
            alenmax = n - alenmax;
            if (rjhi >= rnov2 && alenhi >= alenmax) for (int i2j = alenhi; i2j >= alenmax; i2j--) scan(i2j, ilo, ihi, jlo, jhi);
        }
    }
}
// Would block_search look at the arc from position p (1-based) of length L?  For a block pair it scans the lengths alenlo .. min(alen, n - alen) and n - min(alen, n - alen) .. alenhi,
// alen = the distance between the pair's extremes (CBSTStatistic.cs:233-326): the lengths in between cannot beat the arc between the extremes — but when that arc is shorter than
// the minimum width (or otherwise outside the scanned lengths) nothing vouches for them, and the reference simply does not see those arcs.  The device searches take the maximum
// over ALL admissible arcs; they are the reference's result only if the maximiser is an arc the reference scans (checked with this; otherwise the host replays the search).
static bool ref_scans_arc(const Blocks& B, int n, int al0, int p, int L) {
    const std::vector<int>& bb = B.bb; const int nb = B.nb, q = p + L;
    auto block_of = [&](int pos) { int k = 0; while (k + 1 < nb && pos > bb[k]) k++; return k; };       // 0-based block of a 1-based position
    const int bi = block_of(p), bj = block_of(q);
    const int ilo = bi == 0 ? 1 : bb[bi - 1] + 1, ihi = bb[bi], jlo = bj == 0 ? 1 : bb[bj - 1] + 1, jhi = bb[bj], nal0 = n - al0;
    int alenhi = jhi - ilo; if (alenhi > nal0) alenhi = nal0;
    int alenlo = bi == bj ? 1 : jlo - ihi; if (alenlo < al0) alenlo = al0;
    const double s1 = std::fabs(B.bpsmax[bj] - B.bpsmin[bi]), s2 = std::fabs(B.bpsmax[bi] - B.bpsmin[bj]);
    int alen = s1 > s2 ? std::abs(B.ibmax[bj] - B.ibmin[bi]) : std::abs(B.ibmin[bj] - B.ibmax[bi]);
    int amax = alen > n - alen ? n - alen : alen;
    const double rn = (double)n, rnov2 = rn / 2;
    if ((double)alenlo <= rnov2 && alenlo <= amax && L >= alenlo && L <= amax) return true;
    amax = n - amax;
    return (double)alenhi >= rnov2 && alenhi >= amax && L >= amax && L <= alenhi;
}
static double normalise(double bssmax, double tss, double rn) { if (tss <= bssmax + 0.0001) tss = bssmax + 1.0; return bssmax / ((tss - bssmax) / (rn - 2.0)); }   // CBSTStatistic.cs:334-337
static void tmaxo_host(const double* x, int n, double tss, double* sx, int iseg[2], double& ostat, int al0) {           // CBSTStatistic.cs:19-341
    Blocks B; build_blocks(x, n, sx, B);
    double rn = (double)n, psdiff = B.psmax0 - B.psmin0, rj = (double)std::abs(B.ipsmax0 - B.ipsmin0);
    double bssmax = (rn / (rj * (rn - rj))) * sq(psdiff);
    int ti = std::min(B.ipsmax0, B.ipsmin0), tj = std::max(B.ipsmax0, B.ipsmin0);
    if (psdiff <= 0) bssmax = 0; else block_search(sx, n, al0, B, bssmax, ti, tj);
    ostat = normalise(bssmax, tss, rn); iseg[0] = ti; iseg[1] = tj;
}
static double tmaxp_host(double tss, const double* px, int n, double* sx, int al0) {                                     // CBSTStatistic.cs:599-934
    Blocks B; build_blocks(px, n, sx, B);
    double rn = (double)n, psdiff = B.psmax0 - B.psmin0, rj = (double)std::abs(B.ipsmax0 - B.ipsmin0);
    double bssmax = (rn / (rj * (rn - rj))) * sq(psdiff); int a = 0, b = 0;
    block_search(sx, n, al0, B, bssmax, a, b);
    return normalise(bssmax, tss, rn);
}
static double htmaxp_host(int k, double tss, const double* px, int n, double* sx, int al0) {                             // CBSTStatistic.cs:354-586
    double rn = (double)n; int nb = (int)(rn / k);
    std::vector<double> bmx(nb), bmn(nb); std::vector<int> bb(nb);
    for (int i = 0; i < nb; i++) bb[i] = dn_round(rn * ((double)(i + 1) / nb));
    int ilo = 1; double psum = 0, h = 0.0;
    for (int j = 0; j < nb; j++) {
        sx[ilo - 1] = psum + px[ilo - 1];
        double mn = sx[ilo - 1], mx = mn; int imn = ilo, imx = ilo;
        for (int i = ilo; i < bb[j]; i++) { sx[i] = sx[i - 1] + px[i]; if (sx[i] < mn) { mn = sx[i]; imn = i + 1; } if (sx[i] > mx) { mx = sx[i]; imx = i + 1; } }
        bmn[j] = mn; bmx[j] = mx; psum = sx[bb[j] - 1]; ilo = bb[j] + 1;
        int d = std::abs(imn - imx);
        if (d <= k && d >= al0) { double rj = (double)d; double v = (rn / (rj * (rn - rj))) * sq(bmx[j] - bmn[j]); if (h < v) h = v; }
    }
    auto arcs = [&](double dsq, auto&& inner) {
        for (int j = al0; j <= k; j++) { double rj = (double)j, c = rn / (rj * (rn - rj)); if (c * dsq < h) break; double m = inner(j); double v = c * sq(m); if (h < v) h = v; }
    };
    arcs(sq(bmx[0] - bmn[0]), [&](int j) { double m = 0; for (int i = 1; i <= bb[0] - j; i++) { double a = std::fabs(sx[i + j - 1] - sx[i - 1]); if (m < a) m = a; } return m; });
    arcs(sq(std::max(std::fabs(bmx[0] - bmn[nb - 1]), std::fabs(bmx[nb - 1] - bmn[0]))), [&](int j) { double m = 0; int nmj = n - j; for (int i = 0; i < j; i++) { double a = std::fabs(sx[i + nmj] - sx[i]); if (m < a) m = a; } return m; });
    for (int l = 1; l < nb; l++) {
        int lo = bb[l - 1] + 1, hi = bb[l];
        arcs(sq(bmx[l] - bmn[l]), [&](int j) { double m = 0; for (int i = lo; i <= hi - j; i++) { double a = std::fabs(sx[i + j - 1] - sx[i - 1]); if (m < a) m = a; } return m; });
        arcs(sq(std::max(std::fabs(bmx[l] - bmn[l - 1]), std::fabs(bmx[l - 1] - bmn[l]))), [&](int j) { double m = 0; for (int i = lo - j; i <= lo - 1; i++) { double a = std::fabs(sx[i + j - 1] - sx[i - 1]); if (m < a) m = a; } return m; });
    }
    return normalise(h, tss, rn);
}

// (diagnostic, CANVAS_CBS_TIMING: thread-nanoseconds inside the runtime's allocation / stream-creation calls of the engines — what a cold call pays before its first kernel)
static std::atomic<long long> g_ns_alloc_arc{0}, g_ns_alloc_perm{0}, g_ns_alloc_tail{0}, g_ns_alloc_svc{0};
struct AllocClock { std::atomic<long long>& a; std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now(); explicit AllocClock(std::atomic<long long>& x) : a(x) {}
                    ~AllocClock() { a += (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t).count(); } };
struct Stats { std::atomic<long long> unscanned_max{0}; std::atomic<long long> tailp_dev{0}, tailp_host{0}; std::atomic<long long> ns_tmaxo_host{0}, ns_tailp{0}, ns_prep{0}; std::atomic<long long> ns_ensure{0}, ns_upload{0}, ns_submit{0}, ns_post{0}; std::atomic<long long> ns_dev{0}, ns_hostperm{0}, ns_tpermp{0}, ns_tmaxo{0}, ns_mt{0}; std::atomic<long long> dev_perms{0}, dev_batches{0}, exact_rechecks{0}, verified{0}, violations{0}; std::atomic<long long> tmaxo_calls{0}, tmaxo_elems{0}, perms{0}, perm_elems{0}, tpermp_draws{0}, tpermp_device{0}, tailp_exits{0}, big_t{0}, gpu_searches{0}, gpu_pairs{0}, tie_replays{0}; };

// GPU arc search service shared by the chromosome threads
struct ArcHostReq { ArcReq r; ArcPReq p; bool pruned = true; const void* hSx; void* hMax; void* hFirst; unsigned long long* hOut; bool done = false; int32_t rc = CANVAS_OK; };
struct PermService;
static int32_t service_submit_arc(PermService* svc, ArcHostReq& q);
// One arena per context for what the engines of a call ask for (EngineCache, below): a cold call creates some fifty-six engines from as many threads, and their ~170 small
// hipMalloc / hipHostMalloc / hipStreamCreate calls queued up behind the runtime's locks — 3.7 thread-seconds in the tail engines alone, 0.13 s of the first call
// (CANVAS_CBS_TIMING prints the figure).  canvas_cbs sizes the arena for the engines it is about to create: one device slab, one pinned slab, sixteen shared streams.
struct EngineCache;
static char* cache_take(EngineCache* ec, bool pinned, size_t bytes);       // nullptr: nothing left (the engine then allocates for itself)
static hipStream_t cache_tail_stream(EngineCache* ec);                    // nullptr: no pool
struct SvcRes;
static SvcRes* cache_svc_take(canvas_ctx* ctx);
static void cache_svc_give(canvas_ctx* ctx, SvcRes* r);
struct ArcGpu {       // one per chromosome thread: own buffers; the launches go through the launcher thread (PermService) so that the searches of all chromosomes share one launch
    canvas_ctx* ctx = nullptr; PermService* svc = nullptr; hipStream_t stream = nullptr; EngineCache* cache = nullptr;
    bool owned = true;            // dSx / dPr / pin are allocations of this engine (false: slices of the cache's arena, never freed one by one)
    int reserveN = 0;             // the call's longest chromosome: the first allocation is made for it (an engine that met a short chromosome first was allocated again later)
    double* dSx = nullptr; double* dMax = nullptr; int32_t* dFirst = nullptr; int cap = 0;
    char* dPr = nullptr;          // pruned search: block minima / maxima, pair list, per-pair maxima, result words
    char* pin = nullptr; size_t pinBytes = 0;
    char* pinEx = nullptr; int capEx = 0;      // the exhaustive search's per-length maxima (device + pinned): only flat data or the test hook ever need them
    // (a cold process creates some fifty engines — chromosome threads and helpers — and paid for 20 pinned bytes and three device arrays per bin of each: 0.3 s of the
    // CanvasPartition -m CBS executable; the pruned search needs the prefix sums and a few result words)
    int pinCap = 0;               // bins the pinned staging buffer holds: it grows with the segments THIS engine meets (pinning host memory costs ~1.4 ms per MB — sizing all
                                  // fifty-six buffers for the longest chromosome, 213 MB, was 0.3 s of a cold call; the device side is sized for the longest chromosome at once)
    char* pinSmall = nullptr; bool pinSmallOwned = false;      // the result mailbox (256 bytes) when the prefix sums travel from the caller's pageable array (one-shot hosts)
    bool pageable() const { return ctx->one_shot != 0; }
    int32_t ensure(int n) {
        if (pageable() && !pinSmall) {
            pinSmall = cache_take(cache, true, 256);
            if (!pinSmall) { CANVAS_HIP_TRY(ctx, hipHostMalloc((void**)&pinSmall, 256, hipHostMallocDefault)); pinSmallOwned = true; }
        }
        if (n <= cap && (pageable() || n <= pinCap)) return CANVAS_OK;
        AllocClock ac(g_ns_alloc_arc);
        if (n > cap) {
            if (dSx && owned) { (void)hipFree(dSx); (void)hipFree(dPr); }
            dSx = nullptr; dPr = nullptr;
            const int want = std::max(n, reserveN);
            cap = want + want / 4 + 1024;
            const size_t prBytes = arc_pr_bytes(cap), sxBytes = (size_t)cap * 8;
            char* d = cache_take(cache, false, ((prBytes + 255) & ~size_t(255)) + sxBytes);
            if (d) { dPr = d; dSx = (double*)(d + ((prBytes + 255) & ~size_t(255))); owned = false; }
            else { owned = true; CANVAS_HIP_TRY(ctx, hipMalloc((void**)&dPr, prBytes)); CANVAS_HIP_TRY(ctx, hipMalloc((void**)&dSx, sxBytes)); }
        }
        if (n > pinCap && !pageable()) {
            if (pin) (void)hipHostFree(pin);
            pin = nullptr; pinCap = n + n / 4 + 1024;
            pinBytes = (size_t)pinCap * 8 + 256; CANVAS_HIP_TRY(ctx, hipHostMalloc((void**)&pin, pinBytes, hipHostMallocDefault));
        }
        return CANVAS_OK;
    }
    static size_t arc_pr_bytes(int cap) { const size_t nb = (size_t)cap / AP_BK + 2; return nb * 24 + AP_PAIRCAP * 12 + 4096; }
    static void arena_bytes(int n, size_t& dev, size_t& pin) { const int cap = n + n / 4 + 1024; dev = ((arc_pr_bytes(cap) + 255) & ~size_t(255)) + (size_t)cap * 8 + 256; pin = 0; }
    int32_t ensure_exhaustive() {
        if (capEx >= cap) return CANVAS_OK;
        if (dMax) { (void)hipFree(dMax); (void)hipFree(dFirst); (void)hipHostFree(pinEx); dMax = nullptr; dFirst = nullptr; pinEx = nullptr; }
        CANVAS_HIP_TRY(ctx, hipMalloc((void**)&dMax, (size_t)cap * 8)); CANVAS_HIP_TRY(ctx, hipMalloc((void**)&dFirst, (size_t)cap * 4));
        CANVAS_HIP_TRY(ctx, hipHostMalloc((void**)&pinEx, (size_t)cap * 12 + 256, hipHostMallocDefault));
        capEx = cap;
        return CANVAS_OK;
    }
    ~ArcGpu() { if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); } if (dSx && owned) { (void)hipFree(dSx); (void)hipFree(dPr); } if (pin) (void)hipHostFree(pin); if (pinSmall && pinSmallOwned) (void)hipHostFree(pinSmall);
                if (dMax) { (void)hipFree(dMax); (void)hipFree(dFirst); (void)hipHostFree(pinEx); } }
};

// TMaxO with the O(n^2) search on the GPU.  Returns false when the caller must fall back to the host replay (ambiguous maximum).
static int32_t tmaxo_gpu(ArcGpu& G, const double* x, int n, double tss, double* sx, int iseg[2], double& ostat, int al0, Stats& st, bool& ok) {
    // sequential prefix sums and the initial incumbent exactly as the reference builds them (block structure does not change sx)
    Blocks B; build_blocks(x, n, sx, B);
    double rn = (double)n, psdiff = B.psmax0 - B.psmin0, rj0 = (double)std::abs(B.ipsmax0 - B.ipsmin0);
    double bss0 = (rn / (rj0 * (rn - rj0))) * sq(psdiff);
    int ti = std::min(B.ipsmax0, B.ipsmin0), tj = std::max(B.ipsmax0, B.ipsmin0);
    ok = true;
    if (psdiff <= 0) { ostat = normalise(0.0, tss, rn); iseg[0] = ti; iseg[1] = tj; return CANVAS_OK; }
    canvas_ctx* ctx = G.ctx;
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    int32_t rc = G.ensure(n); if (rc) return rc;
    double* hSx = (double*)G.pin; double* hMax = nullptr; int32_t* hFirst = nullptr; unsigned long long* hOut = (unsigned long long*)(G.pin + (size_t)G.pinCap * 8);
    // a host that makes ONE call and exits (canvas_set_one_shot: the CanvasPartition executable) uploads the prefix sums from the caller's pageable array: the pinned staging
    // buffers of fifty engines are ~100 MB that such a process pins at 1.4 ms per MB and unpins again when it leaves; a host that keeps its context (warm calls: 0.046 against
    // 0.056 s for the germline sample) keeps the staging buffers
    if (G.pageable()) { hSx = sx; hOut = (unsigned long long*)G.pinSmall; }
    else memcpy(hSx, sx, (size_t)n * 8);
    const size_t nbk = (size_t)G.cap / AP_BK + 2;
    ArcHostReq q; q.hSx = hSx; q.hMax = hMax; q.hFirst = hFirst; q.hOut = hOut;
    q.r.sx = G.dSx; q.r.n = n; q.r.dmax = G.dMax; q.r.firstI = G.dFirst;
    q.p.sx = G.dSx; q.p.n = n; q.p.al0 = al0; q.p.tau = bss0; q.p.bmin = (double*)G.dPr; q.p.bmax = q.p.bmin + nbk; q.p.pairs = (int*)(q.p.bmax + nbk);
    q.p.pairMax = (double*)(q.p.pairs + AP_PAIRCAP); q.p.out = (unsigned long long*)(q.p.pairMax + AP_PAIRCAP); q.p.bpos = (int*)(q.p.out + 8);
    q.pruned = cvx_hook("CANVAS_CBS_EXHAUSTIVE_ARCS") == nullptr;
    if (q.pruned) {
        rc = service_submit_arc(G.svc, q); if (rc) return rc;
        st.gpu_searches++;
        if (!hOut[4]) {
            st.gpu_pairs += (long long)hOut[3] * AP_BK * AP_BK;
            const double M = hOut[3] ? __builtin_bit_cast(double, hOut[0]) : -1.0;
            if (!(M > bss0)) { ostat = normalise(bss0, tss, rn); iseg[0] = ti; iseg[1] = tj; return CANVAS_OK; }   // the incumbent survives every strict '>' test
            if (hOut[1] == 1) {
                const int L = (int)(hOut[2] >> 32), i0 = (int)(hOut[2] & 0xffffffffull);
                if (ref_scans_arc(B, n, al0, i0 + 1, L)) { ostat = normalise(M, tss, rn); iseg[0] = i0 + 1; iseg[1] = i0 + 1 + L; return CANVAS_OK; }
                st.unscanned_max++;      // the best admissible arc is one the reference does not look at: its own (smaller) maximum comes from the host replay
            }
            ok = false;       // several arcs attain the maximum: the winner depends on the reference's block visiting order
            return CANVAS_OK;
        }
        q.pruned = false;     // too many candidate block pairs (flat data): exhaustive search
    }
    rc = G.ensure_exhaustive(); if (rc) return rc;
    hMax = (double*)G.pinEx; hFirst = (int32_t*)(hMax + G.cap);
    q.hMax = hMax; q.hFirst = hFirst; q.r.dmax = G.dMax; q.r.firstI = G.dFirst;
    rc = service_submit_arc(G.svc, q); if (rc) return rc;
    const double* dmax = hMax; const int32_t* firstI = hFirst;
    if (!cvx_hook("CANVAS_CBS_EXHAUSTIVE_ARCS")) st.gpu_searches--;
    st.gpu_searches++; st.gpu_pairs += (long long)n * (n - 1) / 2;
    // arcs of length L in [al0, n - al0] (CBSTStatistic.cs:139-151: alenlo >= al0, alenhi <= n - al0)
    double M = -1.0; int bestL = -1, nbest = 0;
    for (int L = al0; L <= n - al0; L++) {
        double rj = (double)L; double v = (rn / (rj * (rn - rj))) * sq(dmax[L]);
        if (v > M) { M = v; bestL = L; nbest = 1; } else if (v == M) nbest++;
    }
    if (!(M > bss0)) { ostat = normalise(bss0, tss, rn); iseg[0] = ti; iseg[1] = tj; return CANVAS_OK; }   // the incumbent survives every strict '>' test
    if (nbest == 1) {
        // is the maximising arc unique among all arcs of that length (including rounding plateaus of v)?
        double rj = (double)bestL, c = rn / (rj * (rn - rj)); int cnt = 0;
        for (int i = 0; i + bestL < n && cnt < 2; i++) { double d = std::fabs(sx[i + bestL] - sx[i]); if (c * sq(d) == M) cnt++; }
        if (cnt == 1 && ref_scans_arc(B, n, al0, firstI[bestL] + 1, bestL)) { ostat = normalise(M, tss, rn); iseg[0] = firstI[bestL] + 1; iseg[1] = firstI[bestL] + 1 + bestL; return CANVAS_OK; }
        if (cnt == 1) st.unscanned_max++;
    }
    ok = false;     // exact tie: the winner depends on the reference's block visiting order
    return CANVAS_OK;
}

// TPermP on the device (CANVAS_CBS_DEVICE_TPERMP=1).  The test is ONE chain: nPerm x min(n1, n2) swaps, each at a position drawn from the chromosome's
// Mersenne Twister (whose state every later permutation of the chromosome continues from) and each reading what the previous ones wrote — nothing in it is independent, so one
// lane of one wave walks it exactly as the reference does: the segment's copy in LDS when it fits (else in global scratch), the generator's 624 words in LDS, twisted in place.
// A lane needs ~0.15 us per swap where a host core needs 5 ns: the host chain stays the default, this kernel is the parity-tested drop-in for a host without spare cores.
#define TPERMP_LDS_MAX 4096              // (32 KB of LDS; longer segments walk a copy in global scratch)
__global__ void __launch_bounds__(64) k_tpermp(const double* __restrict__ gd, int n1, int n2, uint32_t nPerm, uint32_t* __restrict__ mtState /* 624 words + index, in and out */,
                                               double* __restrict__ scratch, int32_t* __restrict__ out /* nrej, swaps made (0: the shortcut) */) {
    __shared__ uint32_t mt[624];
    __shared__ double spx[TPERMP_LDS_MAX];
    const int n = n1 + n2;
    for (int i = threadIdx.x; i < 624; i += 64) mt[i] = mtState[i];
    __syncthreads();
    if (threadIdx.x != 0) return;
    int mti = (int)mtState[624];
    double* px = n <= TPERMP_LDS_MAX ? spx : scratch;
    const double rn1 = (double)n1, rn2 = (double)n2, rn = rn1 + rn2;
    int nrej = 0; long long swaps = 0;
    if (n1 == 1 || n2 == 1) nrej = (int)nPerm;
    else {
        double xs1 = 0.0, tss = 0.0;
        for (int i = 0; i < n1; i++) { px[i] = gd[i]; xs1 = xs1 + gd[i]; tss = tss + gd[i] * gd[i]; }
        double xs2 = 0.0;
        for (int i = n1; i < n; i++) { px[i] = gd[i]; xs2 = xs2 + gd[i]; tss = tss + gd[i] * gd[i]; }
        const double xbar = (xs1 + xs2) / rn; tss = tss - rn * (xbar * xbar);
        int m1; double rm1, ostat, tstat;
        if (n1 <= n2) { m1 = n1; rm1 = rn1; ostat = 0.99999 * fabs(xs1 / rn1 - xbar); tstat = (ostat * ostat) * rn1 * rn / rn2; }
        else { m1 = n2; rm1 = rn2; ostat = 0.99999 * fabs(xs2 / rn2 - xbar); tstat = (ostat * ostat) * rn2 * rn / rn1; }
        tstat = tstat / ((tss - tstat) / (rn - 2.0));
        if (!(tstat > 25 && m1 >= 10)) {
            for (uint32_t np = 0; np < nPerm; np++) {
                xs1 = 0;
                for (int i = n - 1; i >= n - m1; i--) {
                    if (mti >= 624) {                            // MT19937 twist (the generator of MersenneTwister.cs, as cbs::MT::u32)
                        int k; uint32_t y;
                        for (k = 0; k < 227; k++) { y = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu); mt[k] = mt[k + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u); }
                        for (; k < 623; k++) { y = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu); mt[k] = mt[k - 227] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u); }
                        y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu); mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
                        mti = 0;
                    }
                    uint32_t y = mt[mti++];
                    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
                    const double cc = (double)y * (1.0 / 4294967296.0);
                    int j = (int)(cc * (double)(i + 1)); j = j > i ? i : j;
                    const double a = px[i], b = px[j]; px[i] = b; px[j] = a;
                    xs1 = xs1 + px[i];
                }
                if (ostat <= fabs(xs1 / rm1 - xbar)) nrej++;
            }
            swaps = (long long)nPerm * m1;
        }
    }
    for (int i = 0; i < 624; i++) mtState[i] = mt[i];
    mtState[624] = (uint32_t)mti;
    out[0] = nrej; out[1] = swaps > 0 ? 1 : 0;
}
struct PermGpu;
static bool tpermp_device(PermGpu& PG, int n1, int n2, int n, const double* gd, int off, uint32_t nPerm, MT& rnd, Stats& st, double& p);
static bool tpermp_device_impl(canvas_ctx* ctx, hipStream_t stream, int n1, int n2, int n, const double* gd, int off, uint32_t nPerm, MT& rnd, Stats& st, double& p) {
    if (hipSetDevice(ctx->device) != hipSuccess) { (void)hipGetLastError(); return false; }      // (chromosome and helper threads: the context's device, its engine's own stream)
    char* d = nullptr;
    const size_t bytesX = (size_t)n * 8, offState = (bytesX + 255) & ~size_t(255), offScr = offState + 4096, offOut = offScr + ((bytesX + 255) & ~size_t(255));
    if (hipMalloc((void**)&d, offOut + 256) != hipSuccess) { (void)hipGetLastError(); return false; }
    uint32_t state[625]; rnd.get_state(state);
    int32_t out[2] = {0, 0};
    bool ok = hipMemcpyAsync(d, gd + off, bytesX, hipMemcpyHostToDevice, stream) == hipSuccess && hipMemcpyAsync(d + offState, state, sizeof state, hipMemcpyHostToDevice, stream) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(k_tpermp, dim3(1), dim3(64), 0, stream, (const double*)d, n1, n2, nPerm, (uint32_t*)(d + offState), (double*)(d + offScr), (int32_t*)(d + offOut));
        ok = hipMemcpyAsync(out, d + offOut, sizeof out, hipMemcpyDeviceToHost, stream) == hipSuccess && hipMemcpyAsync(state, d + offState, sizeof state, hipMemcpyDeviceToHost, stream) == hipSuccess
             && hipStreamSynchronize(stream) == hipSuccess;
    }
    (void)hipFree(d);
    if (!ok) { (void)hipGetLastError(); return false; }
    rnd.set_state(state);
    if (out[1]) { st.tpermp_draws += (long long)nPerm * std::min(n1, n2); rnd.drawn += (long long)nPerm * std::min(n1, n2); }
    st.tpermp_device++;
    p = (double)out[0] / nPerm;
    return true;
}
static double tpermp(PermGpu& PG, int n1, int n2, int n, const double* gd, int off, double* px, uint32_t nPerm, MT& rnd, Stats& st) {   // CBSTStatistic.cs:947-1024
    { static const bool onDevice = cvx_hook("CANVAS_CBS_DEVICE_TPERMP") != nullptr; double p; if (onDevice && tpermp_device(PG, n1, n2, n, gd, off, nPerm, rnd, st, p)) return p; }
    double rn1 = (double)n1, rn2 = (double)n2, rn = rn1 + rn2; int nrej;
    if (n1 == 1 || n2 == 1) nrej = (int)nPerm;
    else {
        double xs1 = 0.0, tss = 0.0;
        for (int i = 0; i < n1; i++) { px[i] = gd[off + i]; xs1 = xs1 + gd[off + i]; tss = tss + sq(gd[off + i]); }
        double xs2 = 0.0;
        for (int i = n1; i < n; i++) { px[i] = gd[off + i]; xs2 = xs2 + gd[off + i]; tss = tss + sq(gd[off + i]); }
        double xbar = (xs1 + xs2) / rn; tss = tss - rn * sq(xbar);
        int m1; double rm1, ostat, tstat;
        if (n1 <= n2) { m1 = n1; rm1 = rn1; ostat = 0.99999 * std::fabs(xs1 / rn1 - xbar); tstat = sq(ostat) * rn1 * rn / rn2; }
        else { m1 = n2; rm1 = rn2; ostat = 0.99999 * std::fabs(xs2 / rn2 - xbar); tstat = sq(ostat) * rn2 * rn / rn1; }
        nrej = 0; tstat = tstat / ((tss - tstat) / (rn - 2.0));
        if (!(tstat > 25 && m1 >= 10)) {
            for (uint32_t np = 0; np < nPerm; np++) {
                xs1 = 0;
                for (int i = n - 1; i >= n - m1; i--) { double cc = rnd.next_double(); int j = (int)(cc * (i + 1)); j = j > i ? i : j; std::swap(px[i], px[j]); xs1 = xs1 + px[i]; }
                if (ostat <= std::fabs(xs1 / rm1 - xbar)) nrej++;
            }
            st.tpermp_draws += (long long)nPerm * m1;
        }
    }
    return (double)nrej / nPerm;
}
static void xperm(const double* x, double* px, int n, MT& rnd) {                 // ChangePoint.cs:407-421
    for (int i = 0; i < n; i++) px[i] = x[i];
    for (int i = n - 1; i >= 0; i--) { double cc = rnd.next_double(); int j = (int)(cc * (i + 1)); j = j > i ? i : j; std::swap(px[i], px[j]); }
}

// the plan of k_perm_rp for a segment of n bins: ranges, tiles, the capacity of every range's inbox and the layout of a workgroup's scratch (in 4-byte words)
#define PERM_RP_MIN_N 1024
#define PERM_RP_MAX_N (RP_MAXK * RP_R)
#define PERM_RP_GRID 512             // persistent workgroups per request (two per CU are resident)
static void rp_plan(int n, PermReq::RpPlan& P) {
    const int K = (n + RP_R - 1) / RP_R, nT = (n + RP_BK - 1) / RP_BK;
    P.K = K; P.nT = nT;
    uint32_t ent = 0;
    for (int d = 0; d < RP_MAXK; d++) { P.inOff[d] = 0; P.inCap[d] = 0; }
    for (int d = 0; d + 1 < K; d++) {
        // messages into range d: one per step i >= (d + 1) R whose target falls into it, probability R / (i + 1): at most R ln(n / ((d + 1) R)) expected, variance below the mean
        const double E = (double)RP_R * std::log((double)n / (double)((d + 1) * RP_R));
        uint32_t cap = (uint32_t)(E + 8.0 * std::sqrt(E + 1.0) + 64.0); cap = (cap + 1u) & ~1u;
        P.inOff[d] = ent; P.inCap[d] = cap; ent += cap;
    }
    uint32_t o = 0;
    P.oEndsIn = o; o += (uint32_t)((K > 1 ? K - 1 : 0) * (nT + 1));
    P.oEndsOwn = o; o += (uint32_t)(nT + 1);
    o = (o + 1u) & ~1u; P.oInbox = o; o += 2u * ent;
    P.oOutIn = o; o += ent;
    P.oOutOwn = o; o += (uint32_t)n;
    P.stride = (o + 63u) & ~63u;
}
static size_t rp_scratch_bytes(int n, int wgs) { PermReq::RpPlan P; rp_plan(n, P); return (size_t)P.stride * 4 * (size_t)wgs; }

// ================================================================================================ the cache's own generator: jump-ahead check points + the plain recurrence
// A cached stream is CONSUMED in sequence but must be PRODUCED in parallel: one workgroup running the plain recurrence x[n] = x[n - 227] ^ twist(x[n - 624], x[n - 623])
// makes ~4 G words/s (454 words per barrier), the strided GF(2) form above makes 53 G words/s per stream with 134 LDS reads per word — a third of the kernel time of a cold
// tumour / normal call.  So a stream is cut into chunks of MT_CHUNK = 2^22 outputs; the generator state a chunk starts from (the 624 untempered outputs in front of it) is
// obtained from the state of the chunk before it by the jump-ahead polynomial g = x^(2^22) mod phi (cbs_mt_jump.hpp, tools/gen_mt_jump.py: MT19937 is linear over GF(2), so
// u[k + 2^22] = XOR_{i : g_i} u[k + i] for the untempered outputs u of any seed) — k_mt_jump: one workgroup per stream walks its chunks, generating the 19 937 words behind a
// state into LDS and XOR-summing 10 024 of them per state word (~0.1 ms per chunk: LDS bandwidth) — and then ALL chunks of ALL streams are generated side by side by k_mt_chunk
// with the plain recurrence (3 LDS reads per word, stores of 227 consecutive words), which is bound by the HBM writes, not by the generator.
#define MT_CHUNK (1LL << MT_JUMP_LOG2)
#define MTJ_T 640
#define MTJ_W (624 + 19937)
struct MtJumpJob { uint32_t* cp; int kFrom, kTo; };      // check points cp[k] (624 words each): cp[kFrom] exists, cp[kFrom + 1 .. kTo] are computed
struct MtChunkJob { const uint32_t* cp; uint32_t* out; long long words; };
__device__ __forceinline__ uint32_t mt_twist(uint32_t a, uint32_t b, uint32_t c) { const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu); return c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u); }
#define MTJ_NIDX ((MT_JUMP_NTERMS + 7) & ~7)
__global__ void __launch_bounds__(MTJ_T) k_mt_jump(const MtJumpJob* __restrict__ jobs) {
    __shared__ uint32_t W[MTJ_W + 3];
    __shared__ __attribute__((aligned(16))) unsigned short sIdx[MTJ_NIDX];      // the polynomial's terms in LDS: read eight at a time as ONE broadcast 16-byte read (from constant memory the
                                                                                 // scalar loads missed their cache all the way: 0.5 ms per jump instead of 0.1)
    const MtJumpJob J = jobs[blockIdx.x];
    const int t = (int)threadIdx.x;
    for (int i = t; i < MTJ_NIDX; i += MTJ_T) sIdx[i] = i < MT_JUMP_NTERMS ? MT_JUMP_IDX[i] : (unsigned short)0xFFFF;
    if (t < 624) W[t] = J.cp[(size_t)J.kFrom * 624 + t];
    __syncthreads();
    for (int k = J.kFrom; k < J.kTo; k++) {
        // the 19 937 words behind the state, 227 at a time (each needs words at least 227 behind it)
        for (int n = 624; n < MTJ_W; n += 227) {
            if (t < 227 && n + t < MTJ_W) W[n + t] = mt_twist(W[n + t - 624], W[n + t - 623], W[n + t - 227]);
            __syncthreads();
        }
        uint32_t acc = 0;
        if (t < 624) {
            const uint32_t* __restrict__ w = W + t;
            for (int i = 0; i < MTJ_NIDX - 8; i += 8) {           // (the LDS reads of a wave are 64 consecutive words: no bank conflict)
                const uint4 pk = *reinterpret_cast<const uint4*>(&sIdx[i]);
                acc ^= w[pk.x & 0xFFFFu] ^ w[pk.x >> 16] ^ w[pk.y & 0xFFFFu] ^ w[pk.y >> 16] ^ w[pk.z & 0xFFFFu] ^ w[pk.z >> 16] ^ w[pk.w & 0xFFFFu] ^ w[pk.w >> 16];
            }
            for (int i = MTJ_NIDX - 8; i < MT_JUMP_NTERMS; i++) acc ^= w[sIdx[i]];
        }
        __syncthreads();
        if (t < 624) { W[t] = acc; J.cp[(size_t)(k + 1) * 624 + t] = acc; }
        __syncthreads();
    }
}
// one workgroup per chunk: the state in an LDS ring, two blocks of 227 words per barrier (the second block of a lane continues its own first word and otherwise reads words
// that are at least 396 behind: nothing another lane produced in this round)
__global__ void __launch_bounds__(256) k_mt_chunk(const MtChunkJob* __restrict__ jobs) {
    __shared__ uint32_t R[2048];
    const MtChunkJob J = jobs[blockIdx.x];
    const int t = (int)threadIdx.x;
    for (int i = t; i < 624; i += 256) R[i] = J.cp[i];
    __syncthreads();
    const gptr<uint32_t> out = as_global(J.out);
    const long long total = J.words;
    // a round makes the 623 words [n, n + 623): lane t its word n + t, then n + t + 227 (continuing its own first word; the other two operands lie 397 / 396 behind), and — the
    // lanes below 169 — n + t + 454 (operands 170 / 169 behind the round's start: old).  Every operand that is not the lane's own is older than the round.
    for (long long n = 624; n - 624 < total; n += 623) {
        if (t < 227) {
            const long long m = n + t;
            const uint32_t a0 = R[(m - 624) & 2047], b0 = R[(m - 623) & 2047], c0 = R[(m - 227) & 2047], a1 = R[(m - 397) & 2047], b1 = R[(m - 396) & 2047];
            uint32_t a2 = 0, b2 = 0; if (t < 169) { a2 = R[(m - 170) & 2047]; b2 = R[(m - 169) & 2047]; }
            const uint32_t v0 = mt_twist(a0, b0, c0), v1 = mt_twist(a1, b1, v0);
            R[m & 2047] = v0; R[(m + 227) & 2047] = v1;
            if (m - 624 < total) out[m - 624] = mt_temper(v0);
            if (m - 397 < total) out[m - 397] = mt_temper(v1);
            if (t < 169) { const uint32_t v2 = mt_twist(a2, b2, v1); R[(m + 454) & 2047] = v2; if (m - 170 < total) out[m - 170] = mt_temper(v2); }
        }
        // the round's LDS writes must be visible to the next round's reads; the global stores need not have completed (__syncthreads() would wait for them: vmcnt(0) every 623 words)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}
// ================================================================================================ the chromosomes' draw streams, kept in HBM
// The k-th chromosome's generator is MersenneTwister(seed_k) with seed_k drawn from MersenneTwister(0) in file order (CBSRunner.cs:107-112), and XPerm / TPermP consume it
// strictly in sequence (ChangePoint.cs:407-421, CBSTStatistic.cs:1009): HOW FAR a call reads depends on the data, the words do not.  So the tempered outputs of every
// stream are kept per context in device memory — generated once, by the kernels above, in extensions of tens of millions of words that run AHEAD of the permutation loops on a
// stream of their own — and a permutation batch is handed a pointer into them.  What that removes from every batch of every later call (and from all but the first reader of
// a position inside a call): the generator launches (k_mt_draws, up to seven bootstrap launches, k_mt_classes: a third of the device time of the tumour / normal flow),
// k_mt_snapshots and the 2.5 MB of generator states it sent to the host per batch — the host now keeps a POSITION and fetches 624 words only when host code draws (edge
// tests, re-evaluations).  A stream is one virtual address range (hipMemAddressReserve) into which physical memory is mapped 64 MB at a time as it grows, so positions stay
// plain offsets and nothing is ever copied; the whole cache is bounded (CANVAS_CBS_CACHE_GB, default 30 % of the device's memory: the 288 GB of an MI355X are there to be
// used — a 60x germline sample reads 1.6 GB of draws, an 80x / 40x tumour-normal pair 46 GB), and a stream that reaches the bound serves what it holds and hands the rest of
// the loop to the in-batch generator (the path of rounds 1-5, kept as the fallback).
#define MTS_GRANULE (size_t(64) << 20)          // physical memory is mapped into a stream in pieces of this size (16 M draws)
#define MTS_VA_BYTES (size_t(16) << 30)         // address range of a stream: 4 G draws (the longest chromosome of the 80x / 40x tumour-normal pair reads 2 G; a stream that
                                                // reaches the end of its range serves what it holds).  (64 GB each: the kernel took 0.3 s longer to take a process with 25 of them apart)
#define MTS_FIRST_WORDS (8LL << 20)             // the first extension of a stream (it must leave MT_HISTORY words behind for the strided generator to continue from)
#define MTS_JOB_MAX_WORDS (48LL << 20)          // ... and the longest one: a round of the producer serves every stream that is behind and publishes when its LONGEST job is done —
                                                // short rounds (~1 ms) keep a stream that needs little from waiting behind one that needs much
struct MtStream {
    uint32_t seed = 0; char* va = nullptr; size_t mappedBytes = 0; bool plain = false; size_t plainBytes = 0; std::vector<hipMemGenericAllocationHandle_t> handles;
    long long ready = 0, requested = 0; bool full = false;      // words that are valid / that the producer has been asked for; full: no more memory will be mapped
    long long floorWords = 0;                                   // what canvas_cbs_prefetch / the start of a call asked for: the end of a call does not take THAT back
    uint32_t* cp = nullptr; int ncp = 0;                        // the generator states of the stream's chunks (624 untempered words in front of chunk k; cp[0] = init_genrand(seed)), how many exist
    uint32_t* d() const { return (uint32_t*)va; }
};
struct MtStreamCache {
    canvas_ctx* ctx; std::mutex mu; std::condition_variable cvWork, cvReady; std::map<uint32_t, std::unique_ptr<MtStream>> streams;
    size_t capBytes = 0, usedBytes = 0; bool useVmm = true, stop = false, failed = false, started = false, off = false; std::string err;
    std::thread th; hipStream_t stream = nullptr; PermReq* dReqs = nullptr; PermReq* hReqs = nullptr;
    MtJumpJob* dJump = nullptr; MtJumpJob* hJump = nullptr; MtChunkJob* dChunk = nullptr; MtChunkJob* hChunk = nullptr;      // job tables of the cache's own generator (pinned + device)
    static constexpr int CHUNK_CAP = 2048;
    std::atomic<long long> servedWords{0}, generatedWords{0}, fallbackWords{0}, fetches{0};
    std::atomic<long long> nsMap{0}, nsGen{0}, rounds{0}, nsWaited{0}, waits{0};      // producer: mapping memory, generating (launch to synchronisation), rounds; consumers: time spent waiting in acquire()
    static constexpr int CAP = 32;
    explicit MtStreamCache(canvas_ctx* c) : ctx(c) {
        // CANVAS_CBS_CACHE_GB: the bound of the whole cache in GB (0 switches it off); a configuration switch, read without CANVAS_TEST_HOOKS
        const char* e = getenv("CANVAS_CBS_CACHE_GB");
        size_t freeB = 0, totB = 0;
        if (hipSetDevice(ctx->device) != hipSuccess || hipMemGetInfo(&freeB, &totB) != hipSuccess) { (void)hipGetLastError(); off = true; return; }
        double gb = e ? atof(e) : 0.30 * (double)totB / 1e9;
        if (!(gb > 0)) { off = true; return; }
        capBytes = (size_t)(gb * 1e9);
        int vmm = 0; if (hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, ctx->device) != hipSuccess || !vmm) { (void)hipGetLastError(); useVmm = false; }
        if (cvx_hook("CANVAS_CBS_CACHE_NO_VMM")) useVmm = false;      // test hook: the fixed-allotment form
    }
    ~MtStreamCache() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; } cvWork.notify_all(); cvReady.notify_all();
        if (th.joinable()) th.join();
        (void)hipSetDevice(ctx->device);
        if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
        if (dReqs) (void)hipFree(dReqs); if (hReqs) (void)hipHostFree(hReqs);
        if (dJump) (void)hipFree(dJump); if (hJump) (void)hipHostFree(hJump); if (dChunk) (void)hipFree(dChunk); if (hChunk) (void)hipHostFree(hChunk);
        for (auto& kv : streams) {
            MtStream& S = *kv.second;
            if (S.cp) (void)hipFree(S.cp);
            if (S.plain) { if (S.va) (void)hipFree(S.va); continue; }
            if (S.va && S.mappedBytes) (void)hipMemUnmap(S.va, S.mappedBytes);
            for (auto h : S.handles) (void)hipMemRelease(h);
            if (S.va) (void)hipMemAddressFree(S.va, MTS_VA_BYTES);
        }
    }
    // the stream of a seed (created on first use; nullptr: the cache is off or could not reserve an address range)
    MtStream* get(uint32_t seed) {
        if (off) return nullptr;
        std::lock_guard<std::mutex> lk(mu);
        if (failed) return nullptr;
        auto it = streams.find(seed);
        if (it != streams.end()) return it->second.get();
        std::unique_ptr<MtStream> S(new MtStream()); S->seed = seed;
        if (hipSetDevice(ctx->device) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (useVmm) {
            void* base = nullptr;
            if (hipMemAddressReserve(&base, MTS_VA_BYTES, 0, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); useVmm = false; }
            else S->va = (char*)base;
        }
        if (!useVmm) {          // no virtual memory management: one fixed allotment per stream (1 / 32 of the bound), served until it is full
            S->plain = true; S->plainBytes = std::max<size_t>(MTS_GRANULE, std::min<size_t>(size_t(512) << 20, capBytes / 32) & ~(MTS_GRANULE - 1));      // (at most 512 MB each: allocating 25 x 2.9 GB took 2 s)
            if (usedBytes + S->plainBytes > capBytes || hipMalloc((void**)&S->va, S->plainBytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
            S->mappedBytes = S->plainBytes; usedBytes += S->plainBytes;
        }
        MtStream* r = S.get(); streams[seed] = std::move(S);
        return r;
    }
    // ask the producer for [0, upto) without waiting
    void prefetch(MtStream* S, long long upto) {
        if (!S) return;
        { std::lock_guard<std::mutex> lk(mu);
          upto = std::max(upto, MTS_FIRST_WORDS);
          S->floorWords = std::max(S->floorWords, upto);
          if (S->full || failed || upto <= S->requested) return;
          S->requested = upto; ensure_thread(); }
        cvWork.notify_one();
    }
    // [0, end) must be valid before the caller's kernels read it; ahead: how far beyond `end` the producer should go on.  false: the cache cannot serve `end` (bound reached,
    // no memory, a HIP error): the caller generates the batch itself
    bool acquire(MtStream* S, long long end, long long ahead) {
        if (!S) return false;
        std::unique_lock<std::mutex> lk(mu);
        if (S->ready >= end) {
            if (!S->full && !failed && S->ready < end + ahead / 2 && end + ahead > S->requested) { S->requested = end + ahead; ensure_thread(); lk.unlock(); cvWork.notify_one(); }
            return true;
        }
        if (S->full || failed) return false;
        if (end + ahead > S->requested) { S->requested = std::max(end + ahead, MTS_FIRST_WORDS); ensure_thread(); cvWork.notify_one(); }
        const auto tW = std::chrono::steady_clock::now();
        cvReady.wait(lk, [&]() { return stop || failed || S->full || S->ready >= end; });
        nsWaited += (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tW).count(); waits++;
        return S->ready >= end;
    }
    void ensure_thread() { if (!started) { started = true; th = std::thread([this]() { run(); }); } }      // (mu held)
    // the end of a call: what was asked for ahead of the permutation loops and has not been generated yet is not needed any more, and nothing of the cache's is left running
    // on the device when the call returns (a one-shot process that exits with a generator launch in flight pays for it in the driver's teardown: 0.1-0.3 s)
    bool busy = false;
    void quiesce() {
        std::unique_lock<std::mutex> lk(mu);
        for (auto& kv : streams) if (kv.second->requested > kv.second->ready) kv.second->requested = std::max(kv.second->ready, std::min(kv.second->requested, kv.second->floorWords));
        cvReady.wait(lk, [&]() { return !busy || stop || failed; });
    }
    // physical memory behind [0, words) of a stream; returns the words that are backed
    long long back(MtStream& S, long long words) {
        const size_t need = ((size_t)words * 4 + MTS_GRANULE - 1) & ~(MTS_GRANULE - 1);
        if (S.plain) return (long long)(S.mappedBytes / 4);
        while (S.mappedBytes < need && S.mappedBytes + MTS_GRANULE <= MTS_VA_BYTES) {
            { std::lock_guard<std::mutex> lk(mu); if (usedBytes + MTS_GRANULE > capBytes) break; usedBytes += MTS_GRANULE; }
            size_t freeB = 0, totB = 0;
            hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = ctx->device;
            hipMemAccessDesc ad = {}; ad.location.type = hipMemLocationTypeDevice; ad.location.id = ctx->device; ad.flags = hipMemAccessFlagsProtReadWrite;
            hipMemGenericAllocationHandle_t h;
            bool ok = hipMemGetInfo(&freeB, &totB) == hipSuccess && freeB > MTS_GRANULE + (size_t(4) << 30)      // (never the device's last 4 GB: the caller's own arrays come first)
                      && hipMemCreate(&h, MTS_GRANULE, &prop, 0) == hipSuccess;
            if (ok && hipMemMap(S.va + S.mappedBytes, MTS_GRANULE, 0, h, 0) != hipSuccess) { (void)hipMemRelease(h); ok = false; }
            if (ok && hipMemSetAccess(S.va + S.mappedBytes, MTS_GRANULE, &ad, 1) != hipSuccess) { (void)hipMemUnmap(S.va + S.mappedBytes, MTS_GRANULE); (void)hipMemRelease(h); ok = false; }
            if (!ok) { (void)hipGetLastError(); std::lock_guard<std::mutex> lk(mu); usedBytes -= MTS_GRANULE; break; }
            S.handles.push_back(h); S.mappedBytes += MTS_GRANULE;
        }
        return (long long)(S.mappedBytes / 4);
    }
    void run() {
        struct Job { MtStream* S; long long from, to; };
        // Which generator extends the streams.  Default: the strided one of rounds 3-5 (k_mt_draws + bootstrap + k_mt_classes).  CANVAS_CBS_CACHE_JUMP_GENERATOR=1: jump-ahead check
        // points + the plain recurrence (k_mt_jump / k_mt_chunk above) — measured in round 6 on the tumour / normal pair: 0.23 s of kernel time per cold call instead of 0.54 s
        // (profiles/r06_generator_ab.txt), and NO gain in wall time (first call of the flow 0.75-2.1 s against 0.98-1.09 s on the same box, germline first call 0.14 against
        // 0.10-0.125 s): the cold call waits for allocations and for k_perm_rp, whose persistent workgroups hold 159 of every CU's 160 KB of LDS — k_mt_jump's 100 KB window waits
        // for a CU exactly as k_mt_classes' 160 KB ring does — and the batches wait for the producer for ~50 ms per chromosome thread either way.  Both are tested (contents of
        // the cache against the oracle's generator across every seam, CBS parity); the default is the one the six-minute soaks of five rounds have run on.
        const bool oldGen = cvx_hook("CANVAS_CBS_CACHE_JUMP_GENERATOR") == nullptr;
        if (hipSetDevice(ctx->device) != hipSuccess || hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess || hipMalloc((void**)&dReqs, CAP * sizeof(PermReq)) != hipSuccess
            || hipHostMalloc((void**)&hReqs, CAP * sizeof(PermReq), hipHostMallocDefault) != hipSuccess
            || hipMalloc((void**)&dJump, CAP * sizeof(MtJumpJob)) != hipSuccess || hipHostMalloc((void**)&hJump, CAP * sizeof(MtJumpJob), hipHostMallocDefault) != hipSuccess
            || hipMalloc((void**)&dChunk, CHUNK_CAP * sizeof(MtChunkJob)) != hipSuccess || hipHostMalloc((void**)&hChunk, CHUNK_CAP * sizeof(MtChunkJob), hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError(); std::lock_guard<std::mutex> lk(mu); failed = true; err = "the draw-stream cache could not create its stream / request tables"; cvReady.notify_all(); return;
        }
        for (;;) {
            std::vector<Job> jobs;
            { std::unique_lock<std::mutex> lk(mu);
              cvWork.wait(lk, [&]() { if (stop) return true; for (auto& kv : streams) if (!kv.second->full && kv.second->requested > kv.second->ready) return true; return false; });
              if (stop) return;
              for (auto& kv : streams) { MtStream& S = *kv.second; if (!S.full && S.requested > S.ready && (int)jobs.size() < CAP) jobs.push_back({&S, S.ready, std::min(S.requested, S.ready + MTS_JOB_MAX_WORDS)}); }
              busy = true; }
            std::vector<char> shortJob(jobs.size(), 0);
            const auto tM = std::chrono::steady_clock::now();
            for (size_t i = 0; i < jobs.size(); i++) {
                Job& j = jobs[i];
                if (!oldGen) j.to = (j.to + MT_CHUNK - 1) / MT_CHUNK * MT_CHUNK;                       // whole chunks
                long long have = back(*j.S, j.to);
                if (!oldGen) have = have / MT_CHUNK * MT_CHUNK;
                if (have < j.to) { j.to = have; shortJob[i] = 1; }
                if (j.to <= j.from || (j.from == 0 && j.to < MTS_FIRST_WORDS)) j.to = j.from;
            }
            bool ok = true;
            const auto tG = std::chrono::steady_clock::now();
            nsMap += (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(tG - tM).count(); rounds++;
            if (oldGen) {
                int R = 0; bool anyFresh = false; long long maxFresh = 0, maxTotal = 0;
                for (Job& j : jobs) {
                    if (j.to <= j.from) continue;
                    PermReq& q = hReqs[R++];
                    memset((void*)&q, 0, sizeof q);
                    { MT m(j.S->seed); m.get_state(q.state); }
                    q.total = j.to - j.from; q.P.draws = j.S->d() + j.from; q.fy = 0; q.cached = 0;
                    q.cont = j.from > 0 ? 1 : 0; q.hist = q.cont ? j.S->d() + (j.from - MT_HISTORY) : nullptr;
                    if (!q.cont) { anyFresh = true; maxFresh = std::max(maxFresh, q.total); }
                    maxTotal = std::max(maxTotal, q.total + (q.cont ? MT_HISTORY : 0));
                }
                if (R > 0) {
                    ok = hipMemcpyAsync(dReqs, hReqs, (size_t)R * sizeof(PermReq), hipMemcpyHostToDevice, stream) == hipSuccess;
                    if (ok && anyFresh) {
                        hipLaunchKernelGGL(k_mt_draws, dim3(R), dim3(256), 0, stream, dReqs, 1);
                        for (int sd = 1; sd < MT_STRIDE && 19937LL * sd < maxFresh; sd <<= 1) hipLaunchKernelGGL(k_mt_classes, dim3(sd, R), dim3(MTC_T), 0, stream, dReqs, sd, 1);
                    }
                    if (ok && maxTotal > MT_HISTORY) hipLaunchKernelGGL(k_mt_classes, dim3(MT_STRIDE, R), dim3(MTC_T), 0, stream, dReqs, MT_STRIDE, 0);
                    ok = ok && hipStreamSynchronize(stream) == hipSuccess && hipGetLastError() == hipSuccess;
                }
            } else {
                // ---- check points by jump-ahead (one workgroup per stream walks its new chunks), then every chunk of every stream side by side
                int nJ = 0, nC = 0;
                std::vector<std::vector<uint32_t>> seeds0;                                         // (initial states on their way to the device: alive until the round has been waited for)
                const int cpCap = (int)(MTS_VA_BYTES / 4 / MT_CHUNK) + 1;
                for (Job& j : jobs) {
                    if (j.to <= j.from || !ok) continue;
                    MtStream& S = *j.S;
                    if (!S.cp) { if (hipMalloc((void**)&S.cp, (size_t)cpCap * 624 * 4) != hipSuccess) { ok = false; break; } }
                    if (S.ncp == 0) {
                        MT m(S.seed); seeds0.emplace_back(m.mt, m.mt + 624);
                        ok = hipMemcpyAsync(S.cp, seeds0.back().data(), 624 * 4, hipMemcpyHostToDevice, stream) == hipSuccess; S.ncp = 1;
                    }
                    const int c0 = (int)(j.from / MT_CHUNK), c1 = (int)(j.to / MT_CHUNK);              // chunks [c0, c1): their states cp[c0 .. c1 - 1]
                    if (c1 > cpCap || nC + (c1 - c0) > CHUNK_CAP) { j.to = j.from; continue; }           // (cannot happen: the address range holds cpCap - 1 chunks; a round holds at most CAP x 12)
                    if (c1 - 1 > S.ncp - 1) { hJump[nJ++] = MtJumpJob{S.cp, S.ncp - 1, c1 - 1}; S.ncp = c1; }
                    for (int c = c0; c < c1; c++) hChunk[nC++] = MtChunkJob{S.cp + (size_t)c * 624, S.d() + (long long)c * MT_CHUNK, MT_CHUNK};
                }
                if (ok && nJ > 0) {
                    ok = hipMemcpyAsync(dJump, hJump, (size_t)nJ * sizeof(MtJumpJob), hipMemcpyHostToDevice, stream) == hipSuccess;
                    if (ok) hipLaunchKernelGGL(k_mt_jump, dim3(nJ), dim3(MTJ_T), 0, stream, dJump);
                }
                if (ok && nC > 0) {
                    ok = hipMemcpyAsync(dChunk, hChunk, (size_t)nC * sizeof(MtChunkJob), hipMemcpyHostToDevice, stream) == hipSuccess;
                    if (ok) hipLaunchKernelGGL(k_mt_chunk, dim3(nC), dim3(256), 0, stream, dChunk);
                }
                ok = ok && hipStreamSynchronize(stream) == hipSuccess && hipGetLastError() == hipSuccess;
            }
            nsGen += (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tG).count();
            { std::lock_guard<std::mutex> lk(mu);
              if (!ok) { (void)hipGetLastError(); failed = true; err = "the draw-stream cache's generator failed"; }
              else for (size_t i = 0; i < jobs.size(); i++) { MtStream& S = *jobs[i].S; if (jobs[i].to > S.ready) { generatedWords += jobs[i].to - S.ready; S.ready = jobs[i].to; } if (shortJob[i]) S.full = true; }
              busy = false; }
            cvReady.notify_all();
            if (!ok) return;
        }
    }
};

// ---- device permutation engine, one instance per chromosome thread (own stream and buffers)
#define PERM_GPU_MIN_N 201           // every hybrid segment (> 200 bins) takes the device engine (round 2 kept those below 1024 bins on the host: 12 % slower on the device then; with this round's kernels 0.19 vs 0.23 s on the 4.7 M-bin probe); CANVAS_CBS_PERM_GPU_MIN_N overrides
#define PERM_TARGET_ELEMS (64 << 20) // permuted elements per batch (44 B of workspace each)
#define PERM_FY_MIN_N 16384          // segments from this length on take k_perm_fy (block-wise simulation of the swaps); shorter ones k_perm_stat (CANVAS_CBS_FY_MIN_N overrides: test hook)
// (batches of up to 2048 permutations for loops that run long were tried: k_perm_stat then takes 26 ms instead of 3.5 ms for 256 — the same rate per permutation — so
//  a larger batch only saves launcher round trips, 0.45 -> 0.44 s on the 4.7 M-bin sample, for 8.4 GB of workspace per engine)
struct PermService;
struct PermGpu {
    canvas_ctx* ctx = nullptr; PermService* svc = nullptr; hipStream_t stream = nullptr; char* buf = nullptr; size_t bytes = 0; char* pin = nullptr; size_t pinBytes = 0;
    size_t reserveElems = 0, reserveN = 0;     // the call's longest chromosome: the first allocation is made for it (growing means hipFree + hipMalloc, which stall every stream of the device)
    size_t reserveBytes = 0, reservePin = 0;   // ... in bytes of device / pinned memory (perm_reserve_bytes)
    // analytic tail probability on the device (k_tail_nu): own stream, 3 x 128 values on the device and in pinned memory
    hipStream_t tailStream = nullptr; char* tailDev = nullptr; char* tailPin = nullptr; EngineCache* cache = nullptr; unsigned tailSeq = 0;      // tailDev: the completion counter of k_tail_nu; tailPin: its mailbox
    unsigned mailSeq = 0;         // sequence numbers of the permutation batches' result mailboxes
    bool tailOwned = true;        // false: the stream is one of the cache's shared ones (several engines enqueue their short tail kernels on it; each waits for the stream,
                                  // i.e. at worst for a few other 10 us kernels) and the buffers are slices of its arena
    int32_t ensure_tail() {
        if (tailStream) return CANVAS_OK;
        AllocClock ac(g_ns_alloc_tail);
        CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
        hipStream_t sh = cache_tail_stream(cache); char* d = sh ? cache_take(cache, false, 128 * 24) : nullptr; char* p = d ? cache_take(cache, true, 128 * 24) : nullptr;
        if (sh && d && p) { tailStream = sh; tailDev = d; tailPin = p; tailOwned = false; }
        else {
            tailOwned = true;
            CANVAS_HIP_TRY(ctx, hipStreamCreateWithFlags(&tailStream, hipStreamNonBlocking));
            CANVAS_HIP_TRY(ctx, hipMalloc((void**)&tailDev, 128 * 24)); CANVAS_HIP_TRY(ctx, hipHostMalloc((void**)&tailPin, 128 * 24, hipHostMallocDefault));
        }
        CANVAS_HIP_TRY(ctx, hipMemsetAsync(tailDev, 0, 256, tailStream));      // the completion counter (every call leaves it at zero again)
        CANVAS_HIP_TRY(ctx, hipStreamSynchronize(tailStream));
        return CANVAS_OK;
    }
    // need: what the reservation asks for (the call's longest chromosome); minNeed: what this request cannot do without — taken when the reservation does not fit the device
    int32_t ensure(size_t need, size_t needPin, size_t minNeed = 0) {
        CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
        if (need <= bytes && needPin <= pinBytes) return CANVAS_OK;
        AllocClock ac(g_ns_alloc_perm);
        if (need > bytes) {
            if (buf) CANVAS_HIP_TRY(ctx, hipFree(buf));
            buf = nullptr; bytes = 0;
            hipError_t e = hipMalloc((void**)&buf, need);
            if (e != hipSuccess && minNeed && minNeed < need) { (void)hipGetLastError(); buf = nullptr; need = minNeed; e = hipMalloc((void**)&buf, need); }
            if (e != hipSuccess) { buf = nullptr; ctx->err = std::string("hipMalloc of the permutation workspace: ") + hipGetErrorString(e); return CANVAS_ERR_HIP; }
            bytes = need;
        }
        if (needPin > pinBytes) { if (pin) CANVAS_HIP_TRY(ctx, hipHostFree(pin)); pin = nullptr; pinBytes = 0; CANVAS_HIP_TRY(ctx, hipHostMalloc((void**)&pin, needPin, hipHostMallocDefault)); pinBytes = needPin; }
        return CANVAS_OK;
    }
    ~PermGpu() { if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); } if (buf) (void)hipFree(buf); if (pin) (void)hipHostFree(pin);
                 if (tailStream && tailOwned) { (void)hipStreamSynchronize(tailStream); (void)hipStreamDestroy(tailStream); } if (tailDev && tailOwned) (void)hipFree(tailDev); if (tailPin && tailOwned) (void)hipHostFree(tailPin); }
};
static bool tpermp_device(PermGpu& PG, int n1, int n2, int n, const double* gd, int off, uint32_t nPerm, MT& rnd, Stats& st, double& p) {
    if (PG.ensure_tail() != CANVAS_OK) return false;
    return tpermp_device_impl(PG.ctx, PG.tailStream, n1, n2, n, gd, off, nPerm, rnd, st, p);
}
// The chromosome's generator as the recursion sees it: a POSITION in the chromosome's stream plus, while host code draws from it (edge tests, re-evaluations, the host's own
// permutation loops), a real MT19937 at that position.  Device batches served from the stream cache only move the position (jump_to); the state is rebuilt from the 624
// outputs in front of the position when the host next needs it (host(): one 2.5 KB copy out of the cache).  Without a cache stream (CANVAS_CBS_CACHE_GB=0, no memory) the
// generator is always real and the batches hand their states over as in rounds 1-5.
struct Rng {
    MT mt{0u}; uint32_t seed = 0; MtStreamCache* cache = nullptr; MtStream* S = nullptr; long long pos = 0, base = 0; bool valid = true;
    void init(uint32_t sd, MtStreamCache* c) { seed = sd; mt = MT(sd); base = 0; pos = 0; valid = true; cache = c; S = c ? c->get(sd) : nullptr; }
    long long position() const { return valid ? base + mt.drawn : pos; }
    void jump_to(long long p) { pos = p; valid = false; }                                                      // consumed up to p on the device
    void adopt(const uint32_t* state625, long long p) { mt.set_state(state625); mt.drawn = 0; base = p; valid = true; }
    // a generator at stream position p (p <= everything the cache holds, or reachable from the current state): into `out`
    int32_t at(PermGpu& PG, long long p, MT& out) {
        if (valid && p >= position() && p - position() <= 4096) { out = mt; while (base + out.drawn < p) (void)out.u32(); return CANVAS_OK; }
        if (p < 624 || !S) { out = MT(seed); for (long long t = 0; t < p; t++) (void)out.u32(); return CANVAS_OK; }      // (no stream: only reached with small p)
        canvas_ctx* ctx = PG.ctx;
        int32_t rc = PG.ensure_tail(); if (rc) return rc;
        uint32_t* h = (uint32_t*)PG.tailPin;                                                                    // 128 * 24 = 3072 bytes: 624 words fit
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(h, S->d() + (p - 624), 624 * 4, hipMemcpyDeviceToHost, PG.tailStream));
        CANVAS_HIP_TRY(ctx, hipStreamSynchronize(PG.tailStream));
        uint32_t st[625];
        for (int i = 0; i < 624; i++) { uint32_t y = h[i]; y ^= y >> 18; y ^= (y << 15) & 0xefc60000u; uint32_t t = y; t = y ^ ((t << 7) & 0x9d2c5680u); t = y ^ ((t << 7) & 0x9d2c5680u); t = y ^ ((t << 7) & 0x9d2c5680u); t = y ^ ((t << 7) & 0x9d2c5680u); y = t;
                                         t = y; t = y ^ (t >> 11); t = y ^ (t >> 11); st[i] = t; }      // (mt_untemper on the host)
        st[624] = 624u;
        out.set_state(st); out.drawn = 0;
        if (cache) cache->fetches++;
        return CANVAS_OK;
    }
    // the real generator at the current position (nullptr: a HIP error, ctx->err set)
    MT* host(PermGpu& PG) {
        if (valid) return &mt;
        MT m(0u); if (at(PG, pos, m) != CANVAS_OK) return nullptr;
        mt = m; mt.drawn = 0; base = pos; valid = true;
        return &mt;
    }
};
// TailP for the two decisions of FindChangePoints: device approximation, accepted only when every p1 within 1e-8 relative gives the same decisions; else the exact host series
static int32_t tail_p_decide(PermGpu& PG, double b, double delta, int m, double cutoff, uint32_t nPerm, Stats& st, bool& exitNoSplit, int& nrejc) {
    const int nGrid = 100; const double tol = 1E-6;
    auto exact = [&]() { const double p1 = tail_p(b, delta, m, nGrid, tol); st.tailp_host++; exitNoSplit = p1 > cutoff; nrejc = exitNoSplit ? 0 : (int)((cutoff - p1) * nPerm); };
    if (cvx_hook("CANVAS_CBS_HOST_TAILP")) { exact(); return CANVAS_OK; }
    canvas_ctx* ctx = PG.ctx;
    int32_t rc = PG.ensure_tail(); if (rc) return rc;
    double* hNu = (double*)PG.tailPin; int* hFlag = (int*)(hNu + 128); volatile unsigned* hSeq = (volatile unsigned*)(hFlag + 128);      // the mailbox: 1024 + 512 + 4 bytes of the 3072
    const double dincr = (0.5 - delta) / nGrid, bs = b / std::sqrt((double)m);
    double tl = 0.5 - dincr, t = 0.5 - 0.5 * dincr, tls[128];
    TailArgs A;
    for (int i = 0; i < nGrid; i++) { tl = tl + dincr; t = t + dincr; A.x[i] = bs / std::sqrt(t * (1 - t)); tls[i] = tl; }
    unsigned seq = ++PG.tailSeq; if (seq == 0) seq = ++PG.tailSeq;
    *hSeq = 0u; std::atomic_thread_fence(std::memory_order_seq_cst);
    hipLaunchKernelGGL(k_tail_nu, dim3(nGrid), dim3(256), 0, PG.tailStream, A, nGrid, tol, hNu, hFlag, (unsigned*)PG.tailDev, (unsigned*)hSeq, seq);
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(PG.tailStream));
    CANVAS_HIP_TRY(ctx, hipGetLastError());
    { int32_t rcm = cvx_mail_await(ctx, hSeq, seq, "canvas_cbs (tail series)"); if (rcm) return rcm; }
    bool flagged = false; for (int i = 0; i < nGrid; i++) if (hFlag[i] || !(hNu[i] == hNu[i])) flagged = true;
    if (flagged) { exact(); return CANVAS_OK; }
    double tp = 0.0;
    for (int i = 0; i < nGrid; i++) tp = tp + sq(hNu[i]) * integral_inv(tls[i], dincr);
    tp = 9.973557E-2 * (b * b * b) * std::exp(-sq(b) / 2) * tp;
    const double p1 = 2.0 * tp, eps = 1e-8 * std::fabs(p1) + 1e-300;
    const double lo = p1 - eps, hi = p1 + eps;
    if ((lo > cutoff) != (hi > cutoff)) { exact(); return CANVAS_OK; }
    if (lo > cutoff) { exitNoSplit = true; nrejc = 0; st.tailp_dev++; return CANVAS_OK; }
    const int a = (int)((cutoff - lo) * nPerm), c = (int)((cutoff - hi) * nPerm);
    if (a != c) { exact(); return CANVAS_OK; }
    exitNoSplit = false; nrejc = a; st.tailp_dev++;
    return CANVAS_OK;
}

// All chromosome threads hand their batches to ONE launcher thread: whatever is waiting goes into a single k_mt_draws launch (one
// workgroup per request: the generator is sequential per chromosome, the chromosomes are not) and a single k_perm_stat launch (one
// workgroup per permutation of every request).  Concurrency then does not depend on how many hardware queues the runtime maps the
// per-thread streams to.
struct PermHostReq { PermReq r; long long prevTotal = 0; double* hStat; uint32_t* hSnaps; const double* hX = nullptr; double* dX = nullptr; size_t xBytes = 0;
                     const uint32_t* hDraws = nullptr; size_t drawBytes = 0;      // short segments: the draws of the batch, produced by the host generator
                     bool done = false; int32_t rc = CANVAS_OK; };
// the device side of a launcher (its stream and request tables): kept by the context between calls — canvas_cbs builds seven launchers per call, and each paid a stream
// creation and six small allocations on its first round
struct SvcRes { hipStream_t stream = nullptr; PermReq* dReqs = nullptr; PermReq* hReqs = nullptr; ArcReq* dArc = nullptr; ArcReq* hArc = nullptr; ArcPReq* dArcP = nullptr; ArcPReq* hArcP = nullptr;
                char* rpSlab = nullptr; size_t rpSlabBytes = 0; };      // rpSlab: the scratch of k_perm_rp's persistent workgroups for whatever this launcher has in flight
#define SVC_REQ_CAP 32
// the device side of a launcher, created on its first use — or ahead of it by EngineCache::warm (canvas_cbs_prefetch): a stream costs ~5 ms on this runtime, and seven
// launchers creating theirs at the start of the first call were 60 ms of it
// (quiet = the background warmer: it must not write the context's error string, which belongs to the thread that makes the calls)
static int32_t svc_res_create(canvas_ctx* ctx, SvcRes* res, bool quiet = false) {
    if (res->stream && res->hArcP) return CANVAS_OK;
    hipError_t e = hipSuccess;
    auto step = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    if (!res->stream) step(hipStreamCreateWithFlags(&res->stream, hipStreamNonBlocking));
    if (e == hipSuccess && !res->dReqs) step(hipMalloc((void**)&res->dReqs, SVC_REQ_CAP * sizeof(PermReq)));
    if (e == hipSuccess && !res->hReqs) step(hipHostMalloc((void**)&res->hReqs, SVC_REQ_CAP * sizeof(PermReq), hipHostMallocDefault));
    if (e == hipSuccess && !res->dArc) step(hipMalloc((void**)&res->dArc, 64 * sizeof(ArcReq)));
    if (e == hipSuccess && !res->hArc) step(hipHostMalloc((void**)&res->hArc, 64 * sizeof(ArcReq), hipHostMallocDefault));
    if (e == hipSuccess && !res->dArcP) step(hipMalloc((void**)&res->dArcP, 64 * sizeof(ArcPReq)));
    if (e == hipSuccess && !res->hArcP) step(hipHostMalloc((void**)&res->hArcP, 64 * sizeof(ArcPReq), hipHostMallocDefault));
    if (e != hipSuccess) { (void)hipGetLastError(); if (!quiet) ctx->err = std::string("canvas_cbs: creating a launcher's stream / request tables: ") + hipGetErrorString(e); return CANVAS_ERR_HIP; }
    return CANVAS_OK;
}
struct PermService {
    canvas_ctx* ctx; SvcRes* res = nullptr; hipStream_t stream = nullptr; PermReq* dReqs = nullptr; PermReq* hReqs = nullptr; int cap = 32;
    ArcReq* dArc = nullptr; ArcReq* hArc = nullptr; ArcPReq* dArcP = nullptr; ArcPReq* hArcP = nullptr; std::vector<ArcHostReq*> pendingArc;
    static_assert(SVC_REQ_CAP == 32, "cap below");
    long long rounds = 0, nArc = 0, nPermReq = 0, nArcRounds = 0; double secArc = 0, secPerm = 0; unsigned arcSeq = 0;
    // k_perm_rp's scratch belongs to the LAUNCHER, not to the engines: a launch holds the batches of several chromosomes, the device has 512 workgroup slots for all of them, and
    // a launcher has one launch in flight — so one slab per launcher (2 GiB for the longest segments) serves what 24 engines used to reserve 2 GiB EACH for (48 GB of
    // hipMalloc in the first call of a process, which at times took seconds beside the stream cache's mappings: ADVICE r05, profiles/r06_*).  Requests share the slab in
    // proportion to the workgroups they ask for.
    size_t rpSlabWant = 0;
    bool probeTiming = false; double lastMs[3] = {0, 0, 0};      // canvas_cbs_perm_probe: generator (sequential + bootstrap), generator (strided), permutation + statistic of the last launch
    std::mutex mu; std::condition_variable cvWork, cvDone; std::vector<PermHostReq*> pending; bool stop = false; std::thread th; std::string err;
    explicit PermService(canvas_ctx* c) : ctx(c) { th = std::thread([this]() { run(); }); }
    ~PermService() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cvWork.notify_all(); th.join();
        if (stream) (void)hipStreamSynchronize(stream);
        if (res) cache_svc_give(ctx, res); }
    int32_t submit(PermHostReq& q) {
        std::unique_lock<std::mutex> lk(mu);
        pending.push_back(&q);
        cvWork.notify_one();
        cvDone.wait(lk, [&]() { return q.done; });
        if (q.rc) ctx->err = err;
        return q.rc;
    }
    int32_t submit_arc(ArcHostReq& q) {
        std::unique_lock<std::mutex> lk(mu);
        pendingArc.push_back(&q);
        cvWork.notify_one();
        cvDone.wait(lk, [&]() { return q.done; });
        if (q.rc) ctx->err = err;
        return q.rc;
    }
    int32_t init() {
        CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
        if (!stream) {
            AllocClock ac(g_ns_alloc_svc);
            res = cache_svc_take(ctx);
            { int32_t rcr = svc_res_create(ctx, res); if (rcr) return rcr; }
            stream = res->stream; dReqs = res->dReqs; hReqs = res->hReqs; dArc = res->dArc; hArc = res->hArc; dArcP = res->dArcP; hArcP = res->hArcP;
        }
        return CANVAS_OK;
    }
    // all waiting arc searches in shared launches (grid.y = request): pruned pipeline, or the exhaustive kernel for requests that ask for it
    int32_t launch_arc(std::vector<ArcHostReq*>& all) {
        const auto tA0 = std::chrono::steady_clock::now();
        int32_t rc = init(); if (rc) return rc;
        const double msInit = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tA0).count();
        struct FirstRounds { PermService* s; double msInit; std::chrono::steady_clock::time_point t0; size_t n; ~FirstRounds() { if (s->nArcRounds++ < 2 && cvx_hook("CANVAS_CBS_TIMING")) fprintf(stderr, "cbs arc launcher %p round %lld: %zu searches, init %.2f ms, copies + kernels + wait %.2f ms\n", (void*)s, s->nArcRounds, n, msInit, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() - msInit); } } fr{this, msInit, tA0, all.size()};
        std::vector<ArcHostReq*> pr, ex;
        for (auto* q : all) (q->pruned ? pr : ex).push_back(q);
        for (auto* q : all) CANVAS_HIP_TRY(ctx, hipMemcpyAsync((void*)q->r.sx, q->hSx, (size_t)q->r.n * 8, hipMemcpyHostToDevice, stream));
        if (!pr.empty()) {
            const int R = (int)pr.size(); int maxN = 0;
            for (int i = 0; i < R; i++) { unsigned sq_ = ++arcSeq; if (sq_ == 0) sq_ = ++arcSeq; pr[i]->p.hOut = pr[i]->hOut; pr[i]->p.hSeq = (unsigned*)(pr[i]->hOut + 6); pr[i]->p.seq = sq_; *(volatile unsigned*)pr[i]->p.hSeq = 0u;
                                          hArcP[i] = pr[i]->p; maxN = std::max(maxN, pr[i]->p.n); }
            std::atomic_thread_fence(std::memory_order_seq_cst);
            CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dArcP, hArcP, R * sizeof(ArcPReq), hipMemcpyHostToDevice, stream));
            const int nb = (maxN + AP_BK - 1) / AP_BK;
            hipLaunchKernelGGL(k_arcp_blocks, dim3(nb, R), dim3(256), 0, stream, dArcP);
            hipLaunchKernelGGL(k_arcp_bounds, dim3((unsigned)(((long long)nb * nb + 255) / 256), R), dim3(256), 0, stream, dArcP, 0);
            hipLaunchKernelGGL(k_arcp_bounds, dim3((unsigned)(((long long)nb * nb + 255) / 256), R), dim3(256), 0, stream, dArcP, 1);
            hipLaunchKernelGGL(k_arcp_eval, dim3(AP_EVAL_GRID, R), dim3(256), 0, stream, dArcP, 0);
            hipLaunchKernelGGL(k_arcp_eval, dim3(AP_EVAL_GRID, R), dim3(256), 0, stream, dArcP, 1);
            hipLaunchKernelGGL(k_arcp_mail, dim3(R), dim3(64), 0, stream, dArcP);
        }
        if (!ex.empty()) {
            const int R = (int)ex.size(); int maxN = 0;
            for (int i = 0; i < R; i++) { hArc[i] = ex[i]->r; maxN = std::max(maxN, ex[i]->r.n); }
            CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dArc, hArc, R * sizeof(ArcReq), hipMemcpyHostToDevice, stream));
            hipLaunchKernelGGL(k_arc_search, dim3((maxN / 2 + 1 + ARC_THREADS - 1) / ARC_THREADS, R), dim3(ARC_THREADS), 0, stream, dArc);
            for (int i = 0; i < R; i++) {
                CANVAS_HIP_TRY(ctx, hipMemcpyAsync(ex[i]->hMax, ex[i]->r.dmax, (size_t)ex[i]->r.n * 8, hipMemcpyDeviceToHost, stream));
                CANVAS_HIP_TRY(ctx, hipMemcpyAsync(ex[i]->hFirst, ex[i]->r.firstI, (size_t)ex[i]->r.n * 4, hipMemcpyDeviceToHost, stream));
            }
        }
        CANVAS_HIP_TRY(ctx, hipStreamSynchronize(stream));
        CANVAS_HIP_TRY(ctx, hipGetLastError());
        for (auto* q : pr) { int32_t rcm = cvx_mail_await(ctx, (volatile unsigned*)q->p.hSeq, q->p.seq, "canvas_cbs (arc search)"); if (rcm) return rcm; }
        return CANVAS_OK;
    }
    int32_t launch(std::vector<PermHostReq*>& batch) {
        { int32_t rc0 = init(); if (rc0) return rc0; }
        const int R = (int)batch.size();
        int blocks = 0, rpBlocks = 0; long long maxTotal = 0;
        // ---- the launcher's slab shared among the k_perm_rp requests of this launch that bring no scratch of their own (the probe does)
        auto al256 = [](size_t v) { return (v + 255) & ~size_t(255); };
        { size_t want = 0; for (int i = 0; i < R; i++) if (batch[i]->r.fy == 3 && !batch[i]->r.rpScratch) want += al256((size_t)batch[i]->r.rp.stride * 4 * (size_t)std::max(1, batch[i]->r.rpWGs));
          if (want) {
              if (!res->rpSlab || res->rpSlabBytes < std::min<size_t>(want, std::max<size_t>(rpSlabWant, size_t(64) << 20))) {      // (too small for this launch and allowed to grow)
                  AllocClock ac(g_ns_alloc_perm);
                  if (res->rpSlab) { CANVAS_HIP_TRY(ctx, hipStreamSynchronize(stream)); CANVAS_HIP_TRY(ctx, hipFree(res->rpSlab)); res->rpSlab = nullptr; res->rpSlabBytes = 0; }
                  // as large as this launch wants (a power of two from 64 MB), never beyond rpSlabWant: a germline call whose loops are short never allocates the 2 GiB a
                  // tumour / normal call grows to (allocation and, at the end of a one-shot process, release of 4 x 2 GiB were 0.1 s of CanvasPartition -m CBS)
                  size_t bytes = size_t(64) << 20; while (bytes < want && bytes < std::max<size_t>(rpSlabWant, size_t(64) << 20)) bytes <<= 1;
                  bytes = std::min(bytes, std::max<size_t>(rpSlabWant, size_t(64) << 20));
                  for (int i = 0; i < R; i++) if (batch[i]->r.fy == 3 && !batch[i]->r.rpScratch) bytes = std::max(bytes, al256((size_t)batch[i]->r.rp.stride * 4) + 256);      // at least one workgroup of the longest plan
                  CANVAS_HIP_TRY(ctx, hipMalloc((void**)&res->rpSlab, bytes)); res->rpSlabBytes = bytes;
              }
              const double scale = want > res->rpSlabBytes ? (double)res->rpSlabBytes / (double)want * 0.98 : 1.0;
              size_t off = 0;
              for (int i = 0; i < R; i++) if (batch[i]->r.fy == 3 && !batch[i]->r.rpScratch) {
                  PermReq& q = batch[i]->r; const size_t per = (size_t)q.rp.stride * 4;
                  int wg = std::max(1, (int)((double)std::max(1, q.rpWGs) * scale));
                  while (wg > 1 && off + al256(per * (size_t)wg) > res->rpSlabBytes) wg--;
                  if (off + al256(per * (size_t)wg) > res->rpSlabBytes) CANVAS_FAIL(ctx, CANVAS_ERR_HIP, "canvas_cbs: the launcher's scratch slab cannot hold one workgroup of every request");
                  const int rounds = (q.nb + wg - 1) / wg; q.rpWGs = (q.nb + rounds - 1) / rounds;      // (every workgroup the same number of permutations)
                  q.rpScratch = (uint32_t*)(res->rpSlab + off); off += al256(per * (size_t)q.rpWGs);
              }
          } }
        for (int i = 0; i < R; i++) {
            batch[i]->r.rpBase = rpBlocks; if (batch[i]->r.fy == 3) rpBlocks += batch[i]->r.rpWGs; else batch[i]->r.rpWGs = 0;
            batch[i]->r.blockBase = blocks; blocks += batch[i]->r.nb; hReqs[i] = batch[i]->r; if (batch[i]->r.fy != 2 && !batch[i]->r.cached) maxTotal = std::max(maxTotal, batch[i]->r.total + (batch[i]->r.cont ? MT_HISTORY : 0));
            if (batch[i]->xBytes) CANVAS_HIP_TRY(ctx, hipMemcpyAsync(batch[i]->dX, batch[i]->hX, batch[i]->xBytes, hipMemcpyHostToDevice, stream));     // (the caller's array: it stays valid until the launch has been waited for)
            if (batch[i]->drawBytes) CANVAS_HIP_TRY(ctx, hipMemcpyAsync(batch[i]->r.P.draws, batch[i]->hDraws, batch[i]->drawBytes, hipMemcpyHostToDevice, stream));
        }
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dReqs, hReqs, R * sizeof(PermReq), hipMemcpyHostToDevice, stream));
        static const bool dbgEnv = cvx_hook("CANVAS_CBS_DEBUG_BATCHES") != nullptr; const bool dbg = dbgEnv || probeTiming;
        auto tp0 = std::chrono::steady_clock::now(); double msA = 0, msB = 0;
        auto lap = [&]() { (void)hipStreamSynchronize(stream); auto t = std::chrono::steady_clock::now(); const double ms = std::chrono::duration<double, std::milli>(t - tp0).count(); tp0 = t; return ms; };
        if (dbg) lap();
        const bool bootstrap = true;      // (the sequential history is grown by doubling: k_mt_classes with boot = 1)
        bool anyFresh = false; long long maxFresh = 0;
        bool anyOwn = false;      // requests that generate their own draws (the others read them out of the stream cache: no generator, no snapshots)
        for (int i = 0; i < R; i++) if (batch[i]->r.fy != 2 && !batch[i]->r.cached) anyOwn = true;
        for (int i = 0; i < R; i++) if (!batch[i]->r.cont && batch[i]->r.fy != 2 && !batch[i]->r.cached) { anyFresh = true; if (batch[i]->r.total >= MT_BOOT_MIN) maxFresh = std::max(maxFresh, batch[i]->r.total); }
        if (anyFresh) hipLaunchKernelGGL(k_mt_draws, dim3(R), dim3(256), 0, stream, dReqs, bootstrap ? 1 : 0);
        if (anyFresh && bootstrap)
            for (int sd = 1; sd < MT_STRIDE && 19937LL * sd < maxFresh; sd <<= 1) hipLaunchKernelGGL(k_mt_classes, dim3(sd, R), dim3(MTC_T), 0, stream, dReqs, sd, 1);
        if (dbg) msA = lap();
        const int steps = maxTotal > MT_HISTORY ? (int)((maxTotal - MT_HISTORY + MT_WIDTH - 1) / MT_WIDTH) : 0;
        if (steps > 0) hipLaunchKernelGGL(k_mt_classes, dim3(MT_STRIDE, R), dim3(MTC_T), 0, stream, dReqs, MT_STRIDE, 0);
        if (dbg) msB = lap();
        if (anyOwn) hipLaunchKernelGGL(k_mt_snapshots, dim3(blocks), dim3(256), 0, stream, dReqs, R);
        bool anyFy = false, anyOld = false, anySmall = false; for (int i = 0; i < R; i++) (batch[i]->r.fy == 2 ? anySmall : batch[i]->r.fy == 1 ? anyFy : batch[i]->r.fy == 0 ? anyOld : anySmall /* (3: below) */) |= batch[i]->r.fy != 3;
        if (anyOld) hipLaunchKernelGGL(k_perm_stat, dim3(blocks), dim3(PG_T), 0, stream, dReqs, R);
        if (anyFy) hipLaunchKernelGGL(k_perm_fy, dim3(blocks), dim3(PG_T), 0, stream, dReqs, R);
        if (rpBlocks) {
            static const hipError_t ldsAttr = hipFuncSetAttribute((const void*)k_perm_rp, hipFuncAttributeMaxDynamicSharedMemorySize, RPL_TOTAL);      // (more than the 64 KB a kernel gets without asking)
            CANVAS_HIP_TRY(ctx, ldsAttr);
            hipLaunchKernelGGL(k_perm_rp, dim3(rpBlocks), dim3(RP_T), RPL_TOTAL, stream, dReqs, R);
        }
        if (anySmall) hipLaunchKernelGGL(k_perm_small, dim3(blocks), dim3(64), 0, stream, dReqs, R);
        if (probeTiming) { lastMs[0] = msA; lastMs[1] = msB; lastMs[2] = lap(); }
        else if (dbg) { const double msC = lap(); int nc = 0, maxN = 0; for (int i = 0; i < R; i++) { nc += batch[i]->r.cont; maxN = std::max(maxN, batch[i]->r.n); }
                   fprintf(stderr, "cbs batch: %d requests (%d continued), %d permutations, longest segment %d, %d stride steps: sequential %.2f ms, strided %.2f ms, statistics %.2f ms\n", R, nc, blocks, maxN, steps, msA, msB, msC); }
        bool anyMail = false; for (int i = 0; i < R; i++) anyMail |= batch[i]->r.mailStat != nullptr;
        if (anyMail) hipLaunchKernelGGL(k_perm_mail, dim3(R), dim3(256), 0, stream, dReqs);
        for (int i = 0; i < R; i++) {
            if (!batch[i]->r.mailStat) CANVAS_HIP_TRY(ctx, hipMemcpyAsync(batch[i]->hStat, batch[i]->r.pstat, (size_t)batch[i]->r.nb * 16, hipMemcpyDeviceToHost, stream));
            if (batch[i]->hSnaps) CANVAS_HIP_TRY(ctx, hipMemcpyAsync(batch[i]->hSnaps, batch[i]->r.snaps, (size_t)batch[i]->r.nb * 625 * 4, hipMemcpyDeviceToHost, stream));
        }
        CANVAS_HIP_TRY(ctx, hipStreamSynchronize(stream));
        CANVAS_HIP_TRY(ctx, hipGetLastError());
        for (int i = 0; i < R; i++) if (batch[i]->r.mailStat) { int32_t rcm = cvx_mail_await(ctx, (volatile unsigned*)batch[i]->r.mailSeq, batch[i]->r.seq, "canvas_cbs (permutation batch)"); if (rcm) return rcm; }
        return CANVAS_OK;
    }
    void run() {
        for (;;) {
            std::vector<PermHostReq*> batch; std::vector<ArcHostReq*> arcs;
            { std::unique_lock<std::mutex> lk(mu); cvWork.wait(lk, [&]() { return stop || !pending.empty() || !pendingArc.empty(); }); if (pending.empty() && pendingArc.empty()) return;
              while (!pending.empty() && (int)batch.size() < cap) { batch.push_back(pending.front()); pending.erase(pending.begin()); }
              while (!pendingArc.empty() && (int)arcs.size() < 64) { arcs.push_back(pendingArc.front()); pendingArc.erase(pendingArc.begin()); } }
            std::string saved = ctx->err;
            auto t0 = std::chrono::steady_clock::now();
            int32_t rcA = arcs.empty() ? CANVAS_OK : launch_arc(arcs);
            auto t1 = std::chrono::steady_clock::now();
            int32_t rc = batch.empty() ? CANVAS_OK : launch(batch);
            auto t2 = std::chrono::steady_clock::now();
            rounds++; secArc += std::chrono::duration<double>(t1 - t0).count(); secPerm += std::chrono::duration<double>(t2 - t1).count(); nArc += (long long)arcs.size(); nPermReq += (long long)batch.size();
            { std::lock_guard<std::mutex> lk(mu); if (rc || rcA) { err = ctx->err; ctx->err = saved; }
              for (auto* q : batch) { q->rc = rc; q->done = true; } for (auto* q : arcs) { q->rc = rcA; q->done = true; } }
            cvDone.notify_all();
        }
    }
};
static int32_t service_submit_arc(PermService* svc, ArcHostReq& q) { return svc->submit_arc(q); }
// where the wall time of one chromosome goes (CANVAS_CBS_TIMING): seconds waiting for / running phase 1, in the device and host permutation loops, in the edge tests
struct ChromClock { double p1 = 0, dev = 0, host = 0, edge = 0; int segments = 0, devLoops = 0, hostLoops = 0; long long loopPerms = 0, loopBatches = 0, loopComputed = 0; };
static thread_local ChromClock tlClock;
// ---- workspace of a permutation loop.  k_perm_rp (segments of PERM_RP_MIN_N .. PERM_RP_MAX_N bins): the scratch of its persistent workgroups (+ the draws of two batches
// when the loop generates its own, i.e. without a cache stream); the older kernels (shorter / longer segments): 48 bytes per permuted element.
#define PERM_RP_SCRATCH_BYTES (size_t(2) << 30)
#define PERM_RP_MAXB 1024
static inline size_t al256(size_t v) { return (v + 255) & ~size_t(255); }
static inline bool perm_use_rp(int n) { static const bool off = cvx_hook("CANVAS_CBS_NO_RP") != nullptr; static const int minN = cvx_hook("CANVAS_CBS_RP_MIN_N") ? atoi(cvx_hook("CANVAS_CBS_RP_MIN_N")) : PERM_RP_MIN_N; return !off && n >= minN && n <= PERM_RP_MAX_N; }
// cached: the batch reads its draws out of the chromosome's stream — no draw buffers, so a batch of a long segment could be as large as the stopping rule asks for (a 234 k-bin
// segment: 1024 permutations instead of 286).  Measured with a switch that has been removed again (268 435 456 elements per cached batch): the tumour / normal flow's CBS 0.436 against 0.438 s, the germline call 0.061
// against 0.047 s — the long-running workgroups of the larger batches keep the short arc / tail kernels of the other chromosomes waiting for a CU.  The size stays.
static inline long long perm_target_elems(bool /*cached*/) { return (long long)PERM_TARGET_ELEMS; }
static inline int perm_max_batch(int n, bool cached = false) { return perm_use_rp(n) ? (int)std::max<long long>(8, std::min<long long>(PERM_RP_MAXB, perm_target_elems(cached) / n)) : (int)std::max<long long>(8, std::min<long long>(256, PERM_TARGET_ELEMS / n)); }
static inline int perm_rp_wgs(int n) { PermReq::RpPlan P; rp_plan(n, P); const size_t per = (size_t)P.stride * 4; return (int)std::max<size_t>(64, std::min<size_t>(PERM_RP_GRID, PERM_RP_SCRATCH_BYTES / per)); }
// the scratch of k_perm_rp's workgroups for segments of up to nMax bins: what perm_rp_wgs() workgroups of the LONGEST plan take, never more than PERM_RP_SCRATCH_BYTES — a
// small genome reserves megabytes, not 2 GiB
static inline size_t perm_rp_scratch_bytes(size_t nMax) {
    if (nMax < (size_t)PERM_RP_MIN_N) return 0;
    const int n = (int)std::min<size_t>(nMax, PERM_RP_MAX_N);
    if (!perm_use_rp(n)) return 0;
    PermReq::RpPlan P; rp_plan(n, P);
    return al256((size_t)P.stride * 4 * (size_t)perm_rp_wgs(n));
}
// device / pinned bytes that serve every segment of up to nMax bins; withDraws: the loop generates its own draws (no cache stream): two draw buffers with their history
static void perm_reserve_bytes(size_t nMax, bool withDraws, size_t& dev, size_t& pin) {
    const size_t head = al256(nMax * 8) + al256(625 * 4) + al256((size_t)PERM_RP_MAXB * 625 * 4) + al256((size_t)PERM_RP_MAXB * 16);
    const size_t elemsRp = std::max<size_t>(PERM_TARGET_ELEMS, 8 * nMax);
    const size_t drawsB = withDraws ? 2 * al256((elemsRp + (size_t)MT_HISTORY) * 4) : 0;
    dev = head + drawsB + (size_t(4) << 20);      // (k_perm_rp's scratch is the launcher's: PermService::rpSlabWant)
    if (nMax > (size_t)PERM_RP_MAX_N || !perm_use_rp((int)std::min<size_t>(nMax, PERM_RP_MAX_N))) {
        const size_t re = (size_t)std::min<long long>((long long)256 * (long long)nMax, std::max<long long>(PERM_TARGET_ELEMS, 8LL * (long long)nMax));
        dev = std::max(dev, head + (withDraws ? 2 * al256((re + (size_t)MT_HISTORY) * 4) : 0) + 5 * al256(re * 4) + 2 * al256((re + 256) * 4) + 2 * al256(re * 8));
    }
    pin = (withDraws ? al256((size_t)PERM_RP_MAXB * 625 * 4) : 0) + al256((size_t)PERM_RP_MAXB * 16) + 256 + al256((size_t)2048 * 16);      // (the last term: the small-segment loop's intervals)
}

// The sequential stopping rule of FindChangePoints (ChangePoint.cs:337-364) over permutations evaluated in device batches.
// Returns through `outcome`: 0 = not significant (nrej > nrejc), 1 = continue to the edge tests.  rnd ends exactly where the reference's would.
// The draws of a batch are read out of the chromosome's cache stream (MtStreamCache: nothing is generated, no state travels; rnd moves by position); when the stream cannot
// serve a batch (cache off, bound reached) the batch generates its own on the device from the generator's state and hands the state after every permutation back.
static int32_t perm_loop_gpu(PermGpu& PG, const double* gd, int n, double tss, uint32_t nPerm, int hk, int al0, double ostat, int nrejc, int k,
                             const std::vector<uint32_t>& sbdry, Rng& rnd, Stats& st, int& outcome) {
    canvas_ctx* ctx = PG.ctx;
    const bool useRp = perm_use_rp(n) && hk == RP_J1 && al0 == RP_J0;      // (k_perm_rp's statistic is written for FindChangePoints' own arc lengths)
    int maxB = perm_max_batch(n, rnd.S != nullptr);
    auto al = [](size_t v) { return (v + 255) & ~size_t(255); };
    auto now = []() { return std::chrono::steady_clock::now(); };
    uint32_t* dRpScratch = nullptr; int rpWGs = 0; PermReq::RpPlan rpP; memset(&rpP, 0, sizeof rpP);
    auto since = [](std::chrono::steady_clock::time_point t) { return (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t).count(); };
    double* dX = nullptr; uint32_t* dSnaps = nullptr; double* dStat = nullptr; PermBuf P; uint32_t* hSnaps = nullptr; double* hStat = nullptr;
    uint32_t* draws2[2] = {nullptr, nullptr}; int drawBuf = 0;      // (own draws) the loop's batches alternate between two draw buffers
    bool haveDraws = false; volatile unsigned* hMail = nullptr;
    auto setup = [&](int mb, bool withDraws) -> int32_t {
        const size_t e = (size_t)mb * n, e1 = (size_t)mb * (n + 1);
        const size_t drawsB = withDraws ? 2 * al((e + (size_t)MT_HISTORY) * 4) : 0;      // two draw buffers: a batch reads its history where the previous one wrote it
        const size_t oX = 0, oState = oX + al((size_t)n * 8), oSnaps = oState + al(625 * 4), oStat = oSnaps + al((size_t)mb * 625 * 4), oDraws = oStat + al((size_t)mb * 16),
                     oJ = oDraws + drawsB, oOff = oJ + al(e * 4), oCur = oOff + al(e1 * 4), oItems = oCur + al(e1 * 4), oG = oItems + al(e * 4), oSucc = oG + al(e * 4),
                     oPx = oSucc + al(e * 4), oSx = oPx + al(e * 8), total = oSx + al(e * 8);
        // pinned: the generator states of a batch (own draws only), its intervals and the mailbox's sequence word.  The segment itself is uploaded from the caller's array, once
        // per loop (pageable: ~0.4 ms for the longest chromosome, against a loop of milliseconds to seconds) — a pinned staging copy per engine was 3 MB x 24 engines, and with
        // the states 132 MB that a one-shot process pinned at its start and unpinned at its exit (1.4 ms per MB each way)
        const size_t pSnaps = 0, pStat = pSnaps + (withDraws ? al((size_t)mb * 625 * 4) : 0), pMail = pStat + al((size_t)mb * 16), pinTotal = pMail + 256;
        auto tE = now();
        // k_perm_rp: behind the draws the scratch of the persistent workgroups (results travel in streams: no per-element workspace)
        size_t totalRp = 0;
        if (useRp) { rp_plan(n, rpP); rpWGs = std::min(perm_rp_wgs(n), mb); totalRp = oDraws + drawsB + 256; }      // (the workgroups' scratch comes out of the launcher's slab at launch time)
        const size_t need = useRp ? totalRp : total;
        size_t want = need, wantPin = pinTotal;
        if (PG.reserveBytes) { want = std::max(want, PG.reserveBytes); wantPin = std::max(wantPin, PG.reservePin); }      // the first allocation serves the longest segment this call can meet
        int32_t rc0 = PG.ensure(want, wantPin, need); if (rc0) return rc0;
        st.ns_ensure += since(tE);
        char* d = PG.buf; char* h = PG.pin;
        dX = (double*)(d + oX); dSnaps = (uint32_t*)(d + oSnaps); dStat = (double*)(d + oStat);
        memset(&P, 0, sizeof P);
        if (withDraws) { P.draws = (uint32_t*)(d + oDraws) + MT_HISTORY; draws2[0] = P.draws; draws2[1] = (uint32_t*)(d + oDraws + al((e + (size_t)MT_HISTORY) * 4)) + MT_HISTORY; }
        if (!useRp) {
        P.j = (int32_t*)(d + oJ); P.off = (int32_t*)(d + oOff); P.cur = (int32_t*)(d + oCur); P.items = (int32_t*)(d + oItems);
        P.g = (int32_t*)(d + oG); P.succ = (int32_t*)(d + oSucc); P.px = (double*)(d + oPx); P.sx = (double*)(d + oSx); }
        hSnaps = (uint32_t*)(h + pSnaps); hStat = (double*)(h + pStat); hMail = (volatile unsigned*)(h + pMail);
        haveDraws = withDraws;
        return CANVAS_OK;
    };
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    MtStreamCache* SC = rnd.cache; MtStream* S = rnd.S;
    int32_t rc = setup(maxB, S == nullptr); if (rc) return rc;
    bool needUpload = true;                   // the segment is uploaded by the launcher together with the first batch
    // worst-case rounding bound of a prefix-sum difference: both orders of summation are within gamma_n * sum|x| of the exact sum
    double absSum = 0.0; for (int i = 0; i < n; i++) absSum += std::fabs(gd[i]);
    const double errBound = 4.04 * (double)(n + 8) * 1.1102230246251565e-16 * absSum;
    std::vector<double> px, sx;
    long long pos = rnd.position();            // stream position of the next draw
    uint32_t cur[625]; bool haveCur = false;   // (own draws) the generator state at `pos`
    int nrej = 0; uint32_t np = 0;
    long long prevTotal = 0; bool prevOwn = false;
    // first batch: without a single rejection the sequential rule stops at permutation sbdry[k - 1] — known now — and that is what a segment with a real change point
    // does; asking for that many at once saves the 64 / 128 / 256 ramp its two extra launcher round trips (a segment without one leaves after a handful either way)
    int B = std::min(maxB, std::max(64, (int)std::min<uint32_t>(sbdry[k - 1], 4096u)));
    // ... but no more than four times what nrejc + 1 rejections need at a rejection rate of one in four, and no more than one round of the persistent workgroups (a batch of
    // 1 024 permutations is two rounds of 512 workgroups: two batches of 512 take as long and the second is only computed if the rule is still running).  Three in four loops of
    // the tumour / normal pair end "not significant" inside their first batch: with sbdry alone 11 % of all permuted elements were never looked at (profiles/r06_loop_waste.txt)
    B = std::min(B, std::max(64, std::min(512, 4 * (nrejc + 1))));
    outcome = 1;
    while (np < nPerm) {
        int nb = (int)std::min<uint32_t>((uint32_t)B, nPerm - np);
        long long need = (long long)nb * n;
        auto tS = now();
        const bool cached = S && SC->acquire(S, pos + need, std::max<long long>(2 * need, 16LL << 20));
        if (!cached) {
            // own draws: the state at pos (a generator that was moved by position is rebuilt from the cache's words in front of it) and the two draw buffers (batches of
            // the size those buffers are reserved for)
            if (!haveCur) { MT m(0u); rc = rnd.at(PG, pos, m); if (rc) return rc; m.get_state(cur); haveCur = true; prevOwn = false; }
            if (!haveDraws) { maxB = perm_max_batch(n, false); rc = setup(maxB, true); if (rc) return rc; needUpload = true; }
            if (nb > maxB) { nb = maxB; need = (long long)nb * n; }
        }
        PermHostReq q;
        q.r.total = need; q.r.n = n; q.r.nb = nb; q.r.snaps = dSnaps; q.r.x = dX; q.r.hk = hk; q.r.al0 = al0; q.r.tss = tss; q.r.errBound = errBound;
        q.r.P = P; q.r.pstat = dStat; q.r.blockBase = 0; q.hStat = hStat;
        if (needUpload) { q.hX = gd; q.dX = dX; q.xBytes = (size_t)n * 8; needUpload = false; }
        if (cached) {
            memset(q.r.state, 0, sizeof q.r.state); q.r.cached = 1; q.r.cont = 0; q.r.hist = nullptr; q.hSnaps = nullptr; q.r.snaps = nullptr;
            q.r.P.draws = S->d() + pos;
            SC->servedWords += need;
        } else {
            memcpy(q.r.state, cur, sizeof cur); q.r.cached = 0; q.hSnaps = hSnaps;
            q.r.cont = (prevOwn && prevTotal >= MT_HISTORY) ? 1 : 0; q.prevTotal = prevTotal;
            q.r.P.draws = draws2[drawBuf]; q.r.hist = q.r.cont ? draws2[drawBuf ^ 1] + (prevTotal - MT_HISTORY) : nullptr; drawBuf ^= 1;
            if (SC) SC->fallbackWords += need;
        }
        { static const int fyMin = cvx_hook("CANVAS_CBS_FY_MIN_N") ? atoi(cvx_hook("CANVAS_CBS_FY_MIN_N")) : PERM_FY_MIN_N; q.r.fy = useRp ? 3 : (n >= fyMin ? 1 : 0); }
        // persistent workgroups: every one takes the same number of permutations (600 permutations on 512 workgroups would be two rounds with 424 workgroups idle in the second:
        // 300 workgroups with two each take the same time and leave the other CUs to the kernels of the other chromosomes)
        { const int rounds = useRp ? (nb + rpWGs - 1) / rpWGs : 1; q.r.rpWGs = useRp ? (nb + rounds - 1) / rounds : 0; }
        q.r.rpBase = 0; q.r.rpScratch = dRpScratch; q.r.rp = rpP; q.r.rpClk = nullptr;
        { unsigned sq_ = ++PG.mailSeq; if (sq_ == 0) sq_ = ++PG.mailSeq; q.r.mailStat = hStat; q.r.mailSeq = (unsigned*)hMail; q.r.seq = sq_; *hMail = 0u; std::atomic_thread_fence(std::memory_order_seq_cst); }
        prevTotal = need; prevOwn = !cached;
        rc = PG.svc->submit(q); if (rc) return rc;
        st.ns_submit += since(tS);
        auto tP = now();
        struct PostAcc { std::atomic<long long>& a; std::chrono::steady_clock::time_point t; ~PostAcc() { a += (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t).count(); } } postAcc{st.ns_post, tP};
        st.dev_batches++; tlClock.loopBatches++; tlClock.loopComputed += nb;
        // (own draws) generator state behind permutation b of this batch.  The device snapshot is rebuilt from the last 624 outputs in front of that point: fewer than 624 exist when a
        // batch that does not continue another one is cut short inside its first permutations (segments of a few hundred bins) — then the batch's start state is advanced on
        // the host, at most 623 draws.
        const bool contBatch = q.r.cont != 0;
        uint32_t tmpState[625];
        auto state_after = [&](int b) -> const uint32_t* {
            if (contBatch || (long long)(b + 1) * n >= 624) return hSnaps + (size_t)b * 625;
            MT m(0u); m.set_state(cur); for (long long t = 0; t < (long long)(b + 1) * n; t++) (void)m.u32(); m.get_state(tmpState); return tmpState;
        };
        // the rule stops behind permutation b: rnd continues at that position
        auto stop_behind = [&](int b) { const long long p = pos + (long long)(b + 1) * n; if (cached) rnd.jump_to(p); else rnd.adopt(state_after(b), p); };
        // a real generator at the start of permutation b (re-evaluations, the test hook)
        auto gen_before = [&](int b, MT& m) -> int32_t {
            if (cached) return rnd.at(PG, pos + (long long)b * n, m);
            m.set_state(b == 0 ? cur : state_after(b - 1)); return CANVAS_OK;
        };
        if (cvx_hook("CANVAS_CBS_TEST_VERIFY")) {      // test hook: every device interval must contain the statistic computed in the reference's order
            MT m2(0u); rc = gen_before(0, m2); if (rc) return rc;
            if (cached) {      // ... and the cached stream must BE the generator's output: the batch's first words against MersenneTwister(seed) advanced to pos
                const size_t cw = (size_t)std::min<long long>(need, 4096); std::vector<uint32_t> hw(cw);
                CANVAS_HIP_TRY(ctx, hipMemcpy(hw.data(), S->d() + pos, cw * 4, hipMemcpyDeviceToHost));
                MT m4 = m2; for (size_t t = 0; t < cw; t++) if (m4.u32() != hw[t]) { st.violations++; break; }
            }
            for (int b = 0; b < nb; b++) {
                if (!cached) { rc = gen_before(b, m2); if (rc) return rc; }      // (cached: m2 runs on from permutation to permutation — the stream is consumed in sequence)
                px.resize(n); sx.resize(n);
                xperm(gd, px.data(), n, m2);
                const double exact = htmaxp_host(hk, tss, px.data(), n, sx.data(), al0);
                bool same = true;
                if (!cached) { MT m3(0u); m3.set_state(state_after(b)); MT m2c = m2; for (int t = 0; t < 1400; t++) if (m2c.u32() != m3.u32()) { same = false; break; } }      // the device snapshot must continue the stream exactly where the host generator is
                st.verified++;
                if (!(hStat[2 * b] <= exact && exact <= hStat[2 * b + 1]) || !same ||
                    !(hStat[2 * b + 1] - hStat[2 * b] <= 1e-6 * std::fabs(exact) + 1e-300)) st.violations++;
            }
        }
        for (int b = 0; b < nb; b++) {
            np++; tlClock.loopPerms++;
            st.perms++; st.perm_elems += n; st.dev_perms++;
            bool rej;
            const double lo = hStat[2 * b], hi = hStat[2 * b + 1];
            if (ostat <= lo) rej = true;
            else if (ostat > hi) rej = false;
            else {      // inside the rounding interval: this permutation again, in the reference's order of operations
                MT m2(0u); rc = gen_before(b, m2); if (rc) return rc;
                px.resize(n); sx.resize(n);
                xperm(gd, px.data(), n, m2);
                rej = ostat <= htmaxp_host(hk, tss, px.data(), n, sx.data(), al0);
                st.exact_rechecks++;
            }
            if (rej) { nrej++; k++; }
            if (nrej > nrejc) { stop_behind(b); outcome = 0; return CANVAS_OK; }
            if (np >= sbdry[k - 1]) { stop_behind(b); return CANVAS_OK; }
        }
        if (cached) haveCur = false;
        else { const uint32_t* sEnd = state_after(nb - 1); uint32_t keepState[625]; memcpy(keepState, sEnd, sizeof keepState); memcpy(cur, keepState, sizeof cur); }
        pos += need;
        // The next batch: as many permutations as the rule is expected to look at yet (+ 15 %), not simply twice the last one — the permutation kernels are the device's load,
        // and a loop that stops 20 permutations into a batch of 1024 has computed the other thousand for nothing (with the doubling a third of all permutations computed were never
        // looked at).  With the rejection rate seen so far, p, the rule stops where np reaches sbdry[k - 1 + p (np' - np)] or where the rejections exceed nrejc, whichever is first.
        {
            const double p = (double)nrej / (double)np;
            long long stopAt = nPerm;
            if (p > 0.0) stopAt = std::min<long long>(stopAt, (long long)np + (long long)std::ceil((double)(nrejc + 1 - nrej) / p));
            for (long long t = np; t <= stopAt; t += 16) {      // (the boundary moves up with every expected rejection)
                const long long kk = (long long)k + (long long)std::floor(p * (double)(t - np));
                if (kk - 1 >= (long long)sbdry.size()) break;
                if (t >= (long long)sbdry[(size_t)kk - 1]) { stopAt = t; break; }
            }
            const long long want = (long long)std::ceil(1.15 * (double)(stopAt - (long long)np)) + 8;
            B = (int)std::max<long long>(64, std::min<long long>(maxB, want));
        }
    }
    if (haveCur) rnd.adopt(cur, pos); else rnd.jump_to(pos);
    return CANVAS_OK;
}

// The same stopping rule for the segments of at most 200 bins (the non-hybrid test: XPerm + TMaxP, ChangePoint.cs:337-364): batches of up to 2048 permutations, one wave each
// (k_perm_small).  The draws of a batch come out of the chromosome's cache stream; without one they are produced by the chromosome's host generator in the reference's order
// and travel with the request.  The statistic comes back exact, so the comparison ostat <= pstat is the reference's own.  rnd ends behind the last permutation the rule looked at.
static int32_t perm_loop_small_gpu(PermGpu& PG, const double* gd, int n, double tss, uint32_t nPerm, int al0, double ostat, int nrejc, int k, const std::vector<uint32_t>& sbdry, Rng& rnd, Stats& st, int& outcome) {
    canvas_ctx* ctx = PG.ctx;
    const int maxB = 2048;
    auto al = [](size_t v) { return (v + 255) & ~size_t(255); };
    const size_t oX = 0, oStat = al((size_t)n * 8), oDraws = oStat + al((size_t)maxB * 16), total = oDraws + al(((size_t)maxB * n + (size_t)MT_HISTORY) * 4) + 256;
    MtStreamCache* SC = rnd.cache; MtStream* S = rnd.S;
    const size_t pStat = 0, pMail = pStat + al((size_t)maxB * 16), pDraws = pMail + 256;
    double* dX = nullptr; double* dStat = nullptr; uint32_t* dDraws = nullptr; double* hStat = nullptr; uint32_t* hDraws = nullptr; volatile unsigned* hMail = nullptr; bool havePinDraws = false;
    auto setup = [&](bool withDraws) -> int32_t {      // withDraws: the host generator's draws of a batch travel through pinned memory (no cache stream / its bound reached)
        size_t want = total, wantPin = pDraws + (withDraws ? al((size_t)maxB * n * 4) : 0);
        if (PG.reserveBytes) { want = std::max(want, PG.reserveBytes); wantPin = std::max(wantPin, PG.reservePin); }      // never shrink below what perm_loop_gpu will ask for: one allocation per engine
        int32_t rc0 = PG.ensure(want, wantPin, total); if (rc0) return rc0;
        char* d = PG.buf; char* h = PG.pin;
        dX = (double*)(d + oX); dStat = (double*)(d + oStat); dDraws = (uint32_t*)(d + oDraws) + MT_HISTORY;
        hStat = (double*)(h + pStat); hDraws = (uint32_t*)(h + pDraws); hMail = (volatile unsigned*)(h + pMail); havePinDraws = withDraws;
        return CANVAS_OK;
    };
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    int32_t rc = setup(S == nullptr); if (rc) return rc;
    bool needUpload = true;
    int nrej = 0; uint32_t np = 0;
    int B = 256;
    outcome = 1;
    std::vector<double> px, sx;
    long long pos = rnd.position();
    while (np < nPerm) {
        const int nb = (int)std::min<uint32_t>((uint32_t)B, nPerm - np);
        const long long need = (long long)nb * n;
        const bool cached = S && SC->acquire(S, pos + need, std::max<long long>(2 * need, 16LL << 20));
        MT start(0u);
        if (!cached) {      // the batch's draws: nb * n outputs of the chromosome's generator, in order
            if (!havePinDraws) { rc = setup(true); if (rc) return rc; needUpload = true; }
            MT* m = rnd.host(PG); if (!m) return CANVAS_ERR_HIP;
            start = *m;
            for (size_t t = 0; t < (size_t)need; t++) hDraws[t] = m->u32();
            if (SC) SC->fallbackWords += need;
        } else SC->servedWords += need;
        PermHostReq q;
        memset(q.r.state, 0, sizeof q.r.state); q.r.total = need; q.r.n = n; q.r.nb = nb; q.r.snaps = nullptr; q.r.x = dX; q.r.hk = 0; q.r.al0 = al0; q.r.tss = tss; q.r.errBound = 0.0;
        memset(&q.r.P, 0, sizeof q.r.P); q.r.P.draws = cached ? S->d() + pos : dDraws; q.r.cached = cached ? 1 : 0; q.r.pstat = dStat; q.r.blockBase = 0; q.r.cont = 0; q.r.hist = nullptr; q.r.fy = 2; q.hStat = hStat; q.hSnaps = nullptr;
        q.r.rpBase = 0; q.r.rpWGs = 0; q.r.rpScratch = nullptr; q.r.rpClk = nullptr; memset(&q.r.rp, 0, sizeof q.r.rp);
        if (needUpload) { q.hX = gd; q.dX = dX; q.xBytes = (size_t)n * 8; needUpload = false; }
        if (!cached) { q.hDraws = hDraws; q.drawBytes = (size_t)need * 4; }
        { unsigned sq_ = ++PG.mailSeq; if (sq_ == 0) sq_ = ++PG.mailSeq; q.r.mailStat = hStat; q.r.mailSeq = (unsigned*)hMail; q.r.seq = sq_; *hMail = 0u; std::atomic_thread_fence(std::memory_order_seq_cst); }
        auto tS = std::chrono::steady_clock::now();
        rc = PG.svc->submit(q); if (rc) return rc;
        st.ns_submit += (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tS).count();
        st.dev_batches++;
        auto stop_at = [&](int b) {
            const long long p = pos + (long long)(b + 1) * n;
            if (cached) rnd.jump_to(p);
            else { MT m = start; for (size_t t = 0; t < (size_t)(b + 1) * n; t++) (void)m.u32(); uint32_t s625[625]; m.get_state(s625); rnd.adopt(s625, p); }
        };
        for (int b = 0; b < nb; b++) {
            np++;
            st.perms++; st.perm_elems += n; st.dev_perms++;
            double pstat = hStat[2 * b];
            if (!(pstat == pstat) || cvx_hook("CANVAS_CBS_TEST_VERIFY")) {      // a NaN (degenerate extremes) or the test hook: this permutation again on the host, in the reference's order
                MT m2(0u);
                if (cached) { rc = rnd.at(PG, pos + (long long)b * n, m2); if (rc) return rc; }
                else { m2 = start; for (size_t t = 0; t < (size_t)b * n; t++) (void)m2.u32(); }
                px.resize(n); sx.resize(n);
                xperm(gd, px.data(), n, m2);
                const double exact = tmaxp_host(tss, px.data(), n, sx.data(), al0);
                if (cvx_hook("CANVAS_CBS_TEST_VERIFY")) { st.verified++; if (memcmp(&exact, &pstat, 8) != 0 && (exact == exact || pstat == pstat)) st.violations++; }
                else st.exact_rechecks++;
                pstat = exact;
            }
            if (ostat <= pstat) { nrej++; k++; }
            if (nrej > nrejc) { stop_at(b); outcome = 0; return CANVAS_OK; }
            if (np >= sbdry[k - 1]) { stop_at(b); return CANVAS_OK; }
        }
        if (cached) rnd.jump_to(pos + need);      // (own draws: rnd's generator has produced exactly the batch)
        pos += need;
        B = std::min(maxB, B * 2);
    }
    return CANVAS_OK;
}

#define CBS_GPU_MIN_N 4096

// ChangePoint.FindChangePoints (ChangePoint.cs:291-400) in two phases.
//   phase 1  everything in front of the first random number: centring, TSS, TMaxO, the analytic tail probability of the hybrid test.  It depends on the segment's data
//            only, so the segments a split leaves on the stack are worked on AHEAD of the recursion, by a pool of helper threads (their arc searches and tail series share
//            the launcher rounds of all chromosomes), while the chromosome's own thread is busy with the permutations of the segment on top.
//   phase 2  the permutation reference distribution, the sequential stopping rule and the edge tests: they draw from the chromosome's ONE generator, in the reference's
//            order, on the chromosome's thread.
// Every segment that reaches the stack is processed sooner or later, so nothing is computed that the sequential order would not compute; the results and the number of
// random numbers drawn are the reference's.  (Before: per chromosome a chain arc search -> tail probability -> permutation batches, every link a launcher round trip.)
struct Phase1 {
    int32_t rc = CANVAS_OK; int cn = 0; bool trivial = false;          // fewer than 2 * minWidth values, or constant data: no change point
    std::vector<double> cur, sx; double tss = 0; bool hybrid = false; double delta = 0;
    int iseg[2] = {0, 0}; double ostat = 0, ostat1 = 0; bool stop = false;   // stop: sqrt(ostat) <= 0.1, no change point
    bool bigT = false, exitNoSplit = false; int nrejc = 0;
    bool searched = false;                                  // TMaxO ran
};
static void phase1_run(ArcGpu& G, PermGpu& PG, const double* gd, int cn, uint32_t nPerm, double cutoff, Stats& st, Phase1& P) {
    const int minWidth = 2, kMax = 25; const uint32_t nMin = 200;
    P.cn = cn;
    if (cn < 2 * minWidth) { P.trivial = true; return; }
    P.cur.assign(gd, gd + cn);
    std::vector<double>& cur = P.cur;
    if (nMin < (uint32_t)cn) { P.hybrid = true; P.delta = (kMax + 1.0) / cn; }
    double mx = cur[0], mn = cur[0];
    for (double v : cur) { mx = std::max(mx, v); mn = std::min(mn, v); }
    if (mx == mn) { P.trivial = true; return; }
    double sum = 0; for (double v : cur) sum += v;
    const double avg = sum / cn;
    for (double& v : cur) v -= avg;
    double tss = 0.0; for (double v : cur) tss += 1.0 * v * v;
    P.tss = tss;
    P.sx.resize(cn);
    const int n = cn, al0 = minWidth;
    bool done = false;
    if (n >= CBS_GPU_MIN_N) {
        bool ok = false;
        auto tA = std::chrono::steady_clock::now();
        P.rc = tmaxo_gpu(G, cur.data(), n, tss, P.sx.data(), P.iseg, P.ostat, al0, st, ok); if (P.rc) return;
        st.ns_tmaxo += (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tA).count();
        done = ok;
        if (!ok) st.tie_replays++;
    }
    if (!done) { auto tH = std::chrono::steady_clock::now(); tmaxo_host(cur.data(), n, tss, P.sx.data(), P.iseg, P.ostat, al0); st.ns_tmaxo_host += (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tH).count(); }
    P.searched = true;                                      // (counted when the recursion consumes the result: a helper's guess that is never looked at is not a call of the reference)
    P.ostat1 = std::sqrt(P.ostat); P.ostat *= 0.99999;
    if (P.ostat1 <= 0.1) { P.stop = true; return; }
    const int l = std::min(P.iseg[1] - P.iseg[0], n - P.iseg[1] + P.iseg[0]);
    if (!(P.ostat1 >= 7.0 && l >= 10)) {
        if (P.hybrid) {
            auto tTP = std::chrono::steady_clock::now();
            P.rc = tail_p_decide(PG, P.ostat1, P.delta, n, cutoff, nPerm, st, P.exitNoSplit, P.nrejc); if (P.rc) return;      // TailP(ostat1, delta, n, 100, 1E-6): p1 > cutoff, (int)((cutoff - p1) nPerm)
            st.ns_tailp += (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tTP).count();

        } else P.nrejc = (int)(cutoff * nPerm);
    } else P.bigT = true;
}
static int32_t phase2_run(PermGpu& PG, Phase1& P, uint32_t nPerm, double cutoff, int& nCp, int iCp[2], const std::vector<uint32_t>& sbdry, Rng& rnd, Stats& st) {
    const int minWidth = 2, kMax = 25;
    nCp = 0;
    if (P.rc) return P.rc;
    if (P.searched) { st.tmaxo_calls++; st.tmaxo_elems += P.cn; if (P.bigT) st.big_t++; if (P.hybrid && !P.bigT && P.exitNoSplit) st.tailp_exits++; }
    if (P.trivial || P.stop) return CANVAS_OK;
    const int n = P.cn, al0 = minWidth, hk = kMax; const double* gd = P.cur.data(); const double tss = P.tss, ostat = P.ostat; const bool hybrid = P.hybrid;
    const int* iseg = P.iseg;
    std::vector<double> px(n); std::vector<double>& sx = P.sx;
    int nrej = 0;
    if (!P.bigT) {
        if (hybrid && P.exitNoSplit) return CANVAS_OK;
        const int nrejc = P.nrejc;
        int k = nrejc * (nrejc + 1) / 2 + 1;
        auto t0 = std::chrono::steady_clock::now();
        struct Acc { std::atomic<long long>& a; std::chrono::steady_clock::time_point t; ~Acc() { a += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t).count(); } };
        static const int permGpuMinN = cvx_hook("CANVAS_CBS_PERM_GPU_MIN_N") ? atoi(cvx_hook("CANVAS_CBS_PERM_GPU_MIN_N")) : PERM_GPU_MIN_N;
        if (hybrid && n >= permGpuMinN && hk <= PG_MAXK && cvx_hook("CANVAS_CBS_HOST_PERMUTATIONS") == nullptr) {
            Acc acc{st.ns_dev, t0};
            struct L { std::chrono::steady_clock::time_point t; ~L() { tlClock.dev += std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); tlClock.devLoops++; } } lc{t0};
            int outcome = 1;
            tlClock.loopPerms = 0; tlClock.loopBatches = 0; tlClock.loopComputed = 0;
            int32_t rc = perm_loop_gpu(PG, gd, n, tss, nPerm, hk, al0, ostat, nrejc, k, sbdry, rnd, st, outcome); if (rc) return rc;
            { static const bool logLoops = cvx_hook("CANVAS_CBS_TIMING") && atoi(cvx_hook("CANVAS_CBS_TIMING")) >= 2;
              if (logLoops) fprintf(stderr, "cbs loop: n %d nrejc %d stop-if-no-rejection %u outcome %d seconds %.4f perms %lld batches %lld computed %lld\n", n, nrejc, sbdry[(size_t)k - 1], outcome,
                                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), tlClock.loopPerms, tlClock.loopBatches, tlClock.loopComputed); }
            if (outcome == 0) return CANVAS_OK;
        } else if (!hybrid && n <= 200 && n >= 4 && PG.svc && cvx_hook("CANVAS_CBS_HOST_PERMUTATIONS") == nullptr && cvx_hook("CANVAS_CBS_HOST_SMALL") == nullptr) {
            Acc acc{st.ns_dev, t0};
            struct L { std::chrono::steady_clock::time_point t; ~L() { tlClock.dev += std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); tlClock.devLoops++; } } lc{t0};
            int outcome = 1;
            int32_t rc = perm_loop_small_gpu(PG, gd, n, tss, nPerm, al0, ostat, nrejc, k, sbdry, rnd, st, outcome); if (rc) return rc;
            if (outcome == 0) return CANVAS_OK;
        } else {
            Acc acc{st.ns_hostperm, t0};
            struct L { std::chrono::steady_clock::time_point t; ~L() { tlClock.host += std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); tlClock.hostLoops++; } } lc{t0};
            MT* hm = rnd.host(PG); if (!hm) return CANVAS_ERR_HIP;
            for (uint32_t np = 1; np <= nPerm; np++) {
                xperm(gd, px.data(), n, *hm);
                double pstat = hybrid ? htmaxp_host(hk, tss, px.data(), n, sx.data(), al0) : tmaxp_host(tss, px.data(), n, sx.data(), al0);
                st.perms++; st.perm_elems += n;
                if (ostat <= pstat) { nrej++; k++; }
                if (nrej > nrejc) return CANVAS_OK;
                if (np >= sbdry[k - 1]) break;
            }
        }
    }
    auto tT = std::chrono::steady_clock::now();
    struct TAcc { std::atomic<long long>& a; std::chrono::steady_clock::time_point t; ~TAcc() { a += (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t).count(); } } tAcc{st.ns_tpermp, tT};
    struct LE { std::chrono::steady_clock::time_point t; ~LE() { tlClock.edge += std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); } } le{tT};
    if (iseg[1] == n) { nCp = 1; iCp[0] = iseg[0]; }
    else if (iseg[0] == 0) { nCp = 1; iCp[0] = iseg[1]; }
    else {
        // the edge tests draw on the host (TPermP: one chain of nPerm x m1 dependent swaps): the real generator at the current position
        MT* hm = rnd.host(PG); if (!hm) return CANVAS_ERR_HIP;
        int n1 = iseg[0], n12 = iseg[1], n2 = n12 - n1;
        if (tpermp(PG, n1, n2, n12, gd, 0, px.data(), nPerm, *hm, st) <= cutoff) { nCp = 1; iCp[0] = iseg[0]; }
        int off = iseg[0]; n12 = n - iseg[0]; n2 = n - iseg[1]; n1 = n12 - n2;
        if (tpermp(PG, n1, n2, n12, gd, off, px.data(), nPerm, *hm, st) <= cutoff) { nCp++; iCp[nCp - 1] = iseg[1]; }
    }
    return CANVAS_OK;
}
// the engines' buffers outlive a call: a thread borrows an ArcGpu / PermGpu from the context's cache and hands it back (the buffers only grow)
struct EngineCache {
    struct Slab { char* base = nullptr; size_t bytes = 0, off = 0; };
    std::vector<Slab> devSlabs, pinSlabs; std::vector<hipStream_t> tailStreams; size_t nextTail = 0; std::vector<SvcRes*> svcFree, svcAll;
    std::unique_ptr<MtStreamCache> mts;      // the chromosomes' draw streams (created with the first CBS call / canvas_cbs_prefetch of the context)
    MtStreamCache* streams(canvas_ctx* ctx) { std::lock_guard<std::mutex> lk(mu); if (!mts) mts.reset(new MtStreamCache(ctx)); return mts->off ? nullptr : mts.get(); }
    std::thread streamMaker, warmer; bool dying = false;
    // ahead of the first call (canvas_cbs_prefetch): the launchers' streams and tables and the shared tail streams, on a thread of their own
    void warm(canvas_ctx* ctx, int nSvc) {
        std::lock_guard<std::mutex> lk(mu);
        if (warmer.joinable()) return;
        const int dev = ctx->device; const size_t wantStreams = ctx->one_shot ? 4 : 8;
        warmer = std::thread([this, ctx, dev, nSvc, wantStreams]() {
            if (hipSetDevice(dev) != hipSuccess) { (void)hipGetLastError(); return; }
            std::vector<SvcRes*> made;
            for (int i = 0; i < nSvc; i++) {
                { std::lock_guard<std::mutex> lk2(mu); if (dying) break; }
                SvcRes* r = svc_take();
                (void)svc_res_create(ctx, r, true);      // (a failure leaves a partly made entry: the launcher that takes it completes it, or reports the error, on its own thread)
                made.push_back(r);
            }
            for (SvcRes* r : made) svc_give(r);
            for (;;) {
                { std::lock_guard<std::mutex> lk2(mu); if (tailStreams.size() >= wantStreams || dying) return; }
                hipStream_t q = nullptr; if (hipStreamCreateWithFlags(&q, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return; }
                std::lock_guard<std::mutex> lk2(mu); tailStreams.push_back(q);
            }
        });
    }
    ~EngineCache() {
        { std::lock_guard<std::mutex> lk(mu); dying = true; }
        if (streamMaker.joinable()) streamMaker.join();
        if (warmer.joinable()) warmer.join();
        mts.reset();                                          // (joins the producer thread, unmaps the streams)
        arcs.clear(); perms.clear(); tails.clear();           // (the engines first: they may still wait on a shared stream)
        for (hipStream_t q : tailStreams) { (void)hipStreamSynchronize(q); (void)hipStreamDestroy(q); }
        for (SvcRes* r : svcAll) { if (r->stream) { (void)hipStreamSynchronize(r->stream); (void)hipStreamDestroy(r->stream); } if (r->dReqs) (void)hipFree(r->dReqs); if (r->hReqs) (void)hipHostFree(r->hReqs);
                                   if (r->dArc) (void)hipFree(r->dArc); if (r->hArc) (void)hipHostFree(r->hArc); if (r->dArcP) (void)hipFree(r->dArcP); if (r->hArcP) (void)hipHostFree(r->hArcP); if (r->rpSlab) (void)hipFree(r->rpSlab); delete r; }
        for (Slab& b : devSlabs) (void)hipFree(b.base);
        for (Slab& b : pinSlabs) (void)hipHostFree(b.base);
    }
    // what the call is about to create: newArc arc engines for chromosomes of up to nMax bins, newTail tail engines.  A failure here only means the engines allocate for themselves.
    void reserve(canvas_ctx* ctx, int wantArc, int wantTail, int nMax) {
        std::lock_guard<std::mutex> lk(mu);
        AllocClock ac(g_ns_alloc_svc);
        if (hipSetDevice(ctx->device) != hipSuccess) return;
        // engines in the cache whose buffers are too small for this call will allocate again: count them as new
        int haveArc = 0; for (auto& a : arcs) if (a->cap >= nMax) haveArc++;
        int haveTail = 0; for (auto& t : perms) if (t->tailStream) haveTail++; for (auto& t : tails) if (t->tailStream) haveTail++;
        const int newArc = std::max(0, wantArc - haveArc), newTail = std::max(0, wantTail - haveTail);
        size_t aDev = 0, aPin = 0; ArcGpu::arena_bytes(nMax, aDev, aPin);
        const size_t dev = (size_t)newArc * (aDev + 512) + (size_t)newTail * (128 * 24 + 512), pin = (size_t)newArc * (aPin + 512) + (size_t)newTail * (128 * 24 + 512);
        auto left = [](std::vector<Slab>& v) { return v.empty() ? size_t(0) : v.back().bytes - v.back().off; };
        auto t0 = std::chrono::steady_clock::now(); auto ms = [&]() { auto t = std::chrono::steady_clock::now(); const double v = std::chrono::duration<double, std::milli>(t - t0).count(); t0 = t; return v; };
        if (dev > left(devSlabs)) { Slab b; b.bytes = dev + (1u << 16); if (hipMalloc((void**)&b.base, b.bytes) == hipSuccess) devSlabs.push_back(b); else (void)hipGetLastError(); }
        const double msDev = ms();
        if (pin > left(pinSlabs)) { Slab b; b.bytes = pin + (1u << 16); if (hipHostMalloc((void**)&b.base, b.bytes, hipHostMallocDefault) == hipSuccess) pinSlabs.push_back(b); else (void)hipGetLastError(); }
        const double msPin = ms();
        // the shared tail streams: creating one takes ~5 ms on this runtime (sixteen: 77 of the 170 ms of a cold call) — two are made here, the others on a thread of their
        // own while the chromosome threads start (tail_stream() hands out what exists)
        while (newTail > 0 && tailStreams.size() < 2) { hipStream_t q = nullptr; if (hipStreamCreateWithFlags(&q, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; } tailStreams.push_back(q); }
        // (a one-shot process pays ~5 ms per stream again when it leaves: four shared tail streams there, eight for a host that keeps its context)
        const size_t wantStreams = ctx->one_shot ? 4 : 8;
        if (newTail > 0 && tailStreams.size() < wantStreams && !streamMaker.joinable()) {
            const int dev = ctx->device;
            streamMaker = std::thread([this, dev, wantStreams]() {
                if (hipSetDevice(dev) != hipSuccess) { (void)hipGetLastError(); return; }
                for (;;) {
                    { std::lock_guard<std::mutex> lk2(mu); if (tailStreams.size() >= wantStreams || dying) return; }
                    hipStream_t q = nullptr; if (hipStreamCreateWithFlags(&q, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return; }
                    std::lock_guard<std::mutex> lk2(mu); tailStreams.push_back(q);
                }
            });
        }
        if (cvx_hook("CANVAS_CBS_TIMING")) fprintf(stderr, "cbs arena: device slab of %.1f MB %.1f ms, pinned slab of %.2f MB %.1f ms, %zu tail streams (the others follow in the background) %.1f ms\n", dev / 1e6, msDev, pin / 1e6, msPin, tailStreams.size(), ms());
    }
    char* take(bool pinned, size_t bytes) {
        std::lock_guard<std::mutex> lk(mu);
        std::vector<Slab>& v = pinned ? pinSlabs : devSlabs;
        if (v.empty()) return nullptr;
        Slab& b = v.back(); const size_t at = (b.off + 255) & ~size_t(255);
        if (at + bytes > b.bytes) return nullptr;
        b.off = at + bytes; return b.base + at;
    }
    hipStream_t tail_stream() { std::lock_guard<std::mutex> lk(mu); if (tailStreams.empty()) return nullptr; return tailStreams[nextTail++ % tailStreams.size()]; }
    SvcRes* svc_take() { std::lock_guard<std::mutex> lk(mu); if (!svcFree.empty()) { SvcRes* r = svcFree.back(); svcFree.pop_back(); return r; } SvcRes* r = new SvcRes(); svcAll.push_back(r); return r; }
    void svc_give(SvcRes* r) { std::lock_guard<std::mutex> lk(mu); svcFree.push_back(r); }
    std::mutex mu; std::vector<std::unique_ptr<ArcGpu>> arcs; std::vector<std::unique_ptr<PermGpu>> perms, tails;      // tails: engines of the helper threads (tail series only: they never grow a permutation workspace and must not take one of those away from a chromosome thread)
    static EngineCache& of(canvas_ctx* ctx) {
        static std::mutex g; std::lock_guard<std::mutex> lk(g);
        if (!ctx->cbs_cache) ctx->cbs_cache = std::shared_ptr<void>(new EngineCache(), [](void* p) { delete (EngineCache*)p; });
        return *(EngineCache*)ctx->cbs_cache.get();
    }
    std::unique_ptr<ArcGpu> arc(canvas_ctx* ctx, PermService* svc) { std::unique_ptr<ArcGpu> g; { std::lock_guard<std::mutex> lk(mu); if (!arcs.empty()) { g = std::move(arcs.back()); arcs.pop_back(); } } if (!g) { g.reset(new ArcGpu()); g->ctx = ctx; } g->svc = svc; g->cache = this; return g; }
    std::unique_ptr<PermGpu> perm(canvas_ctx* ctx, PermService* svc) { std::unique_ptr<PermGpu> g; { std::lock_guard<std::mutex> lk(mu); if (!perms.empty()) { g = std::move(perms.back()); perms.pop_back(); } } if (!g) { g.reset(new PermGpu()); g->ctx = ctx; } g->svc = svc; g->cache = this; return g; }
    std::unique_ptr<PermGpu> tail(canvas_ctx* ctx) { std::unique_ptr<PermGpu> g; { std::lock_guard<std::mutex> lk(mu); if (!tails.empty()) { g = std::move(tails.back()); tails.pop_back(); } } if (!g) { g.reset(new PermGpu()); g->ctx = ctx; } g->svc = nullptr; g->cache = this; return g; }
    void give_tail(std::unique_ptr<PermGpu> g) { std::lock_guard<std::mutex> lk(mu); tails.push_back(std::move(g)); }
    void give(std::unique_ptr<ArcGpu> g) { std::lock_guard<std::mutex> lk(mu); arcs.push_back(std::move(g)); }
    void give(std::unique_ptr<PermGpu> g) { std::lock_guard<std::mutex> lk(mu); perms.push_back(std::move(g)); }
};
static char* cache_take(EngineCache* ec, bool pinned, size_t bytes) { return ec ? ec->take(pinned, bytes) : nullptr; }
static hipStream_t cache_tail_stream(EngineCache* ec) { return ec ? ec->tail_stream() : nullptr; }
static SvcRes* cache_svc_take(canvas_ctx* ctx) { return EngineCache::of(ctx).svc_take(); }
static void cache_svc_give(canvas_ctx* ctx, SvcRes* r) { EngineCache::of(ctx).svc_give(r); }
// helper threads of phase 1 (own arc-search and tail-series buffers each); a chromosome thread that needs a segment no helper has started yet runs it itself
#define CBS_SPEC_MIN_N 4
struct SpecTask { const double* gd; int cn; std::atomic<bool> guess{false}; Phase1 out; std::atomic<int> state{0}; };      // 0 queued, 1 running, 2 done; guess: put there by a helper, not (yet) asked for by the recursion
struct SpecPool {
    canvas_ctx* ctx; PermService** arcSvcs; int nArcSvc; uint32_t nPerm; double cutoff; Stats* st; std::atomic_int nextArc{0}; int reserveN = 0;      // reserveN: the call's longest chromosome (the helpers' arc engines are allocated for it)
    std::mutex mu; std::condition_variable cvWork, cvDone; std::deque<std::shared_ptr<SpecTask>> queue; bool stopping = false; std::vector<std::thread> workers;
    // every task of the call by (data pointer, length): a segment is looked up here before anything is computed for it.  Helpers put the segments a finished phase 1 makes
    // LIKELY there as well — its two candidate change points are the arc maximiser's (iseg); if the tests of phase 2 keep both, the children are [0, i0), [i0, i1), [i1, n) —
    // so that the first child's arc search and tail series run while the parent is still permuting, one level ahead of the recursion.  A guess that does not come true (an
    // edge test drops a change point, the permutation test rejects) costs a helper some work and is never looked at.
    std::map<std::pair<const double*, int>, std::shared_ptr<SpecTask>> reg;
    std::atomic<long long> guessed{0}, guessedUsed{0};
    std::shared_ptr<SpecTask> find_or_submit(const double* gd, int cn, bool guess) {
        std::shared_ptr<SpecTask> t;
        { std::shared_ptr<SpecTask> hit; bool promote = false;
          { std::lock_guard<std::mutex> lk(mu);
            auto it = reg.find({gd, cn});
            if (it != reg.end()) { hit = it->second; if (!guess && hit->guess) { hit->guess = false; guessedUsed++; promote = hit->state.load() == 2; } } }
          if (hit) { if (promote && guessAhead) guess_children(*hit); return hit; } }     // a guess that came true and is finished: ITS likely children go out now (one level ahead, never more)
        { std::lock_guard<std::mutex> lk(mu);
          auto it = reg.find({gd, cn});
          if (it != reg.end()) return it->second;
          t = std::make_shared<SpecTask>(); t->gd = gd; t->cn = cn; t->guess = guess; reg[{gd, cn}] = t;
          if (guess) { guessed++; queue.push_back(t); } else queue.push_front(t); }       // what the recursion asks for goes first
        cvWork.notify_one();
        return t;
    }
    void guess_children(const SpecTask& t) {
        const Phase1& P = t.out;
        if (P.rc || P.trivial || P.stop || (P.hybrid && !P.bigT && P.exitNoSplit)) return;
        const int n = P.cn, i0 = P.iseg[0], i1 = P.iseg[1];
        auto go = [&](int a, int b) { if (b - a >= CBS_SPEC_MIN_N) find_or_submit(t.gd + a, b - a, true); };
        if (i1 == n) { go(0, i0); go(i0, n); }
        else if (i0 == 0) { go(0, i1); go(i1, n); }
        else { go(0, i0); go(i0, i1); go(i1, n); }
    }
    void start(int n) {
        for (int i = 0; i < n; i++) workers.emplace_back([this]() {
            EngineCache& cache = EngineCache::of(ctx);
            std::unique_ptr<ArcGpu> gp = cache.arc(ctx, arcSvcs[nextArc++ % nArcSvc]); std::unique_ptr<PermGpu> pgp = cache.tail(ctx);
            gp->reserveN = reserveN;
            struct Back { EngineCache& c; std::unique_ptr<ArcGpu>& a; std::unique_ptr<PermGpu>& p; ~Back() { c.give(std::move(a)); c.give_tail(std::move(p)); } } back{cache, gp, pgp};
            ArcGpu& G = *gp; PermGpu& PG = *pgp;
            for (;;) {
                std::shared_ptr<SpecTask> t;
                { std::unique_lock<std::mutex> lk(mu); cvWork.wait(lk, [&]() { return stopping || !queue.empty(); }); if (stopping || queue.empty()) return; t = queue.front(); queue.pop_front(); }      // (stopping: guesses nobody will read are dropped, not computed)
                int expect = 0;
                if (!t->state.compare_exchange_strong(expect, 1)) continue;          // the chromosome thread took it
                phase1_run(G, PG, t->gd, t->cn, nPerm, cutoff, *st, t->out);
                { std::lock_guard<std::mutex> lk(mu); t->state = 2; }
                cvDone.notify_all();
                if (guessAhead && !t->guess) guess_children(*t);      // only for segments the recursion has asked for: guesses do not breed guesses (2 279 guessed, 358 used when they did)
            }
        });
    }
    bool guessAhead = cvx_hook("CANVAS_CBS_NO_GUESSES") == nullptr;
    // the result of a task: run here if nobody has started it, otherwise wait for the helper
    Phase1& get(const std::shared_ptr<SpecTask>& t, ArcGpu& G, PermGpu& PG) {
        int expect = 0;
        if (t->state.compare_exchange_strong(expect, 1)) { phase1_run(G, PG, t->gd, t->cn, nPerm, cutoff, *st, t->out); { std::lock_guard<std::mutex> lk(mu); t->state = 2; } cvDone.notify_all(); if (guessAhead && !t->guess) guess_children(*t); return t->out; }
        std::unique_lock<std::mutex> lk(mu); cvDone.wait(lk, [&]() { return t->state.load() == 2; });
        return t->out;
    }
    ~SpecPool() { { std::lock_guard<std::mutex> lk(mu); stopping = true; } cvWork.notify_all(); for (auto& w : workers) w.join(); }
};

// ChangePoint.ChangePoints (ChangePoint.cs:44-153), undo = None
static int32_t change_points(ArcGpu& G, PermGpu& PG, SpecPool* pool, const double* gd, int n, const std::vector<uint32_t>& sbdry, Rng& rnd, double alpha, uint32_t nPerm, std::vector<int>& lengthSeg, Stats& st) {
    std::vector<int> segEnd = {0, n}, changeLoc;
    int k = 2, nCp = 0, iCp[2] = {0, 0};
    while (k > 1) {
        const int s0 = segEnd[k - 2], cn = segEnd[k - 1] - s0;
        Phase1 local; Phase1* P = &local;
        std::shared_ptr<SpecTask> hold;
        auto tP1 = std::chrono::steady_clock::now();
        if (pool && cn >= CBS_SPEC_MIN_N) { hold = pool->find_or_submit(gd + s0, cn, false); P = &pool->get(hold, G, PG); }
        else phase1_run(G, PG, gd + s0, cn, nPerm, alpha, st, local);
        tlClock.p1 += std::chrono::duration<double>(std::chrono::steady_clock::now() - tP1).count(); tlClock.segments++;
        int32_t rc = phase2_run(PG, *P, nPerm, alpha, nCp, iCp, sbdry, rnd, st); if (rc) return rc;
        if (nCp == 0) changeLoc.push_back(segEnd[k - 1]);
        for (int i = 0; i < nCp; i++) iCp[i] += s0;
        if (nCp == 0) segEnd.erase(segEnd.begin() + (k - 1));
        else if (nCp == 1) segEnd.insert(segEnd.begin() + (k - 1), iCp[0]);
        else segEnd.insert(segEnd.begin() + (k - 1), iCp, iCp + 2);
        if (nCp > 0 && pool) {
            // the split left nCp + 1 segments where one was: all of them are submitted (the top one is needed next: the chromosome thread takes it itself unless a helper is faster)
            const int first = k - 2;                            // segEnd[first] = s0
            for (int j = first; j <= first + nCp; j++) { const int a = segEnd[j], b = segEnd[j + 1]; if (b - a >= CBS_SPEC_MIN_N) pool->find_or_submit(gd + a, b - a, false); }      // (front of the queue: the last one pushed — the top of the stack — is served first)
        }
        k = (int)segEnd.size();
    }
    std::reverse(changeLoc.begin(), changeLoc.end());
    lengthSeg.clear(); int prev = 0;
    for (int e : changeLoc) { lengthSeg.push_back(e - prev); prev = e; }
    return CANVAS_OK;
}

// ---- undo = SDUndo (ChangePoint.cs:155-196) with the genome-wide trimmed SD (ChangePoint.cs:423-474); host scalar code on a few
// hundred segments.  Normal.InverseCDF / Density are MathNet (parity unpinned): erfc-based here, as in the oracle.
static double qnorm(double p) { double lo = -40, hi = 40; for (int it = 0; it < 200; it++) { double mid = 0.5 * (lo + hi); if (pnorm(mid) < p) lo = mid; else hi = mid; } return 0.5 * (lo + hi); }
static double inflation_factor(double trim) {
    double a = qnorm(1 - trim), step = 2 * a / 10000, from = -a + step / 2, to = a - step / 2, stp = (to - from) / (10000 - 1), e = 0.0, x1 = from;
    for (int i = 0; i < 10000; i++) { double xv = i == 0 ? from : (i == 9999 ? to : (x1 = x1 + stp)); e += (xv * xv) * (std::exp(-0.5 * xv * xv) / std::sqrt(2 * M_PI)); }
    return 1 / (e * step / (1 - 2 * trim));
}
static double trimmed_variance(const double* cov, const int64_t* off, int nchr, double trim) {
    int64_t n = off[nchr];
    std::vector<double> diff((size_t)(n > 0 ? n - 1 : 0));
    int64_t i = 0; double last = 0;
    for (int c = 0; c < nchr; c++) {
        int64_t len = off[c + 1] - off[c];
        if (len <= 0) continue;
        const double* x = cov + off[c];
        if (i > 0) { diff[i] = x[0] - last; i++; }
        for (int64_t t = 0; t + 1 < len; t++) diff[i + t] = x[t + 1] - x[t];
        i += len - 1; last = x[len - 1];
    }
    int nKeep = dn_round(std::nearbyint((1 - 2 * trim) * (n - 1)));
    for (double& d : diff) d = std::fabs(d);
    std::sort(diff.begin(), diff.end());
    double sp = 0.0; for (int t = 0; t < nKeep; t++) sp += sq(diff[t]);
    return inflation_factor(trim) * sp / (2 * nKeep);
}
static double helper_median(const double* x, int a, int b) {
    std::vector<double> y(x + a, x + b); int mid = (int)y.size() / 2;
    std::nth_element(y.begin(), y.begin() + mid, y.end());
    double m = y[mid];
    if (y.size() % 2 == 0) m = (m + *std::max_element(y.begin(), y.begin() + mid)) / 2;
    return m;
}
static void sd_undo(const double* gd, std::vector<int>& lengthSeg, double trimmedSD, double changeSD) {
    if (lengthSeg.size() <= 1) return;
    changeSD *= trimmedSD;
    std::vector<int> cpl(lengthSeg.size()); int acc = 0;
    for (size_t i = 0; i < lengthSeg.size(); i++) { acc += lengthSeg[i]; cpl[i] = acc; }
    for (;;) {
        int k = (int)cpl.size();
        if (k <= 1) break;
        std::vector<double> med(k);
        for (int i = 0; i < k; i++) med[i] = helper_median(gd, i == 0 ? 0 : cpl[i - 1], cpl[i]);
        double mn = std::fabs(med[1] - med[0]); int iMin = 0;
        for (int i = 1; i < k - 1; i++) { double d = std::fabs(med[i + 1] - med[i]); if (d < mn) { mn = d; iMin = i; } }
        if (mn < changeSD) cpl.erase(cpl.begin() + iMin); else break;
    }
    lengthSeg.clear(); int prev = 0;
    for (int e : cpl) { lengthSeg.push_back(e - prev); prev = e; }
}

// ---- undo = Prune (ChangePoint.cs:205-271, Prune.cs): exhaustive search over subsets of the change points for the smallest
// within-segment sum of squares with j change points, j = K-1 .. 1, stopping when it exceeds (1 + cutoff) x the full model's.
// Host scalar code on <= a few dozen segments per chromosome; the number of subsets is capped (the reference would simply not return).
static double error_ssq(const std::vector<int>& len, const std::vector<double>& sum, int k, const std::vector<int>& loc) {
    auto term = [&](int a, int b) { double sx = 0.0; int nx = 0; for (int i = a; i < b; i++) { sx += sum[i]; nx += len[i]; } return std::pow(sx, 2) / nx; };
    double e = 0.0;
    e += term(0, loc[0]);
    for (int j = 1; j < k; j++) e += term(loc[j - 1], loc[j]);
    e += term(loc[k - 1], (int)len.size());
    return e;
}
static int32_t prune(const double* gd, int n, std::vector<int>& lengthSeg, double cutoff, std::string& err) {
    const int nseg = (int)lengthSeg.size(), ncp = nseg - 1;
    if (ncp < 1) return CANVAS_OK;
    // work bound: sum over j of C(ncp, j) subsets, each O(nseg)
    { double total = 0, cmb = 1; for (int j = 1; j <= ncp - 1; j++) { cmb = cmb * (ncp - j + 1) / j; total += cmb; if (total * nseg > 4e9) { err = "CBS -s Prune: too many change points for the exhaustive subset search (ChangePoint.cs:231-246)"; return CANVAS_ERR_UNSUPPORTED; } } }
    std::vector<double> sx(nseg);
    double ssq = 0.0;
    for (int i = 0; i < n; i++) ssq += std::pow(gd[i], 2);
    { int k = 0; for (int i = 0; i < nseg; i++) { double sp = 0.0; for (int t = k; t < k + lengthSeg[i]; t++) sp += std::pow(gd[t], 1); sx[i] = sp; k += lengthSeg[i]; } }
    std::vector<int> loc(ncp), best(ncp), kept(ncp);
    for (int i = 0; i < ncp; i++) { loc[i] = i + 1; kept[i] = i + 1; }
    const double wssqk = ssq - error_ssq(lengthSeg, sx, ncp, loc);
    int pruned = 0;                                   // stays 0 when no j exceeds the cut-off: ONE segment (reference behaviour)
    for (int j = ncp - 1; j > 0; j--) {
        const int kmj = ncp - j;
        for (int i = 0; i < j; i++) { loc[i] = i + 1; best[i] = i + 1; }
        double wssqj = ssq - error_ssq(lengthSeg, sx, j, loc);
        for (bool left = true; left;) {
            int i = j - 1;                            // next combination (Prune.cs:62-72)
            while (loc[i] == kmj + i + 1) i--;
            loc[i]++;
            for (int q = i + 1; q < j; q++) loc[q] = loc[q - 1] + 1;
            if (loc[0] == kmj + 1) left = false;
            const double w1 = ssq - error_ssq(lengthSeg, sx, j, loc);
            if (w1 <= wssqj) { wssqj = w1; for (int q = 0; q < j; q++) best[q] = loc[q]; }
        }
        if (wssqj / wssqk > 1 + cutoff) { pruned = j + 1; for (int q = 0; q < pruned; q++) loc[q] = kept[q]; break; }
        for (int q = 0; q < j; q++) kept[q] = best[q];
    }
    std::vector<int> cum(nseg); { int a = 0; for (int i = 0; i < nseg; i++) { a += lengthSeg[i]; cum[i] = a; } }
    std::vector<int> ends;
    for (int i = 0; i < pruned; i++) ends.push_back(cum[loc[i] - 1]);
    ends.push_back(n);
    lengthSeg.clear(); int prev = 0;
    for (int e : ends) { lengthSeg.push_back(e - prev); prev = e; }
    return CANVAS_OK;
}

}  // namespace cbs

// device-engine counters of the last canvas_cbs call: permutations evaluated on the device / on the host, permutations re-evaluated in
// the reference's exact order because the observed statistic fell inside the rounding interval, device batches; with the test hook
// CANVAS_CBS_TEST_VERIFY=1 also [4] intervals checked against the exact statistic and [5] violations (must be 0)
// host-only: the table of sequential stopping points canvas_cbs works with (GetBoundary.cs:19-157, eta = 0.05 as CBSRunner passes it), evaluated on the host pool
extern "C" int64_t canvas_cbs_boundary(uint32_t nperm, double alpha, uint32_t* h_out, int64_t cap) {
    if (nperm == 0 || !(alpha > 0 && alpha < 1) || !h_out) return CANVAS_ERR_INVALID;
    std::vector<uint32_t> sb;
    cbs::compute_boundary(nperm, alpha, 0.05, sb);
    if ((int64_t)sb.size() > cap) return CANVAS_ERR_CAPACITY;
    for (size_t i = 0; i < sb.size(); i++) h_out[i] = sb[i];
    return (int64_t)sb.size();
}
// host-only: the per-chromosome seeds in file order (CBSRunner.cs:107-112) under the byte convention of include/canvas_mathnet.h
static void cbs_chromosome_seeds(int nchr, int32_t* out) { cbs::MT seeder(0u); for (int c = 0; c < nchr; c++) out[c] = seeder.next_full_range_int32(); }
extern "C" int32_t canvas_cbs_seeds(int32_t nchr, int32_t* h_out, int32_t* h_variant) {
    if (nchr < 0 || (nchr > 0 && !h_out)) return CANVAS_ERR_INVALID;
    cbs_chromosome_seeds(nchr, h_out);
    if (h_variant) *h_variant = canvas_mathnet_seed_variant();
    return CANVAS_OK;
}
// Diagnostic / test entry: ONE batch of nb permutations of the (centred) segment h_x[n] through the device permutation engine, exactly as FindChangePoints' hybrid test
// runs it (XPerm + HTMaxP with hk = 25, al0 = 2; ChangePoint.cs:337-364,407-421; CBSTStatistic.cs:354-586) from a generator seeded with `seed`.  kernel: 0 = k_perm_stat,
// 1 = k_perm_fy.  h_lohi[2 nb]: the interval of every permutation's statistic (the exact value lies inside); h_ms3: milliseconds of the generator's sequential part, of its
// strided part and of the permutation + statistic kernel.  tests/test_cbs_gpu.py compares the intervals with the oracle's XPerm + HTMaxP; tools/perm_probe.py times the kernels.
extern "C" int32_t canvas_cbs_perm_probe(canvas_ctx* ctx, const double* h_x, int32_t n, uint32_t seed, int32_t nb, int32_t kernel, double tss, double* h_lohi, double* h_ms3) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (!h_x || !h_lohi || n < 1024 || nb < 1 || kernel < 0 || kernel > 2 || (long long)n * nb > (1ll << 30) || (kernel == 2 && n > PERM_RP_MAX_N)) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_cbs_perm_probe: bad arguments (n >= 1024, n * nb <= 2^30, kernel 0, 1 or 2 (n <= 524288))");
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    cbs::PermService svc(ctx); svc.probeTiming = true;
    auto al = [](size_t v) { return (v + 255) & ~size_t(255); };
    const size_t e = (size_t)nb * n, e1 = (size_t)nb * (n + 1);
    const size_t oX = 0, oSnaps = oX + al((size_t)n * 8), oStat = oSnaps + al((size_t)nb * 625 * 4), oDraws = oStat + al((size_t)nb * 16), oJ = oDraws + al((e + (size_t)MT_HISTORY) * 4), oOff = oJ + al(e * 4),
                 oCur = oOff + al(kernel == 0 ? e1 * 4 : 0), oItems = oCur + al(kernel == 0 ? e1 * 4 : 0), oG = oItems + al(kernel == 0 ? e * 4 : 0), oSucc = oG + al(kernel == 0 ? e * 4 : 0), oPx = oSucc + al(kernel == 0 ? e * 4 : 0),
                 oSx = oPx + al(kernel == 2 ? 0 : e * 8), oScr = oSx + al(kernel == 0 ? e * 8 : 0);
    PermReq::RpPlan rpP; memset(&rpP, 0, sizeof rpP); int rpWGs = 0;
    if (kernel == 2) { cbs::rp_plan(n, rpP); rpWGs = std::min(cbs::perm_rp_wgs(n), (int)nb); }
    const size_t total = oScr + al((size_t)rpP.stride * 4 * (size_t)rpWGs);
    char* d = nullptr; char* h = nullptr;
    CANVAS_HIP_TRY(ctx, hipMalloc((void**)&d, total));
    struct Free { char*& d; char*& h; ~Free() { if (d) (void)hipFree(d); if (h) (void)hipHostFree(h); } } fr{d, h};
    const size_t pSnaps = al((size_t)n * 8), pStat = pSnaps + al((size_t)nb * 625 * 4);
    CANVAS_HIP_TRY(ctx, hipHostMalloc((void**)&h, pStat + al((size_t)nb * 16), hipHostMallocDefault));
    memcpy(h, h_x, (size_t)n * 8);
    double absSum = 0.0; for (int i = 0; i < n; i++) absSum += std::fabs(h_x[i]);
    cbs::PermHostReq q;
    { cbs::MT m(seed); m.get_state(q.r.state); }
    q.r.total = (long long)nb * n; q.r.n = n; q.r.nb = nb; q.r.snaps = (uint32_t*)(d + oSnaps); q.r.x = (double*)(d + oX); q.r.hk = 25; q.r.al0 = 2; q.r.tss = tss;
    q.r.errBound = 4.04 * (double)(n + 8) * 1.1102230246251565e-16 * absSum;
    q.r.P.draws = (uint32_t*)(d + oDraws) + MT_HISTORY; q.r.P.j = (int32_t*)(d + oJ); q.r.P.off = (int32_t*)(d + oOff); q.r.P.cur = (int32_t*)(d + oCur); q.r.P.items = (int32_t*)(d + oItems);
    q.r.P.g = (int32_t*)(d + oG); q.r.P.succ = (int32_t*)(d + oSucc); q.r.P.px = (double*)(d + oPx); q.r.P.sx = (double*)(d + oSx);
    q.r.pstat = (double*)(d + oStat); q.r.blockBase = 0; q.r.cont = 0; q.r.hist = nullptr; q.r.fy = kernel == 2 ? 3 : kernel; q.hStat = (double*)(h + pStat); q.hSnaps = (uint32_t*)(h + pSnaps);
    q.r.rpBase = 0; q.r.rpWGs = rpWGs; q.r.rpScratch = (uint32_t*)(d + oScr); q.r.rp = rpP; q.r.rpClk = nullptr;
    long long* dClk = nullptr; const bool wantClk = kernel == 2 && cvx_hook("CANVAS_CBS_PROBE_CLOCKS");
    if (wantClk) { CANVAS_HIP_TRY(ctx, hipMalloc((void**)&dClk, 128)); CANVAS_HIP_TRY(ctx, hipMemset(dClk, 0, 128)); q.r.rpClk = dClk; }
    q.hX = (const double*)h; q.dX = (double*)(d + oX); q.xBytes = (size_t)n * 8;
    int32_t rc = svc.submit(q); if (rc) return rc;
    memcpy(h_lohi, q.hStat, (size_t)nb * 16);
    if (dClk) { long long c[16]; (void)hipMemcpy(c, dClk, 128, hipMemcpyDeviceToHost); (void)hipFree(dClk);
        fprintf(stderr, "  own-step blocks in detail: marks + next targets %lld, barrier %lld, classification + independent steps %lld, reservations + stores %lld, slots of the ordered steps %lld, barrier %lld\n", c[8], c[9], c[10], c[11], c[12], c[2]);
        fprintf(stderr, "k_perm_rp workgroup 0 (n %d, %d permutations): cycles range set-up %lld, inbox %lld, own steps: targets + independent steps %lld, ordered replay %lld, last steps (one wave) %lld, statistic: tables %lld, gather %lld, sums + arcs %lld\n",
                n, (nb + rpWGs - 1) / rpWGs, c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]); }
    if (h_ms3) for (int i = 0; i < 3; i++) h_ms3[i] = svc.lastMs[i];
    return CANVAS_OK;
}
// the chromosomes' draw streams ahead of the first call: returns at once, the generator runs on its own thread and stream (cbs::MtStreamCache)
extern "C" int32_t canvas_cbs_prefetch(canvas_ctx* ctx, int32_t nchr, int64_t words_per_chromosome) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (nchr < 0 || nchr > 100000 || words_per_chromosome < 0) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_cbs_prefetch: bad arguments");
    if (nchr == 0 || cvx_hook("CANVAS_CBS_NO_STREAM_CACHE")) return CANVAS_OK;
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    cbs::MtStreamCache* mts = cbs::EngineCache::of(ctx).streams(ctx);
    if (!mts) return CANVAS_OK;
    std::vector<int32_t> seeds((size_t)nchr);
    cbs_chromosome_seeds(nchr, seeds.data());
    for (int c = 0; c < nchr; c++) mts->prefetch(mts->get((uint32_t)seeds[(size_t)c]), words_per_chromosome);
    cbs::EngineCache::of(ctx).warm(ctx, 7);      // (the seven launchers of a call: their streams and request tables, then the shared tail streams)
    return CANVAS_OK;
}
// diagnostic / test entry: Nu(x) of TailProbability.cs:52-85 for up to 100 arguments through the device series (k_tail_nu), with the flags that send a call back to the host series
extern "C" int32_t canvas_cbs_tail_probe(canvas_ctx* ctx, const double* h_x, int32_t n, double tol, double* h_nu, int32_t* h_flag) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (!h_x || !h_nu || !h_flag || n < 1 || n > 100 || !(tol > 0)) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_cbs_tail_probe: 1 .. 100 arguments");
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    cbs::PermGpu PG; PG.ctx = ctx;
    int32_t rc = PG.ensure_tail(); if (rc) return rc;
    double* hNu = (double*)PG.tailPin; int* hFlag = (int*)(hNu + 128); volatile unsigned* hSeq = (volatile unsigned*)(hFlag + 128);
    TailArgs A; memset(&A, 0, sizeof A); for (int i = 0; i < n; i++) A.x[i] = h_x[i];
    const unsigned seq = 1u; *hSeq = 0u; std::atomic_thread_fence(std::memory_order_seq_cst);
    hipLaunchKernelGGL(k_tail_nu, dim3(n), dim3(256), 0, PG.tailStream, A, n, tol, hNu, hFlag, (unsigned*)PG.tailDev, (unsigned*)hSeq, seq);
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(PG.tailStream));
    CANVAS_HIP_TRY(ctx, hipGetLastError());
    rc = cvx_mail_await(ctx, hSeq, seq, "canvas_cbs_tail_probe"); if (rc) return rc;
    for (int i = 0; i < n; i++) { h_nu[i] = hNu[i]; h_flag[i] = hFlag[i]; }
    return CANVAS_OK;
}
// diagnostic / test entry: nwords draws of the chromosome-th stream from `position` on, out of the context's cache (generated now if they are not there yet)
extern "C" int32_t canvas_cbs_stream_read(canvas_ctx* ctx, int32_t chromosome, int64_t position, int64_t nwords, uint32_t* h_out) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (chromosome < 0 || chromosome > 100000 || position < 0 || nwords < 0 || (nwords > 0 && !h_out)) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_cbs_stream_read: bad arguments");
    if (nwords == 0) return CANVAS_OK;
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    cbs::MtStreamCache* mts = cvx_hook("CANVAS_CBS_NO_STREAM_CACHE") ? nullptr : cbs::EngineCache::of(ctx).streams(ctx);
    if (!mts) CANVAS_FAIL(ctx, CANVAS_ERR_UNSUPPORTED, "canvas_cbs_stream_read: the draw-stream cache is switched off (CANVAS_CBS_CACHE_GB=0)");
    std::vector<int32_t> seeds((size_t)chromosome + 1);
    cbs_chromosome_seeds(chromosome + 1, seeds.data());
    cbs::MtStream* S = mts->get((uint32_t)seeds[(size_t)chromosome]);
    if (!S || !mts->acquire(S, position + nwords, 0)) CANVAS_FAIL(ctx, CANVAS_ERR_CAPACITY, "canvas_cbs_stream_read: the cache cannot hold the stream up to that position (CANVAS_CBS_CACHE_GB bounds it)" + (mts->err.empty() ? std::string() : ": " + mts->err));
    CANVAS_HIP_TRY(ctx, hipMemcpy(h_out, S->d() + position, (size_t)nwords * 4, hipMemcpyDeviceToHost));
    return CANVAS_OK;
}
extern "C" int32_t canvas_cbs_cache_stats(canvas_ctx* ctx, int64_t* h_out6) {
    if (!ctx || !h_out6) return CANVAS_ERR_INVALID;
    for (int i = 0; i < 6; i++) h_out6[i] = ctx->cbs_cache_stats[i];
    if (ctx->cbs_cache) {      // what the cache holds NOW (the first four are the last canvas_cbs call's)
        cbs::EngineCache& ec = cbs::EngineCache::of(ctx);
        std::lock_guard<std::mutex> lk(ec.mu);
        if (ec.mts && !ec.mts->off) { std::lock_guard<std::mutex> lk2(ec.mts->mu); h_out6[4] = (int64_t)ec.mts->usedBytes; long long rdy = 0; for (auto& kv : ec.mts->streams) rdy += kv.second->ready; h_out6[5] = rdy; }
    }
    return CANVAS_OK;
}
extern "C" int32_t canvas_cbs_tailp_stats(canvas_ctx* ctx, int64_t* h_out2) {
    if (!ctx || !h_out2) return CANVAS_ERR_INVALID;
    h_out2[0] = ctx->cbs_tailp[0]; h_out2[1] = ctx->cbs_tailp[1];
    return CANVAS_OK;
}
extern "C" int32_t canvas_cbs_device_stats(canvas_ctx* ctx, int64_t* h_out6) {
    if (!ctx || !h_out6) return CANVAS_ERR_INVALID;
    for (int i = 0; i < 6; i++) h_out6[i] = ctx->cbs_dev[i];
    return CANVAS_OK;
}

extern "C" int32_t canvas_cbs_tpermp_stats(canvas_ctx* ctx, int64_t* h_out2) {
    if (!ctx || !h_out2) return CANVAS_ERR_INVALID;
    h_out2[0] = ctx->cbs_tpermp[0]; h_out2[1] = ctx->cbs_tpermp[1];
    return CANVAS_OK;
}
extern "C" int32_t canvas_cbs_undo(canvas_ctx* ctx, int32_t nchr, const double* d_cov, const int64_t* h_chr_offset, double alpha, uint32_t nperm,
                                   int32_t undo, double undo_sd, int32_t* d_seg_len, int32_t* h_nseg, int64_t* h_stats);
extern "C" int32_t canvas_cbs(canvas_ctx* ctx, int32_t nchr, const double* d_cov, const int64_t* h_chr_offset, double alpha, uint32_t nperm,
                              int32_t* d_seg_len, int32_t* h_nseg, int64_t* h_stats) {
    return canvas_cbs_undo(ctx, nchr, d_cov, h_chr_offset, alpha, nperm, 0, 3.0, d_seg_len, h_nseg, h_stats);
}
extern "C" int32_t canvas_cbs_undo(canvas_ctx* ctx, int32_t nchr, const double* d_cov, const int64_t* h_chr_offset, double alpha, uint32_t nperm,
                                   int32_t undo, double undo_sd, int32_t* d_seg_len, int32_t* h_nseg, int64_t* h_stats) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (nchr <= 0 || !h_chr_offset || !d_seg_len || !h_nseg) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_cbs: bad arguments");
    std::vector<std::vector<int>> segs;
    int32_t rc = cvx_cbs_masked(ctx, nchr, d_cov, h_chr_offset, alpha, nperm, undo, undo_sd, nullptr, segs, h_stats); if (rc) return rc;
    const int64_t N = h_chr_offset[nchr];
    std::vector<int32_t> flat((size_t)N + 1, 0);
    for (int c = 0; c < nchr; c++) { h_nseg[c] = (int32_t)segs[c].size(); for (size_t i = 0; i < segs[c].size(); i++) flat[h_chr_offset[c] + i] = segs[c][i]; }
    if (N > 0) CANVAS_HIP_TRY(ctx, hipMemcpyAsync(d_seg_len, flat.data(), (size_t)N * 4, hipMemcpyHostToDevice, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CANVAS_OK;
}
// the segmentation itself; h_mask (optional) = the chromosomes to segment (canvas_cbs_sharded: the ones this rank owns).  Everything genome-wide — the seeds, drawn for every
// chromosome in file order (CBSRunner.cs:107-112), and the trimmed SD of SDUndo (CBSRunner.cs:102) — is computed from the whole coverage whatever the mask says.
int32_t cvx_cbs_masked(canvas_ctx* ctx, int32_t nchr, const double* d_cov, const int64_t* h_chr_offset, double alpha, uint32_t nperm, int32_t undo, double undo_sd,
                       const uint8_t* h_mask, std::vector<std::vector<int>>& segs, int64_t* h_stats) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (undo != 0 && undo != 1 && undo != 2) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "undo must be 0 (None), 1 (Prune) or 2 (SDUndo)");
    if (nchr <= 0 || !d_cov || !h_chr_offset || nperm == 0 || !(alpha > 0 && alpha < 1)) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_cbs: bad arguments");
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const auto tEntry = std::chrono::steady_clock::now(); double lapLast = 0; std::string laps;
    auto lap = [&](const char* name) { const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - tEntry).count(); char b[96]; snprintf(b, sizeof b, "%s %.1f ms, ", name, (t - lapLast) * 1e3); laps += b; lapLast = t; };
    const int64_t N = h_chr_offset[nchr];
    std::vector<double> cov((size_t)N);
    if (N > 0) CANVAS_HIP_TRY(ctx, hipMemcpyAsync(cov.data(), d_cov, (size_t)N * 8, hipMemcpyDeviceToHost, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int64_t i = 0; i < N; i++) if (!std::isfinite(cov[i])) CANVAS_FAIL(ctx, CANVAS_ERR_UNSUPPORTED, "canvas_cbs: non-finite coverage (CBSRunner.cs:63-89 filter) is not built");
    lap("coverage to the host");
    // sequential stopping boundary (GetBoundary.cs), cached per (nperm, alpha)
    static std::mutex bmu; static std::vector<uint32_t> sb; static uint32_t sbN = 0; static double sbA = 0;
    std::vector<uint32_t> sbdry;
    { std::lock_guard<std::mutex> lk(bmu); auto tB = std::chrono::steady_clock::now(); const bool fresh = sbN != nperm || sbA != alpha;
      if (fresh) { cbs::compute_boundary(nperm, alpha, 0.05, sb); sbN = nperm; sbA = alpha; } sbdry = sb;
      if (fresh && cvx_hook("CANVAS_CBS_TIMING")) fprintf(stderr, "cbs: sequential boundary table (GetBoundary.cs) computed in %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - tB).count()); }
    // per-chromosome seeds in file order (CBSRunner.cs:107-112)
    std::vector<int32_t> seeds(nchr);
    cbs_chromosome_seeds(nchr, seeds.data());
    double trimmedSD = 1.0;
    if (undo == 2 && N > 1) trimmedSD = std::sqrt(cbs::trimmed_variance(cov.data(), h_chr_offset, nchr, 0.025));   // CBSRunner.cs:102
    cbs::Stats st;
    segs.assign((size_t)nchr, std::vector<int>());
    std::vector<int32_t> rcs(nchr, 0);
    std::vector<std::string> errs(nchr);
    std::atomic_int next{0};
    long long nMax = 0; for (int c = 0; c < nchr; c++) if (!h_mask || h_mask[c]) nMax = std::max<long long>(nMax, h_chr_offset[c + 1] - h_chr_offset[c]);
    // launcher threads (own streams): arc searches on one, permutation batches spread over four, so that the device always has several
    // independent kernels in flight (a batch of one chromosome is a chain of latency-bound launches)
    cbs::PermService arcService(ctx), arcService1(ctx), arcService2(ctx), service(ctx), service1(ctx), service2(ctx), service3(ctx);
    const int nPermSvc = std::max(1, std::min(16, cvx_hook("CANVAS_CBS_PERM_SERVICES") ? atoi(cvx_hook("CANVAS_CBS_PERM_SERVICES")) : 4));
    std::vector<std::unique_ptr<cbs::PermService>> moreServices;
    for (int i = 4; i < nPermSvc; i++) moreServices.emplace_back(new cbs::PermService(ctx));
    const int nArcSvc = std::max(1, std::min(16, cvx_hook("CANVAS_CBS_ARC_SERVICES") ? atoi(cvx_hook("CANVAS_CBS_ARC_SERVICES")) : 3));
    std::vector<std::unique_ptr<cbs::PermService>> moreArc;
    for (int i = 3; i < nArcSvc; i++) moreArc.emplace_back(new cbs::PermService(ctx));
    cbs::PermService* arcServices[16] = {&arcService, &arcService1, &arcService2};      // (one launcher synchronises after every round: requests that arrive meanwhile would wait a whole round)
    for (int i = 3; i < nArcSvc; i++) arcServices[i] = moreArc[(size_t)i - 3].get();
    std::atomic_int nextArc{0};
    std::vector<cbs::PermService*> permServices = {&service, &service1, &service2, &service3};
    for (auto& m : moreServices) permServices.push_back(m.get());
    permServices.resize((size_t)nPermSvc);
    std::atomic_int nextService{0};
    for (auto* sv : permServices) sv->rpSlabWant = cbs::perm_rp_scratch_bytes((size_t)std::min<long long>(nMax, 0x7FFFFFF0ll));
    std::mutex chromMu; double maxChromSec = 0, sumChromSec = 0, slowSec = 0; std::string slowLine; const bool timing = cvx_hook("CANVAS_CBS_TIMING") != nullptr;
    // helper threads for the deterministic front half of every segment on a recursion stack (cbs::SpecPool); CANVAS_CBS_NO_SPECULATION=1: the plain sequential order (test hook)
    std::unique_ptr<cbs::SpecPool> specPool;
    unsigned hw = std::thread::hardware_concurrency(); if (hw == 0) hw = 4;
    const int nthreads = (int)std::min<unsigned>(hw, (unsigned)nchr);
    const int nHelpers = cvx_hook("CANVAS_CBS_NO_SPECULATION") ? 0 : (int)std::min<unsigned>(32u, std::max(4u, std::thread::hardware_concurrency() / 4));
    // one device slab, one pinned slab and the shared tail streams for every engine this call is about to create (chromosome threads + helpers): cbs::EngineCache
    lap("launchers");
    if (!cvx_hook("CANVAS_CBS_NO_ARENA")) cbs::EngineCache::of(ctx).reserve(ctx, nthreads + nHelpers, nthreads + nHelpers, (int)std::min<long long>(nMax, 0x7FFFFFF0ll));
    lap("arena");
    if (nHelpers) { specPool.reset(new cbs::SpecPool{ctx, arcServices, nArcSvc, nperm, alpha, &st}); specPool->reserveN = (int)nMax; specPool->start(nHelpers); }
    size_t perEngineBudget = ~size_t(0);
    { size_t freeB = 0, totB = 0; if (hipMemGetInfo(&freeB, &totB) == hipSuccess) perEngineBudget = (freeB / 2) / (size_t)std::max(1, nthreads); }
    // the chromosomes' draw streams: every chromosome that will permute asks for the first extension of its stream NOW, so that the generator runs while the first arc
    // searches do (in a process that called canvas_cbs_prefetch — the executable does, during its file read — the words are there already)
    cbs::MtStreamCache* mts = cvx_hook("CANVAS_CBS_NO_STREAM_CACHE") ? nullptr : cbs::EngineCache::of(ctx).streams(ctx);
    long long mtsBefore[4] = {0, 0, 0, 0};
    if (mts) {
        mtsBefore[0] = mts->servedWords; mtsBefore[1] = mts->fallbackWords; mtsBefore[2] = mts->generatedWords; mtsBefore[3] = mts->fetches;
        for (int c = 0; c < nchr; c++) { const long long n = h_chr_offset[c + 1] - h_chr_offset[c]; if (n >= 4 && (!h_mask || h_mask[c])) mts->prefetch(mts->get((uint32_t)seeds[c]), std::max<long long>(MTS_FIRST_WORDS, 32 * n)); }
    }
    auto work = [&]() {
        cbs::EngineCache& cache = cbs::EngineCache::of(ctx);                             // per thread: own buffers, borrowed from the context's cache (created on first use)
        std::unique_ptr<cbs::PermGpu> pgp = cache.perm(ctx, permServices[(size_t)(nextService++ % nPermSvc)]); std::unique_ptr<cbs::ArcGpu> gp = cache.arc(ctx, arcServices[nextArc++ % nArcSvc]);
        struct Back { cbs::EngineCache& c; std::unique_ptr<cbs::ArcGpu>& a; std::unique_ptr<cbs::PermGpu>& p; ~Back() { c.give(std::move(a)); c.give(std::move(p)); } } back{cache, gp, pgp};
        cbs::PermGpu& PG = *pgp; cbs::ArcGpu& G = *gp;
        G.reserveN = (int)nMax;
        // the first allocation of an engine is made for the call's longest chromosome (growing later means hipFree + hipMalloc, which stall every stream of the device) —
        // as long as all engines of the call together stay inside half of what the device has free: many contigs on a many-core host, or several contexts on one GPU,
        // would otherwise run out of memory where grow-on-demand engines fit
        PG.reserveN = (size_t)nMax; cbs::perm_reserve_bytes((size_t)nMax, mts == nullptr, PG.reserveBytes, PG.reservePin);      // (with cache streams no engine holds draw buffers: 0.5 GB each)
        if (PG.reserveBytes > perEngineBudget) { PG.reserveBytes = 0; PG.reservePin = 0; PG.reserveN = 0; }       // the engines then grow on demand
        for (;;) {
            int c = next++; if (c >= nchr) break;
            int n = (int)(h_chr_offset[c + 1] - h_chr_offset[c]);
            if (n <= 0 || (h_mask && !h_mask[c])) continue;
            cbs::Rng rnd; rnd.init((uint32_t)seeds[c], mts);
            auto tC = std::chrono::steady_clock::now();
            struct CAcc { std::mutex& m; double& mx; double& sm; std::chrono::steady_clock::time_point t; ~CAcc() { double d = std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); std::lock_guard<std::mutex> lk(m); mx = std::max(mx, d); sm += d; } } cAcc{chromMu, maxChromSec, sumChromSec, tC};
            cbs::tlClock = cbs::ChromClock();
            rcs[c] = cbs::change_points(G, PG, specPool.get(), cov.data() + h_chr_offset[c], n, sbdry, rnd, alpha, nperm, segs[c], st);
            if (timing) { const double d = std::chrono::duration<double>(std::chrono::steady_clock::now() - tC).count(); const cbs::ChromClock& k = cbs::tlClock; std::lock_guard<std::mutex> lk(chromMu);
                if (d > slowSec) { slowSec = d; char b[400]; snprintf(b, sizeof b, "slowest chromosome %d (%d bins): %.3f s = phase 1 %.3f + device permutation loops %.3f (%d) + host permutation loops %.3f (%d) + edge tests %.3f; %d segments tested", c, n, d, k.p1, k.dev, k.devLoops, k.host, k.hostLoops, k.edge, k.segments); slowLine = b; } }
            if (rcs[c] == 0 && undo == 2) cbs::sd_undo(cov.data() + h_chr_offset[c], segs[c], trimmedSD, undo_sd);
            if (rcs[c] == 0 && undo == 1 && segs[c].size() > 1) rcs[c] = cbs::prune(cov.data() + h_chr_offset[c], n, segs[c], 0.05, errs[c]);   // undoPrune = 0.05 (CBSRunner.cs:42)
        }
    };
    lap("helpers + draw streams asked for");
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) th.emplace_back(work);
    for (auto& t : th) t.join();
    if (mts) mts->quiesce();
    lap("chromosome threads");
    if (timing) fprintf(stderr, "cbs call phases: %s\n", laps.c_str());
    for (int c = 0; c < nchr; c++) if (rcs[c]) { if (!errs[c].empty()) ctx->err = errs[c]; return rcs[c]; }
    ctx->cbs_tpermp[0] = st.tpermp_device; ctx->cbs_tpermp[1] = st.tpermp_draws;
    ctx->cbs_dev[0] = st.dev_perms; ctx->cbs_dev[1] = st.perms - st.dev_perms; ctx->cbs_dev[2] = st.exact_rechecks; ctx->cbs_dev[3] = st.dev_batches; ctx->cbs_dev[4] = st.verified; ctx->cbs_dev[5] = st.violations;
    ctx->cbs_tailp[0] = st.tailp_dev; ctx->cbs_tailp[1] = st.tailp_host;
    for (int i = 0; i < 6; i++) ctx->cbs_cache_stats[i] = 0;
    if (mts) {
        ctx->cbs_cache_stats[0] = mts->servedWords - mtsBefore[0]; ctx->cbs_cache_stats[1] = mts->fallbackWords - mtsBefore[1]; ctx->cbs_cache_stats[2] = mts->generatedWords - mtsBefore[2]; ctx->cbs_cache_stats[3] = mts->fetches - mtsBefore[3];
        std::lock_guard<std::mutex> lk(mts->mu); ctx->cbs_cache_stats[4] = (long long)mts->usedBytes; long long rdy = 0; for (auto& kv : mts->streams) rdy += kv.second->ready; ctx->cbs_cache_stats[5] = rdy;
    }
    if (timing && mts) fprintf(stderr, "cbs draw streams: %lld words read out of the cache, %lld generated inside batches (no stream / bound reached), %lld generated by the cache's producer in this call, %lld states fetched for host code; %.2f GB mapped, %lld words held\n",
                               ctx->cbs_cache_stats[0], ctx->cbs_cache_stats[1], ctx->cbs_cache_stats[2], ctx->cbs_cache_stats[3], ctx->cbs_cache_stats[4] / 1e9, ctx->cbs_cache_stats[5]);
    if (timing && mts) fprintf(stderr, "cbs draw streams (since the context was created): producer %lld rounds, %.3f s mapping memory, %.3f s generating; batches waited for the producer %lld times, %.3f thread-seconds\n",
                               (long long)mts->rounds, mts->nsMap * 1e-9, mts->nsGen * 1e-9, (long long)mts->waits, mts->nsWaited * 1e-9);
    if (timing) fprintf(stderr, "cbs allocation / stream creation, thread-seconds: arc engines %.3f, permutation engines %.3f, tail engines %.3f, launchers %.3f\n", cbs::g_ns_alloc_arc.exchange(0) * 1e-9, cbs::g_ns_alloc_perm.exchange(0) * 1e-9, cbs::g_ns_alloc_tail.exchange(0) * 1e-9, cbs::g_ns_alloc_svc.exchange(0) * 1e-9);
    if (timing) fprintf(stderr, "cbs %s\n", slowLine.c_str());
    if (timing) fprintf(stderr, "cbs arc searches whose best admissible arc the reference does not scan (replayed on the host): %lld\n", (long long)st.unscanned_max.load());
    if (timing && specPool) fprintf(stderr, "cbs helpers: %lld segments guessed from a finished phase 1, %lld of them asked for by the recursion\n", (long long)specPool->guessed.load(), (long long)specPool->guessedUsed.load());
    if (cvx_hook("CANVAS_CBS_TIMING")) fprintf(stderr, "cbs thread-seconds: TMaxO on the host %.3f, TailP %.3f (%lld decided from the device series, %lld by the host series)\n", st.ns_tmaxo_host.load() * 1e-9, st.ns_tailp.load() * 1e-9, (long long)st.tailp_dev.load(), (long long)st.tailp_host.load()),
                                      fprintf(stderr, "cbs launcher: %lld rounds, %lld arc searches in %.3f s, %lld permutation batches in %.3f s\n", service.rounds + arcService.rounds + arcService1.rounds + arcService2.rounds, arcService.nArc + arcService1.nArc + arcService2.nArc, std::max(arcService.secArc, std::max(arcService1.secArc, arcService2.secArc)), service.nPermReq + service1.nPermReq + service2.nPermReq + service3.nPermReq, std::max(std::max(service.secPerm, service1.secPerm), std::max(service2.secPerm, service3.secPerm))),
                                      fprintf(stderr, "cbs thread-seconds: TMaxO on the device incl. waiting %.3f, edge tests (TPermP) %.3f; per-chromosome wall max %.3f sum %.3f; ", st.ns_tmaxo.load() * 1e-9, st.ns_tpermp.load() * 1e-9, maxChromSec, sumChromSec),
                                      fprintf(stderr, "device permutation loop %.3f (buffers %.3f, uploads %.3f, waiting for the launcher %.3f, stopping rule %.3f), host permutation loop %.3f\n", st.ns_dev.load() * 1e-9, st.ns_ensure.load() * 1e-9, st.ns_upload.load() * 1e-9, st.ns_submit.load() * 1e-9, st.ns_post.load() * 1e-9, st.ns_hostperm.load() * 1e-9);
    if (h_stats) { h_stats[0] = st.tmaxo_calls; h_stats[1] = st.tmaxo_elems; h_stats[2] = st.perms; h_stats[3] = st.perm_elems; h_stats[4] = st.tpermp_draws; h_stats[5] = st.tailp_exits; h_stats[6] = st.gpu_searches; h_stats[7] = st.tie_replays; }
    return CANVAS_OK;
}
