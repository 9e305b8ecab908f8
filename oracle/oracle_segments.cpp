// TEST INFRASTRUCTURE ONLY (see oracle_common.h). Segment derivation / post-processing restatement and the extern "C"
// surface of the oracle (loaded by tests/ via ctypes). Paths relative to /root/reference/Src/Canvas/.
#include "oracle_common.h"
#include "oracle_api.h"
#include "oracle_partition.h"
#include <map>
#include <set>
#include <thread>
#include <atomic>

namespace oracle {

// SegmentationInput.DeriveSegments (CanvasPartition/Segmentation.cs:83-125): breakpoints (bin indices) -> genomic segments
static int DeriveSegments(std::vector<int> breakpoints, int segmentsLength, const uint32_t* startByChr, const uint32_t* endByChr,
                          uint32_t* segStart, uint32_t* segEnd) {
    std::vector<int> sb, eb;
    if (breakpoints.size() >= 2 && segmentsLength > 10) {
        if (breakpoints[0] != 0) breakpoints.insert(breakpoints.begin(), 0);
        sb.push_back(breakpoints[0]);
        eb.push_back(breakpoints[1] - 1);
        for (size_t i = 1; i + 1 < breakpoints.size(); i++) { sb.push_back(breakpoints[i]); eb.push_back(breakpoints[i + 1] - 1); }
        sb.push_back(breakpoints.back());
        eb.push_back(segmentsLength - 1);
    } else { sb.push_back(0); eb.push_back(segmentsLength - 1); }
    for (size_t i = 0; i < sb.size(); i++) { segStart[i] = startByChr[sb[i]]; segEnd[i] = endByChr[eb[i]]; }
    return (int)sb.size();
}

// GenomeSegmentationResults.SplitOverlappingSegments (CanvasPartition/GenomeSegmentationResults.cs:35-55), one chromosome
static int SplitOverlapping(int nSamples, const uint32_t* const* starts, const uint32_t* const* ends, const int* nseg,
                            uint32_t* outStart, uint32_t* outEnd, int cap) {
    std::vector<std::pair<uint32_t, int>> ev;  // (position, isStart ? 0 : 1) -- starts before ends on ties (irrelevant to output)
    for (int s = 0; s < nSamples; s++) for (int i = 0; i < nseg[s]; i++) { ev.push_back({starts[s][i], 0}); ev.push_back({ends[s][i], 1}); }
    std::stable_sort(ev.begin(), ev.end(), [](auto& a, auto& b) { return a.first < b.first; });
    int overlapping = 0, n = 0;
    uint32_t cur = 0;
    for (auto& e : ev) {
        if (overlapping > 0 && cur != e.first) { if (n < cap) { outStart[n] = cur; outEnd[n] = e.first; } n++; }
        cur = e.first;
        overlapping += (e.second == 0) ? 1 : -1;
    }
    return n;
}

}  // namespace oracle

using namespace oracle;

extern "C" {

double orc_bin_rate(const uint8_t* hits, const uint8_t* mask, int64_t len) { return bin_rate(hits, mask, len); }
int orc_bin_size(const double* rates, int n, int countsPerBin) { return bin_size_from_rates(rates, n, countsPerBin); }
int64_t orc_bin_chromosome(const uint8_t* bases, const uint8_t* mask, const uint8_t* hits, int64_t len, int binSize, int mode,
                           int64_t cap, int32_t* start, int32_t* stop, int32_t* gc, int32_t* count) {
    return bin_chromosome(bases, mask, hits, len, binSize, mode, cap, start, stop, gc, count);
}
int64_t orc_bin_chromosome_predefined(const uint8_t* bases, const uint8_t* mask, const uint8_t* hits, int64_t len, int mode, int64_t nbins, const int32_t* binStart, const int32_t* binStop,
                                      int32_t* gc, int32_t* count) {
    return bin_chromosome_predefined(bases, mask, hits, len, mode, nbins, binStart, binStop, gc, count);
}
int64_t orc_bin_chromosome_predefined_weighted(const uint8_t* bases, const uint8_t* mask, const uint8_t* hits, const uint8_t* readGC, const float* obsVsExp, int64_t len, int64_t nbins,
                                               const int32_t* binStart, const int32_t* binStop, int32_t* gc, int32_t* count) {
    return bin_chromosome_predefined_weighted(bases, mask, hits, readGC, obsVsExp, len, nbins, binStart, binStop, gc, count);
}
// multi-threaded helper for the CPU baseline: one std::thread per chromosome, as Parallel.ForEach in CanvasBin.cs:539
void orc_bin_genome(int nchr, const uint8_t* const* bases, const uint8_t* const* mask, const uint8_t* const* hits, const int64_t* len,
                    int binSize, int mode, const int64_t* cap, int32_t* const* start, int32_t* const* stop, int32_t* const* gc,
                    int32_t* const* count, int64_t* nbins, int threads) {
    std::vector<std::thread> th;
    std::atomic_int next{0};
    auto work = [&]() {
        for (;;) { int c = next++; if (c >= nchr) break;
            nbins[c] = bin_chromosome(bases[c], mask[c], hits[c], len[c], binSize, mode, cap[c], start[c], stop[c], gc[c], count[c]); }
    };
    for (int t = 0; t < std::max(1, threads); t++) th.emplace_back(work);
    for (auto& t : th) t.join();
}
void orc_bin_rates_genome(int nchr, const uint8_t* const* mask, const uint8_t* const* hits, const int64_t* len, double* rates, int threads) {
    std::vector<std::thread> th;
    std::atomic_int next{0};
    auto work = [&]() { for (;;) { int c = next++; if (c >= nchr) break; rates[c] = bin_rate(hits[c], mask[c], len[c]); } };
    for (int t = 0; t < std::max(1, threads); t++) th.emplace_back(work);
    for (auto& t : th) t.join();
}

int orc_mean_fragment_size(int nchr, const int16_t* const* fl, const int64_t* len) { return mean_fragment_size(nchr, fl, len); }
void orc_read_gc_content(const uint8_t* bases, const int16_t* fl, int64_t L, int meanFrag, uint8_t* out) { read_gc_content(bases, fl, L, meanFrag, out); }
void orc_observed_vs_expected_gc(int nchr, const uint8_t* const* readGC, const uint8_t* const* hits, const int64_t* len, float* out101) { observed_vs_expected_gc(nchr, readGC, hits, len, out101); }
int64_t orc_bin_chromosome_weighted(const uint8_t* bases, const uint8_t* mask, const uint8_t* hits, const uint8_t* readGC, const float* w, int64_t len, int binSize,
                                    int64_t cap, int32_t* start, int32_t* stop, int32_t* gc, int32_t* count) {
    return bin_chromosome_weighted(bases, mask, hits, readGC, w, len, binSize, cap, start, stop, gc, count);
}
int64_t orc_clean(int64_t n, int32_t* chr, int32_t* start, int32_t* stop, float* count, int32_t* gc, int nchr,
                  const uint8_t* isAuto, const uint8_t* isY, uint32_t flags, int minBinsWeighted, double* localSdOut, int32_t* stageCounts) {
    return clean(n, chr, start, stop, count, gc, nchr, isAuto, isY, flags, minBinsWeighted, localSdOut, stageCounts);
}

static int put(const std::string& s, char* buf, int cap) { int n = (int)s.size(); if (n + 1 <= cap) memcpy(buf, s.c_str(), n + 1); return n; }
int orc_format_f2(float v, char* buf, int cap) { return put(format_float_f2(v), buf, cap); }
int orc_format_g15(double v, char* buf, int cap) { return put(format_double_g15(v), buf, cap); }
int orc_format_g7(float v, char* buf, int cap) { return put(format_float_g7(v), buf, cap); }

void orc_quartiles(const float* x, int n, float* out3) { std::vector<float> v(x, x + n); Quartiles(v, out3[0], out3[1], out3[2]); }
float orc_median_f32(const float* x, int n) { std::vector<float> v(x, x + n); return sorted_median(v); }
// Utilities.MedianFilter semantics are pinned through the window medians (CanvasTest/TestUtilities.cs:195-206)
void orc_loess_fit(const double* x, const double* y, int n, double bw, int rob, double xStep, double* fitted, double* predicted) {
    loess_fit(x, y, n, bw, rob, xStep, fitted, predicted);
}
double orc_golden_section_square(double a, double b) { return golden_section_square(a, b); }

void orc_negbin(double mean, double variance, int maxValue, double* out) {
    auto d = NegativeBinomialWrapper(mean, variance, maxValue);
    std::copy(d.begin(), d.end(), out);
}
int orc_genotype_combos(int nStates, int cur, int* out, int cap) {
    auto c = GetGenotypeCombinations(nStates, cur);
    int k = 0;
    for (auto& v : c) for (int s : v) { if (k < cap) out[k] = s; k++; }
    return (int)c.size();
}

void orc_hmm_global_params(int nchr, const double* const* cov, const int64_t* n, double* median, double* pv) { hmm_global_params(nchr, cov, n, median, pv); }
int orc_hmm_chromosome(int nSamples, int perSample, const double* const* cov, int T, const double* medians, const double* pvs, int32_t* path) {
    return hmm_chromosome(nSamples, perSample != 0, cov, T, medians, pvs, path);
}
// whole-genome PerSampleHMM state paths, one std::thread per chromosome (HiddenMarkovModelsRunner.cs:51-58)
void orc_hmm_genome_per_sample(int nchr, const double* const* cov, const int64_t* n, int32_t* const* path, int32_t* ran, int threads) {
    double med, pv;
    hmm_global_params(nchr, cov, n, &med, &pv);
    std::vector<std::thread> th;
    std::atomic_int next{0};
    auto work = [&]() { for (;;) { int c = next++; if (c >= nchr) break; const double* p = cov[c]; ran[c] = hmm_chromosome(1, true, &p, (int)n[c], &med, &pv, path[c]); } };
    for (int t = 0; t < std::max(1, threads); t++) th.emplace_back(work);
    for (auto& t : th) t.join();
}
// breakpoints where the state changes (HiddenMarkovModelsRunner.cs:88-95) -> DeriveSegments
int orc_segments_from_path(const int32_t* path, int T, int ran, const uint32_t* start, const uint32_t* end, uint32_t* segStart, uint32_t* segEnd) {
    if (!ran) return 0;  // chromosome skipped: no entry in segmentByChr
    std::vector<int> bp = {0};
    for (int i = 1; i < T; i++) if (path[i] - path[i - 1] != 0) bp.push_back(i);
    return DeriveSegments(bp, T, start, end, segStart, segEnd);
}
int orc_derive_segments(const int* breakpoints, int nb, int T, const uint32_t* start, const uint32_t* end, uint32_t* segStart, uint32_t* segEnd) {
    return DeriveSegments(std::vector<int>(breakpoints, breakpoints + nb), T, start, end, segStart, segEnd);
}
int orc_split_overlapping(int nSamples, const uint32_t* const* starts, const uint32_t* const* ends, const int* nseg, uint32_t* outStart, uint32_t* outEnd, int cap) {
    return SplitOverlapping(nSamples, starts, ends, nseg, outStart, outEnd, cap);
}

// PloidyInfo.getPloidyCounts + IsUniformReferencePloidy (CanvasCommon/PloidyInfo.cs:78-110) for the query interval [oneBasedStart, oneBasedEnd]
// (Isas.SequencingFiles.Interval is not in /root/reference: one-based inclusive, Length = end - start + 1 — assumption, see oracle_common.h).
// Intervals are the chromosome's ploidy.vcf records (one-based Start = POS, End = INFO/END, Ploidy = CN) in file order.
// Returns 1 when uniform, 0 when not, -1 when a ploidy outside 0..4 would index past baseCounts (the C# throws IndexOutOfRangeException).
static int IsUniformReferencePloidy(int64_t oneBasedStart, int64_t oneBasedEnd, int nIv, const int32_t* ivStart, const int32_t* ivEnd, const int32_t* ivPloidy) {
    int baseCounts[5] = {0, 0, 0, 0, 0};
    baseCounts[2] = (int)(oneBasedEnd - oneBasedStart + 1);
    for (int k = 0; k < nIv; k++) {
        if (ivPloidy[k] == 2) continue;
        int overlapStart = std::max((int)oneBasedStart - 1, ivStart[k] - 1);
        if (overlapStart > ivEnd[k]) continue;
        int overlapEnd = std::min((int)oneBasedEnd, ivEnd[k]);
        int overlapBases = overlapEnd - overlapStart;
        if (overlapBases <= 0) continue;
        if (ivPloidy[k] < 0 || ivPloidy[k] > 4) return -1;
        baseCounts[2] -= overlapBases;
        baseCounts[ivPloidy[k]] += overlapBases;
    }
    int nonZeroCount = 0;
    for (int cn = 0; cn < 5; cn++) if (baseCounts[cn] > 0) nonZeroCount++;
    return nonZeroCount < 2 ? 1 : 0;
}
int orc_is_uniform_reference_ploidy(int64_t oneBasedStart, int64_t oneBasedEnd, int nIv, const int32_t* ivStart, const int32_t* ivEnd, const int32_t* ivPloidy) {
    return IsUniformReferencePloidy(oneBasedStart, oneBasedEnd, nIv, ivStart, ivEnd, ivPloidy);
}

// SegmentationResultsProcessor.PostProcessSegments (CanvasPartition/SegmentationResultsProcessor.cs:17-129).
// Chromosomes are given in CoverageInfo (file) order. segStart[c] lists the segment starts of chromosome c (may be empty: Q17).
// excl*: forbidden intervals per chromosome.  nploidy == NULL: referencePloidy == null; otherwise nploidy[c] = -1 when the chromosome is not a key of
// PloidyByChromosome (IsUniformReferencePloidy returns true, PloidyInfo.cs:80-81), else the number of its intervals.
// Output: segment id per bin; returns the final counter value (-1000 if a ploidy outside 0..4 was hit: the reference throws).
int orc_postprocess_ploidy(int nchr, const int64_t* nbins, const uint32_t* const* binStart, const uint32_t* const* binEnd,
                           const int* nseg, const uint32_t* const* segStart, const int* nexcl, const int32_t* const* exclStart,
                           const int32_t* const* exclStop, int maxInterBinDist, const int* nploidy, const int32_t* const* plStart,
                           const int32_t* const* plEnd, const int32_t* const* plCn, int32_t* const* segId) {
    int segmentNum = -1;
    for (int c = 0; c < nchr; c++) {
        std::set<uint32_t> starts(segStart[c], segStart[c] + nseg[c]);
        int excludeIndex = 0;
        uint32_t previousBinEnd = 0;
        bool haveCurrent = false;
        for (int64_t b = 0; b < nbins[c]; b++) {
            uint32_t start = binStart[c][b], end = binEnd[c][b];
            bool newSegment = starts.count(start) > 0;
            if (nexcl && nexcl[c] > 0) {
                while (excludeIndex < nexcl[c] && (int64_t)exclStop[c][excludeIndex] < (int64_t)previousBinEnd) excludeIndex++;
                if (excludeIndex < nexcl[c]) {
                    int forbiddenZoneMid = (exclStart[c][excludeIndex] + exclStop[c][excludeIndex]) / 2;
                    if ((int64_t)previousBinEnd < forbiddenZoneMid && (int64_t)end >= forbiddenZoneMid) newSegment = true;
                }
            }
            if (previousBinEnd > 0 && maxInterBinDist >= 0 && (int64_t)previousBinEnd + maxInterBinDist < (int64_t)start && !newSegment) newSegment = true;
            if (!newSegment && nploidy && nploidy[c] >= 0) {                                   // :117-128
                int u = IsUniformReferencePloidy(previousBinEnd > 0 ? previousBinEnd : 1, end, nploidy[c], plStart[c], plEnd[c], plCn[c]);
                if (u < 0) return -1000;
                if (!u) newSegment = true;
            }
            if (newSegment) { segmentNum++; haveCurrent = true; }
            else if (!haveCurrent) haveCurrent = true;   // new SegmentWithBins(segmentNum, bin) re-using the current counter (Q17)
            segId[c][b] = segmentNum;
            previousBinEnd = end;
        }
    }
    return segmentNum;
}
int orc_postprocess(int nchr, const int64_t* nbins, const uint32_t* const* binStart, const uint32_t* const* binEnd,
                    const int* nseg, const uint32_t* const* segStart, const int* nexcl, const int32_t* const* exclStart,
                    const int32_t* const* exclStop, int maxInterBinDist, int32_t* const* segId) {
    return orc_postprocess_ploidy(nchr, nbins, binStart, binEnd, nseg, segStart, nexcl, exclStart, exclStop, maxInterBinDist, nullptr, nullptr, nullptr, nullptr, segId);
}

// SegmentationInput.reportScoresByWindow (CanvasPartition/Segmentation.cs:275-296): per chromosome, windows at index = 0, windowSize, ... while
// index < length - windowSize; each window takes windowSize - 1 values (Skip(index).Take(windowSize - 1)).  LINQ Average() and Sum() of doubles
// are sequential sums in list order.  Scores that are infinite or NaN are dropped.  (The ConcurrentBag's order is not deterministic, but both
// consumers sort.)
static std::vector<double> reportScoresByWindow(int nchr, const double* const* cov, const int64_t* n, int windowSize) {
    std::vector<double> evennessScores;
    for (int c = 0; c < nchr; c++)
        for (int64_t index = 0; index < n[c] - windowSize; index += windowSize) {
            const double* tmp = cov[c] + index; const int64_t cnt = windowSize - 1;
            double sum = 0; for (int64_t i = 0; i < cnt; i++) sum += tmp[i];
            double average = sum / (double)cnt;
            double tmpEvenness = 0;
            for (int coverageBin = 0; coverageBin <= average; coverageBin++) {
                int count = 0; for (int64_t i = 0; i < cnt; i++) if (tmp[i] >= coverageBin) count++;
                tmpEvenness += count / sum;
            }
            if (!std::isinf(tmpEvenness) && !std::isnan(tmpEvenness)) evennessScores.push_back(tmpEvenness);
        }
    return evennessScores;
}
// SegmentationInput.GetEvennessScore (Segmentation.cs:260-269).  Returns 0 and *score, or 1 when the reference would throw inside Quartiles / Median
// (fewer than 2 scores at the 10000-bin window or none at the requested one: WaveletsRunner.cs:58-67 catches it and writes no file).
int orc_evenness_score(int nchr, const double* const* cov, const int64_t* n, int windowSize, double* score) {
    const double IQRthreshold = 0.015; const int windowSizeIQR = 10000;
    auto evennessScoresIQR = reportScoresByWindow(nchr, cov, n, windowSizeIQR);
    if (evennessScoresIQR.size() < 2) return 1;                       // Quartiles indexes sorted[-1]
    std::vector<float> f; for (double v : evennessScoresIQR) f.push_back((float)v);   // Convert.ToSingle
    float q1, q2, q3; Quartiles(f, q1, q2, q3);
    auto evennessScores = reportScoresByWindow(nchr, cov, n, windowSize);
    if (evennessScores.empty()) return 1;
    double median = median_copy(evennessScores);              // Utilities.Median(IEnumerable<double>) -> SortedList<double>.Median()
    *score = (q3 - q1 > IQRthreshold) ? q3 * 100.0 : median * 100.0;
    return 0;
}
int orc_evenness_window_scores(int nchr, const double* const* cov, const int64_t* n, int windowSize, double* out, int cap) {
    auto v = reportScoresByWindow(nchr, cov, n, windowSize);
    for (size_t i = 0; i < v.size() && (int)i < cap; i++) out[i] = v[i];
    return (int)v.size();
}

// ---- CBS
int orc_cbs_boundary(uint32_t nPerm, double alpha, double eta, uint32_t* out, int cap) {
    std::vector<uint32_t> s;
    ComputeBoundary(nPerm, alpha, eta, s);
    for (size_t i = 0; i < s.size() && (int)i < cap; i++) out[i] = s[i];
    return (int)s.size();
}
double orc_phyper(double x, double NR, double NB, double n) { return phyper_lower(x, NR, NB, n); }
double orc_tailp(double b, double delta, int m, int nGrid, double tol) { return TailP(b, delta, m, nGrid, tol); }
void orc_tmaxo(const double* x, int n, double tss, double* sx, int* iseg, double* ostat, int al0) { TMaxO(x, n, tss, sx, iseg, *ostat, al0); }
double orc_htmaxp(int k, double tss, const double* px, int n, double* sx, int al0) { return HTMaxP(k, tss, px, n, sx, al0); }
double orc_tmaxp(double tss, const double* px, int n, double* sx, int al0) { return TMaxP(tss, px, n, sx, al0); }
void orc_mt_u32(uint32_t seed, int n, uint32_t* out) { MT19937 r(seed); for (int i = 0; i < n; i++) out[i] = r.next_u32(); }
// CBSRunner.cs:107-112: one NextFullRangeInt32() per chromosome in dictionary order from MersenneTwister(0)
void orc_cbs_seeds(int nchr, int32_t* seeds) { MT19937 g(0); for (int c = 0; c < nchr; c++) seeds[c] = g.next_full_range_int32(); }
void orc_xperm(const double* x, double* px, int n, uint32_t seed, int skip_perms) {
    MT19937 r(seed);
    for (int p = 0; p <= skip_perms; p++) {
        for (int i = 0; i < n; i++) px[i] = x[i];
        for (int i = n - 1; i >= 0; i--) { double cc = r.next_double(); int j = (int)(cc * (i + 1)); j = (j > i) ? i : j; std::swap(px[i], px[j]); }
    }
}
// ChangePoints for one chromosome. stats: 7 int64 (CbsStats). Returns number of segments.
int orc_cbs_chromosome(const double* x, int n, int32_t seed, const uint32_t* sbdry, int nsbdry, double alpha, uint32_t nPerm, int undo,
                       double trimmedSD, int32_t* lengthSeg, int cap, int64_t* stats) {
    std::vector<uint32_t> sb(sbdry, sbdry + nsbdry);
    MT19937 rnd((uint32_t)seed);
    CbsStats st;
    auto ls = ChangePoints(x, n, sb, rnd, alpha, nPerm, 2, 25, 200, undo, trimmedSD, 0.05, 3, &st);
    for (size_t i = 0; i < ls.size() && (int)i < cap; i++) lengthSeg[i] = ls[i];
    if (stats) { stats[0] = st.tmaxo_calls; stats[1] = st.tmaxo_elems; stats[2] = st.perms; stats[3] = st.perm_elems; stats[4] = st.tpermp_draws; stats[5] = st.tailp_exits; stats[6] = st.big_t_splits; }
    return (int)ls.size();
}
// whole genome CBS (finite data assumed), one std::thread per chromosome (CBSRunner.cs:115-147)
void orc_cbs_genome_undo(int nchr, const double* const* x, const int64_t* n, const uint32_t* sbdry, int nsbdry, double alpha, uint32_t nPerm,
                         int undo, int32_t* const* lengthSeg, const int* cap, int32_t* nseg, int64_t* stats7, int threads);
void orc_cbs_genome(int nchr, const double* const* x, const int64_t* n, const uint32_t* sbdry, int nsbdry, double alpha, uint32_t nPerm,
                    int32_t* const* lengthSeg, const int* cap, int32_t* nseg, int64_t* stats7, int threads) {
    orc_cbs_genome_undo(nchr, x, n, sbdry, nsbdry, alpha, nPerm, 0, lengthSeg, cap, nseg, stats7, threads);
}
void orc_cbs_genome_undo(int nchr, const double* const* x, const int64_t* n, const uint32_t* sbdry, int nsbdry, double alpha, uint32_t nPerm,
                         int undo, int32_t* const* lengthSeg, const int* cap, int32_t* nseg, int64_t* stats7, int threads) {
    double trimmedSD = 1.0;
    if (undo == 2) {   // CBSRunner.cs:102
        std::vector<const double*> sc(x, x + nchr); std::vector<int> ln(nchr); for (int c = 0; c < nchr; c++) ln[c] = (int)n[c];
        trimmedSD = std::sqrt(TrimmedVariance(sc, ln, 0.025));
    }
    std::vector<int32_t> seeds(nchr);
    orc_cbs_seeds(nchr, seeds.data());
    std::vector<int64_t> st((size_t)nchr * 7, 0);
    std::vector<std::thread> th;
    std::atomic_int next{0};
    auto work = [&]() { for (;;) { int c = next++; if (c >= nchr) break;
        nseg[c] = n[c] > 0 ? orc_cbs_chromosome(x[c], (int)n[c], seeds[c], sbdry, nsbdry, alpha, nPerm, undo, trimmedSD, lengthSeg[c], cap[c], &st[(size_t)c * 7]) : 0; } };
    for (int t = 0; t < std::max(1, threads); t++) th.emplace_back(work);
    for (auto& t : th) t.join();
    if (stats7) { for (int k = 0; k < 7; k++) { stats7[k] = 0; for (int c = 0; c < nchr; c++) stats7[k] += st[(size_t)c * 7 + k]; } }
}
double orc_trimmed_variance(int nchr, const double* const* x, const int* n, double trim) {
    std::vector<const double*> s(x, x + nchr); std::vector<int> l(n, n + nchr);
    return TrimmedVariance(s, l, trim);
}
int orc_changepoints_prune(const double* x, int n, const int32_t* lengthSeg, int nseg, double cutoff, int32_t* out, int cap) {
    std::vector<int> ls(lengthSeg, lengthSeg + nseg);
    auto r = ChangePointsPrune(x, n, ls, cutoff);
    for (size_t i = 0; i < r.size() && (int)i < cap; i++) out[i] = r[i];
    return (int)r.size();
}
// Utilities.MergeMultiSampleCleanedBedFile (CanvasCommon/Utilities.cs:834-920) on SoA inputs (chromosome as an index): chromosomes in
// the order the HashSet first saw them, bins of a chromosome in the order their start first appeared (Dictionary enumeration order
// without removals), kept when every file contributed a count; stop = the value read last.  Returns the number of merged bins;
// outCount is [nsamples][cap].
int64_t orc_merge_cleaned(int nsamples, const int64_t* n, const int32_t* const* chr, const int32_t* const* start, const int32_t* const* stop, const float* const* count,
                          int32_t* outChr, int32_t* outStart, int32_t* outStop, float* const* outCount, int64_t cap) {
    std::vector<int32_t> chromOrder; std::map<int32_t, int> chromIndex;
    for (int s = 0; s < nsamples; s++) for (int64_t i = 0; i < n[s]; i++) if (!chromIndex.count(chr[s][i])) { chromIndex[chr[s][i]] = (int)chromOrder.size(); chromOrder.push_back(chr[s][i]); }
    struct Entry { int32_t stop; std::vector<float> counts; };
    std::vector<std::vector<int32_t>> keyOrder(chromOrder.size());
    std::vector<std::map<int32_t, Entry>> byPos(chromOrder.size());
    for (int s = 0; s < nsamples; s++)
        for (int64_t i = 0; i < n[s]; i++) {
            int ci = chromIndex[chr[s][i]]; int32_t pos = start[s][i];
            auto it = byPos[ci].find(pos);
            if (it == byPos[ci].end()) { keyOrder[ci].push_back(pos); it = byPos[ci].insert({pos, Entry{}}).first; }
            it->second.stop = stop[s][i];
            it->second.counts.push_back(count[s][i]);
        }
    int64_t k = 0;
    for (size_t ci = 0; ci < chromOrder.size(); ci++)
        for (int32_t pos : keyOrder[ci]) {
            const Entry& e = byPos[ci][pos];
            if ((int)e.counts.size() < nsamples) continue;
            if (k < cap) { outChr[k] = chromOrder[ci]; outStart[k] = pos; outStop[k] = e.stop; for (int s = 0; s < nsamples; s++) outCount[s][k] = e.counts[s]; }
            k++;
        }
    return k;
}
void orc_sort_keys_items(double* keys, int* items, int n) { dotnet_sort_keys_items(keys, items, 0, n, n); }

}  // extern "C"
