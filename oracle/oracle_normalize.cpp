// TEST INFRASTRUCTURE ONLY — CPU restatement of CanvasNormalize's ratio path:
//   WeightedAverageReferenceGenerator.Run (CanvasNormalize/WeightedAverageReferenceGenerator.cs:28-70),
//   BinCounts.OnTargetMedianBinCount (BinCounts.cs:36-60), LSNormRatioCalculator.Run (LSNormRatioCalculator.cs:20-48),
//   RawRatioCalculator.Run (RawRatioCalculator.cs:21-46), CanvasNormalizeUtilities.RatiosToCounts (CanvasNormalizeUtilities.cs:23-33).
// The reference has no known-answer test for these: parity unpinned, the statements are followed one by one.  The manifest
// (Isas.Manifests.NexteraManifest, not in /root/reference) only decides which bins are "on target": that index list is an input here.
#include "oracle_common.h"
#include "oracle_api.h"

using namespace oracle;

static double median_on(const std::vector<double>& counts, const int32_t* onIdx, int64_t nOn) {
    std::vector<double> v;
    if (onIdx) { v.resize((size_t)nOn); for (int64_t i = 0; i < nOn; i++) v[(size_t)i] = counts[(size_t)onIdx[i]]; } else v = counts;
    return sorted_median(v);
}

extern "C" {
// weights[i] = 1 / median_i (0 when the median is not positive), normalised to sum 1; out[j] = sum_i weights[i] * counts_i[j], i ascending
void orc_norm_weighted_reference(int nsamples, const double* const* counts, int64_t n, const int32_t* onIdx, int64_t nOn, double* out, double* weights) {
    for (int s = 0; s < nsamples; s++) {
        double median = median_on(std::vector<double>(counts[s], counts[s] + n), onIdx, nOn);
        weights[s] = median > 0 ? 1.0 / median : 0;
    }
    double weightSum = 0; for (int s = 0; s < nsamples; s++) weightSum += weights[s];
    for (int s = 0; s < nsamples; s++) weights[s] /= weightSum;
    for (int64_t j = 0; j < n; j++) { double w = 0; for (int s = 0; s < nsamples; s++) w += weights[s] * counts[s][j]; out[j] = w; }
}
// mode 0: LSNorm (library-size factor from the on-target medians, bins with reference < 1 dropped); mode 1: Raw (reference outside
// [minRef, maxRef] dropped).  ploidy: reference copy number per bin or NULL (2).  Returns the number of bins kept.
int64_t orc_norm_ratio(int64_t n, const float* sample, const float* reference, const int32_t* onIdx, int64_t nOn, int mode, double minRef, double maxRef,
                       const int32_t* ploidy, int32_t* keepIdx, float* ratio, float* count) {
    double lsf = 1;
    if (mode == 0) {
        std::vector<double> s(n), r(n);
        for (int64_t i = 0; i < n; i++) { s[i] = (double)sample[i]; r[i] = (double)reference[i]; }
        double sm = median_on(s, onIdx, nOn), rm = median_on(r, onIdx, nOn);
        lsf = (sm > 0 && rm > 0) ? rm / sm : 1;
    }
    int64_t k = 0;
    for (int64_t i = 0; i < n; i++) {
        if (mode == 0) { if (reference[i] < 1) continue; }
        else { if ((double)reference[i] < minRef) continue; if ((double)reference[i] > maxRef) continue; }
        float q = sample[i] / reference[i];                       // float / float (C#)
        double rt = mode == 0 ? (double)q * lsf : (double)q;
        float rf = (float)rt;
        double factor = 40.0 * (ploidy ? ploidy[i] : 2) / 2.0;    // CanvasDiploidBinRatioFactor * ploidy / 2.0
        keepIdx[k] = (int32_t)i; ratio[k] = rf; count[k] = (float)((double)rf * factor);
        k++;
    }
    return k;
}
}
