// TEST INFRASTRUCTURE ONLY (see oracle_common.h). CPU restatement of CanvasBin's merge step and of CanvasClean.
// Every function cites the reference file:line it follows (paths relative to /root/reference/Src/Canvas/).
#include "oracle_common.h"
#include "oracle_api.h"

namespace oracle {

// ------------------------------------------------------------------ formatting (CanvasCommon/IO.cs:21; Q16)
static std::string round_digits_fixed(std::string digits, int scale, bool neg, int decimals) {
    // digits: significant decimal digits d1d2..dn meaning 0.d1d2..dn * 10^scale. Round half-up (on the digit string)
    // at `decimals` fractional digits, as .NET Core 2.x Number.RoundNumber does.
    int pos = scale + decimals;  // number of digits kept
    if (pos < 0) { digits = ""; }
    else if (pos < (int)digits.size()) {
        bool up = digits[pos] >= '5';
        digits.resize(pos);
        if (up) {
            int i = pos - 1;
            while (i >= 0 && digits[i] == '9') { digits[i] = '0'; i--; }
            if (i >= 0) digits[i]++;
            else { digits.insert(digits.begin(), '1'); scale++; }
        }
    }
    // strip to canonical: build integer and fraction parts
    std::string ip, fp;
    for (int i = 0; i < scale; i++) ip.push_back(i < (int)digits.size() ? digits[i] : '0');
    if (ip.empty()) ip = "0";
    for (int i = 0; i < decimals; i++) {
        int di = scale + i;
        fp.push_back((di >= 0 && di < (int)digits.size()) ? digits[di] : '0');
    }
    bool allzero = true;
    for (char c : ip) if (c != '0') allzero = false;
    for (char c : fp) if (c != '0') allzero = false;
    std::string out;
    if (neg && !allzero) out.push_back('-');  // .NET Core 2.x prints "-0.00" as "0.00"? It prints "-0.00" only from 3.0 on.
    out += ip;
    if (decimals > 0) { out.push_back('.'); out += fp; }
    return out;
}

static void to_sig_digits(double v, int prec, std::string& digits, int& scale) {
    // correctly rounded `prec` significant digits: 0.DIGITS * 10^scale
    char buf[64];
    snprintf(buf, sizeof buf, "%.*e", prec - 1, std::fabs(v));
    // d.ddddde[+-]xx
    digits.clear();
    const char* p = buf;
    for (; *p && *p != 'e'; p++) if (*p >= '0' && *p <= '9') digits.push_back(*p);
    int ex = atoi(p + 1);
    scale = ex + 1;
    // strip trailing zeros (Number struct keeps only significant digits)
    while (!digits.empty() && digits.back() == '0') digits.pop_back();
    if (digits.empty()) scale = 0;
}

std::string format_float_f2(float v) {
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v > 0 ? "Infinity" : "-Infinity";
    std::string digits; int scale;
    to_sig_digits((double)v, 7, digits, scale);
    return round_digits_fixed(digits, scale, std::signbit(v), 2);
}

static std::string format_general(double v, int prec) {
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v > 0 ? "Infinity" : "-Infinity";
    std::string digits; int scale;
    to_sig_digits(v, prec, digits, scale);
    if (digits.empty()) return "0";
    std::string out;
    if (std::signbit(v)) out.push_back('-');
    int exp10 = scale - 1;
    if (exp10 >= prec || exp10 < -5) {  // scientific: d.dddE+xx
        out.push_back(digits[0]);
        if (digits.size() > 1) { out.push_back('.'); out += digits.substr(1); }
        char eb[16]; snprintf(eb, sizeof eb, "E%c%02d", exp10 < 0 ? '-' : '+', std::abs(exp10));
        out += eb;
        return out;
    }
    if (scale <= 0) { out += "0."; out.append(-scale, '0'); out += digits; return out; }
    for (int i = 0; i < scale; i++) out.push_back(i < (int)digits.size() ? digits[i] : '0');
    if ((int)digits.size() > scale) { out.push_back('.'); out += digits.substr(scale); }
    return out;
}
std::string format_double_g15(double v) { return format_general(v, 15); }
std::string format_float_g7(float v) { return format_general((double)v, 7); }

// ------------------------------------------------------------------ CanvasBin
// HitArray.CountSetBits (CanvasBin/HitArray.cs:24-32) and CanvasBin.CountSetBits (CanvasBin/CanvasBin.cs:146-156);
// SampleHitArrays.GetRates (CanvasBin.cs:30-71): rate = numberObserved / (double)numberPossible.
double bin_rate(const uint8_t* hits, const uint8_t* mask, int64_t len) {
    int numberObserved = 0, numberPossible = 0;
    for (int64_t i = 0; i < len; i++) if (hits[i] > 0) numberObserved++;
    for (int64_t i = 0; i < len; i++) if ((mask[i >> 3] >> (i & 7)) & 1) numberPossible++;
    return numberObserved / (double)numberPossible;
}

// SampleHitArrays.GetBinSize (CanvasBin.cs:79-83): (int)(countsPerBin / Median(rates)); Median via SortedList<double>.
int bin_size_from_rates(const double* rates, int n, int countsPerBin) {
    std::vector<double> r(rates, rates + n);
    double medianRate = sorted_median(r);
    return (int)(countsPerBin / medianRate);
}

// BinCountsForChromosome (CanvasBin.cs:568-661), non-predefined-bins path, modes Binary(0) / TruncatedDynamicRange(3).
// Quirks kept: Q1 (NucleotideCount counts every position, leading-'n' skip is lowercase only), Q3 (no trailing partial
// bin), Q4 (gc in float32).
int64_t bin_chromosome(const uint8_t* bases, const uint8_t* mask, const uint8_t* hits, int64_t len, int binSize, int mode,
                       int64_t cap, int32_t* start, int32_t* stop, int32_t* gc, int32_t* count) {
    int64_t pos = 0;
    while (pos < len && bases[pos] == 'n') pos++;   // :582 (throws past the end in C#; we just emit nothing)
    int NucleotideCount = 0, GCCount = 0, PossibleCount = 0, ObservedCount = 0, TruncObserved = 0;
    int64_t StartPosition = -1;
    int64_t nb = 0;
    for (; pos < len; pos++) {
        if (StartPosition == -1) StartPosition = pos;            // :588
        NucleotideCount++;                                        // :592 (char.Equals(string) is always false)
        switch (bases[pos]) { case 'C': case 'c': case 'G': case 'g': GCCount++; break; default: break; }  // :595-602
        if ((mask[pos >> 3] >> (pos & 7)) & 1) {                  // :604
            PossibleCount++;
            ObservedCount += hits[pos];
            TruncObserved += std::min(10, (int)hits[pos]);        // :618-625 applied at close time over binObservations
        }
        if (PossibleCount == binSize) {                           // :615
            int obs = (mode == 3) ? TruncObserved : ObservedCount;
            float gcf = 100.0f * (float)GCCount;                  // :638  (int)(100f * GCCount / NucleotideCount)
            gcf = gcf / (float)NucleotideCount;
            int g = (int)gcf;
            if (nb < cap) { start[nb] = (int32_t)StartPosition; stop[nb] = (int32_t)(pos + 1); gc[nb] = g; count[nb] = obs; }
            nb++;
            NucleotideCount = GCCount = PossibleCount = ObservedCount = TruncObserved = 0;
            StartPosition = -1;
        }
    }
    return nb;
}

// BinCountsForChromosome with predefined bins (CanvasBin -n; CanvasBin.cs:568-661 with usePredefinedBins): the cursor starts at the first bin's Start, skips leading 'n',
// closes a bin when it stands on Stop - 1 and jumps to the next bin's Start.  Returns the number of bins closed (the others keep gc / count as loaded from the BED file),
// or -1 where the C# would index past the end of the chromosome while skipping 'n'.  Modes 0 / 3.
int64_t bin_chromosome_predefined(const uint8_t* bases, const uint8_t* mask, const uint8_t* hits, int64_t len, int mode, int64_t nbins, const int32_t* binStart, const int32_t* binStop,
                                  int32_t* gc, int32_t* count) {
    if (nbins == 0) return 0;
    int64_t predefinedBinIndex = 0;
    int64_t pos = binStart[0];
    while (true) { if (pos >= len) return -1; if (bases[pos] != 'n') break; pos++; }       // :582-584
    int NucleotideCount = 0, GCCount = 0, ObservedCount = 0, TruncObserved = 0;
    for (; pos < len; pos++) {
        NucleotideCount++;                                                                  // :592-593 (char.Equals(string): always counted)
        switch (bases[pos]) { case 'C': case 'c': case 'G': case 'g': GCCount++; break; default: break; }
        if ((mask[pos >> 3] >> (pos & 7)) & 1) { ObservedCount += hits[pos]; TruncObserved += std::min(10, (int)hits[pos]); }
        if (pos == (int64_t)binStop[predefinedBinIndex] - 1) {                              // :616
            float gcf = 100.0f * (float)GCCount; gcf = gcf / (float)NucleotideCount;
            gc[predefinedBinIndex] = (int)gcf; count[predefinedBinIndex] = (mode == 3) ? TruncObserved : ObservedCount;
            predefinedBinIndex++;
            if (predefinedBinIndex >= nbins) break;
            pos = (int64_t)binStart[predefinedBinIndex] - 1;                                // :646
            NucleotideCount = GCCount = ObservedCount = TruncObserved = 0;
        }
    }
    return predefinedBinIndex;
}

// ---- GCContentWeighted mode (CanvasBin.cs:416-506, 330-405, 626-636)
// Utilities.NonZeroMean(Int16[]) (CanvasCommon/Utilities.cs:135-151)
static int16_t NonZeroMean(const int16_t* x, int64_t n) {
    long long sum = 0, counter = 0;
    for (int64_t i = 0; i < n; i++) if (x[i] > 0) { sum += x[i]; counter++; }
    if (counter == 0) return 0;
    return (int16_t)(sum / counter);
}
// MeanFragmentSize (CanvasBin.cs:164-174)
int16_t mean_fragment_size(int nchr, const int16_t* const* fl, const int64_t* len) {
    std::vector<int16_t> means;
    for (int c = 0; c < nchr; c++) means.push_back(NonZeroMean(fl[c], len[c]));
    return NonZeroMean(means.data(), (int64_t)means.size());
}
// read GC content per position (CanvasBin.cs:451-500)
void read_gc_content(const uint8_t* bases, const int16_t* fl, int64_t L, int meanFragmentSize, uint8_t* gcContent) {
    const int meanFragmentCutoff = 3;
    const uint8_t gcCap = 101;
    for (int64_t i = 0; i < L; i++) gcContent[i] = 0;
    for (int64_t pos = 0; pos < L - (int64_t)meanFragmentSize * meanFragmentCutoff - 1; pos++) {
        int16_t currentFragment;
        if (fl[pos] == 0) currentFragment = (int16_t)meanFragmentSize;
        else currentFragment = (int16_t)std::min((int)fl[pos], meanFragmentSize * meanFragmentCutoff);
        uint32_t gcCounter = 0;
        for (int64_t i = pos; i < pos + currentFragment; i++) switch (bases[i]) { case 'C': case 'c': case 'G': case 'g': gcCounter++; break; default: break; }
        long long v = (long long)100 * (long long)gcCounter / (long long)currentFragment;
        gcContent[pos] = (uint8_t)std::min<long long>(v, gcCap);
    }
}
// ComputeObservedVsExpectedGC (CanvasBin.cs:330-405), manifest == null
void observed_vs_expected_gc(int nchr, const uint8_t* const* readGC, const uint8_t* const* hits, const int64_t* len, float* out101) {
    long long expectedC[101] = {0}, observedC[101] = {0};
    for (int c = 0; c < nchr; c++) for (int64_t i = 0; i < len[c]; i++) { expectedC[readGC[c][i]]++; observedC[readGC[c][i]] += hits[c][i]; }
    long long sumObserved = 0, sumExpected = 0;
    for (int b = 0; b < 101; b++) { sumObserved += observedC[b]; sumExpected += expectedC[b]; }
    for (int b = 0; b < 101; b++) {
        if (expectedC[b] == 0) expectedC[b] = 1;
        if (observedC[b] == 0) observedC[b] = 1;
        out101[b] = ((float)observedC[b] / (float)expectedC[b]) * ((float)sumExpected / (float)sumObserved);
    }
}
// BinCountsForChromosome, GCContentWeighted branch (CanvasBin.cs:626-636): float32 accumulation in position order, Math.Round half-even (Q5)
int64_t bin_chromosome_weighted(const uint8_t* bases, const uint8_t* mask, const uint8_t* hits, const uint8_t* readGC, const float* obsVsExp, int64_t len, int binSize,
                                int64_t cap, int32_t* start, int32_t* stop, int32_t* gc, int32_t* count) {
    int64_t pos = 0;
    while (pos < len && bases[pos] == 'n') pos++;
    int NucleotideCount = 0, GCCount = 0, PossibleCount = 0;
    float tmpObservedCount = 0;
    int64_t StartPosition = -1, nb = 0;
    for (; pos < len; pos++) {
        if (StartPosition == -1) StartPosition = pos;
        NucleotideCount++;
        switch (bases[pos]) { case 'C': case 'c': case 'G': case 'g': GCCount++; break; default: break; }
        if ((mask[pos >> 3] >> (pos & 7)) & 1) {
            PossibleCount++;
            float q = (float)(int)hits[pos] / obsVsExp[readGC[pos]];
            tmpObservedCount += std::min(10.0f, q);       // Math.Min(10, float): the int literal widens to float
        }
        if (PossibleCount == binSize) {
            int obs = (int)round_half_even((double)tmpObservedCount);
            float gcf = 100.0f * (float)GCCount; gcf = gcf / (float)NucleotideCount;
            if (nb < cap) { start[nb] = (int32_t)StartPosition; stop[nb] = (int32_t)(pos + 1); gc[nb] = (int)gcf; count[nb] = obs; }
            nb++;
            NucleotideCount = GCCount = PossibleCount = 0; tmpObservedCount = 0; StartPosition = -1;
        }
    }
    return nb;
}

// BinCountsForChromosome with predefined bins AND the GCContentWeighted branch (CanvasBin.cs:575-655: the close at :616-617 is shared, the weighted count at :626-636).
// Same cursor as bin_chromosome_predefined; returns the bins closed or -1.
int64_t bin_chromosome_predefined_weighted(const uint8_t* bases, const uint8_t* mask, const uint8_t* hits, const uint8_t* readGC, const float* obsVsExp, int64_t len, int64_t nbins,
                                           const int32_t* binStart, const int32_t* binStop, int32_t* gc, int32_t* count) {
    if (nbins == 0) return 0;
    int64_t predefinedBinIndex = 0;
    int64_t pos = binStart[0];
    while (true) { if (pos >= len) return -1; if (bases[pos] != 'n') break; pos++; }       // :582-584
    int NucleotideCount = 0, GCCount = 0;
    float tmpObservedCount = 0;
    for (; pos < len; pos++) {
        NucleotideCount++;
        switch (bases[pos]) { case 'C': case 'c': case 'G': case 'g': GCCount++; break; default: break; }
        if ((mask[pos >> 3] >> (pos & 7)) & 1) {                                            // :604-611 binObservations / binPositions, summed at the close in the same order
            float q = (float)(int)hits[pos] / obsVsExp[readGC[pos]];
            tmpObservedCount += std::min(10.0f, q);
        }
        if (pos == (int64_t)binStop[predefinedBinIndex] - 1) {                              // :616-617
            float gcf = 100.0f * (float)GCCount; gcf = gcf / (float)NucleotideCount;
            gc[predefinedBinIndex] = (int)gcf; count[predefinedBinIndex] = (int)round_half_even((double)tmpObservedCount);
            predefinedBinIndex++;
            if (predefinedBinIndex >= nbins) break;
            pos = (int64_t)binStart[predefinedBinIndex] - 1;                                // :646
            NucleotideCount = GCCount = 0; tmpObservedCount = 0;
        }
    }
    return predefinedBinIndex;
}

// ------------------------------------------------------------------ CanvasClean
struct Bins {
    std::vector<int32_t> chr, start, stop, gc;
    std::vector<float> count;
    std::vector<double> dev;  // SampleGenomicBin.CountDeviation, initialised to -1 (CanvasCommon/GenomicBin.cs:83)
    size_t size() const { return chr.size(); }
    void push_from(const Bins& o, size_t i) {
        chr.push_back(o.chr[i]); start.push_back(o.start[i]); stop.push_back(o.stop[i]); gc.push_back(o.gc[i]);
        count.push_back(o.count[i]); dev.push_back(o.dev[i]);
    }
};

// RemoveBigBins (CanvasClean/CanvasClean.cs:328-355)
static Bins RemoveBigBins(const Bins& bins) {
    std::vector<int> sizes(bins.size());
    for (size_t i = 0; i < bins.size(); i++) sizes[i] = bins.stop[i] - bins.start[i];
    std::sort(sizes.begin(), sizes.end());
    int index = (int)(0.98 * (double)bins.size());
    if (index >= (int)sizes.size()) return bins;
    int thresh = sizes[index];
    Bins out;
    for (size_t i = 0; i < bins.size(); i++) if (bins.stop[i] - bins.start[i] <= thresh) out.push_from(bins, i);
    return out;
}

// SignificantlyDifferent (CanvasClean.cs:363-381)
static bool SignificantlyDifferent(float a, float b) {
    double mu = ((double)a + (double)b) / 2;
    if (a + b == 0) return false;   // float add
    double da = (double)a - mu, db = (double)b - mu;
    double chi2 = (da * da + db * db) / mu;
    return chi2 > 6.635;
}

// RemoveOutliers (CanvasClean.cs:387-413)
static Bins RemoveOutliers(const Bins& bins) {
    Bins out;
    int64_t n = (int64_t)bins.size();
    for (int64_t i = 0; i < n; i++) {
        bool hasPrev = i > 0, hasNext = i < n - 1;
        bool prevSame = hasPrev && bins.chr[i] == bins.chr[i - 1];
        bool nextSame = hasNext && bins.chr[i] == bins.chr[i + 1];
        if ((hasPrev && !prevSame) && (hasNext && !nextSame)) continue;
        if ((prevSame && !SignificantlyDifferent(bins.count[i], bins.count[i - 1])) ||
            (nextSame && !SignificantlyDifferent(bins.count[i], bins.count[i + 1])) || (!hasPrev && !hasNext))
            out.push_from(bins, i);
    }
    return out;
}

// Utilities.StandardDeviation(double[], start, end) (CanvasCommon/Utilities.cs:246-262) + Mean (:197-210)
static double StandardDeviationRange(const std::vector<double>& x, int start, int end) {
    double sum = 0;
    for (int i = start; i < end; i++) sum += x[i];
    double mu = sum / (end - start);
    double s2 = 0;
    for (int i = start; i < end; i++) { double d = x[i] - mu; s2 += d * d; }
    return std::sqrt(s2 / (end - start - 1));
}

// Utilities.Mad (Utilities.cs:451-462) on a slice
static double MadRange(const std::vector<double>& x, int start, int end) {
    std::vector<double> s(x.begin() + start, x.begin() + end);
    double median = median_copy(s);
    std::vector<double> diffs(s.size());
    for (size_t i = 0; i < s.size(); i++) diffs[i] = std::fabs(s[i] - median);
    return sorted_median(diffs);
}

// GetLocalStandardDeviation (CanvasClean.cs:268-298) + GetLocalStandardDeviationAverage (:243-258). Q8 kept.
static double GetLocalStandardDeviation(Bins& bins) {
    int n = (int)bins.size();
    std::vector<double> countsDiffs(n > 0 ? n - 1 : 0);
    for (int i = 0; i < n - 1; i++) countsDiffs[i] = (double)(float)(bins.count[i + 1] - bins.count[i]);  // float subtract, Convert.ToDouble
    std::vector<double> localSDs;
    std::vector<int> chromosomeBin;
    const int windowSize = 20;
    for (int windowEnd = windowSize, windowStart = 0; windowEnd < (int)countsDiffs.size(); windowStart += windowSize, windowEnd += windowSize) {
        double localSD = StandardDeviationRange(countsDiffs, windowStart, windowEnd);
        localSDs.push_back(localSD);
        chromosomeBin.push_back(bins.chr[windowStart]);
        for (int b = windowStart; b < windowEnd; b++) bins.dev[b] = localSD;
    }
    std::vector<double> mads;
    int iStart = 0;
    for (int i = 0; i < (int)localSDs.size(); i++) {
        if (chromosomeBin[i] != chromosomeBin[iStart]) { mads.push_back(MadRange(localSDs, iStart, i)); iStart = i; }
    }
    mads.push_back(MadRange(localSDs, iStart, (int)localSDs.size()));  // throws in C# if empty; we require n>=50000 upstream
    double s = 0;
    for (double m : mads) s += m;
    return s / (double)mads.size();   // List<double>.Average()
}

// RemoveBinsWithExtremeLocalSD (CanvasClean.cs:308-322)
static Bins RemoveBinsWithExtremeLocalSD(const Bins& bins, double localSDaverage, double threshold) {
    Bins out;
    for (size_t i = 0; i < bins.size(); i++) {
        if (bins.dev[i] > threshold * 2.0 && localSDaverage > 5.0) continue;
        out.push_from(bins, i);
    }
    return out;
}

static const int numberOfGCbins = 101;          // EnrichmentUtilities.cs:58
static const int defaultMinNumberOfBinsPerGC = 100;  // CanvasClean.cs:14

// RemoveBinsWithExtremeGC (CanvasClean.cs:207-237), manifest == null
static Bins RemoveBinsWithExtremeGC(const Bins& bins, int threshold, const uint8_t* isAuto, int minBinsWeighted) {
    std::vector<int> counts(numberOfGCbins, 0);
    double totalCount = 0;
    for (size_t i = 0; i < bins.size(); i++) {
        if (!isAuto[bins.chr[i]]) continue;
        counts[bins.gc[i]]++;
        totalCount++;
    }
    int averageCountPerGC = std::max(minBinsWeighted, (int)(totalCount / counts.size()));
    threshold = std::min(threshold, averageCountPerGC);
    Bins out;
    for (size_t i = 0; i < bins.size(); i++) {
        if (counts[bins.gc[i]] < threshold) continue;
        out.push_from(bins, i);
    }
    return out;
}

// EnrichmentUtilities.GetCountsByGC (CanvasClean/EnrichmentUtilities.cs:65-84), manifest == null
static void GetCountsByGC(const Bins& bins, const uint8_t* isAuto, std::vector<std::vector<float>>& countsByGC, std::vector<float>& counts) {
    countsByGC.assign(numberOfGCbins, {});
    counts.clear();
    for (size_t i = 0; i < bins.size(); i++) {
        if (!isAuto[bins.chr[i]]) continue;
        countsByGC[bins.gc[i]].push_back(bins.count[i]);
        counts.push_back(bins.count[i]);
    }
}

struct WC { float v, w; };
// GetWeightedCounts (CanvasClean.cs:107-132)
static std::vector<WC> GetWeightedCounts(const std::vector<std::vector<float>>& countsByGC, int gcBin) {
    std::vector<WC> wc;
    int radius = 0;
    float weight = 1;
    while ((int)wc.size() < defaultMinNumberOfBinsPerGC) {
        int gcWindowEnd = gcBin + radius, gcWindowStart = gcBin - radius;
        if (gcWindowEnd >= (int)countsByGC.size() && gcWindowStart < 0) break;
        if (gcWindowEnd < (int)countsByGC.size()) for (float c : countsByGC[gcWindowEnd]) wc.push_back({c, weight});
        if (gcWindowStart != gcWindowEnd && gcWindowStart >= 0) for (float c : countsByGC[gcWindowStart]) wc.push_back({c, weight});
        radius++;
        weight /= 2;
    }
    return wc;
}

// Utilities.WeightedQuantiles (CanvasCommon/Utilities.cs:493-515). Q9 kept (LINQ Sum<float> semantics, stable OrderBy).
void WeightedQuantiles(const std::vector<WC>& x, const float* probs, int nprobs, double* quantiles) {
    double acc = 0;
    for (auto& t : x) acc += t.w;
    double totalWeight = (double)(float)acc;
    double cumulativeWeight = 0, cumulativeProb = 0;
    for (int i = 0; i < nprobs; i++) quantiles[i] = 0;
    std::vector<WC> s(x);
    std::stable_sort(s.begin(), s.end(), [](const WC& a, const WC& b) { return a.v < b.v; });
    for (auto& t : s) {
        cumulativeWeight += t.w;
        cumulativeProb = cumulativeWeight / totalWeight;
        for (int i = 0; i < nprobs; i++) if (cumulativeProb <= (double)probs[i]) quantiles[i] = t.v;
    }
}

// Utilities.Quartiles (Utilities.cs:361-419), float arithmetic
void Quartiles(const std::vector<float>& x, float& fQ1, float& fQ2, float& fQ3) {
    std::vector<float> sorted(x);
    std::sort(sorted.begin(), sorted.end());
    int iSize = (int)sorted.size();
    int iMid = iSize / 2;
    fQ1 = fQ2 = fQ3 = 0;
    if (iSize == 0) return;  // C# would throw
    if (iSize % 2 == 0) {
        fQ2 = (sorted[iMid - 1] + sorted[iMid]) / 2;
        int iMidMid = iMid / 2;
        if (iMid % 2 == 0) {
            fQ1 = (sorted[iMidMid - 1] + sorted[iMidMid]) / 2;
            fQ3 = (sorted[iMid + iMidMid - 1] + sorted[iMid + iMidMid]) / 2;
        } else {
            fQ1 = sorted[iMidMid];
            fQ3 = sorted[iMidMid + iMid];
        }
    } else {
        fQ2 = sorted[iMid];
        if ((iSize - 1) % 4 == 0) {
            int n = (iSize - 1) / 4;
            if (n >= 1) {
                fQ1 = (sorted[n - 1] * 0.25f) + (sorted[n] * 0.75f);
                fQ3 = (sorted[3 * n] * 0.75f) + (sorted[3 * n + 1] * 0.25f);
            } else { fQ1 = fQ3 = sorted[0]; }  // iSize==1: C# indexes [-1] and throws; keep defined
        } else if ((iSize - 3) % 4 == 0) {
            int n = (iSize - 3) / 4;
            fQ1 = (sorted[n] * 0.75f) + (sorted[n + 1] * 0.25f);
            fQ3 = (sorted[3 * n + 1] * 0.25f) + (sorted[3 * n + 2] * 0.75f);
        }
    }
}

// NormalizeByGC, MedianByGC flavour (CanvasClean.cs:163-196)
static void NormalizeByGC(Bins& bins, const uint8_t* isAuto) {
    std::vector<std::vector<float>> countsByGC;
    std::vector<float> counts;
    GetCountsByGC(bins, isAuto, countsByGC, counts);
    double globalMedian = (double)median_copy(counts);
    std::vector<double> medians(numberOfGCbins);
    for (int g = 0; g < numberOfGCbins; g++) {
        if ((int)countsByGC[g].size() >= defaultMinNumberOfBinsPerGC) medians[g] = (double)median_copy(countsByGC[g]);
        else {
            auto wc = GetWeightedCounts(countsByGC, g);
            const float p = 0.5f;
            double q;
            WeightedQuantiles(wc, &p, 1, &q);
            medians[g] = q;
        }
    }
    for (size_t i = 0; i < bins.size(); i++) {
        double median = medians[bins.gc[i]];
        if (median > 0) bins.count[i] = (float)(globalMedian * (double)bins.count[i] / median);
    }
}

// NormalizeVarianceByGC (CanvasClean.cs:34-97)
static bool NormalizeVarianceByGC(Bins& bins, const uint8_t* isAuto) {
    std::vector<std::vector<float>> countsByGC;
    std::vector<float> counts;
    GetCountsByGC(bins, isAuto, countsByGC, counts);
    float gq1, gq2, gq3;
    Quartiles(counts, gq1, gq2, gq3);
    std::vector<float> localIQR, lq2;
    for (int i = 0; i < numberOfGCbins; i++) {
        if (countsByGC[i].empty()) { localIQR.push_back(-1.0f); lq2.push_back(-1.0f); }
        else if ((int)countsByGC[i].size() >= defaultMinNumberOfBinsPerGC) {
            float q1, q2, q3;
            Quartiles(countsByGC[i], q1, q2, q3);
            lq2.push_back(q2);
            localIQR.push_back(q3 - q1);
        } else {
            auto wc = GetWeightedCounts(countsByGC, i);
            const float p[3] = {0.25f, 0.5f, 0.75f};
            double q[3];
            WeightedQuantiles(wc, p, 3, q);
            lq2.push_back((float)q[1]);
            localIQR.push_back((float)(q[2] - q[0]));
        }
    }
    float globalIQR = gq3 - gq1;
    int significantIQRcounter = 0;
    for (int i = 10; i < 90; i++) if (globalIQR * 2.0f < localIQR[i]) significantIQRcounter++;
    if (significantIQRcounter <= 0) return false;
    for (size_t b = 0; b < bins.size(); b++) {
        float scaledLocalIqr = localIQR[bins.gc[b]] * 0.8f;
        if (globalIQR >= scaledLocalIqr) continue;
        float iqrRatio = scaledLocalIqr / globalIQR;
        float medianGCCount = lq2[bins.gc[b]];
        bins.count[b] = medianGCCount + (bins.count[b] - medianGCCount) / iqrRatio;
    }
    return true;
}

// ---- LOESS (CanvasClean/LoessInterpolator.cs, LoessGCNormalizer.cs)
struct LoessInterval { double xmin, xmax; int l, r; };
struct LoessModel {
    std::vector<double> xs, ys, fitted, rw;  // sorted by x
    std::vector<int> order;                  // ascendingOrder
    std::vector<LoessInterval> intervals;
    bool hasFitted = false, hasRw = false;
};

static bool updateBandwidthInterval(double x, const std::vector<double>& xval, int& l, int& r) {  // LoessInterpolator.cs:253-283
    bool updated = false;
    int n = (int)xval.size();
    while (r < n - 1 && x > xval[r]) { l++; r++; updated = true; }
    while (r < n - 1 && xval[r + 1] - x < x - xval[l]) { l++; r++; updated = true; }
    return updated;
}
static inline double tricube(double x) { double t = 1 - x * x * x; return t * t * t; }  // :294-298

static void computeCoefficients(double x, const std::vector<double>& xval, const std::vector<double>& yval, const double* rw,
                                int iLeft, int iRight, double& alpha, double& beta) {  // :187-238
    int edge = (x - xval[iLeft] > xval[iRight] - x) ? iLeft : iRight;
    double sumWeights = 0, sumX = 0, sumXSquared = 0, sumY = 0, sumXY = 0;
    double denom = std::fabs(1.0 / (xval[edge] - x));
    for (int k = iLeft; k <= iRight; ++k) {
        double xk = xval[k], yk = yval[k];
        double dist = std::fabs(x - xk);
        double robustnessWeight = rw ? rw[k] : 1.0;
        double w = tricube(dist * denom) * robustnessWeight;
        double xkw = xk * w;
        sumWeights += w; sumX += xkw; sumXSquared += xk * xkw; sumY += yk * w; sumXY += yk * xkw;
    }
    double meanX = sumX / sumWeights, meanY = sumY / sumWeights, meanXY = sumXY / sumWeights, meanXSquared = sumXSquared / sumWeights;
    if (meanXSquared == meanX * meanX) beta = 0;
    else beta = (meanXY - meanX * meanY) / (meanXSquared - meanX * meanX);
    alpha = meanY - beta * meanX;
}
static inline double loess_predict_poly(double x, double alpha, double beta) {  // :240-249: y = 0 + pow(x,0)*a + pow(x,1)*b
    double y = 0;
    y += 1.0 * alpha;
    y += x * beta;
    return y;
}

// LoessInterpolator.Train (:61-77) + train (:86-169) + computeIntervals (:171-190)
static LoessModel LoessTrain(const std::vector<double>& xv, const std::vector<double>& yv, double bandwidth, int robustnessIters,
                             double xStep, bool computeFitted) {
    LoessModel m;
    int n = (int)xv.size();
    m.order.resize(n);
    std::iota(m.order.begin(), m.order.end(), 0);
    std::stable_sort(m.order.begin(), m.order.end(), [&](int a, int b) { return xv[a] < xv[b]; });  // OrderBy is stable
    m.xs.resize(n); m.ys.resize(n);
    for (int i = 0; i < n; i++) { m.xs[i] = xv[m.order[i]]; m.ys[i] = yv[m.order[i]]; }
    if (n <= 1) { m.fitted = {yv[0]}; m.hasFitted = true; return m; }
    int bandwidthInPoints = (int)std::ceil(bandwidth * n);
    if (robustnessIters > 0) computeFitted = true;
    if (computeFitted) { m.fitted.assign(n, 0.0); m.hasFitted = true; }
    std::vector<double> residuals;
    if (robustnessIters > 0) { residuals.assign(n, 0.0); m.rw.assign(n, 1.0); m.hasRw = true; }
    for (int iter = 0; iter <= robustnessIters; ++iter) {
        int l = 0, r = bandwidthInPoints - 1;
        for (int i = 0; i < n; ++i) {
            double x = m.xs[i];
            if (i > 0) updateBandwidthInterval(x, m.xs, l, r);
            if (computeFitted) {
                double a, b;
                computeCoefficients(x, m.xs, m.ys, m.hasRw ? m.rw.data() : nullptr, l, r, a, b);
                m.fitted[i] = loess_predict_poly(x, a, b);
            }
            if (robustnessIters > 0) residuals[i] = std::fabs(m.ys[i] - m.fitted[i]);
        }
        if (iter == robustnessIters) break;
        double medianResidual = median_copy(residuals);
        if (medianResidual == 0) break;
        for (int i = 0; i < n; ++i) {
            double arg = residuals[i] / (6 * medianResidual);
            double t = 1 - arg * arg;
            m.rw[i] = (arg >= 1) ? 0 : t * t;   // Math.Pow(.,2) := exact square (Q13)
        }
    }
    // computeIntervals
    int l = 0, r = bandwidthInPoints - 1;
    double xMin = -std::numeric_limits<double>::infinity();
    for (double x = m.xs[0]; x <= m.xs[n - 1]; x += xStep) {
        int nl = l, nr = r;
        if (updateBandwidthInterval(x, m.xs, nl, nr)) {
            m.intervals.push_back({xMin, x, l, r});
            xMin = x; l = nl; r = nr;
        }
    }
    m.intervals.push_back({xMin, std::numeric_limits<double>::infinity(), l, r});
    return m;
}

// LoessModel.Predict(double) (:445-459) via FindInterval (:461-481)
static double LoessPredictOne(const LoessModel& m, double x) {
    int iLeft = 0, iRight = (int)m.intervals.size() - 1;
    while (iLeft <= iRight) {
        int iMid = (iLeft + iRight) / 2;
        const LoessInterval& iv = m.intervals[iMid];
        if (x < iv.xmin) iRight = iMid - 1;
        else if (iv.xmax <= x) iLeft = iMid + 1;
        else {
            double a, b;
            computeCoefficients(x, m.xs, m.ys, m.hasRw ? m.rw.data() : nullptr, iv.l, iv.r, a, b);
            return loess_predict_poly(x, a, b);
        }
    }
    return std::numeric_limits<double>::quiet_NaN();
}

// LoessModel.Predict(IEnumerable<double>) (:419-443)
static std::vector<double> LoessPredictMany(const LoessModel& m, const std::vector<double>& xArr) {
    std::vector<int> ord(xArr.size());
    std::iota(ord.begin(), ord.end(), 0);
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return xArr[a] < xArr[b]; });
    std::vector<double> y(xArr.size());
    int idx = 0;
    for (size_t i = 0; i < xArr.size(); i++) {
        double x = xArr[ord[i]];
        while (idx < (int)m.intervals.size() - 1 && m.intervals[idx].xmax <= x) idx++;
        const LoessInterval& iv = m.intervals[idx];
        double a, b;
        computeCoefficients(x, m.xs, m.ys, m.hasRw ? m.rw.data() : nullptr, iv.l, iv.r, a, b);
        y[ord[i]] = loess_predict_poly(x, a, b);
    }
    return y;
}

// Utilities.GoldenSectionSearch (CanvasCommon/Utilities.cs:1014-1044)
template <class F>
static double GoldenSectionSearch(F f, double a, double b, double tol = 1E-5) {
    const double goldenRatio = 0.618034;
    double c = b - goldenRatio * (b - a);
    double d = a + goldenRatio * (b - a);
    double fc = f(c), fd = f(d);
    while (std::fabs(d - c) > tol) {
        if (fc < fd) { b = d; d = c; fd = fc; c = b - goldenRatio * (b - a); fc = f(c); }
        else { a = c; c = d; fc = fd; d = a + goldenRatio * (b - a); fd = f(d); }
    }
    return (b + a) / 2;
}

static double StandardDeviationAll(const std::vector<double>& x) {  // Utilities.cs:265-277
    double sum = 0;
    for (double v : x) sum += v;
    double mu = sum / x.size();
    double s = 0;
    for (double v : x) { double d = v - mu; s += d * d; }
    return std::sqrt(s / (x.size() - 1));
}

static std::vector<double> gc_range(int minGC, int maxGC) {  // Enumerable.Range(minGC, maxGC) — (start, COUNT): Q7
    std::vector<double> v;
    for (int i = 0; i < maxGC; i++) v.push_back((double)(minGC + i));
    return v;
}

// LoessGCNormalizer.objective (LoessGCNormalizer.cs:98-131)
static double loess_objective(double bandwidth, const std::vector<double>& gcs, const std::vector<double>& counts) {
    double medianY = median_copy(counts);
    int minGC = (int)*std::min_element(gcs.begin(), gcs.end());
    int maxGC = (int)*std::max_element(gcs.begin(), gcs.end());
    std::vector<double> normalized(counts.size());
    {
        LoessModel model = LoessTrain(gcs, counts, bandwidth, 0, 1, false);
        auto fittedByGC = LoessPredictMany(model, gc_range(minGC, maxGC));
        for (size_t i = 0; i < normalized.size(); i++) { int gc = (int)gcs[i]; normalized[i] = counts[i] - fittedByGC[gc - minGC] + medianY; }
    }
    std::vector<double> fitted(counts.size());
    {
        LoessModel model = LoessTrain(gcs, normalized, bandwidth, 0, 1, false);
        auto fittedByGC = LoessPredictMany(model, gc_range(minGC, maxGC));
        for (size_t i = 0; i < fitted.size(); i++) { int gc = (int)gcs[i]; fitted[i] = fittedByGC[gc - minGC]; }
    }
    return StandardDeviationAll(fitted);
}

// LoessGCNormalizer.initialize/Normalize (LoessGCNormalizer.cs:36-90) with CanvasClean's log/exp transformers (CanvasClean.cs:147-151)
static void NormalizeByGC_Loess(Bins& bins, const uint8_t* isChrY) {
    std::vector<double> gcs, counts;
    std::vector<int> withoutChrY;
    int i = 0;
    for (size_t b = 0; b < bins.size(); b++) {
        double c = std::log((double)bins.count[b]);
        if (!std::isinf(c)) {
            gcs.push_back((double)bins.gc[b]); counts.push_back(c);
            if (!isChrY[bins.chr[b]]) withoutChrY.push_back(i);
            i++;
        }
    }
    std::vector<double> g2, c2;
    for (int k : withoutChrY) { g2.push_back(gcs[k]); c2.push_back(counts[k]); }
    double minBandwidth = std::max(2.0 / g2.size(), 0.3), maxBandwidth = std::min(1.0, 0.75);
    if (maxBandwidth < minBandwidth) maxBandwidth = minBandwidth;
    double best = GoldenSectionSearch([&](double b) { return loess_objective(b, g2, c2); }, minBandwidth, maxBandwidth);
    double medianY = median_copy(counts);
    int minGC = (int)*std::min_element(gcs.begin(), gcs.end());
    int maxGC = (int)*std::max_element(gcs.begin(), gcs.end());
    LoessModel model = LoessTrain(gcs, counts, best, 0, 1, false);
    auto fittedByGC = LoessPredictMany(model, gc_range(minGC, maxGC));
    for (size_t b = 0; b < bins.size(); b++) {
        int k = std::min((int)fittedByGC.size() - 1, std::max(0, bins.gc[b] - minGC));
        double smoothed = std::log((double)bins.count[b]) - fittedByGC[k] + medianY;
        bins.count[b] = (float)std::exp(smoothed);
    }
}

// CanvasClean.Main (CanvasClean.cs:415-533), manifest == null
int64_t clean(int64_t n, int32_t* chr, int32_t* start, int32_t* stop, float* count, int32_t* gc, int nchr,
              const uint8_t* chrIsAutosome, const uint8_t* chrIsY, uint32_t flags, int minBinsWeighted, double* localSdOut,
              int32_t* stageCounts) {
    Bins bins;
    bins.chr.assign(chr, chr + n); bins.start.assign(start, start + n); bins.stop.assign(stop, stop + n);
    bins.gc.assign(gc, gc + n); bins.count.assign(count, count + n); bins.dev.assign(n, -1.0);
    (void)nchr;
    bool doGC = flags & CLEAN_GCNORM, doSize = flags & CLEAN_FILTSIZE, doOutl = flags & CLEAN_OUTLIERS;
    bool haveLocalSdFile = flags & CLEAN_LOCALSD, loess = flags & CLEAN_LOESS;
    int sc = 0;
    auto note = [&](size_t v) { if (stageCounts) stageCounts[sc] = (int32_t)v; sc++; };
    if (doSize) bins = RemoveBigBins(bins);
    note(bins.size());
    if (doOutl) bins = RemoveOutliers(bins);
    note(bins.size());
    if (haveLocalSdFile && bins.size() < 50000) haveLocalSdFile = false;
    double localSd = -1.0;
    if (haveLocalSdFile) localSd = GetLocalStandardDeviation(bins);
    int varNorm = 0;
    if (doGC) {
        Bins stripped = loess ? bins : RemoveBinsWithExtremeGC(bins, defaultMinNumberOfBinsPerGC, chrIsAutosome, minBinsWeighted);
        if (stripped.size() != 0) {
            bins = stripped;
            if (loess) NormalizeByGC_Loess(bins, chrIsY); else NormalizeByGC(bins, chrIsAutosome);
            if (haveLocalSdFile && bins.size() > 500000) {
                bool v = NormalizeVarianceByGC(bins, chrIsAutosome);
                varNorm = v ? 1 : 0;
                if (v) { if (loess) NormalizeByGC_Loess(bins, chrIsY); else NormalizeByGC(bins, chrIsAutosome); }
            }
        }
    }
    note(bins.size());
    if (haveLocalSdFile) bins = RemoveBinsWithExtremeLocalSD(bins, localSd, 20);
    note(bins.size());
    if (stageCounts) stageCounts[sc] = varNorm;
    if (localSdOut) *localSdOut = localSd;
    int64_t m = (int64_t)bins.size();
    std::copy(bins.chr.begin(), bins.chr.end(), chr); std::copy(bins.start.begin(), bins.start.end(), start);
    std::copy(bins.stop.begin(), bins.stop.end(), stop); std::copy(bins.gc.begin(), bins.gc.end(), gc);
    std::copy(bins.count.begin(), bins.count.end(), count);
    return m;
}

// exported for golden tests
void loess_fit(const double* x, const double* y, int n, double bandwidth, int robIters, double xStep, double* fittedOrig, double* predicted) {
    std::vector<double> xv(x, x + n), yv(y, y + n);
    LoessModel m = LoessTrain(xv, yv, bandwidth, robIters, xStep, true);
    // model.Fitted: OriginalOrder.Select(i => SortedFitted[i]) with OriginalOrder[ascendingOrder[i]] = i
    for (int i = 0; i < n; i++) fittedOrig[m.order[i]] = m.fitted[i];
    if (predicted) for (int i = 0; i < n; i++) predicted[i] = LoessPredictOne(m, x[i]);
}
double golden_section_square(double a, double b) { return GoldenSectionSearch([](double x) { return x * x; }, a, b); }

}  // namespace oracle
