// TEST INFRASTRUCTURE ONLY — CPU restatement of CanvasPartition's Wavelets method (the reference's default -m):
// unbalanced Haar decomposition, thresholding, reconstruction, healing of badly supported splits and (germline) refinement,
// plus the coverage-variability inputs WaveletsRunner.Run feeds it.  Statement by statement after
//   CanvasPartition/WaveletSegmentation.cs:19-428, WaveletsRunner.cs:52-150, Segmentation.cs:297-429 (SegmentationInput),
//   CanvasCommon/Utilities.cs:340-462 (Median / Mad), Segmentation.cs:83-125 (DeriveSegments).
// Pinned by the reference's own known-answer test (CanvasTest/CanvasPartition/WaveletTests.cs: 530 bins -> 12 breakpoints;
// tests/golden/wavelets_minimal.json) for the non-germline flavour.  The germline flavour additionally depends on the tie order
// of .NET's unstable Array.Sort (HardThresh, WaveletSegmentation.cs:84) — restated from coreclr's introsort, parity unpinned —
// and both depend on the platform's log() (Math.Log -> libm on Linux).
#include "oracle_common.h"
#include "oracle_api.h"
#include <atomic>
#include <thread>
#include <functional>

namespace oracle {
void Quartiles(const std::vector<float>& x, float& fQ1, float& fQ2, float& fQ3);   // oracle_bin_clean.cpp (Utilities.cs:361-419)

namespace wv {

// SortedList<T>(x).Median(): NaN sorts in front of every number under .NET's default comparer
template <class T>
static T dotnet_median(std::vector<T> v) {
    auto mid = std::partition(v.begin(), v.end(), [](T a) { return a != a; });
    std::sort(mid, v.end());
    size_t n = v.size();
    if (n == 0) return T(0);
    if (n % 2 == 1) return v[n / 2];
    return (T)((v[n / 2 - 1] + v[n / 2]) / (T)2);
}
static double MedianRange(const double* x, int64_t start, int64_t end) { return dotnet_median(std::vector<double>(x + start, x + end)); }
static double MadRange(const double* x, int64_t start, int64_t end) {            // Utilities.cs:451-462
    double median = MedianRange(x, start, end);
    std::vector<double> d((size_t)(end - start));
    for (int64_t i = start; i < end; i++) d[(size_t)(i - start)] = std::fabs(x[i] - median);
    return dotnet_median(d);
}

// SegmentationInput.reportVariabilityByWindow (Segmentation.cs:334-349)
static std::vector<float> VariabilityByWindow(int windowSize, int nchr, const double* cov, const int64_t* off) {
    std::vector<float> out;
    for (int c = 0; c < nchr; c++) {
        const double* x = cov + off[c]; const int64_t L = off[c + 1] - off[c];
        for (int64_t index = 0; index < L - windowSize; index += windowSize) {
            double MAD = MadRange(x, index, index + windowSize);
            double median = MedianRange(x, index, index + windowSize);
            out.push_back((float)(MAD / median));                                // Convert.ToSingle
        }
    }
    return out;
}
// SegmentationInput.GetCoverageVariability (Segmentation.cs:308-328); returns false for null
static bool CoverageVariability(int windowSize, int nchr, const double* cov, const int64_t* off, double& cv) {
    if (off[nchr] - off[0] < 10 * (int64_t)windowSize) return false;
    const int windowSizeIQR = 10000;
    if (windowSize > windowSizeIQR) {
        const double IQRthreshold = 0.015;
        std::vector<float> rv = VariabilityByWindow(windowSizeIQR, nchr, cov, off);
        float q1, q2, q3; Quartiles(rv, q1, q2, q3);
        if ((q3 - q1) / q2 > IQRthreshold) { cv = q1; return true; }
    }
    std::vector<float> rv = VariabilityByWindow(windowSize, nchr, cov, off);
    cv = (double)dotnet_median(rv);                                               // Median(IEnumerable<float>)
    return true;
}
// SegmentationInput.FactorOfThreeCoverageVariabilities (Segmentation.cs:366-402) + GetTripletMediansAndCMADs (404-429)
static std::vector<double> FactorOfThree(int nchr, const double* cov, const int64_t* off, int maxExponent = 8) {
    std::vector<double> f3{0.0};
    std::vector<std::vector<double>> results(nchr);
    for (int c = 0; c < nchr; c++) results[c].assign(cov + off[c], cov + off[c + 1]);
    int exponent = 1;
    while (exponent <= maxExponent) {
        std::vector<double> CMADs;
        for (int c = 0; c < nchr; c++) {
            const std::vector<double>& data = results[c];
            int n = (int)data.size() / 3;
            std::vector<double> med((size_t)n);
            for (int i = 0; i < n; i++) {
                int j = i * 3 + 1;
                double a = data[j - 1], b = data[j], cc = data[j + 1];
                if (a > b) std::swap(a, b);
                if (a > cc) std::swap(a, cc);
                if (b > cc) std::swap(b, cc);
                med[i] = b;
                CMADs.push_back((cc - a) / 2.0 / b);
            }
            results[c] = med;
        }
        if ((int)CMADs.size() < 50) {
            int add = maxExponent - (int)f3.size() + 1;
            double last = f3.back();
            for (int i = 0; i < add; i++) f3.push_back(last);
            break;
        }
        f3.push_back(dotnet_median(CMADs));
        ++exponent;
    }
    return f3;
}

// WaveletSegmentation.GetInnerProdIter (WaveletSegmentation.cs:19-48)
static void GetInnerProdIter(const double* x, int64_t n, std::vector<double>& I_prod, double& mean) {
    std::vector<double> I_plus((size_t)(n - 1)), I_minus((size_t)(n - 1));
    I_plus[0] = std::sqrt(1 - 1.0 / n) * x[0];
    double sumX = 0;
    for (int64_t i = 1; i < n; i++) sumX += x[i];
    mean = (x[0] + sumX) / n;
    I_minus[0] = (1.0 / std::sqrt((double)(n * (n - 1)))) * sumX;
    if (n > 2) {
        for (int64_t m = 1; m < n - 1; m++) {
            double factor = std::sqrt((double)(n - m - 1) * (double)m / (double)(m + 1) / (double)(n - m));
            I_plus[m] = I_plus[m - 1] * factor + x[m] * std::sqrt(1.0 / (double)(m + 1) - 1.0 / n);
            I_minus[m] = I_minus[m - 1] / factor - x[m] / std::sqrt(((double)n * n / (double)(m + 1)) - (double)n);
        }
    }
    I_prod.resize((size_t)(n - 1));
    for (int64_t i = 0; i < n - 1; i++) I_prod[i] = I_plus[i] - I_minus[i];
}
// Enumerable.Max over |ipi| (leading NaNs skipped, later NaNs never win), then the first index that equals it (cs:54-68)
static int GetInnerProdMax(const std::vector<double>& ipi) {
    size_t i = 0; double mx = std::fabs(ipi[0]);
    while (mx != mx && i + 1 < ipi.size()) mx = std::fabs(ipi[++i]);
    for (size_t k = i + 1; k < ipi.size(); k++) { double a = std::fabs(ipi[k]); if (a > mx) mx = a; }
    int index = 0;
    for (; index < (int)ipi.size(); index++) if (std::fabs(ipi[index]) == mx) break;
    return index + 1;
}

typedef std::vector<std::vector<double>> Tree;

// WaveletSegmentation.FindBestUnbalancedHaarDecomposition (cs:252-366)
static double FindBestUnbalancedHaarDecomposition(const double* x, int64_t n, Tree& tree) {
    tree.clear();
    std::vector<double> ipi; double mean;
    GetInnerProdIter(x, n, ipi, mean);
    int ind_max = GetInnerProdMax(ipi);
    const double meanscale = 200.0;
    tree.push_back({1.0, ipi[ind_max - 1] / std::max(0.5, mean / meanscale), 1.0, (double)ind_max, (double)n});
    size_t j = 0;
    double bpSum = 0;
    for (size_t i = 0; i < tree[j].size() / 5; i++) bpSum += tree[j][5 + i * 5 - 1] - tree[j][3 + i * 5 - 1] - 1.0;
    while (bpSum != 0) {
        size_t parents = tree[j].size() / 5;
        std::vector<double> next;
        for (size_t i = 0; i < parents; i++) {
            const double idx = tree[j][5 * i], s = tree[j][5 * i + 2], b = tree[j][5 * i + 3], e = tree[j][5 * i + 4];
            if (b - s >= 1) {
                int64_t skip = (int64_t)s - 1, take = (int64_t)b - skip;
                double m2; GetInnerProdIter(x + skip, take, ipi, m2);
                ind_max = GetInnerProdMax(ipi);
                next.insert(next.end(), {2 * idx - 1, ipi[ind_max - 1] / std::max(0.5, m2 / meanscale), s, ind_max + s - 1, b});
            }
            if (e - b >= 2) {
                int64_t skip = (int64_t)b, take = (int64_t)e - skip;
                double m2; GetInnerProdIter(x + skip, take, ipi, m2);
                ind_max = GetInnerProdMax(ipi);
                next.insert(next.end(), {2 * idx, ipi[ind_max - 1] / std::max(0.5, m2 / meanscale), b + 1, ind_max + b, e});
            }
        }
        tree.push_back(next);      // (the C# would throw on an empty level; bpSum != 0 guarantees at least one child)
        j++;
        bpSum = 0;
        for (size_t k = 0; k < tree[j].size() / 5; k++) bpSum += tree[j][5 + k * 5 - 1] - tree[j][3 + k * 5 - 1] - 1;
    }
    double smooth = 0;
    for (int64_t i = 0; i < n; i++) smooth += x[i];
    return smooth / std::sqrt((double)n);
}

// Array.Sort<int>(indices, comparison) of .NET Core 2.0: ArraySortHelper<T>.IntrospectiveSort with a Comparison<T>
namespace isort {
typedef std::function<int(int, int)> Cmp;
static void swap_if_greater(int* k, const Cmp& c, int a, int b) { if (a != b && c(k[a], k[b]) > 0) std::swap(k[a], k[b]); }
static void insertion_sort(int* k, int lo, int hi, const Cmp& c) {
    for (int i = lo; i < hi; i++) { int j = i; int t = k[i + 1]; while (j >= lo && c(t, k[j]) < 0) { k[j + 1] = k[j]; j--; } k[j + 1] = t; }
}
static void down_heap(int* k, int i, int n, int lo, const Cmp& c) {
    int d = k[lo + i - 1];
    while (i <= n / 2) {
        int child = 2 * i;
        if (child < n && c(k[lo + child - 1], k[lo + child]) < 0) child++;
        if (!(c(d, k[lo + child - 1]) < 0)) break;
        k[lo + i - 1] = k[lo + child - 1];
        i = child;
    }
    k[lo + i - 1] = d;
}
static void heap_sort(int* k, int lo, int hi, const Cmp& c) {
    int n = hi - lo + 1;
    for (int i = n / 2; i >= 1; i--) down_heap(k, i, n, lo, c);
    for (int i = n; i > 1; i--) { std::swap(k[lo], k[lo + i - 1]); down_heap(k, 1, i - 1, lo, c); }
}
static int pick_pivot_and_partition(int* k, int lo, int hi, const Cmp& c) {
    int mid = lo + (hi - lo) / 2;
    swap_if_greater(k, c, lo, mid); swap_if_greater(k, c, lo, hi); swap_if_greater(k, c, mid, hi);
    int pivot = k[mid];
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
static void intro_sort(int* k, int lo, int hi, int depthLimit, const Cmp& c) {
    while (hi > lo) {
        int partitionSize = hi - lo + 1;
        if (partitionSize <= 16) {
            if (partitionSize == 1) return;
            if (partitionSize == 2) { swap_if_greater(k, c, lo, hi); return; }
            if (partitionSize == 3) { swap_if_greater(k, c, lo, hi - 1); swap_if_greater(k, c, lo, hi); swap_if_greater(k, c, hi - 1, hi); return; }
            insertion_sort(k, lo, hi, c);
            return;
        }
        if (depthLimit == 0) { heap_sort(k, lo, hi, c); return; }
        depthLimit--;
        int p = pick_pivot_and_partition(k, lo, hi, c);
        intro_sort(k, p + 1, hi, depthLimit, c);
        hi = p - 1;
    }
}
static void sort(std::vector<int>& k, const Cmp& c) {
    int n = (int)k.size();
    if (n < 2) return;
    int fl = 0; for (int v = n; v >= 1; v /= 2) fl++;
    intro_sort(k.data(), 0, n - 1, 2 * fl, c);
}
}  // namespace isort

// WaveletSegmentation.HardThresh (cs:73-117)
static void HardThresh(Tree& tree, double sigma, bool isGermline) {
    int treeSize = (int)tree.size();
    std::vector<double> thresholds;
    std::vector<int> indices((size_t)treeSize);
    if (isGermline) {
        std::vector<int> counts((size_t)treeSize);
        for (int i = 0; i < treeSize; i++) { counts[i] = (int)std::floor(tree[i].size() / 5.0); indices[i] = i; }
        isort::sort(indices, [&](int a, int b) { return counts[b] < counts[a] ? -1 : (counts[b] > counts[a] ? 1 : 0); });   // counts[b].CompareTo(counts[a])
        const double NewMax = 1.0, NewMin = 0.8;
        for (int x = 1; x <= treeSize; x++) thresholds.push_back(((double)x * (NewMax - NewMin)) / treeSize + NewMin);
    } else {
        for (int i = 0; i < treeSize; i++) { thresholds.push_back(1.0); indices[i] = i; }
    }
    const int subtreeSize = 5;
    double n = tree[0][subtreeSize - 1];
    for (int node = 0; node < treeSize; node++) {
        int K = (int)std::floor(tree[node].size() / 5.0);
        for (int k = 0; k < K; k++)
            if (std::fabs(tree[node][k * subtreeSize + 1]) <= 2 * sigma * (thresholds[indices[node]]) * std::sqrt(2 * std::log(n)))
                tree[node][k * subtreeSize + 1] = 0;
    }
}
// GetUnbalHaarVector + GetReconstructedVector + GetSegments (cs:120-185)
static void GetSegments(const Tree& tree, double smooth, std::vector<int>& breakpoints) {
    int n = (int)tree[0][4];
    std::vector<double> rec((size_t)n);
    for (int i = 0; i < n; i++) rec[i] = 1.0 / std::sqrt((double)n) * smooth;
    for (size_t j = 0; j < tree.size(); j++) {
        size_t K = tree[j].size() / 5;
        for (size_t k = 0; k < K; k++) {
            const double a0 = tree[j][k * 5 + 2], a1 = tree[j][k * 5 + 3], a2 = tree[j][k * 5 + 4], coef = tree[j][k * 5 + 1];
            double nn = a2 - a0 + 1, m = a1 - a0 + 1;
            double val1 = std::sqrt(1 / m - 1 / nn), val2 = -1.0 / std::sqrt(nn * nn / m - nn);
            int s = (int)a0 - 1;
            for (int i = s; i < a2; i++) rec[i] = rec[i] + ((double)(i - s) < m ? val1 : val2) * coef;
        }
    }
    breakpoints.push_back(0);
    for (int i = 1; i < n; i++) if (rec[i] - rec[i - 1] != 0) breakpoints.push_back(i);
}
// GetBreakpointsAfterHealingBadSplits (cs:195-225)
static void Heal(std::vector<int>& breakpoints, const std::vector<int>& prelim, const double* ratio, int N, const std::vector<double>& f3) {
    int L = (int)prelim.size();
    breakpoints.push_back(prelim[0]);
    for (int i = 1; i < L; ++i) {
        int leftStart = breakpoints.back(), rightStart = prelim[i], rightEnd = (i < L - 1) ? prelim[i + 1] : N;
        int leftLength = rightStart - leftStart, rightLength = rightEnd - rightStart;
        double leftMedian = MedianRange(ratio, leftStart, leftStart + leftLength);
        double rightMedian = MedianRange(ratio, rightStart, rightStart + rightLength);
        double weightedMedian = (leftLength * leftMedian + rightLength * rightMedian) / (rightEnd - leftStart);
        int smallerLength = std::min(leftLength, rightLength);
        int scale = std::min((int)f3.size() - 1, (int)std::ceil(std::log((double)smallerLength) / std::log(3.0)));
        double cutoff = f3[scale];
        if (std::fabs(leftMedian - rightMedian) > cutoff * 4 * std::max(weightedMedian, 50.0)) breakpoints.push_back(prelim[i]);
    }
}
// RefineSegments (cs:230-250)
static void Refine(std::vector<int>& bp, const double* cov, int N) {
    const int halfWindow = 5;
    double totalMedian = MedianRange(cov, 0, N);
    for (int i = 1; i < (int)bp.size() - 1; i++) {
        int leftInterval = std::min(halfWindow, (bp[i] - bp[i - 1]) / 2);
        int rightInterval = std::min(halfWindow, (bp[i + 1] - bp[i]) / 2);
        double best = std::fabs(MedianRange(cov, bp[i - 1], bp[i]) - totalMedian);
        int bestBp = bp[i];
        for (int j = bp[i] - leftInterval; j < bp[i] + rightInterval; j++) {
            double t = std::fabs(MedianRange(cov, bp[i - 1], j) - totalMedian);
            if (t > best) { best = t; bestBp = j; }
        }
        bp[i] = bestBp;
    }
}
// WaveletSegmentation.HaarWavelets (cs:373-425)
static void HaarWavelets(const double* ratio, int N, double thresholdlower, double thresholdupper, std::vector<int>& breakpoints, bool isGermline,
                         double madFactor, bool hasCV, double cv, const std::vector<double>& f3) {
    Tree tree;
    double smooth = FindBestUnbalancedHaarDecomposition(ratio, N, tree);
    double median = MedianRange(ratio, 0, N);
    double variabilityMeasure = hasCV ? median * cv : MadRange(ratio, 0, N);
    double threshold = madFactor * variabilityMeasure;
    if (threshold < thresholdlower) threshold = thresholdlower;
    if (threshold > thresholdupper) threshold = thresholdupper;
    HardThresh(tree, threshold, isGermline);
    std::vector<int> prelim;
    GetSegments(tree, smooth, prelim);
    Heal(breakpoints, prelim, ratio, N, f3);
    if (isGermline) Refine(breakpoints, ratio, N);
}
}  // namespace wv
}  // namespace oracle

using namespace oracle;

extern "C" {
// one chromosome, explicit CV / factor-of-three inputs (the shape of WaveletTests.MinimalWaveletTest)
int64_t orc_haar_wavelets(const double* x, int64_t n, double thr_lower, double thr_upper, int is_germline, double mad_factor, int has_cv, double cv,
                          const double* f3, int nf3, int32_t* out, int64_t cap) {
    std::vector<int> bp;
    wv::HaarWavelets(x, (int)n, thr_lower, thr_upper, bp, is_germline != 0, mad_factor, has_cv != 0, cv, std::vector<double>(f3, f3 + nf3));
    if ((int64_t)bp.size() > cap) return -1;
    for (size_t i = 0; i < bp.size(); i++) out[i] = bp[i];
    return (int64_t)bp.size();
}
int orc_coverage_variability(int window, int nchr, const double* cov, const int64_t* off, double* cv) { return wv::CoverageVariability(window, nchr, cov, off, *cv) ? 1 : 0; }
int orc_factor_of_three(int nchr, const double* cov, const int64_t* off, double* out9) {
    std::vector<double> f = wv::FactorOfThree(nchr, cov, off);
    for (size_t i = 0; i < f.size() && i < 9; i++) out9[i] = f[i];
    return (int)f.size();
}
// WaveletsRunner.Run up to the breakpoints (WaveletsRunner.cs:52-150): out_off[c]..out_off[c+1] = breakpoints of chromosome c
int64_t orc_wavelets(int nchr, const double* cov, const int64_t* off, int is_germline, double thr_lower, double thr_upper, double mad_factor, int window,
                     int min_size, int32_t* out, int64_t cap, int64_t* out_off) {
    double cv = 0; bool hasCV = wv::CoverageVariability(window, nchr, cov, off, cv);
    std::vector<double> f3 = wv::FactorOfThree(nchr, cov, off);
    int64_t total = 0;
    for (int c = 0; c < nchr; c++) {
        out_off[c] = total;
        const int64_t L = off[c + 1] - off[c];
        std::vector<int> bp;
        if (std::max<int64_t>(L, 1) > min_size) wv::HaarWavelets(cov + off[c], (int)L, thr_lower, thr_upper, bp, is_germline != 0, mad_factor, hasCV, cv, f3);
        if (total + (int64_t)bp.size() > cap) return -1;
        for (int v : bp) out[total++] = v;
    }
    out_off[nchr] = total;
    return total;
}
// the same with one task per chromosome on `threads` host threads, as WaveletsRunner.Run does (Parallel.ForEach over the chromosomes, WaveletsRunner.cs:89-135):
// the timing baseline of bench.py; results are those of orc_wavelets
int64_t orc_wavelets_threads(int nchr, const double* cov, const int64_t* off, int is_germline, double thr_lower, double thr_upper, double mad_factor, int window,
                             int min_size, int32_t* out, int64_t cap, int64_t* out_off, int threads) {
    double cv = 0; bool hasCV = wv::CoverageVariability(window, nchr, cov, off, cv);
    std::vector<double> f3 = wv::FactorOfThree(nchr, cov, off);
    std::vector<std::vector<int>> bps((size_t)nchr);
    std::atomic<int> next{0};
    auto worker = [&]() {
        for (int c = next.fetch_add(1); c < nchr; c = next.fetch_add(1)) {
            const int64_t L = off[c + 1] - off[c];
            if (std::max<int64_t>(L, 1) > min_size) wv::HaarWavelets(cov + off[c], (int)L, thr_lower, thr_upper, bps[(size_t)c], is_germline != 0, mad_factor, hasCV, cv, f3);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < std::max(1, std::min(threads, nchr)); t++) pool.emplace_back(worker);
    worker();
    for (auto& t : pool) t.join();
    int64_t total = 0;
    for (int c = 0; c < nchr; c++) {
        out_off[c] = total;
        if (total + (int64_t)bps[(size_t)c].size() > cap) return -1;
        for (int v : bps[(size_t)c]) out[total++] = v;
    }
    out_off[nchr] = total;
    return total;
}
}
