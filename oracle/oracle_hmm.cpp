// TEST INFRASTRUCTURE ONLY (see oracle_common.h). CPU restatement of CanvasPartition's HMM / PerSampleHMM path.
// Paths relative to /root/reference/Src/Canvas/.
#include "oracle_common.h"
#include "oracle_api.h"
#include "oracle_partition.h"

namespace oracle {

// CanvasCommon/DistributionUtilities.cs:51-69 (adjustClumpingParameter = false). GammaLn/FactorialLn are MathNet
// (not in /root/reference; parity unpinned) -> lgamma.  Math.Pow(x,2) := x*x (Q13).
std::vector<double> NegativeBinomialWrapper(double mean, double variance, int maxValue) {
    std::vector<double> density(maxValue > 0 ? maxValue : 0, 0.0);
    double m = std::max(mean, 0.1);
    double clumpingParameter = (m * m) / (std::max(variance, mean * 1.2) - mean);
    clumpingParameter = std::max(2.0, clumpingParameter);
    for (int x = 0; x < maxValue; x++) {
        double tmpDensity = std::exp(std::log(std::pow(1 + mean / clumpingParameter, -clumpingParameter)) +
                                     std::log(std::pow(mean / (mean + clumpingParameter), (double)x)) +
                                     std::lgamma(clumpingParameter + x) - std::lgamma((double)x + 1.0) - std::lgamma(clumpingParameter));
        density[x] = (std::isnan(tmpDensity) || std::isinf(tmpDensity)) ? 0 : tmpDensity;
    }
    return density;
}

// CanvasCommon/DistributionUtilities.cs:11-40 ; permutation order = Combinatorics/Permutations.cs:399-433 (sort, then
// lexicographic next-permutation over distinct arrangements).
std::vector<std::vector<int>> GetGenotypeCombinations(int numberOfStates, int currentState) {
    const int diploidState = 2, maxNumberOfStates = 4;
    if (numberOfStates > maxNumberOfStates) numberOfStates = maxNumberOfStates;
    std::vector<std::vector<int>> all;
    if (currentState == diploidState) { all.push_back(std::vector<int>(numberOfStates, diploidState)); return all; }
    for (int nd = 0; nd < numberOfStates; nd++) {
        std::vector<int> states(numberOfStates - nd, currentState);
        states.insert(states.end(), nd, diploidState);
        std::sort(states.begin(), states.end());
        do { all.push_back(states); } while (std::next_permutation(states.begin(), states.end()));
    }
    if (all.empty()) all.push_back({currentState});
    return all;
}

struct HmmModel {
    int nStates = 5, nSamples = 1;
    bool perSample = true;
    // pmf[state][sample][x]
    std::vector<std::vector<std::vector<double>>> pmf;
    std::vector<std::vector<std::vector<int>>> combos;  // per state
    double A[5][5];
    double pi[5];
};

// CanvasPartition/Distributions.cs:257-323
static double EstimateViterbiLikelihood(const HmmModel& m, const double* data /*nSamples*/, int currentCnState, const double* transition) {
    double maxLikelyhood = -std::numeric_limits<double>::max();  // Double.MinValue
    const std::vector<int>* bestState = nullptr;
    static const std::vector<int> empty;
    bestState = &empty;
    for (const auto& perm : m.combos[currentCnState]) {
        double emissionLikelihood = 1.0;
        int s = 0;
        for (int cnGenotype : perm) {
            int pointCoverage = to_int32_round(data[s]);
            if (m.perSample) emissionLikelihood *= m.pmf[cnGenotype][s][pointCoverage];
            else {
                if (cnGenotype == 0 || cnGenotype == 1)
                    emissionLikelihood *= std::max(m.pmf[0][s][pointCoverage], m.pmf[1][s][pointCoverage]);
                else if (cnGenotype == 3 || cnGenotype == 4)
                    emissionLikelihood *= std::max(m.pmf[3][s][pointCoverage], m.pmf[4][s][pointCoverage]);
                else emissionLikelihood *= m.pmf[cnGenotype][s][pointCoverage];
            }
            s++;
        }
        if (std::isnan(emissionLikelihood) || std::isinf(emissionLikelihood)) emissionLikelihood = 0;
        if (maxLikelyhood < emissionLikelihood) { bestState = &perm; maxLikelyhood = emissionLikelihood; }
    }
    double transitionLikelihood = 1.0;
    double tmax = transition[0];
    for (int i = 1; i < m.nStates; i++) tmax = std::max(tmax, transition[i]);
    bool transitionFromDiploid = tmax == transition[2];
    auto minOver = [&](bool skipDiploid) {
        double mn = std::numeric_limits<double>::infinity();
        bool any = false;
        for (int st : *bestState) { if (skipDiploid && st == 2) continue; mn = any ? std::min(mn, transition[st]) : transition[st]; any = true; }
        return mn;  // C# Min() on empty throws
    };
    if (transitionFromDiploid) transitionLikelihood = minOver(false);
    else if (currentCnState == 2) transitionLikelihood = transition[2];
    else transitionLikelihood = minOver(true);
    return std::log(maxLikelyhood) + std::log(transitionLikelihood);
}

// HiddenMarkovModelsRunner.InitializeNegativeBinomialEmission (HiddenMarkovModelsRunner.cs:111-152) + RemoveOutliers (:154-162)
// + HiddenMarkovModel ctor (HMM.cs:24-51). `data` is [T][nSamples] row-major and is capped in place (the reference caps a copy).
static void build_model(HmmModel& m, std::vector<double>& data, int T, int nSamples, bool perSample,
                        const double* medians, const double* pseudoVariances) {
    m.nSamples = nSamples; m.perSample = perSample;
    std::vector<double> haploidMean, variance;
    for (int d = 0; d < nSamples; d++) {
        if (!perSample) {
            std::vector<double> col(T);
            for (int t = 0; t < T; t++) col[t] = data[(size_t)t * nSamples + d];
            double median = std::max(1.0, median_copy(col));
            haploidMean.push_back(median / 2.0);
            double sum = 0; for (double v : col) sum += v;           // Utilities.Variance (Utilities.cs:287-300)
            double mu = sum / col.size(), s2 = 0;
            for (double v : col) { double df = v - mu; s2 += df * df; }
            variance.push_back(s2 / (col.size() - 1));
        } else {
            haploidMean.push_back(medians[d] / 2.0);
            variance.push_back(pseudoVariances[d]);
        }
    }
    double maxThreshold = *std::max_element(haploidMean.begin(), haploidMean.end()) * m.nStates;
    for (auto& v : data) v = v > maxThreshold ? maxThreshold : v;
    int maxValues = std::numeric_limits<int>::min();
    for (int t = 0; t < T; t++) {
        double mx = data[(size_t)t * nSamples];
        for (int d = 1; d < nSamples; d++) mx = std::max(mx, data[(size_t)t * nSamples + d]);
        maxValues = std::max(maxValues, to_int32_round(mx));
    }
    m.pmf.assign(m.nStates, {});
    for (int CN = 0; CN < m.nStates; CN++)
        for (int d = 0; d < nSamples; d++)
            m.pmf[CN].push_back(NegativeBinomialWrapper(std::max((double)CN, 0.1) * haploidMean[d], variance[d], maxValues + 10));
    m.combos.clear();
    for (int CN = 0; CN < m.nStates; CN++) m.combos.push_back(GetGenotypeCombinations(nSamples, CN));
    const double selfTransition = 0.99;
    for (int i = 0; i < m.nStates; i++) {
        for (int j = 0; j < m.nStates; j++) m.A[i][j] = (i == j) ? selfTransition : (1.0 - selfTransition) / (m.nStates - 1);
        m.pi[i] = (double)(1.0f / m.nStates);   // HMM.cs:41, float then widened (Q12)
    }
}

// HiddenMarkovModel.BestPathViterbi (HMM.cs:62-130)
static std::vector<int> BestPathViterbi(const HmmModel& m, const std::vector<double>& x, int size) {
    int nS = m.nStates, S = m.nSamples;
    std::vector<double> bestScore((size_t)size * nS);
    std::vector<int8_t> bp((size_t)size * nS);
    for (int j = 0; j < nS; j++) {
        bestScore[j] = std::log(m.pi[j]) + EstimateViterbiLikelihood(m, &x[0], j, m.A[0]) - std::log(m.A[0][j]);
        bp[j] = -1;
    }
    for (int t = 1; t < size; t++) {
        for (int j = 0; j < nS; j++) {
            int state = 0;
            double mx = -std::numeric_limits<double>::max();
            for (int i = 0; i < nS; i++) {
                double vitLogL = EstimateViterbiLikelihood(m, &x[(size_t)t * S], j, m.A[i]);
                double tmpMax = bestScore[(size_t)(t - 1) * nS + i] + vitLogL;
                if (tmpMax > mx) { state = i; mx = tmpMax; }
            }
            bestScore[(size_t)t * nS + j] = mx;
            bp[(size_t)t * nS + j] = (int8_t)state;
        }
    }
    int bestState = -1;
    double max1 = -std::numeric_limits<double>::max();
    for (int i = 0; i < nS; i++) {
        double v = bestScore[(size_t)(size - 1) * nS + i];
        if (v > max1) { bestState = i; max1 = v; }
    }
    std::vector<int> states(size);
    int backtrack = size - 1;
    while (backtrack > 0) {
        states[backtrack] = bestState;
        bestState = bestState < 0 ? -1 : bp[(size_t)backtrack * nS + bestState];  // C# would throw on -1
        backtrack--;
    }
    states[0] = bestState;
    return states;
}

// HiddenMarkovModelsRunner.Run (HiddenMarkovModelsRunner.cs:23-109), one chromosome.
// cov: [nSamples][T] pointers. medians/pseudoVariances: per-sample global values (perSample) or null.
// Returns 0 and leaves path untouched when T <= minSize (chromosome skipped, :69).
int hmm_chromosome(int nSamples, bool perSample, const double* const* cov, int T, const double* medians,
                   const double* pseudoVariances, int32_t* path) {
    const int minSize = 10;
    if (!(T > minSize)) return 0;
    std::vector<double> data((size_t)T * nSamples);
    for (int t = 0; t < T; t++) for (int d = 0; d < nSamples; d++) data[(size_t)t * nSamples + d] = cov[d][t];
    HmmModel m;
    build_model(m, data, T, nSamples, perSample, medians, pseudoVariances);
    std::vector<int> st = BestPathViterbi(m, data, T);
    for (int t = 0; t < T; t++) path[t] = st[t];
    return 1;
}

// global per-sample quartiles (HiddenMarkovModelsRunner.cs:36-50): float Quartiles over all chromosomes' coverage cast to float
void hmm_global_params(int nchr, const double* const* cov, const int64_t* n, double* median, double* pseudoVariance) {
    std::vector<float> v;
    for (int c = 0; c < nchr; c++) for (int64_t i = 0; i < n[c]; i++) v.push_back((float)cov[c][i]);
    float q1, q2, q3;
    Quartiles(v, q1, q2, q3);
    *median = (double)q2;
    float iqr = q3 - q1;
    *pseudoVariance = (double)(iqr * iqr);
}

}  // namespace oracle
