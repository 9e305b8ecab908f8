// TEST INFRASTRUCTURE ONLY (see oracle_common.h). CPU restatement of CanvasPartition's CBS path.
// Paths relative to /root/reference/Src/Canvas/CanvasPartition/.
#include "oracle_common.h"
#include "oracle_partition.h"

namespace oracle {

static inline double sq(double v) { return v * v; }   // Math.Pow(v, 2) := exact square

// ------------------------------------------------------------------ Array.Sort<double,int> (.NET Core 2.0 introsort)
// Helper.QuickSort (Helper.cs:283-298) forwards to Array.Sort(keys, items, index, length): an unstable introspective sort
// (coreclr ArraySortHelper<TKey,TValue>, restated from its published algorithm; tie order is parity-unpinned, Q11).
namespace {
struct KI { double* k; int* v; };
inline void swp(KI a, int i, int j) { if (i != j) { std::swap(a.k[i], a.k[j]); std::swap(a.v[i], a.v[j]); } }
inline void swap_if_greater(KI a, int i, int j) { if (i != j && a.k[i] > a.k[j]) { std::swap(a.k[i], a.k[j]); std::swap(a.v[i], a.v[j]); } }
void insertion_sort(KI a, int lo, int hi) {
    for (int i = lo; i < hi; i++) {
        int j = i;
        double t = a.k[i + 1]; int tv = a.v[i + 1];
        while (j >= lo && t < a.k[j]) { a.k[j + 1] = a.k[j]; a.v[j + 1] = a.v[j]; j--; }
        a.k[j + 1] = t; a.v[j + 1] = tv;
    }
}
void down_heap(KI a, int i, int n, int lo) {
    double d = a.k[lo + i - 1]; int dv = a.v[lo + i - 1];
    while (i <= n / 2) {
        int child = 2 * i;
        if (child < n && a.k[lo + child - 1] < a.k[lo + child]) child++;
        if (a.k[lo + child - 1] < d) break;
        a.k[lo + i - 1] = a.k[lo + child - 1]; a.v[lo + i - 1] = a.v[lo + child - 1];
        i = child;
    }
    a.k[lo + i - 1] = d; a.v[lo + i - 1] = dv;
}
void heap_sort(KI a, int lo, int hi) {
    int n = hi - lo + 1;
    for (int i = n / 2; i >= 1; i--) down_heap(a, i, n, lo);
    for (int i = n; i > 1; i--) { swp(a, lo, lo + i - 1); down_heap(a, 1, i - 1, lo); }
}
int pick_pivot_and_partition(KI a, int lo, int hi) {
    int mid = lo + (hi - lo) / 2;
    swap_if_greater(a, lo, mid); swap_if_greater(a, lo, hi); swap_if_greater(a, mid, hi);
    double pivot = a.k[mid];
    swp(a, mid, hi - 1);
    int left = lo, right = hi - 1;
    while (left < right) {
        while (pivot > a.k[++left]) ;
        while (pivot < a.k[--right]) ;
        if (left >= right) break;
        swp(a, left, right);
    }
    swp(a, left, hi - 1);
    return left;
}
void intro_sort(KI a, int lo, int hi, int depthLimit) {
    while (hi > lo) {
        int partitionSize = hi - lo + 1;
        if (partitionSize <= 16) {
            if (partitionSize == 1) return;
            if (partitionSize == 2) { swap_if_greater(a, lo, hi); return; }
            if (partitionSize == 3) { swap_if_greater(a, lo, hi - 1); swap_if_greater(a, lo, hi); swap_if_greater(a, hi - 1, hi); return; }
            insertion_sort(a, lo, hi);
            return;
        }
        if (depthLimit == 0) { heap_sort(a, lo, hi); return; }
        depthLimit--;
        int p = pick_pivot_and_partition(a, lo, hi);
        intro_sort(a, p + 1, hi, depthLimit);
        hi = p - 1;
    }
}
int floor_log2(int n) { int r = 0; while (n >= 1) { r++; n /= 2; } return r; }
}  // namespace
void dotnet_sort_keys_items(double* keys, int* items, int index, int length, int arrayLength) {
    if (length < 2) return;
    KI a{keys, items};
    intro_sort(a, index, length + index - 1, 2 * floor_log2(arrayLength));
}

// ------------------------------------------------------------------ R nmath pieces used by GetBoundary (R.cs:8-160,528-548)
static double bd0(double x, double np) {
    if (std::fabs(x - np) < 0.1 * (x + np)) {
        double v = (x - np) / (x + np);
        double s = (x - np) * v;
        double ej = 2 * x * v;
        v = v * v;
        for (int j = 1;; j++) {
            ej *= v;
            double s1 = s + ej / ((j << 1) + 1);
            if (s1 == s) return s1;
            s = s1;
        }
    }
    return x * std::log(x / np) + np - x;
}
static double stirlerr(double n) {
    static const double S0 = 0.083333333333333333333, S1 = 0.00277777777777777777778, S2 = 0.00079365079365079365079365,
                        S3 = 0.000595238095238095238095238, S4 = 0.0008417508417508417508417508;
    static const double sferr_halves[31] = {
        0.0, 0.1534264097200273452913848, 0.0810614667953272582196702, 0.0548141210519176538961390, 0.0413406959554092940938221,
        0.03316287351993628748511048, 0.02767792568499833914878929, 0.02374616365629749597132920, 0.02079067210376509311152277,
        0.01848845053267318523077934, 0.01664469118982119216319487, 0.01513497322191737887351255, 0.01387612882307074799874573,
        0.01281046524292022692424986, 0.01189670994589177009505572, 0.01110455975820691732662991, 0.010411265261972096497478567,
        0.009799416126158803298389475, 0.009255462182712732917728637, 0.008768700134139385462952823, 0.008330563433362871256469318,
        0.007934114564314020547248100, 0.007573675487951840794972024, 0.007244554301320383179543912, 0.006942840107209529865664152,
        0.006665247032707682442354394, 0.006408994188004207068439631, 0.006171712263039457647532867, 0.005951370112758847735624416,
        0.005746216513010115682023589, 0.005554733551962801371038690};
    const double M_LN_SQRT_2PI_ = 0.918938533204672741780329736406;
    if (n <= 15.0) {
        double nn = n + n;
        if (nn == (int)nn) return sferr_halves[(int)nn];
        return std::lgamma(n + 1.0) - (n + 0.5) * std::log(n) + n - M_LN_SQRT_2PI_;
    }
    double nn = n * n;
    if (n > 500) return (S0 - S1 / nn) / n;
    if (n > 80) return (S0 - (S1 - S2 / nn) / nn) / n;
    if (n > 35) return (S0 - (S1 - (S2 - S3 / nn) / nn) / nn) / n;
    return (S0 - (S1 - (S2 - (S3 - S4 / nn) / nn) / nn) / nn) / n;
}
static double dbinom_raw(double x, double n, double p, double q) {  // R.cs:97-125, giveLog = false
    if (p == 0) return (x == 0) ? 1.0 : 0.0;
    if (q == 0) return (x == n) ? 1.0 : 0.0;
    if (x == 0) {
        if (n == 0) return 1.0;
        double lc = (p < 0.1) ? (-bd0(n, n * q) - n * p) : (n * std::log(q));
        return std::exp(lc);
    }
    if (x == n) {
        double lc = (q < 0.1) ? -bd0(n, n * p) - n * q : n * std::log(p);
        return std::exp(lc);
    }
    if (x < 0 || x > n) return 0.0;
    double lc = stirlerr(n) - stirlerr(x) - stirlerr(n - x) - bd0(x, n * p) - bd0(n - x, n * q);
    double lf = std::log(2 * M_PI) + std::log(x) + std::log1p(-x / n);
    return std::exp(lc - 0.5 * lf);
}
static double dhyper(double x, double r, double b, double n) {  // R.cs:43-67
    if (n < x || r < x || (n - x) > b) return 0.0;
    if (n == 0) return (x == 0) ? 1.0 : 0.0;
    double p = n / (r + b), q = (r + b - n) / (r + b);
    double p1 = dbinom_raw(x, r, p, q), p2 = dbinom_raw(n - x, b, p, q), p3 = dbinom_raw(n, r + b, p, q);
    return p1 * p2 / p3;
}
static double pdhyper(double x, double NR, double NB, double n) {  // R.cs:69-95
    double sum = 0, term = 1;
    while (x > 0 && term >= 2.2204460492503131E-16 * sum) {
        term *= x * (NB - n + x) / (n + 1 - x) / (NR + 1 - x);
        sum += term;
        x--;
    }
    return 1 + sum;
}
double phyper_lower(double x, double NR, double NB, double n) {  // R.cs:8-41, lowerTail = true, logP = false
    x = std::floor(x + 1e-7);
    NR = std::floor(NR + 0.5); NB = std::floor(NB + 0.5); n = std::floor(n + 0.5);
    bool lowerTail = true;
    if (NR < 0 || NB < 0 || !std::isfinite(NR + NB) || n < 0 || n > NR + NB) return std::numeric_limits<double>::quiet_NaN();
    if (x * (NR + NB) > n * NR) {
        double oldNB = NB; NB = NR; NR = oldNB;
        x = n - x - 1;
        lowerTail = !lowerTail;
    }
    if (x < 0) return lowerTail ? 0.0 : 1.0;
    if (x >= NR || x >= n) return lowerTail ? 1.0 : 0.0;
    double d = dhyper(x, NR, NB, n);
    double pd = pdhyper(x, NR, NB, n);
    double p = d * pd;
    return lowerTail ? p : (0.5 - p + 0.5);   // R_D_Lval
}

// ------------------------------------------------------------------ GetBoundary.cs
static double BinomialLn(int n, int k) {  // MathNet SpecialFunctions.BinomialLn (not in /root/reference; parity unpinned)
    if (k < 0 || n < 0 || k > n) return -std::numeric_limits<double>::infinity();
    return std::lgamma(n + 1.0) - std::lgamma(k + 1.0) - std::lgamma(n - k + 1.0);
}
static void EtaBoundary(uint32_t nPerm, double eta0, uint32_t n1s, std::vector<uint32_t>& sbdry, uint32_t off) {  // GetBoundary.cs:72-89
    double dn = (double)nPerm - (double)n1s;
    uint32_t k = 0;
    for (uint32_t i = 1; i <= nPerm; i++) {
        double tProb = phyper_lower((double)k, (double)n1s, dn, (double)i);
        if (tProb <= eta0) { sbdry[off + k] = i; k += 1; }
    }
}
static double PExceed(uint32_t nPerm, uint32_t n1s, const std::vector<uint32_t>& sbdry, uint32_t off) {  // GetBoundary.cs:100-152
    int n = (int)nPerm, k = (int)n1s;
    int n1 = (int)(nPerm - sbdry[off]);
    double dlcnk = BinomialLn(n, k);
    double pExcd = std::exp(BinomialLn(n1, k) - dlcnk);
    if (n1s >= 2) {
        n1 = (int)sbdry[off];
        n = (int)(nPerm - sbdry[off + 1]);
        k = (int)(n1s - 1);
        pExcd += std::exp(std::log((double)n1) + BinomialLn(n, k) - dlcnk);
    }
    if (n1s >= 3) {
        n1 = (int)sbdry[off];
        int n2 = (int)sbdry[off + 1];
        n = (int)(nPerm - sbdry[off + 2]);
        k = (int)(n1s - 2);
        pExcd += std::exp(std::log((double)n1) + std::log(n1 - 1.0) - std::log(2.0) + BinomialLn(n, k) - dlcnk) +
                 std::exp(std::log((double)n1) + std::log((double)(n2 - n1)) + BinomialLn(n, k) - dlcnk);
    }
    if (n1s > 3) {
        for (int i = 4; i <= (int)n1s; i++) {
            n1 = (int)sbdry[off + i - 4];
            int k1 = i - 1, k2 = i - 2, k3 = i - 3;
            int n2 = (int)sbdry[off + i - 3], n3 = (int)sbdry[off + i - 2];
            n = (int)(nPerm - sbdry[off + i - 1]);
            k = (int)(n1s - i + 1);
            pExcd += std::exp(BinomialLn(n1, k1) + BinomialLn(n, k) - dlcnk) +
                     std::exp(BinomialLn(n1, k2) + std::log((double)(n3 - n1)) + BinomialLn(n, k) - dlcnk) +
                     std::exp(BinomialLn(n1, k3) + std::log((double)(n2 - n1)) + std::log((double)(n3 - n2)) + BinomialLn(n, k) - dlcnk) +
                     std::exp(BinomialLn(n1, k3) + std::log((double)(n2 - n1)) - std::log(2.0) + std::log(n2 - n1 - 1.0) + BinomialLn(n, k) - dlcnk);
        }
    }
    return pExcd;
}
// GetBoundary.ComputeBoundary (GetBoundary.cs:19-60,154-157)
void ComputeBoundary(uint32_t nPerm, double alpha, double eta, std::vector<uint32_t>& sbdry) {
    uint32_t maxOnes = (uint32_t)(std::floor(nPerm * alpha) + 1);
    const double tol = 1E-2;
    sbdry.assign((size_t)maxOnes * (maxOnes + 1) / 2, 0);
    uint32_t l = 0;
    sbdry[0] = nPerm - (uint32_t)(nPerm * eta);
    double eta0 = eta;
    for (uint32_t j = 2; j <= maxOnes; j++) {
        double etaHi = eta0 * 1.1;
        EtaBoundary(nPerm, etaHi, j, sbdry, l + 1);
        double pHi = PExceed(nPerm, j, sbdry, l + 1);
        double etaLo = eta0 * 0.25;
        EtaBoundary(nPerm, etaLo, j, sbdry, l + 1);
        double pLo = PExceed(nPerm, j, sbdry, l + 1);
        while ((etaHi - etaLo) / etaLo > tol) {
            eta0 = etaLo + (etaHi - etaLo) * (eta - pLo) / (pHi - pLo);
            EtaBoundary(nPerm, eta0, j, sbdry, l + 1);
            double pExcd = PExceed(nPerm, j, sbdry, l + 1);
            if (pExcd > eta) { etaHi = eta0; pHi = pExcd; }
            else { etaLo = eta0; pLo = pExcd; }
        }
        l += j;
    }
}

// ------------------------------------------------------------------ TailProbability.cs
static double pnorm(double x) { return 0.5 * std::erfc(-x / M_SQRT2); }  // MathNet Normal.CumulativeDistribution (parity unpinned)
static double Nu(double x, double tol);
extern "C" double orc_nu(double x, double tol) { return Nu(x, tol); }      // (exported for the test of the device series: tests/test_cbs_gpu.py)
static double Nu(double x, double tol) {  // TailProbability.cs:45-85
    double lnu1;
    if (x > 0.01) {
        lnu1 = std::log(2.0) - 2 * std::log(x);
        double lnu0 = lnu1;
        int k = 2;
        double dk = 0;
        for (int i = 0; i < k; i++) {
            dk = dk + 1;
            double xk = -x * std::sqrt(dk) / 2.0;
            lnu1 = lnu1 - 2.0 * pnorm(xk) / dk;
        }
        while (std::fabs((lnu1 - lnu0) / lnu1) > tol) {
            lnu0 = lnu1;
            for (int i = 0; i < k; i++) {
                dk = dk + 1;
                double xk = -x * std::sqrt(dk) / 2.0;
                lnu1 = lnu1 - 2.0 * pnorm(xk) / dk;
            }
            k *= 2;
        }
    } else lnu1 = -0.583 * x;
    return std::exp(lnu1);
}
static double IntegralInvT1tSq(double x, double a) {  // TailProbability.cs:93-105
    double y = x + a - 0.5;
    double integral = (8.0 * y) / (1.0 - 4.0 * sq(y)) + 2.0 * std::log((1.0 + 2.0 * y) / (1.0 - 2.0 * y));
    y = x - 0.5;
    integral = integral - (8.0 * y) / (1.0 - 4.0 * sq(y)) - 2.0 * std::log((1.0 + 2.0 * y) / (1.0 - 2.0 * y));
    return integral;
}
double TailP(double b, double delta, int m, int nGrid, double tol) {  // TailProbability.cs:21-43
    double dincr = (0.5 - delta) / nGrid;
    double bsqrtm = b / std::sqrt((double)m);
    double tl = 0.5 - dincr, t = 0.5 - 0.5 * dincr;
    double tailP = 0.0;
    for (int i = 0; i < nGrid; i++) {
        tl = tl + dincr;
        t = t + dincr;
        double x = bsqrtm / std::sqrt(t * (1 - t));
        double nux = Nu(x, tol);
        tailP = tailP + sq(nux) * IntegralInvT1tSq(tl, dincr);
    }
    tailP = 9.973557E-2 * (b * b * b) * std::exp(-sq(b) / 2) * tailP;   // Math.Pow(b,3) := b*b*b
    tailP = 2.0 * tailP;
    return tailP;
}

// ------------------------------------------------------------------ CBSTStatistic.cs
static inline int dn_round(double v) { return to_int32_round(v); }

// shared by TMaxO/TMaxP: block boundaries, sequential prefix sums with per-block min/max (CBSTStatistic.cs:44-110 / :621-690)
struct Blocks {
    int nb;
    std::vector<int> bb, ibmin, ibmax;
    std::vector<double> bpsmin, bpsmax;
    double psmin0, psmax0; int ipsmin0, ipsmax0;
};
static void build_blocks(const double* x, int n, double* sx, Blocks& B) {
    double rn = (double)n;
    B.nb = (n >= 50) ? dn_round(std::sqrt((double)n)) : 1;
    int nb = B.nb;
    B.bb.resize(nb); B.ibmin.resize(nb); B.ibmax.resize(nb); B.bpsmin.resize(nb); B.bpsmax.resize(nb);
    for (int i = 0; i < nb; i++) B.bb[i] = dn_round(rn * ((i + 1.0) / nb));
    int ilo = 1;
    double psum = 0;
    B.psmin0 = 0; B.psmax0 = 0; B.ipsmin0 = n; B.ipsmax0 = n;
    for (int j = 0; j < nb; j++) {
        sx[ilo - 1] = psum + x[ilo - 1];
        double psmin = sx[ilo - 1], psmax = sx[ilo - 1];
        int ipsmin = ilo, ipsmax = ilo;
        for (int i = ilo + 1; i <= B.bb[j]; i++) {
            sx[i - 1] = sx[i - 2] + x[i - 1];
            if (sx[i - 1] < psmin) { psmin = sx[i - 1]; ipsmin = i; }
            if (sx[i - 1] > psmax) { psmax = sx[i - 1]; ipsmax = i; }
        }
        B.ibmin[j] = ipsmin; B.ibmax[j] = ipsmax; B.bpsmin[j] = psmin; B.bpsmax[j] = psmax;
        if (psmin < B.psmin0) { B.psmin0 = psmin; B.ipsmin0 = ipsmin; }
        if (psmax > B.psmax0) { B.psmax0 = psmax; B.ipsmax0 = ipsmax; }
        psum = sx[B.bb[j] - 1];
        ilo = B.bb[j] + 1;
    }
}

// the block-pair search shared by TMaxO (tracks arg) and TMaxP (value only); CBSTStatistic.cs:128-326 / :707-905
static void block_search(const double* sx, int n, int al0, const Blocks& B, double& bssmax, int& tmaxi, int& tmaxj) {
    double rn = (double)n;
    int nb = B.nb, nb2 = nb * (nb + 1) / 2;
    std::vector<double> bssbij(nb2), bssijmax(nb2);
    std::vector<int> bloci(nb2), blocj(nb2), loc(nb2), alen(nb2);
    double rnov2 = rn / 2;
    int l = 0, nal0 = n - al0;
    const std::vector<int>& bb = B.bb;
    for (int i = 1; i <= nb; i++) {
        for (int j = i; j <= nb; j++) {
            int ilo = (i == 1) ? 1 : bb[i - 2] + 1, ihi = bb[i - 1];
            int jlo = (j == 1) ? 1 : bb[j - 2] + 1, jhi = bb[j - 1];
            int alenhi = jhi - ilo;
            if (alenhi > nal0) alenhi = nal0;
            double rjhi = (double)alenhi;
            int alenlo = (i == j) ? 1 : jlo - ihi;
            if (alenlo < al0) alenlo = al0;
            double sij1 = std::fabs(B.bpsmax[j - 1] - B.bpsmin[i - 1]);
            double sij2 = std::fabs(B.bpsmax[i - 1] - B.bpsmin[j - 1]);
            double sijmx0 = std::max(sij1, sij2);
            double rjlo = (double)alenlo;
            double rnjov1 = rn / std::min(rjlo * (rn - rjlo), rjhi * (rn - rjhi));
            double bsslim = rnjov1 * sq(sijmx0);
            if (bssmax <= bsslim) {
                loc[l] = l + 1; bloci[l] = i; blocj[l] = j; bssijmax[l] = bsslim;
                if (sij1 > sij2) {
                    alen[l] = std::abs(B.ibmax[j - 1] - B.ibmin[i - 1]);
                    double rj = (double)alen[l];
                    rnjov1 = rn / (rj * (rn - rj));
                    bssbij[l] = rnjov1 * sq(sij1);
                } else {
                    alen[l] = std::abs(B.ibmin[j - 1] - B.ibmax[i - 1]);
                    double rj = (double)alen[l];
                    rnjov1 = rn / (rj * (rn - rj));
                    bssbij[l] = rnjov1 * sq(sij2);
                }
                l++;
            }
        }
    }
    int nb1 = l;
    for (int k = 0; k < nb1; k++) loc[k] = k + 1;
    dotnet_sort_keys_items(bssbij.data(), loc.data(), 0, nb1, nb2);
    for (l = nb1 - 1; l >= 0; l--) {
        int k = loc[l] - 1;
        double bsslim = bssijmax[k];
        if (bssmax <= bsslim) {
            int bi = bloci[k], bj = blocj[k];
            int alenmax = alen[k];
            int ilo = (bi == 1) ? 1 : bb[bi - 2] + 1, ihi = bb[bi - 1];
            int jlo = (bj == 1) ? 1 : bb[bj - 2] + 1, jhi = bb[bj - 1];
            int alenhi = jhi - ilo;
            if (alenhi > nal0) alenhi = nal0;
            double rjhi = (double)alenhi;
            int alenlo = (bi == bj) ? 1 : (jlo - ihi);
            if (alenlo < al0) alenlo = al0;
            double rjlo = (double)alenlo;
            if (alenmax > n - alenmax) alenmax = n - alenmax;
            if ((rjlo <= rnov2) && (alenlo <= alenmax)) {
                for (int i2j = alenlo; i2j <= alenmax; i2j++) {
                    int ixlo = std::max(0, jlo - ilo - i2j), ixhi = std::max(0, ihi + i2j - jhi);
                    double sxmx = 0;
                    int sxmxi = ilo + ixlo - 1;
                    for (int i = ilo + ixlo; i <= ihi - ixhi; i++) {
                        int j = i + i2j;
                        double absx = std::fabs(sx[j - 1] - sx[i - 1]);
                        if (sxmx < absx) { sxmx = absx; sxmxi = i; }
                    }
                    double rj = (double)i2j;
                    double rnjov1 = rn / (rj * (rn - rj));
                    double bijbss = rnjov1 * sq(sxmx);
                    if (bijbss > bssmax) { bssmax = bijbss; tmaxi = sxmxi; tmaxj = sxmxi + i2j; }
                }
            }
            alenmax = n - alenmax;
            if ((rjhi >= rnov2) && (alenhi >= alenmax)) {
                for (int i2j = alenhi; i2j >= alenmax; i2j--) {
                    int ixlo = std::max(0, jlo - ilo - i2j), ixhi = std::max(0, ihi + i2j - jhi);
                    double sxmx = 0;
                    int sxmxi = ilo + ixlo - 1;
                    for (int i = ilo + ixlo; i <= ihi - ixhi; i++) {
                        int j = i + i2j;
                        double absx = std::fabs(sx[j - 1] - sx[i - 1]);
                        if (sxmx < absx) { sxmx = absx; sxmxi = i; }
                    }
                    double rj = (double)i2j;
                    double rnjov1 = rn / (rj * (rn - rj));
                    double bijbss = rnjov1 * sq(sxmx);
                    if (bijbss > bssmax) { bssmax = bijbss; tmaxi = sxmxi; tmaxj = sxmxi + i2j; }
                }
            }
        }
    }
}

// CBSTStatistic.TMaxO (CBSTStatistic.cs:19-341), isBinary = false
void TMaxO(const double* x, int n, double tss, double* sx, int iseg[2], double& ostat, int al0) {
    Blocks B;
    build_blocks(x, n, sx, B);
    double rn = (double)n;
    double psdiff = B.psmax0 - B.psmin0;
    double rj = (double)std::abs(B.ipsmax0 - B.ipsmin0);
    double rnjov1 = rn / (rj * (rn - rj));
    double bssmax = rnjov1 * sq(psdiff);
    int tmaxi = std::min(B.ipsmax0, B.ipsmin0), tmaxj = std::max(B.ipsmax0, B.ipsmin0);
    if (psdiff <= 0) bssmax = 0;
    else block_search(sx, n, al0, B, bssmax, tmaxi, tmaxj);
    if (tss <= bssmax + 0.0001) tss = bssmax + 1.0;
    bssmax = bssmax / ((tss - bssmax) / (rn - 2.0));
    ostat = bssmax;
    iseg[0] = tmaxi; iseg[1] = tmaxj;
}

// CBSTStatistic.TMaxP (CBSTStatistic.cs:599-934), isBinary = false
double TMaxP(double tss, const double* px, int n, double* sx, int al0) {
    Blocks B;
    build_blocks(px, n, sx, B);
    double rn = (double)n;
    double psdiff = B.psmax0 - B.psmin0;
    double rj = (double)std::abs(B.ipsmax0 - B.ipsmin0);
    double rnjov1 = rn / (rj * (rn - rj));
    double bssmax = rnjov1 * sq(psdiff);
    int ti = 0, tj = 0;
    block_search(sx, n, al0, B, bssmax, ti, tj);
    if (tss <= bssmax + 0.0001) tss = bssmax + 1.0;
    return bssmax / ((tss - bssmax) / (rn - 2.0));
}

// CBSTStatistic.HTMaxP (CBSTStatistic.cs:354-586), isBinary = false
double HTMaxP(int k, double tss, const double* px, int n, double* sx, int al0) {
    double rn = (double)n;
    int nb = (int)(rn / k);
    std::vector<double> bpsmax(nb), bpsmin(nb);
    std::vector<int> bb(nb);
    for (int i = 0; i < nb; i++) bb[i] = dn_round(rn * ((double)(i + 1) / nb));
    int ilo = 1;
    double psum = 0, h = 0.0;
    for (int j = 0; j < nb; j++) {
        sx[ilo - 1] = psum + px[ilo - 1];
        double psmin = sx[ilo - 1], psmax = sx[ilo - 1];
        int ipsmin = ilo, ipsmax = ilo;
        for (int i = ilo; i < bb[j]; i++) {
            sx[i] = sx[i - 1] + px[i];
            if (sx[i] < psmin) { psmin = sx[i]; ipsmin = i + 1; }
            if (sx[i] > psmax) { psmax = sx[i]; ipsmax = i + 1; }
        }
        bpsmin[j] = psmin; bpsmax[j] = psmax;
        psum = sx[bb[j] - 1];
        ilo = bb[j] + 1;
        int i = std::abs(ipsmin - ipsmax);
        if ((i <= k) && (i >= al0)) {
            double rj = (double)i;
            double rnjov1 = rn / (rj * (rn - rj));
            double bssmx = rnjov1 * sq(bpsmax[j] - bpsmin[j]);
            if (h < bssmx) h = bssmx;
        }
    }
    auto arcs = [&](double psdiffsq, auto&& inner) {
        for (int j = al0; j <= k; j++) {
            double rj = (double)j;
            double rnjov1 = rn / (rj * (rn - rj));
            double bsslim = rnjov1 * psdiffsq;
            if (bsslim < h) break;
            double sxmx = inner(j);
            double bssmx = rnjov1 * sq(sxmx);
            if (h < bssmx) h = bssmx;
        }
    };
    // first block
    {
        int lo = 1, hi = bb[0];
        arcs(sq(bpsmax[0] - bpsmin[0]), [&](int j) {
            double sxmx = 0.0;
            for (int i = lo; i <= hi - j; i++) { double a = std::fabs(sx[i + j - 1] - sx[i - 1]); if (sxmx < a) sxmx = a; }
            return sxmx;
        });
    }
    // minor arcs spanning the end
    {
        double psdiff = std::max(std::fabs(bpsmax[0] - bpsmin[nb - 1]), std::fabs(bpsmax[nb - 1] - bpsmin[0]));
        arcs(sq(psdiff), [&](int j) {
            double sxmx = 0.0;
            int nmj = n - j;
            for (int i = 0; i < j; i++) { double a = std::fabs(sx[i + nmj] - sx[i]); if (sxmx < a) sxmx = a; }
            return sxmx;
        });
    }
    for (int l = 1; l < nb; l++) {
        int lo = bb[l - 1] + 1, hi = bb[l];
        arcs(sq(bpsmax[l] - bpsmin[l]), [&](int j) {
            double sxmx = 0.0;
            for (int i = lo; i <= hi - j; i++) { double a = std::fabs(sx[i + j - 1] - sx[i - 1]); if (sxmx < a) sxmx = a; }
            return sxmx;
        });
        double psdiff = std::max(std::fabs(bpsmax[l] - bpsmin[l - 1]), std::fabs(bpsmax[l - 1] - bpsmin[l]));
        arcs(sq(psdiff), [&](int j) {
            double sxmx = 0.0;
            for (int i = lo - j; i <= lo - 1; i++) { double a = std::fabs(sx[i + j - 1] - sx[i - 1]); if (sxmx < a) sxmx = a; }
            return sxmx;
        });
    }
    if (tss <= h + 0.0001) tss = h + 1.0;
    return h / ((tss - h) / (rn - 2.0));
}

// CBSTStatistic.TPermP (CBSTStatistic.cs:947-1024)
static double TPermP(int n1, int n2, int n, const double* genomeData, int off, double* px, uint32_t nPerm, MT19937& rnd, CbsStats* st) {
    double rn1 = (double)n1, rn2 = (double)n2, rn = rn1 + rn2;
    int nrej;
    if (n1 == 1 || n2 == 1) nrej = (int)nPerm;
    else {
        double xsum1 = 0.0, tss = 0.0;
        for (int i = 0; i < n1; i++) { px[i] = genomeData[off + i]; xsum1 = xsum1 + genomeData[off + i]; tss = tss + sq(genomeData[off + i]); }
        double xsum2 = 0.0;
        for (int i = n1; i < n; i++) { px[i] = genomeData[off + i]; xsum2 = xsum2 + genomeData[off + i]; tss = tss + sq(genomeData[off + i]); }
        double xbar = (xsum1 + xsum2) / rn;
        tss = tss - rn * sq(xbar);
        int m1; double rm1, ostat, tstat;
        if (n1 <= n2) { m1 = n1; rm1 = rn1; ostat = 0.99999 * std::fabs(xsum1 / rn1 - xbar); tstat = sq(ostat) * rn1 * rn / rn2; }
        else { m1 = n2; rm1 = rn2; ostat = 0.99999 * std::fabs(xsum2 / rn2 - xbar); tstat = sq(ostat) * rn2 * rn / rn1; }
        nrej = 0;
        tstat = tstat / ((tss - tstat) / (rn - 2.0));
        if ((tstat > 25) && (m1 >= 10)) {}
        else {
            for (uint32_t np = 0; np < nPerm; np++) {
                xsum1 = 0;
                for (int i = n - 1; i >= n - m1; i--) {
                    double cc = rnd.next_double();
                    int j = (int)(cc * (i + 1));
                    j = (j > i) ? i : j;
                    std::swap(px[i], px[j]);
                    xsum1 = xsum1 + px[i];
                }
                double pstat = std::fabs(xsum1 / rm1 - xbar);
                if (ostat <= pstat) nrej = nrej + 1;
            }
            if (st) st->tpermp_draws += (int64_t)nPerm * m1;
        }
    }
    return (double)nrej / nPerm;
}

// ------------------------------------------------------------------ ChangePoint.cs
static void XPerm(const double* x, double* px, int n, MT19937& rnd) {  // ChangePoint.cs:407-421
    for (int i = 0; i < n; i++) px[i] = x[i];
    for (int i = n - 1; i >= 0; i--) {
        double cc = rnd.next_double();
        int j = (int)(cc * (i + 1));
        j = (j > i) ? i : j;
        std::swap(px[i], px[j]);
    }
}

// ChangePoint.FindChangePoints (ChangePoint.cs:291-400), isBinary = false
static void FindChangePoints(const double* gd, int n, double tss, uint32_t nPerm, double cutoffPValue, int& nChangePoints,
                             int iChangePoint[2], bool hybrid, int al0, int hk, double delta, int nGrid,
                             const std::vector<uint32_t>& sbdry, double tol, MT19937& rnd, CbsStats* st) {
    std::vector<double> px(n), sx(n);
    int iseg[2];
    double ostat;
    int nrej = 0;
    nChangePoints = 0;
    TMaxO(gd, n, tss, sx.data(), iseg, ostat, al0);
    if (st) { st->tmaxo_calls++; st->tmaxo_elems += n; }
    double ostat1 = std::sqrt(ostat);
    ostat *= 0.99999;
    if (ostat1 <= 0.1) return;
    int l = std::min(iseg[1] - iseg[0], n - iseg[1] + iseg[0]);
    if (!((ostat1 >= 7.0) && (l >= 10))) {
        if (hybrid) {
            double pValue1 = TailP(ostat1, delta, n, nGrid, tol);
            if (pValue1 > cutoffPValue) { if (st) st->tailp_exits++; return; }
            double pValue2 = cutoffPValue - pValue1;
            int nrejc = (int)(pValue2 * nPerm);
            int k = nrejc * (nrejc + 1) / 2 + 1;
            for (uint32_t np = 1; np <= nPerm; np++) {
                XPerm(gd, px.data(), n, rnd);
                double pstat = HTMaxP(hk, tss, px.data(), n, sx.data(), al0);
                if (st) { st->perms++; st->perm_elems += n; }
                if (ostat <= pstat) { nrej++; k++; }
                if (nrej > nrejc) return;
                if (np >= sbdry[k - 1]) break;
            }
        } else {
            int nrejc = (int)(cutoffPValue * nPerm);
            int k = nrejc * (nrejc + 1) / 2 + 1;
            for (uint32_t np = 1; np <= nPerm; np++) {
                XPerm(gd, px.data(), n, rnd);
                double pstat = TMaxP(tss, px.data(), n, sx.data(), al0);
                if (st) { st->perms++; st->perm_elems += n; }
                if (ostat <= pstat) { nrej++; k++; }
                if (nrej > nrejc) return;
                if (np >= sbdry[k - 1]) break;
            }
        }
    } else if (st) st->big_t_splits++;
    if (iseg[1] == n) { nChangePoints = 1; iChangePoint[0] = iseg[0]; }
    else if (iseg[0] == 0) { nChangePoints = 1; iChangePoint[0] = iseg[1]; }
    else {
        int off = 0, n1 = iseg[0], n12 = iseg[1], n2 = n12 - n1;
        double tPValue = TPermP(n1, n2, n12, gd, off, px.data(), nPerm, rnd, st);
        if (tPValue <= cutoffPValue) { nChangePoints = 1; iChangePoint[0] = iseg[0]; }
        off = iseg[0];
        n12 = n - iseg[0];
        n2 = n - iseg[1];
        n1 = n12 - n2;
        tPValue = TPermP(n1, n2, n12, gd, off, px.data(), nPerm, rnd, st);
        if (tPValue <= cutoffPValue) { nChangePoints++; iChangePoint[nChangePoints - 1] = iseg[1]; }
    }
}

// Helper.Median (Helper.cs:30-44) via QuickSelect (:53-83): value semantics == order statistics
static double HelperMedian(const double* x, int iStart, int iEnd) {
    std::vector<double> y(x + iStart, x + iEnd);
    int mid = (int)y.size() / 2;
    std::nth_element(y.begin(), y.begin() + mid, y.end());
    double median = y[mid];
    if (y.size() % 2 == 0) {
        double lower = *std::max_element(y.begin(), y.begin() + mid);
        median = (median + lower) / 2;
    }
    return median;
}

// ChangePoint.ChangePointsSDUndo (ChangePoint.cs:155-196)
static std::vector<int> ChangePointsSDUndo(const double* gd, const std::vector<int>& lengthSeg, double trimmedSD, double changeSD) {
    changeSD *= trimmedSD;
    std::vector<int> cpl(lengthSeg.size());
    std::partial_sum(lengthSeg.begin(), lengthSeg.end(), cpl.begin());
    bool sdUndo = true;
    while (sdUndo) {
        int k = (int)cpl.size();
        if (k > 1) {
            std::vector<int> starts(cpl.begin(), cpl.end() - 1);
            starts.insert(starts.begin(), 0);
            std::vector<double> med(k);
            for (int i = 0; i < k; i++) med[i] = HelperMedian(gd, starts[i], cpl[i]);
            double mn = std::fabs(med[1] - med[0]);
            int iMin = 0;
            for (int i = 1; i < k - 1; i++) { double d = std::fabs(med[i + 1] - med[i]); if (d < mn) { mn = d; iMin = i; } }
            if (mn < changeSD) cpl.erase(cpl.begin() + iMin);
            else sdUndo = false;
        } else sdUndo = false;
    }
    cpl.insert(cpl.begin(), 0);
    std::vector<int> out(cpl.size() - 1);
    for (size_t i = 0; i + 1 < cpl.size(); i++) out[i] = cpl[i + 1] - cpl[i];
    return out;
}

// Prune.ErrorSumOfSquares (Prune.cs:18-51): Fortran errssq
static double ErrorSumOfSquares(const std::vector<int>& lengthSeg, const std::vector<double>& segmentSums, int k, const std::vector<int>& locations) {
    double errorSumOfSquares = 0.0;
    double segsx = 0.0; int segnx = 0;
    for (int i = 0; i < locations[0]; i++) { segsx += segmentSums[i]; segnx += lengthSeg[i]; }
    errorSumOfSquares += std::pow(segsx, 2) / segnx;
    for (int j = 1; j < k; j++) {
        segsx = 0.0; segnx = 0;
        for (int i = locations[j - 1]; i < locations[j]; i++) { segsx += segmentSums[i]; segnx += lengthSeg[i]; }
        errorSumOfSquares += std::pow(segsx, 2) / segnx;
    }
    segsx = 0.0; segnx = 0;
    for (int i = locations[k - 1]; i < (int)lengthSeg.size(); i++) { segsx += segmentSums[i]; segnx += lengthSeg[i]; }
    errorSumOfSquares += std::pow(segsx, 2) / segnx;
    return errorSumOfSquares;
}
// Prune.Combination (Prune.cs:62-72): next r-combination of 1..(r+nmr), AS 88
static void Combination(int r, int nmr, std::vector<int>& locations, bool& rleft) {
    int i = r - 1;
    while (locations[i] == nmr + i + 1) i--;
    locations[i]++;
    for (int j = i + 1; j < r; j++) locations[j] = locations[j - 1] + 1;
    if (locations[0] == nmr + 1) rleft = false;
}
// ChangePoint.ChangePointsPrune (ChangePoint.cs:205-271).  Quirks kept: with a single change point the j-loop never runs and the
// change point is dropped; when no j exceeds the cut-off the result is ONE segment.
std::vector<int> ChangePointsPrune(const double* gd, int n, const std::vector<int>& lengthSeg, double changeCutoff) {
    const int nseg = (int)lengthSeg.size(), ncp = nseg - 1;
    std::vector<double> sx(nseg);
    std::vector<int> loc(ncp), loc1a(ncp), loc1b(ncp);     // loc1[0,*], loc1[1,*]
    int prunedNChangePoints = 0;
    double ssq = 0.0;
    for (int i = 0; i < n; i++) ssq += std::pow(gd[i], 2);          // Helper.PartialSumOfPowers(x, 2, 0, n)
    int k = 0;
    for (int i = 0; i < nseg; i++) { double sp = 0.0; for (int t = k; t < k + lengthSeg[i]; t++) sp += std::pow(gd[t], 1); sx[i] = sp; k += lengthSeg[i]; }
    for (int i = 0; i < ncp; i++) { loc[i] = i + 1; loc1b[i] = i + 1; }
    const double wssqk = ssq - ErrorSumOfSquares(lengthSeg, sx, ncp, loc);
    for (int j = ncp - 1; j > 0; j--) {
        const int kmj = ncp - j;
        bool jleft = true;
        for (int i = 0; i < j; i++) { loc[i] = i + 1; loc1a[i] = i + 1; }
        double wssqj = ssq - ErrorSumOfSquares(lengthSeg, sx, j, loc);
        while (jleft) {
            Combination(j, kmj, loc, jleft);
            double wssq1 = ssq - ErrorSumOfSquares(lengthSeg, sx, j, loc);
            if (wssq1 <= wssqj) { wssqj = wssq1; for (int i = 0; i < j; i++) loc1a[i] = loc[i]; }
        }
        if (wssqj / wssqk > 1 + changeCutoff) {
            prunedNChangePoints = j + 1;
            for (int i = 0; i < prunedNChangePoints; i++) loc[i] = loc1b[i];
            break;
        } else {
            for (int i = 0; i < j; i++) loc1b[i] = loc1a[i];
        }
    }
    std::vector<int> cum(nseg);
    std::partial_sum(lengthSeg.begin(), lengthSeg.end(), cum.begin());
    std::vector<int> pcp(prunedNChangePoints + 2);
    for (int i = 0; i < prunedNChangePoints; i++) pcp[i + 1] = cum[loc[i] - 1];
    pcp[0] = 0; pcp[pcp.size() - 1] = n;
    std::vector<int> out(pcp.size() - 1);
    for (size_t i = 0; i + 1 < pcp.size(); i++) out[i] = pcp[i + 1] - pcp[i];
    return out;
}

// ChangePoint.ChangePoints (ChangePoint.cs:44-153). undoSplits: 0 None, 1 Prune, 2 SDUndo.
std::vector<int> ChangePoints(const double* genomeData, int n, const std::vector<uint32_t>& sbdry, MT19937& rnd, double alpha,
                              uint32_t nPerm, int minWidth, int kMax, uint32_t nMin, int undoSplits, double trimmedSD,
                              double undoPrune, double undoSD, CbsStats* stats) {
    const int nGrid = 100; const double tol = 1E-6;
    std::vector<int> segEnd = {0, n};
    int k = (int)segEnd.size();
    std::vector<int> changeLocations;
    int nChangePoints = 0;
    int iChangePoint[2] = {0, 0};
    while (k > 1) {
        int currentN = segEnd[k - 1] - segEnd[k - 2];
        if (currentN >= 2 * minWidth) {
            std::vector<double> cur(genomeData + segEnd[k - 2], genomeData + segEnd[k - 2] + currentN);
            bool hybrid = false;
            double delta = 0.0;
            if (nMin < (uint32_t)currentN) { hybrid = true; delta = (kMax + 1.0) / currentN; }
            double mx = *std::max_element(cur.begin(), cur.end()), mn = *std::min_element(cur.begin(), cur.end());
            if (mx == mn) nChangePoints = 0;
            else {
                double sum = 0;
                for (double v : cur) sum += v;
                double currentAverage = sum / currentN;   // Enumerable.Average
                for (double& v : cur) v -= currentAverage;
                double currentTSS = 0.0;
                for (double v : cur) currentTSS += 1.0 * v * v;    // Helper.WeightedSumOfSquares: wi * x * x
                FindChangePoints(cur.data(), currentN, currentTSS, nPerm, alpha, nChangePoints, iChangePoint, hybrid, minWidth, kMax,
                                 delta, nGrid, sbdry, tol, rnd, stats);
            }
        } else nChangePoints = 0;
        if (nChangePoints == 0) changeLocations.push_back(segEnd[k - 1]);
        for (int i = 0; i < nChangePoints; i++) iChangePoint[i] += segEnd[k - 2];
        switch (nChangePoints) {
            case 0: segEnd.erase(segEnd.begin() + (k - 1)); break;
            case 1: segEnd.insert(segEnd.begin() + (k - 1), iChangePoint[0]); break;
            case 2: segEnd.insert(segEnd.begin() + (k - 1), iChangePoint, iChangePoint + 2); break;
        }
        k = (int)segEnd.size();
    }
    std::reverse(changeLocations.begin(), changeLocations.end());
    std::vector<int> segEnds = changeLocations;
    int nSeg = (int)segEnds.size();
    segEnds.insert(segEnds.begin(), 0);
    std::vector<int> lengthSeg(nSeg);
    for (int i = 0; i < nSeg; i++) lengthSeg[i] = segEnds[i + 1] - segEnds[i];
    if (nSeg > 1 && undoSplits == 1) lengthSeg = ChangePointsPrune(genomeData, n, lengthSeg, undoPrune);
    if (nSeg > 1 && undoSplits == 2) lengthSeg = ChangePointsSDUndo(genomeData, lengthSeg, trimmedSD, undoSD);
    return lengthSeg;
}

// ChangePoint.TrimmedVariance / InflationFactor (ChangePoint.cs:423-474). Normal.InverseCDF / Density are MathNet
// (parity unpinned); only used when undo = SDUndo.
static double qnorm_upper(double p) {  // inverse CDF by bisection on erfc (monotone, 1e-16 resolution)
    double lo = -40, hi = 40;
    for (int it = 0; it < 200; it++) { double mid = 0.5 * (lo + hi); if (pnorm(mid) < p) lo = mid; else hi = mid; }
    return 0.5 * (lo + hi);
}
static double InflationFactor(double trim) {
    double a = qnorm_upper(1 - trim);
    double step = 2 * a / 10000;
    double from = -a + step / 2, to = a - step / 2;
    double st = (to - from) / (10000 - 1);
    double eX2 = 0.0, x1 = from;
    for (int i = 0; i < 10000; i++) {
        double xv = (i == 0) ? from : (i == 9999 ? to : (x1 = x1 + st));
        eX2 += (xv * xv) * (std::exp(-0.5 * xv * xv) / std::sqrt(2 * M_PI));
    }
    eX2 = eX2 * step / (1 - 2 * trim);
    return 1 / eX2;
}
double TrimmedVariance(const std::vector<const double*>& scores, const std::vector<int>& lens, double trim) {
    int n = 0;
    for (int l : lens) n += l;
    std::vector<double> diff(n > 0 ? n - 1 : 0);
    int i = 0;
    double last = std::numeric_limits<double>::quiet_NaN();
    for (size_t c = 0; c < lens.size(); c++) {
        if (lens[c] <= 0) continue;
        if (i > 0) { diff[i] = scores[c][0] - last; i++; }
        for (int t = 0; t + 1 < lens[c]; t++) diff[i + t] = scores[c][t + 1] - scores[c][t];
        i += lens[c] - 1;
        last = scores[c][lens[c] - 1];
    }
    int nKeep = to_int32_round(round_half_even((1 - 2 * trim) * (n - 1)));
    for (double& d : diff) d = std::fabs(d);
    std::sort(diff.begin(), diff.end());
    double sp = 0.0;
    for (int t = 0; t < nKeep; t++) sp += sq(diff[t]);
    return InflationFactor(trim) * sp / (2 * nKeep);
}

}  // namespace oracle
