"""Second, independent restatements of the stages the reference's own tests do not pin (BinCountsForChromosome, the bin
size, PerSampleHMM / Viterbi), written directly from the C# as plain Python loops and compared with the C++ oracle on seeded inputs.
They share no code with oracle/*.cpp: an error in reading the C# would have to be made twice, in two languages, to go unnoticed.
CPU only; small sizes (pure-Python loops)."""
import math
import sys

import numpy as np
import pytest

import oracle_lib as O


# ---------------------------------------------------------------------------------------------------------------------------------------
# CanvasBin.BinCountsForChromosome, CanvasBin.cs:568-661, without predefined bins; modes Binary (0) and TruncatedDynamicRange (3)
def py_bin_counts_for_chromosome(bases, possible, observed, bin_size, mode):
    bins = []
    pos = 0
    while bases[pos] == ord("n"):                                  # :581-582 (an all-'n' sequence throws in the reference)
        pos += 1
    nucleotides = gc = poss = obs = 0
    start = -1
    seen = []
    while pos < len(bases):
        if start == -1:
            start = pos
        nucleotides += 1                                            # :592 compares a char with the string "n": never equal, every base counts
        if chr(bases[pos]) in "CcGg":
            gc += 1
        if possible[pos]:
            poss += 1
            obs += int(observed[pos])
            seen.append(int(observed[pos]))
        if poss == bin_size:
            if mode == 3:
                obs = sum(min(10, v) for v in seen)
            gc_pct = int(np.float32(100.0) * np.float32(gc) / np.float32(nucleotides))     # (int)(100f * GCCount / NucleotideCount)
            bins.append((start, pos + 1, gc_pct, obs))
            nucleotides = gc = poss = obs = 0
            start = -1
            seen = []
        pos += 1
    return bins


def _random_chromosome(rng, L):
    bases = rng.choice(np.frombuffer(b"ACGTacgtnN", np.uint8), size=L, p=[.2, .2, .2, .2, .04, .04, .04, .04, .03, .01]).astype(np.uint8)
    lead = int(rng.randint(0, 40)) if rng.rand() < 0.5 else 0
    bases[:lead] = ord("n")
    if (bases == ord("n")).all():                                 # the reference throws on an all-'n' sequence
        bases[-1] = ord("A")
    possible = (rng.rand(L) < rng.uniform(0.2, 0.95))
    observed = np.where(possible, rng.poisson(rng.uniform(0.2, 6.0), L), 0).clip(0, 255).astype(np.uint8)
    if rng.rand() < 0.3:
        observed[rng.randint(0, L, size=3)] = 255
        observed[~possible] = 0
    return bases, possible, observed


def _pack_mask(possible):
    L = len(possible)
    padded = np.zeros((L + 63) // 64 * 64, np.uint8)
    padded[:L] = possible
    return np.packbits(padded, bitorder="little").view(np.uint64)


@pytest.mark.parametrize("mode", [0, 3])
def test_bin_counts_for_chromosome_two_restatements(mode):
    rng = np.random.RandomState(1234 + mode)
    for it in range(150):
        L = int(rng.randint(1, 700))
        bases, possible, observed = _random_chromosome(rng, L)
        bin_size = int(rng.randint(1, 60))
        want = py_bin_counts_for_chromosome(bases, possible, observed, bin_size, mode)
        got = O.bin_chromosome(bases, _pack_mask(possible), observed, bin_size, mode)
        got = list(zip(*[g.tolist() for g in got]))
        assert got == want, (it, L, bin_size)


# SampleHitArrays.GetRates / GetBinSize (CanvasBin.cs:30-83), HitArray.CountSetBits (HitArray.cs:24-32), Utilities.Median (Utilities.cs:340-344)
def py_bin_size(possible_by_chr, observed_by_chr, counts_per_bin):
    rates = []
    for p, o in zip(possible_by_chr, observed_by_chr):
        rates.append(sum(1 for v in o if v > 0) / float(sum(1 for b in p if b)))
    s = sorted(rates)
    n = len(s)
    median = s[n // 2] if n % 2 else (s[n // 2 - 1] + s[n // 2]) / 2
    return int(counts_per_bin / median)


def test_bin_size_two_restatements():
    rng = np.random.RandomState(99)
    for it in range(60):
        nchr = int(rng.randint(1, 7))
        chroms = [_random_chromosome(rng, int(rng.randint(50, 900))) for _ in range(nchr)]
        cpb = int(rng.randint(1, 300))
        want = py_bin_size([c[1] for c in chroms], [c[2] for c in chroms], cpb)
        rates = [O.bin_rate(c[2], _pack_mask(c[1])) for c in chroms]
        assert O.bin_size(rates, cpb) == want, it


# ---------------------------------------------------------------------------------------------------------------------------------------
# PerSampleHMM for one sample: HiddenMarkovModelsRunner.cs:23-162, HMM.cs:24-130, Distributions.cs:22-78,255-316,
# DistributionUtilities.cs:51-69, Utilities.Quartiles (Utilities.cs:361-420)
def _to_int32(v):                                                   # Convert.ToInt32(double): round half to even
    return int(np.rint(v))


def py_quartiles(values):
    s = np.sort(np.asarray(values, np.float32))
    n = len(s)
    mid = n // 2
    f = np.float32
    if n % 2 == 0:
        q2 = (s[mid - 1] + s[mid]) / f(2)
        mm = mid // 2
        if mid % 2 == 0:
            q1 = (s[mm - 1] + s[mm]) / f(2)
            q3 = (s[mid + mm - 1] + s[mid + mm]) / f(2)
        else:
            q1 = s[mm]
            q3 = s[mm + mid]
    else:
        q2 = s[mid]
        q1 = q3 = f(0)
        if (n - 1) % 4 == 0:
            k = (n - 1) // 4
            q1 = s[k - 1] * f(.25) + s[k] * f(.75)
            q3 = s[3 * k] * f(.75) + s[3 * k + 1] * f(.25)
        elif (n - 3) % 4 == 0:
            k = (n - 3) // 4
            q1 = s[k] * f(.75) + s[k + 1] * f(.25)
            q3 = s[3 * k + 1] * f(.25) + s[3 * k + 2] * f(.75)
    return f(q1), f(q2), f(q3)


def py_negative_binomial(mean, variance, max_value):
    def log(v):
        return math.log(v) if v > 0 else -math.inf
    r = max(max(mean, 0.1) ** 2 / (max(variance, mean * 1.2) - mean), 2.0)
    out = []
    for x in range(max_value):
        try:
            head = (1 + mean / r) ** (-r)
        except OverflowError:
            head = math.inf
        t = log(head) + log((mean / (mean + r)) ** x) + math.lgamma(r + x) - math.lgamma(x + 1.0) - math.lgamma(r)
        try:
            d = math.exp(t)
        except OverflowError:
            d = math.inf
        out.append(0.0 if (math.isnan(d) or math.isinf(d)) else d)
    return out


def py_hmm_per_sample(chromosomes, n_states=5, min_size=10):
    all_values = np.concatenate([np.asarray(c, np.float64).astype(np.float32) for c in chromosomes])
    q1, q2, q3 = py_quartiles(all_values)
    iqr = np.float32(q3 - q1)
    median, pseudo_variance = float(q2), float(np.float32(iqr * iqr))
    log = lambda v: math.log(v) if v > 0 else -math.inf
    trans = [[0.99 if i == j else (1.0 - 0.99) / (n_states - 1) for j in range(n_states)] for i in range(n_states)]
    prior = float(np.float32(1.0) / np.float32(n_states))
    paths = []
    for cov in chromosomes:
        T = len(cov)
        if T <= min_size:
            paths.append(None)
            continue
        haploid = median / 2.0
        cap = haploid * n_states
        data = [cap if v > cap else v for v in cov]
        max_value = max(_to_int32(v) for v in data)
        tables = [py_negative_binomial(max(cn, 0.1) * haploid, pseudo_variance, max_value + 10) for cn in range(n_states)]

        def viterbi_likelihood(x, j, row):                          # one sample: the only genotype list is [j]
            return log(tables[j][_to_int32(x)]) + log(trans[row][j])
        score = [log(prior) + viterbi_likelihood(data[0], j, 0) - log(trans[0][j]) for j in range(n_states)]
        back = []
        for t in range(1, T):
            new, frm = [], []
            for j in range(n_states):
                state, best = 0, -sys.float_info.max
                for i in range(n_states):
                    cand = score[i] + viterbi_likelihood(data[t], j, i)
                    if cand > best:
                        state, best = i, cand
                new.append(best)
                frm.append(state)
            score = new
            back.append(frm)
        state, best = -1, -sys.float_info.max
        for i in range(n_states):
            if score[i] > best:
                state, best = i, score[i]
        path = [state]
        for frm in reversed(back):
            state = frm[state]
            path.append(state)
        paths.append(path[::-1])
    return paths, median, pseudo_variance


def _random_coverage(rng, T, depth):
    cn = np.full(T, 2)
    for _ in range(int(rng.randint(0, 4))):
        a = int(rng.randint(0, T)); b = min(T, a + int(rng.randint(1, max(2, T // 3))))
        cn[a:b] = rng.choice([0, 1, 3, 4, 6])
    lam = np.maximum(cn, 0.02) * depth / 2.0
    shape = rng.uniform(3, 40)
    v = rng.poisson(rng.gamma(shape, lam / shape)).astype(np.float64)
    v += rng.randint(0, 100, T) / 100.0                            # two decimals, as parsed from the F2 text of the cleaned file
    if rng.rand() < 0.3:
        v[rng.randint(0, T, 2)] *= 7                                # outliers beyond the cap
    return np.round(v, 2)


def test_per_sample_hmm_two_restatements():
    rng = np.random.RandomState(4321)
    segmented = with_events = 0
    for it in range(25):
        depth = float(rng.choice([8, 30, 70, 150]))
        chroms = [_random_coverage(rng, int(rng.choice([4, 10, 11, 37, 120, 300])), depth) for _ in range(int(rng.randint(1, 5)))]
        if sum(len(c) for c in chroms) < 8:
            continue
        want, median, pv = py_hmm_per_sample(chroms)
        med, var = O.hmm_global_params(chroms)
        assert (med, var) == (median, pv), it
        paths, ran = O.hmm_genome_per_sample(chroms, threads=1)
        for c, w in enumerate(want):
            if w is None:
                assert ran[c] == 0, (it, c)
            else:
                assert ran[c] == 1 and paths[c].tolist() == w, (it, c)
                segmented += 1
                with_events += len(set(w)) > 1
    assert segmented > 20 and with_events > 10


# ---------------------------------------------------------------------------------------------------------------------------------------
# CanvasClean with -s -r -g (MedianByGC): CanvasClean.cs:328-352 (RemoveBigBins), :363-412 (SignificantlyDifferent, RemoveOutliers),
# :207-239 (RemoveBinsWithExtremeGC), :163-199 (NormalizeByGC), EnrichmentUtilities.cs:65-86 (GetCountsByGC), Utilities.cs:470-474 (Median)
def _significantly_different(a, b):
    mu = (float(a) + float(b)) / 2
    if np.float32(a) + np.float32(b) == 0:
        return False
    da, db = float(a) - mu, float(b) - mu
    return (da * da + db * db) / mu > 6.635


def _median_f32(values):
    s = np.sort(np.asarray(values, np.float32))
    n = len(s)
    return float(s[n // 2]) if n % 2 else float((s[n // 2 - 1] + s[n // 2]) / np.float32(2))


# CanvasClean.GetWeightedCounts (CanvasClean.cs:107-134) + Utilities.WeightedQuantiles / WeightedMedian (Utilities.cs:493-520): GC values with
# fewer than 100 bins borrow their neighbours' counts at half the weight per step; reached with -w below 100
def _py_weighted_median(by_gc, gc):
    pairs = []
    radius, weight = 0, np.float32(1)
    while len(pairs) < 100:
        hi, lo = gc + radius, gc - radius
        if hi >= len(by_gc) and lo < 0:
            break
        if hi < len(by_gc):
            pairs += [(c, weight) for c in by_gc[hi]]
        if lo != hi and lo >= 0:
            pairs += [(c, weight) for c in by_gc[lo]]
        radius += 1
        weight = np.float32(weight / np.float32(2))
    acc = 0.0
    for _, w in pairs:
        acc += float(w)
    total = float(np.float32(acc))                                     # LINQ Sum over a float selector: double accumulator, float result
    cumulative, quantile = 0.0, 0.0
    for c, w in sorted(pairs, key=lambda t: t[0]):                      # OrderBy: stable
        cumulative += float(w)
        if cumulative / total <= float(np.float32(0.5)):
            quantile = float(c)
    return quantile


def py_clean(bins, is_autosome, min_bins_weighted=100):
    """bins: list of [chr, start, stop, count(float32), gc]"""
    sizes = sorted(b[2] - b[1] for b in bins)                      # RemoveBigBins
    index = int(0.98 * float(len(bins)))
    if index < len(sizes):
        bins = [b for b in bins if b[2] - b[1] <= sizes[index]]
    kept = []                                                      # RemoveOutliers
    for i, b in enumerate(bins):
        prev = bins[i - 1] if i > 0 else None
        nxt = bins[i + 1] if i < len(bins) - 1 else None
        if prev is not None and prev[0] != b[0] and nxt is not None and nxt[0] != b[0]:
            continue
        if (prev is not None and prev[0] == b[0] and not _significantly_different(b[3], prev[3])) \
                or (nxt is not None and nxt[0] == b[0] and not _significantly_different(b[3], nxt[3])) or (prev is None and nxt is None):
            kept.append(b)
    bins = kept
    per_gc = [0] * 101                                             # RemoveBinsWithExtremeGC(bins, 100)
    total = 0.0
    for b in bins:
        if is_autosome[b[0]]:
            per_gc[b[4]] += 1
            total += 1
    threshold = min(100, max(min_bins_weighted, int(total / 101)))
    stripped = [b for b in bins if per_gc[b[4]] >= threshold]
    if not stripped:
        return bins
    bins = [list(b) for b in stripped]
    by_gc = [[] for _ in range(101)]                               # NormalizeByGC
    autosomal = []
    for b in bins:
        if is_autosome[b[0]]:
            by_gc[b[4]].append(b[3])
            autosomal.append(b[3])
    global_median = _median_f32(autosomal)
    medians = [_median_f32(v) if len(v) >= 100 else _py_weighted_median(by_gc, g) for g, v in enumerate(by_gc)]
    for b in bins:
        m = medians[b[4]]
        if m is not None and m > 0:
            b[3] = np.float32(global_median * float(b[3]) / m)
    return bins


def test_clean_median_by_gc_two_restatements():
    from canvas_amd import CLEAN_GCNORM, CLEAN_FILTSIZE, CLEAN_OUTLIERS
    rng = np.random.RandomState(777)
    normalised = dropped = 0
    for it in range(6):
        nchr = int(rng.randint(2, 6))
        is_auto = np.ones(nchr, np.uint8); is_auto[-1] = 0
        bins = []
        for c in range(nchr):
            n = int(rng.randint(1, 1500))
            pos = 0
            for _ in range(n):
                size = int(rng.choice([100, 101, 105, 140, 400], p=[.5, .3, .15, .04, .01]))
                gc = int(np.clip(rng.normal(45, 4 if it % 2 else 9), 0, 100))
                lam = 80.0 * (1 + (gc - 45) * 0.012) * (1 if c else 1.5)
                cnt = np.float32(rng.poisson(max(lam, 0.0)) if rng.rand() > 0.02 else rng.choice([0, 0, 400]))
                bins.append([c, pos, pos + size, cnt, gc])
                pos += size + int(rng.randint(0, 50))
        want = py_clean(bins, is_auto)
        a = [np.array([b[k] for b in bins]) for k in range(5)]
        got = O.clean(a[0], a[1], a[2], a[3], a[4], is_auto, np.zeros(nchr, np.uint8), CLEAN_GCNORM | CLEAN_FILTSIZE | CLEAN_OUTLIERS)
        assert len(got["chr"]) == len(want), it
        dropped += len(bins) - len(want)
        normalised += sum(1 for b in want if float(b[3]) != float(int(b[3])))
        assert got["chr"].tolist() == [b[0] for b in want] and got["start"].tolist() == [b[1] for b in want]
        assert (got["count"].view(np.uint32) == np.array([b[3] for b in want], np.float32).view(np.uint32)).all(), it
    assert normalised > 1000 and dropped > 100                     # the cases did reach the normalisation and the filters


# ---------------------------------------------------------------------------------------------------------------------------------------
# CBS statistics: the block-pruned searches of CBSTStatistic.cs (TMaxO :19-343, HTMaxP :354-590, TMaxP :599-930) against a search over
# every arc.  What the three compute, read from the C#: the maximum over circular arcs of n / (len (n - len)) * (arc sum)^2, arc lengths
# al0..n-al0 (HTMaxP: al0..k), TMaxO / TMaxP seeded with the arc between the global maximum and minimum of the partial sums whatever its
# length (:686-700), then scaled by the residual variance.  Arc and complement are the same arc mathematically but not in rounding, hence
# the 1e-11 relative tolerance; bit-level agreement between oracle and device is what tests/test_cbs_gpu.py and tools/soak_cbs.py check.
def _max_min_seed(px):
    n = len(px)
    hi = lo = acc = 0.0
    at_hi = at_lo = n
    for i in range(n):
        acc += px[i]
        if acc < lo:
            lo, at_lo = acc, i + 1
        if acc > hi:
            hi, at_hi = acc, i + 1
    rj = abs(at_hi - at_lo)
    return n / (rj * (float(n) - rj)) * (hi - lo) ** 2


def _every_arc_statistic(px, tss, longest, al0, seeded):
    n = len(px)
    sums = np.concatenate([[0.0], np.cumsum(np.concatenate([px, px]))])
    best = _max_min_seed(px) if seeded else 0.0
    for length in range(al0, longest + 1):
        arc = np.abs(sums[length:length + n] - sums[:n]).max()
        best = max(best, n / (length * (float(n) - length)) * arc ** 2)
    if tss <= best + 0.0001:
        tss = best + 1.0
    return best / ((tss - best) / (n - 2.0))


def test_cbs_statistics_against_every_arc():
    rng = np.random.RandomState(7)
    for it in range(250):
        n = int(rng.randint(5, 450))
        x = rng.normal(0, 1, n)
        if rng.rand() < 0.5:
            a = int(rng.randint(0, n)); x[a:a + int(rng.randint(1, 60))] += rng.normal(0, 3)
        if rng.rand() < 0.3:
            x = np.round(x, 1)                                       # exact ties
        x -= x.mean()
        tss = float(np.sum(x * x))
        if tss == 0:
            continue
        want = _every_arc_statistic(x, tss, n - 2, 2, True)
        ostat, iseg, _ = O.tmaxo(x, 2)
        assert abs(ostat - want) <= 1e-11 * max(1.0, want), (it, n)
        assert 0 <= iseg[0] < iseg[1] <= n
        assert abs(O.tmaxp(x, tss, 2) - want) <= 1e-11 * max(1.0, want), (it, n)
        if n >= 60:
            k = int(rng.choice([10, 25, 40]))
            assert abs(O.htmaxp(x, tss, k, 2) - _every_arc_statistic(x, tss, k, 2, False)) <= 1e-11 * max(1.0, want), (it, n, k)


# ---------------------------------------------------------------------------------------------------------------------------------------
# CanvasNormalize: WeightedAverageReferenceGenerator.cs:38-68, BinCounts.cs:30-62, LSNormRatioCalculator.cs:20-47,
# CanvasNormalizeUtilities.cs:23-33
def _median_f64(values):
    s = sorted(float(v) for v in values)
    n = len(s)
    return s[n // 2] if n % 2 else (s[n // 2 - 1] + s[n // 2]) / 2


def test_normalize_two_restatements():
    rng = np.random.RandomState(31)
    for it in range(20):
        n = int(rng.randint(1, 400))
        on = None if it % 3 else np.sort(rng.choice(n, size=max(1, n // 3), replace=False)).astype(np.int32)
        controls = [np.round(rng.gamma(4, 20 * (s + 1), n), int(rng.randint(0, 4))) * (rng.rand(n) > 0.1) for s in range(int(rng.randint(2, 5)))]
        weights = []
        for c in controls:
            m = _median_f64(c if on is None else [c[i] for i in on])
            weights.append(1.0 / m if m > 0 else 0.0)
        total = sum(weights)
        weights = [w / total for w in weights]
        reference = []
        for j in range(n):
            acc = 0.0
            for w, c in zip(weights, controls):
                acc += w * float(c[j])
            reference.append(acc)
        got_ref, got_w = O.norm_weighted_reference(controls, on)
        assert got_w.tolist() == weights and got_ref.tolist() == reference, it
        sample = np.round(rng.gamma(4, 25, n), 2).astype(np.float32)
        ref32 = np.asarray(reference, np.float32)                      # the reference file is read back as float counts
        ploidy = rng.choice([1, 2, 2, 2, 3], n).astype(np.int32) if it % 2 else None
        sm = _median_f64(sample if on is None else [sample[i] for i in on])
        rm = _median_f64(ref32 if on is None else [ref32[i] for i in on])
        factor = rm / sm if (sm > 0 and rm > 0) else 1.0
        keep, ratios, counts = [], [], []
        for j in range(n):
            if ref32[j] < 1:
                continue
            ratio = np.float32(float(np.float32(sample[j] / ref32[j])) * factor)
            keep.append(j)
            ratios.append(ratio)
            counts.append(np.float32(float(ratio) * (40.0 * (2 if ploidy is None else int(ploidy[j])) / 2.0)))
        k, r, c = O.norm_ratio(sample, ref32, on, mode=0, ploidy=ploidy)
        assert k.tolist() == keep, it
        assert (r.view(np.uint32) == np.asarray(ratios, np.float32).view(np.uint32)).all() and (c.view(np.uint32) == np.asarray(counts, np.float32).view(np.uint32)).all(), it


# ---------------------------------------------------------------------------------------------------------------------------------------
# CanvasClean, whole-genome branches (more than 500 000 bins): Main's order of stages (CanvasClean.cs:473-528), GetLocalStandardDeviation
# (:268-300, :243-258), NormalizeVarianceByGC (:34-97, the GC values that survive the strip all have 100+ bins, so the quartiles are the plain
# ones), the second NormalizeByGC, RemoveBinsWithExtremeLocalSD (:308-322); Utilities.cs:199-257 (Mean / StandardDeviation), :428-462 (Median / Mad).
# Vectorised with numpy where the C# order of the floating-point operations is kept (sequential sums run column by column).
def _np_normalize_by_gc(count, gc, auto):
    global_median = _median_f32(count[auto])
    out = count.copy()
    for g in np.unique(gc):
        sel = auto & (gc == g)
        if sel.sum() >= 100:
            m = _median_f32(count[sel])
            if m > 0:
                rows = gc == g
                out[rows] = (global_median * count[rows].astype(np.float64) / m).astype(np.float32)
    return out


def np_clean_whole_genome(chrom, start, stop, count, gc, is_autosome):
    size = stop - start                                              # RemoveBigBins
    keep = size <= np.sort(size)[int(0.98 * float(len(size)))]
    chrom, start, stop, count, gc = [a[keep] for a in (chrom, start, stop, count, gc)]
    n = len(count)                                                   # RemoveOutliers
    c64 = count.astype(np.float64)

    def different(a, b):
        mu = (a + b) / 2
        with np.errstate(invalid="ignore", divide="ignore"):
            chi2 = ((a - mu) ** 2 + (b - mu) ** 2) / mu
        return np.where(a + b == 0, False, chi2 > 6.635)
    same_prev = np.zeros(n, bool); same_next = np.zeros(n, bool)
    same_prev[1:] = chrom[1:] == chrom[:-1]; same_next[:-1] = chrom[:-1] == chrom[1:]
    ok_prev = np.zeros(n, bool); ok_next = np.zeros(n, bool)
    ok_prev[1:] = same_prev[1:] & ~different(c64[1:], c64[:-1]); ok_next[:-1] = same_next[:-1] & ~different(c64[:-1], c64[1:])
    keep = ok_prev | ok_next
    chrom, start, stop, count, gc = [a[keep] for a in (chrom, start, stop, count, gc)]
    n = len(count)
    assert n >= 50000
    diffs = (count[1:] - count[:-1]).astype(np.float64)              # GetLocalStandardDeviation
    nw = (len(diffs) - 1) // 20                                      # windows [20w, 20w+20) with 20w+20 < len(diffs)
    w = diffs[: nw * 20].reshape(nw, 20)
    total = np.zeros(nw)
    for k in range(20):
        total = total + w[:, k]
    mu = total / 20
    ss = np.zeros(nw)
    for k in range(20):
        d = w[:, k] - mu
        ss = ss + d * d
    local_sd = np.sqrt(ss / 19)
    deviation = np.full(n, -1.0)
    deviation[: nw * 20] = np.repeat(local_sd, 20)
    window_chrom = chrom[np.arange(nw) * 20]
    mads = []
    edges = [0] + [i for i in range(1, nw) if window_chrom[i] != window_chrom[i - 1]] + [nw]
    for a, b in zip(edges[:-1], edges[1:]):                          # the C# compares with the first window of the run: the same for sorted input
        med = _median_f64(local_sd[a:b])
        mads.append(_median_f64(np.abs(local_sd[a:b] - med)))
    acc = 0.0
    for m in mads:
        acc += m
    local_sd_average = acc / len(mads)
    auto = is_autosome[chrom].astype(bool)                           # RemoveBinsWithExtremeGC
    per_gc = np.bincount(gc[auto], minlength=101)
    threshold = min(100, max(100, int(float(auto.sum()) / 101)))
    keep = per_gc[gc] >= threshold
    assert keep.any()
    chrom, start, stop, count, gc, deviation, auto = [a[keep] for a in (chrom, start, stop, count, gc, deviation, auto)]
    count = _np_normalize_by_gc(count, gc, auto)
    assert len(count) > 500000
    f = np.float32                                                   # NormalizeVarianceByGC
    gq = py_quartiles(count[auto])
    global_iqr = f(gq[2] - gq[0])
    local_iqr = np.full(101, -1, np.float32); local_median = np.full(101, -1, np.float32)
    for g in range(101):
        v = count[auto & (gc == g)]
        if len(v):
            q = py_quartiles(v)
            local_iqr[g] = f(q[2] - q[0]); local_median[g] = q[1]
    variance_normalised = bool((global_iqr * f(2) < local_iqr[10:90]).any())
    if variance_normalised:
        scaled = local_iqr[gc] * f(0.8)
        rows = ~(global_iqr >= scaled)
        ratio = (scaled[rows] / global_iqr).astype(np.float32)
        med = local_median[gc][rows]
        count = count.copy()
        count[rows] = med + (count[rows] - med) / ratio
        count = _np_normalize_by_gc(count, gc, auto)
    keep = ~((deviation > 20 * 2.0) & (local_sd_average > 5.0))      # RemoveBinsWithExtremeLocalSD
    return chrom[keep], start[keep], count[keep], local_sd_average, variance_normalised, int((~keep).sum())


def test_clean_whole_genome_branches_two_restatements():
    from canvas_amd import CLEAN_GCNORM, CLEAN_FILTSIZE, CLEAN_OUTLIERS, CLEAN_LOCALSD
    rng = np.random.RandomState(2024)
    per_chr = [270_000, 240_000, 160_000, 50_000]
    is_auto = np.array([1, 1, 1, 0], np.uint8)
    n = sum(per_chr)
    chrom = np.repeat(np.arange(4), per_chr).astype(np.int32)
    size = rng.choice([100, 101, 105, 140, 400], n, p=[.5, .3, .15, .04, .01]).astype(np.int32)
    start = np.concatenate([np.cumsum(np.r_[0, s[:-1] + 7]) for s in np.split(size, np.cumsum(per_chr)[:-1])]).astype(np.int32)
    gc = np.clip(rng.normal(45, 6, n), 0, 100).astype(np.int32)
    mean = 90.0 * (1 + (gc - 45) * 0.01)
    spread = np.where((gc >= 30) & (gc <= 36), 4.0, 1.0)              # GC values whose spread is far beyond the genome's: variance normalisation runs
    noise = np.repeat(rng.choice([1.0, 5.0], n // 1500 + 1), 1500)[:n]  # FFPE-like: the local SD wanders, part of it beyond the filter's threshold
    count = np.rint(np.maximum(rng.normal(mean, np.sqrt(mean) * spread * noise), 0)).astype(np.float32)
    want = np_clean_whole_genome(chrom, start, start + size, count, gc, is_auto)
    assert want[4] and want[5] > 1000 and want[3] > 5.0               # both branches taken
    got = O.clean(chrom, start, start + size, count, gc, is_auto, np.zeros(4, np.uint8), CLEAN_GCNORM | CLEAN_FILTSIZE | CLEAN_OUTLIERS | CLEAN_LOCALSD)
    assert got["local_sd"] == want[3]
    assert len(got["chr"]) == len(want[0]) and (got["chr"] == want[0]).all() and (got["start"] == want[1]).all()
    assert (got["count"].view(np.uint32) == want[2].view(np.uint32)).all()


# ---------------------------------------------------------------------------------------------------------------------------------------
# Joint HMM over several samples of a pedigree (isPerSample = false): HiddenMarkovModelsRunner.cs:74-152 (per-chromosome median and variance,
# RemoveOutliers, NB tables), Distributions.cs:255-316 (the best genotype combination per copy-number state and what the transition is charged
# on), DistributionUtilities.cs:11-40 (GetGenotypeCombinations; distinct permutations in lexicographic order, Combinatorics/Permutations.cs),
# Utilities.cs:290-302 (Variance), HMM.cs:60-128
def py_genotype_combinations(n_samples, state):
    import itertools
    n_samples = min(n_samples, 4)
    if state == 2:
        return [[2] * n_samples]
    combos = []
    for n_diploid in range(n_samples):
        combos += [list(p) for p in sorted(set(itertools.permutations([state] * (n_samples - n_diploid) + [2] * n_diploid)))]
    return combos or [[state]]


def py_hmm_joint_chromosome(samples, n_states=5, min_size=10):
    S, T = len(samples), len(samples[0])
    if T <= min_size:
        return None
    log = lambda v: math.log(v) if v > 0 else -math.inf
    haploid, variance = [], []
    for s in range(S):
        col = [float(v) for v in samples[s]]
        haploid.append(max(1.0, _median_f64(col)) / 2.0)
        acc = 0.0
        for v in col:
            acc += v
        mu = acc / T
        ss = 0.0
        for v in col:
            ss += (v - mu) * (v - mu)
        variance.append(ss / (T - 1))
    cap = max(haploid) * n_states
    data = [[cap if samples[s][t] > cap else float(samples[s][t]) for s in range(S)] for t in range(T)]
    max_value = max(_to_int32(max(row)) for row in data)
    tables = [[py_negative_binomial(max(cn, 0.1) * haploid[s], variance[s], max_value + 10) for s in range(S)] for cn in range(n_states)]
    combos = [py_genotype_combinations(S, cn) for cn in range(n_states)]
    trans = [[0.99 if i == j else (1.0 - 0.99) / (n_states - 1) for j in range(n_states)] for i in range(n_states)]
    prior = float(np.float32(1.0) / np.float32(n_states))

    def best_combination(row_data, cn):                             # does not depend on the state we come from
        best, best_l = [], -sys.float_info.max
        for combo in combos[cn]:
            e = 1.0
            for s, g in enumerate(combo):
                c = _to_int32(row_data[s])
                if g in (0, 1):
                    e *= max(tables[0][s][c], tables[1][s][c])
                elif g in (3, 4):
                    e *= max(tables[3][s][c], tables[4][s][c])
                else:
                    e *= tables[g][s][c]
            if math.isnan(e) or math.isinf(e):
                e = 0.0
            if best_l < e:
                best, best_l = combo, e
        return best, best_l

    def viterbi_likelihood(chosen, cn, row):
        best, best_l = chosen
        if max(trans[row]) == trans[row][2]:
            charge = min(trans[row][g] for g in best)
        elif cn == 2:
            charge = trans[row][2]
        else:
            charge = min(trans[row][g] for g in best if g != 2)
        return log(best_l) + log(charge)
    chosen = [best_combination(data[0], j) for j in range(n_states)]
    score = [log(prior) + viterbi_likelihood(chosen[j], j, 0) - log(trans[0][j]) for j in range(n_states)]
    back = []
    for t in range(1, T):
        chosen = [best_combination(data[t], j) for j in range(n_states)]
        new, frm = [], []
        for j in range(n_states):
            state, best = 0, -sys.float_info.max
            for i in range(n_states):
                cand = score[i] + viterbi_likelihood(chosen[j], j, i)
                if cand > best:
                    state, best = i, cand
            new.append(best)
            frm.append(state)
        score = new
        back.append(frm)
    state, best = -1, -sys.float_info.max
    for i in range(n_states):
        if score[i] > best:
            state, best = i, score[i]
    path = [state]
    for frm in reversed(back):
        state = frm[state]
        path.append(state)
    return path[::-1]


def test_joint_hmm_two_restatements():
    rng = np.random.RandomState(8642)
    segmented = with_events = 0
    for it in range(24):
        S = int(rng.choice([1, 2, 3, 4]))
        T = int(rng.choice([9, 11, 40, 150, 260]))
        depths = rng.choice([12.0, 35.0, 80.0], S)
        samples = [_random_coverage(rng, T, float(d)) for d in depths]
        if rng.rand() < 0.5 and T > 30:                              # an event the samples share
            a = int(rng.randint(0, T - 10)); b = a + int(rng.randint(5, T // 2))
            f = rng.choice([0.5, 1.5, 2.0])
            for s in range(S):
                samples[s][a:b] = np.round(samples[s][a:b] * f, 2)
        want = py_hmm_joint_chromosome(samples)
        ran, path = O.hmm_chromosome(samples, per_sample=False)
        if want is None:
            assert ran == 0, it
        else:
            assert ran == 1 and path.tolist() == want, (it, S, T)
            segmented += 1
            with_events += len(set(want)) > 1
    assert segmented > 15 and with_events > 8


# ---------------------------------------------------------------------------------------------------------------------------------------
# GCContentWeighted binning (-m 5): Utilities.NonZeroMean (Utilities.cs:136-151), MeanFragmentSize (CanvasBin.cs:164-174), the read-GC profile
# (:452-496), ComputeObservedVsExpectedGC (:330-394, no manifest), the weighted count of BinCountsForChromosome (:606-636)
def py_gc_weighted(bases, possible, observed, fragment_lengths, bin_size):
    def non_zero_mean(values):
        total = count = 0
        for v in values:
            if v > 0:
                total += int(v); count += 1
        return total // count if count else 0
    mean_fragment = non_zero_mean([non_zero_mean(f) for f in fragment_lengths])
    read_gc = []
    for b, fl in zip(bases, fragment_lengths):
        is_gc = [chr(c) in "CcGg" for c in b]
        g = [0] * len(b)
        for pos in range(len(b) - mean_fragment * 3 - 1):
            frag = mean_fragment if fl[pos] == 0 else min(int(fl[pos]), mean_fragment * 3)
            g[pos] = min(100 * sum(is_gc[pos:pos + frag]) // frag, 101)
        read_gc.append(g)
    expected = [0] * 101; seen = [0] * 101
    for g, o in zip(read_gc, observed):
        for i, v in enumerate(g):
            expected[v] += 1; seen[v] += int(o[i])
    f = np.float32
    sum_expected, sum_seen = sum(expected), sum(seen)
    weights = []
    for k in range(101):
        e = expected[k] or 1; s = seen[k] or 1
        weights.append((f(s) / f(e)) * (f(sum_expected) / f(sum_seen)))
    out = []
    for b, p, o, g in zip(bases, possible, observed, read_gc):
        bins = []
        pos = 0
        while b[pos] == ord("n"):
            pos += 1
        nucleotides = gc = poss = 0
        start = -1
        acc = f(0)
        while pos < len(b):
            if start == -1:
                start = pos
            nucleotides += 1
            gc += chr(b[pos]) in "CcGg"
            if p[pos]:
                poss += 1
                acc = f(acc + min(f(10), f(f(int(o[pos])) / weights[g[pos]])))
            if poss == bin_size:
                bins.append((start, pos + 1, int(f(100.0) * f(gc) / f(nucleotides)), int(np.rint(float(acc)))))
                nucleotides = gc = poss = 0
                start = -1
                acc = f(0)
            pos += 1
        out.append(bins)
    return out, mean_fragment, weights, read_gc


def test_gc_content_weighted_two_restatements():
    rng = np.random.RandomState(555)
    for it in range(12):
        nchr = int(rng.randint(1, 4))
        chroms = [_random_chromosome(rng, int(rng.randint(400, 1500))) for _ in range(nchr)]
        frag = []
        for b, p, o in chroms:
            fl = np.where(o > 0, rng.randint(20, 140, len(b)), 0).astype(np.int16)
            fl[rng.randint(0, len(b), 3)] = 600                        # beyond three mean fragments: capped
            frag.append(fl)
        bin_size = int(rng.randint(3, 50))
        want, mean_fragment, weights, read_gc = py_gc_weighted([c[0] for c in chroms], [c[1] for c in chroms], [c[2] for c in chroms], frag, bin_size)
        res, m, w, rgc = O.bin_gc_weighted([c[0] for c in chroms], [_pack_mask(c[1]) for c in chroms], [c[2] for c in chroms], frag, bin_size)
        assert m == mean_fragment and (w.view(np.uint32) == np.asarray(weights, np.float32).view(np.uint32)).all(), it
        for c in range(nchr):
            assert rgc[c].tolist() == read_gc[c], (it, c)
            assert list(zip(*[a.tolist() for a in res[c]])) == want[c], (it, c)


def py_predefined_gc_weighted(bases, possible, observed, read_gc, weights, starts, stops):
    """BinCountsForChromosome with usePredefinedBins and the GCContentWeighted branch (CanvasBin.cs:575-655), a second reading: the cursor starts at the first bin's Start,
    skips leading 'n', closes a bin on Stop - 1 and jumps to the next bin's Start; the weighted count is the float32 sum over the bin's possible positions"""
    f = np.float32
    out = []
    if not len(starts):
        return out
    k = 0
    pos = int(starts[0])
    while bases[pos] == ord("n"):
        pos += 1
    nucleotides = gc = 0
    acc = f(0)
    while pos < len(bases):
        nucleotides += 1
        gc += chr(bases[pos]) in "CcGg"
        if possible[pos]:
            acc = f(acc + min(f(10), f(f(int(observed[pos])) / weights[read_gc[pos]])))
        if pos == int(stops[k]) - 1:
            out.append((int(f(100.0) * f(gc) / f(nucleotides)), int(np.rint(float(acc)))))
            k += 1
            if k >= len(starts):
                break
            pos = int(starts[k]) - 1
            nucleotides = gc = 0
            acc = f(0)
        pos += 1
    return out


def test_predefined_bins_gc_content_weighted_two_restatements():
    """CanvasBin -n with -m GCContentWeighted: the oracle's bin_chromosome_predefined_weighted against the reading above, the profile and the weights from py_gc_weighted
    (every chromosome enters them, also one without bins)"""
    rng = np.random.RandomState(777)
    for it in range(10):
        nchr = int(rng.randint(2, 4))
        chroms = [_random_chromosome(rng, int(rng.randint(500, 1400))) for _ in range(nchr)]
        frag = [np.where(o > 0, rng.randint(20, 140, len(b)), 0).astype(np.int16) for b, p, o in chroms]
        _, mean_fragment, weights, read_gc = py_gc_weighted([c[0] for c in chroms], [c[1] for c in chroms], [c[2] for c in chroms], frag, 10)
        starts, stops = [], []
        for c, (b, p, o) in enumerate(chroms):
            if c == 1:
                starts.append(np.zeros(0, np.int32)); stops.append(np.zeros(0, np.int32)); continue
            L = len(b)
            s0 = np.sort(rng.choice(L - 60, 12, replace=False)); e0 = np.minimum(s0 + rng.randint(1, 60, 12), L)
            s0[0] = 0; e0[0] = max(int(e0[0]), 40)                          # the first bin starts inside the leading 'n' stretch (if any) and reaches past it
            e0[-1] = L
            starts.append(s0.astype(np.int32)); stops.append(e0.astype(np.int32))
        res = O.bin_predefined_gc_weighted([c[0] for c in chroms], [_pack_mask(c[1]) for c in chroms], [c[2] for c in chroms], frag, starts, stops)
        for c, (b, p, o) in enumerate(chroms):
            want = py_predefined_gc_weighted(b, p, o, read_gc[c], weights, starts[c], stops[c])
            k, gcs, cnts = res[c]
            assert k == len(want), (it, c)
            assert list(zip(gcs.tolist(), cnts.tolist())) == want, (it, c)


# ---------------------------------------------------------------------------------------------------------------------------------------
# Wavelets: WaveletSegmentation.cs:19-48 (GetInnerProdIter), :54-68 (GetInnerProdMax), :264-383 (FindBestUnbalancedHaarDecomposition),
# :72-115 (HardThresh), :118-171 (GetUnbalHaarVector, GetReconstructedVector), :174-185 (GetSegments), :194-234 (healing), :237-258
# (RefineSegments), :385-425 (HaarWavelets).  The reference's own known-answer test pins the non-germline flavour on one vector
# (tests/test_oracle_golden.py); this second reading runs both flavours on many.  Germline's Array.Sort with a comparison is restated as
# .NET Core's introsort (insertion sort up to 16 elements, median-of-three partition above; a case that would reach its heapsort is skipped).
def _inner_products(x):
    n = len(x)
    plus = [0.0] * (n - 1); minus = [0.0] * (n - 1)
    plus[0] = math.sqrt(1 - 1.0 / n) * x[0]
    rest = 0.0
    for v in x[1:]:
        rest += v
    mean = (x[0] + rest) / n
    minus[0] = (1.0 / math.sqrt(n * (n - 1))) * rest
    for m in range(1, n - 1):
        factor = math.sqrt(float(n - m - 1) * m / float(m + 1) / float(n - m))
        plus[m] = plus[m - 1] * factor + x[m] * math.sqrt(1.0 / (m + 1) - 1.0 / n)
        minus[m] = minus[m - 1] / factor - x[m] / math.sqrt((float(n) * n / float(m + 1)) - float(n))
    return [p - q for p, q in zip(plus, minus)], mean


def _first_largest(ipi):
    top = max(abs(v) for v in ipi)
    for k, v in enumerate(ipi):
        if abs(v) == top:
            return k + 1


class _NeedsHeapsort(Exception):
    pass


def _dotnet_sort(keys, compare):
    def swap_if_greater(a, b):
        if a != b and compare(keys[a], keys[b]) > 0:
            keys[a], keys[b] = keys[b], keys[a]

    def intro(lo, hi, depth):
        while hi > lo:
            size = hi - lo + 1
            if size <= 16:
                if size == 2:
                    swap_if_greater(lo, hi)
                elif size == 3:
                    swap_if_greater(lo, hi - 1); swap_if_greater(lo, hi); swap_if_greater(hi - 1, hi)
                elif size > 3:
                    for i in range(lo, hi):
                        j, t = i, keys[i + 1]
                        while j >= lo and compare(t, keys[j]) < 0:
                            keys[j + 1] = keys[j]
                            j -= 1
                        keys[j + 1] = t
                return
            if depth == 0:
                raise _NeedsHeapsort()
            depth -= 1
            mid = lo + (hi - lo) // 2
            swap_if_greater(lo, mid); swap_if_greater(lo, hi); swap_if_greater(mid, hi)
            pivot = keys[mid]
            keys[mid], keys[hi - 1] = keys[hi - 1], keys[mid]
            left, right = lo, hi - 1
            while left < right:
                left += 1
                while compare(keys[left], pivot) < 0:
                    left += 1
                right -= 1
                while compare(pivot, keys[right]) < 0:
                    right -= 1
                if left >= right:
                    break
                keys[left], keys[right] = keys[right], keys[left]
            keys[left], keys[hi - 1] = keys[hi - 1], keys[left]
            intro(left + 1, hi, depth)
            hi = left - 1
    n = len(keys)
    if n < 2:
        return
    depth, k = 0, n
    while k >= 1:
        depth += 1
        k //= 2
    intro(0, n - 1, 2 * depth)


def py_haar_wavelets(x, thr_lower, thr_upper, germline, mad_factor, cv, factor_of_three):
    x = [float(v) for v in x]
    n = len(x)
    ipi, mean = _inner_products(x)
    k = _first_largest(ipi)
    tree = [[[1.0, ipi[k - 1] / max(0.5, mean / 200.0), 1, k, n]]]     # node = [index, coefficient, start, split, end], positions 1-based
    while sum(node[4] - node[2] - 1 for node in tree[-1]) != 0:
        level = []
        for index, _, start, split, end in tree[-1]:
            if split - start >= 1:
                ipi, mean = _inner_products(x[start - 1:split])
                k = _first_largest(ipi)
                level.append([2 * index - 1, ipi[k - 1] / max(0.5, mean / 200.0), start, k + start - 1, split])
            if end - split >= 2:
                ipi, mean = _inner_products(x[split:end])
                k = _first_largest(ipi)
                level.append([2 * index, ipi[k - 1] / max(0.5, mean / 200.0), split + 1, k + split, end])
        tree.append(level)
    smooth = 0.0
    for v in x:
        smooth += v
    smooth /= math.sqrt(n)
    median = _median_f64(x)
    variability = median * cv if cv is not None else _median_f64([abs(v - median) for v in x])
    threshold = mad_factor * variability
    if threshold < thr_lower:
        threshold = thr_lower
    if threshold > thr_upper:
        threshold = thr_upper
    depth = len(tree)                                                   # HardThresh
    order = list(range(depth))
    if germline:
        counts = [len(level) for level in tree]
        _dotnet_sort(order, lambda a, b: (counts[b] > counts[a]) - (counts[b] < counts[a]))
        weights = [(float(r) * (1.0 - 0.8)) / depth + 0.8 for r in range(1, depth + 1)]
    else:
        weights = [1.0] * depth
    for j, level in enumerate(tree):
        for node in level:
            if abs(node[1]) <= 2 * threshold * weights[order[j]] * math.sqrt(2 * math.log(float(n))):
                node[1] = 0.0
    rec = np.full(n, 1.0 / math.sqrt(n) * smooth)                       # GetReconstructedVector
    for level in tree:
        for _, coefficient, start, split, end in level:
            span = float(end - start + 1); head = float(split - start + 1)
            vector = np.empty(end - start + 1)
            vector[: split - start + 1] = math.sqrt(1 / head - 1 / span)
            vector[split - start + 1:] = -1.0 / math.sqrt(span * span / head - span)
            rec[start - 1:end] = rec[start - 1:end] + vector * coefficient
    prelim = [0] + [i for i in range(1, n) if rec[i] - rec[i - 1] != 0]
    breakpoints = [prelim[0]]                                            # GetBreakpointsAfterHealingBadSplits
    for i in range(1, len(prelim)):
        left_start, right_start = breakpoints[-1], prelim[i]
        right_end = prelim[i + 1] if i < len(prelim) - 1 else n
        left_len, right_len = right_start - left_start, right_end - right_start
        left_median, right_median = _median_f64(x[left_start:right_start]), _median_f64(x[right_start:right_end])
        weighted = (left_len * left_median + right_len * right_median) / (right_end - left_start)
        scale = min(len(factor_of_three) - 1, int(math.ceil(math.log(min(left_len, right_len)) / math.log(3))))
        if abs(left_median - right_median) > factor_of_three[scale] * 4 * max(weighted, 50.0):
            breakpoints.append(prelim[i])
    if germline:                                                         # RefineSegments
        for i in range(1, len(breakpoints) - 1):
            left = min(5, (breakpoints[i] - breakpoints[i - 1]) // 2)
            right = min(5, (breakpoints[i + 1] - breakpoints[i]) // 2)
            best_difference = abs(_median_f64(x[breakpoints[i - 1]:breakpoints[i]]) - median)
            best = breakpoints[i]
            for j in range(breakpoints[i] - left, breakpoints[i] + right):
                difference = abs(_median_f64(x[breakpoints[i - 1]:j]) - median)
                if difference > best_difference:
                    best_difference, best = difference, j
            breakpoints[i] = best
    return breakpoints, depth


def test_haar_wavelets_two_restatements():
    rng = np.random.RandomState(606)
    ran = deep = events = 0
    for it in range(80):
        n = int(rng.choice([2, 3, 5, 21, 80, 250, 600, 600, 900]))
        level = float(rng.choice([40.0, 100.0]))
        x = rng.normal(level, level * rng.uniform(0.03, 0.15), n)
        for _ in range(int(rng.randint(0, 4))):
            if n > 40:
                a = int(rng.randint(0, n - 10)); x[a:a + int(rng.randint(3, n // 2))] *= rng.choice([0.5, 1.5, 2.0])
        x = np.round(np.maximum(x, 0), int(rng.choice([0, 2])))          # whole numbers: exact ties between inner products and medians
        germline = bool(it % 2)
        cv = None if it % 3 == 0 else float(rng.uniform(0.02, 0.2))
        f3 = np.sort(rng.uniform(0.01, 0.08, int(rng.randint(1, 9))))[::-1].copy()
        mad_factor = float(rng.choice([2.0, 5.0]))
        try:
            want, depth = py_haar_wavelets(x, 5.0, 80.0, germline, mad_factor, cv, f3.tolist())
        except _NeedsHeapsort:
            continue
        got = O.haar_wavelets(x, 5.0, 80.0, germline, mad_factor, cv, f3)
        assert got.tolist() == want, (it, n, germline)
        ran += 1; deep += depth > 16 and germline; events += len(want) > 1
    assert ran > 60 and deep > 8 and events > 20, (ran, deep, events)


# ---------------------------------------------------------------------------------------------------------------------------------------
# Utilities.MergeMultiSampleCleanedBedFile + CanvasRunner.NormalizeCanvasClean (Utilities.cs:834-920, CanvasRunner.cs:883-903): dictionaries
# keyed by chromosome and start in insertion order, a bin survives when every sample contributed a count, the stop is the last one read
def py_merge_cleaned(samples):
    stops, counts = {}, {}
    for s in samples:
        for c, a, b, v in zip(s["chr"], s["start"], s["stop"], s["count"]):
            stops.setdefault(int(c), {})[int(a)] = int(b)
            counts.setdefault(int(c), {}).setdefault(int(a), []).append(v)
    rows = []
    for c in stops:
        for a in stops[c]:
            if len(counts[c][a]) < len(samples):
                continue
            rows.append((c, a, stops[c][a], counts[c][a]))
    return rows


def test_merge_cleaned_two_restatements():
    rng = np.random.RandomState(17)
    for it in range(30):
        nchr = int(rng.randint(1, 5)); S = int(rng.randint(1, 5))
        base = []
        for c in range(nchr):
            starts = np.cumsum(rng.randint(1, 200, int(rng.randint(0, 300))))
            base += [(c, int(a), int(a) + int(rng.randint(1, 150))) for a in starts]
        samples = []
        for s in range(S):
            keep = [r for r in base if rng.rand() > 0.15]
            samples.append(dict(chr=np.array([r[0] for r in keep], np.int32), start=np.array([r[1] for r in keep], np.int32),
                                stop=np.array([r[2] + (s if rng.rand() < 0.05 else 0) for r in keep], np.int32),
                                count=rng.gamma(5, 20, len(keep)).astype(np.float32)))
        want = py_merge_cleaned(samples)
        oc, os_, oe, ocnt = O.merge_cleaned(samples)
        assert len(oc) == len(want), it
        assert list(zip(oc.tolist(), os_.tolist(), oe.tolist())) == [r[:3] for r in want], it
        for s in range(S):
            assert (ocnt[s].view(np.uint32) == np.array([r[3][s] for r in want], np.float32).view(np.uint32)).all(), (it, s)


# ---------------------------------------------------------------------------------------------------------------------------------------
# SegmentationInput.GetCoverageVariability / reportVariabilityByWindow (Segmentation.cs:304-338) and FactorOfThreeCoverageVariabilities /
# GetTripletMediansAndCMADs (:345-429): the two genome-wide inputs of HaarWavelets
def py_coverage_variability(window, per_chr):
    if sum(len(c) for c in per_chr) < 10 * window:
        return None

    def by_window(w):
        out = []
        for c in per_chr:
            for index in range(0, len(c) - w, w):
                chunk = [float(v) for v in c[index:index + w]]
                med = _median_f64(chunk)
                out.append(np.float32(_median_f64([abs(v - med) for v in chunk]) / med))
        return out
    if window > 10000:
        q1, q2, q3 = py_quartiles(by_window(10000))
        if float(np.float32(np.float32(q3 - q1) / q2)) > 0.015:
            return float(q1)
    return _median_f32(by_window(window))


def py_factor_of_three(per_chr, max_exponent=8):
    out = [0.0]
    data = [[float(v) for v in c] for c in per_chr]
    exponent = 1
    while exponent <= max_exponent:
        cmads, nxt = [], []
        for c in data:
            medians = []
            for i in range(len(c) // 3):
                a, b, d = sorted(c[3 * i:3 * i + 3])
                medians.append(b)
                cmads.append((d - a) / 2.0 / b)
            nxt.append(medians)
        data = nxt
        if len(cmads) < 50:
            out += [out[-1]] * (max_exponent - len(out) + 1)
            break
        out.append(_median_f64(cmads))
        exponent += 1
    return out


def test_wavelet_genome_inputs_two_restatements():
    rng = np.random.RandomState(909)
    for it in range(25):
        per_chr = [np.round(np.maximum(rng.normal(100, rng.uniform(3, 20), int(rng.randint(3, 9000))), 1.0), 2) for _ in range(int(rng.randint(1, 5)))]
        if it % 4 == 0:                                                # a long chromosome: the 10 000-bin windows of the large-window branch
            per_chr.append(np.round(np.maximum(rng.normal(100, 8, 125_000) * np.repeat(rng.choice([1.0, 1.0, 1.5], 125), 1000), 1.0), 2))
        window = int(rng.choice([11, 100, 2000, 10001, 12000]))
        assert O.coverage_variability(window, per_chr) == py_coverage_variability(window, per_chr), (it, window)
        assert O.factor_of_three(per_chr).tolist() == py_factor_of_three(per_chr), it


# ---------------------------------------------------------------------------------------------------------------------------------------
# CBS, SegmentSplitUndo.SDUndo (ChangePoint.cs:155-202; Helper.cs:30-44 Median, :245-263 ArgMin): applied here to the segmentation the oracle
# finds without undo (same seed, so the same change points before the undo step) and compared with the oracle's own SDUndo run
def py_sd_undo(x, lengths, trimmed_sd, change_sd=3.0):
    change_sd *= trimmed_sd
    ends = np.cumsum(lengths).tolist()
    while len(ends) > 1:
        starts = [0] + ends[:-1]
        medians = [_median_f64(x[a:b]) for a, b in zip(starts, ends)]
        gaps = [abs(b - a) for a, b in zip(medians[:-1], medians[1:])]
        smallest, at = gaps[0], 0
        for i in range(1, len(gaps)):
            if gaps[i] < smallest:
                smallest, at = gaps[i], i
        if smallest < change_sd:
            del ends[at]
        else:
            break
    return np.diff([0] + ends).tolist()


def test_cbs_sd_undo_two_restatements():
    rng = np.random.RandomState(1212)
    undone = 0
    for it in range(12):
        parts = [rng.normal(m, 1.0, int(rng.randint(15, 120))) for m in rng.choice([0.0, 0.8, 2.5, 4.0, -3.0], int(rng.randint(2, 7)))]
        x = np.round(np.concatenate(parts), 2)
        sd = float(rng.choice([0.4, 1.0, 2.0]))
        plain, _ = O.cbs_chromosome(x, seed=it + 1, undo=0)
        with_undo, _ = O.cbs_chromosome(x, seed=it + 1, undo=2, trimmed_sd=sd)
        want = py_sd_undo(x, plain.tolist(), sd) if len(plain) > 1 else plain.tolist()
        assert with_undo.tolist() == want, it
        undone += len(want) < len(plain)
    assert undone >= 3


# ---------------------------------------------------------------------------------------------------------------------------------------
# CBS: the recursion and the permutation tests (ChangePoint.cs:44-136 ChangePoints, :291-404 FindChangePoints, :407-421 XPerm,
# CBSTStatistic.cs:935-1015 TPermP) restated on top of the oracle's statistics, which are checked on their own above (TMaxO / HTMaxP / TMaxP
# against every arc) — what is second-sourced here is the control flow around them: segment stack, hybrid switch, early stopping against the
# boundary, the two TPermP tests, and the order in which random numbers are drawn (numpy's MT19937 stream, one 32-bit output per NextDouble).
class _DotNetRandom:
    def __init__(self, seed):
        self.rs = np.random.RandomState(int(seed) & 0xFFFFFFFF)
        self.buf, self.at, self.drawn = [], 0, 0

    def next_double(self):
        if self.at == len(self.buf):
            self.buf = self.rs.randint(0, 2 ** 32, size=1 << 16, dtype=np.uint64).tolist()
            self.at = 0
        v = self.buf[self.at]
        self.at += 1; self.drawn += 1
        return v * (1.0 / 4294967296.0)


def _py_tpermp(n1, n2, n, data, offset, n_perm, rnd):
    import ctypes as C
    if n1 == 1 or n2 == 1:
        return float(n_perm) / n_perm
    px = [0.0] * n
    sum1 = sum2 = tss = 0.0
    for i in range(n1):
        px[i] = data[offset + i]; sum1 += px[i]; tss += px[i] ** 2
    for i in range(n1, n):
        px[i] = data[offset + i]; sum2 += px[i]; tss += px[i] ** 2
    rn1, rn2 = float(n1), float(n2)
    rn = rn1 + rn2
    xbar = (sum1 + sum2) / rn
    tss -= rn * xbar ** 2
    if n1 <= n2:
        m1, rm1 = n1, rn1
        ostat = 0.99999 * abs(sum1 / rn1 - xbar)
        tstat = ostat ** 2 * rn1 * rn / rn2
    else:
        m1, rm1 = n2, rn2
        ostat = 0.99999 * abs(sum2 / rn2 - xbar)
        tstat = ostat ** 2 * rn2 * rn / rn1
    tstat = tstat / ((tss - tstat) / (rn - 2.0))
    rejected = 0
    if not (tstat > 25 and m1 >= 10):
        for _ in range(n_perm):
            s = 0.0
            for i in range(n - 1, n - m1 - 1, -1):
                j = int(rnd.next_double() * (i + 1))
                j = i if j > i else j
                px[i], px[j] = px[j], px[i]
                s += px[i]
            if ostat <= abs(s / rm1 - xbar):
                rejected += 1
    return float(rejected) / n_perm


def _py_find_change_points(x, tss, n_perm, alpha, hybrid, delta, sbdry, rnd, al0=2, hk=25):
    import ctypes as C
    n = len(x)
    arr = np.ascontiguousarray(x, np.float64); sx = np.zeros(n); iseg = np.zeros(2, np.int32); ostat = C.c_double()
    O.lib.orc_tmaxo(O._p(arr), n, C.c_double(tss), O._p(sx), O._p(iseg), C.byref(ostat), al0)
    ostat = ostat.value
    ostat1 = math.sqrt(ostat)
    ostat *= 0.99999
    if ostat1 <= 0.1:
        return []
    shorter = min(int(iseg[1] - iseg[0]), n - int(iseg[1]) + int(iseg[0]))
    if not (ostat1 >= 7.0 and shorter >= 10):
        if hybrid:
            p1 = O.lib.orc_tailp(ostat1, delta, n, 100, 1e-6)
            if p1 > alpha:
                return []
            limit = int((alpha - p1) * n_perm)
        else:
            limit = int(alpha * n_perm)
        k = limit * (limit + 1) // 2 + 1
        rejected = 0
        px = np.zeros(n)
        for np_ in range(1, n_perm + 1):
            perm = list(x)                                             # XPerm
            for i in range(n - 1, -1, -1):
                j = int(rnd.next_double() * (i + 1))
                j = i if j > i else j
                perm[i], perm[j] = perm[j], perm[i]
            px[:] = perm
            pstat = O.lib.orc_htmaxp(hk, C.c_double(tss), O._p(px), n, O._p(sx), al0) if hybrid else O.lib.orc_tmaxp(C.c_double(tss), O._p(px), n, O._p(sx), al0)
            if ostat <= pstat:
                rejected += 1; k += 1
            if rejected > limit:
                return []
            if np_ >= sbdry[k - 1]:
                break
    a, b = int(iseg[0]), int(iseg[1])
    if b == n:
        return [a]
    if a == 0:
        return [b]
    found = []
    if _py_tpermp(a, b - a, b, x, 0, n_perm, rnd) <= alpha:
        found.append(a)
    if _py_tpermp((n - a) - (n - b), n - b, n - a, x, a, n_perm, rnd) <= alpha:
        found.append(b)
    return found


def py_cbs_change_points(data, seed, sbdry, n_perm, alpha=0.01, min_width=2, k_max=25, n_min=200):
    rnd = _DotNetRandom(seed)
    data = [float(v) for v in data]
    ends = [0, len(data)]
    closed = []
    while len(ends) > 1:
        lo, hi = ends[-2], ends[-1]
        n = hi - lo
        found = []
        if n >= 2 * min_width:
            cur = data[lo:hi]
            if max(cur) != min(cur):
                acc = 0.0
                for v in cur:
                    acc += v
                mean = acc / n
                cur = [v - mean for v in cur]
                tss = 0.0
                for v in cur:
                    tss += v * v
                hybrid = n_min < n
                found = _py_find_change_points(cur, tss, n_perm, alpha, hybrid, (k_max + 1.0) / n if hybrid else 0.0, sbdry, rnd)
        if not found:
            closed.append(hi)
            ends.pop()
        else:
            ends[-1:-1] = [lo + f for f in found]
    closed.reverse()
    return np.diff([0] + closed).tolist(), rnd.drawn


def test_cbs_recursion_two_restatements():
    rng = np.random.RandomState(3434)
    n_perm = 200
    sbdry = O.cbs_boundary(n_perm, 0.01)
    split = hybrid_runs = 0
    for it in range(14):
        parts = [rng.normal(m, 1.0, int(rng.randint(8, 160))) for m in rng.choice([0.0, 0.6, 1.2, 3.0, -2.0], int(rng.randint(1, 6)))]
        x = np.round(np.concatenate(parts), 2)
        seed = int(O.cbs_seeds(24)[it % 24])
        want, drawn = py_cbs_change_points(x, seed, sbdry, n_perm)
        got, stats = O.cbs_chromosome(x, seed=seed, sbdry=sbdry, n_perm=n_perm, undo=0)
        assert got.tolist() == want, (it, len(x))
        assert int(stats[3] + stats[4]) == drawn, (it, stats.tolist(), drawn)     # the same number of random numbers consumed
        split += len(want) > 1; hybrid_runs += len(x) > 200
    assert split >= 6 and hybrid_runs >= 4


# ---------------------------------------------------------------------------------------------------------------------------------------
# TailProbability.TailP / Nu / IntegralInvT1tSq (TailProbability.cs:8-107) with the normal CDF written through erfc in place of MathNet's:
# agreement to 1e-10 relative — the value is only compared with the p-value cutoff
def py_tailp(b, delta, m, n_grid=100, tol=1e-6):
    cdf = lambda z: 0.5 * math.erfc(-z / math.sqrt(2.0))

    def nu(x):
        if x <= 0.01:
            return math.exp(-0.583 * x)
        l1 = math.log(2.0) - 2 * math.log(x)
        l0 = l1
        k, dk = 2, 0.0
        for _ in range(k):
            dk += 1
            l1 -= 2.0 * cdf(-x * math.sqrt(dk) / 2.0) / dk
        while abs((l1 - l0) / l1) > tol:
            l0 = l1
            for _ in range(k):
                dk += 1
                l1 -= 2.0 * cdf(-x * math.sqrt(dk) / 2.0) / dk
            k *= 2
        return math.exp(l1)

    def integral(x, a):
        y = x + a - 0.5
        v = (8.0 * y) / (1.0 - 4.0 * y ** 2) + 2.0 * math.log((1.0 + 2.0 * y) / (1.0 - 2.0 * y))
        y = x - 0.5
        return v - (8.0 * y) / (1.0 - 4.0 * y ** 2) - 2.0 * math.log((1.0 + 2.0 * y) / (1.0 - 2.0 * y))
    step = (0.5 - delta) / n_grid
    scaled = b / math.sqrt(m)
    tl, t, total = 0.5 - step, 0.5 - 0.5 * step, 0.0
    for _ in range(n_grid):
        tl += step; t += step
        total += nu(scaled / math.sqrt(t * (1 - t))) ** 2 * integral(tl, step)
    return 2.0 * (9.973557e-2 * b ** 3 * math.exp(-b ** 2 / 2) * total)


def test_tail_probability_two_restatements():
    rng = np.random.RandomState(88)
    for it in range(10):
        m = int(rng.randint(201, 3000))
        b = float(rng.uniform(2.5, 7.0))                             # small b: the Nu series needs 10^5 terms per grid point
        delta = 26.0 / m
        want = py_tailp(b, delta, m)
        got = O.lib.orc_tailp(b, delta, m, 100, 1e-6)
        assert abs(got - want) <= 1e-10 * max(abs(want), 1e-300), (it, b, m, got, want)


# ---------------------------------------------------------------------------------------------------------------------------------------
# GetBoundary.ComputeBoundary / EtaBoundary / PExceed (GetBoundary.cs:9-150): the sequential stopping boundary of the permutation tests, with
# scipy's hypergeometric CDF in place of the reference's R.phyper port and lgamma for MathNet's BinomialLn
def py_cbs_boundary(n_perm, alpha, eta=0.05, tol=1e-2):
    from scipy.stats import hypergeom
    max_ones = int(math.floor(n_perm * alpha) + 1)
    sbdry = [0] * (max_ones * (max_ones + 1) // 2)
    draws = np.arange(1, n_perm + 1)

    def binomial_ln(n, k):
        if k < 0 or n < 0 or k > n:
            return -math.inf
        return math.lgamma(n + 1) - math.lgamma(k + 1) - math.lgamma(n - k + 1)

    def eta_boundary(eta0, ones, at, table):
        k = 0
        for i in range(1, n_perm + 1):
            if table[k][i - 1] <= eta0:
                sbdry[at + k] = i
                k += 1

    def p_exceed(ones, at):
        log = lambda v: math.log(v) if v > 0 else -math.inf
        whole = binomial_ln(n_perm, ones)
        p = math.exp(binomial_ln(n_perm - sbdry[at], ones) - whole)
        if ones >= 2:
            p += math.exp(log(sbdry[at]) + binomial_ln(n_perm - sbdry[at + 1], ones - 1) - whole)
        if ones >= 3:
            n1, n2, n, k = sbdry[at], sbdry[at + 1], n_perm - sbdry[at + 2], ones - 2
            p += math.exp(log(n1) + log(n1 - 1.0) - math.log(2.0) + binomial_ln(n, k) - whole) + math.exp(log(n1) + log(n2 - n1) + binomial_ln(n, k) - whole)
        for i in range(4, ones + 1):
            n1, n2, n3 = sbdry[at + i - 4], sbdry[at + i - 3], sbdry[at + i - 2]
            n, k = n_perm - sbdry[at + i - 1], ones - i + 1
            tail = binomial_ln(n, k) - whole
            p += (math.exp(binomial_ln(n1, i - 1) + tail) + math.exp(binomial_ln(n1, i - 2) + log(n3 - n1) + tail)
                  + math.exp(binomial_ln(n1, i - 3) + log(n2 - n1) + log(n3 - n2) + tail)
                  + math.exp(binomial_ln(n1, i - 3) + log(n2 - n1) - math.log(2.0) + log(n2 - n1 - 1.0) + tail))
        return p
    sbdry[0] = n_perm - int(n_perm * eta)
    eta0 = eta
    at = 0
    for ones in range(2, max_ones + 1):
        table = [hypergeom.cdf(k, n_perm, ones, draws) for k in range(ones + 1)]
        hi = eta0 * 1.1
        eta_boundary(hi, ones, at + 1, table); p_hi = p_exceed(ones, at + 1)
        lo = eta0 * 0.25
        eta_boundary(lo, ones, at + 1, table); p_lo = p_exceed(ones, at + 1)
        while (hi - lo) / lo > tol:
            eta0 = lo + (hi - lo) * (eta - p_lo) / (p_hi - p_lo)
            eta_boundary(eta0, ones, at + 1, table); p = p_exceed(ones, at + 1)
            if p > eta:
                hi, p_hi = eta0, p
            else:
                lo, p_lo = eta0, p
        at += ones
    return sbdry


@pytest.mark.parametrize("n_perm,alpha", [(200, 0.01), (500, 0.01), (1000, 0.01), (400, 0.05)])
def test_cbs_boundary_two_restatements(n_perm, alpha):
    assert O.cbs_boundary(n_perm, alpha).tolist() == py_cbs_boundary(n_perm, alpha)


# ---------------------------------------------------------------------------------------------------------------------------------------
# ChangePoint.TrimmedVariance / InflationFactor (ChangePoint.cs:423-470; Helper.Seq :305-314): the SD estimate behind SDUndo, with scipy's
# normal quantile and density in place of MathNet's — 1e-10 relative
def py_trimmed_variance(per_chr, trim=0.025):
    from scipy.stats import norm
    n = sum(len(c) for c in per_chr)
    diff = [0.0] * (n - 1)                                             # slots the loop does not reach stay 0 and are sorted with the rest
    i, last = 0, float("nan")
    for c in per_chr:
        if len(c) == 0:
            continue
        if i > 0:
            diff[i] = float(c[0]) - last
            i += 1
        for a, b in zip(c[:-1], c[1:]):
            diff[i] = float(b) - float(a)
            i += 1
        last = float(c[-1])
    keep = int(np.rint((1 - 2 * trim) * (n - 1)))
    kept = sorted(abs(d) for d in diff)[:keep]
    total = 0.0
    for d in kept:
        total += d ** 2
    a = norm.ppf(1 - trim)
    step = 2 * a / 10000
    lo, hi = -a + step / 2, a - step / 2
    inc = (hi - lo) / 9999
    xs = [lo]
    for _ in range(9998):
        xs.append(xs[-1] + inc)
    xs.append(hi)
    ex2 = 0.0
    for v in xs:
        ex2 += (v * v) * norm.pdf(v)
    ex2 = ex2 * step / (1 - 2 * trim)
    return (1 / ex2) * total / (2 * keep)


def test_trimmed_variance_two_restatements():
    import ctypes as C
    rng = np.random.RandomState(66)
    for it in range(4):
        per_chr = [np.round(rng.normal(0, rng.uniform(0.5, 3), int(rng.randint(2, 400))), 2) for _ in range(int(rng.randint(1, 5)))]
        if it == 3:
            per_chr.insert(0, np.array([1.25]))                         # a one-bin first chromosome: no junction difference is formed after it
        arrs = [np.ascontiguousarray(c, np.float64) for c in per_chr]
        lens = np.array([len(c) for c in arrs], np.int32)
        got = O.lib.orc_trimmed_variance(len(arrs), O._pp(arrs), O._p(lens), C.c_double(0.025))
        want = py_trimmed_variance(per_chr)
        assert abs(got - want) <= 1e-10 * want, (it, got, want)


def test_clean_weighted_median_two_restatements():
    from canvas_amd import CLEAN_GCNORM, CLEAN_FILTSIZE, CLEAN_OUTLIERS
    rng = np.random.RandomState(4040)
    weighted_used = 0
    for it in range(6):
        nchr = 3
        is_auto = np.array([1, 1, 0], np.uint8)
        bins = []
        for c in range(nchr):
            pos = 0
            for _ in range(int(rng.randint(300, 900))):
                gc = int(np.clip(rng.normal(45, 7), 0, 100))
                cnt = np.float32(rng.poisson(60.0 * (1 + (gc - 45) * 0.015)))
                bins.append([c, pos, pos + 100, cnt, gc])
                pos += 100
        w = int(rng.choice([5, 20, 40]))
        want = py_clean(bins, is_auto, min_bins_weighted=w)
        a = [np.array([b[k] for b in bins]) for k in range(5)]
        got = O.clean(a[0], a[1], a[2], a[3], a[4], is_auto, np.zeros(nchr, np.uint8), CLEAN_GCNORM | CLEAN_FILTSIZE | CLEAN_OUTLIERS, min_bins_weighted=w)
        assert len(got["chr"]) == len(want), it
        assert (got["count"].view(np.uint32) == np.array([b[3] for b in want], np.float32).view(np.uint32)).all(), it
        per_gc = np.bincount([b[4] for b in want if is_auto[b[0]]], minlength=101)
        weighted_used += int(((per_gc > 0) & (per_gc < 100)).sum())
    assert weighted_used > 20


# RawRatioCalculator.Run (RawRatioCalculator.cs:23-44) followed by RatiosToCounts
def test_normalize_raw_ratio_two_restatements():
    rng = np.random.RandomState(32)
    for it in range(10):
        n = int(rng.randint(1, 300))
        sample = np.round(rng.gamma(4, 25, n), 2).astype(np.float32)
        reference = np.round(rng.gamma(2, 30, n), 2).astype(np.float32) * (rng.rand(n) > 0.1)
        lo, hi = float(rng.choice([1.0, 5.0])), float(rng.choice([np.inf, 150.0]))
        keep, ratios = [], []
        for j in range(n):
            if float(reference[j]) < lo or float(reference[j]) > hi:
                continue
            keep.append(j)
            ratios.append(np.float32(sample[j] / reference[j]))
        counts = [np.float32(float(r) * (40.0 * 2 / 2.0)) for r in ratios]
        k, r, c = O.norm_ratio(sample, reference, None, mode=1, min_ref=lo, max_ref=hi)
        assert k.tolist() == keep, it
        assert (r.view(np.uint32) == np.asarray(ratios, np.float32).view(np.uint32)).all() and (c.view(np.uint32) == np.asarray(counts, np.float32).view(np.uint32)).all(), it


# ---- CanvasPartition -p: PloidyInfo.IsUniformReferencePloidy + the reference-ploidy branch of IsNewSegment (SegmentationResultsProcessor.cs:117-128),
# read a second time with inclusive one-based overlaps [max(qs, a), min(qe, b)] instead of the C#'s zero-based overlapStart / overlapEnd pair
def py_is_uniform(q_start, q_end, ivs):
    counts = [0] * 5
    counts[2] = q_end - q_start + 1
    for a, b, cn in ivs:                                  # one-based inclusive [a, b]
        if cn == 2:
            continue
        # overlapStart = max(qs - 1, a - 1); "if overlapStart > End continue"; overlapEnd = min(qe, b); bases = overlapEnd - overlapStart
        lo = max(q_start, a); hi = min(q_end, b)
        bases = hi - lo + 1
        if max(q_start - 1, a - 1) > b or bases <= 0:
            continue
        counts[2] -= bases; counts[cn] += bases
    return sum(1 for v in counts if v > 0) < 2


def py_postprocess_ploidy(bin_start, bin_end, seg_starts, ploidy, max_dist):
    seg_num = -1; out = []
    for c in range(len(bin_start)):
        starts = set(int(v) for v in seg_starts[c]); prev_end = 0; ids = []
        for s, e in zip(bin_start[c], bin_end[c]):
            s, e = int(s), int(e)
            new = s in starts
            if prev_end > 0 and max_dist >= 0 and prev_end + max_dist < s and not new:
                new = True
            if not new and ploidy[c] is not None and not py_is_uniform(prev_end if prev_end > 0 else 1, e, list(zip(*[[int(v) for v in col] for col in ploidy[c]]))):
                new = True
            if new:
                seg_num += 1
            ids.append(seg_num); prev_end = e
        out.append(np.array(ids, np.int32))
    return out, seg_num


def test_reference_ploidy_postprocess_two_restatements():
    rng = np.random.RandomState(20260928)
    for trial in range(30):
        nchr = 4
        bs, be, segs, ploidy = [], [], [], []
        for c in range(nchr):
            nb = int(rng.randint(5, 120))
            gaps = rng.randint(0, 40, nb); sizes = rng.randint(50, 400, nb)
            st = np.cumsum(gaps + np.concatenate([[0], sizes[:-1]])) + 100
            en = st + sizes
            bs.append(st.astype(np.uint32)); be.append(en.astype(np.uint32))
            k = int(rng.randint(0, 5))
            segs.append(np.sort(rng.choice(st, k, replace=False)).astype(np.uint32) if k else np.zeros(0, np.uint32))
            if c == 0 and trial % 3 == 0:
                ploidy.append(None)                        # chromosome absent from the VCF
            else:
                m = int(rng.randint(0, 5)); a = rng.randint(1, int(en[-1]) + 50, m); b = a + rng.randint(-5, int(en[-1]) // 2 + 1, m)
                ploidy.append((a.astype(np.int32), b.astype(np.int32), rng.choice([0, 1, 2, 3, 4], m).astype(np.int32)))
        exp, last = py_postprocess_ploidy(bs, be, segs, ploidy, 1000000)
        got, last_o = O.postprocess_ploidy(bs, be, segs, None, ploidy, 1000000)
        assert last == last_o
        for c in range(nchr):
            assert (got[c] == exp[c]).all(), (trial, c)
    # the query interval itself: a record ending exactly on the previous bin's end / starting on this bin's end
    assert O.is_uniform_reference_ploidy(100, 200, ([201], [300], [1])) == 1
    assert O.is_uniform_reference_ploidy(100, 200, ([200], [300], [1])) == 0
    assert O.is_uniform_reference_ploidy(100, 200, ([1], [99], [1])) == 1
    assert O.is_uniform_reference_ploidy(100, 200, ([1], [100], [1])) == 0
    assert O.is_uniform_reference_ploidy(100, 200, ([1], [1000], [1])) == 1      # all haploid
    assert O.is_uniform_reference_ploidy(100, 200, ([1], [1000], [7])) == -1     # baseCounts[7]: IndexOutOfRangeException in the C#


# ---- GetEvennessScore (Segmentation.cs:260-296): written from the formula in the paper the C# cites — per window, with S = sum(x) and
# a = mean(x): score = sum_{k=0..floor(a)} #{x >= k} / S — on numpy, sequential sums kept (np.cumsum adds left to right like LINQ's Sum)
def py_evenness(per_chr, window):
    def scores(w):
        out = []
        for x in per_chr:
            for index in range(0, max(0, len(x) - w), w):
                if not index < len(x) - w:
                    break
                t = x[index:index + w - 1]
                s = float(np.cumsum(t)[-1]); avg = s / len(t)
                if not avg >= 0:
                    out.append(0.0); continue
                acc = 0.0
                with np.errstate(divide="ignore", invalid="ignore"):
                    for k in range(0, int(math.floor(avg)) + 1):
                        acc = acc + np.float64(int(np.count_nonzero(t >= k))) / np.float64(s)
                if math.isfinite(acc):
                    out.append(float(acc))
        return out
    iqr = scores(10000); med = scores(window)
    if len(iqr) < 2 or not med:
        return None
    q1, _, q3 = py_quartiles([np.float32(v) for v in iqr])
    m = sorted(med); n = len(m)
    median = m[n // 2] if n % 2 else (m[n // 2 - 1] + m[n // 2]) / 2
    return float(np.float32(q3)) * 100.0 if float(np.float32(q3) - np.float32(q1)) > 0.015 else median * 100.0


def test_evenness_score_two_restatements():
    rng = np.random.RandomState(77)
    for trial, (sizes, window) in enumerate([((25_000, 31_000, 900), 4000), ((12_000, 10_001, 10_000), 700), ((45_000,), 1000), ((9_000, 8_000), 500), ((33_000, 21_000), 40_000)]):
        per = []
        for n in sizes:
            x = np.round(rng.gamma(25, 4, n), 2)
            if trial == 0: x[5_000:16_000] = 0.0                      # windows whose sum is 0
            if trial == 1: x[100:400] = -2.0
            if trial == 2: x[20_000:] *= 3
            per.append(np.ascontiguousarray(x))
        exp = py_evenness(per, window)
        got = O.evenness_score(per, window)
        assert (exp is None) == (got is None), (trial, exp, got)
        if exp is not None:
            assert np.float64(exp).tobytes() == np.float64(got).tobytes(), (trial, exp, got)
    assert O.evenness_score([np.ones(9_000)], 500) is None


# ---------------------------------------------------------------------------------------------------------------------------------------
# TMaxP is NOT always the maximum over every admissible arc.  For a pair of blocks the C# scans the arc lengths alenlo .. min(alen, n - alen)
# and n - min(alen, n - alen) .. alenhi, alen = the distance between the pair's extremes (CBSTStatistic.cs:860-934; TMaxO has the same loop,
# :233-326): the lengths in between cannot beat the arc between the extremes — unless that arc is shorter than the minimum width, in which
# case nothing of the pair is scanned although it holds admissible arcs.  With blocks of a dozen elements that changes the value now and
# then (a randomised device soak found it; the device kernel for segments <= 200 bins had taken the exhaustive maximum).  This restatement
# follows the scanned ranges literally; the oracle must agree with IT on every case, and on some cases both are below the every-arc value.
def _round_half_even(v):
    return int(np.rint(v))


def _scanned_arc_statistic(px, tss, al0):
    n = len(px); rn = float(n)
    nb = _round_half_even(np.sqrt(rn)) if n >= 50 else 1
    bb = [_round_half_even(rn * ((i + 1.0) / nb)) for i in range(nb)]                # 1-based last position of every block
    sx = np.cumsum(px)
    lo_pos = [1 if k == 0 else bb[k - 1] + 1 for k in range(nb)]
    bmn, bmx, imn, imx = [], [], [], []
    for k in range(nb):
        seg = sx[lo_pos[k] - 1:bb[k]]
        bmn.append(float(seg.min())); bmx.append(float(seg.max()))
        imn.append(lo_pos[k] + int(np.argmin(seg))); imx.append(lo_pos[k] + int(np.argmax(seg)))      # first occurrence
    best = _max_min_seed(px)
    nal0 = n - al0
    for bi in range(nb):
        for bj in range(bi, nb):
            ilo, ihi, jlo, jhi = lo_pos[bi], bb[bi], lo_pos[bj], bb[bj]
            alenhi = min(jhi - ilo, nal0)
            alenlo = max(1 if bi == bj else jlo - ihi, al0)
            s1, s2 = abs(bmx[bj] - bmn[bi]), abs(bmx[bi] - bmn[bj])
            alen = abs(imx[bj] - imn[bi]) if s1 > s2 else abs(imn[bj] - imx[bi])
            amax = min(alen, n - alen)
            lengths = []
            if alenlo <= rn / 2 and alenlo <= amax:
                lengths += list(range(alenlo, amax + 1))
            if alenhi >= rn / 2 and alenhi >= n - amax:
                lengths += list(range(n - amax, alenhi + 1))
            for L in lengths:
                p0 = max(ilo, jlo - L); p1 = min(ihi, jhi - L)                          # 1-based start positions with the end inside block bj
                if p1 < p0:
                    continue
                d = np.abs(sx[p0 + L - 1:p1 + L] - sx[p0 - 1:p1]).max()
                best = max(best, rn / (L * (rn - L)) * d * d)
    t = tss
    if t <= best + 0.0001:
        t = best + 1.0
    return best / ((t - best) / (rn - 2.0))


def test_tmaxp_follows_the_scanned_ranges_not_every_arc():
    rng = np.random.RandomState(4242)
    below = 0
    for it in range(3000):
        n = int(rng.randint(8, 201))
        kind = it % 4
        if kind == 0:
            x = rng.normal(0, 1, n)
        elif kind == 1:
            x = np.round(rng.normal(0, 1, n), 1)                       # exact ties
        elif kind == 2:
            x = rng.normal(0, 1, n); x[:n // 2] += rng.choice([1.0, 3.0, 6.0]); x = rng.permutation(x)      # a permuted segment with a change
        else:
            x = rng.standard_cauchy(n)                                 # spikes: a block's extremes next to each other — where the scanned ranges leave arcs out
        x -= x.mean()
        tss = float(np.sum(x * x))
        if tss == 0:
            continue
        got = O.tmaxp(x, tss, 2)
        want = _scanned_arc_statistic(x, tss, 2)
        assert abs(got - want) <= 1e-11 * max(1.0, want), (it, n, got, want)
        assert abs(O.tmaxo(x, 2)[0] - want) <= 1e-11 * max(1.0, want), (it, n)          # TMaxO: the same search on the observed data
        every = _every_arc_statistic(x, tss, n - 2, 2, True)
        assert got <= every * (1 + 1e-11)
        if got < every * (1 - 1e-9):
            below += 1
    assert below > 10                                                  # the two differ on spiky inputs: the exhaustive maximum is not the reference's
