"""PerSampleHMM (Viterbi) on the GPU vs the CPU oracle: state paths and segment ids bit-identical."""
import numpy as np
import pytest

import oracle_lib as O
from canvas_amd import synth
from gpu_common import get_canvas, to_dev

pytestmark = pytest.mark.gpu


def _coverage(seed, n, nchr=24):
    bins = synth.generate_bins(seed, n, nchr=nchr)
    # what CanvasPartition reads: the F2-rounded text of CanvasClean's float counts, parsed as double (CanvasSegment.cs:1146)
    cov = np.round(bins["count"].astype(np.float64), 2)
    off = np.concatenate([[0], np.cumsum(np.bincount(bins["chr"], minlength=nchr))]).astype(np.int64)
    return bins, cov, off


def _check(cv, bins, cov, off, max_dist=1000000):
    nchr = len(off) - 1
    per = [np.ascontiguousarray(cov[off[c]:off[c + 1]]) for c in range(nchr)]
    paths, ran = O.hmm_genome_per_sample(per, threads=8)
    dcov = to_dev(cov, cv.device)
    state = cv.hmm_per_sample(dcov, off)
    got = state.cpu().numpy()
    for c in range(nchr):
        exp = paths[c] if ran[c] else np.full(len(per[c]), -1, np.int32)
        g = got[off[c]:off[c + 1]]
        assert (g == exp).all(), (c, int((g != exp).sum()), len(exp))
    # segment ids
    bs = [bins["start"][off[c]:off[c + 1]].astype(np.uint32) for c in range(nchr)]
    be = [bins["stop"][off[c]:off[c + 1]].astype(np.uint32) for c in range(nchr)]
    segstarts = [O.segments_from_path(paths[c], ran[c], bs[c], be[c])[0] for c in range(nchr)]
    ids, last = O.postprocess(bs, be, segstarts, None, max_dist)
    seg, nseg = cv.segment_ids(off, state, to_dev(bins["start"], cv.device), to_dev(bins["stop"], cv.device), max_dist)
    gseg = seg.cpu().numpy()
    assert (gseg == np.concatenate(ids)).all()
    assert nseg == last + 1
    return got


@pytest.mark.parametrize("n,nchr", [(30_000, 24), (5_000, 3), (200_000, 24)])
def test_viterbi_matches_oracle(n, nchr):
    cv = get_canvas()
    bins, cov, off = _coverage(20260927 + 10, n, nchr)
    got = _check(cv, bins, cov, off)
    if n >= 200_000:
        assert len(np.unique(got)) >= 2   # planted CN segments are found


def test_viterbi_short_and_skipped_chromosomes():
    cv = get_canvas()
    bins, cov, off = _coverage(20260927 + 11, 4_000, 6)
    # carve chromosome runs of 11, 10 (skipped) and 1 bins out of the tail
    chr_id = bins["chr"].copy()
    n = len(chr_id)
    chr_id[n - 22:n - 11] = 6; chr_id[n - 11:n - 1] = 7; chr_id[n - 1:] = 8
    bins["chr"] = chr_id
    off = np.concatenate([[0], np.cumsum(np.bincount(chr_id, minlength=9))]).astype(np.int64)
    got = _check(cv, bins, cov, off, max_dist=5000)
    assert (got[off[7]:off[8]] == -1).all() and (got[off[8]:] == -1).all()


def test_viterbi_saturated_and_zero_coverage():
    cv = get_canvas()
    bins, cov, off = _coverage(20260927 + 12, 20_000, 4)
    cov[100:400] = 0.0          # homozygous deletion
    cov[1000:1300] = 5000.0     # far above 5 x haploid mean: capped (HiddenMarkovModelsRunner.cs:154-162)
    cov[2000] = 124.5; cov[2001] = 125.5   # half-way cases of Convert.ToInt32
    _check(cv, bins, cov, off)


def test_speculation_is_verified_and_falls_back(monkeypatch):
    """A corrupted back-pointer guess must be caught by k_vit_verify; the chromosome is then recomputed sequentially and the
    result is still identical to the oracle.  Also: the purely sequential path (CANVAS_HMM_SEQUENTIAL) gives the same states."""
    cv = get_canvas()
    cv.profile_enable(True)
    bins, cov, off = _coverage(20260927 + 13, 60_000, 6)
    cv.profile_get("viterbi_sequential", reset=True)
    base = _check(cv, bins, cov, off)
    assert cv.profile_get("viterbi_sequential")[1] == 0          # speculation verified everywhere
    for mode in ("chain", "scan"):           # the wave-sequential backbone / the single-kernel parity scan instead of the predicted pieces
        monkeypatch.setenv("CANVAS_HMM_BACKBONE", mode)
        got0 = _check(cv, bins, cov, off)
        assert cv.profile_get("viterbi_sequential")[1] == 0 and (got0 == base).all()
    monkeypatch.delenv("CANVAS_HMM_BACKBONE")
    monkeypatch.setenv("CANVAS_HMM_TEST_CORRUPT", "1")
    got = _check(cv, bins, cov, off)
    assert cv.profile_get("viterbi_sequential")[1] == 1          # the corrupted chromosome was recomputed
    assert (got == base).all()
    monkeypatch.delenv("CANVAS_HMM_TEST_CORRUPT")
    monkeypatch.setenv("CANVAS_HMM_SEQUENTIAL", "1")
    got2 = _check(cv, bins, cov, off)
    assert (got2 == base).all()


def test_segment_ids_with_forbidden_intervals():
    """SegmentationResultsProcessor.PostProcessSegments with the -b BED intervals (forbidden-zone midpoint rule)"""
    cv = get_canvas()
    bins, cov, off = _coverage(20260927 + 14, 40_000, 5)
    nchr = 5
    per = [np.ascontiguousarray(cov[off[c]:off[c + 1]]) for c in range(nchr)]
    paths, ran = O.hmm_genome_per_sample(per, threads=4)
    rng = np.random.RandomState(3)
    excl = []
    for c in range(nchr):
        s = bins["start"][off[c]:off[c + 1]]; e = bins["stop"][off[c]:off[c + 1]]
        pick = np.sort(rng.choice(len(s) - 2, 25, replace=False))
        # intervals in the gaps / over bins, sorted by end
        st = e[pick] - rng.randint(0, 400, 25); en = st + rng.randint(10, 900, 25)
        order = np.argsort(en, kind="stable")
        excl.append((st[order].astype(np.int32), en[order].astype(np.int32)))
    bs = [bins["start"][off[c]:off[c + 1]].astype(np.uint32) for c in range(nchr)]
    be = [bins["stop"][off[c]:off[c + 1]].astype(np.uint32) for c in range(nchr)]
    segstarts = [O.segments_from_path(paths[c], ran[c], bs[c], be[c])[0] for c in range(nchr)]
    ids, last = O.postprocess(bs, be, segstarts, excl, 1000000)
    state = to_dev(np.concatenate(paths), cv.device)
    seg, nseg = cv.segment_ids(off, state, to_dev(bins["start"], cv.device), to_dev(bins["stop"], cv.device), 1000000, excluded=excl)
    assert (seg.cpu().numpy() == np.concatenate(ids)).all()
    ids0, _ = O.postprocess(bs, be, segstarts, None, 1000000)
    assert last > _      # the forbidden zones added splits


@pytest.mark.parametrize("nsamples,n", [(3, 30_000), (1, 12_000), (2, 20_000), (5, 9_000)])
def test_joint_hmm_matches_oracle(nsamples, n):
    """-m HMM (joint mode): per-chromosome emission parameters, genotype combinations over the samples (only the first four samples enter
    the combinations, DistributionUtilities.cs:14-16), grouped 0/1 and 3/4 probabilities; chromosomes with <= 10 bins are skipped"""
    cv = get_canvas()
    cv.profile_enable(True)
    nchr = 5
    bins, cov0, off = _coverage(20260927 + 30 + nsamples, n, nchr)
    rng = np.random.RandomState(nsamples)
    covs = [cov0]
    for s in range(1, nsamples):
        scale = [1.0, 0.8, 1.3, 0.6, 1.1][s]
        c = np.round(cov0 * scale + rng.normal(0, 4, len(cov0)), 2).clip(0)
        if s == 1: c[off[1] + 200:off[1] + 600] *= 0.5            # a deletion only sample 1 carries
        covs.append(np.ascontiguousarray(c))
    covs[0] = covs[0].copy(); covs[0][off[2] + 100:off[2] + 500] *= 1.5     # a gain only sample 0 carries
    # products of several pmf values underflow to 0 for outlying bins, Math.Log gives -inf there: such chromosomes are (correctly)
    # handed to the sequential kernel, the others go through speculate / verify — the result must be the oracle's either way
    got = cv.hmm_joint([to_dev(c, cv.device) for c in covs], off).cpu().numpy()
    for c in range(nchr):
        ran, path = O.hmm_chromosome([np.ascontiguousarray(x[off[c]:off[c + 1]]) for x in covs], per_sample=False)
        g = got[off[c]:off[c + 1]]
        if not ran: assert (g == -1).all()
        else: assert (g == path).all(), (c, np.nonzero(g != path)[0][:10])
    assert len(np.unique(got)) >= 2


@pytest.mark.parametrize("case", ["radix_forced", "not_f2_text", "quartiles_outside_the_window", "negative_zero"])
def test_quartiles_by_counting_fall_back_to_the_radix_select(case, monkeypatch):
    """The genome-wide quartiles are read off per-value counts when the coverage is F2 text (k / 100); anything else — a value that is not of that form, a -0.0,
    quartiles further than the counting window from the sample's level — must end in the radix select with the same states as the oracle"""
    cv = get_canvas()
    bins, cov, off = _coverage(20260927 + 140, 60_000, 24)
    if case == "radix_forced":
        monkeypatch.setenv("CANVAS_HMM_RADIX_SELECT", "1")
    elif case == "not_f2_text":
        cov = cov + 1e-7 * np.arange(len(cov)) / len(cov)
    elif case == "quartiles_outside_the_window":
        cov = cov.copy(); cov[::2] = np.round(cov[::2] * 3 + 500, 2)            # bimodal: the upper quartile sits hundreds of units above the median
    elif case == "negative_zero":
        cov = cov.copy(); cov[5] = -0.0
    _check(cv, bins, cov, off)


def test_sample_whose_median_coverage_is_zero_takes_the_sequential_kernel():
    """The one kind of PerSampleHMM call the speculative attempts give up on, caught in a noise-40 soak and kept as a fixture (tests/golden/viterbi_fallback_median0.npz: the 413
    bins CanvasClean left of a 600 000-bin sample whose counts were clipped noise; seed 1531490205 of tools/soak.py): three quarters of the coverage are 0, so the sample's
    median — the model's haploid mean — is 0, every state's emission probability is 0 for every bin with coverage, and the reference's best state is -1 at every position
    (HMM.cs:100-111).  The speculative passes compare log-likelihoods that are all -inf: every attempt fails its exact comparison (CANVAS_HMM_DEBUG_FAIL: why = 0x19) and the
    sequential kernel reproduces the reference's all -1 path."""
    import os, torch
    cv = get_canvas()
    cv.profile_enable(True)
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "viterbi_fallback_median0.npz"))
    cov, off = np.ascontiguousarray(d["cov"], np.float64), np.asarray(d["off"], np.int64)
    assert np.median(cov) == 0.0 and cov.max() > 100.0
    per = [np.ascontiguousarray(cov[off[c]:off[c + 1]]) for c in range(len(off) - 1)]
    paths, ran = O.hmm_genome_per_sample(per, threads=1)
    assert ran[0] == 1 and (paths[0] == -1).all()
    cv.profile_get("viterbi_sequential", reset=True)
    st = cv.hmm_per_sample(torch.from_numpy(cov).to(cv.device), off).cpu().numpy()
    assert (st == paths[0]).all()
    assert cv.profile_get("viterbi_sequential")[1] == 1
