"""CanvasPartition -m Wavelets on the GPU (canvas_wavelets) against the oracle, which the reference's own known-answer test pins
(tests/test_oracle_golden.py::test_wavelets_known_answer).  Breakpoints are integers: identical or wrong."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from gpu_common import get_canvas, to_dev

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _run(cv, per_chr, **kw):
    cov = np.ascontiguousarray(np.concatenate(per_chr), np.float64)
    off = np.concatenate([[0], np.cumsum([len(a) for a in per_chr])]).astype(np.int64)
    return cv.wavelets(to_dev(cov, cv.device), off, **kw)


def _coverage(rng, n, mean=100.0, events=6, wave=0.0):
    x = rng.poisson(mean, n).astype(np.float64)
    for _ in range(events if n > 40 else 0):
        a = int(rng.randint(0, n - 20)); b = min(n, a + int(rng.choice([12, 40, 300, 2500, n // 4 + 1])))
        x[a:b] = np.round(x[a:b] * float(rng.choice([0.0, 0.5, 1.5, 2.0])))
    if wave:
        x = np.round(x * (1 + wave * np.sin(np.arange(n) / 700.0)))
    return np.round(x * 100) / 100      # what the cleaned file holds: F2 text


def test_reference_known_answer_on_device():
    cv = get_canvas()
    d = json.load(open(os.path.join(G, "wavelets_minimal.json")))
    cov = np.array(d["coverage"])
    got = _run(cv, [cov], is_germline=d["is_germline"], threshold_lower=d["threshold_lower"], threshold_upper=d["threshold_upper"], mad_factor=d["mad_factor"],
               window=d["variability_window"])
    assert got[0].tolist() == d["expected_breakpoints"]


@pytest.mark.parametrize("germline", [False, True])
@pytest.mark.parametrize("seed,lengths,window", [(1, [3000, 11, 10, 800], 100), (2, [40_000, 9_001], 1000), (3, [120_000], 100000), (4, [257, 256, 258, 1025], 11),
                                                 (5, [60_000, 30_000, 5], 20000)])
def test_wavelets_match_oracle(seed, lengths, window, germline):
    cv = get_canvas()
    rng = np.random.RandomState(seed)
    per = [_coverage(rng, n, mean=float(rng.choice([30, 100, 400])), wave=0.05 * (seed % 2)) for n in lengths]
    exp = O.wavelets_genome(per, is_germline=germline, window=window)
    got = _run(cv, per, is_germline=germline, window=window)
    assert len(got) == len(exp)
    for c in range(len(per)):
        assert got[c].tolist() == exp[c].tolist(), (c, lengths[c])
    st = cv.wavelets_stats()
    assert st[0] > 0 and st[1] == 0          # the shortcut division never disagreed with the IEEE one
    assert sum(len(e) for e in exp) > len([n for n in lengths if n > 10])   # something beyond the chromosome starts was found


def test_wavelets_flat_and_degenerate_input():
    cv = get_canvas()
    per = [np.full(500, 100.0), np.zeros(300), np.arange(400, dtype=np.float64)]
    for germline in (False, True):
        exp = O.wavelets_genome(per, is_germline=germline, window=50)
        got = _run(cv, per, is_germline=germline, window=50)
        for c in range(3):
            assert got[c].tolist() == exp[c].tolist()
    # non-finite coverage is refused, not silently segmented
    bad = np.full(100, 50.0); bad[7] = np.nan
    with pytest.raises(Exception):
        _run(cv, [bad], window=11)


def test_exact_chain_fallback_gives_the_same_tree(monkeypatch):
    """CANVAS_WV_TEST_EXACT=1 recomputes every long node with IEEE divisions in the chain (the path taken when a checkpoint of the shortcut
    chain is not reproduced): same breakpoints, and the run is not counted as a disagreement."""
    cv = get_canvas()
    rng = np.random.RandomState(11)
    per = [_coverage(rng, 50_000, wave=0.05), _coverage(rng, 3_000)]
    exp = O.wavelets_genome(per, is_germline=True, window=5000)
    monkeypatch.setenv("CANVAS_WV_TEST_EXACT", "1")
    got = _run(cv, per, is_germline=True, window=5000)
    assert [g.tolist() for g in got] == [e.tolist() for e in exp]
    assert cv.wavelets_stats()[1] == 0


def test_nan_threshold_keeps_every_coefficient():
    """Most windows have median 0, so MAD / median is NaN, the coverage variability is NaN and so is the threshold: HardThresh's
    '<=' is then never true and no coefficient is zeroed (found by tools/soak_wavelets.py)."""
    cv = get_canvas()
    rng = np.random.RandomState(5)
    x = np.zeros(1200); x[800:] = rng.poisson(40, 400)
    assert np.isnan(O.coverage_variability(100, [x]))
    for germline in (False, True):
        exp = O.wavelets_genome([x], is_germline=germline, window=100)
        got = _run(cv, [x], is_germline=germline, window=100)
        assert got[0].tolist() == exp[0].tolist() and len(exp[0]) > 20


def test_closed_form_decisions_agree_with_the_chains(monkeypatch):
    """Default path: the arg-max of every long node is decided from exact prefix sums + the rounding-error bound of the reference's recurrences (canvas_wavelets_decisions);
    exact ties (a flat chromosome) cannot be decided and go through the chain.  CANVAS_WV_CHAIN_ONLY=1 runs every long node through the chain: same breakpoints.  A coverage
    that is not made of two-decimal values (no exact integer sums) takes the chains by itself."""
    cv = get_canvas()
    rng = np.random.RandomState(21)
    per = [_coverage(rng, 90_000, wave=0.05), _coverage(rng, 20_000), np.full(3000, 77.0)]
    exp = O.wavelets_genome(per, window=5000)
    got = _run(cv, per, window=5000)
    dec = cv.wavelets_decisions()
    assert [g.tolist() for g in got] == [e.tolist() for e in exp]
    assert dec[3] == 1 and dec[0] > 50 and dec[1] >= 1 and dec[2] >= 1, dec
    monkeypatch.setenv("CANVAS_WV_CHAIN_ONLY", "1")
    got2 = _run(cv, per, window=5000)
    assert [g.tolist() for g in got2] == [e.tolist() for e in exp] and cv.wavelets_decisions()[3] == 0
    monkeypatch.delenv("CANVAS_WV_CHAIN_ONLY")
    per3 = [per[0] * 1.0000001, per[1]]
    exp3 = O.wavelets_genome(per3, window=5000)
    got3 = _run(cv, per3, window=5000)
    assert [g.tolist() for g in got3] == [e.tolist() for e in exp3] and cv.wavelets_decisions()[3] == 0


def test_factor_of_three_statistics_on_the_device_equal_the_host_thread(monkeypatch):
    """SegmentationInput.FactorOfThreeCoverageVariabilities runs on the device (triplet medians + a radix selection of the median ratio per exponent); CANVAS_WV_F3_CHECK=1
    makes the call compute the host version as well and fail on any difference (NaN ratios — triplets with median 0 — sort in front of every number, as in .NET)."""
    cv = get_canvas()
    monkeypatch.setenv("CANVAS_WV_F3_CHECK", "1")
    monkeypatch.setenv("CANVAS_WV_VAR_CHECK", "1")           # the same for the per-window MAD / median of GetCoverageVariability (one workgroup per window, radix selections)
    rng = np.random.RandomState(77)
    zeros = _coverage(rng, 30_000); zeros[5000:21000] = 0.0           # most triplets have median 0: NaN and infinite ratios
    for per in ([_coverage(rng, 120_000, wave=0.05), _coverage(rng, 7_001), _coverage(rng, 200)], [zeros, _coverage(rng, 9_000)], [_coverage(rng, 160)]):
        for window in (5000, 12000, 999):                   # 12000: windows of 10 000 first (Segmentation.cs:314-322), of 12 000 if their spread is small
            exp = O.wavelets_genome(per, window=window)
            got = _run(cv, per, window=window)
            assert [g.tolist() for g in got] == [e.tolist() for e in exp]


def test_tile_edges_of_the_prefix_sums_and_the_stretch_medians(monkeypatch):
    """Round 6: the prefix sums run over tiles of 1 024 bins (carries from the tiles in front, more than 1 024 of them for a chromosome above a million bins) and the
    medians of the chromosomes / healing stretches over tiles of 4 096 integers of all stretches at once: chromosome lengths on and around the tile sizes, a chromosome
    of 1.2 M bins, chromosomes below MinSize between them; the host-comparison hook checks every device median against the host's."""
    monkeypatch.setenv("CANVAS_TEST_HOOKS", "1"); monkeypatch.setenv("CANVAS_WV_VAR_CHECK", "1")
    cv = get_canvas()
    rng = np.random.RandomState(66)
    lengths = [1_200_000, 1025, 1024, 1, 4096, 2, 4097, 5000, 8192, 9]
    per = [_coverage(rng, n, mean=100.0, wave=0.03) for n in lengths]
    exp = O.wavelets_genome(per, is_germline=True, window=1000)
    got = _run(cv, per, is_germline=True, window=1000)
    for c in range(len(per)):
        assert got[c].tolist() == exp[c].tolist(), (c, lengths[c])
    # the same list through the per-workgroup medians (the path of a coverage that is not two-decimal text)
    monkeypatch.setenv("CANVAS_WV_MEDIAN_PER_WG", "1")
    got2 = _run(cv, per, is_germline=True, window=1000)
    for c in range(len(per)):
        assert got2[c].tolist() == exp[c].tolist(), (c, lengths[c])


def test_a_coverage_without_two_decimal_values_takes_the_other_medians():
    """x = k / 100 fails for some bins: no integers, so the chromosome medians come from the per-workgroup kernel after the first synchronisation and every long node
    takes the chain; the result is still the oracle's."""
    cv = get_canvas()
    rng = np.random.RandomState(67)
    per = [_coverage(rng, n) + 0.001 * (np.arange(n) % 7 == 0) for n in (30_000, 4_000)]
    exp = O.wavelets_genome(per, is_germline=False, window=1000)
    got = _run(cv, per, is_germline=False, window=1000)
    for c in range(2):
        assert got[c].tolist() == exp[c].tolist()
