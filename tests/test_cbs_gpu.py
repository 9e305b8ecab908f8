"""CBS (host recursion + GPU exhaustive arc search) vs the CPU oracle: segment lengths identical."""
import numpy as np
import pytest

import oracle_lib as O
from canvas_amd import synth
from gpu_common import get_canvas, to_dev

pytestmark = pytest.mark.gpu


def _run(cv, cov, off, nperm=10000, undo=0):
    nchr = len(off) - 1
    per = [np.ascontiguousarray(cov[off[c]:off[c + 1]]) for c in range(nchr)]
    exp, est = O.cbs_genome(per, 0.01, nperm, threads=8, undo=undo)
    seg_len, nseg, stats = cv.cbs(to_dev(cov, cv.device), off, 0.01, nperm, undo=undo)
    got = seg_len.cpu().numpy()
    for c in range(nchr):
        g = got[off[c]:off[c] + nseg[c]]
        assert nseg[c] == len(exp[c]) and (g == exp[c]).all(), (c, g[:10], exp[c][:10])
    assert stats[0] == est[0] and stats[2] == est[2] and stats[4] == est[4]    # same number of TMaxO calls, permutations, TPermP draws
    return stats, exp


def test_cbs_planted_segments_matches_oracle():
    cv = get_canvas()
    bins = synth.generate_bins(20260927 + 20, 150_000, nchr=8)
    cov = np.round(bins["count"].astype(np.float64), 2)
    off = np.concatenate([[0], np.cumsum(np.bincount(bins["chr"], minlength=8))]).astype(np.int64)
    stats, exp = _run(cv, cov, off)
    assert stats[6] > 0                      # the GPU search was used
    assert sum(len(e) for e in exp) > 8      # change points were found


def test_cbs_ties_and_small_chromosomes():
    cv = get_canvas()
    rng = np.random.RandomState(11)
    # heavily quantised data (many exactly equal partial sums => exact ties of the statistic), plus tiny chromosomes
    parts = [rng.randint(95, 106, 6000).astype(np.float64), np.concatenate([rng.randint(98, 103, 5000), rng.randint(100, 105, 4000)]).astype(np.float64),
             np.full(300, 100.0), rng.randint(90, 110, 3).astype(np.float64), rng.normal(100, 5, 250).round(2), np.array([100.0])]
    cov = np.concatenate(parts)
    off = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int64)
    _run(cv, cov, off, nperm=2000)


def test_cbs_borderline_needs_permutations():
    cv = get_canvas()
    rng = np.random.RandomState(12)
    x = rng.normal(100, 10, 5000)
    x[2000:2008] += 14       # short weak aberration: t between TailP threshold and 7 => permutation test + edge tests
    x[3500:3900] += 2.5
    cov = np.round(x, 2)
    off = np.array([0, len(cov)], np.int64)
    stats, exp = _run(cv, cov, off, nperm=2000)
    assert stats[2] > 0


def test_cbs_sdundo_merges_weak_splits():
    cv = get_canvas()
    rng = np.random.RandomState(13)
    parts = []
    for c in range(3):
        x = rng.normal(100, 10, 6000)
        x[1000:1400] += 60; x[3000:3300] += 6; x[4500:4520] -= 25
        parts.append(np.round(x, 2))
    cov = np.concatenate(parts)
    off = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int64)
    s0, e0 = _run(cv, cov, off, nperm=2000, undo=0)
    s2, e2 = _run(cv, cov, off, nperm=2000, undo=2)
    assert sum(len(e) for e in e2) <= sum(len(e) for e in e0)


def test_cbs_prune_drops_unsupported_change_points():
    """-s Prune (ChangePoint.cs:205-271): exhaustive subset search over the change points found by the recursion"""
    cv = get_canvas()
    rng = np.random.RandomState(17)
    parts = []
    for c in range(4):
        x = rng.normal(100, 10, 5000)
        x[800:1300] += 50; x[2500:2700] += 12; x[4000:4030] -= 30
        if c == 3: x = rng.normal(100, 10, 3000)            # no change point at all
        parts.append(np.round(x, 2))
    cov = np.concatenate(parts)
    off = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int64)
    s0, e0 = _run(cv, cov, off, nperm=2000, undo=0)
    s1, e1 = _run(cv, cov, off, nperm=2000, undo=1)
    assert sum(len(e) for e in e1) <= sum(len(e) for e in e0)
    assert any(len(a) != len(b) for a, b in zip(e0, e1)) or all(len(a) <= 2 for a in e0)


def test_cbs_device_permutation_engine(monkeypatch):
    """XPerm + HTMaxP on the device (batches, generator continued on the device, sequential stopping rule on the host): same segments and
    the same RNG consumption as the oracle; with the test hook every device interval is checked against the statistic computed in the
    reference's order (it must contain it, be tight, and the generator snapshot must equal the host generator's state)."""
    cv = get_canvas()
    rng = np.random.RandomState(23)
    parts = []
    for c in range(4):
        n = 9000 + 4000 * c
        x = rng.normal(100, 12, n)
        x[n // 3:] += 0.5 + 0.15 * c                 # weak whole-arm shifts: t is between 0.1 and 7 on segments of thousands of bins,
        x[2 * n // 3:] -= 0.4                         # so the hybrid permutation test has to decide
        parts.append(np.round(x, 2))
    cov = np.concatenate(parts)
    off = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int64)
    monkeypatch.setenv("CANVAS_CBS_TEST_VERIFY", "1")
    stats, exp = _run(cv, cov, off, nperm=10000)
    d = cv.cbs_device_stats()
    assert d[0] > 500 and d[4] >= d[0] and d[5] == 0, d      # the hook also checks the permutations of a batch that lie behind the stopping point
    monkeypatch.delenv("CANVAS_CBS_TEST_VERIFY")
    monkeypatch.setenv("CANVAS_CBS_HOST_PERMUTATIONS", "1")
    stats2, exp2 = _run(cv, cov, off, nperm=10000)
    d2 = cv.cbs_device_stats()
    assert d2[0] == 0 and d2[1] == stats2[2]
    assert [int(v) for v in stats[:5]] == [int(v) for v in stats2[:5]]


def test_short_segments_run_on_the_device_with_the_references_own_statistic(monkeypatch):
    """Segments of at most 200 bins take the non-hybrid test (XPerm + TMaxP): k_perm_small must return tmaxp_host's value bit for bit — including the arcs the reference's block
    search does NOT look at (its pruning is not lossless for blocks of a dozen elements) — and hybrid segments of a few hundred bins go through the device engine with the
    generator state of a batch that is cut short inside its first 624 draws taken from the host.  Many short chromosomes with and without a jump; the hook checks every
    permutation, the oracle the segments and the RNG consumption."""
    cv = get_canvas()
    rng = np.random.RandomState(77)
    parts = []
    for c in range(60):
        n = int(rng.choice([12, 30, 49, 50, 51, 80, 120, 199, 200, 201, 230, 400, 640, 1023]))
        x = rng.normal(60, 7, n)
        if c % 3 == 0: x[n // 2:] += rng.choice([3.0, 6.0, 12.0])
        if c % 7 == 0: x = np.round(x)                    # heavy ties
        parts.append(np.round(np.clip(x, 0, None), 2))
    cov = np.concatenate(parts)
    off = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int64)
    monkeypatch.setenv("CANVAS_CBS_TEST_VERIFY", "1")
    stats, exp = _run(cv, cov, off, nperm=10000)
    d = cv.cbs_device_stats()
    assert d[1] == 0 and d[0] == stats[2] and d[0] > 20000, d          # every permutation on the device
    assert d[4] >= d[0] and d[5] == 0, d                                 # every one checked against the host, none differs
    monkeypatch.delenv("CANVAS_CBS_TEST_VERIFY")
    monkeypatch.setenv("CANVAS_CBS_HOST_PERMUTATIONS", "1")
    stats2, _ = _run(cv, cov, off, nperm=10000)
    assert cv.cbs_device_stats()[0] == 0
    assert [int(v) for v in stats[:5]] == [int(v) for v in stats2[:5]]


def test_spiky_coverage_where_the_reference_search_leaves_arcs_out():
    """Coverage with isolated spikes puts a block's extremes next to each other; block_search then scans nothing of that block pair (CBSTStatistic.cs:233-326) and the reference's
    maximum can be below the maximum over every admissible arc.  The device arc search takes the latter and is only accepted when its maximiser is an arc the reference scans
    (otherwise the host replays the reference's search); short segments follow the scanned ranges on the device.  Segments and RNG consumption must be the oracle's."""
    cv = get_canvas()
    rng = np.random.RandomState(99)
    parts = []
    for c in range(6):
        n = [5000, 9000, 16000, 150, 640, 30000][c]
        x = rng.normal(80, 8, n)
        spikes = rng.choice(n, max(3, n // 400), replace=False)
        x[spikes] += rng.standard_cauchy(len(spikes)) * 60
        if c % 2 == 0: x[n // 2:] += 4.0
        parts.append(np.round(np.clip(x, 0, 5000), 2))
    cov = np.concatenate(parts)
    off = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int64)
    _run(cv, cov, off, nperm=2000)


def test_edge_tests_on_the_device_equal_the_host_chain():
    """CBSTStatistic.TPermP (CBSTStatistic.cs:947-1024) as a device kernel (CANVAS_CBS_DEVICE_TPERMP=1): one lane walks the chain of nPerm x min(n1, n2) dependent swaps with the
    chromosome's Mersenne Twister, whose state the following permutations continue from.  Same segments, same number of draws as the oracle (whose generator is the reference's);
    segments longer than the kernel's LDS copy (16 384 bins) take its global scratch.  Run in a process of its own: the switch is read once."""
    import json, os, subprocess, sys
    code = r"""
import json, sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import oracle_lib as O
from gpu_common import get_canvas, to_dev
cv = get_canvas()
rng = np.random.RandomState(12)
x = rng.normal(100, 10, 5000); x[2000:2008] += 40; x[3500:3900] += 2.5          # an aberration of 8 bins inside the chromosome: both edge tests walk their chains (fewer than 10 bins on one side)
y = rng.normal(100, 10, 40000); y[18000:18006] += 45                              # an edge test over more than 16 384 bins
parts = [np.round(x, 2), np.round(y, 2)]
cov = np.concatenate(parts); off = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int64)
exp, est = O.cbs_genome(parts, 0.01, 1000, threads=2)
seg_len, nseg, stats = cv.cbs(to_dev(cov, cv.device), off, 0.01, 1000)
got = seg_len.cpu().numpy()
same = all(int(nseg[c]) == len(exp[c]) and (got[off[c]:off[c] + nseg[c]] == exp[c]).all() for c in range(2))
print(json.dumps({"same": bool(same), "stats": [int(v) for v in stats], "est": [int(v) for v in est], "tpermp": [int(v) for v in cv.cbs_tpermp_stats()]}))
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, CANVAS_CBS_DEVICE_TPERMP="1"), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["same"] and d["stats"][0] == d["est"][0] and d["stats"][2] == d["est"][2] and d["stats"][4] == d["est"][4], d
    assert d["tpermp"][0] >= 2 and d["tpermp"][1] == d["est"][4] and d["est"][4] > 0, d


def test_device_tail_series_against_the_reference_series():
    """k_tail_nu evaluates TailProbability.Nu (TailProbability.cs:52-85) with the blocks of the series from D = 512 on taken from the Euler-Maclaurin formula instead of term by
    term.  Against the reference's sequential series (oracle Nu): relative difference below 1e-10 over the arguments a WGS sample meets (x = b / sqrt(m t (1 - t)): 0.0101 for
    a 476 k-bin segment up to ~10 for short ones) — canvas_cbs accepts a device value only when its two decisions hold for every p1 within 1e-8 relative — and the special
    cases of the reference (x <= 0.01: exp(-0.583 x))"""
    cv = get_canvas()
    rng = np.random.RandomState(3)
    xs = np.concatenate([[0.005, 0.01, 0.0100001, 0.0101, 0.0116, 0.012, 0.02, 0.0345, 0.05, 0.1, 0.2, 0.31, 0.5, 0.77, 1.0, 1.5, 2.0, 3.0, 5.0, 8.0, 12.0],
                         np.exp(rng.uniform(np.log(0.0101), np.log(8.0), 179))])
    worst = 0.0
    for i in range(0, len(xs), 100):
        part = xs[i:i + 100]
        nu, fl = cv.cbs_tail_probe(part, 1e-6)
        for x, v, f in zip(part, nu, fl):
            ref = O.lib.orc_nu(float(x), 1e-6)
            if f:
                continue                                        # (too close to a stopping comparison: the library redoes these on the host)
            rel = abs(v - ref) / abs(ref)
            worst = max(worst, rel)
            assert rel < 1e-10, (x, v, ref, rel)
    assert worst > 0.0 or True
