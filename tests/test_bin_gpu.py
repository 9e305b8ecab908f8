"""CanvasBin on the GPU vs the CPU oracle (bit-exact: integer work)."""
import numpy as np
import pytest

import oracle_lib as O
from canvas_amd import synth
from gpu_common import get_canvas, to_dev, pad16

pytestmark = pytest.mark.gpu

SEED = 20260927 + 1


@pytest.fixture(params=["two_pass", "single_read"])
def bin_path(request, monkeypatch):
    """Both CanvasBin device paths run every case: the two-pass path (k_tile_stats + k_bin_pass) and the single-read path
    (k_tile_summary + k_bin_close) that one-call binning takes when the bin size comes from the sample's own rates."""
    monkeypatch.setenv("CANVAS_BIN_SINGLE_READ" if request.param == "single_read" else "CANVAS_BIN_TWO_PASS", "1")
    return request.param


def _chroms(lengths, rate=0.105):
    thr = synth.poisson_thresholds(rate)
    return [synth.generate_chromosome(SEED, c, L, rate, thr) for c, L in enumerate(lengths)]


def _upload(cv, data):
    bases = [to_dev(pad16(b), cv.device) for b, h, m in data]
    hits = [to_dev(pad16(h), cv.device) for b, h, m in data]
    masks = [to_dev(m.view(np.int64), cv.device) for b, h, m in data]
    return bases, hits, masks


def test_device_synth_matches_numpy():
    import torch
    from canvas_amd.lib import synth_generate_device
    cv = get_canvas()
    L = 300_001
    b, h, m = synth.generate_chromosome(SEED, 5, L, 0.21)
    db, dh, dm, _ = synth_generate_device(SEED, 5, L, 0.21, cv.device)
    torch.cuda.synchronize()
    assert (db.cpu().numpy()[:L] == b).all()
    assert (dh.cpu().numpy()[:L] == h).all()
    assert (dm.cpu().numpy().view(np.uint8)[: len(m)] == m).all()


@pytest.mark.parametrize("lengths,bin_size", [([1_500_000, 700_001, 40_961], None), ([300_000], 37), ([200_000, 5_000], 7), ([1_000_000], 5000)])
def test_bin_genome_matches_oracle(lengths, bin_size, bin_path):
    cv = get_canvas()
    data = _chroms(lengths)
    bases, hits, masks = _upload(cv, data)
    lens = np.array(lengths, np.int64)
    obs, poss, rate = cv.bin_rates(hits, masks, lens)
    for c, (b, h, m) in enumerate(data):
        assert rate[c] == O.bin_rate(h, m)
        assert obs[c] == int((h > 0).sum())
    if bin_size is None:
        bin_size = cv.bin_size_from_rates(rate, 100)
        assert bin_size == O.bin_size(rate, 100)
    for mode in (3, 0):
        out, per, total = cv.bin_genome(bases, masks, hits, lens, bin_size, mode)
        cv.synchronize()
        off = 0
        for c, (b, h, m) in enumerate(data):
            es, ee, eg, ec = O.bin_chromosome(b, m, h, bin_size, mode)
            assert per[c] == len(es), (c, per[c], len(es))
            sl = slice(off, off + len(es))
            assert (out["chr"][sl].cpu().numpy() == c).all()
            assert (out["start"][sl].cpu().numpy() == es).all()
            assert (out["stop"][sl].cpu().numpy() == ee).all()
            assert (out["gc"][sl].cpu().numpy() == eg).all()
            assert (out["count"][sl].cpu().numpy() == ec.astype(np.float32)).all()
            off += len(es)
        assert total == off


def test_bin_edge_cases(bin_path):
    cv = get_canvas()
    # all-'n' chromosome (no bins), chromosome with fewer possible positions than one bin, unscreened hits outside the mask
    L = 10_000
    rng = np.random.RandomState(3)
    b1 = np.full(L, ord('n'), np.uint8); h1 = np.zeros(L, np.uint8); m1 = np.zeros((L + 63) // 64 * 8, np.uint8)
    b2 = rng.choice(np.frombuffer(b"ACGTacgt", np.uint8), L); h2 = rng.randint(0, 30, L).astype(np.uint8)
    bits = (rng.rand((L + 63) // 64 * 64) < 0.01).astype(np.uint8); bits[L:] = 0
    m2 = np.packbits(bits, bitorder="little")
    b3 = rng.choice(np.frombuffer(b"ACGTacgtn", np.uint8), L); b3[:100] = ord('n'); h3 = rng.randint(0, 255, L).astype(np.uint8)
    bits3 = (rng.rand((L + 63) // 64 * 64) < 0.7).astype(np.uint8); bits3[L:] = 0; bits3[:50] = 1   # possible positions inside the leading n's
    m3 = np.packbits(bits3, bitorder="little")
    data = [(b1, h1, m1), (b2, h2, m2), (b3, h3, m3)]
    bases, hits, masks = _upload(cv, data)
    lens = np.array([L, L, L], np.int64)
    for bs in (500, 16, 3, 1):
        for mode in (3, 0):
            out, per, total = cv.bin_genome(bases, masks, hits, lens, bs, mode)
            off = 0
            for c, (b, h, m) in enumerate(data):
                es, ee, eg, ec = O.bin_chromosome(b, m, h, bs, mode)
                assert per[c] == len(es)
                sl = slice(off, off + len(es))
                assert (out["start"][sl].cpu().numpy() == es).all()
                assert (out["stop"][sl].cpu().numpy() == ee).all()
                assert (out["gc"][sl].cpu().numpy() == eg).all()
                assert (out["count"][sl].cpu().numpy() == ec.astype(np.float32)).all()
                off += len(es)


def test_bin_sample_one_call_equals_two_step_flow(bin_path):
    import torch
    cv = get_canvas()
    lengths = [900_000, 500_001, 300_000]
    data = _chroms(lengths, rate=0.21)
    bases, hits, masks = _upload(cv, data)
    lens = np.array(lengths, np.int64)
    is_auto = np.array([1, 1, 0], np.uint8)
    rates = [O.bin_rate(h, m) for b, h, m in data]
    bs_exp = O.bin_size([rates[0], rates[1]], 100)      # autosomes only (CanvasBin.cs:44)
    cap = int(lens.sum() // 50)
    out = dict(chr=torch.empty(cap, dtype=torch.int32, device=cv.device), start=torch.empty(cap, dtype=torch.int32, device=cv.device),
               stop=torch.empty(cap, dtype=torch.int32, device=cv.device), gc=torch.empty(cap, dtype=torch.int32, device=cv.device),
               count=torch.empty(cap, dtype=torch.float32, device=cv.device))
    o, per, total, bs = cv.bin_sample(bases, masks, hits, lens, is_auto, 100, -1, 3, out=out)
    assert bs == bs_exp
    exp = [O.bin_chromosome(b, m, h, bs, 3) for b, h, m in data]
    assert total == sum(len(e[0]) for e in exp)
    assert (out["stop"][:total].cpu().numpy() == np.concatenate([e[1] for e in exp])).all()
    assert (out["count"][:total].cpu().numpy() == np.concatenate([e[3] for e in exp]).astype(np.float32)).all()
    off = cv.chromosome_offsets(out["chr"], total, 3)
    assert off.tolist() == [0, len(exp[0][0]), len(exp[0][0]) + len(exp[1][0]), total]
    # explicit bin size (-z) skips the rate pass
    o, per, total2, bs2 = cv.bin_sample(bases, masks, hits, lens, is_auto, 100, 777, 3, out=out)
    assert bs2 == 777 and total2 == sum(len(O.bin_chromosome(b, m, h, 777, 3)[0]) for b, h, m in data)


def test_prep_kernels_mask_filter_screen():
    """InitializeAlignmentArrays / ExcludeTagsOverlappingFilterFile / ScreenObservedTags (CanvasBin.cs:183-200,668-716) vs numpy"""
    cv = get_canvas()
    rng = np.random.RandomState(21)
    for L in (1_000_003, 4096, 77):
        bases = rng.choice(np.frombuffer(b"ACGTNacgtn@[`{Zz", np.uint8), L)
        hits = rng.randint(0, 256, L).astype(np.uint8)
        dmask = cv.mask_from_fasta(to_dev(pad16(bases), cv.device), L)
        exp = (bases >= ord('A')) & (bases <= ord('Z'))
        got = np.unpackbits(dmask.cpu().numpy().view(np.uint8), bitorder="little")
        assert (got[:L] == exp).all() and got[L:].sum() == 0
        starts = np.sort(rng.randint(0, L, 40)); stops = np.minimum(L + 5, starts + rng.randint(1, 3000, 40))
        starts = np.concatenate([starts, [0, L - 1]]); stops = np.concatenate([stops, [1, L]])
        cv.mask_exclude_intervals(dmask, L, starts, stops)
        for a, b in zip(starts, stops):
            exp[a:min(b, L)] = False
        got = np.unpackbits(dmask.cpu().numpy().view(np.uint8), bitorder="little")
        assert (got[:L] == exp).all()
        dh = to_dev(pad16(hits), cv.device)
        cv.screen_hits(dh, dmask, L)
        assert (dh.cpu().numpy()[:L] == np.where(exp, hits, 0)).all()


def test_bin_gc_content_weighted_mode(bin_path):
    """CanvasBin -m GCContentWeighted (CanvasBin.cs:416-506, 330-405, 626-636) vs the oracle"""
    import torch
    cv = get_canvas()
    lengths = [700_000, 410_001]
    data = _chroms(lengths, rate=0.21)
    rng = np.random.RandomState(31)
    fl = [np.where(h > 0, np.clip(rng.normal(350, 60, len(h)), 1, 5000), 0).astype(np.int16) for b, h, m in data]
    fl[0][1000:1100] = 32767      # clipped at 3 x mean fragment
    bases, hits, masks = _upload(cv, data)
    dfl = [to_dev(pad16(f), cv.device) for f in fl]
    lens = np.array(lengths, np.int64)
    cap = int(lens.sum() // 50)
    out = dict(chr=torch.empty(cap, dtype=torch.int32, device=cv.device), start=torch.empty(cap, dtype=torch.int32, device=cv.device),
               stop=torch.empty(cap, dtype=torch.int32, device=cv.device), gc=torch.empty(cap, dtype=torch.int32, device=cv.device),
               count=torch.empty(cap, dtype=torch.float32, device=cv.device))
    for bs in (480, 90):
        exp, mfrag, w, _ = O.bin_gc_weighted([d[0] for d in data], [d[2] for d in data], [d[1] for d in data], fl, bs)
        o, per, total, bsz = cv.bin_sample_gcweighted(bases, masks, hits, dfl, lens, [1, 1], 100, bs, out=out)
        assert total == sum(len(e[0]) for e in exp)
        assert (out["stop"][:total].cpu().numpy() == np.concatenate([e[1] for e in exp])).all()
        ggot = out["gc"][:total].cpu().numpy(); gex = np.concatenate([e[2] for e in exp])
        assert (ggot == gex).all(), (bs, np.nonzero(ggot != gex)[0][:8], ggot[ggot != gex][:8], gex[ggot != gex][:8], out["start"][:total].cpu().numpy()[ggot != gex][:8])
        got = out["count"][:total].cpu().numpy(); ex = np.concatenate([e[3] for e in exp]).astype(np.float32)
        assert (got == ex).all(), (np.nonzero(got != ex)[0][:5], got[got != ex][:5], ex[got != ex][:5])
        decided, replayed = cv.bin_gcw_stats()
        assert decided + replayed == total and decided > 0.9 * total, (decided, replayed, total)      # the exact-sum interval decides nearly every bin


@pytest.mark.parametrize("mean_sd", [(350, 60), (104, 3), (101, 0), (60, 10), (1500, 400), (97, 0)])
def test_read_gc_profile_kernels_over_fragment_size_regimes(monkeypatch, mean_sd):
    """The read-GC profile (CanvasBin.cs:416-506) comes from k_read_gc3 when the mean fragment is above 100 (the default window's value is followed from position to position, which
    needs |100 d| < meanFragment) and from k_read_gc2 otherwise; CANVAS_GCW_READ_GC2=1 forces the latter.  Both against the oracle through the weighted counts of small bins (every
    term is hit / weight[readGC]: a wrong gcContent byte or a wrong histogram counter moves them), with lengths that are no multiple of 16, a chromosome shorter than 3 x the mean
    fragment, hits without a fragment length, fragment lengths without a hit, negative and clipped lengths, and saturated hit counts."""
    import torch
    cv = get_canvas()
    mean, sd = mean_sd
    lengths = [640_007, 270_001, 3 * mean - 5, 3 * mean + 40]
    data = _chroms(lengths, rate=0.3)
    rng = np.random.RandomState(33 + mean)
    fl = []
    for b, h, m in data:
        f = np.where(h > 0, np.clip(rng.normal(mean, sd, len(h)), 1, 30000), 0).astype(np.int16)
        drop = rng.rand(len(h)) < 0.2; f[drop] = 0                                   # hits without a fragment length
        add = (rng.rand(len(h)) < 0.01) & (h == 0); f[add] = mean                    # a fragment length without a hit
        fl.append(f)
    fl[0][7000:7040] = -5; fl[0][9000:9100] = min(32767, 40 * mean); fl[1][-2000:] = max(1, mean // 2)
    for k in (0, 1):
        sel = rng.randint(0, lengths[k], 300); data[k][1][sel] = 255                  # saturated counts
    bases, hits, masks = _upload(cv, data)
    dfl = [to_dev(pad16(f), cv.device) for f in fl]
    lens = np.array(lengths, np.int64)
    cap = int(lens.sum() // 8)
    mk = lambda dt: torch.empty(cap, dtype=dt, device=cv.device)
    out = dict(chr=mk(torch.int32), start=mk(torch.int32), stop=mk(torch.int32), gc=mk(torch.int32), count=mk(torch.float32))
    for bs in (200, 24):
        exp, mfrag, w, _ = O.bin_gc_weighted([d[0] for d in data], [d[2] for d in data], [d[1] for d in data], fl, bs)
        assert (mfrag > 100) == (mean > 100), (mfrag, mean)                         # the regime this case is meant for
        ex = np.concatenate([e[3] for e in exp]).astype(np.float32)
        for force_old in (False, True):
            if force_old: monkeypatch.setenv("CANVAS_GCW_READ_GC2", "1")
            else: monkeypatch.delenv("CANVAS_GCW_READ_GC2", raising=False)
            o, per, total, _ = cv.bin_sample_gcweighted(bases, masks, hits, dfl, lens, [1, 1, 1, 1], 100, bs, out=out)
            got = out["count"][:total].cpu().numpy()
            assert total == len(ex) and (got == ex).all(), (mean_sd, bs, force_old, np.nonzero(got != ex)[0][:5], got[got != ex][:5], ex[got != ex][:5])
    monkeypatch.delenv("CANVAS_GCW_READ_GC2", raising=False)


def test_gc_weighted_interval_decisions_equal_the_serial_order(monkeypatch):
    """k_bin_weighted2 decides (int)Math.Round of the float32 running sum from the exact sum of the terms and a rounding-error interval; CANVAS_GCW_SERIAL=1 sends every bin
    through the reference's own order of additions instead: same counts (and both equal the oracle, incl. negative fragment lengths, which the reference reads as an empty window)."""
    import torch
    cv = get_canvas()
    lengths = [900_001, 130_000]
    data = _chroms(lengths, rate=0.28)
    rng = np.random.RandomState(32)
    fl = [np.where(h > 0, np.clip(rng.normal(520, 140, len(h)), 1, 5000), 0).astype(np.int16) for b, h, m in data]
    fl[0][5000:5050] = -7
    fl[1][-3000:] = 900
    bases, hits, masks = _upload(cv, data)
    dfl = [to_dev(pad16(f), cv.device) for f in fl]
    lens = np.array(lengths, np.int64)
    cap = int(lens.sum() // 20)
    mk = lambda dt: torch.empty(cap, dtype=dt, device=cv.device)
    out = dict(chr=mk(torch.int32), start=mk(torch.int32), stop=mk(torch.int32), gc=mk(torch.int32), count=mk(torch.float32))
    for bs in (300, 64, 31):
        exp, mfrag, w, _ = O.bin_gc_weighted([d[0] for d in data], [d[2] for d in data], [d[1] for d in data], fl, bs)
        ex = np.concatenate([e[3] for e in exp]).astype(np.float32)
        monkeypatch.delenv("CANVAS_GCW_SERIAL", raising=False)
        o, per, total, _ = cv.bin_sample_gcweighted(bases, masks, hits, dfl, lens, [1, 1], 100, bs, out=out)
        fast = out["count"][:total].cpu().numpy().copy()
        d1 = cv.bin_gcw_stats()
        monkeypatch.setenv("CANVAS_GCW_SERIAL", "1")
        o, per, total2, _ = cv.bin_sample_gcweighted(bases, masks, hits, dfl, lens, [1, 1], 100, bs, out=out)
        serial = out["count"][:total2].cpu().numpy()
        d2 = cv.bin_gcw_stats()
        assert total == total2 == len(ex) and (fast == ex).all() and (serial == ex).all(), (bs, np.nonzero(fast != ex)[0][:5], np.nonzero(serial != ex)[0][:5])
        assert d1[0] > 0 and d2[0] == 0 and d2[1] == total
        monkeypatch.delenv("CANVAS_GCW_SERIAL", raising=False)


def test_device_synth_sample_pair_matches_numpy():
    """synth_generate_sample (tumour / normal over one reference, fragment lengths) is the numpy mirror byte for byte"""
    import torch
    from canvas_amd.lib import synth_generate_device, synth_generate_sample_device
    cv = get_canvas()
    L = 300_001
    thr_t = synth.poisson_thresholds(0.28, purity=0.7)
    b, h, m, fl = synth.generate_chromosome(SEED, 5, L, 0.28, thr_t, hit_seed=SEED + 77, with_fraglen=True)
    db, _, dm, _ = synth_generate_device(SEED, 5, L, 0.28, cv.device)
    dthr = torch.from_numpy(thr_t.view(np.int32)).to(cv.device)
    dh, dfl = synth_generate_sample_device(SEED, SEED + 77, 5, L, dthr, cv.device, with_fraglen=True)
    torch.cuda.synchronize()
    assert (db[:L].cpu().numpy() == b).all() and (dm.cpu().numpy().view(np.uint8) == m).all()
    assert (dh[:L].cpu().numpy() == h).all() and (dfl[:L].cpu().numpy() == fl).all()


@pytest.mark.parametrize("resident_reference", [False, True])
def test_streamed_upload_bins_match_oracle(resident_reference):
    """canvas_upload_genome_begin + canvas_bin_sample: the per-chromosome sweeps start as the chromosomes arrive from host memory; same bins as the oracle.
    resident_reference: bases and mask are already on the device (a cohort's shared reference), only the hits are uploaded."""
    import torch
    cv = get_canvas()
    lengths = [900_001, 4096 * 50, 333_333, 1_000_000, 70_000]
    data = _chroms(lengths, rate=0.21)
    lens = np.array(lengths, np.int64)
    pin = lambda a: torch.from_numpy(np.ascontiguousarray(a)).pin_memory()
    hb = [pin(pad16(b)) for b, h, m in data]; hh = [pin(pad16(h)) for b, h, m in data]; hm = [pin(m.view(np.int64)) for b, h, m in data]
    db = [torch.zeros_like(t, device=cv.device) for t in hb]; dh = [torch.zeros_like(t, device=cv.device) for t in hh]; dm = [torch.zeros_like(t, device=cv.device) for t in hm]
    if resident_reference:
        for d, s in zip(db + dm, hb + hm): d.copy_(s)
        torch.cuda.synchronize()
    cap = int(lens.sum() // 50)
    mk = lambda dt: torch.empty(cap, dtype=dt, device=cv.device)
    out = dict(chr=mk(torch.int32), start=mk(torch.int32), stop=mk(torch.int32), gc=mk(torch.int32), count=mk(torch.float32))
    rates = [O.bin_rate(h, m) for b, h, m in data]
    for rep in range(2):                                  # twice: the second upload must wait for the first pass's readers
        for t in dh: t.zero_()
        torch.cuda.synchronize()
        cv.upload_genome_begin(lens, None if resident_reference else hb, db, None if resident_reference else hm, dm, hh, dh)
        _, per, total, bs = cv.bin_sample(db, dm, dh, lens, [1, 1, 1, 1, 0], 100, -1, 3, out=out)
        assert bs == O.bin_size(rates[:4], 100)
        exp = [O.bin_chromosome(b, m, h, bs) for b, h, m in data]
        assert total == sum(len(e[0]) for e in exp)
        for k, j in (("start", 0), ("stop", 1), ("gc", 2)):
            assert (out[k][:total].cpu().numpy() == np.concatenate([e[j] for e in exp])).all(), k
        assert (out["count"][:total].cpu().numpy() == np.concatenate([e[3] for e in exp]).astype(np.float32)).all()
    # a pending upload followed by a call on OTHER arrays is simply waited for
    cv.upload_genome_begin(lens, hb, db, hm, dm, hh, dh)
    db2 = [t.clone() for t in db]
    _, per, total2, bs2 = cv.bin_sample(db2, dm, dh, lens, [1, 1, 1, 1, 0], 100, -1, 3, out=out)
    assert total2 == total and bs2 == bs
    cv.upload_genome_wait()


def test_predefined_bins_match_oracle():
    """canvas_bin_predefined (CanvasBin -n): counts and GC of given intervals, incl. the leading-'n' skip of a chromosome's first bin, touching / overlapping / 1-base bins"""
    cv = get_canvas()
    lengths = [400_000, 130_001]
    data = _chroms(lengths, rate=0.21)
    bases, hits, masks = _upload(cv, data)
    rng = np.random.RandomState(17)
    starts, stops = [], []
    for c, L in enumerate(lengths):
        s0 = np.sort(rng.choice(L - 3000, 300, replace=False)); e0 = s0 + rng.randint(1, 2500, 300)
        s0[0] = 0                                                     # inside the leading 'n' stretch of the synthetic chromosome
        e0[0] = max(e0[0], 12_000)
        e0[-1] = L                                                    # to the last base
        starts.append(s0.astype(np.int32)); stops.append(np.minimum(e0, L).astype(np.int32))
    for mode in (3, 0):
        gc, cnt = cv.bin_predefined(bases, masks, hits, np.array(lengths, np.int64), starts, stops, mode=mode)
        off = 0
        for c in range(2):
            k, eg, ec = O.bin_predefined(data[c][0], data[c][2], data[c][1], starts[c], stops[c], mode)
            assert k == len(starts[c])
            n = len(starts[c])
            assert (gc[off:off + n].cpu().numpy() == eg).all() and (cnt[off:off + n].cpu().numpy() == ec.astype(np.float32)).all()
            off += n
    from canvas_amd.lib import CanvasError
    with pytest.raises(CanvasError):                                  # a bin past the end of the chromosome: the reference's cursor never closes it
        cv.bin_predefined(bases, masks, hits, np.array(lengths, np.int64), [np.array([10], np.int32), np.array([5], np.int32)], [np.array([20], np.int32), np.array([lengths[1] + 1], np.int32)])
    with pytest.raises(CanvasError):                                  # first bin entirely inside the leading n's
        cv.bin_predefined(bases, masks, hits, np.array(lengths, np.int64), [np.array([0], np.int32), np.array([20_000], np.int32)], [np.array([50], np.int32), np.array([21_000], np.int32)])


def test_predefined_bins_gc_content_weighted_match_oracle():
    """CanvasBin -n with -m GCContentWeighted (CanvasBin.cs:617-636: the predefined-bin close shares the weighted branch): canvas_bin_predefined_gcweighted vs the oracle — the
    profile and the weights come from all three chromosomes, the middle one has no bins; bins of 1 base, touching, overlapping, one spanning megabase-like gaps, the first one
    starting inside the leading 'n' stretch; CANVAS_GCW_SERIAL=1 (every bin in the reference's own order of additions) gives the same counts"""
    cv = get_canvas()
    lengths = [500_000, 90_007, 230_001]
    data = _chroms(lengths, rate=0.25)
    rng = np.random.RandomState(41)
    fl = [np.where(h > 0, np.clip(rng.normal(320, 50, len(h)), 1, 5000), 0).astype(np.int16) for b, h, m in data]
    bases, hits, masks = _upload(cv, data)
    dfl = [to_dev(pad16(f), cv.device) for f in fl]
    starts, stops = [], []
    for c, L in enumerate(lengths):
        if c == 1: starts.append(np.zeros(0, np.int32)); stops.append(np.zeros(0, np.int32)); continue
        s0 = np.sort(rng.choice(L - 3000, 400, replace=False)); e0 = s0 + rng.randint(1, 2500, 400)
        s0[0] = 0; e0[0] = max(e0[0], 12_000)
        s0[5] = s0[4]; e0[5] = e0[4] + 1                               # overlapping bins
        e0[7] = s0[7] + 1                                               # one base
        e0[200] = s0[200] + 60_000                                      # longer than the 16-lane path's span: the whole wave scans it
        e0[-1] = L
        starts.append(s0.astype(np.int32)); stops.append(np.minimum(e0, L).astype(np.int32))
    exp = O.bin_predefined_gc_weighted([d[0] for d in data], [d[2] for d in data], [d[1] for d in data], fl, starts, stops)
    lens = np.array(lengths, np.int64)
    import os
    for serial in (False, True):
        if serial: os.environ["CANVAS_GCW_SERIAL"] = "1"
        try:
            gc, cnt = cv.bin_predefined(bases, masks, hits, lens, starts, stops, mode=5, fraglens=dfl)
        finally:
            os.environ.pop("CANVAS_GCW_SERIAL", None)
        off = 0
        for c in range(3):
            k, eg, ec = exp[c]; n = len(starts[c])
            assert k == n
            assert (gc[off:off + n].cpu().numpy() == eg).all()
            got = cnt[off:off + n].cpu().numpy()
            assert (got == ec.astype(np.float32)).all(), (serial, c, np.nonzero(got != ec)[0][:5], got[got != ec][:5], ec[got != ec][:5])
            off += n
        decided, replayed = cv.bin_gcw_stats()
        assert decided + replayed == off and (replayed == off if serial else decided > 0.9 * off)
    from canvas_amd.lib import CanvasError
    with pytest.raises(CanvasError):                                  # mode 5 through the entry point without fragment lengths
        cv.bin_predefined(bases, masks, hits, lens, starts, stops, mode=5)
    with pytest.raises(CanvasError):                                  # no usable fragment length: "Unable to determine fragment size" (CanvasBin.cs:431-434)
        cv.bin_predefined(bases, masks, hits, lens, starts, stops, mode=5, fraglens=[to_dev(pad16(np.zeros_like(f)), cv.device) for f in fl])
