"""CanvasBin over the packed planes (canvas_bin_sample_packed, bin_packed.hpp): same bins as the CPU oracle and as the byte-array path, bit for bit.
The planes come from the device packer AND from the host packer (both must produce the same bytes)."""
import numpy as np
import pytest

import oracle_lib as O
from canvas_amd import synth, CLEAN_GCNORM, CLEAN_FILTSIZE, CLEAN_OUTLIERS, CLEAN_LOCALSD
from canvas_amd.lib import pack_reference_host, pack_hits_host, packed_plane_words
from gpu_common import get_canvas, to_dev, pad16

pytestmark = pytest.mark.gpu

SEED = 20260927 + 41


def _chroms(lengths, rate=0.21):
    thr = synth.poisson_thresholds(rate)
    return [synth.generate_chromosome(SEED, c, L, rate, thr) for c, L in enumerate(lengths)]


def _upload(cv, data):
    bases = [to_dev(pad16(b), cv.device) for b, h, m in data]
    hits = [to_dev(pad16(h), cv.device) for b, h, m in data]
    masks = [to_dev(np.ascontiguousarray(m).view(np.int64), cv.device) for b, h, m in data]
    return bases, hits, masks


def _out(cv, cap):
    import torch
    mk = lambda dt: torch.empty(cap, dtype=dt, device=cv.device)
    return dict(chr=mk(torch.int32), start=mk(torch.int32), stop=mk(torch.int32), gc=mk(torch.int32), count=mk(torch.float32))


def _host_planes(data):
    ref, planes, pos0, sat = [], [], [], 0
    for b, h, m in data:
        L = len(b)
        r, p0 = pack_reference_host(np.ascontiguousarray(b), np.ascontiguousarray(m).view(np.uint64), L, threads=3)
        p, s = pack_hits_host(np.ascontiguousarray(h), L, threads=3)
        ref.append(r); planes.append(p); pos0.append(p0); sat += s
    return ref, planes, np.array(pos0, np.int64), sat


def _check_against_oracle(out, per, total, data, bs, mode):
    off = 0
    for c, (b, h, m) in enumerate(data):
        es, ee, eg, ec = O.bin_chromosome(b, m, h, bs, mode)
        assert per[c] == len(es), (c, per[c], len(es))
        sl = slice(off, off + len(es))
        assert (out["chr"][sl].cpu().numpy() == c).all()
        assert (out["start"][sl].cpu().numpy() == es).all()
        assert (out["stop"][sl].cpu().numpy() == ee).all()
        assert (out["gc"][sl].cpu().numpy() == eg).all()
        assert (out["count"][sl].cpu().numpy() == ec.astype(np.float32)).all()
        off += len(es)
    assert total == off


def test_device_and_host_packers_agree():
    cv = get_canvas()
    lengths = [700_001, 4096 * 3, 65, 333_333]
    data = _chroms(lengths)
    data[0][1][5000:5100] = 200                                   # pile-up: saturates at 15
    bases, hits, masks = _upload(cv, data)
    dref, dpl, pos0, sat = cv.pack_genome_device(bases, masks, hits, np.array(lengths, np.int64))
    href, hpl, hpos0, hsat = _host_planes(data)
    assert pos0.tolist() == hpos0.tolist() and sat == hsat and sat >= 100
    for c in range(len(lengths)):
        assert (dref[c].cpu().numpy().view(np.uint64) == href[c]).all(), c
        assert (dpl[c].cpu().numpy().view(np.uint64) == hpl[c]).all(), c
    # hits only (second sample over a packed reference)
    _, dpl2, p0none, _ = cv.pack_genome_device(None, None, hits, np.array(lengths, np.int64))
    assert p0none is None and all((a == b).all() for a, b in zip(dpl, dpl2))


@pytest.mark.parametrize("lengths,bin_size", [([1_500_000, 700_001, 40_961], None), ([300_000], 37), ([200_000, 5_000], 7), ([1_000_000, 4096 * 16], 5000)])
def test_packed_bins_match_oracle(lengths, bin_size):
    cv = get_canvas()
    data = _chroms(lengths)
    lens = np.array(lengths, np.int64)
    href, hpl, pos0, _ = _host_planes(data)
    dref = [to_dev(r.view(np.int64), cv.device) for r in href]; dpl = [to_dev(p.view(np.int64), cv.device) for p in hpl]
    is_auto = np.ones(len(lengths), np.uint8)
    out = _out(cv, int(lens.sum() // (bin_size or 50)) + 8)
    for mode in (3, 0):
        d2 = data if mode == 3 else [(b, np.minimum(h, 1), m) for b, h, m in data]          # Binary mode: the hit array holds 0 / 1 (CanvasBin.cs:259-262)
        dp = dpl if mode == 3 else [to_dev(pack_hits_host(np.ascontiguousarray(h), len(h))[0].view(np.int64), cv.device) for b, h, m in d2]
        _, per, total, bs = cv.bin_sample_packed(dref, dp, lens, pos0, is_auto, 100, bin_size or -1, mode, out=out)
        if bin_size is None:
            assert bs == O.bin_size([O.bin_rate(h, m) for b, h, m in d2], 100)
        else:
            assert bs == bin_size
        _check_against_oracle(out, per, total, d2, bs, mode)


def test_packed_edge_cases():
    """all-'n' chromosome, fewer possible positions than one bin, possible positions inside the leading n's, hits far above 15, bin sizes down to 1"""
    cv = get_canvas()
    L = 10_000
    rng = np.random.RandomState(3)
    b1 = np.full(L, ord('n'), np.uint8); h1 = np.zeros(L, np.uint8); m1 = np.zeros((L + 63) // 64 * 8, np.uint8)
    b2 = rng.choice(np.frombuffer(b"ACGTacgt", np.uint8), L); h2 = rng.randint(0, 30, L).astype(np.uint8)
    bits = (rng.rand((L + 63) // 64 * 64) < 0.01).astype(np.uint8); bits[L:] = 0
    m2 = np.packbits(bits, bitorder="little")
    b3 = rng.choice(np.frombuffer(b"ACGTacgtn", np.uint8), L); b3[:100] = ord('n'); h3 = rng.randint(0, 255, L).astype(np.uint8)
    bits3 = (rng.rand((L + 63) // 64 * 64) < 0.7).astype(np.uint8); bits3[L:] = 0; bits3[:50] = 1
    m3 = np.packbits(bits3, bitorder="little")
    L4 = 4096 * 2 + 17                                            # pos0 in the middle of a word of the second tile
    b4 = rng.choice(np.frombuffer(b"ACGT", np.uint8), L4); b4[:4096 + 700] = ord('n'); h4 = rng.randint(0, 12, L4).astype(np.uint8)
    m4 = np.packbits(np.concatenate([np.ones(L4, np.uint8), np.zeros((-L4) % 64, np.uint8)]), bitorder="little")
    data = [(b1, h1, m1), (b2, h2, m2), (b3, h3, m3), (b4, h4, m4)]
    lens = np.array([L, L, L, L4], np.int64)
    href, hpl, pos0, sat = _host_planes(data)
    assert pos0.tolist() == [L, 0, 100, 4096 + 700] and sat > 0
    dref = [to_dev(r.view(np.int64), cv.device) for r in href]; dpl = [to_dev(p.view(np.int64), cv.device) for p in hpl]
    out = _out(cv, int(lens.sum()) + 8)
    for bs in (500, 16, 3, 1):
        _, per, total, _ = cv.bin_sample_packed(dref, dpl, lens, pos0, [1, 1, 1, 1], 100, bs, 3, out=out)
        _check_against_oracle(out, per, total, data, bs, 3)
    # the rates too: derived bin size as the oracle derives it
    _, per, total, bs = cv.bin_sample_packed(dref, dpl, lens, pos0, [0, 1, 1, 1], 100, -1, 3, out=out)
    assert bs == O.bin_size([O.bin_rate(h, m) for b, h, m in data[1:]], 100)
    _check_against_oracle(out, per, total, data, bs, 3)


def test_packed_equals_byte_path_and_streamed_upload():
    """the byte-array path and the packed path leave the same arrays; so does the packed path fed by canvas_upload_packed_begin (sweeps overlap the upload)"""
    import torch
    cv = get_canvas()
    lengths = [900_001, 4096 * 50, 333_333, 1_000_000, 70_000]
    data = _chroms(lengths)
    lens = np.array(lengths, np.int64)
    bases, hits, masks = _upload(cv, data)
    is_auto = [1, 1, 1, 1, 0]
    cap = int(lens.sum() // 50)
    o1 = _out(cv, cap); o2 = _out(cv, cap); o3 = _out(cv, cap)
    _, per1, tot1, bs1 = cv.bin_sample(bases, masks, hits, lens, is_auto, 100, -1, 3, out=o1)
    dref, dpl, pos0, _ = cv.pack_genome_device(bases, masks, hits, lens)
    _, per2, tot2, bs2 = cv.bin_sample_packed(dref, dpl, lens, pos0, is_auto, 100, -1, 3, out=o2)
    assert (bs1, tot1, per1.tolist()) == (bs2, tot2, per2.tolist())
    for k in o1:
        assert torch.equal(o1[k][:tot1], o2[k][:tot1]), k
    pin = lambda t: t.cpu().pin_memory()
    href = [pin(t) for t in dref]; hpl = [pin(t) for t in dpl]
    for resident in (False, True):
        for rep in range(2):
            dpl3 = [torch.zeros_like(t) for t in dpl]
            dref3 = [t.clone() for t in dref] if resident else [torch.zeros_like(t) for t in dref]
            torch.cuda.synchronize()
            cv.upload_packed_begin(lens, None if resident else href, dref3, hpl, dpl3)
            _, per3, tot3, bs3 = cv.bin_sample_packed(dref3, dpl3, lens, pos0, is_auto, 100, -1, 3, out=o3)
            assert (bs1, tot1, per1.tolist()) == (bs3, tot3, per3.tolist())
            for k in o1:
                assert torch.equal(o1[k][:tot1], o3[k][:tot1]), (k, resident, rep)
    cv.upload_genome_wait()


def test_packed_pipeline_equals_pipeline():
    import torch
    cv = get_canvas()
    lengths = [3_000_000, 2_200_000, 1_500_000]
    is_auto = np.array([1, 1, 0], np.uint8)
    thr = synth.poisson_thresholds(0.21)
    data = [synth.generate_chromosome(20260927 + 70, c, L, 0.21, thr) for c, L in enumerate(lengths)]
    bases, hits, masks = _upload(cv, data)
    lens = np.array(lengths, np.int64)
    flags = CLEAN_GCNORM | CLEAN_FILTSIZE | CLEAN_OUTLIERS | CLEAN_LOCALSD
    cap = int(lens.sum() // 100) + 16

    def bufs():
        return _out(cv, cap), torch.empty(cap, dtype=torch.float64, device=cv.device), torch.empty(cap, dtype=torch.int32, device=cv.device), torch.empty(cap, dtype=torch.int32, device=cv.device)

    out1, cov1, st1, seg1 = bufs(); out2, cov2, st2, seg2 = bufs()
    r1 = cv.sample_pipeline(bases, masks, hits, lens, is_auto, out1, cov1, st1, seg1, counts_per_bin=100, bin_size=-1, mode=3, flags=flags)
    dref, dpl, pos0, _ = cv.pack_genome_device(bases, masks, hits, lens)
    r2 = cv.sample_pipeline(dref, None, dpl, lens, is_auto, out2, cov2, st2, seg2, counts_per_bin=100, bin_size=-1, mode=3, flags=flags, pos0=pos0)
    r2b = cv.sample_pipeline(None, None, None, None, None, None, None, None, None, prepared=r2["prepared"])       # the cached call
    cv.synchronize()
    for r in (r2, r2b):
        assert (r["bin_size"], r["total"], r["n_out"], r["nseg"]) == (r1["bin_size"], r1["total"], r1["n_out"], r1["nseg"]) and r["off"].tolist() == r1["off"].tolist()
    n = r1["n_out"]
    for k in out1:
        assert torch.equal(out1[k][:n], out2[k][:n]), k
    assert torch.equal(cov1[:n], cov2[:n]) and torch.equal(st1[:n], st2[:n]) and torch.equal(seg1[:n], seg2[:n])
    assert n > 5_000 and r1["nseg"] >= 3


def test_two_bit_upload_expands_to_the_same_planes_and_bins():
    """canvas_upload_packed2_begin: the hit planes travel in their two-bit wire form (lo / header / extras) and are expanded on the device behind each chromosome's
    transfer; the expanded planes and the bins are those of the four-plane upload (pile-ups and scattered positions with four hits and more included)"""
    import torch
    from canvas_amd.lib import pack_hits2_host
    cv = get_canvas()
    lengths = [900_001, 4096 * 50, 333_333, 70_000]
    data = _chroms(lengths)
    rng = np.random.RandomState(11)
    for b, h, m in data:
        idx = rng.randint(0, len(h), len(h) // 300); h[idx] = rng.randint(4, 40, len(idx)).astype(np.uint8)
    data[1][1][:] = np.minimum(data[1][1], 3)                     # a chromosome without any extras
    lens = np.array(lengths, np.int64)
    href, hpl, pos0, _ = _host_planes(data)
    pin = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).pin_memory()
    parts = [pack_hits2_host(np.ascontiguousarray(h), len(h), threads=2) for b, h, m in data]
    assert parts[1][3] == 0 and parts[0][3] > 100
    h_lo = [pin(p[0]) for p in parts]; h_hdr = [pin(p[1]) for p in parts]; h_ex = [pin(p[2]) for p in parts]; nx = [p[3] for p in parts]
    p_ref = [pin(r) for r in href]
    dref = [torch.zeros(len(r), dtype=torch.int64, device=cv.device) for r in href]
    dpl = [torch.zeros(len(p), dtype=torch.int64, device=cv.device) for p in hpl]
    cap = int(lens.sum() // 50)
    o1 = _out(cv, cap); o2 = _out(cv, cap)
    d1 = [to_dev(r.view(np.int64), cv.device) for r in href]; p1 = [to_dev(p.view(np.int64), cv.device) for p in hpl]
    _, per1, tot1, bs1 = cv.bin_sample_packed(d1, p1, lens, pos0, [1, 1, 1, 0], 100, -1, 3, out=o1)
    for rep in range(2):
        for t in dpl: t.zero_()
        torch.cuda.synchronize()
        cv.upload_packed2_begin(lens, p_ref if rep == 0 else None, dref, h_lo, h_hdr, h_ex, nx, dpl)
        _, per2, tot2, bs2 = cv.bin_sample_packed(dref, dpl, lens, pos0, [1, 1, 1, 0], 100, -1, 3, out=o2)
        assert (bs1, tot1, per1.tolist()) == (bs2, tot2, per2.tolist())
        for k in o1:
            assert torch.equal(o1[k][:tot1], o2[k][:tot1]), (k, rep)
        for c in range(len(lengths)):
            assert (dpl[c].cpu().numpy().view(np.uint64) == hpl[c]).all(), c
    _check_against_oracle(o2, per2, tot2, data, bs2, 3)
