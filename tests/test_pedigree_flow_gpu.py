"""BASELINE configs[3] in memory: a trio that shares the possible mask — multi-sample bin size (MultiSampleHitArrays, CanvasBin.cs:86-110),
CanvasBin + CanvasClean per sample, the bin intersection between CanvasClean and CanvasPartition (Utilities.cs:834-920), PerSampleHMM per
sample, SplitOverlappingSegments across the samples and the per-sample segment ids — every hand-off compared with the chained oracle."""
import numpy as np
import pytest

import oracle_lib as O
from canvas_amd import synth, CLEAN_GCNORM, CLEAN_FILTSIZE, CLEAN_OUTLIERS
from gpu_common import get_canvas, to_dev, pad16

pytestmark = pytest.mark.gpu


def test_trio_bin_clean_merge_partition():
    import ctypes as C
    cv = get_canvas()
    lengths = [2_600_000, 1_900_000, 1_200_000]           # two autosomes and "X"
    nchr = len(lengths); is_auto = np.array([1, 1, 0], np.uint8); is_y = np.zeros(nchr, np.uint8)
    lens = np.array(lengths, np.int64)
    thr = synth.poisson_thresholds(0.21)
    ref = [synth.generate_chromosome(20260927 + 50, c, L, 0.21, thr) for c, L in enumerate(lengths)]
    bases = [r[0] for r in ref]; masks = [r[2] for r in ref]
    rng = np.random.RandomState(50)
    samples_hits = []
    for s in range(3):
        hs = []
        for c, (b, h, m) in enumerate(ref):
            other = synth.generate_chromosome(20260927 + 51 + s, c, lengths[c], 0.21, thr)[1]
            poss = np.unpackbits(m.view(np.uint8), bitorder="little")[: lengths[c]].astype(bool)
            h2 = np.where(poss, other, 0).astype(np.uint8)                   # same mask, the sample's own reads
            if c == 0 and s == 2: h2[900_000:1_300_000] = (h2[900_000:1_300_000] // 2)   # a deletion only the child carries
            hs.append(h2)
        samples_hits.append(hs)
    d_bases = [to_dev(pad16(b), cv.device) for b in bases]
    d_masks = [to_dev(m.view(np.int64), cv.device) for m in masks]
    # ---- multi-sample bin size: median over samples x autosomes
    rates_dev, rates_exp = [], []
    d_hits = []
    for hs in samples_hits:
        dh = [to_dev(pad16(h), cv.device) for h in hs]; d_hits.append(dh)
        _, _, rate = cv.bin_rates(dh, d_masks, lens)
        rates_dev += [rate[c] for c in range(nchr) if is_auto[c]]
        rates_exp += [O.bin_rate(hs[c], masks[c]) for c in range(nchr) if is_auto[c]]
    assert rates_dev == rates_exp
    bin_size = cv.bin_size_from_rates(rates_dev, 100)
    assert bin_size == O.bin_size(rates_exp, 100)
    # ---- bin + clean per sample
    flags = CLEAN_GCNORM | CLEAN_FILTSIZE | CLEAN_OUTLIERS
    dev_clean, exp_clean = [], []
    for s, hs in enumerate(samples_hits):
        out, per, total = cv.bin_genome(d_bases, d_masks, d_hits[s], lens, bin_size, 3)
        res = O.bin_genome(bases, masks, hs, bin_size, mode=3, threads=3)
        e = dict(chr=np.concatenate([np.full(len(res[0][c]), c, np.int32) for c in range(nchr)]), start=np.concatenate(res[0]), stop=np.concatenate(res[1]),
                 gc=np.concatenate(res[2]), count=np.concatenate(res[3]).astype(np.float32))
        assert total == len(e["chr"]) and (out["count"][:total].cpu().numpy() == e["count"]).all()
        ex = O.clean(e["chr"], e["start"], e["stop"], e["count"], e["gc"], is_auto, is_y, flags)
        n_out, _, _ = cv.clean(out, total, is_auto, flags)
        assert n_out == len(ex["chr"]) and (out["count"][:n_out].cpu().numpy().view(np.uint32) == ex["count"].view(np.uint32)).all()
        dev_clean.append((out, n_out)); exp_clean.append(ex)
    # ---- bins every sample still has
    mc, ms, me, mcnt, k = cv.merge_cleaned([o for o, _ in dev_clean], [n for _, n in dev_clean])
    ec, es, ee, ecnt = O.merge_cleaned(exp_clean)
    assert k == len(ec) and (ms.cpu().numpy() == es).all() and (me.cpu().numpy() == ee).all()
    off = cv.chromosome_offsets(mc, k, nchr)
    # ---- PerSampleHMM per sample, then the common segmentation
    seg_starts_dev, seg_starts_exp = [], []
    bs = [es[off[c]:off[c + 1]].astype(np.uint32) for c in range(nchr)]; be = [ee[off[c]:off[c + 1]].astype(np.uint32) for c in range(nchr)]
    states = []
    for s in range(3):
        assert (mcnt[s].cpu().numpy().view(np.uint32) == ecnt[s].view(np.uint32)).all()
        cov = cv.quantize_f2(mcnt[s], k)
        cov_exp = np.array([float(O.format_f2(float(v))) for v in ecnt[s][:2000]])
        assert (cov[:2000].cpu().numpy() == cov_exp).all()
        st = cv.hmm_per_sample(cov, off)
        states.append(st)
        covh = cov.cpu().numpy()
        per = [np.ascontiguousarray(covh[off[c]:off[c + 1]]) for c in range(nchr)]
        paths, ran = O.hmm_genome_per_sample(per, threads=3)
        got = st.cpu().numpy()
        for c in range(nchr):
            assert (got[off[c]:off[c + 1]] == (paths[c] if ran[c] else -1)).all()
        seg_starts_exp.append([O.segments_from_path(paths[c], ran[c], bs[c], be[c]) for c in range(nchr)])
    # SplitOverlappingSegments per chromosome (GenomeSegmentationResults.cs:18-55) through the product's host entry point
    merged_starts = []
    for c in range(nchr):
        st_list = [seg_starts_exp[s][c][0] for s in range(3)]; en_list = [seg_starts_exp[s][c][1] for s in range(3)]
        xs, xe = O.split_overlapping(st_list, en_list)
        ns = np.array([len(a) for a in st_list], np.int32); cap = 2 * int(ns.sum()) + 2
        os_ = np.zeros(cap, np.uint32); oe = np.zeros(cap, np.uint32); nout = C.c_int32(0)
        ps = (C.c_void_p * 3)(*[a.ctypes.data for a in st_list]); pe = (C.c_void_p * 3)(*[a.ctypes.data for a in en_list])
        assert cv.lib.canvas_split_overlapping(3, ps, pe, ns.ctypes.data_as(C.c_void_p), os_.ctypes.data_as(C.c_void_p), oe.ctypes.data_as(C.c_void_p), cap, C.byref(nout)) == 0
        assert nout.value == len(xs) and (os_[:len(xs)] == xs).all() and (oe[:len(xs)] == xe).all()
        merged_starts.append(xs)
    ids, last = O.postprocess(bs, be, merged_starts, None, 1000000)
    assert last + 1 >= 3          # the child's deletion and the chromosome starts
