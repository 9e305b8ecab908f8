"""TEST INFRASTRUCTURE ONLY: whole flows chained on the CPU oracle (tests/, smoke() and bench.py's cpu_baseline leg)."""
import numpy as np

import oracle_lib as O


def f2_float(count):
    """the float CanvasClean reads back from a "{count:F2}" column (IO.cs:21 -> float.Parse, IO.cs:40)"""
    return np.array([np.float32(float(O.format_f2(float(v)))) for v in count], np.float32)


def f2_double(count):
    """the double CanvasPartition reads from the cleaned file's F2 column (CanvasSegment.cs:1146)"""
    return np.array([float(O.format_f2(float(v))) for v in count], np.float64)


def tumor_normal(bases, masks, hits_t, fraglen_t, hits_n, is_autosome, clean_flags, alpha=0.01, nperm=10000, counts_per_bin=100, threads=8):
    """BASELINE configs[4] on the oracle: tumour GCContentWeighted bins (bin size from the tumour's autosome rates) + normal TruncatedDynamicRange bins of the
    same size -> LSNorm ratio x 40 -> F2 file -> CanvasClean -> F2 file -> CBS.  Mirrors canvas_amd.Canvas.tumor_normal_flow hand-off by hand-off."""
    nchr = len(bases)
    rates = O.bin_rates_genome(masks, hits_t, threads=threads)
    bs = O.bin_size([r for r, a in zip(rates, is_autosome) if a], counts_per_bin)
    tb, mfrag, w, _ = O.bin_gc_weighted(bases, masks, hits_t, fraglen_t, bs)
    nb = O.bin_genome(bases, masks, hits_n, bs, 3, threads=threads)
    T = dict(chr=np.concatenate([np.full(len(tb[c][0]), c, np.int32) for c in range(nchr)]), start=np.concatenate([tb[c][0] for c in range(nchr)]),
             stop=np.concatenate([tb[c][1] for c in range(nchr)]), gc=np.concatenate([tb[c][2] for c in range(nchr)]),
             count=np.concatenate([tb[c][3] for c in range(nchr)]).astype(np.float32))
    ncount = np.concatenate(nb[3]).astype(np.float32)
    assert (np.concatenate(nb[1]) == T["stop"]).all()
    keep, ratio, count = O.norm_ratio(T["count"], ncount, None, mode=0)
    R = dict(chr=T["chr"][keep], start=T["start"][keep], stop=T["stop"][keep], gc=T["gc"][keep], count=f2_float(count))
    is_y = np.zeros(nchr, np.uint8)
    ex = O.clean(R["chr"], R["start"], R["stop"], R["count"], R["gc"], is_autosome, is_y, clean_flags)
    cov = f2_double(ex["count"])
    off = np.concatenate([[0], np.cumsum(np.bincount(ex["chr"], minlength=nchr))]).astype(np.int64)
    per = [np.ascontiguousarray(cov[off[c]:off[c + 1]]) for c in range(nchr)]
    seg, stats = O.cbs_genome(per, alpha, nperm, threads=threads)
    return dict(bin_size=bs, tumour=T, normal_count=ncount, keep_idx=keep, ratio=ratio, ratio_count=count, to_clean=R, cleaned=ex, cov=cov, chr_offset=off, seg_len=seg, cbs_stats=stats,
                mean_fragment=mfrag)


def germline_single(bases, masks, hits, is_autosome, names, counts_per_bin=100, threads=8, alpha=0.01, nperm=10000):
    """BASELINE configs[0] plumbing on the oracle: CanvasBin -d 100 -m TruncatedDynamicRange -> CanvasClean -g -s -r --local-sd-metric-file -> F2 ->
    CanvasPartition (PerSampleHMM and CBS), rows of the three files as text."""
    nchr = len(bases)
    rates = O.bin_rates_genome(masks, hits, threads=threads)
    bs = O.bin_size([r for r, a in zip(rates, is_autosome) if a], counts_per_bin)
    st, en, gc, cnt = O.bin_genome(bases, masks, hits, bs, 3, threads=threads)
    B = dict(chr=np.concatenate([np.full(len(st[c]), c, np.int32) for c in range(nchr)]), start=np.concatenate(st), stop=np.concatenate(en), gc=np.concatenate(gc),
             count=np.concatenate(cnt).astype(np.float32))
    flags = O.CLEAN_GCNORM | O.CLEAN_FILTSIZE | O.CLEAN_OUTLIERS | O.CLEAN_LOCALSD
    ex = O.clean(B["chr"], B["start"], B["stop"], B["count"], B["gc"], is_autosome, np.zeros(nchr, np.uint8), flags)
    cov = f2_double(ex["count"])
    off = np.concatenate([[0], np.cumsum(np.bincount(ex["chr"], minlength=nchr))]).astype(np.int64)
    per = [np.ascontiguousarray(cov[off[c]:off[c + 1]]) for c in range(nchr)]
    bsr = [ex["start"][off[c]:off[c + 1]].astype(np.uint32) for c in range(nchr)]; ber = [ex["stop"][off[c]:off[c + 1]].astype(np.uint32) for c in range(nchr)]
    paths, ran = O.hmm_genome_per_sample(per, threads=threads)
    hmm_starts = [O.segments_from_path(paths[c], ran[c], bsr[c], ber[c])[0] for c in range(nchr)]
    hmm_ids, _ = O.postprocess(bsr, ber, hmm_starts)
    seg, stats = O.cbs_genome(per, alpha, nperm, threads=threads)
    cbs_starts = []
    for c in range(nchr):
        pos = np.concatenate([[0], np.cumsum(seg[c])[:-1]]).astype(np.int64) if len(seg[c]) else np.zeros(0, np.int64)
        cbs_starts.append(bsr[c][pos].astype(np.uint32) if len(pos) else np.zeros(0, np.uint32))
    cbs_ids, _ = O.postprocess(bsr, ber, cbs_starts)
    binned_rows = [f"{names[c]}\t{s}\t{e}\t{O.format_f2(float(n))}\t{g}" for c, s, e, n, g in zip(B["chr"], B["start"], B["stop"], B["count"], B["gc"])]
    cleaned_rows = [f"{names[c]}\t{s}\t{e}\t{O.format_f2(float(n))}\t{g}" for c, s, e, n, g in zip(ex["chr"], ex["start"], ex["stop"], ex["count"], ex["gc"])]
    part = lambda ids: [f"{names[c]}\t{s_}\t{e_}\t{O.format_g15(float(v))}\t{i}" for c in range(nchr) for s_, e_, v, i in zip(bsr[c], ber[c], per[c], ids[c])]
    return dict(bin_size=bs, binned=B, cleaned=ex, cov=cov, chr_offset=off, paths=paths, ran=ran, seg_len=seg, cbs_stats=stats, binned_rows=binned_rows, cleaned_rows=cleaned_rows,
                partitioned_hmm_rows=part(hmm_ids), partitioned_cbs_rows=part(cbs_ids), hmm_ids=hmm_ids, cbs_ids=cbs_ids)
