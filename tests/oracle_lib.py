"""ctypes binding of the CPU oracle (oracle/libcanvas_oracle.so).  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by canvas_amd/."""
import ctypes as C
import os
import subprocess
import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "libcanvas_oracle.so")

CLEAN_GCNORM, CLEAN_FILTSIZE, CLEAN_OUTLIERS, CLEAN_LOCALSD, CLEAN_LOESS = 1, 2, 4, 8, 16


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "oracle")])


def _load():
    if not os.path.exists(_SO):
        build()
    lib = C.CDLL(_SO)
    lib.orc_bin_rate.restype = C.c_double
    lib.orc_bin_chromosome.restype = C.c_int64
    lib.orc_bin_chromosome_weighted.restype = C.c_int64
    lib.orc_clean.restype = C.c_int64
    lib.orc_median_f32.restype = C.c_float
    lib.orc_golden_section_square.restype = C.c_double
    lib.orc_golden_section_square.argtypes = [C.c_double, C.c_double]
    lib.orc_nu.restype = C.c_double
    lib.orc_nu.argtypes = [C.c_double, C.c_double]
    lib.orc_tailp.restype = C.c_double
    lib.orc_tailp.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int, C.c_double]
    lib.orc_htmaxp.restype = C.c_double
    lib.orc_tmaxp.restype = C.c_double
    lib.orc_phyper.restype = C.c_double
    lib.orc_phyper.argtypes = [C.c_double] * 4
    lib.orc_trimmed_variance.restype = C.c_double
    return lib


lib = _load()


def _p(a, t=None):
    return a.ctypes.data_as(C.c_void_p)


def _pp(arrs):
    """array of pointers to numpy arrays"""
    T = C.c_void_p * len(arrs)
    return T(*[a.ctypes.data for a in arrs])


def bin_rate(hits, mask):
    return lib.orc_bin_rate(_p(hits), _p(mask), C.c_int64(len(hits)))


def bin_size(rates, counts_per_bin):
    r = np.ascontiguousarray(rates, dtype=np.float64)
    return lib.orc_bin_size(_p(r), len(r), counts_per_bin)


def bin_chromosome(bases, mask, hits, bin_size, mode=3):
    L = len(bases)
    cap = L // max(1, bin_size) + 2
    out = [np.zeros(cap, np.int32) for _ in range(4)]
    n = lib.orc_bin_chromosome(_p(bases), _p(mask), _p(hits), C.c_int64(L), bin_size, mode, C.c_int64(cap), *[_p(o) for o in out])
    return [o[:n].copy() for o in out]


def bin_predefined(bases, mask, hits, bin_start, bin_stop, mode=3):
    """BinCountsForChromosome with predefined bins (CanvasBin -n): (bins closed, gc, count); bins never closed keep gc = count = 0"""
    bs = np.ascontiguousarray(bin_start, np.int32); be = np.ascontiguousarray(bin_stop, np.int32)
    gc = np.zeros(len(bs), np.int32); cnt = np.zeros(len(bs), np.int32)
    lib.orc_bin_chromosome_predefined.restype = C.c_int64
    k = lib.orc_bin_chromosome_predefined(_p(bases), _p(mask), _p(hits), C.c_int64(len(bases)), int(mode), C.c_int64(len(bs)), _p(bs), _p(be), _p(gc), _p(cnt))
    return int(k), gc, cnt


def bin_predefined_gc_weighted(bases, masks, hits, fraglens, bin_starts, bin_stops):
    """CanvasBin -n -m GCContentWeighted: the profile and the weights from every chromosome (BinCounts, CanvasBin.cs:427-505), then the predefined bins of the chromosomes that
    have any (bin_starts[c] may be empty); returns per chromosome (bins closed, gc, count)"""
    nchr = len(bases)
    lens = np.array([len(b) for b in bases], np.int64)
    m = lib.orc_mean_fragment_size(nchr, _pp(fraglens), _p(lens))
    rgc = []
    for c in range(nchr):
        g = np.zeros(len(bases[c]), np.uint8)
        lib.orc_read_gc_content(_p(bases[c]), _p(fraglens[c]), C.c_int64(len(bases[c])), m, _p(g))
        rgc.append(g)
    w = np.zeros(101, np.float32)
    lib.orc_observed_vs_expected_gc(nchr, _pp(rgc), _pp(hits), _p(lens), _p(w))
    lib.orc_bin_chromosome_predefined_weighted.restype = C.c_int64
    res = []
    for c in range(nchr):
        bs = np.ascontiguousarray(bin_starts[c], np.int32); be = np.ascontiguousarray(bin_stops[c], np.int32)
        gc = np.zeros(len(bs), np.int32); cnt = np.zeros(len(bs), np.int32)
        k = lib.orc_bin_chromosome_predefined_weighted(_p(bases[c]), _p(masks[c]), _p(hits[c]), _p(rgc[c]), _p(w), C.c_int64(len(bases[c])), C.c_int64(len(bs)), _p(bs), _p(be), _p(gc), _p(cnt))
        res.append((int(k), gc, cnt))
    return res


def bin_gc_weighted(bases, masks, hits, fraglens, bin_size):
    """GCContentWeighted binning of a genome (CanvasBin.cs:416-506,626-636): returns (per-chromosome [start,stop,gc,count], mean fragment, weights)"""
    nchr = len(bases)
    lens = np.array([len(b) for b in bases], np.int64)
    m = lib.orc_mean_fragment_size(nchr, _pp(fraglens), _p(lens))
    rgc = []
    for c in range(nchr):
        g = np.zeros(len(bases[c]), np.uint8)
        lib.orc_read_gc_content(_p(bases[c]), _p(fraglens[c]), C.c_int64(len(bases[c])), m, _p(g))
        rgc.append(g)
    w = np.zeros(101, np.float32)
    lib.orc_observed_vs_expected_gc(nchr, _pp(rgc), _pp(hits), _p(lens), _p(w))
    res = []
    for c in range(nchr):
        L = len(bases[c]); cap = L // max(1, bin_size) + 2
        out = [np.zeros(cap, np.int32) for _ in range(4)]
        n = lib.orc_bin_chromosome_weighted(_p(bases[c]), _p(masks[c]), _p(hits[c]), _p(rgc[c]), _p(w), C.c_int64(L), bin_size, C.c_int64(cap), *[_p(o) for o in out])
        res.append([o[:n].copy() for o in out])
    return res, m, w, rgc


def bin_genome(bases, masks, hits, bin_size, mode=3, threads=1):
    nchr = len(bases)
    lens = np.array([len(b) for b in bases], np.int64)
    caps = lens // max(1, bin_size) + 2
    outs = [[np.zeros(int(c), np.int32) for c in caps] for _ in range(4)]
    nb = np.zeros(nchr, np.int64)
    lib.orc_bin_genome(nchr, _pp(bases), _pp(masks), _pp(hits), _p(lens), bin_size, mode, _p(caps),
                       _pp(outs[0]), _pp(outs[1]), _pp(outs[2]), _pp(outs[3]), _p(nb), threads)
    return [[outs[k][c][:nb[c]] for c in range(nchr)] for k in range(4)]


def bin_rates_genome(masks, hits, threads=1):
    nchr = len(masks)
    lens = np.array([len(h) for h in hits], np.int64)
    rates = np.zeros(nchr, np.float64)
    lib.orc_bin_rates_genome(nchr, _pp(masks), _pp(hits), _p(lens), _p(rates), threads)
    return rates


def clean(chr_id, start, stop, count, gc, is_auto, is_y, flags, min_bins_weighted=100):
    chr_id, start, stop, gc = [np.array(a, np.int32) for a in (chr_id, start, stop, gc)]
    count = np.array(count, np.float32)
    is_auto = np.ascontiguousarray(is_auto, np.uint8)
    is_y = np.ascontiguousarray(is_y, np.uint8)
    local_sd = C.c_double(-1.0)
    stages = np.zeros(8, np.int32)
    n = lib.orc_clean(C.c_int64(len(chr_id)), _p(chr_id), _p(start), _p(stop), _p(count), _p(gc), len(is_auto), _p(is_auto), _p(is_y),
                      C.c_uint32(flags), min_bins_weighted, C.byref(local_sd), _p(stages))
    return dict(chr=chr_id[:n], start=start[:n], stop=stop[:n], count=count[:n], gc=gc[:n], local_sd=local_sd.value, stages=stages)


def merge_cleaned(samples):
    """Utilities.MergeMultiSampleCleanedBedFile (Utilities.cs:834-920); samples = list of dicts with chr/start/stop/count arrays"""
    S = len(samples)
    n = np.array([len(s["chr"]) for s in samples], np.int64)
    cap = int(n.sum()) + 1
    chrs = [np.ascontiguousarray(s["chr"], np.int32) for s in samples]; st = [np.ascontiguousarray(s["start"], np.int32) for s in samples]
    en = [np.ascontiguousarray(s["stop"], np.int32) for s in samples]; cnt = [np.ascontiguousarray(s["count"], np.float32) for s in samples]
    oc = np.zeros(cap, np.int32); os_ = np.zeros(cap, np.int32); oe = np.zeros(cap, np.int32); ocnt = [np.zeros(cap, np.float32) for _ in range(S)]
    lib.orc_merge_cleaned.restype = C.c_int64
    k = lib.orc_merge_cleaned(S, _p(n), _pp(chrs), _pp(st), _pp(en), _pp(cnt), _p(oc), _p(os_), _p(oe), _pp(ocnt), C.c_int64(cap))
    return oc[:k].copy(), os_[:k].copy(), oe[:k].copy(), [c[:k].copy() for c in ocnt]


def _fmt(fn, v):
    buf = C.create_string_buffer(64)
    fn(v, buf, 64)
    return buf.value.decode()


def format_f2(v):
    return _fmt(lib.orc_format_f2, C.c_float(v))


def format_g15(v):
    return _fmt(lib.orc_format_g15, C.c_double(v))


def format_g7(v):
    return _fmt(lib.orc_format_g7, C.c_float(v))


def quartiles(x):
    x = np.ascontiguousarray(x, np.float32)
    out = np.zeros(3, np.float32)
    lib.orc_quartiles(_p(x), len(x), _p(out))
    return out


def median_f32(x):
    x = np.ascontiguousarray(x, np.float32)
    return lib.orc_median_f32(_p(x), len(x))


def loess_fit(x, y, bandwidth, rob_iters, x_step):
    x = np.ascontiguousarray(x, np.float64); y = np.ascontiguousarray(y, np.float64)
    fitted = np.zeros_like(x); pred = np.zeros_like(x)
    lib.orc_loess_fit(_p(x), _p(y), len(x), C.c_double(bandwidth), rob_iters, C.c_double(x_step), _p(fitted), _p(pred))
    return fitted, pred


def negbin(mean, variance, max_value):
    out = np.zeros(max_value, np.float64)
    lib.orc_negbin(C.c_double(mean), C.c_double(variance), max_value, _p(out))
    return out


def genotype_combos(n_states, cur):
    out = np.zeros(4096, np.int32)
    n = lib.orc_genotype_combos(n_states, cur, _p(out), 4096)
    w = min(n_states, 4)
    return out[: n * w].reshape(n, w).tolist()


def hmm_global_params(cov):
    n = np.array([len(c) for c in cov], np.int64)
    med = C.c_double(); pv = C.c_double()
    lib.orc_hmm_global_params(len(cov), _pp(cov), _p(n), C.byref(med), C.byref(pv))
    return med.value, pv.value


def hmm_chromosome(cov_samples, per_sample, medians=None, pvs=None):
    S = len(cov_samples); T = len(cov_samples[0])
    path = np.full(T, -9, np.int32)
    m = np.ascontiguousarray(medians if medians is not None else np.zeros(S), np.float64)
    v = np.ascontiguousarray(pvs if pvs is not None else np.zeros(S), np.float64)
    ran = lib.orc_hmm_chromosome(S, int(per_sample), _pp(cov_samples), T, _p(m) if medians is not None else None,
                                 _p(v) if pvs is not None else None, _p(path))
    return ran, path


def hmm_genome_per_sample(cov, threads=1):
    n = np.array([len(c) for c in cov], np.int64)
    paths = [np.full(len(c), -9, np.int32) for c in cov]
    ran = np.zeros(len(cov), np.int32)
    lib.orc_hmm_genome_per_sample(len(cov), _pp(cov), _p(n), _pp(paths), _p(ran), threads)
    return paths, ran


def segments_from_path(path, ran, start, end):
    T = len(path)
    ss = np.zeros(T + 1, np.uint32); se = np.zeros(T + 1, np.uint32)
    n = lib.orc_segments_from_path(_p(path), T, int(ran), _p(start), _p(end), _p(ss), _p(se))
    return ss[:n].copy(), se[:n].copy()


def split_overlapping(starts, ends):
    S = len(starts)
    nseg = np.array([len(s) for s in starts], np.int32)
    cap = 2 * int(nseg.sum()) + 2
    os_ = np.zeros(cap, np.uint32); oe = np.zeros(cap, np.uint32)
    n = lib.orc_split_overlapping(S, _pp(starts), _pp(ends), _p(nseg), _p(os_), _p(oe), cap)
    return os_[:n].copy(), oe[:n].copy()


def is_uniform_reference_ploidy(q_start, q_end, ivs):
    """PloidyInfo.IsUniformReferencePloidy for the one-based interval [q_start, q_end]; ivs = (starts, ends, cns) of the chromosome"""
    a = np.ascontiguousarray(ivs[0], np.int32); b = np.ascontiguousarray(ivs[1], np.int32); c = np.ascontiguousarray(ivs[2], np.int32)
    return lib.orc_is_uniform_reference_ploidy(C.c_int64(q_start), C.c_int64(q_end), len(a), _p(a), _p(b), _p(c))


def postprocess_ploidy(bin_start, bin_end, seg_start, excl, ploidy, max_inter_bin_dist=1000000):
    """PostProcessSegments with a reference ploidy: ploidy[c] = None (chromosome not in the VCF) or (starts, ends, cns), one-based"""
    nchr = len(bin_start)
    nb = np.array([len(b) for b in bin_start], np.int64)
    nseg = np.array([len(s) for s in seg_start], np.int32)
    seg_id = [np.zeros(len(b), np.int32) for b in bin_start]
    z = np.zeros(0, np.int32)
    excl = excl if excl is not None else [(z, z)] * nchr
    es = [np.ascontiguousarray(e[0], np.int32) for e in excl]; ee = [np.ascontiguousarray(e[1], np.int32) for e in excl]
    ne = np.array([len(e) for e in es], np.int32)
    pl = [p if p is not None else (z, z, z) for p in ploidy]
    ps = [np.ascontiguousarray(q[0], np.int32) for q in pl]; pe = [np.ascontiguousarray(q[1], np.int32) for q in pl]; pc = [np.ascontiguousarray(q[2], np.int32) for q in pl]
    npl = np.array([-1 if p is None else len(p[0]) for p in ploidy], np.int32)
    last = lib.orc_postprocess_ploidy(nchr, _p(nb), _pp(bin_start), _pp(bin_end), _p(nseg), _pp(seg_start), _p(ne), _pp(es), _pp(ee), max_inter_bin_dist,
                                      _p(npl), _pp(ps), _pp(pe), _pp(pc), _pp(seg_id))
    return seg_id, last


def evenness_score(per_chr_cov, window=100000):
    """SegmentationInput.GetEvennessScore (Segmentation.cs:260-296); None when the reference would throw (no metric file written)"""
    n = np.array([len(c) for c in per_chr_cov], np.int64)
    out = C.c_double(0)
    rc = lib.orc_evenness_score(len(per_chr_cov), _pp(per_chr_cov), _p(n), int(window), C.byref(out))
    return None if rc else out.value


def evenness_window_scores(per_chr_cov, window):
    n = np.array([len(c) for c in per_chr_cov], np.int64)
    cap = int(sum(max(0, len(c) // max(1, window)) for c in per_chr_cov)) + 8
    out = np.zeros(cap, np.float64)
    k = lib.orc_evenness_window_scores(len(per_chr_cov), _pp(per_chr_cov), _p(n), int(window), _p(out), cap)
    return out[:k].copy()


def postprocess(bin_start, bin_end, seg_start, excl=None, max_inter_bin_dist=1000000):
    nchr = len(bin_start)
    nb = np.array([len(b) for b in bin_start], np.int64)
    nseg = np.array([len(s) for s in seg_start], np.int32)
    seg_id = [np.zeros(len(b), np.int32) for b in bin_start]
    if excl is None:
        last = lib.orc_postprocess(nchr, _p(nb), _pp(bin_start), _pp(bin_end), _p(nseg), _pp(seg_start), None, None, None,
                                   max_inter_bin_dist, _pp(seg_id))
    else:
        es = [np.ascontiguousarray(e[0], np.int32) for e in excl]; ee = [np.ascontiguousarray(e[1], np.int32) for e in excl]
        ne = np.array([len(e) for e in es], np.int32)
        last = lib.orc_postprocess(nchr, _p(nb), _pp(bin_start), _pp(bin_end), _p(nseg), _pp(seg_start), _p(ne), _pp(es), _pp(ee),
                                   max_inter_bin_dist, _pp(seg_id))
    return seg_id, last


_SBDRY = {}


def cbs_boundary(n_perm=10000, alpha=0.01, eta=0.05):
    key = (n_perm, alpha, eta)
    if key not in _SBDRY:
        cap = 8192 * 8
        out = np.zeros(cap, np.uint32)
        n = lib.orc_cbs_boundary(C.c_uint32(n_perm), C.c_double(alpha), C.c_double(eta), _p(out), cap)
        _SBDRY[key] = out[:n].copy()
    return _SBDRY[key]


def tmaxo(x, al0=2):
    x = np.ascontiguousarray(x, np.float64)
    tss = float(np.sum(x * x))
    sx = np.zeros_like(x); iseg = np.zeros(2, np.int32); ostat = C.c_double()
    lib.orc_tmaxo(_p(x), len(x), C.c_double(tss), _p(sx), _p(iseg), C.byref(ostat), al0)
    return ostat.value, iseg, sx


def htmaxp(px, tss, k=25, al0=2):
    px = np.ascontiguousarray(px, np.float64); sx = np.zeros_like(px)
    return lib.orc_htmaxp(k, C.c_double(tss), _p(px), len(px), _p(sx), al0)


def tmaxp(px, tss, al0=2):
    px = np.ascontiguousarray(px, np.float64); sx = np.zeros_like(px)
    return lib.orc_tmaxp(C.c_double(tss), _p(px), len(px), _p(sx), al0)


def mt_u32(seed, n):
    out = np.zeros(n, np.uint32)
    lib.orc_mt_u32(C.c_uint32(seed), n, _p(out))
    return out


def cbs_seeds(nchr):
    s = np.zeros(nchr, np.int32)
    lib.orc_cbs_seeds(nchr, _p(s))
    return s


def xperm(x, seed, skip=0):
    x = np.ascontiguousarray(x, np.float64); px = np.zeros_like(x)
    lib.orc_xperm(_p(x), _p(px), len(x), C.c_uint32(seed & 0xFFFFFFFF), skip)
    return px


def cbs_chromosome(x, seed, sbdry=None, alpha=0.01, n_perm=10000, undo=0, trimmed_sd=1.0):
    x = np.ascontiguousarray(x, np.float64)
    sb = cbs_boundary(n_perm, alpha) if sbdry is None else sbdry
    cap = len(x) + 1
    ls = np.zeros(cap, np.int32); stats = np.zeros(7, np.int64)
    n = lib.orc_cbs_chromosome(_p(x), len(x), C.c_int32(int(seed)), _p(sb), len(sb), C.c_double(alpha), C.c_uint32(n_perm), undo,
                               C.c_double(trimmed_sd), _p(ls), cap, _p(stats))
    return ls[:n].copy(), stats


def changepoints_prune(x, length_seg, cutoff=0.05):
    """ChangePoint.ChangePointsPrune (ChangePoint.cs:205-271)"""
    x = np.ascontiguousarray(x, np.float64); ls = np.ascontiguousarray(length_seg, np.int32)
    out = np.zeros(len(ls) + 1, np.int32)
    k = lib.orc_changepoints_prune(_p(x), len(x), _p(ls), len(ls), C.c_double(cutoff), _p(out), len(out))
    return out[:k].copy()


def cbs_genome(xs, alpha=0.01, n_perm=10000, threads=1, undo=0):
    sb = cbs_boundary(n_perm, alpha)
    n = np.array([len(x) for x in xs], np.int64)
    caps = np.array([len(x) + 1 for x in xs], np.int32)
    ls = [np.zeros(int(c), np.int32) for c in caps]
    nseg = np.zeros(len(xs), np.int32); stats = np.zeros(7, np.int64)
    lib.orc_cbs_genome_undo(len(xs), _pp(xs), _p(n), _p(sb), len(sb), C.c_double(alpha), C.c_uint32(n_perm), undo, _pp(ls), _p(caps), _p(nseg), _p(stats), threads)
    return [l[:k].copy() for l, k in zip(ls, nseg)], stats


# ---- Wavelets (oracle_wavelets.cpp)
lib.orc_haar_wavelets.restype = C.c_int64
lib.orc_wavelets.restype = C.c_int64


def haar_wavelets(x, thr_lower, thr_upper, is_germline, mad_factor, cv, f3):
    """WaveletSegmentation.HaarWavelets (WaveletSegmentation.cs:373-425); cv None = no coverage variability (MAD is used)"""
    x = np.ascontiguousarray(x, np.float64); f3 = np.ascontiguousarray(f3, np.float64)
    out = np.zeros(len(x) + 1, np.int32)
    k = lib.orc_haar_wavelets(_p(x), C.c_int64(len(x)), C.c_double(thr_lower), C.c_double(thr_upper), int(bool(is_germline)), C.c_double(mad_factor),
                              0 if cv is None else 1, C.c_double(0.0 if cv is None else cv), _p(f3), len(f3), _p(out), C.c_int64(len(out)))
    assert k >= 0
    return out[:k].copy()


def coverage_variability(window, per_chr):
    """SegmentationInput.GetCoverageVariability (Segmentation.cs:308-328); None when there are fewer than 10 windows of data"""
    cov = np.ascontiguousarray(np.concatenate(per_chr), np.float64)
    off = np.concatenate([[0], np.cumsum([len(a) for a in per_chr])]).astype(np.int64)
    cv = C.c_double(0)
    return cv.value if lib.orc_coverage_variability(int(window), len(per_chr), _p(cov), _p(off), C.byref(cv)) else None


def factor_of_three(per_chr):
    """SegmentationInput.FactorOfThreeCoverageVariabilities (Segmentation.cs:366-402)"""
    cov = np.ascontiguousarray(np.concatenate(per_chr), np.float64)
    off = np.concatenate([[0], np.cumsum([len(a) for a in per_chr])]).astype(np.int64)
    out = np.zeros(9, np.float64)
    k = lib.orc_factor_of_three(len(per_chr), _p(cov), _p(off), _p(out))
    return out[:k].copy()


def wavelets_genome(per_chr, is_germline=False, thr_lower=0.05, thr_upper=80.0, mad_factor=5.0, window=100000, min_size=10, threads=1):
    """WaveletsRunner.Run up to the breakpoints (WaveletsRunner.cs:52-150): list of breakpoint arrays, one per chromosome; threads > 1: one task per chromosome
    (the reference's Parallel.ForEach, WaveletsRunner.cs:89-135), same results"""
    cov = np.ascontiguousarray(np.concatenate(per_chr), np.float64)
    off = np.concatenate([[0], np.cumsum([len(a) for a in per_chr])]).astype(np.int64)
    out = np.zeros(len(cov) + len(per_chr) + 1, np.int32); oo = np.zeros(len(per_chr) + 1, np.int64)
    if threads > 1:
        k = lib.orc_wavelets_threads(len(per_chr), _p(cov), _p(off), int(bool(is_germline)), C.c_double(thr_lower), C.c_double(thr_upper), C.c_double(mad_factor), int(window),
                                     int(min_size), _p(out), C.c_int64(len(out)), _p(oo), int(threads))
    else:
        k = lib.orc_wavelets(len(per_chr), _p(cov), _p(off), int(bool(is_germline)), C.c_double(thr_lower), C.c_double(thr_upper), C.c_double(mad_factor), int(window),
                             int(min_size), _p(out), C.c_int64(len(out)), _p(oo))
    assert k >= 0
    return [out[oo[c]:oo[c + 1]].copy() for c in range(len(per_chr))]


# ---- CanvasNormalize (oracle_normalize.cpp)
lib.orc_norm_ratio.restype = C.c_int64
lib.orc_wavelets_threads.restype = C.c_int64


def norm_weighted_reference(counts, on_idx=None):
    counts = [np.ascontiguousarray(c, np.float64) for c in counts]
    n = len(counts[0]); out = np.zeros(n, np.float64); w = np.zeros(len(counts), np.float64)
    oi = None if on_idx is None else np.ascontiguousarray(on_idx, np.int32)
    lib.orc_norm_weighted_reference(len(counts), _pp(counts), C.c_int64(n), None if oi is None else _p(oi), C.c_int64(0 if oi is None else len(oi)), _p(out), _p(w))
    return out, w


def norm_ratio(sample, reference, on_idx=None, mode=0, min_ref=1.0, max_ref=float("inf"), ploidy=None):
    s = np.ascontiguousarray(sample, np.float32); r = np.ascontiguousarray(reference, np.float32); n = len(s)
    oi = None if on_idx is None else np.ascontiguousarray(on_idx, np.int32); pl = None if ploidy is None else np.ascontiguousarray(ploidy, np.int32)
    keep = np.zeros(n, np.int32); ratio = np.zeros(n, np.float32); count = np.zeros(n, np.float32)
    k = lib.orc_norm_ratio(C.c_int64(n), _p(s), _p(r), None if oi is None else _p(oi), C.c_int64(0 if oi is None else len(oi)), int(mode), C.c_double(min_ref), C.c_double(max_ref),
                           None if pl is None else _p(pl), _p(keep), _p(ratio), _p(count))
    return keep[:k].copy(), ratio[:k].copy(), count[:k].copy()
