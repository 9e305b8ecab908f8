"""ctypes binding of libcanvas_hip.so.  Device memory is handled with torch tensors (plumbing only)."""
import sys
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(HERE, "libcanvas_hip.so")
_SYNTH_SO = os.path.join(HERE, "libcanvas_synth.so")

MODE_BINARY, MODE_TDR, MODE_GCW = 0, 3, 5
CLEAN_GCNORM, CLEAN_FILTSIZE, CLEAN_OUTLIERS, CLEAN_LOCALSD, CLEAN_LOESS = 1, 2, 4, 8, 16

# every symbol include/canvas_hip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "canvas_create", "canvas_destroy", "canvas_last_error", "canvas_version", "canvas_set_stream", "canvas_set_one_shot", "canvas_synchronize",
    "canvas_device_malloc", "canvas_device_free", "canvas_memcpy_h2d", "canvas_memcpy_d2h", "canvas_host_register", "canvas_host_unregister", "canvas_upload_genome_begin", "canvas_upload_genome_wait",
    "canvas_packed_plane_bytes", "canvas_pack_reference_host", "canvas_pack_hits_host", "canvas_pack_genome_device", "canvas_upload_packed_begin", "canvas_bin_sample_packed", "canvas_sample_pipeline_packed", "canvas_pack_hits2_host", "canvas_upload_packed2_begin",
    "canvas_mask_from_fasta", "canvas_mask_exclude_intervals", "canvas_screen_hits",
    "canvas_bin_rates", "canvas_bin_size_from_rates", "canvas_bin_count_upper_bound", "canvas_bin_genome", "canvas_bin_sample", "canvas_bin_sample_gcweighted", "canvas_bin_predefined", "canvas_bin_predefined_gcweighted",
    "canvas_clean", "canvas_clean2", "canvas_clean_batch", "canvas_merge_cleaned", "canvas_chromosome_offsets", "canvas_quantize_f2", "canvas_hmm_per_sample", "canvas_hmm_joint", "canvas_segment_ids", "canvas_segment_ids_filtered", "canvas_segment_ids_ploidy", "canvas_evenness_score", "canvas_split_overlapping", "canvas_cbs", "canvas_cbs_undo", "canvas_cbs_device_stats", "canvas_cbs_tailp_stats", "canvas_cbs_tail_probe", "canvas_cbs_boundary", "canvas_cbs_seeds", "canvas_cbs_prefetch", "canvas_cbs_stream_read", "canvas_cbs_cache_stats", "canvas_wavelets", "canvas_wavelets_stats", "canvas_wavelets_decisions", "canvas_normalize_reference", "canvas_normalize_ratio", "canvas_sample_pipeline",
    "canvas_comm_unique_id", "canvas_comm_init", "canvas_comm_init_host", "canvas_allgather_boundaries", "canvas_sample_pipeline_sharded", "canvas_sample_pipeline_sharded_packed", "canvas_sharded_stats", "canvas_cbs_sharded", "canvas_wavelets_sharded", "canvas_allgather_host", "canvas_merge_cleaned_sharded", "canvas_profile_enable", "canvas_profile_get", "canvas_bin_gcw_stats", "canvas_cbs_tpermp_stats", "canvas_comm_split", "canvas_comm_restore", "canvas_comm_rank", "canvas_bin_sample_sharded", "canvas_hmm_per_sample_sharded", "canvas_cbs_perm_probe", "canvas_stale_reads",
]


class CanvasError(RuntimeError):
    pass


_lib = None


def load_library():
    """Loads the HIP library; fails loudly if it has not been built (python -m canvas_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch wheels bundle their own HIP runtime (SONAME libamdhip64.so.7): load torch FIRST so that the dynamic loader binds
    # this library to the runtime torch already loaded (one runtime per process, shared device pointers and streams).
    import torch  # noqa: F401
    if not os.path.exists(_SO):
        raise CanvasError(f"{_SO} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)")
    lib = C.CDLL(_SO)
    lib.canvas_create.restype = C.c_void_p
    lib.canvas_create.argtypes = [C.c_int]
    lib.canvas_destroy.argtypes = [C.c_void_p]
    lib.canvas_last_error.restype = C.c_char_p
    lib.canvas_last_error.argtypes = [C.c_void_p]
    lib.canvas_version.restype = C.c_char_p
    lib.canvas_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.canvas_synchronize.argtypes = [C.c_void_p]
    lib.canvas_device_malloc.restype = C.c_void_p
    lib.canvas_device_malloc.argtypes = [C.c_void_p, C.c_int64]
    lib.canvas_device_free.argtypes = [C.c_void_p, C.c_void_p]
    lib.canvas_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    lib.canvas_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    lib.canvas_bin_count_upper_bound.restype = C.c_int64
    lib.canvas_bin_count_upper_bound.argtypes = [C.c_int32, C.c_void_p, C.c_int32]
    lib.canvas_bin_size_from_rates.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    lib.canvas_packed_plane_bytes.argtypes = [C.c_int64, C.c_void_p, C.c_void_p]
    lib.canvas_pack_reference_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32]
    lib.canvas_pack_hits_host.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32]
    lib.canvas_pack_hits2_host.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32]
    _lib = lib
    return lib


def _ptr_table(tensors):
    T = C.c_void_p * len(tensors)
    return T(*[t.data_ptr() for t in tensors])


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def packed_plane_words(length):
    """64-position words of the packed planes of one chromosome (whole tiles of 4096 positions): the reference planes hold 2 u64 per word, the hit planes 4"""
    return ((int(length) + 4095) // 4096) * 64


def _host_addr(a):
    return C.c_void_p(a.data_ptr() if hasattr(a, "data_ptr") else a.ctypes.data)


def pack_reference_host(bases, mask, length, out=None, threads=0):
    """canvas_pack_reference_host (plain host code, no GPU): (bases u8[len], mask u64[ceil(len/64)]) -> ({possible, gc} u64 pairs, pos0).
    Arrays are numpy arrays or (pinned) CPU torch tensors; `out` = u64[2 * packed_plane_words(len)]"""
    lib = load_library()
    if out is None:
        out = np.zeros(2 * packed_plane_words(length), np.uint64)
    p0 = C.c_int64(0)
    rc = lib.canvas_pack_reference_host(_host_addr(bases), _host_addr(mask), C.c_int64(int(length)), _host_addr(out), C.byref(p0), int(threads))
    if rc:
        raise CanvasError(f"canvas_pack_reference_host: error {rc}")
    return out, p0.value


def pack_hits_host(hits, length, out=None, threads=0):
    """canvas_pack_hits_host (plain host code, no GPU): hits u8[len] -> bit-sliced 4-bit planes u64[4 * packed_plane_words(len)]; returns (planes, #saturated positions)"""
    lib = load_library()
    if out is None:
        out = np.zeros(4 * packed_plane_words(length), np.uint64)
    sat = C.c_int64(0)
    rc = lib.canvas_pack_hits_host(_host_addr(hits), C.c_int64(int(length)), _host_addr(out), C.byref(sat), int(threads))
    if rc:
        raise CanvasError(f"canvas_pack_hits_host: error {rc}")
    return out, sat.value


def pack_hits2_host(hits, length, lo=None, hdr=None, extras=None, threads=0):
    """canvas_pack_hits2_host (plain host code): hits u8[len] -> the two-bit wire form (lo u64[2 W], hdr u64[2 W / 64], extras u64[2 n_extras]); returns
    (lo, hdr, extras, n_extras, #saturated).  The extras buffer is sized W / 16 entries unless given; it is grown once if the sample needs more."""
    lib = load_library()
    W = packed_plane_words(length)
    lo = np.zeros(2 * W, np.uint64) if lo is None else lo
    hdr = np.zeros(2 * (W // 64), np.uint64) if hdr is None else hdr
    extras = np.zeros(2 * (W // 16 + 64), np.uint64) if extras is None else extras
    for attempt in range(2):
        nx = C.c_int64(0); sat = C.c_int64(0)
        cap = int((extras.numel() if hasattr(extras, "numel") else len(extras)) // 2)
        rc = lib.canvas_pack_hits2_host(_host_addr(hits), C.c_int64(int(length)), _host_addr(lo), _host_addr(hdr), _host_addr(extras), C.c_int64(cap), C.byref(nx), C.byref(sat), int(threads))
        if rc == -4 and attempt == 0 and not hasattr(extras, "numel"):           # CANVAS_ERR_CAPACITY: more words with four hits and more than the default holds
            extras = np.zeros(2 * (nx.value + 64), np.uint64); continue
        if rc:
            raise CanvasError(f"canvas_pack_hits2_host: error {rc} (extras needed: {nx.value})")
        return lo, hdr, extras, nx.value, sat.value


class Canvas:
    """One context = one GPU + one stream.  Method names follow the reference functions they replace."""

    def __init__(self, device=0, stream=None):
        import torch
        self.torch = torch
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise CanvasError("no GPU visible: canvas_amd has no CPU fallback")
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(device)
        raw = self.lib.canvas_create(device)
        if not raw:
            raise CanvasError("canvas_create failed: no usable GPU (no CPU fallback)")
        self.ctx = C.c_void_p(raw)   # always pass as a 64-bit pointer
        self.comm_size = 1           # ranks of the library communicator (parallel.init_library_comm / init_host_comm set it)
        if stream is not None:
            self._check(self.lib.canvas_set_stream(self.ctx, C.c_void_p(stream)))

    def close(self):
        if getattr(self, "ctx", None) is not None:
            self.lib.canvas_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        # not at interpreter shutdown: the HIP runtime (and a profiler attached to it) may already be finalised, and the process exit frees everything anyway
        try:
            if not sys.is_finalizing():
                self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise CanvasError(f"libcanvas_hip error {rc}: {self.lib.canvas_last_error(self.ctx).decode()}")

    def use_torch_stream(self):
        """run on torch's current stream so torch.cuda.Event timing sees the kernels"""
        s = self.torch.cuda.current_stream(self.device).cuda_stream
        self._check(self.lib.canvas_set_stream(self.ctx, C.c_void_p(s)))

    def synchronize(self):
        self._check(self.lib.canvas_synchronize(self.ctx))

    def upload_genome_begin(self, lens, h_bases, d_bases, h_mask, d_mask, h_hits, d_hits):
        """queue the per-chromosome upload of host arrays (pinned torch tensors / numpy arrays; None = already resident) on the context's copy stream;
        the next binning call on the same destination tensors overlaps it (canvas_upload_genome_begin)"""
        n = len(d_bases)
        hl = np.ascontiguousarray(lens, np.int64)
        hp = lambda ts: None if ts is None else (C.c_void_p * n)(*[None if t is None else C.c_void_p(t.data_ptr() if hasattr(t, "data_ptr") else t.ctypes.data) for t in ts])
        self._check(self.lib.canvas_upload_genome_begin(self.ctx, n, _np_ptr(hl), hp(h_bases), _ptr_table(d_bases), hp(h_mask), _ptr_table(d_mask), hp(h_hits), _ptr_table(d_hits)))

    def upload_packed_begin(self, lens, h_ref, d_ref, h_planes, d_planes):
        """canvas_upload_packed_begin: the packed planes, chromosome by chromosome on the copy stream (h_ref None = reference planes already resident)"""
        n = len(d_ref)
        hl = np.ascontiguousarray(lens, np.int64)
        hp = lambda ts: None if ts is None else (C.c_void_p * n)(*[None if t is None else _host_addr(t) for t in ts])
        self._check(self.lib.canvas_upload_packed_begin(self.ctx, n, _np_ptr(hl), hp(h_ref), _ptr_table(d_ref), hp(h_planes), _ptr_table(d_planes)))

    def upload_packed2_begin(self, lens, h_ref, d_ref, h_lo, h_hdr, h_extras, n_extras, d_planes):
        """canvas_upload_packed2_begin: the hit planes in their two-bit wire form (pack_hits2_host), expanded into d_planes on the device behind each chromosome's transfer"""
        n = len(d_ref)
        hl = np.ascontiguousarray(lens, np.int64); nx = np.ascontiguousarray(n_extras, np.int64)
        hp = lambda ts: None if ts is None else (C.c_void_p * n)(*[None if t is None else _host_addr(t) for t in ts])
        self._check(self.lib.canvas_upload_packed2_begin(self.ctx, n, _np_ptr(hl), hp(h_ref), _ptr_table(d_ref), hp(h_lo), hp(h_hdr), hp(h_extras), _np_ptr(nx), _ptr_table(d_planes)))

    def pack_genome_device(self, bases, masks, hits, lens):
        """canvas_pack_genome_device: per-base arrays in HBM -> (ref planes, hit planes, pos0, #saturated); bases/masks or hits may be None"""
        torch = self.torch
        n = len(lens)
        hl = np.ascontiguousarray(lens, np.int64)
        ref = [torch.empty(2 * packed_plane_words(L), dtype=torch.int64, device=self.device) for L in lens] if bases is not None else None
        planes = [torch.empty(4 * packed_plane_words(L), dtype=torch.int64, device=self.device) for L in lens] if hits is not None else None
        pos0 = np.zeros(n, np.int64); sat = C.c_int64(0)
        tab = lambda ts: None if ts is None else _ptr_table(ts)
        self._check(self.lib.canvas_pack_genome_device(self.ctx, n, tab(bases), tab(masks), tab(hits), _np_ptr(hl), tab(ref), tab(planes),
                                                       _np_ptr(pos0) if bases is not None else None, C.byref(sat)))
        return ref, planes, (pos0 if bases is not None else None), sat.value

    def bin_sample_packed(self, ref, planes, lens, pos0, is_autosome, counts_per_bin=100, bin_size=-1, mode=MODE_TDR, out=None):
        """canvas_bin_sample over the packed planes (bit-identical outputs)"""
        n = len(ref)
        lens = np.ascontiguousarray(lens, np.int64); p0 = np.ascontiguousarray(pos0, np.int64)
        ia = np.ascontiguousarray(is_autosome, np.uint8)
        per = np.zeros(n, np.int64); total = C.c_int64(0); bs = C.c_int32(0)
        self._check(self.lib.canvas_bin_sample_packed(self.ctx, n, _ptr_table(ref), _ptr_table(planes), _np_ptr(lens), _np_ptr(p0), _np_ptr(ia), counts_per_bin, bin_size, mode,
                                                      C.c_void_p(out["chr"].data_ptr()), C.c_void_p(out["start"].data_ptr()), C.c_void_p(out["stop"].data_ptr()),
                                                      C.c_void_p(out["gc"].data_ptr()), C.c_void_p(out["count"].data_ptr()), C.c_int64(out["chr"].numel()),
                                                      C.byref(bs), _np_ptr(per), C.byref(total)))
        self.synchronize()
        return out, per, total.value, bs.value

    def upload_genome_wait(self):
        self._check(self.lib.canvas_upload_genome_wait(self.ctx))

    def memcpy_d2h(self, h_dst, d_src, nbytes):
        self._check(self.lib.canvas_memcpy_d2h(self.ctx, C.c_void_p(h_dst.data_ptr() if hasattr(h_dst, "data_ptr") else h_dst.ctypes.data), C.c_void_p(d_src.data_ptr()), C.c_int64(nbytes)))

    # ---- CanvasBin
    def mask_from_fasta(self, bases, length):
        """InitializeAlignmentArrays (CanvasBin.cs:183-200)"""
        mask = self.torch.empty((length + 63) // 64, dtype=self.torch.int64, device=self.device)
        self._check(self.lib.canvas_mask_from_fasta(self.ctx, C.c_void_p(bases.data_ptr()), C.c_int64(length), C.c_void_p(mask.data_ptr())))
        self.synchronize()
        return mask

    def mask_exclude_intervals(self, mask, length, starts, stops):
        """ExcludeTagsOverlappingFilterFile (CanvasBin.cs:668-692)"""
        a = np.ascontiguousarray(starts, np.int32); b = np.ascontiguousarray(stops, np.int32)
        self._check(self.lib.canvas_mask_exclude_intervals(self.ctx, C.c_void_p(mask.data_ptr()), C.c_int64(length), len(a), _np_ptr(a), _np_ptr(b)))

    def screen_hits(self, hits, mask, length):
        """ScreenObservedTags (CanvasBin.cs:699-716)"""
        self._check(self.lib.canvas_screen_hits(self.ctx, C.c_void_p(hits.data_ptr()), C.c_void_p(mask.data_ptr()), C.c_int64(length)))
        self.synchronize()

    def bin_rates(self, hits, masks, lens):
        """SampleHitArrays.GetRates (CanvasBin.cs:30-71)"""
        n = len(hits)
        lens = np.ascontiguousarray(lens, np.int64)
        obs = np.zeros(n, np.int64); poss = np.zeros(n, np.int64); rate = np.zeros(n, np.float64)
        self._check(self.lib.canvas_bin_rates(self.ctx, n, _ptr_table(hits), _ptr_table(masks), _np_ptr(lens), _np_ptr(obs), _np_ptr(poss), _np_ptr(rate)))
        return obs, poss, rate

    def bin_size_from_rates(self, rates, counts_per_bin):
        """SampleHitArrays.GetBinSize (CanvasBin.cs:73-83)"""
        r = np.ascontiguousarray(rates, np.float64)
        return int(self.lib.canvas_bin_size_from_rates(_np_ptr(r), len(r), counts_per_bin))

    def bin_genome(self, bases, masks, hits, lens, bin_size, mode=MODE_TDR, out=None):
        """BinCounts (CanvasBin.cs:416-550): returns dict(chr,start,stop,gc,count) device tensors trimmed to the bin count"""
        torch = self.torch
        n = len(bases)
        lens = np.ascontiguousarray(lens, np.int64)
        cap = int(self.lib.canvas_bin_count_upper_bound(n, _np_ptr(lens), bin_size))
        if cap < 0:
            raise CanvasError("bad bin size")
        if out is None:
            out = dict(chr=torch.empty(cap + 1, dtype=torch.int32, device=self.device), start=torch.empty(cap + 1, dtype=torch.int32, device=self.device),
                       stop=torch.empty(cap + 1, dtype=torch.int32, device=self.device), gc=torch.empty(cap + 1, dtype=torch.int32, device=self.device),
                       count=torch.empty(cap + 1, dtype=torch.float32, device=self.device))
        per = np.zeros(n, np.int64); total = C.c_int64(0)
        self._check(self.lib.canvas_bin_genome(self.ctx, n, _ptr_table(bases), _ptr_table(masks), _ptr_table(hits), _np_ptr(lens), bin_size, mode,
                                               C.c_void_p(out["chr"].data_ptr()), C.c_void_p(out["start"].data_ptr()), C.c_void_p(out["stop"].data_ptr()),
                                               C.c_void_p(out["gc"].data_ptr()), C.c_void_p(out["count"].data_ptr()), C.c_int64(out["chr"].numel()),
                                               _np_ptr(per), C.byref(total)))
        self.synchronize()   # own non-blocking stream: make the bins visible to torch's stream before handing tensors back
        return out, per, total.value

    def bin_sample(self, bases, masks, hits, lens, is_autosome, counts_per_bin=100, bin_size=-1, mode=MODE_TDR, out=None):
        """CanvasBin.RunSingleSample (CanvasBin.cs:914-931): rates -> bin size -> bins in one call"""
        n = len(bases)
        lens = np.ascontiguousarray(lens, np.int64)
        ia = np.ascontiguousarray(is_autosome, np.uint8)
        per = np.zeros(n, np.int64); total = C.c_int64(0); bs = C.c_int32(0)
        self._check(self.lib.canvas_bin_sample(self.ctx, n, _ptr_table(bases), _ptr_table(masks), _ptr_table(hits), _np_ptr(lens), _np_ptr(ia), counts_per_bin, bin_size, mode,
                                               C.c_void_p(out["chr"].data_ptr()), C.c_void_p(out["start"].data_ptr()), C.c_void_p(out["stop"].data_ptr()),
                                               C.c_void_p(out["gc"].data_ptr()), C.c_void_p(out["count"].data_ptr()), C.c_int64(out["chr"].numel()),
                                               C.byref(bs), _np_ptr(per), C.byref(total)))
        self.synchronize()
        return out, per, total.value, bs.value

    def bin_sample_gcweighted(self, bases, masks, hits, fraglens, lens, is_autosome, counts_per_bin=100, bin_size=-1, out=None):
        """CanvasBin -m GCContentWeighted (CanvasBin.cs:416-506,626-636)"""
        n = len(bases)
        lens = np.ascontiguousarray(lens, np.int64)
        ia = np.ascontiguousarray(is_autosome, np.uint8)
        per = np.zeros(n, np.int64); total = C.c_int64(0); bs = C.c_int32(0)
        self._check(self.lib.canvas_bin_sample_gcweighted(self.ctx, n, _ptr_table(bases), _ptr_table(masks), _ptr_table(hits), _ptr_table(fraglens), _np_ptr(lens), _np_ptr(ia),
                                                          counts_per_bin, bin_size, C.c_void_p(out["chr"].data_ptr()), C.c_void_p(out["start"].data_ptr()),
                                                          C.c_void_p(out["stop"].data_ptr()), C.c_void_p(out["gc"].data_ptr()), C.c_void_p(out["count"].data_ptr()),
                                                          C.c_int64(out["chr"].numel()), C.byref(bs), _np_ptr(per), C.byref(total)))
        self.synchronize()
        return out, per, total.value, bs.value

    def bin_sample_sharded(self, owner, bases, masks, hits, lens, is_autosome, out, counts_per_bin=100, bin_size=-1, mode=MODE_TDR, fraglens=None):
        """canvas_bin_sample_sharded: CanvasBin with the chromosomes sharded over the ranks (entries of chromosomes this rank does not own may be None); every rank gets the
        whole genome's bins in `out`.  fraglens: mode 5 (GCContentWeighted).  Returns (bin size, number of bins)."""
        n = len(bases)
        lens = np.ascontiguousarray(lens, np.int64); ia = np.ascontiguousarray(is_autosome, np.uint8); ow = np.ascontiguousarray(owner, np.int32)
        bs = C.c_int32(0); total = C.c_int64(0)
        tab = lambda ts: (C.c_void_p * n)(*[None if t is None else C.c_void_p(t.data_ptr()) for t in ts])
        self._check(self.lib.canvas_bin_sample_sharded(self.ctx, n, _np_ptr(ow), tab(bases), tab(masks), tab(hits), tab(fraglens) if fraglens is not None else None,
                                                       _np_ptr(lens), _np_ptr(ia), counts_per_bin, bin_size, mode, C.c_void_p(out["chr"].data_ptr()), C.c_void_p(out["start"].data_ptr()),
                                                       C.c_void_p(out["stop"].data_ptr()), C.c_void_p(out["gc"].data_ptr()), C.c_void_p(out["count"].data_ptr()),
                                                       C.c_int64(out["chr"].numel()), C.byref(bs), C.byref(total)))
        self.synchronize()
        return bs.value, total.value

    def bin_gcw_stats(self):
        """last bin_sample_gcweighted: (bins whose weighted count was decided from the exact sum + error interval, bins replayed in the reference's order)"""
        v = np.zeros(2, np.int64)
        self._check(self.lib.canvas_bin_gcw_stats(self.ctx, _np_ptr(v)))
        return int(v[0]), int(v[1])

    def bin_predefined(self, bases, masks, hits, lens, bin_starts, bin_stops, mode=MODE_TDR, fraglens=None):
        """CanvasBin -n (BinCountsForChromosome with predefined bins): bin_starts / bin_stops = one array per chromosome; returns (gc, count) tensors over the concatenated bins.
        mode 5 (GCContentWeighted) needs fraglens (int16 per base, per chromosome): canvas_bin_predefined_gcweighted"""
        torch = self.torch
        n = len(bases)
        lens = np.ascontiguousarray(lens, np.int64)
        off = np.concatenate([[0], np.cumsum([len(b) for b in bin_starts])]).astype(np.int64)
        hs = np.ascontiguousarray(np.concatenate([np.asarray(b, np.int32) for b in bin_starts] + [np.zeros(0, np.int32)]), np.int32)
        he = np.ascontiguousarray(np.concatenate([np.asarray(b, np.int32) for b in bin_stops] + [np.zeros(0, np.int32)]), np.int32)
        ds = torch.from_numpy(hs if len(hs) else np.zeros(1, np.int32)).to(self.device); de = torch.from_numpy(he if len(he) else np.zeros(1, np.int32)).to(self.device)
        gc = torch.empty(max(1, len(hs)), dtype=torch.int32, device=self.device); cnt = torch.empty(max(1, len(hs)), dtype=torch.float32, device=self.device)
        self.torch.cuda.synchronize()
        if fraglens is not None:
            self._check(self.lib.canvas_bin_predefined_gcweighted(self.ctx, n, _ptr_table(bases), _ptr_table(masks), _ptr_table(hits), _ptr_table(fraglens), _np_ptr(lens), _np_ptr(off), _np_ptr(hs),
                                                                  _np_ptr(he), C.c_void_p(ds.data_ptr()), C.c_void_p(de.data_ptr()), C.c_void_p(gc.data_ptr()), C.c_void_p(cnt.data_ptr())))
            return gc[:len(hs)], cnt[:len(hs)]
        self._check(self.lib.canvas_bin_predefined(self.ctx, n, _ptr_table(bases), _ptr_table(masks), _ptr_table(hits), _np_ptr(lens), int(mode), _np_ptr(off), _np_ptr(hs), _np_ptr(he),
                                                   C.c_void_p(ds.data_ptr()), C.c_void_p(de.data_ptr()), C.c_void_p(gc.data_ptr()), C.c_void_p(cnt.data_ptr())))
        return gc[:len(hs)], cnt[:len(hs)]

    def chromosome_offsets(self, chr_ids, n, nchr):
        off = np.zeros(nchr + 1, np.int64)
        self._check(self.lib.canvas_chromosome_offsets(self.ctx, C.c_void_p(chr_ids.data_ptr()), C.c_int64(n), nchr, _np_ptr(off)))
        return off

    # ---- CanvasClean
    def clean(self, bins, n, is_autosome, flags, min_bins_per_gc=100, is_y=None):
        """CanvasClean.Main (CanvasClean.cs:415-533) in place on device SoA; returns (n_out, local_sd, info)"""
        ia = np.ascontiguousarray(is_autosome, np.uint8)
        iy = np.ascontiguousarray(is_y if is_y is not None else np.zeros(len(ia)), np.uint8)
        lsd = C.c_double(-1.0); nout = C.c_int64(0); info = np.zeros(8, np.int32)
        self._check(self.lib.canvas_clean2(self.ctx, C.c_int64(n), C.c_void_p(bins["chr"].data_ptr()), C.c_void_p(bins["start"].data_ptr()),
                                           C.c_void_p(bins["stop"].data_ptr()), C.c_void_p(bins["count"].data_ptr()), C.c_void_p(bins["gc"].data_ptr()),
                                           len(ia), _np_ptr(ia), _np_ptr(iy), C.c_uint32(flags), min_bins_per_gc, C.byref(lsd), C.byref(nout), _np_ptr(info)))
        return nout.value, lsd.value, info

    def clean_batch(self, samples, ns, is_autosome, flags, min_bins_per_gc=100, is_y=None):
        """canvas_clean_batch: CanvasClean of several samples at once (list of SoA dicts, bins per sample); returns ([n_out], [local_sd], info[S][8])"""
        S = len(samples)
        ia = np.ascontiguousarray(is_autosome, np.uint8)
        iy = np.ascontiguousarray(is_y if is_y is not None else np.zeros(len(ia)), np.uint8)
        arr = lambda key: (C.c_void_p * S)(*[C.c_void_p(s[key].data_ptr()) for s in samples])
        hn = np.ascontiguousarray(ns, np.int64); lsd = np.full(S, -1.0, np.float64); nout = np.zeros(S, np.int64); info = np.zeros((S, 8), np.int32)
        self.torch.cuda.synchronize()
        self._check(self.lib.canvas_clean_batch(self.ctx, S, _np_ptr(hn), arr("chr"), arr("start"), arr("stop"), arr("count"), arr("gc"), len(ia), _np_ptr(ia), _np_ptr(iy),
                                                C.c_uint32(flags), int(min_bins_per_gc), _np_ptr(lsd), _np_ptr(nout), _np_ptr(info)))
        return nout, lsd, info

    def merge_cleaned(self, samples, ns):
        """MergeMultiSampleCleanedBedFile (Utilities.cs:834-920): bins every sample still has.  samples = list of SoA dicts (chr, start, stop,
        count), ns = bins per sample.  Returns (chr, start, stop, [count per sample], n_out)."""
        torch = self.torch
        S = len(samples); n0 = int(ns[0])
        oc = torch.empty(max(n0, 1), dtype=torch.int32, device=self.device); os_ = torch.empty_like(oc); oe = torch.empty_like(oc)
        ocnt = [torch.empty(max(n0, 1), dtype=torch.float32, device=self.device) for _ in range(S)]
        arr = lambda key: (C.c_void_p * S)(*[C.c_void_p(s[key].data_ptr()) for s in samples])
        hn = np.ascontiguousarray(ns, np.int64); nout = C.c_int64(0)
        self._check(self.lib.canvas_merge_cleaned(self.ctx, S, _np_ptr(hn), arr("chr"), arr("start"), arr("stop"), arr("count"), C.c_void_p(oc.data_ptr()),
                                                  C.c_void_p(os_.data_ptr()), C.c_void_p(oe.data_ptr()), (C.c_void_p * S)(*[C.c_void_p(t.data_ptr()) for t in ocnt]), C.byref(nout)))
        k = nout.value
        return oc[:k], os_[:k], oe[:k], [t[:k] for t in ocnt], k

    def quantize_f2(self, count, n, out=None):
        """count.ToString("F2") -> Convert.ToDouble (IO.cs:21 -> CanvasSegment.cs:1146), in memory"""
        cov = out[:n] if out is not None else self.torch.empty(n, dtype=self.torch.float64, device=self.device)
        self._check(self.lib.canvas_quantize_f2(self.ctx, C.c_void_p(count.data_ptr()), C.c_int64(n), C.c_void_p(cov.data_ptr())))
        self.synchronize()   # the library runs on its own non-blocking stream; make the result visible to torch's stream
        return cov

    def profile_enable(self, on=True):
        self._check(self.lib.canvas_profile_enable(self.ctx, int(on)))      # True / 1: every scope; 2: only the dominant kernel's scope

    def profile_get(self, name, reset=True):
        ms = C.c_double(0); k = C.c_int32(0)
        self._check(self.lib.canvas_profile_get(self.ctx, name.encode(), C.byref(ms), C.byref(k), int(reset)))
        return ms.value, k.value

    # ---- CanvasPartition
    def hmm_per_sample(self, cov, chr_offset, out=None):
        """HiddenMarkovModelsRunner.Run(isPerSample) (HiddenMarkovModelsRunner.cs:23-109): Viterbi state per bin"""
        torch = self.torch
        off = np.ascontiguousarray(chr_offset, np.int64)
        state = out[:int(off[-1])] if out is not None else torch.empty(int(off[-1]), dtype=torch.int32, device=self.device)
        self._check(self.lib.canvas_hmm_per_sample(self.ctx, len(off) - 1, C.c_void_p(cov.data_ptr()), _np_ptr(off), C.c_void_p(state.data_ptr())))
        return state

    def cbs_device_stats(self):
        """[device permutations, host permutations, exact re-evaluations, device batches, verified, violations] of the last cbs() call"""
        out = np.zeros(6, np.int64)
        self._check(self.lib.canvas_cbs_device_stats(self.ctx, _np_ptr(out)))
        return out

    def cbs_tail_probe(self, xs, tol=1e-6):
        """TailProbability.Nu of up to 100 arguments through the device series (canvas_cbs_tail_probe): (nu, flags)"""
        xs = np.ascontiguousarray(xs, np.float64); nu = np.zeros(len(xs), np.float64); fl = np.zeros(len(xs), np.int32)
        self._check(self.lib.canvas_cbs_tail_probe(self.ctx, _np_ptr(xs), len(xs), C.c_double(tol), _np_ptr(nu), _np_ptr(fl)))
        return nu, fl

    def cbs_cache_stats(self):
        """[draws read out of the stream cache, draws generated inside batches, draws the cache's generator produced, states fetched for host code] of the last cbs() call,
        then [bytes of device memory the cache holds, draws it holds]"""
        out = np.zeros(6, np.int64)
        self._check(self.lib.canvas_cbs_cache_stats(self.ctx, _np_ptr(out)))
        return out

    def cbs_stream_read(self, chromosome, position, nwords):
        """nwords draws of the chromosome-th generator's output stream from `position` on, out of the context's cache (canvas_cbs_stream_read)"""
        out = np.zeros(int(nwords), np.uint32)
        self._check(self.lib.canvas_cbs_stream_read(self.ctx, int(chromosome), C.c_int64(int(position)), C.c_int64(int(nwords)), _np_ptr(out)))
        return out

    def cbs_prefetch(self, nchr, words_per_chromosome=0):
        """start generating the draw streams of the first nchr chromosomes in the background (canvas_cbs_prefetch)"""
        self._check(self.lib.canvas_cbs_prefetch(self.ctx, int(nchr), C.c_int64(int(words_per_chromosome))))

    def cbs_perm_probe(self, x, seed, nb, kernel, tss):
        """one batch of nb permutations of the centred segment x through the device permutation engine (canvas_cbs_perm_probe): (lohi[nb, 2], ms[3])"""
        x = np.ascontiguousarray(x, np.float64)
        lohi = np.zeros((nb, 2), np.float64); ms = np.zeros(3, np.float64)
        self._check(self.lib.canvas_cbs_perm_probe(self.ctx, _np_ptr(x), C.c_int32(len(x)), C.c_uint32(seed & 0xFFFFFFFF), C.c_int32(nb), C.c_int32(kernel), C.c_double(tss), _np_ptr(lohi), _np_ptr(ms)))
        return lohi, ms

    def stale_reads(self):
        """process-wide [pinned results looked at, looks that came before the result had arrived (polled until it did)] (canvas_stale_reads)"""
        out = np.zeros(2, np.int64)
        self._check(self.lib.canvas_stale_reads(_np_ptr(out)))
        return out

    def cbs_tpermp_stats(self):
        """[edge tests (TPermP) run by the device kernel, swaps of all edge tests] of the last cbs() call"""
        out = np.zeros(2, np.int64)
        self._check(self.lib.canvas_cbs_tpermp_stats(self.ctx, _np_ptr(out)))
        return out

    def cbs_tailp_stats(self):
        """[TailP calls decided from the device series, calls recomputed by the host series] of the last cbs() call"""
        out = np.zeros(2, np.int64)
        self._check(self.lib.canvas_cbs_tailp_stats(self.ctx, _np_ptr(out)))
        return out

    def hmm_joint(self, covs, chr_offset, out=None):
        """-m HMM: joint Viterbi over several samples (HiddenMarkovModelsRunner.Run(isPerSample=false), Distributions.cs:257-323)"""
        torch = self.torch
        off = np.ascontiguousarray(chr_offset, np.int64)
        state = out[:int(off[-1])] if out is not None else torch.empty(int(off[-1]), dtype=torch.int32, device=self.device)
        ptrs = (C.c_void_p * len(covs))(*[C.c_void_p(c.data_ptr()) for c in covs])
        self._check(self.lib.canvas_hmm_joint(self.ctx, len(covs), len(off) - 1, ptrs, _np_ptr(off), C.c_void_p(state.data_ptr())))
        return state

    def evenness_score(self, cov, chr_offset, window=100000):
        """SegmentationInput.GetEvennessScore (Segmentation.cs:260-296): the value of --evenness-metric-file, or None when the reference writes no file"""
        off = np.ascontiguousarray(chr_offset, np.int64)
        score = C.c_double(0); valid = C.c_int32(0)
        self._check(self.lib.canvas_evenness_score(self.ctx, len(off) - 1, C.c_void_p(cov.data_ptr()), _np_ptr(off), int(window), C.byref(score), C.byref(valid)))
        return score.value if valid.value else None

    def segment_ids(self, chr_offset, state, start, stop, max_inter_bin_dist=1000000, excluded=None, out=None, ploidy=None):
        """DeriveSegments + PostProcessSegments; excluded = per-chromosome list of (starts, stops) of the -b BED file; ploidy = per-chromosome list
        of (one-based starts, ends, copy numbers) of the -p VCF"""
        torch = self.torch
        off = np.ascontiguousarray(chr_offset, np.int64)
        seg = out[:int(off[-1])] if out is not None else torch.empty(int(off[-1]), dtype=torch.int32, device=self.device)
        nseg = C.c_int64(0)
        if ploidy is not None:
            cat = lambda k, src: np.ascontiguousarray(np.concatenate([np.asarray(e[k], np.int32) for e in src] + [np.zeros(1, np.int32)]), np.int32)
            po = np.concatenate([[0], np.cumsum([len(e[0]) for e in ploidy])]).astype(np.int64)
            ps, pe, pc = cat(0, ploidy), cat(1, ploidy), cat(2, ploidy)
            if excluded is not None:
                eo = np.concatenate([[0], np.cumsum([len(e[0]) for e in excluded])]).astype(np.int64); es, ee = cat(0, excluded), cat(1, excluded)
            self._check(self.lib.canvas_segment_ids_ploidy(self.ctx, len(off) - 1, _np_ptr(off), C.c_void_p(state.data_ptr()), C.c_void_p(start.data_ptr()),
                                                           C.c_void_p(stop.data_ptr()), max_inter_bin_dist,
                                                           _np_ptr(eo) if excluded is not None else None, _np_ptr(es) if excluded is not None else None,
                                                           _np_ptr(ee) if excluded is not None else None, _np_ptr(po), _np_ptr(ps), _np_ptr(pe), _np_ptr(pc),
                                                           C.c_void_p(seg.data_ptr()), C.byref(nseg)))
        elif excluded is None:
            self._check(self.lib.canvas_segment_ids(self.ctx, len(off) - 1, _np_ptr(off), C.c_void_p(state.data_ptr()), C.c_void_p(start.data_ptr()),
                                                    C.c_void_p(stop.data_ptr()), max_inter_bin_dist, C.c_void_p(seg.data_ptr()), C.byref(nseg)))
        else:
            eo = np.concatenate([[0], np.cumsum([len(e[0]) for e in excluded])]).astype(np.int64)
            es = np.ascontiguousarray(np.concatenate([np.asarray(e[0], np.int32) for e in excluded]) if eo[-1] else np.zeros(1), np.int32)
            ee = np.ascontiguousarray(np.concatenate([np.asarray(e[1], np.int32) for e in excluded]) if eo[-1] else np.zeros(1), np.int32)
            self._check(self.lib.canvas_segment_ids_filtered(self.ctx, len(off) - 1, _np_ptr(off), C.c_void_p(state.data_ptr()), C.c_void_p(start.data_ptr()),
                                                             C.c_void_p(stop.data_ptr()), max_inter_bin_dist, _np_ptr(eo), _np_ptr(es), _np_ptr(ee),
                                                             C.c_void_p(seg.data_ptr()), C.byref(nseg)))
        return seg, nseg.value

    def cbs(self, cov, chr_offset, alpha=0.01, nperm=10000, undo=0, undo_sd=3.0):
        """CBSRunner.Run (CBSRunner.cs:40-151); undo: 0 None, 1 Prune, 2 SDUndo"""
        torch = self.torch
        off = np.ascontiguousarray(chr_offset, np.int64)
        seg_len = torch.zeros(int(off[-1]) + 1, dtype=torch.int32, device=self.device)
        nseg = np.zeros(len(off) - 1, np.int32); stats = np.zeros(8, np.int64)
        self._check(self.lib.canvas_cbs_undo(self.ctx, len(off) - 1, C.c_void_p(cov.data_ptr()), _np_ptr(off), C.c_double(alpha), C.c_uint32(nperm), undo, C.c_double(undo_sd),
                                             C.c_void_p(seg_len.data_ptr()), _np_ptr(nseg), _np_ptr(stats)))
        return seg_len, nseg, stats

    def wavelets(self, cov, chr_offset, is_germline=False, threshold_lower=0.05, threshold_upper=80.0, mad_factor=5.0, window=100000, min_size=10):
        """WaveletsRunner.Run up to the breakpoints (WaveletsRunner.cs:52-150): one array of segment-start bin indices per chromosome"""
        off = np.ascontiguousarray(chr_offset, np.int64)
        nchr = len(off) - 1
        out = np.zeros(int(off[-1] - off[0]) + nchr + 1, np.int32); oo = np.zeros(nchr + 1, np.int64)
        self._check(self.lib.canvas_wavelets(self.ctx, nchr, C.c_void_p(cov.data_ptr()), _np_ptr(off), int(bool(is_germline)), C.c_double(threshold_lower),
                                             C.c_double(threshold_upper), C.c_double(mad_factor), int(window), int(min_size), _np_ptr(out), C.c_int64(len(out)), _np_ptr(oo)))
        return [out[oo[c]:oo[c + 1]].copy() for c in range(nchr)]

    def sample_pipeline(self, bases, masks, hits, lens, is_autosome, out, cov, state, seg, counts_per_bin=100, bin_size=-1, mode=3, flags=0, min_bins_per_gc=100,
                        max_inter_bin_dist=1000000, is_y=None, prepared=None, pos0=None):
        """bin_sample -> clean -> quantize_f2 -> chromosome_offsets -> hmm_per_sample -> segment_ids in ONE library call (canvas_sample_pipeline).
        Returns dict(bin_size, total, n_out, lsd, off, nseg, prepared); results are left in `out`, `cov`, `state`, `seg`.  Passing the returned `prepared` back repeats the
        call on the SAME tensors and options without marshalling the arguments again (every other argument is then ignored).
        pos0 given: `bases` / `hits` are the packed reference / hit planes (canvas_sample_pipeline_packed; `masks` is ignored)."""
        nchr = len(bases) if prepared is None else 0
        # the whole marshalled call (pointer tables, scalars, out-parameters) is cached per set of tensors: a native host keeps these arrays anyway, and per pass
        # the Python side is then one foreign call (building ~30 ctypes arguments costs ~0.1 ms, 3 % of a pass)
        if prepared is None:
            arr = lambda ts: (C.c_void_p * nchr)(*[C.c_void_p(t.data_ptr()) for t in ts])
            pb, pm, ph = arr(bases), (arr(masks) if pos0 is None else None), arr(hits)
            hl = np.ascontiguousarray(lens, np.int64); ia = np.ascontiguousarray(is_autosome, np.uint8); iy = None if is_y is None else np.ascontiguousarray(is_y, np.uint8)
            bs = C.c_int32(0); total = C.c_int64(0); nclean = C.c_int64(0); lsd = C.c_double(-1.0); nseg = C.c_int64(0); off = np.zeros(nchr + 1, np.int64)
            p0 = None if pos0 is None else np.ascontiguousarray(pos0, np.int64)
            head = (self.ctx, nchr, pb, pm, ph, _np_ptr(hl)) if p0 is None else (self.ctx, nchr, pb, ph, _np_ptr(hl), _np_ptr(p0))
            args = head + (_np_ptr(ia), None if iy is None else _np_ptr(iy),
                    int(counts_per_bin), int(bin_size), int(mode), C.c_uint32(flags), int(min_bins_per_gc), int(max_inter_bin_dist),
                    C.c_void_p(out["chr"].data_ptr()), C.c_void_p(out["start"].data_ptr()), C.c_void_p(out["stop"].data_ptr()),
                    C.c_void_p(out["gc"].data_ptr()), C.c_void_p(out["count"].data_ptr()), C.c_int64(int(out["chr"].numel())),
                    C.c_void_p(cov.data_ptr()), C.c_void_p(state.data_ptr()), C.c_void_p(seg.data_ptr()),
                    C.byref(bs), C.byref(total), C.byref(nclean), C.byref(lsd), _np_ptr(off), C.byref(nseg))
            prepared = dict(args=args, keep=(pb, pm, ph, hl, ia, iy, p0), outs=(bs, total, nclean, lsd, nseg, off), packed=p0 is not None)     # valid for exactly these tensors and options
        bs, total, nclean, lsd, nseg, off = prepared["outs"]
        self._check((self.lib.canvas_sample_pipeline_packed if prepared["packed"] else self.lib.canvas_sample_pipeline)(*prepared["args"]))
        return dict(bin_size=bs.value, total=total.value, n_out=nclean.value, lsd=lsd.value, off=off.copy(), nseg=nseg.value, prepared=prepared)

    def tumor_normal_flow(self, bases, masks, hits_t, fraglen_t, hits_n, lens, is_autosome, clean_flags, alpha=0.01, nperm=10000, counts_per_bin=100, is_y=None, keep=False, owner=None):
        """BASELINE configs[4] in memory, the hand-offs of the reference's tumour / normal flow: CanvasBin -m GCContentWeighted on the tumour (bin size from
        the tumour's own rates) and -m TruncatedDynamicRange -z <that size> on the normal (same mask => same bins), CanvasNormalize's LSNorm ratio x 40
        (LSNormRatioCalculator.cs:31-44, CanvasNormalizeUtilities.cs:23-33), its "{count:F2}" file read back with float.Parse (IO.cs:21,40), CanvasClean,
        the F2 hand-off to CanvasPartition and CBS (alpha, nperm).  Returns a dict; with keep=True every intermediate array is kept for checking.
        owner given (one entry per chromosome): the chromosome-sharded flow over the communicator of the context — this rank holds the arrays of its own chromosomes only (None
        elsewhere); both binnings are canvas_bin_sample_sharded, the ratio and CanvasClean run on the whole genome on every rank, CBS is canvas_cbs_sharded: same result on every rank."""
        import time
        torch = self.torch
        nchr = len(bases)
        cap = int(sum(int(l) for l in lens) // 50) + 64
        mk = lambda dt: torch.empty(cap, dtype=dt, device=self.device)
        stage = {}; t_prev = [time.perf_counter()]

        def tick(name):
            self.synchronize(); torch.cuda.synchronize()
            now = time.perf_counter(); stage[name] = round(now - t_prev[0], 4); t_prev[0] = now
        T = dict(chr=mk(torch.int32), start=mk(torch.int32), stop=mk(torch.int32), gc=mk(torch.int32), count=mk(torch.float32))
        N = dict(chr=mk(torch.int32), start=mk(torch.int32), stop=mk(torch.int32), gc=mk(torch.int32), count=mk(torch.float32))
        if owner is None:
            _, perT, nT, bs = self.bin_sample_gcweighted(bases, masks, hits_t, fraglen_t, lens, is_autosome, counts_per_bin, -1, out=T)
            tick("bin_tumour_gcweighted")
            _, perN, nN, _ = self.bin_sample(bases, masks, hits_n, lens, is_autosome, counts_per_bin, bs, MODE_TDR, out=N)
        else:
            bs, nT = self.bin_sample_sharded(owner, bases, masks, hits_t, lens, is_autosome, T, counts_per_bin, -1, MODE_GCW, fraglens=fraglen_t)
            tick("bin_tumour_gcweighted")
            _, nN = self.bin_sample_sharded(owner, bases, masks, hits_n, lens, is_autosome, N, counts_per_bin, bs, MODE_TDR)
        tick("bin_normal")
        if nN != nT:
            raise CanvasError(f"tumour and normal bins differ ({nT} vs {nN}): they must share the reference mask")
        kidx, ratio, count, lsf = self.normalize_ratio(T["count"][:nT], N["count"][:nN], None, mode=0)
        k = int(kidx.numel())
        ki = kidx.long()
        R = dict(chr=T["chr"][:nT][ki].contiguous(), start=T["start"][:nT][ki].contiguous(), stop=T["stop"][:nT][ki].contiguous(), gc=T["gc"][:nT][ki].contiguous(),
                 count=self.quantize_f2(count, k).float().contiguous())       # the ratio file's F2 text, float.Parse'd by CanvasClean
        tick("normalize_ratio+f2")
        res = dict(bin_size=bs, n_bins=nT, n_ratio=k, library_size_factor=lsf)
        if keep:
            res.update(tumour={a: T[a][:nT].clone() for a in T}, normal_count=N["count"][:nN].clone(), keep_idx=kidx.clone(), ratio=ratio.clone(), ratio_count=count.clone(),
                       to_clean={a: R[a].clone() for a in R})
        n_out, lsd, info = self.clean(R, k, is_autosome, clean_flags, is_y=is_y)
        tick("clean")
        cov = self.quantize_f2(R["count"], n_out)
        off = self.chromosome_offsets(R["chr"], n_out, nchr)
        tick("f2+offsets")
        seg_len, nseg, stats = self.cbs(cov, off, alpha, nperm) if owner is None else self.cbs_sharded(owner, cov, off, alpha, nperm)
        tick("cbs")
        res.update(n_clean=n_out, local_sd=lsd, chr_offset=off, nseg=nseg, cbs_stats=stats, segments=int(nseg.sum()), stage_seconds=stage)
        if keep:
            res.update(cleaned={a: R[a][:n_out].clone() for a in R}, cov=cov.clone(), seg_len=seg_len)
        return res

    def sample_pipeline_sharded(self, owner, bases, masks, hits, lens, is_autosome, out, cov, state, seg, counts_per_bin=100, bin_size=-1, mode=3, flags=0, min_bins_per_gc=100,
                                max_inter_bin_dist=1000000, is_y=None, pos0=None):
        """canvas_sample_pipeline_sharded: `owner[c]` = rank that holds chromosome c; bases / masks / hits are lists over ALL chromosomes with None for the ones this rank
        does not own.  Every rank gets the whole result (same dict as sample_pipeline)."""
        nchr = len(owner)
        tab = lambda ts: (C.c_void_p * nchr)(*[None if t is None else C.c_void_p(t.data_ptr()) for t in ts])
        ow = np.ascontiguousarray(owner, np.int32); hl = np.ascontiguousarray(lens, np.int64); ia = np.ascontiguousarray(is_autosome, np.uint8)
        iy = None if is_y is None else np.ascontiguousarray(is_y, np.uint8)
        bs = C.c_int32(0); total = C.c_int64(0); nclean = C.c_int64(0); lsd = C.c_double(-1.0); nseg = C.c_int64(0); off = np.zeros(nchr + 1, np.int64)
        if pos0 is not None:       # packed planes: `bases` = reference planes, `hits` = hit planes (None for chromosomes of other ranks), `masks` ignored
            p0 = np.ascontiguousarray(pos0, np.int64)
            self._check(self.lib.canvas_sample_pipeline_sharded_packed(self.ctx, nchr, _np_ptr(ow), tab(bases), tab(hits), _np_ptr(hl), _np_ptr(p0), _np_ptr(ia), None if iy is None else _np_ptr(iy),
                                                                       int(counts_per_bin), int(bin_size), int(mode), C.c_uint32(flags), int(min_bins_per_gc), int(max_inter_bin_dist),
                                                                       C.c_void_p(out["chr"].data_ptr()), C.c_void_p(out["start"].data_ptr()), C.c_void_p(out["stop"].data_ptr()),
                                                                       C.c_void_p(out["gc"].data_ptr()), C.c_void_p(out["count"].data_ptr()), C.c_int64(int(out["chr"].numel())),
                                                                       C.c_void_p(cov.data_ptr()), C.c_void_p(state.data_ptr()), C.c_void_p(seg.data_ptr()),
                                                                       C.byref(bs), C.byref(total), C.byref(nclean), C.byref(lsd), _np_ptr(off), C.byref(nseg)))
            return dict(bin_size=bs.value, total=total.value, n_out=nclean.value, lsd=lsd.value, off=off, nseg=nseg.value)
        self._check(self.lib.canvas_sample_pipeline_sharded(self.ctx, nchr, _np_ptr(ow), tab(bases), tab(masks), tab(hits), _np_ptr(hl), _np_ptr(ia), None if iy is None else _np_ptr(iy),
                                                            int(counts_per_bin), int(bin_size), int(mode), C.c_uint32(flags), int(min_bins_per_gc), int(max_inter_bin_dist),
                                                            C.c_void_p(out["chr"].data_ptr()), C.c_void_p(out["start"].data_ptr()), C.c_void_p(out["stop"].data_ptr()),
                                                            C.c_void_p(out["gc"].data_ptr()), C.c_void_p(out["count"].data_ptr()), C.c_int64(int(out["chr"].numel())),
                                                            C.c_void_p(cov.data_ptr()), C.c_void_p(state.data_ptr()), C.c_void_p(seg.data_ptr()),
                                                            C.byref(bs), C.byref(total), C.byref(nclean), C.byref(lsd), _np_ptr(off), C.byref(nseg)))
        return dict(bin_size=bs.value, total=total.value, n_out=nclean.value, lsd=lsd.value, off=off, nseg=nseg.value)

    def hmm_per_sample_sharded(self, owner, cov, chr_offset):
        """canvas_hmm_per_sample_sharded: every rank holds the whole coverage, decodes the chromosomes it owns and receives everybody's state runs (same return as hmm_per_sample)"""
        torch = self.torch
        off = np.ascontiguousarray(chr_offset, np.int64); ow = np.ascontiguousarray(owner, np.int32)
        state = torch.empty(max(int(off[-1]), 1), dtype=torch.int32, device=self.device)
        self._check(self.lib.canvas_hmm_per_sample_sharded(self.ctx, len(off) - 1, _np_ptr(ow), C.c_void_p(cov.data_ptr()), _np_ptr(off), C.c_void_p(state.data_ptr())))
        return state[:int(off[-1])]

    def cbs_sharded(self, owner, cov, chr_offset, alpha=0.01, nperm=10000, undo=0, undo_sd=3.0):
        """canvas_cbs_sharded: every rank holds the whole coverage, segments the chromosomes it owns and receives everybody's segments (same return as cbs)"""
        torch = self.torch
        off = np.ascontiguousarray(chr_offset, np.int64); ow = np.ascontiguousarray(owner, np.int32)
        seg_len = torch.zeros(int(off[-1]) + 1, dtype=torch.int32, device=self.device)
        nseg = np.zeros(len(off) - 1, np.int32); stats = np.zeros(8, np.int64)
        self._check(self.lib.canvas_cbs_sharded(self.ctx, len(off) - 1, _np_ptr(ow), C.c_void_p(cov.data_ptr()), _np_ptr(off), C.c_double(alpha), C.c_uint32(nperm), int(undo), C.c_double(undo_sd),
                                                C.c_void_p(seg_len.data_ptr()), _np_ptr(nseg), _np_ptr(stats)))
        return seg_len, nseg, stats

    def wavelets_sharded(self, owner, cov, chr_offset, is_germline=False, threshold_lower=0.05, threshold_upper=80.0, mad_factor=5.0, window=100000, min_size=10):
        """canvas_wavelets_sharded: same return as wavelets, on every rank"""
        off = np.ascontiguousarray(chr_offset, np.int64); ow = np.ascontiguousarray(owner, np.int32)
        nchr = len(off) - 1
        out = np.zeros(int(off[-1] - off[0]) + nchr + 1, np.int32); oo = np.zeros(nchr + 1, np.int64)
        self._check(self.lib.canvas_wavelets_sharded(self.ctx, nchr, _np_ptr(ow), C.c_void_p(cov.data_ptr()), _np_ptr(off), int(bool(is_germline)), C.c_double(threshold_lower),
                                                     C.c_double(threshold_upper), C.c_double(mad_factor), int(window), int(min_size), _np_ptr(out), C.c_int64(len(out)), _np_ptr(oo)))
        return [out[oo[c]:oo[c + 1]].copy() for c in range(nchr)]

    def allgather_host(self, arr):
        """canvas_allgather_host: a small numpy array from every rank -> array of shape (ranks,) + arr.shape"""
        a = np.ascontiguousarray(arr)
        out = np.zeros((self.comm_size,) + a.shape, a.dtype)
        self._check(self.lib.canvas_allgather_host(self.ctx, _np_ptr(a), C.c_int64(a.nbytes), _np_ptr(out)))
        return out

    def merge_cleaned_sharded(self, bins, n):
        """canvas_merge_cleaned_sharded: this rank's cleaned SoA (dict chr/start/stop/count, n bins) -> (chr, start, stop, this sample's counts, n_out); one sample per rank"""
        torch = self.torch
        cap = int(bins["chr"].numel())
        mk = lambda dt: torch.empty(cap, dtype=dt, device=self.device)
        oc, os_, oe, ov = mk(torch.int32), mk(torch.int32), mk(torch.int32), mk(torch.float32)
        k = C.c_int64(0)
        self._check(self.lib.canvas_merge_cleaned_sharded(self.ctx, C.c_int64(int(n)), C.c_void_p(bins["chr"].data_ptr()), C.c_void_p(bins["start"].data_ptr()), C.c_void_p(bins["stop"].data_ptr()),
                                                          C.c_void_p(bins["count"].data_ptr()), C.c_void_p(oc.data_ptr()), C.c_void_p(os_.data_ptr()), C.c_void_p(oe.data_ptr()), C.c_void_p(ov.data_ptr()),
                                                          C.c_int64(cap), C.byref(k)))
        return oc[:k.value], os_[:k.value], oe[:k.value], ov[:k.value], k.value

    def sharded_stats(self):
        out = np.zeros(6, np.int64)
        self._check(self.lib.canvas_sharded_stats(self.ctx, _np_ptr(out)))
        return out

    # ---- CanvasNormalize (ratio path)
    def normalize_reference(self, counts, on_target_idx=None):
        """WeightedAverageReferenceGenerator.Run for several control samples: (weighted counts f64 tensor, weights)"""
        torch = self.torch
        n = int(counts[0].numel())
        ptrs = (C.c_void_p * len(counts))(*[C.c_void_p(c.data_ptr()) for c in counts])
        out = torch.empty(n, dtype=torch.float64, device=self.device); w = np.zeros(len(counts), np.float64)
        self._check(self.lib.canvas_normalize_reference(self.ctx, len(counts), ptrs, C.c_int64(n), C.c_void_p(on_target_idx.data_ptr()) if on_target_idx is not None else None,
                                                        C.c_int64(int(on_target_idx.numel()) if on_target_idx is not None else 0), C.c_void_p(out.data_ptr()), _np_ptr(w)))
        self.synchronize()   # the weighted sum is still in flight on the library's stream
        return out, w

    def normalize_ratio(self, sample, reference, on_target_idx=None, mode=0, min_ref=1.0, max_ref=float("inf"), ploidy=None):
        """LSNorm (mode 0) / Raw (mode 1) ratio + RatiosToCounts: (kept bin indices, ratios, counts, library-size factor)"""
        torch = self.torch
        n = int(sample.numel())
        keep = torch.empty(n, dtype=torch.int32, device=self.device); ratio = torch.empty(n, dtype=torch.float32, device=self.device); count = torch.empty(n, dtype=torch.float32, device=self.device)
        n_out = C.c_int64(0); lsf = C.c_double(0)
        self._check(self.lib.canvas_normalize_ratio(self.ctx, C.c_int64(n), C.c_void_p(sample.data_ptr()), C.c_void_p(reference.data_ptr()),
                                                    C.c_void_p(on_target_idx.data_ptr()) if on_target_idx is not None else None, C.c_int64(int(on_target_idx.numel()) if on_target_idx is not None else 0),
                                                    int(mode), C.c_double(min_ref), C.c_double(max_ref), C.c_void_p(ploidy.data_ptr()) if ploidy is not None else None,
                                                    C.c_void_p(keep.data_ptr()), C.c_void_p(ratio.data_ptr()), C.c_void_p(count.data_ptr()), C.byref(n_out), C.byref(lsf)))
        k = n_out.value
        return keep[:k], ratio[:k], count[:k], lsf.value

    def wavelets_decisions(self):
        """[long nodes decided from the closed form, undecided -> exact chain, chained for their coefficient, closed form in use] of the last wavelets() call"""
        out = np.zeros(4, np.int64)
        self._check(self.lib.canvas_wavelets_decisions(self.ctx, _np_ptr(out)))
        return [int(v) for v in out]

    def wavelets_stats(self):
        """[tree levels processed, nodes recomputed by the exact chain] of the last wavelets() call"""
        out = np.zeros(2, np.int64)
        self._check(self.lib.canvas_wavelets_stats(self.ctx, _np_ptr(out)))
        return out


# ---- synthetic generator (bench/test tooling, separate library)
_synth = None


def synth_generate_sample_device(seed, hit_seed, chrom, length, thr_dev, device, with_fraglen=False, bases=None, mask=None):
    """a further sample over the reference of `seed` (tumour / normal pairs): (hits, fraglen or None); mirrors synth.generate_chromosome(hit_seed=...).
    bases / mask tensors are (re)written when given."""
    import torch
    from . import synth
    global _synth
    if _synth is None:
        load_library()
        if not os.path.exists(_SYNTH_SO):
            raise CanvasError(f"{_SYNTH_SO} missing: run build()")
        _synth = C.CDLL(_SYNTH_SO)
    gap0, g1s, g1e, base_cn = synth.chrom_params(chrom, length)
    pad = (length + 63) // 64 * 64
    hits = torch.zeros(pad, dtype=torch.uint8, device=device)
    fl = torch.zeros(pad, dtype=torch.int16, device=device) if with_fraglen else None
    rc = _synth.synth_generate_sample(C.c_uint32(seed), C.c_uint32(hit_seed), C.c_uint32(chrom), C.c_int64(length), C.c_int64(gap0), C.c_int64(g1s), C.c_int64(g1e), C.c_uint32(base_cn),
                                      C.c_void_p(thr_dev.data_ptr()), C.c_void_p(bases.data_ptr()) if bases is not None else None, C.c_void_p(hits.data_ptr()),
                                      C.c_void_p(mask.data_ptr()) if mask is not None else None, C.c_void_p(fl.data_ptr()) if fl is not None else None,
                                      C.c_void_p(torch.cuda.current_stream(device).cuda_stream))
    if rc != 0:
        raise CanvasError(f"synth_generate_sample failed: hip error {rc}")
    return hits, fl


def synth_generate_device(seed, chrom, length, rate, device, thr_dev=None):
    """generate (bases, hits, mask) for one chromosome directly in HBM; mirrors canvas_amd.synth.generate_chromosome"""
    import torch
    from . import synth
    global _synth
    if _synth is None:
        load_library()
        if not os.path.exists(_SYNTH_SO):
            raise CanvasError(f"{_SYNTH_SO} missing: run build()")
        _synth = C.CDLL(_SYNTH_SO)
    if thr_dev is None:
        thr_dev = torch.from_numpy(synth.poisson_thresholds(rate).view(np.int32)).to(device)
    gap0, g1s, g1e, base_cn = synth.chrom_params(chrom, length)
    pad = (length + 63) // 64 * 64
    bases = torch.empty(pad, dtype=torch.uint8, device=device)
    hits = torch.empty(pad, dtype=torch.uint8, device=device)
    mask = torch.empty(pad // 64, dtype=torch.int64, device=device)
    rc = _synth.synth_generate(C.c_uint32(seed), C.c_uint32(chrom), C.c_int64(length), C.c_int64(gap0), C.c_int64(g1s), C.c_int64(g1e), C.c_uint32(base_cn),
                               C.c_void_p(thr_dev.data_ptr()), C.c_void_p(bases.data_ptr()), C.c_void_p(hits.data_ptr()), C.c_void_p(mask.data_ptr()),
                               C.c_void_p(torch.cuda.current_stream(device).cuda_stream))
    if rc != 0:
        raise CanvasError(f"synth_generate failed: hip error {rc}")
    return bases, hits, mask, thr_dev
