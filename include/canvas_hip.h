/* canvas_hip.h — C ABI of libcanvas_hip.so: the MI355X (gfx950) implementation of Canvas's read-depth hot path
 * (CanvasBin merge step -> CanvasClean -> CanvasPartition).
 *
 * The reference (Illumina/canvas, C#) has no in-process FFI for this path: its boundary is three executables that
 * exchange gzip text files.  This header is the boundary a thin C# `Main` P/Invokes instead of running
 * the C# loops; each entry point names the reference code it replaces (paths relative to Src/Canvas/).
 * INTEGRATION.md shows the DllImport stubs.
 *
 * Conventions
 *   - plain C, no exceptions cross the boundary; every function returns int32 status: 0 = ok, <0 = CANVAS_ERR_*;
 *     the message is available from canvas_last_error(ctx).
 *   - "d_" pointers are DEVICE pointers (from canvas_device_malloc or any HIP allocation of the same process);
 *     "h_" pointers are host pointers.  The library never frees or retains caller memory past the call.
 *   - one context = one GPU + one HIP stream; calls on a context are serialized by the caller.
 *   - alignment: d_bases / d_hits 16 bytes, d_mask 8 bytes and padded to a multiple of 8 bytes.
 *   - possible-alignment mask layout = System.Collections.BitArray: bit i of the chromosome is bit (i & 7) of byte i >> 3.
 */
#ifndef CANVAS_HIP_H
#define CANVAS_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct canvas_ctx canvas_ctx;

enum {
    CANVAS_OK = 0,
    CANVAS_ERR_INVALID = -1,     /* bad argument */
    CANVAS_ERR_HIP = -2,         /* HIP runtime error (no device, launch failure, OOM) */
    CANVAS_ERR_UNSUPPORTED = -3, /* reference feature outside the built scope */
    CANVAS_ERR_CAPACITY = -4,    /* caller buffer too small */
    CANVAS_ERR_COMM = -5         /* RCCL error */
};

/* coverage modes: CanvasCommon/Utilities.cs:56-74 (ParseCanvasCoverageMode) */
enum { CANVAS_MODE_BINARY = 0, CANVAS_MODE_TRUNCATED_DYNAMIC_RANGE = 3, CANVAS_MODE_GC_CONTENT_WEIGHTED = 5 };

/* CanvasClean switches: CanvasClean/CanvasClean.cs:431-446 (-g, -s, -r, --local-sd-metric-file, -m LOESS) */
enum { CANVAS_CLEAN_GCNORM = 1, CANVAS_CLEAN_FILTSIZE = 2, CANVAS_CLEAN_OUTLIERS = 4, CANVAS_CLEAN_LOCALSD = 8, CANVAS_CLEAN_LOESS = 16 };

/* ---- context ------------------------------------------------------------------------------------------------- */
canvas_ctx* canvas_create(int device);                 /* NULL when no usable GPU (the product has no CPU fallback) */
void canvas_destroy(canvas_ctx* ctx);
const char* canvas_last_error(canvas_ctx* ctx);
const char* canvas_version(void);
/* Process-wide counters of the results that kernels write straight into pinned host memory (bin size and totals of CanvasBin, quartiles and segment count of
 * CanvasPartition, Wavelets reports ...).  Each carries a sequence word that the kernel stores last (system-scope release) and the host checks after its synchronisation:
 * h_out2[0] = results looked at, h_out2[1] = looks that came before the result had arrived, after which the library polled the word until it did (0 on a quiet system;
 * a synchronisation that returns early was observed about once in eight process starts in round 4).  No reference counterpart. */
int32_t canvas_stale_reads(int64_t* h_out2);
int32_t canvas_set_stream(canvas_ctx* ctx, void* hip_stream); /* run on a caller-owned hipStream_t (NULL = own stream) */
/* A hint, never a change of results: the host makes ONE call per method with this context and then exits (the reference launches CanvasBin / CanvasClean / CanvasPartition as
 * one OS process per sample, Canvas/CanvasRunner.cs:123-128).  The library then uploads from the caller's pageable arrays where it would otherwise stage through pinned host
 * memory of its own: pinning costs ~1.4 ms per MB and the same again when the process leaves, which a second call would amortise and a one-shot process cannot. */
int32_t canvas_set_one_shot(canvas_ctx* ctx, int32_t on);
int32_t canvas_synchronize(canvas_ctx* ctx);
void* canvas_device_malloc(canvas_ctx* ctx, int64_t bytes);
int32_t canvas_device_free(canvas_ctx* ctx, void* d_ptr);
int32_t canvas_memcpy_h2d(canvas_ctx* ctx, void* d_dst, const void* h_src, int64_t bytes);
int32_t canvas_memcpy_d2h(canvas_ctx* ctx, void* h_dst, const void* d_src, int64_t bytes);

/* Pins a host array (hipHostRegister) so that uploads from it run at the full PCIe rate and asynchronously: what a C# host does once with the arrays
 * LoadIntermediateData left in memory (GCHandle.Alloc(..., Pinned) + this call). */
int32_t canvas_host_register(canvas_ctx* ctx, void* h_ptr, int64_t bytes);
int32_t canvas_host_unregister(canvas_ctx* ctx, void* h_ptr);
/* The per-base arrays of CanvasBin always start in host memory (CanvasBin.LoadIntermediateData, CanvasBin/CanvasBin.cs:965-969).  This call queues their upload
 * chromosome by chromosome on a copy stream of the context (h_* = host sources, d_* = caller-owned device destinations of at least h_len[c] bytes, the mask
 * ceil(len/64) words; a NULL source table or entry = that array is already resident, e.g. the reference bases and the mask of a cohort) and returns at once.
 * The next canvas_bin_sample / canvas_bin_genome / canvas_sample_pipeline call on this context that is given exactly these destination tables sweeps every
 * chromosome as soon as it has arrived, so the upload of chromosome c + 1 overlaps the sweep of chromosome c: pass time ~ max(PCIe, compute), not the sum.
 * Any other call must be preceded by canvas_upload_genome_wait.  Sources must stay valid (and should be pinned) until the binning call has returned. */
int32_t canvas_upload_genome_begin(canvas_ctx* ctx, int32_t nchr, const int64_t* h_len, const uint8_t* const* h_bases, uint8_t* const* d_bases,
                                   const uint64_t* const* h_mask, uint64_t* const* d_mask, const uint8_t* const* h_hits, uint8_t* const* d_hits);
int32_t canvas_upload_genome_wait(canvas_ctx* ctx);

/* ---- packed per-base inputs -------------------------------------------------------------------------------------
 * The loops this library replaces read, per position, one BitArray bit (CanvasBin.cs:593), whether the base is G/C (CanvasBin.cs:599-606), whether it is 'n'
 * (only to find the first one that is not, CanvasBin.cs:582-584) and min(10, hits) (TruncatedDynamicRange, CanvasBin.cs:618-619) or a 0/1 hit (Binary).
 * The packed planes carry exactly that, 0.75 B/base instead of 2.125 B/base (PCIe, which bounds a whole pass, and the one full HBM sweep both shrink 2.8x):
 *   reference planes  per 64 positions {u64 possible, u64 gc}                        16 B   (depends on the reference genome only)
 *   hit planes        per 64 positions {u64 b0, b1, b2, b3}, bit i of b_k = bit k of min(15, hits[i])   32 B
 *   pos0              first position whose base is not 'n' (len if there is none)
 * Both planes cover whole tiles of 4096 positions (canvas_packed_plane_bytes) and are zero beyond len.  Results are bit-identical to the byte-array entry
 * points in Binary and TruncatedDynamicRange modes (hits above 15 cannot occur in Binary mode and read as 10 in TruncatedDynamicRange either way; the packers
 * report the number of saturated positions).  GCContentWeighted mode reads fragment lengths per base and stays on the byte arrays. */
int32_t canvas_packed_plane_bytes(int64_t len, int64_t* ref_bytes, int64_t* hit_bytes);
/* Host-side packers (plain CPU code, no device, thread-parallel; `threads` <= 0 = one per hardware thread, at most 32): what a host that holds the reference's byte
 * arrays (CanvasBin.cs:965-969) runs once per reference / once per sample before the upload.  A host that fills the planes while parsing needs neither. */
int32_t canvas_pack_reference_host(const uint8_t* bases, const uint64_t* mask, int64_t len, uint64_t* ref_out, int64_t* pos0_out, int32_t threads);
int32_t canvas_pack_hits_host(const uint8_t* hits, int64_t len, uint64_t* planes_out, int64_t* saturated_out, int32_t threads);
/* The same packing for arrays that already are in HBM.  d_bases + d_mask + d_ref_out + h_pos0_out may all be NULL (hits only: a second sample over a packed
 * reference), or d_hits + d_hit_planes_out (reference only). */
int32_t canvas_pack_genome_device(canvas_ctx* ctx, int32_t nchr, const uint8_t* const* d_bases, const uint64_t* const* d_mask, const uint8_t* const* d_hits, const int64_t* h_len,
                                  uint64_t* const* d_ref_out, uint64_t* const* d_hit_planes_out, int64_t* h_pos0_out, int64_t* h_saturated_out);
/* canvas_upload_genome_begin for the planes (h_ref NULL or h_ref[c] NULL: already resident); the next canvas_bin_sample_packed / canvas_sample_pipeline_packed call
 * over these destination tables sweeps chromosome c while chromosome c + 1 is in flight.  canvas_upload_genome_wait applies. */
int32_t canvas_upload_packed_begin(canvas_ctx* ctx, int32_t nchr, const int64_t* h_len, const uint64_t* const* h_ref, uint64_t* const* d_ref,
                                   const uint64_t* const* h_hit_planes, uint64_t* const* d_hit_planes);
/* Two-bit wire form of the hit planes.  At WGS depth four hits or more at one position are rare (6.6e-5 of the positions at 60x), so the planes b2 and b3 are almost
 * entirely zero: over PCIe the hit planes can travel as  lo: per 64 positions {u64 b0, b1} (16 B);  hdr: per tile of 64 words {u64 xmask, u64 xoff} (bit w of xmask: word w
 * of the tile has a non-zero b2 or b3, xoff: index of the tile's first entry in extras);  extras: {u64 b2, b3} of those words, in word order  — 0.25 B/base plus a few MB
 * instead of 0.5 B/base.  canvas_pack_hits2_host builds the three pieces from the byte array (lo_out: 2 u64 per word, hdr_out: 2 u64 per tile, extras_out: room for
 * extras_cap_words entries; CANVAS_ERR_CAPACITY with *n_extras_out = the number needed if that is too small).  canvas_upload_packed2_begin is canvas_upload_packed_begin
 * for this form: the pieces are expanded into d_hit_planes[c] (the four planes) on the device right behind their transfer, so canvas_bin_sample_packed /
 * canvas_sample_pipeline_packed are called exactly as after canvas_upload_packed_begin. */
int32_t canvas_pack_hits2_host(const uint8_t* hits, int64_t len, uint64_t* lo_out, uint64_t* hdr_out, uint64_t* extras_out, int64_t extras_cap_words, int64_t* n_extras_out,
                               int64_t* saturated_out, int32_t threads);
int32_t canvas_upload_packed2_begin(canvas_ctx* ctx, int32_t nchr, const int64_t* h_len, const uint64_t* const* h_ref, uint64_t* const* d_ref,
                                    const uint64_t* const* h_lo, const uint64_t* const* h_hdr, const uint64_t* const* h_extras, const int64_t* h_n_extras, uint64_t* const* d_hit_planes);
/* canvas_bin_sample (below) over the planes: same arguments otherwise, same outputs bit for bit (modes 0 and 3). */
int32_t canvas_bin_sample_packed(canvas_ctx* ctx, int32_t nchr, const uint64_t* const* d_ref, const uint64_t* const* d_hit_planes, const int64_t* h_len, const int64_t* h_pos0,
                                 const uint8_t* h_chr_is_autosome, int32_t counts_per_bin, int32_t bin_size_in, int32_t mode,
                                 int32_t* d_chr, int32_t* d_start, int32_t* d_stop, int32_t* d_gc, float* d_count, int64_t cap,
                                 int32_t* h_bin_size_out, int64_t* h_nbins_per_chr, int64_t* h_nbins_total);

/* ---- CanvasBin ----------------------------------------------------------------------------------------------- */
/* InitializeAlignmentArrays (CanvasBin/CanvasBin.cs:183-200): possible[i] = char.IsUpper(referenceBases[i]).
 * d_mask must hold ceil(len/64) words; bits at and beyond len are written as 0. */
int32_t canvas_mask_from_fasta(canvas_ctx* ctx, const uint8_t* d_bases, int64_t len, uint64_t* d_mask);
/* ExcludeTagsOverlappingFilterFile (CanvasBin.cs:668-692): clear the possible bits of [start, stop) for the n BED intervals
 * of this chromosome (intervals are clipped to [0, len); the reference would throw past the end). */
int32_t canvas_mask_exclude_intervals(canvas_ctx* ctx, uint64_t* d_mask, int64_t len, int32_t n, const int32_t* h_start, const int32_t* h_stop);
/* ScreenObservedTags (CanvasBin.cs:699-716): hits[i] = 0 wherever the position is not a possible alignment. */
int32_t canvas_screen_hits(canvas_ctx* ctx, uint8_t* d_hits, const uint64_t* d_mask, int64_t len);
/* SampleHitArrays.GetRates (CanvasBin/CanvasBin.cs:30-71) + HitArray.CountSetBits (HitArray.cs:24-32) +
 * CanvasBin.CountSetBits (CanvasBin.cs:146-156): per chromosome #positions with hit>0 and popcount(mask).
 * h_observed/h_possible/h_rate have nchr entries (rate = observed/(double)possible). */
int32_t canvas_bin_rates(canvas_ctx* ctx, int32_t nchr, const uint8_t* const* d_hits, const uint64_t* const* d_mask,
                         const int64_t* h_len, int64_t* h_observed, int64_t* h_possible, double* h_rate);
/* SampleHitArrays.GetBinSize (CanvasBin.cs:73-83): (int)(countsPerBin / median(rates)); pass autosome rates only
 * (MultiSampleHitArrays, :86-110: concatenate the samples' rates). Host scalar code. */
int32_t canvas_bin_size_from_rates(const double* h_rates, int32_t n, int32_t counts_per_bin);
/* upper bound for the bin arrays of canvas_bin_genome */
int64_t canvas_bin_count_upper_bound(int32_t nchr, const int64_t* h_len, int32_t bin_size);
/* BinCounts / BinCountsForChromosome (CanvasBin.cs:416-661), no predefined bins: emits the genome's bins in chromosome
 * order as SoA (chromosome index, start, stop, gc, count as float like SampleGenomicBin.Count).  cap = capacity of the
 * output arrays; *h_nbins_total and h_nbins_per_chr (nchr entries, may be NULL) are written on return (one sync). */
int32_t canvas_bin_genome(canvas_ctx* ctx, int32_t nchr, const uint8_t* const* d_bases, const uint64_t* const* d_mask,
                          const uint8_t* const* d_hits, const int64_t* h_len, int32_t bin_size, int32_t mode,
                          int32_t* d_chr, int32_t* d_start, int32_t* d_stop, int32_t* d_gc, float* d_count, int64_t cap,
                          int64_t* h_nbins_per_chr, int64_t* h_nbins_total);

/* CanvasBin.RunSingleSample (CanvasBin.cs:914-931) in one call: when bin_size_in <= 0 the bin size is derived from the
 * autosomes' rates with counts_per_bin (-d) exactly as canvas_bin_rates + canvas_bin_size_from_rates would, and the mask
 * popcounts of the rate pass are reused for the binning pass (saves one 0.125 B/base sweep).  *h_bin_size_out = size used. */
int32_t canvas_bin_sample(canvas_ctx* ctx, int32_t nchr, const uint8_t* const* d_bases, const uint64_t* const* d_mask,
                          const uint8_t* const* d_hits, const int64_t* h_len, const uint8_t* h_chr_is_autosome,
                          int32_t counts_per_bin, int32_t bin_size_in, int32_t mode,
                          int32_t* d_chr, int32_t* d_start, int32_t* d_stop, int32_t* d_gc, float* d_count, int64_t cap,
                          int32_t* h_bin_size_out, int64_t* h_nbins_per_chr, int64_t* h_nbins_total);

/* GCContentWeighted mode (-m 5, Somatic-WGS): CanvasBin.cs:416-506 (read-GC profile from the per-position fragment lengths, mean
 * fragment size), :330-405 (observed-vs-expected weights), :626-636 (count = Round(sum min(10, hit / weight[readGC])) in float32,
 * position order).  d_fraglen = Int16 fragment length per position (0 = no read), one array per chromosome. */
int32_t canvas_bin_sample_gcweighted(canvas_ctx* ctx, int32_t nchr, const uint8_t* const* d_bases, const uint64_t* const* d_mask,
                                     const uint8_t* const* d_hits, const int16_t* const* d_fraglen, const int64_t* h_len,
                                     const uint8_t* h_chr_is_autosome, int32_t counts_per_bin, int32_t bin_size_in,
                                     int32_t* d_chr, int32_t* d_start, int32_t* d_stop, int32_t* d_gc, float* d_count, int64_t cap,
                                     int32_t* h_bin_size_out, int64_t* h_nbins_per_chr, int64_t* h_nbins_total);
/* last canvas_bin_sample_gcweighted of the context: h_out2[0] = bins whose weighted count (CanvasBin.cs:626-636) was decided from the exact sum of its terms and the rounding-error
   interval of the reference's float32 accumulation, h_out2[1] = bins that replayed the reference's additions in position order (an interval that straddles a rounding boundary) */
int32_t canvas_bin_gcw_stats(canvas_ctx* ctx, int64_t* h_out2);

/* Predefined bins (CanvasBin -n; BinCountsForChromosome with usePredefinedBins, CanvasBin.cs:575-655): count and GC of the given intervals instead of bins of a fixed
 * number of possible positions.  Bins of all chromosomes are concatenated in chromosome order, h_bin_offset[nchr+1] indexes them; start / stop (0-based, half open) are
 * given twice, on the host (validated as Utilities.LoadBedFile does) and on the device.  d_count = sum over the possible positions of hit (mode 0) or min(10, hit) (mode 3),
 * d_gc = (int)(100f * #[CcGg] / #positions) — the first bin of a chromosome starts counting at its first base that is not 'n' (:582-584). */
int32_t canvas_bin_predefined(canvas_ctx* ctx, int32_t nchr, const uint8_t* const* d_bases, const uint64_t* const* d_mask, const uint8_t* const* d_hits, const int64_t* h_len,
                              int32_t mode, const int64_t* h_bin_offset, const int32_t* h_bin_start, const int32_t* h_bin_stop, const int32_t* d_bin_start, const int32_t* d_bin_stop,
                              int32_t* d_gc, float* d_count);
/* The same with -m GCContentWeighted (the predefined-bin close shares the weighted branch, CanvasBin.cs:617-636): d_count = (int)Math.Round of the float32 sum over the bin's
 * possible positions of Math.Min(10, hit / observedVsExpectedGC[readGC]).  The mean fragment size, the read-GC profile and the weights are the whole call's — every one of the
 * nchr chromosomes enters them (BinCounts, CanvasBin.cs:427-505), whether it has bins (h_bin_offset[c] < h_bin_offset[c + 1]) or not.  Arrays 16-byte aligned as for
 * canvas_bin_sample_gcweighted; canvas_bin_gcw_stats reports decided / replayed bins. */
int32_t canvas_bin_predefined_gcweighted(canvas_ctx* ctx, int32_t nchr, const uint8_t* const* d_bases, const uint64_t* const* d_mask, const uint8_t* const* d_hits,
                                         const int16_t* const* d_fraglen, const int64_t* h_len, const int64_t* h_bin_offset, const int32_t* h_bin_start, const int32_t* h_bin_stop,
                                         const int32_t* d_bin_start, const int32_t* d_bin_stop, int32_t* d_gc, float* d_count);

/* ---- CanvasClean --------------------------------------------------------------------------------------------- */
/* CanvasClean.Main (CanvasClean/CanvasClean.cs:415-533) on the whole-genome SoA in file order, in place; bins that
 * survive are compacted to the front, *h_n_out = surviving count.  h_chr_is_autosome[nchr] answers
 * GenomeMetadata.SequenceMetadata.IsAutosome for each chromosome index.  min_bins_per_gc = the -w option (default 100).
 * h_local_sd_out receives the #localSD metric (IO.cs:83-98) or -1.  h_info (may be NULL) gets 8 int32 diagnostics:
 * [0] after size filter, [1] after outlier filter, [2] after GC strip, [3] after local-SD filter, [4] variance-normalised,
 * [5] 1 = the medians / quartiles were read off exact per-value counters (counts that are two-decimal values, as the F2 text
 * of a .binned file always is), 0 = radix selects (any other input; same results), [6] 1 = flags were -g alone (CANVAS_CLEAN_GCNORM) on whole-number counts and
 * the stage ran as RemoveBinsWithExtremeGC + NormalizeByGC (CanvasClean.cs:163-237,497-505) in three launches, in place (same results as the general chain). */
int32_t canvas_clean(canvas_ctx* ctx, int64_t n, int32_t* d_chr, int32_t* d_start, int32_t* d_stop, float* d_count,
                     int32_t* d_gc, int32_t nchr, const uint8_t* h_chr_is_autosome, uint32_t flags, int32_t min_bins_per_gc,
                     double* h_local_sd_out, int64_t* h_n_out, int32_t* h_info);

/* same with -m LOESS support (flag CANVAS_CLEAN_LOESS): h_chr_is_y[nchr] marks chrY/Y, which LoessGCNormalizer leaves out of the
 * bandwidth search (LoessGCNormalizer.cs:49-50,63-68).  LOESS-mode counts match the reference within 1e-5 relative (the floating
 * sums are re-associated); MedianByGC mode is bit-exact.  NULL h_chr_is_y = no chromosome is chrY. */
int32_t canvas_clean2(canvas_ctx* ctx, int64_t n, int32_t* d_chr, int32_t* d_start, int32_t* d_stop, float* d_count,
                      int32_t* d_gc, int32_t nchr, const uint8_t* h_chr_is_autosome, const uint8_t* h_chr_is_y, uint32_t flags,
                      int32_t min_bins_per_gc, double* h_local_sd_out, int64_t* h_n_out, int32_t* h_info);
/* canvas_clean2 for a cohort (CanvasRunner launches one CanvasClean per sample of a pedigree, CanvasRunner.cs:1010-1070): sample s uses h_n[s] and the device arrays
 * h_d_*[s] (1 .. 64 samples).  The MedianByGC stage is batch-native: the samples share every kernel launch (one launch chain and one wait for the whole cohort); LOESS
 * mode and -w < 100 run sample after sample.  Outputs per sample: h_n_out[s], h_local_sd_out[s] (may be NULL), h_info[8 * s ..] (may be NULL).  Results are exactly
 * those of nsamples canvas_clean2 calls. */
int32_t canvas_clean_batch(canvas_ctx* ctx, int32_t nsamples, const int64_t* h_n, int32_t* const* h_d_chr, int32_t* const* h_d_start, int32_t* const* h_d_stop, float* const* h_d_count,
                           int32_t* const* h_d_gc, int32_t nchr, const uint8_t* h_chr_is_autosome, const uint8_t* h_chr_is_y, uint32_t flags, int32_t min_bins_per_gc,
                           double* h_local_sd_out, int64_t* h_n_out, int32_t* h_info);
/* Pedigree workflows: Utilities.MergeMultiSampleCleanedBedFile + CanvasRunner.NormalizeCanvasClean (CanvasCommon/Utilities.cs:834-920,
 * CanvasRunner.cs:883-903): keep the bins (keyed by chromosome and start) that every sample's cleaned list still has.  Inputs: per
 * sample the SoA of CanvasClean (sorted by chromosome index, then start).  Outputs: one bin list in the first sample's order (stop taken
 * from the last sample) and, per sample, the counts of the surviving bins (h_d_out_count[s], capacity h_n[0]). */
int32_t canvas_merge_cleaned(canvas_ctx* ctx, int32_t nsamples, const int64_t* h_n, const int32_t* const* h_d_chr, const int32_t* const* h_d_start,
                             const int32_t* const* h_d_stop, const float* const* h_d_count, int32_t* d_out_chr, int32_t* d_out_start, int32_t* d_out_stop,
                             float* const* h_d_out_count, int64_t* h_n_out);
/* bins are grouped by chromosome in file order: h_chr_offset[c] = index of the first bin of chromosome c (nchr+1 entries,
 * h_chr_offset[nchr] = n); chromosomes without bins get an empty range.  Chromosome indices must be non-decreasing. */
int32_t canvas_chromosome_offsets(canvas_ctx* ctx, const int32_t* d_chr, int64_t n, int32_t nchr, int64_t* h_chr_offset);

/* The text hand-off between CanvasClean and CanvasPartition, done in memory: CanvasIO.WriteToTextFile prints the float
 * count with "{3:F2}" (CanvasCommon/IO.cs:21; .NET Core 2.x: 7 significant digits, then half-up at 2 decimals) and
 * CanvasSegment.ReadBedInput parses that text as double (CanvasCommon/CanvasSegment.cs:1146).  d_cov[i] = that double. */
int32_t canvas_quantize_f2(canvas_ctx* ctx, const float* d_count, int64_t n, double* d_cov);

/* ---- CanvasPartition ------------------------------------------------------------------------------------------ */
/* HiddenMarkovModelsRunner.Run with isPerSample (HiddenMarkovModelsRunner.cs:23-109) + BestPathViterbi (HMM.cs:62-130):
 * one sample, all chromosomes.  d_cov = concatenated coverage (double, file order), h_chr_offset[nchr+1] = chromosome
 * boundaries in bins.  d_state receives the Viterbi state (CN 0..4) per bin; chromosomes with <= 10 bins are skipped
 * (state -1), as in :69.  */
int32_t canvas_hmm_per_sample(canvas_ctx* ctx, int32_t nchr, const double* d_cov, const int64_t* h_chr_offset, int32_t* d_state);
/* -m HMM: HiddenMarkovModelsRunner.Run with isPerSample == false (HiddenMarkovModelsRunner.cs:23-152; CanvasPartition.cs:146-158) and
 * NegativeBinomialMixture.EstimateViterbiLikelihood over the genotype combinations (Distributions.cs:257-323,
 * DistributionUtilities.cs:11-40): one state path for nsamples (1..16) samples that share the bins.  h_d_cov[s] = device pointer to
 * sample s's concatenated coverage (same layout and h_chr_offset for all samples).  Emission parameters are per chromosome (median and
 * variance of that chromosome, :117-131); chromosomes with <= 10 bins are skipped (state -1). */
int32_t canvas_hmm_joint(canvas_ctx* ctx, int32_t nsamples, int32_t nchr, const double* const* h_d_cov, const int64_t* h_chr_offset, int32_t* d_state);
/* breakpoints -> segments -> segment id per bin: SegmentationInput.DeriveSegments (Segmentation.cs:83-125) +
 * SegmentationResultsProcessor.PostProcessSegments (SegmentationResultsProcessor.cs:17-129, no forbidden intervals /
 * ploidy file: those stay in the host tool).  d_is_start: 1 where a segment starts at this bin. */
int32_t canvas_segment_ids(canvas_ctx* ctx, int32_t nchr, const int64_t* h_chr_offset, const int32_t* d_state,
                           const int32_t* d_start, const int32_t* d_stop, int32_t max_inter_bin_dist, int32_t* d_segment_id,
                           int64_t* h_nsegments);
/* same, with the forbidden intervals of the -b BED file (SegmentationResultsProcessor.cs:88-111): a segment is split where the
 * midpoint of an interval lies between two bins.  h_excl_offset[nchr+1] indexes h_excl_start/stop per chromosome; intervals must be
 * sorted by end inside a chromosome (the reference walks them with a forward-only cursor). */
int32_t canvas_segment_ids_filtered(canvas_ctx* ctx, int32_t nchr, const int64_t* h_chr_offset, const int32_t* d_state,
                                    const int32_t* d_start, const int32_t* d_stop, int32_t max_inter_bin_dist,
                                    const int64_t* h_excl_offset, const int32_t* h_excl_start, const int32_t* h_excl_stop,
                                    int32_t* d_segment_id, int64_t* h_nsegments);
/* same, with the reference ploidy of CanvasPartition -p (CanvasPartition.cs:114; CanvasRunner.InvokeCanvasPartition always passes it, CanvasRunner.cs:950):
 * a new segment also starts where PloidyInfo.IsUniformReferencePloidy is false for the one-based interval [previous bin end (or 1), this bin's end]
 * (SegmentationResultsProcessor.cs:117-128, PloidyInfo.cs:78-110).  h_ploidy_offset[nchr+1] indexes the records of the ploidy VCF per chromosome
 * (PloidyInfo.LoadPloidyFromVcfFile, PloidyInfo.cs:128-165): h_ploidy_start = POS (one-based), h_ploidy_end = INFO/END, h_ploidy_cn = the CN genotype
 * field ("." = 2), 0..4.  A chromosome without records behaves like one that is not in the VCF.  h_ploidy_offset == NULL: no -p.  The forbidden
 * intervals are optional as above (h_excl_offset == NULL: no -b). */
int32_t canvas_segment_ids_ploidy(canvas_ctx* ctx, int32_t nchr, const int64_t* h_chr_offset, const int32_t* d_state, const int32_t* d_start,
                                  const int32_t* d_stop, int32_t max_inter_bin_dist,
                                  const int64_t* h_excl_offset, const int32_t* h_excl_start, const int32_t* h_excl_stop,
                                  const int64_t* h_ploidy_offset, const int32_t* h_ploidy_start, const int32_t* h_ploidy_end, const int32_t* h_ploidy_cn,
                                  int32_t* d_segment_id, int64_t* h_nsegments);
/* CanvasPartition --evenness-metric-file (Somatic-WGS, CanvasRunner.cs:958-960): SegmentationInput.GetEvennessScore (Segmentation.cs:260-296) as
 * WaveletsRunner.Run computes it before segmenting (WaveletsRunner.cs:58-67).  d_cov / h_chr_offset as for canvas_cbs; window_size =
 * CanvasPartitionParameters.EvennessScoreWindow (100000).  *h_valid = 0 when the reference's Quartiles / Median throw (fewer than two 10000-bin windows,
 * or no window of window_size: the reference then writes no metric file), otherwise *h_score is the value written as "#evenness\t<score>"
 * (IO.cs:88-98).  Bit-exact: the window sums are evaluated in list order. */
int32_t canvas_evenness_score(canvas_ctx* ctx, int32_t nchr, const double* d_cov, const int64_t* h_chr_offset, int32_t window_size, double* h_score, int32_t* h_valid);
/* GenomeSegmentationResults.SplitOverlappingSegments (GenomeSegmentationResults.cs:18-55) for one chromosome: host scalar code. */
int32_t canvas_split_overlapping(int32_t nsamples, const uint32_t* const* h_start, const uint32_t* const* h_end, const int32_t* h_nseg,
                                 uint32_t* h_out_start, uint32_t* h_out_end, int32_t cap, int32_t* h_nout);
/* CBSRunner.Run / ChangePoint.ChangePoints (CBSRunner.cs:40-151, ChangePoint.cs:44-153), undo = None.
 * d_seg_len receives per chromosome the segment lengths, written at d_seg_len + h_chr_offset[c]; h_nseg[c] = count. */
int32_t canvas_cbs(canvas_ctx* ctx, int32_t nchr, const double* d_cov, const int64_t* h_chr_offset, double alpha, uint32_t nperm,
                   int32_t* d_seg_len, int32_t* h_nseg, int64_t* h_stats);

/* same with -s Prune (undo = 1, ChangePoint.cs:205-271 + Prune.cs, cut-off 0.05), -s SDUndo (undo = 2, ChangePoint.cs:155-196; undo_sd = 3 in
   the reference) or None (0). */
int32_t canvas_cbs_undo(canvas_ctx* ctx, int32_t nchr, const double* d_cov, const int64_t* h_chr_offset, double alpha, uint32_t nperm,
                        int32_t undo, double undo_sd, int32_t* d_seg_len, int32_t* h_nseg, int64_t* h_stats);

/* counters of the device permutation engine for the last canvas_cbs / canvas_cbs_undo call: [0] permutations evaluated on the device
 * (XPerm + HTMaxP, CBSTStatistic.cs:354-586), [1] on the host (segments shorter than 1024 bins, TMaxP), [2] device permutations that
 * were re-evaluated in the reference's exact order because the observed statistic fell inside the rounding interval, [3] batches,
 * [4]/[5] test hook CANVAS_CBS_TEST_VERIFY=1: device intervals checked against the exact statistic / violations. */
int32_t canvas_cbs_device_stats(canvas_ctx* ctx, int64_t* h_out6);
/* the edge tests (CBSTStatistic.TPermP, CBSTStatistic.cs:947-1024) of the last canvas_cbs / canvas_cbs_undo call: [0] tests run by the device kernel (one lane walks the
 * chain of nPerm x min(n1, n2) dependent swaps; CANVAS_CBS_DEVICE_TPERMP=1 — the default is the host chain, which is 30x faster per swap), [1] swaps of all edge tests. */
int32_t canvas_cbs_tpermp_stats(canvas_ctx* ctx, int64_t* h_out2);
/* last canvas_cbs / canvas_cbs_undo call: the analytic tail probability of the hybrid test (TailProbability.TailP, TailProbability.cs:21-85) only feeds two decisions of
 * FindChangePoints (ChangePoint.cs:318-323).  [0] calls decided from the device evaluation of its series (accepted only when every value within 1e-8 relative gives the same
 * decisions), [1] calls recomputed with the host libm in the reference's order. */
int32_t canvas_cbs_tailp_stats(canvas_ctx* ctx, int64_t* h_out2);
/* Diagnostic / test entry: TailProbability.Nu (TailProbability.cs:52-85) of n <= 100 arguments through the device series exactly as canvas_cbs evaluates it (k_tail_nu: the
 * first 512 terms one by one, every later block of the series from the Euler-Maclaurin formula); h_flag[i] != 0: a stopping comparison of the series was too close to call and
 * canvas_cbs would redo the call with the host series. */
int32_t canvas_cbs_tail_probe(canvas_ctx* ctx, const double* h_x, int32_t n, double tol, double* h_nu, int32_t* h_flag);
/* Diagnostic / test entry: ONE batch of nb permutations of the centred segment h_x[n] (n >= 1024, n * nb <= 2^30) through the device permutation engine exactly as the hybrid test of
 * FindChangePoints runs it — XPerm (ChangePoint.cs:407-421) + HTMaxP with k = 25, minimum width 2 (CBSTStatistic.cs:354-586) — from MersenneTwister(seed).  kernel selects the
 * permutation kernel: 0 counting sort + pointer doubling, 1 block-wise simulation of the swaps in global memory, 2 range-partitioned simulation in LDS.  h_lohi[2 nb]: the interval
 * the engine returns for every permutation's statistic (the exact value lies inside; the stopping rule re-evaluates a permutation on the host only when the observed statistic does too).
 * h_ms3 (optional): milliseconds of the generator's sequential part, its strided part, and the permutation + statistic kernel.  No reference counterpart: the engine's test bench. */
int32_t canvas_cbs_perm_probe(canvas_ctx* ctx, const double* h_x, int32_t n, uint32_t seed, int32_t nb, int32_t kernel, double tss, double* h_lohi, double* h_ms3);
/* Host-only (no context, no GPU): the sequential stopping boundary canvas_cbs uses for (nperm, alpha) — GetBoundary.ComputeBoundary (GetBoundary.cs:19-157) with eta = 0.05 as
 * CBSRunner passes it: maxOnes (maxOnes + 1) / 2 entries with maxOnes = floor(nperm alpha) + 1.  Returns the number of entries (or a negative error code).  The library
 * evaluates the table's scans on its host thread pool with the scans' own evaluations and comparisons; exposed so that the table can be checked without a device. */
int64_t canvas_cbs_boundary(uint32_t nperm, double alpha, uint32_t* h_out, int64_t cap);
/* The draw streams of canvas_cbs ahead of its first call.  The k-th chromosome's generator is MersenneTwister(seed_k) with seed_k from MersenneTwister(0) in file order
 * (CBSRunner.cs:107-112) and is consumed strictly in sequence by XPerm / TPermP (ChangePoint.cs:407-421, CBSTStatistic.cs:1009): the words are constants of the method.  The
 * library keeps them per context in device memory (generated once, extended on demand, bounded by CANVAS_CBS_CACHE_GB — default 30 % of the device's memory, 0 = off) and
 * canvas_cbs reads its permutations' draws out of them.  canvas_cbs_prefetch starts the generator for the first `words_per_chromosome` draws of the first nchr streams on a
 * thread and stream of its own — and creates, on another, the streams and request tables of canvas_cbs's launchers (a stream costs ~5 ms on this runtime: 60 ms of a first
 * call) — and returns at once: a host calls it while it is still reading its input (CanvasPartition does).  Optional — canvas_cbs asks for what it needs itself.  canvas_cbs_cache_stats: h_out6 = {draws the last canvas_cbs call read out of the cache, draws it generated inside its batches (cache off / bound reached),
 * draws the cache's generator produced during the call, generator states fetched for host code, bytes of device memory the cache holds, draws it holds}. */
int32_t canvas_cbs_prefetch(canvas_ctx* ctx, int32_t nchr, int64_t words_per_chromosome);
/* Diagnostic / test entry: nwords tempered outputs of the chromosome-th generator (0-based, file order) from output number `position` on, read out of the context's cache
 * (generated now if they are not there yet; CANVAS_ERR_CAPACITY when the bound of the cache does not reach that far).  What canvas_cbs's permutation kernels read. */
int32_t canvas_cbs_stream_read(canvas_ctx* ctx, int32_t chromosome, int64_t position, int64_t nwords, uint32_t* h_out);
int32_t canvas_cbs_cache_stats(canvas_ctx* ctx, int64_t* h_out6);
/* Host-only (no context, no GPU): the seeds of the per-chromosome generators canvas_cbs uses, in file order — new MersenneTwister(0) followed by one NextFullRangeInt32() per
 * chromosome (CBSRunner.cs:107-112).  h_out[nchr]; h_variant (optional): which reading of MathNet's NextBytes is in force (0 / 1 / 2, include/canvas_mathnet.h: the one
 * assumption of this path that could not be checked without a .NET SDK; CANVAS_MATHNET_SEED_BYTES selects it at run time).  Returns 0 or a negative error code. */
int32_t canvas_cbs_seeds(int32_t nchr, int32_t* h_out, int32_t* h_variant);

/* CanvasPartition -m Wavelets, the reference's default method: WaveletsRunner.Run up to the breakpoints (WaveletsRunner.cs:52-150 =
 * SegmentationInput.GetCoverageVariability / FactorOfThreeCoverageVariabilities, Segmentation.cs:297-429, then
 * WaveletSegmentation.HaarWavelets per chromosome, WaveletSegmentation.cs:373-425: unbalanced Haar decomposition, HardThresh,
 * reconstruction, GetBreakpointsAfterHealingBadSplits and, with is_germline (-g), RefineSegments).  d_cov / h_chr_offset as for canvas_cbs.
 * The parameters are WaveletsRunnerParams' (WaveletsRunner.cs:28-41): threshold_lower = ThresholdLowerMaf (0.05), threshold_upper = 80,
 * mad_factor = MadFactor (5), variability_window = EvennessScoreWindow (100000), min_size = 10.  h_breakpoints[h_bp_offset[c] .. h_bp_offset[c+1])
 * are the bin indices where a segment starts on chromosome c (empty for a chromosome of at most min_size bins): what the reference
 * hands to SegmentationInput.DeriveSegments (Segmentation.cs:83-125).  Coverage must be finite. */
int32_t canvas_wavelets(canvas_ctx* ctx, int32_t nchr, const double* d_cov, const int64_t* h_chr_offset, int32_t is_germline,
                        double threshold_lower, double threshold_upper, double mad_factor, int32_t variability_window, int32_t min_size,
                        int32_t* h_breakpoints, int64_t cap, int64_t* h_bp_offset);
/* last canvas_wavelets call: [0] tree levels processed, [1] nodes whose shortcut division disagreed with the IEEE one and were recomputed */
int32_t canvas_wavelets_stats(canvas_ctx* ctx, int64_t* h_out2);
/* last canvas_wavelets call: [0] long nodes whose arg-max was decided from the closed form (exact prefix sums + rounding-error bound of WaveletSegmentation.cs:19-48),
   [1] long nodes the bound could not decide (exact chain), [2] long nodes chained for their coefficient (candidates to survive HardThresh, WaveletSegmentation.cs:73-117),
   [3] 1 if the closed form was in use (coverage of non-negative two-decimal values) */
int32_t canvas_wavelets_decisions(canvas_ctx* ctx, int64_t* h_out4);

/* ---- one sample through the whole path in one call -------------------------------------------------------------
 * canvas_bin_sample -> canvas_clean2 -> canvas_quantize_f2 -> canvas_chromosome_offsets -> canvas_hmm_per_sample -> canvas_segment_ids with
 * the arguments of those calls (h_chr_is_y may be NULL).  Outputs: the cleaned bins in d_chr..d_count (first *h_nbins_clean entries), d_cov,
 * d_state and d_segment_id per cleaned bin, h_chr_offset[nchr+1], the bin size, bin counts, #localSD and the number of segments.  Nothing
 * is computed that the individual entry points do not compute: it only saves the caller's per-call overhead between the stages. */
int32_t canvas_sample_pipeline(canvas_ctx* ctx, int32_t nchr, const uint8_t* const* d_bases, const uint64_t* const* d_mask, const uint8_t* const* d_hits,
                               const int64_t* h_len, const uint8_t* h_chr_is_autosome, const uint8_t* h_chr_is_y, int32_t counts_per_bin, int32_t bin_size_in,
                               int32_t mode, uint32_t clean_flags, int32_t min_bins_per_gc, int32_t max_inter_bin_dist,
                               int32_t* d_chr, int32_t* d_start, int32_t* d_stop, int32_t* d_gc, float* d_count, int64_t cap,
                               double* d_cov, int32_t* d_state, int32_t* d_segment_id,
                               int32_t* h_bin_size, int64_t* h_nbins, int64_t* h_nbins_clean, double* h_local_sd, int64_t* h_chr_offset, int64_t* h_nsegments);

/* canvas_sample_pipeline over the packed planes (canvas_bin_sample_packed). */
int32_t canvas_sample_pipeline_packed(canvas_ctx* ctx, int32_t nchr, const uint64_t* const* d_ref, const uint64_t* const* d_hit_planes, const int64_t* h_len, const int64_t* h_pos0,
                                      const uint8_t* h_chr_is_autosome, const uint8_t* h_chr_is_y, int32_t counts_per_bin, int32_t bin_size_in,
                                      int32_t mode, uint32_t clean_flags, int32_t min_bins_per_gc, int32_t max_inter_bin_dist,
                                      int32_t* d_chr, int32_t* d_start, int32_t* d_stop, int32_t* d_gc, float* d_count, int64_t cap,
                                      double* d_cov, int32_t* d_state, int32_t* d_segment_id,
                                      int32_t* h_bin_size, int64_t* h_nbins, int64_t* h_nbins_clean, double* h_local_sd, int64_t* h_chr_offset, int64_t* h_nsegments);

/* ---- CanvasNormalize, ratio path (enrichment / tumour-normal workflows) -------------------------------------------------
 * canvas_normalize_reference = WeightedAverageReferenceGenerator.Run for more than one control sample (WeightedAverageReferenceGenerator.cs:
 * 38-68): weight_i = 1 / median_i (0 if the median is not positive), normalised to sum 1, median_i = BinCounts.OnTargetMedianBinCount
 * (BinCounts.cs:36-60) over the bins listed in d_on_target_idx (NULL: all bins; the list is what BinCounts.LoadBinCounts derives from the
 * Nextera manifest, BinCounts.cs:118-166); d_weighted[j] = sum_i weight_i * counts_i[j].  h_d_counts: nsamples device pointers to n doubles
 * each (the 4th column of the control .binned files, double.Parse).  With one control sample the reference copies the file instead. */
int32_t canvas_normalize_reference(canvas_ctx* ctx, int32_t nsamples, const double* const* h_d_counts, int64_t n, const int32_t* d_on_target_idx, int64_t n_on_target,
                                   double* d_weighted, double* h_weights);
/* LSNormRatioCalculator.Run (mode 0, LSNormRatioCalculator.cs:20-48: library-size factor = reference median / sample median over the on-target
 * bins, bins whose reference count is below 1 are dropped) or RawRatioCalculator.Run (mode 1, RawRatioCalculator.cs:21-46: bins whose reference
 * count lies outside [min_ref, max_ref] are dropped), followed by CanvasNormalizeUtilities.RatiosToCounts (CanvasNormalizeUtilities.cs:23-33:
 * count = ratio * 40 * ploidy / 2; d_ploidy = reference copy number per bin from the ploidy VCF, NULL = 2).  d_sample / d_reference are the float
 * counts of the two .binned files.  Outputs (capacity n): d_keep_idx = indices of the bins that are kept, d_ratio / d_count per kept bin. */
int32_t canvas_normalize_ratio(canvas_ctx* ctx, int64_t n, const float* d_sample, const float* d_reference, const int32_t* d_on_target_idx, int64_t n_on_target,
                               int32_t mode, double min_ref, double max_ref, const int32_t* d_ploidy, int32_t* d_keep_idx, float* d_ratio, float* d_count,
                               int64_t* h_n_out, double* h_library_size_factor);

/* ---- multi-GPU (one process per GPU; chromosomes sharded across ranks) ------------------------------------------ */
int32_t canvas_comm_unique_id(void* h_id128);  /* ncclGetUniqueId, 128 bytes, rank 0 */
int32_t canvas_comm_init(canvas_ctx* ctx, int32_t rank, int32_t nranks, const void* h_id128);
/* transport for ranks that cannot form an RCCL communicator (several ranks on one GPU, a host with its own MPI): fn must all-gather bytes_per_rank bytes from
 * `send` of every rank into `recv` (rank order), host memory, and return 0.  The library stages the device buffers through pinned host memory around it. */
typedef int32_t (*canvas_host_allgather_fn)(void* user, const void* send, int64_t bytes_per_rank, void* recv);
int32_t canvas_comm_init_host(canvas_ctx* ctx, int32_t rank, int32_t nranks, canvas_host_allgather_fn fn, void* user);
/* Sub-communicators for samples x chromosome groups (BASELINE configs[3]; the reference runs the samples of a pedigree as independent CanvasBin / CanvasClean tasks,
 * CanvasRunner.cs:123-128): the ranks that pass the same color form an RCCL communicator of their own (ncclCommSplit), ordered by key, and every sharded call that
 * follows runs inside it; canvas_comm_restore goes back to the communicator of canvas_comm_init (the bin size of a pedigree and its bin intersection span all samples).
 * Collective over the parent communicator.  The host-callback transport has no split: its caller hands canvas_comm_init_host the callback of the group.
 * canvas_comm_rank: rank and size in the communicator the next collective will use. */
int32_t canvas_comm_split(canvas_ctx* ctx, int32_t color, int32_t key);
int32_t canvas_comm_restore(canvas_ctx* ctx);
int32_t canvas_comm_rank(canvas_ctx* ctx, int32_t* h_rank, int32_t* h_nranks);
/* the single RCCL all-gather of the path: every rank contributes nlocal int32 boundary records (padded to max_per_rank) */
int32_t canvas_allgather_boundaries(canvas_ctx* ctx, const int32_t* d_local, int32_t nlocal, int32_t max_per_rank,
                                    int32_t* d_all, int32_t* h_counts);

/* ONE sample, chromosomes sharded over the ranks (BASELINE configs[3], [4]): canvas_sample_pipeline for a rank that holds the per-base arrays of the
 * chromosomes with h_chr_owner[c] == its rank only (entries of the other chromosomes in d_bases / d_mask / d_hits are ignored; h_len, the autosome flags and the owner
 * table describe the whole genome and are the same on every rank).  The reference's per-chromosome tasks (CanvasBin.cs:513-539, HiddenMarkovModelsRunner.cs:51-104) run on
 * the owner; the genome-wide couplings are exchanged with three all-gathers on the communicator of canvas_comm_init / canvas_comm_init_host: the per-chromosome
 * (observed, possible) table (one bin size for everybody, CanvasBin.cs:73-83), the owned bins (16 B/bin), and — the collective north_star names — the segment boundary
 * records [n, (chr, startBin, endBin, state) ...] through canvas_allgather_boundaries.  CanvasClean runs on the gathered whole-genome SoA on every rank (deterministic,
 * redundant).  Every rank returns the same outputs as canvas_sample_pipeline on one GPU, bit for bit: all cleaned bins, coverage, states and the running segment ids
 * in file order (SegmentationResultsProcessor.cs:57-62).  Must be called by all ranks.  Modes 0 and 3. */
/* CanvasBin alone with the chromosomes sharded over the ranks (BASELINE configs[4]: the tumour's GCContentWeighted bins): canvas_bin_sample / canvas_bin_sample_gcweighted for
 * a rank that holds the arrays of its own chromosomes only; every rank ends with the bins of the whole genome in file order, bit-identical to the single-GPU call.
 * d_fraglen: mode 5 only (NULL otherwise).  Mode 5 adds two reductions over the ranks in front of the rate table: the per-chromosome NonZeroMeans of the fragment
 * lengths (MeanFragmentSize, CanvasBin.cs:164-174) and the 2 x 101 counters of the read-GC profile (CanvasBin.cs:372-391).  Must be called by all ranks; a failure on one
 * rank is announced in every exchange and fails all of them. */
int32_t canvas_bin_sample_sharded(canvas_ctx* ctx, int32_t nchr, const int32_t* h_chr_owner, const uint8_t* const* d_bases, const uint64_t* const* d_mask,
                                  const uint8_t* const* d_hits, const int16_t* const* d_fraglen, const int64_t* h_len, const uint8_t* h_chr_is_autosome,
                                  int32_t counts_per_bin, int32_t bin_size_in, int32_t mode, int32_t* d_chr, int32_t* d_start, int32_t* d_stop, int32_t* d_gc, float* d_count,
                                  int64_t cap, int32_t* h_bin_size, int64_t* h_nbins);
int32_t canvas_sample_pipeline_sharded(canvas_ctx* ctx, int32_t nchr, const int32_t* h_chr_owner, const uint8_t* const* d_bases, const uint64_t* const* d_mask,
                                       const uint8_t* const* d_hits, const int64_t* h_len, const uint8_t* h_chr_is_autosome, const uint8_t* h_chr_is_y,
                                       int32_t counts_per_bin, int32_t bin_size_in, int32_t mode, uint32_t clean_flags, int32_t min_bins_per_gc, int32_t max_inter_bin_dist,
                                       int32_t* d_chr, int32_t* d_start, int32_t* d_stop, int32_t* d_gc, float* d_count, int64_t cap,
                                       double* d_cov, int32_t* d_state, int32_t* d_segment_id,
                                       int32_t* h_bin_size, int64_t* h_nbins, int64_t* h_nbins_clean, double* h_local_sd, int64_t* h_chr_offset, int64_t* h_nsegments);
/* canvas_sample_pipeline_sharded over the packed planes of canvas_bin_sample_packed (0.75 B/base; d_ref / d_hit_planes / h_pos0 as for canvas_sample_pipeline_packed,
 * entries of the chromosomes this rank does not own are ignored): same exchanges, same outputs, bit for bit. */
int32_t canvas_sample_pipeline_sharded_packed(canvas_ctx* ctx, int32_t nchr, const int32_t* h_chr_owner, const uint64_t* const* d_ref, const uint64_t* const* d_hit_planes,
                                              const int64_t* h_len, const int64_t* h_pos0, const uint8_t* h_chr_is_autosome, const uint8_t* h_chr_is_y,
                                              int32_t counts_per_bin, int32_t bin_size_in, int32_t mode, uint32_t clean_flags, int32_t min_bins_per_gc, int32_t max_inter_bin_dist,
                                              int32_t* d_chr, int32_t* d_start, int32_t* d_stop, int32_t* d_gc, float* d_count, int64_t cap,
                                              double* d_cov, int32_t* d_state, int32_t* d_segment_id,
                                              int32_t* h_bin_size, int64_t* h_nbins, int64_t* h_nbins_clean, double* h_local_sd, int64_t* h_chr_offset, int64_t* h_nsegments);
/* last canvas_sample_pipeline_sharded call: [0] ranks, [1] chromosomes owned, [2] bins binned locally, [3] bytes this rank contributed to the bins all-gather,
 * [4] boundary records of this rank, [5] bytes per rank of the boundary all-gather */
int32_t canvas_sharded_stats(canvas_ctx* ctx, int64_t* h_out6);
/* CanvasPartition -m CBS / -m Wavelets with the chromosomes sharded over the ranks (the tumour / normal flow of BASELINE configs[4] ends in CBS, the
 * reference's default method is Wavelets).  d_cov / h_chr_offset describe the WHOLE cleaned coverage and are the same on every rank (what canvas_sample_pipeline_sharded
 * leaves everywhere); a rank segments the chromosomes with h_chr_owner[c] == its rank — the reference's own per-chromosome tasks (CBSRunner.cs:62-89, WaveletsRunner.cs:115-135)
 * — while everything that couples chromosomes is computed from the whole coverage on every rank: the per-chromosome seeds drawn in file order and the genome-wide trimmed SD of
 * SDUndo (CBSRunner.cs:102-112), the coverage variability (Segmentation.cs:297-330).  ONE exchange of variable-length lists (the mechanism of
 * canvas_allgather_boundaries) then gives every rank the complete result, identical to canvas_cbs_undo / canvas_wavelets on one GPU.  Must be called by all ranks; a rank
 * that fails locally still takes part in the exchange and every rank returns an error.  h_stats of canvas_cbs_sharded counts this rank's chromosomes only. */
int32_t canvas_cbs_sharded(canvas_ctx* ctx, int32_t nchr, const int32_t* h_chr_owner, const double* d_cov, const int64_t* h_chr_offset, double alpha, uint32_t nperm,
                           int32_t undo, double undo_sd, int32_t* d_seg_len, int32_t* h_nseg, int64_t* h_stats);
/* PerSampleHMM with the chromosomes sharded over the ranks, on a coverage EVERY rank holds (d_cov / h_chr_offset: the whole sample, h_chr_offset[0] = 0) — e.g. behind the bin
 * intersection of a pedigree, which lies between CanvasClean and CanvasPartition (Utilities.cs:834-920).  Emission parameters from the quartiles of the whole coverage
 * (HiddenMarkovModelsRunner.cs:36-50); a rank decodes its own chromosomes, one exchange of state runs gives every rank d_state[0 .. N): identical to canvas_hmm_per_sample. */
int32_t canvas_hmm_per_sample_sharded(canvas_ctx* ctx, int32_t nchr, const int32_t* h_chr_owner, const double* d_cov, const int64_t* h_chr_offset, int32_t* d_state);
int32_t canvas_wavelets_sharded(canvas_ctx* ctx, int32_t nchr, const int32_t* h_chr_owner, const double* d_cov, const int64_t* h_chr_offset, int32_t is_germline,
                                double threshold_lower, double threshold_upper, double mad_factor, int32_t variability_window, int32_t min_size,
                                int32_t* h_breakpoints, int64_t cap, int64_t* h_bp_offset);

/* The sample axis ("samples of a trio add a second axis"; BASELINE configs[3]): one sample of a pedigree per rank.  CanvasBin / CanvasClean / CanvasPartition of
 * a sample run on its rank with the single-GPU entry points; the two places where the reference couples the samples are exchanges:
 *   canvas_allgather_host         bytes_per_rank host bytes from every rank, rank order — the per-chromosome rates of canvas_bin_rates, from which every rank derives the
 *                                 multi-sample bin size (MultiSampleHitArrays / GetBinSize over all samples' autosomes, CanvasBin.cs:86-110) with canvas_bin_size_from_rates;
 *   canvas_merge_cleaned_sharded  MergeMultiSampleCleanedBedFile (Utilities.cs:834-920) with the samples in rank order: this rank's cleaned SoA in, the merged bin list
 *                                 (identical on every rank; capacity cap >= the first sample's bin count) and THIS sample's counts of the surviving bins out — what
 *                                 canvas_merge_cleaned returns for that sample on one GPU.  The (chr, start, stop) columns travel (12 B per bin), padded to the largest sample.
 * Both must be called by all ranks; a rank that fails locally announces it in the exchange and every rank returns an error. */
int32_t canvas_allgather_host(canvas_ctx* ctx, const void* h_send, int64_t bytes_per_rank, void* h_recv);
int32_t canvas_merge_cleaned_sharded(canvas_ctx* ctx, int64_t n_mine, const int32_t* d_chr, const int32_t* d_start, const int32_t* d_stop, const float* d_count,
                                     int32_t* d_out_chr, int32_t* d_out_start, int32_t* d_out_stop, float* d_out_count, int64_t cap, int64_t* h_n_out);

/* ---- profiling hooks (hipEvent pairs recorded on the context's stream around the named kernels) --------------------- */
/* on: 0 off; 1 every named scope; 2 only the scopes around the dominant (HBM-bound) kernel of CanvasBin — "bin_summary", "bin_summary_packed", "bin_pass",
   "bin_tile_stats" — so that a timed pass carries two event records instead of a dozen (each scope costs two barrier packets on the stream) */
int32_t canvas_profile_enable(canvas_ctx* ctx, int32_t on);
/* name: "bin_pass", "bin_tile_stats", "viterbi"; returns accumulated ms and launch count since the last reset */
int32_t canvas_profile_get(canvas_ctx* ctx, const char* name, double* h_ms_total, int32_t* h_launches, int32_t reset);

#ifdef __cplusplus
}
#endif
#endif
