// One sample through the whole read-depth path in one call: CanvasBin (rates, bin size, bins) -> CanvasClean -> the F2 text hand-off ->
// CanvasPartition -m PerSampleHMM -> segment ids.  A host that keeps the three modules in one process makes this one
// call instead of six: nothing is computed here that the individual entry points do not compute, the call only removes the host
// language's per-call overhead between the stages (arrays stay in HBM, five small results cross PCIe).
#include "common.hpp"
#include <vector>
#include <chrono>
#include <cstdio>

// h_pos0 != NULL: d_bases / d_hits are the packed planes of canvas_bin_sample_packed (d_mask unused)
static int32_t sample_pipeline_impl(canvas_ctx* ctx, int32_t nchr, const uint8_t* const* d_bases, const uint64_t* const* d_mask, const uint8_t* const* d_hits, const int64_t* h_pos0,
                                    const int64_t* h_len, const uint8_t* h_chr_is_autosome, const uint8_t* h_chr_is_y, int32_t counts_per_bin, int32_t bin_size_in,
                                    int32_t mode, uint32_t clean_flags, int32_t min_bins_per_gc, int32_t max_inter_bin_dist,
                                    int32_t* d_chr, int32_t* d_start, int32_t* d_stop, int32_t* d_gc, float* d_count, int64_t cap,
                                    double* d_cov, int32_t* d_state, int32_t* d_segment_id,
                                    int32_t* h_bin_size, int64_t* h_nbins, int64_t* h_nbins_clean, double* h_local_sd, int64_t* h_chr_offset, int64_t* h_nsegments) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (!d_cov || !d_state || !d_segment_id || !h_chr_offset) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_sample_pipeline: bad arguments");
    static const bool timing = cvx_hook("CANVAS_PIPELINE_TIMING") != nullptr;      // host wall time of every stage call (each ends in a synchronisation) and of the pause since the previous call returned
    static thread_local std::chrono::steady_clock::time_point lastReturn; static thread_local bool haveLast = false;
    auto tNow = []() { return std::chrono::steady_clock::now(); };
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    const auto t0 = tNow();
    ProfScope psSpan(ctx, "pass_span", true);      // the pass on the device timeline: from where the stream stands when the call starts to behind its last kernel (bench.py: hand-over between passes = wall - span)
    int32_t binSize = 0; int64_t total = 0, nClean = 0, nseg = 0; double lsd = -1.0; int32_t info[8];
    std::vector<int64_t> perChr((size_t)nchr);
    int32_t rc = h_pos0 ? canvas_bin_sample_packed(ctx, nchr, (const uint64_t* const*)d_bases, (const uint64_t* const*)d_hits, h_len, h_pos0, h_chr_is_autosome, counts_per_bin, bin_size_in, mode,
                                                   d_chr, d_start, d_stop, d_gc, d_count, cap, &binSize, perChr.data(), &total)
                        : canvas_bin_sample(ctx, nchr, d_bases, d_mask, d_hits, h_len, h_chr_is_autosome, counts_per_bin, bin_size_in, mode, d_chr, d_start, d_stop, d_gc, d_count, cap,
                                            &binSize, perChr.data(), &total);
    if (rc) return rc;
    const auto t1 = tNow();
    if (h_bin_size) *h_bin_size = binSize;
    if (h_nbins) *h_nbins = total;
    // CanvasClean, the F2 hand-off and the chromosome offsets under ONE synchronisation (the quantisation and the offsets are enqueued behind CanvasClean with the bin count
    // read on the device; the quantisation also counts the genome-wide quartiles PerSampleHMM starts from)
    const void* hCovQ = nullptr;
    rc = cvx_clean_f2_offsets(ctx, total, d_chr, d_start, d_stop, d_count, d_gc, nchr, h_chr_is_autosome, h_chr_is_y, clean_flags, min_bins_per_gc, d_cov, &lsd, &nClean, info, h_chr_offset, &hCovQ);
    if (rc) return rc;
    const auto t2 = tNow();
    if (h_nbins_clean) *h_nbins_clean = nClean;
    if (h_local_sd) *h_local_sd = lsd;
    const auto t3 = tNow();
    // PerSampleHMM with the segment ids enqueued behind the verification of its speculative pass (hCovQ == NULL: the quartiles are still to be taken, inside)
    rc = cvx_hmm_per_sample_segments(ctx, nchr, d_cov, h_chr_offset, d_state, hCovQ, d_start, d_stop, max_inter_bin_dist, d_segment_id, &nseg); if (rc) return rc;
    const auto t4 = tNow();
    if (h_nsegments) *h_nsegments = nseg;
    if (timing) { const auto t5 = tNow(); fprintf(stderr, "pipeline us: since the previous call returned %.0f | bin %.0f clean %.0f f2+offsets %.0f hmm %.0f segment ids %.0f | total %.0f\n", haveLast ? us(lastReturn, t0) : 0.0, us(t0, t1), us(t1, t2), us(t2, t3), us(t3, t4), us(t4, t5), us(t0, t5)); }
    lastReturn = tNow(); haveLast = true;
    return CANVAS_OK;
}

extern "C" int32_t canvas_sample_pipeline(canvas_ctx* ctx, int32_t nchr, const uint8_t* const* d_bases, const uint64_t* const* d_mask, const uint8_t* const* d_hits,
                                          const int64_t* h_len, const uint8_t* h_chr_is_autosome, const uint8_t* h_chr_is_y, int32_t counts_per_bin, int32_t bin_size_in,
                                          int32_t mode, uint32_t clean_flags, int32_t min_bins_per_gc, int32_t max_inter_bin_dist,
                                          int32_t* d_chr, int32_t* d_start, int32_t* d_stop, int32_t* d_gc, float* d_count, int64_t cap,
                                          double* d_cov, int32_t* d_state, int32_t* d_segment_id,
                                          int32_t* h_bin_size, int64_t* h_nbins, int64_t* h_nbins_clean, double* h_local_sd, int64_t* h_chr_offset, int64_t* h_nsegments) {
    return sample_pipeline_impl(ctx, nchr, d_bases, d_mask, d_hits, nullptr, h_len, h_chr_is_autosome, h_chr_is_y, counts_per_bin, bin_size_in, mode, clean_flags, min_bins_per_gc,
                                max_inter_bin_dist, d_chr, d_start, d_stop, d_gc, d_count, cap, d_cov, d_state, d_segment_id, h_bin_size, h_nbins, h_nbins_clean, h_local_sd, h_chr_offset, h_nsegments);
}
// the same call over the packed planes (canvas_bin_sample_packed)
extern "C" int32_t canvas_sample_pipeline_packed(canvas_ctx* ctx, int32_t nchr, const uint64_t* const* d_ref, const uint64_t* const* d_hit_planes, const int64_t* h_len, const int64_t* h_pos0,
                                                 const uint8_t* h_chr_is_autosome, const uint8_t* h_chr_is_y, int32_t counts_per_bin, int32_t bin_size_in,
                                                 int32_t mode, uint32_t clean_flags, int32_t min_bins_per_gc, int32_t max_inter_bin_dist,
                                                 int32_t* d_chr, int32_t* d_start, int32_t* d_stop, int32_t* d_gc, float* d_count, int64_t cap,
                                                 double* d_cov, int32_t* d_state, int32_t* d_segment_id,
                                                 int32_t* h_bin_size, int64_t* h_nbins, int64_t* h_nbins_clean, double* h_local_sd, int64_t* h_chr_offset, int64_t* h_nsegments) {
    if (ctx && !h_pos0) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_sample_pipeline_packed: pos0 missing");
    return sample_pipeline_impl(ctx, nchr, (const uint8_t* const*)d_ref, d_ref, (const uint8_t* const*)d_hit_planes, h_pos0, h_len, h_chr_is_autosome, h_chr_is_y, counts_per_bin, bin_size_in, mode,
                                clean_flags, min_bins_per_gc, max_inter_bin_dist, d_chr, d_start, d_stop, d_gc, d_count, cap, d_cov, d_state, d_segment_id, h_bin_size, h_nbins, h_nbins_clean,
                                h_local_sd, h_chr_offset, h_nsegments);
}
