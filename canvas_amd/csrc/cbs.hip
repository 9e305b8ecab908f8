// CanvasPartition CBS entry point (placeholder until the CBS kernels land; see DESIGN.md "CBS").
#include "common.hpp"
extern "C" int32_t canvas_cbs(canvas_ctx* ctx, int32_t nchr, const double* d_cov, const int64_t* h_chr_offset, double alpha, uint32_t nperm,
                              int32_t* d_seg_len, int32_t* h_nseg, int64_t* h_stats) {
    (void)nchr; (void)d_cov; (void)h_chr_offset; (void)alpha; (void)nperm; (void)d_seg_len; (void)h_nseg; (void)h_stats;
    if (!ctx) return CANVAS_ERR_INVALID;
    CANVAS_FAIL(ctx, CANVAS_ERR_UNSUPPORTED, "CBS segmentation (CBSRunner.cs) is not built yet; use PerSampleHMM");
}
