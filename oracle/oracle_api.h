// TEST INFRASTRUCTURE ONLY (see oracle_common.h). Internal + extern "C" declarations of the CPU oracle.
#pragma once
#include <cstdint>
#include <vector>

enum : uint32_t {  // same bit meanings as include/canvas_hip.h CANVAS_CLEAN_*
    CLEAN_GCNORM = 1u, CLEAN_FILTSIZE = 2u, CLEAN_OUTLIERS = 4u, CLEAN_LOCALSD = 8u, CLEAN_LOESS = 16u
};

namespace oracle {
double bin_rate(const uint8_t* hits, const uint8_t* mask, int64_t len);
int bin_size_from_rates(const double* rates, int n, int countsPerBin);
int64_t bin_chromosome(const uint8_t* bases, const uint8_t* mask, const uint8_t* hits, int64_t len, int binSize, int mode,
                       int64_t cap, int32_t* start, int32_t* stop, int32_t* gc, int32_t* count);
int64_t bin_chromosome_predefined(const uint8_t* bases, const uint8_t* mask, const uint8_t* hits, int64_t len, int mode, int64_t nbins, const int32_t* binStart, const int32_t* binStop,
                                  int32_t* gc, int32_t* count);
int64_t bin_chromosome_predefined_weighted(const uint8_t* bases, const uint8_t* mask, const uint8_t* hits, const uint8_t* readGC, const float* obsVsExp, int64_t len, int64_t nbins,
                                           const int32_t* binStart, const int32_t* binStop, int32_t* gc, int32_t* count);
int16_t mean_fragment_size(int nchr, const int16_t* const* fl, const int64_t* len);
void read_gc_content(const uint8_t* bases, const int16_t* fl, int64_t L, int meanFragmentSize, uint8_t* gcContent);
void observed_vs_expected_gc(int nchr, const uint8_t* const* readGC, const uint8_t* const* hits, const int64_t* len, float* out101);
int64_t bin_chromosome_weighted(const uint8_t* bases, const uint8_t* mask, const uint8_t* hits, const uint8_t* readGC, const float* obsVsExp, int64_t len, int binSize,
                                int64_t cap, int32_t* start, int32_t* stop, int32_t* gc, int32_t* count);
int64_t clean(int64_t n, int32_t* chr, int32_t* start, int32_t* stop, float* count, int32_t* gc, int nchr,
              const uint8_t* chrIsAutosome, const uint8_t* chrIsY, uint32_t flags, int minBinsWeighted, double* localSdOut,
              int32_t* stageCounts);
void Quartiles(const std::vector<float>& x, float& q1, float& q2, float& q3);
void loess_fit(const double* x, const double* y, int n, double bandwidth, int robIters, double xStep, double* fittedOrig, double* predicted);
double golden_section_square(double a, double b);
}  // namespace oracle
