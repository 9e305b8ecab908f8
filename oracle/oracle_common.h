// TEST INFRASTRUCTURE ONLY — CPU restatement ("oracle") of the Canvas read-depth hot path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
// The product (canvas_amd/, libcanvas_hip.so) never includes, links or calls this code.
//
// Parity status: the C# reference cannot be built in this image (no dotnet, private NuGet packages), so this
// restatement is pinned only by the reference's own known-answer tests (tests/golden/*, see tests/test_oracle_golden.py).
// Everything the reference's tests do not cover (BinCountsForChromosome, CanvasClean stages, CBS, Viterbi) is
// "parity unpinned" against the real binaries: it follows the cited C# statement by statement.  Those stages are
// cross-checked against a second reading of the C#, plain Python that shares no code with this directory
// (tests/test_oracle_independent.py); that excludes a slip made once, it is not a reference-generated vector.
//
// .NET semantics restated here:
//   (int)x            -> truncation toward zero
//   Convert.ToInt32(d), Math.Round(d) -> round half to even
//   float expressions -> evaluated in IEEE binary32 (RyuJIT x64 SSE), no FMA contraction (-ffp-contract=off)
//   LINQ Sum<float>   -> double accumulator, cast to float
//   SortedList<T>.Median() (Illumina.Common, not in /root/reference) -> sorted; odd: a[n/2]; even: (a[n/2-1]+a[n/2])/2 in T
#pragma once
#include <cstdint>
#include "../include/canvas_mathnet.h"
#include <cstdio>
#include <cstring>
#include <cmath>
#include <cfenv>
#include <vector>
#include <string>
#include <algorithm>
#include <numeric>
#include <limits>

namespace oracle {

// Convert.ToInt32(double) / Math.Round(double): banker's rounding.
static inline double round_half_even(double x) { return std::nearbyint(x); }  // default FE_TONEAREST
static inline int to_int32_round(double x) { return (int)std::nearbyint(x); }

// SortedList<T>.Median()
template <class T>
static inline T sorted_median(std::vector<T>& v) {
    std::sort(v.begin(), v.end());
    size_t n = v.size();
    if (n == 0) return T(0);
    if (n % 2 == 1) return v[n / 2];
    return (T)((v[n / 2 - 1] + v[n / 2]) / (T)2);
}
template <class T>
static inline T median_copy(const std::vector<T>& v) { std::vector<T> c(v); return sorted_median(c); }

// ---- MT19937 as MathNet.Numerics.Random.MersenneTwister 3.17 is assumed to behave (not in /root/reference;
// parity unpinned): init_genrand((uint)seed); NextDouble() = genrand_int32() * 2^-32;
// NextFullRangeInt32() = BitConverter.ToInt32 of 4 bytes, each byte = (byte)(genrand_int32() % 256) by default — variants 1 / 2: include/canvas_mathnet.h.
struct MT19937 {
    uint32_t mt[624];
    int mti;
    explicit MT19937(uint32_t seed = 5489u) { init(seed); }
    void init(uint32_t s) {
        mt[0] = s;
        for (mti = 1; mti < 624; mti++) mt[mti] = 1812433253u * (mt[mti - 1] ^ (mt[mti - 1] >> 30)) + (uint32_t)mti;
    }
    uint32_t next_u32() {
        static const uint32_t mag01[2] = {0u, 0x9908b0dfu};
        uint32_t y;
        if (mti >= 624) {
            int kk;
            for (kk = 0; kk < 624 - 397; kk++) {
                y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
                mt[kk] = mt[kk + 397] ^ (y >> 1) ^ mag01[y & 1u];
            }
            for (; kk < 623; kk++) {
                y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
                mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ mag01[y & 1u];
            }
            y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
            mt[623] = mt[396] ^ (y >> 1) ^ mag01[y & 1u];
            mti = 0;
        }
        y = mt[mti++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    double next_double() { return next_u32() * (1.0 / 4294967296.0); }
    int32_t next_full_range_int32() {      // which eight bits make a byte: include/canvas_mathnet.h (the one switch the product and this oracle share)
        uint32_t g[4];
        for (int b = 0; b < 4; b++) g[b] = next_u32();
        return canvas_mathnet_full_range_int32(g, canvas_mathnet_seed_variant());
    }
};

// ---- .NET Core 2.x number formatting.
// float.ToString("F2"): value -> 7 significant decimal digits (correctly rounded), then half-away-from-zero
// on the digit string at 2 decimals.
std::string format_float_f2(float v);
// double.ToString() == "G15".
std::string format_double_g15(double v);
// float.ToString() == "G7".
std::string format_float_g7(float v);

}  // namespace oracle
