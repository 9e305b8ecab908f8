// "{0:F2}" of a float count, parsed back as a double: the hand-off between CanvasClean and CanvasPartition (CanvasCommon/IO.cs:21 -> CanvasSegment.cs:1146), in memory.
// Shared by clean.hip (canvas_quantize_f2) and hmm.hip (the quantisation fused with the counting of the genome-wide quartiles, pipeline.hip).
#pragma once
// ---------------------------------------------------------------- "{count:F2}" text hand-off in memory (IO.cs:21 -> CanvasSegment.cs:1146)
// Exact integer arithmetic: float = m * 2^e; 7 significant decimal digits (ties-to-even on the exact value, as the oracle's
// correctly rounded printf), then half-up on the decimal digits at 2 decimals, then N/100 as a correctly rounded double
// (= parsing the printed text).
// The counts of the path lie in [0.001, 1e7): there the seven significant digits fit 32 bits and the whole conversion is a 24 x 30-bit product, one shift with
// round-half-even, one 32-bit division by a power of ten and one by ten (the general code below: 64-bit divisions and loops, ~2000 cycles per wave and element — it made
// k_quant_covq VALU-bound at 58 us per WGS sample).  Same arithmetic, same results (tests/test_quantize_gpu.py compares the two on random bit patterns).
__device__ __forceinline__ bool quantize_f2_fast(float v, float af, double& out, long long* kOut) {
    if (!(af >= 0.001f && af < 1.0e7f)) return false;
    const uint32_t bits = __float_as_uint(af);
    const uint32_t m = (bits & 0x7FFFFFu) | 0x800000u;          // normal: af >= 0.001
    const int s = 150 - (int)(bits >> 23);                      // af = m * 2^-s, 0 <= s <= 33
    int d;                                                      // decimals kept by the 7-significant-digit stage = 7 - number of integer digits
    if (af >= 1.0f) d = 6 - (af >= 10.0f) - (af >= 100.0f) - (af >= 1000.0f) - (af >= 10000.0f) - (af >= 100000.0f) - (af >= 1000000.0f);
    else { const double a = (double)af; d = a >= 0.1 ? 7 : (a >= 0.01 ? 8 : 9); }
    uint32_t p = 1u;                                            // 10^d
    p = d >= 1 ? 10u : p; p = d >= 2 ? 100u : p; p = d >= 3 ? 1000u : p; p = d >= 4 ? 10000u : p; p = d >= 5 ? 100000u : p;
    p = d >= 6 ? 1000000u : p; p = d >= 7 ? 10000000u : p; p = d >= 8 ? 100000000u : p; p = d >= 9 ? 1000000000u : p;
    const unsigned long long num = (unsigned long long)m * (unsigned long long)p;      // < 2^24 * 10^9 < 2^54
    uint32_t R7;                                                // round-half-even of num / 2^s: in [10^6, 10^7]
    if (s == 0) R7 = (uint32_t)num;
    else {
        unsigned long long q = num >> s; const unsigned long long rem = num & ((1ull << s) - 1ull), half = 1ull << (s - 1);
        if (rem > half || (rem == half && (q & 1ull))) q++;
        R7 = (uint32_t)q;
    }
    const int dd = d - 2;                                       // digits dropped by the two-decimal stage (half-up on the decimal digits)
    uint32_t N2;
    if (dd <= 0) N2 = R7 * (dd == 0 ? 1u : (dd == -1 ? 10u : 100u));
    else {
        // (R7 / 10^(dd-1) + 5) / 10 = (R7 + 5 * 10^(dd-1)) / 10^dd: ONE division of a number below 2^24 by a power of ten — a float estimate (exact operand, reciprocal
        // and product good to 2^-22: off by one at most) and a correction on the exact remainder, instead of two 32-bit divisions (~40 instructions each)
        uint32_t P = 10u; float inv = 1.0e-1f;                  // 10^dd, dd = 1..7
        P = dd >= 2 ? 100u : P; inv = dd >= 2 ? 1.0e-2f : inv; P = dd >= 3 ? 1000u : P; inv = dd >= 3 ? 1.0e-3f : inv; P = dd >= 4 ? 10000u : P; inv = dd >= 4 ? 1.0e-4f : inv;
        P = dd >= 5 ? 100000u : P; inv = dd >= 5 ? 1.0e-5f : inv; P = dd >= 6 ? 1000000u : P; inv = dd >= 6 ? 1.0e-6f : inv; P = dd >= 7 ? 10000000u : P; inv = dd >= 7 ? 1.0e-7f : inv;
        const uint32_t x = R7 + (P >> 1);                       // < 1.5e7 < 2^24: exact as a float
        uint32_t q = (uint32_t)((float)x * inv);
        int32_t rem = (int32_t)(x - q * P);
        if (rem < 0) { q--; rem += (int32_t)P; }
        if (rem >= (int32_t)P) q++;
        N2 = q;
    }
    // N2 / 100.0 correctly rounded without the division sequence: q0 = N2 * RN(1/100) is within an ulp, the fused residual is exact, one correction step rounds correctly
    // (Markstein); checked against the division for EVERY N2 below 2^30 (tests/test_quantize_division.py restates the check; the fast path only sees N2 < 10^9)
    const double a2 = (double)N2, q0 = a2 * 0.01, r0 = fma(-100.0, q0, a2);
    const double r = fma(r0, 0.01, q0);
    if (kOut && !(v < 0) && N2 < (1u << 30)) *kOut = (long long)N2;
    out = v < 0 ? -r : r;
    return true;
}
__device__ inline double quantize_f2_one(float v, long long* kOut = nullptr, bool generalOnly = false) {
    if (kOut) *kOut = -1;                                       // the integer N with result = N / 100, when the value went through the digit arithmetic and is not negative
    const float af = fabsf(v);
    { double fast; if (!generalOnly && quantize_f2_fast(v, af, fast, kOut)) return fast; }
    if (af != af) return (double)v;                             // NaN passes through
    if (af < 0.001f) { if (kOut) *kOut = 0; return 0.0; }       // prints 0.00
    if (af >= 1.0e15f) return (double)v;                                                  // outside the supported count range
    const uint32_t bits = __float_as_uint(af);
    const int ex = (int)(bits >> 23);
    unsigned long long m = ex ? ((bits & 0x7FFFFFu) | 0x800000u) : (bits & 0x7FFFFFu);
    const int e = (ex ? ex : 1) - 150;                          // af = m * 2^e exactly
    const double a = (double)af;
    const double p10[17] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16};
    const unsigned long long ip10[17] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull, 100000000ull, 1000000000ull,
                                         10000000000ull, 100000000000ull, 1000000000000ull, 10000000000000ull, 100000000000000ull, 1000000000000000ull, 10000000000000000ull};
    int scale;                                                  // number of integer digits (<= 0 below 1)
    if (a >= 1.0) { scale = 1; while (scale < 16 && a >= p10[scale]) scale++; }
    else if (a >= 0.1) scale = 0; else if (a >= 0.01) scale = -1; else scale = -2;
    const int d = 7 - scale;                                    // decimals kept by the 7-significant-digit stage
    unsigned long long R7;
    if (d >= 0) {
        unsigned long long num = m * ip10[d];                   // < 2^24 * 10^9 < 2^54
        if (e >= 0) R7 = num << e;                              // only when d == 0..: af >= 2^23, fits
        else {
            int s = -e;                                         // <= 34 for af >= 0.001
            unsigned long long q = num >> s, rem = num & ((1ull << s) - 1ull), half = 1ull << (s - 1);
            if (rem > half || (rem == half && (q & 1ull))) q++;
            R7 = q;
        }
    } else {
        unsigned long long A = e >= 0 ? (m << e) : (m >> (-e));  // af >= 1e7 > 2^23: integer valued, e >= 0 except the first binade
        unsigned long long P = ip10[-d], q = A / P, rem = A % P;
        if (rem * 2 > P || (rem * 2 == P && (q & 1ull))) q++;
        R7 = q;
    }
    unsigned long long N2;
    const int dd = d - 2;
    if (dd <= 0) N2 = R7 * ip10[-dd];
    else if (dd > 16) N2 = 0;
    else {
        unsigned long long P = ip10[dd];
        N2 = R7 / P + (((R7 / ip10[dd - 1]) % 10ull) >= 5ull ? 1ull : 0ull);
    }
    double r = (double)N2 / 100.0;
    if (kOut && !(v < 0) && N2 < (1ull << 30)) *kOut = (long long)N2;
    return v < 0 ? -r : r;
}
