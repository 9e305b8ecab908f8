"""The two shortcuts of quantize_f2_fast (canvas_amd/csrc/quantize.hpp) against the divisions they replace, for every operand the path can see:
N2 / 100.0 as a multiplication by RN(1/100) + one fused correction, and x / 10^dd (x < 2^24) as a float estimate + a remainder correction.
Plain C with the same IEEE operations the device uses (fma, float multiply, truncating conversion); ~10 s."""
import shutil, subprocess
import pytest

SRC = r'''
#include <stdio.h>
#include <math.h>
#include <stdint.h>
int main(void) {
    uint64_t bad = 0, plain = 0, badDiv = 0, badId = 0;
    for (uint64_t N = 0; N < (1ull << 30); N++) {
        const double a = (double)N, q0 = a * 0.01, r0 = fma(-100.0, q0, a), q1 = fma(r0, 0.01, q0);
        if (q1 != a / 100.0) bad++;
        if (q0 != a / 100.0) plain++;
    }
    const float inv[8] = {1.0f, 1.0e-1f, 1.0e-2f, 1.0e-3f, 1.0e-4f, 1.0e-5f, 1.0e-6f, 1.0e-7f};
    uint32_t P = 1;
    for (int dd = 1; dd <= 7; dd++) {
        P *= 10u;
        for (uint32_t R7 = 0; R7 <= 10000000u; R7++) {
            const uint32_t x = R7 + (P >> 1);
            uint32_t q = (uint32_t)((float)x * inv[dd]);
            int32_t rem = (int32_t)(x - q * P);
            if (rem < 0) { q--; rem += (int32_t)P; }
            if (rem >= (int32_t)P) q++;
            if (q != x / P) badDiv++;
            if (q != (R7 / (P / 10u) + 5u) / 10u) badId++;
        }
    }
    printf("%llu %llu %llu %llu\n", (unsigned long long)bad, (unsigned long long)plain, (unsigned long long)badDiv, (unsigned long long)badId);
    return 0;
}
'''

@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_the_shortcuts_equal_the_divisions_for_every_operand(tmp_path):
    c = tmp_path / "d.c"; c.write_text(SRC)
    exe = tmp_path / "d"
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", str(exe), str(c), "-lm"], check=True)
    bad, plain, bad_div, bad_id = map(int, subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split())
    assert bad == 0 and bad_div == 0 and bad_id == 0
    assert plain > 0          # (the correction step is needed: the bare product differs from the quotient for about one N in seven)
