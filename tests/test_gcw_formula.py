"""k_read_gc3's default-window value (bin_gcw.hpp, step (a)): floor(100 * count / meanFragment) from one float multiply-add per position.

The kernel computes  v = (uint32) fma(d, c100, bf)  with  c100 = float(100 / M),  bf = (float(100 * cnt0) - 799.5f) * (1.0f / M)  and d = 8 + (count - cnt0) in
[1, 15]; the reference (CanvasBin.cs:476-480) computes 100 * gcCounter / meanFragmentSize in integers.  This test walks EVERY value boundary of every mean fragment size the
float path is used for (101 .. RG3_FAST_M): the smallest count of each value v and the count below it, reached through each of the fifteen offsets d.  float32 fma is
emulated in float64 (the product of two float32 is exact in float64; the sum's rounding is 2^-29 of a float32 ulp).
"""
import numpy as np

RG3_FAST_M = 16384      # bin_gcw.hpp


def _kernel_value(M, cnt0, d):
    """the kernel's arithmetic, element-wise (M, cnt0, d: integer arrays)"""
    Mf = M.astype(np.float32)
    r_mean = np.float32(1.0) / Mf                                  # IEEE division, correctly rounded
    c100 = (100.0 / M.astype(np.float64)).astype(np.float32)
    bf = ((100 * cnt0).astype(np.float32) - np.float32(799.5)) * r_mean
    t = (d.astype(np.float64) * c100.astype(np.float64) + bf.astype(np.float64)).astype(np.float32)
    return t.astype(np.int64)                                      # truncation (t > 0)


def test_every_boundary_of_the_float_path():
    d = np.arange(1, 16, dtype=np.int64)[None, None, None, :]
    v = np.arange(1, 101, dtype=np.int64)[None, :, None, None]
    below = np.array([0, 1], dtype=np.int64)[None, None, :, None]
    bad = 0
    for lo in range(101, RG3_FAST_M + 1, 512):
        M = np.arange(lo, min(lo + 512, RG3_FAST_M + 1), dtype=np.int64)[:, None, None, None]
        c = (v * M + 99) // 100 - below                            # smallest count with 100 c >= v M, and the one below it
        cnt0 = c - (d - 8)
        ok = (cnt0 >= 0) & (cnt0 <= M) & (c >= 0) & (c <= M)
        Mb, cb, c0b, db = np.broadcast_arrays(M, c, cnt0, d)
        got = _kernel_value(Mb[ok], c0b[ok], db[ok])
        want = (100 * cb[ok]) // Mb[ok]
        bad += int((got != want).sum())
    assert bad == 0


def test_random_counts_of_the_float_path():
    rng = np.random.default_rng(5)
    M = rng.integers(101, RG3_FAST_M + 1, size=2_000_000)
    cnt0 = (rng.random(M.size) * (M + 1)).astype(np.int64)
    d = rng.integers(1, 16, size=M.size)
    c = cnt0 + d - 8
    ok = (c >= 0) & (c <= M)
    assert np.array_equal(_kernel_value(M[ok], cnt0[ok], d[ok]), (100 * c[ok]) // M[ok])


def test_eight_counts_from_one_multiplication():
    """nibble j of ((spread(B) - spread(A)) * 0x11111111 + 0x88888888) << 4 | 8  =  8 + sum_{i<j} (B_i - A_i)"""
    def spread8(x):
        s = (x | (x << 12)) & 0x000F000F
        s = (s | (s << 6)) & 0x03030303
        return (s | (s << 3)) & 0x11111111
    A, B = np.meshgrid(np.arange(256, dtype=np.int64), np.arange(256, dtype=np.int64))
    dex = (((((spread8(B) - spread8(A)) * 0x11111111 + 0x88888888) & 0xFFFFFFFF) << 4) | 8) & 0xFFFFFFFF
    run = np.zeros_like(A)
    for j in range(8):
        assert np.array_equal((dex >> (4 * j)) & 15, 8 + run)
        run = run + ((B >> j) & 1) - ((A >> j) & 1)
