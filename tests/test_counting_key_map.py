"""The map the counting selects of CanvasClean rest on (canvas_amd/csrc/clean_fast.hpp: cq_key / cq_value), restated in numpy: k -> float32(float64(k) * 0.01) is
non-decreasing, gives back every integer count, and gives back the float a two-decimal text parses to except for a tiny fraction of the values (those samples take the
radix selects instead: the result never depends on it).  CPU only."""
import numpy as np


def cq_value(k):
    return (k.astype(np.float64) * 0.01).astype(np.float32)


def cq_key(x):
    y = x.astype(np.float64) * 100.0
    with np.errstate(invalid="ignore"):
        k = np.rint(np.where(np.isfinite(y), y, -1.0)).astype(np.int64)
    ok = (y >= -0.5) & (y < 1073741823.0) & (k >= 0) & (cq_value(k) == x) & ~((k == 0) & np.signbit(x))
    return k, ok


def test_map_is_monotone_and_reproduces_integer_counts():
    rng = np.random.RandomState(1)
    k = np.unique(np.concatenate([np.arange(0, 2_000_000), rng.randint(0, 1 << 30, 2_000_000), (1 << 30) - 1 - np.arange(1000)])).astype(np.int64)
    v = cq_value(k)
    assert (np.diff(v.astype(np.float64)) >= 0).all()                      # non-decreasing in k: order statistics of accepted keys are order statistics of k
    n = np.concatenate([np.arange(0, 3_000_000), rng.randint(0, (1 << 30) // 100, 1_000_000)]).astype(np.int64)
    x = n.astype(np.float32)                                               # integer read counts (every one below 2^24 is a float)
    exact = n < (1 << 24)
    kk, ok = cq_key(x[exact])
    assert ok.all() and (kk == 100 * n[exact]).all()


def test_accepted_keys_are_in_strictly_increasing_correspondence_with_k():
    rng = np.random.RandomState(2)
    x = np.concatenate([rng.gamma(30.0, 3.3, 3_000_000), rng.uniform(0, 5000, 1_000_000)])
    x = (np.round(x * 100.0) / 100.0).astype(np.float32)                   # what float.Parse makes of an F2 text: the float nearest to k / 100
    k, ok = cq_key(x)
    assert ok.mean() > 0.99999                                             # expected misses: ~7e-9 of the values (the double product within an ulp of a float rounding boundary)
    xs, ks = x[ok], k[ok]
    o = np.argsort(ks, kind="stable")
    xs, ks = xs[o], ks[o]
    same_k = np.diff(ks) == 0
    assert (np.diff(xs.astype(np.float64))[~same_k] > 0).all() and (np.diff(xs.astype(np.float64))[same_k] == 0).all()


def test_rejected_inputs():
    x = np.array([np.nan, -1.0, -0.0, 1.005, 123.456, 2.0e7, np.inf], np.float32)
    _, ok = cq_key(x)
    assert not ok.any()
    _, ok = cq_key(np.array([0.0, 0.01, 98.0, 12345.67], np.float32))
    assert ok.all()
