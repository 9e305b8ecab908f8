"""The in-memory "{count:F2}" hand-off (IO.cs:21 -> CanvasSegment.cs:1146) vs the oracle's formatter + strtod."""
import numpy as np
import pytest

import oracle_lib as O
from gpu_common import get_canvas, to_dev

pytestmark = pytest.mark.gpu


def test_quantize_f2_matches_text_roundtrip():
    cv = get_canvas()
    rng = np.random.RandomState(5)
    vals = np.concatenate([
        rng.uniform(0, 300, 20000), rng.uniform(0, 2, 3000), 10 ** rng.uniform(-4, 6.5, 5000),
        np.arange(0, 2000) / 32.0,                       # exact ties of the 7-digit stage
        np.arange(0, 4000) * 0.005, [0.0, 0.004999, 0.005, 0.00499999, 99.995, 99.99499, 1e-9, 123456.789, 9999999.0, 2.675, 100.125],
    ]).astype(np.float32)
    got = cv.quantize_f2(to_dev(vals, cv.device), len(vals)).cpu().numpy()
    exp = np.array([float(O.format_f2(float(v))) for v in vals])
    bad = np.nonzero(got != exp)[0]
    assert len(bad) == 0, [(vals[i], got[i], exp[i]) for i in bad[:10]]


def test_fast_path_equals_the_general_digit_arithmetic(monkeypatch):
    """quantize.hpp takes a 32-bit fast path for counts in [0.001, 1e7); CANVAS_F2_GENERAL=1 sends every value through the general arithmetic: bit-identical on
    random float bit patterns across the whole range (incl. the range borders, the exact ties of both rounding stages and negative values)."""
    cv = get_canvas()
    rng = np.random.RandomState(77)
    bits = rng.randint(0x3A000000, 0x4B800000, size=2_000_000, dtype=np.int64).astype(np.uint32)      # ~4.9e-4 .. 1.7e7: straddles both borders of the fast path
    vals = np.concatenate([bits.view(np.float32), -bits[:1000].view(np.float32), (np.arange(0, 400000) * 0.0025).astype(np.float32), (np.arange(0, 100000) / 64.0).astype(np.float32),
                           np.array([0.001, 0.00099999994, 9999999.0, 1.0e7, 9999999.5, 0.1, 0.01, 0.099999994, 0.0099999998, 1.0, 10.0, 100.0, 99.995, 0.995, 0.005], np.float32)])
    d = to_dev(vals, cv.device)
    fast = cv.quantize_f2(d, len(vals)).cpu().numpy()
    monkeypatch.setenv("CANVAS_F2_GENERAL", "1")
    general = cv.quantize_f2(d, len(vals)).cpu().numpy()
    bad = np.nonzero(fast.view(np.uint64) != general.view(np.uint64))[0]
    assert len(bad) == 0, [(vals[i], fast[i], general[i]) for i in bad[:10]]
    exp = np.array([float(O.format_f2(float(v))) for v in vals[:20000]])
    assert (fast[:20000] == exp).all()
