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
