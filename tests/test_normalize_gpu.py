"""CanvasNormalize's ratio path on the GPU (canvas_normalize_reference / canvas_normalize_ratio) against the oracle: doubles and floats
compared as bit patterns."""
import numpy as np
import pytest

import oracle_lib as O
from gpu_common import get_canvas, to_dev

pytestmark = pytest.mark.gpu


def _panel(rng, n, nsamples):
    base = rng.gamma(2.0, 60.0, n)
    base[rng.rand(n) < 0.3] = 0.0                                  # off-target bins of an enrichment panel: mostly empty
    out = []
    for s in range(nsamples):
        c = np.round(rng.poisson(base * (0.5 + s)) * 1.0 + rng.randint(0, 3, n) * 0.25, 2)
        out.append(c.astype(np.float64))
    return out


@pytest.mark.parametrize("n,nsamples,with_manifest", [(50_000, 3, True), (1001, 2, False), (300_000, 5, True), (10, 2, False)])
def test_weighted_reference(n, nsamples, with_manifest):
    cv = get_canvas()
    rng = np.random.RandomState(n + nsamples)
    counts = _panel(rng, n, nsamples)
    on = np.sort(rng.choice(n, max(1, n // 3), replace=False)).astype(np.int32) if with_manifest else None
    exp, ew = O.norm_weighted_reference(counts, on)
    got, gw = cv.normalize_reference([to_dev(c, cv.device) for c in counts], None if on is None else to_dev(on, cv.device))
    assert (gw.view(np.uint64) == ew.view(np.uint64)).all()
    assert (got.cpu().numpy().view(np.uint64) == exp.view(np.uint64)).all()
    assert abs(gw.sum() - 1.0) < 1e-12


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("n,with_manifest,with_ploidy", [(80_000, True, True), (513, False, False), (256, True, False), (200_001, False, True)])
def test_ratio_and_counts(n, with_manifest, with_ploidy, mode):
    cv = get_canvas()
    rng = np.random.RandomState(7 * n + mode)
    sample = _panel(rng, n, 1)[0].astype(np.float32)
    ref = (O.norm_weighted_reference(_panel(rng, n, 3))[0]).astype(np.float32)      # what float.Parse leaves of the reference file
    ref[rng.rand(n) < 0.02] = 0.5                                                   # below the LSNorm cut
    on = np.sort(rng.choice(n, max(1, n // 4), replace=False)).astype(np.int32) if with_manifest else None
    ploidy = rng.choice([1, 2, 2, 2, 3], n).astype(np.int32) if with_ploidy else None
    kw = dict(mode=mode, min_ref=2.0 if mode else 1.0, max_ref=400.0 if mode else float("inf"))
    ek, er, ec = O.norm_ratio(sample, ref, on, ploidy=ploidy, **kw)
    gk, gr, gc, lsf = cv.normalize_ratio(to_dev(sample, cv.device), to_dev(ref, cv.device), None if on is None else to_dev(on, cv.device),
                                         ploidy=None if ploidy is None else to_dev(ploidy, cv.device), **kw)
    assert len(gk) == len(ek) and 0 < len(ek) < n
    assert (gk.cpu().numpy() == ek).all()
    assert (gr.cpu().numpy().view(np.uint32) == er.view(np.uint32)).all()
    assert (gc.cpu().numpy().view(np.uint32) == ec.view(np.uint32)).all()
    if mode == 1:
        assert lsf == 1.0
