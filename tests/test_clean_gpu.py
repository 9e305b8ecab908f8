"""CanvasClean on the GPU vs the CPU oracle: surviving bins identical, counts bit-identical (MedianByGC mode)."""
import numpy as np
import pytest

import oracle_lib as O
from canvas_amd import synth, CLEAN_GCNORM, CLEAN_FILTSIZE, CLEAN_OUTLIERS, CLEAN_LOCALSD
from gpu_common import get_canvas, to_dev

pytestmark = pytest.mark.gpu
ALL = CLEAN_GCNORM | CLEAN_FILTSIZE | CLEAN_OUTLIERS | CLEAN_LOCALSD


def _run(cv, bins, flags, nchr=24):
    is_auto = synth.IS_AUTOSOME[:nchr]
    is_y = np.zeros(nchr, np.uint8); is_y[-1] = 1
    exp = O.clean(bins["chr"], bins["start"], bins["stop"], bins["count"], bins["gc"], is_auto, is_y, flags)
    dev = {k: to_dev(v, cv.device) for k, v in bins.items()}
    n_out, lsd, info = cv.clean(dev, len(bins["chr"]), is_auto, flags)
    assert n_out == len(exp["chr"]), (n_out, len(exp["chr"]), info, exp["stages"])
    for k in ("chr", "start", "stop", "gc"):
        assert (dev[k][:n_out].cpu().numpy() == exp[k]).all(), k
    got = dev["count"][:n_out].cpu().numpy()
    assert (got.view(np.uint32) == exp["count"].view(np.uint32)).all(), np.abs(got - exp["count"]).max()
    assert lsd == exp["local_sd"]
    return info, exp


@pytest.mark.parametrize("n,flags", [(60_000, ALL), (60_000, CLEAN_GCNORM), (20_000, ALL), (3_000, CLEAN_FILTSIZE | CLEAN_OUTLIERS),
                                     (120_000, CLEAN_GCNORM | CLEAN_LOCALSD), (60_001, CLEAN_OUTLIERS)])
def test_clean_matches_oracle(n, flags):
    cv = get_canvas()
    bins = synth.generate_bins(20260927 + 2, n)
    _run(cv, bins, flags)


def test_clean_wgs_variance_normalisation_path():
    cv = get_canvas()
    bins = synth.generate_bins(20260927 + 3, 700_000)
    # make GC buckets 50..58 very noisy so that NormalizeVarianceByGC (CanvasClean.cs:34-97) fires
    rng = np.random.RandomState(7)
    noisy = (bins["gc"] >= 50) & (bins["gc"] <= 58)
    bins["count"] = np.where(noisy, np.maximum(0, bins["count"] + rng.normal(0, 60, len(noisy))), bins["count"]).astype(np.float32)
    info, exp = _run(cv, bins, CLEAN_GCNORM | CLEAN_FILTSIZE | CLEAN_LOCALSD)   # the outlier filter would remove the noisy bins
    assert info[4] == 1 and exp["stages"][4] == 1
    info, exp = _run(cv, bins, ALL)


def test_clean_ffpe_like_sample_drops_noisy_windows():
    cv = get_canvas()
    bins = synth.generate_bins(20260927 + 4, 80_000)
    rng = np.random.RandomState(9)
    # the metric is the per-chromosome MAD of the window SDs: make the windows heterogeneous (alternating quiet / noisy
    # 500-bin blocks) so that it exceeds 5, and one stretch with SD > 40 that RemoveBinsWithExtremeLocalSD must drop
    n = len(bins["count"])
    block = (np.arange(n) // 500) % 2 == 1
    bins["count"] = (bins["count"] + np.where(block, rng.normal(0, 35, n), 0)).clip(0).astype(np.float32)
    bins["count"][30_000:31_000] = (bins["count"][30_000:31_000] + rng.normal(0, 90, 1000)).clip(0).astype(np.float32)
    info, exp = _run(cv, bins, CLEAN_GCNORM | CLEAN_FILTSIZE | CLEAN_LOCALSD)
    assert exp["local_sd"] > 5.0
    assert exp["stages"][3] < exp["stages"][2]


def test_clean_unsorted_chromosome_runs_and_tiny_inputs():
    cv = get_canvas()
    bins = synth.generate_bins(20260927 + 5, 52_000, nchr=6)
    # chromosome 2 appears twice (non-contiguous runs), a chromosome with a single isolated bin
    bins["chr"][100:140] = 2
    bins["chr"][5000] = 5
    _run(cv, bins, ALL, nchr=6)
    for n in (1, 2, 25):
        b = {k: v[:n].copy() for k, v in bins.items()}
        _run(cv, b, CLEAN_FILTSIZE | CLEAN_OUTLIERS, nchr=6)


def test_clean_loess_mode_within_tolerance():
    """-m LOESS (LoessGCNormalizer): sufficient-statistics LOESS on the GPU vs the oracle's point-by-point LOESS; the reference tolerance
    for this mode is 1e-5 relative (sums are re-associated), survivors must be identical"""
    from canvas_amd import CLEAN_LOESS
    cv = get_canvas()
    nchr = 24
    bins = synth.generate_bins(20260927 + 6, 30_000)
    bins["gc"] = np.clip(bins["gc"], 12, 80)        # GC = 0 makes the reference itself index out of range
    is_auto = synth.IS_AUTOSOME; is_y = np.zeros(nchr, np.uint8); is_y[-1] = 1
    flags = CLEAN_GCNORM | CLEAN_FILTSIZE | CLEAN_OUTLIERS | CLEAN_LOESS
    exp = O.clean(bins["chr"], bins["start"], bins["stop"], bins["count"], bins["gc"], is_auto, is_y, flags)
    dev = {k: to_dev(v, cv.device) for k, v in bins.items()}
    n_out, lsd, info = cv.clean(dev, len(bins["chr"]), is_auto, flags, is_y=is_y)
    assert n_out == len(exp["chr"])
    for k in ("chr", "start", "stop", "gc"):
        assert (dev[k][:n_out].cpu().numpy() == exp[k]).all()
    got = dev["count"][:n_out].cpu().numpy().astype(np.float64); ex = exp["count"].astype(np.float64)
    rel = np.abs(got - ex) / np.maximum(np.abs(ex), 1e-12)
    assert rel[ex > 0].max() < 1e-5, rel.max()
    assert ((ex == 0) == (got == 0)).all()
