"""BASELINE configs[4] (tumour 80x / normal 40x: joint normalisation + CBS) as ONE flow, every hand-off compared with the chained oracle:
tumour GCContentWeighted bins -> normal bins on the tumour's bin size -> LSNorm ratio x 40 -> "{count:F2}" file -> CanvasClean -> F2 -> CBS
(segments and RNG consumption).  Sizes the oracle finishes in seconds; bench.py runs the same flow at whole-genome size (`somatic_flow`)."""
import numpy as np
import pytest

import oracle_flows as OF
import oracle_lib as O
from canvas_amd import synth, CLEAN_GCNORM, CLEAN_FILTSIZE, CLEAN_OUTLIERS, CLEAN_LOCALSD
from gpu_common import get_canvas, to_dev, pad16

pytestmark = pytest.mark.gpu
SEED = 20260927 + 5                       # seeds = 20260927 + config#
ALL = CLEAN_GCNORM | CLEAN_FILTSIZE | CLEAN_OUTLIERS | CLEAN_LOCALSD


def _pair(lengths, rate_t=0.28, rate_n=0.14, purity=0.7):
    thr_t = synth.poisson_thresholds(rate_t, purity=purity); thr_n = synth.poisson_thresholds(rate_n, flat=True)
    T = [synth.generate_chromosome(SEED, c, L, rate_t, thr_t, hit_seed=SEED + 1000, with_fraglen=True) for c, L in enumerate(lengths)]
    N = [synth.generate_chromosome(SEED, c, L, rate_n, thr_n, hit_seed=SEED + 2000) for c, L in enumerate(lengths)]
    return T, N


@pytest.mark.parametrize("lengths,flags", [([9_000_000, 7_500_000, 6_100_003, 5_000_000, 4_200_000, 3_000_000], ALL), ([2_500_000, 1_300_001], CLEAN_GCNORM | CLEAN_FILTSIZE | CLEAN_OUTLIERS)])
def test_tumor_normal_flow_matches_chained_oracle(lengths, flags):
    cv = get_canvas()
    nchr = len(lengths)
    T, N = _pair(lengths)
    for t, n in zip(T, N):
        assert (t[0] == n[0]).all() and (t[2] == n[2]).all()            # one reference, one mask
    is_auto = np.ones(nchr, np.uint8); is_auto[-1] = 0
    bases = [to_dev(pad16(t[0]), cv.device) for t in T]; masks = [to_dev(t[2].view(np.int64), cv.device) for t in T]
    hits_t = [to_dev(pad16(t[1]), cv.device) for t in T]; fl = [to_dev(pad16(t[3]), cv.device) for t in T]; hits_n = [to_dev(pad16(n[1]), cv.device) for n in N]
    lens = np.array(lengths, np.int64)
    got = cv.tumor_normal_flow(bases, masks, hits_t, fl, hits_n, lens, is_auto, flags, alpha=0.01, nperm=10000, keep=True)
    exp = OF.tumor_normal([t[0] for t in T], [t[2] for t in T], [t[1] for t in T], [t[3] for t in T], [n[1] for n in N], is_auto, flags, 0.01, 10000)
    h = lambda t: t.cpu().numpy()
    # 1. CanvasBin, tumour (GCContentWeighted) and normal (TruncatedDynamicRange on the tumour's bin size)
    assert got["bin_size"] == exp["bin_size"]
    for k in ("chr", "start", "stop", "gc"):
        assert (h(got["tumour"][k]) == exp["tumour"][k]).all(), k
    assert (h(got["tumour"]["count"]) == exp["tumour"]["count"]).all()
    assert (h(got["normal_count"]) == exp["normal_count"]).all()
    # 2. CanvasNormalize: kept bins, ratios and ratio x 40 counts as bit patterns
    assert (h(got["keep_idx"]) == exp["keep_idx"]).all()
    assert (h(got["ratio"]).view(np.uint32) == exp["ratio"].view(np.uint32)).all()
    assert (h(got["ratio_count"]).view(np.uint32) == exp["ratio_count"].view(np.uint32)).all()
    assert 0 < got["n_ratio"] <= got["n_bins"]
    # 3. the F2 file CanvasClean reads
    assert (h(got["to_clean"]["count"]).view(np.uint32) == exp["to_clean"]["count"].view(np.uint32)).all()
    # 4. CanvasClean
    assert got["n_clean"] == len(exp["cleaned"]["chr"])
    for k in ("chr", "start", "stop", "gc"):
        assert (h(got["cleaned"][k]) == exp["cleaned"][k]).all(), k
    assert (h(got["cleaned"]["count"]).view(np.uint32) == exp["cleaned"]["count"].view(np.uint32)).all()
    assert got["local_sd"] == exp["cleaned"]["local_sd"]
    # 5. the F2 column CanvasPartition parses, then CBS: segment lengths AND the random numbers consumed
    assert (h(got["cov"]) == exp["cov"]).all()
    assert (got["chr_offset"] == exp["chr_offset"]).all()
    sl = h(got["seg_len"])
    for c in range(nchr):
        g = sl[got["chr_offset"][c]:got["chr_offset"][c] + got["nseg"][c]]
        assert got["nseg"][c] == len(exp["seg_len"][c]) and (g == exp["seg_len"][c]).all(), c
    st, es = got["cbs_stats"], exp["cbs_stats"]
    assert st[0] == es[0] and st[2] == es[2] and st[4] == es[4]         # TMaxO calls, permutations, TPermP draws
    if len(lengths) > 2:
        assert got["segments"] > nchr and es[2] > 0                     # the tumour's CN segments were found, permutations were drawn
        med = np.median(exp["cov"])
        assert 30.0 < med < 50.0                                        # a diploid bin sits at ratio 1 x 40
