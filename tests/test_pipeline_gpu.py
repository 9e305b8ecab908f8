"""canvas_sample_pipeline (one call for bin -> clean -> F2 -> PerSampleHMM -> segment ids) must leave exactly what the six staged calls leave."""
import numpy as np
import pytest

from canvas_amd import synth, CLEAN_GCNORM, CLEAN_FILTSIZE, CLEAN_OUTLIERS, CLEAN_LOCALSD
from gpu_common import get_canvas, to_dev, pad16

pytestmark = pytest.mark.gpu


def test_one_call_pipeline_equals_staged_calls():
    import torch
    cv = get_canvas()
    lengths = [3_000_000, 2_200_000, 1_500_000]
    is_auto = np.array([1, 1, 0], np.uint8)
    thr = synth.poisson_thresholds(0.21)
    data = [synth.generate_chromosome(20260927 + 70, c, L, 0.21, thr) for c, L in enumerate(lengths)]
    bases = [to_dev(pad16(b), cv.device) for b, h, m in data]; hits = [to_dev(pad16(h), cv.device) for b, h, m in data]
    masks = [to_dev(m.view(np.int64), cv.device) for b, h, m in data]
    lens = np.array(lengths, np.int64)
    flags = CLEAN_GCNORM | CLEAN_FILTSIZE | CLEAN_OUTLIERS | CLEAN_LOCALSD
    cap = int(lens.sum() // 100) + 16

    def bufs():
        out = dict(chr=torch.empty(cap, dtype=torch.int32, device=cv.device), start=torch.empty(cap, dtype=torch.int32, device=cv.device),
                   stop=torch.empty(cap, dtype=torch.int32, device=cv.device), gc=torch.empty(cap, dtype=torch.int32, device=cv.device),
                   count=torch.empty(cap, dtype=torch.float32, device=cv.device))
        return out, torch.empty(cap, dtype=torch.float64, device=cv.device), torch.empty(cap, dtype=torch.int32, device=cv.device), torch.empty(cap, dtype=torch.int32, device=cv.device)

    out1, cov1, st1, seg1 = bufs()
    o, per, total, bs = cv.bin_sample(bases, masks, hits, lens, is_auto, 100, -1, 3, out=out1)
    n_out, lsd, info = cv.clean(out1, total, is_auto, flags)
    cov = cv.quantize_f2(out1["count"], n_out, out=cov1)
    off = cv.chromosome_offsets(out1["chr"], n_out, 3)
    state = cv.hmm_per_sample(cov, off, out=st1)
    seg, nseg = cv.segment_ids(off, state, out1["start"], out1["stop"], out=seg1)
    out2, cov2, st2, seg2 = bufs()
    r = cv.sample_pipeline(bases, masks, hits, lens, is_auto, out2, cov2, st2, seg2, counts_per_bin=100, bin_size=-1, mode=3, flags=flags)
    cv.synchronize()
    assert (r["bin_size"], r["total"], r["n_out"], r["nseg"]) == (bs, total, n_out, nseg) and r["off"].tolist() == list(off)
    assert np.float64(r["lsd"]).view(np.uint64) == np.float64(lsd).view(np.uint64)
    for k in out1:
        assert torch.equal(out1[k][:n_out], out2[k][:n_out]), k
    assert torch.equal(cov1[:n_out], cov2[:n_out]) and torch.equal(st1[:n_out], st2[:n_out]) and torch.equal(seg1[:n_out], seg2[:n_out])
    assert n_out > 5_000 and nseg >= 3
