"""BASELINE configs[0] on the GPU: the chr20-size chain of tests/test_chr20_chain.py (L = 64 444 167, 30x) through the HIP path — one-call pipeline
(bin -> clean -> F2 -> PerSampleHMM -> segment ids) and CBS — against the oracle chain on the same bytes, file rows compared through their digests too."""
import numpy as np
import pytest

import oracle_lib as O
import test_chr20_chain as C20
from canvas_amd import CLEAN_GCNORM, CLEAN_FILTSIZE, CLEAN_OUTLIERS, CLEAN_LOCALSD
from gpu_common import get_canvas, to_dev, pad16

pytestmark = pytest.mark.gpu


def test_chr20_hip_path_matches_oracle_chain():
    import torch
    from canvas_amd.lib import synth_generate_device
    cv = get_canvas()
    L = C20.L_CHR20
    db, dh, dm, _ = synth_generate_device(C20.SEED, 19, L, C20.RATE, cv.device)
    torch.cuda.synchronize()
    b = db[:L].cpu().numpy(); h = dh[:L].cpu().numpy(); m = dm.cpu().numpy().view(np.uint8)
    import oracle_flows as OF
    exp = OF.germline_single([b], [m], [h], np.array([1], np.uint8), ["chr20"])
    assert C20.digest(exp["binned_rows"]) == C20.EXPECTED["binned"]          # the device generator made the bytes the CPU test pins
    cap = L // 100 + 16
    mk = lambda dt: torch.empty(cap, dtype=dt, device=cv.device)
    out = dict(chr=mk(torch.int32), start=mk(torch.int32), stop=mk(torch.int32), gc=mk(torch.int32), count=mk(torch.float32))
    cov, state, seg = mk(torch.float64), mk(torch.int32), mk(torch.int32)
    flags = CLEAN_GCNORM | CLEAN_FILTSIZE | CLEAN_OUTLIERS | CLEAN_LOCALSD
    r = cv.sample_pipeline([db], [dm], [dh], np.array([L], np.int64), [1], out, cov, state, seg, counts_per_bin=100, bin_size=-1, mode=3, flags=flags)
    cv.synchronize()
    assert r["bin_size"] == exp["bin_size"] == C20.EXPECTED["bin_size"]
    assert r["total"] == C20.EXPECTED["n_binned"] and r["n_out"] == C20.EXPECTED["n_cleaned"]
    n = r["n_out"]
    ex = exp["cleaned"]
    for k in ("start", "stop", "gc"):
        assert (out[k][:n].cpu().numpy() == ex[k]).all(), k
    cnt = out["count"][:n].cpu().numpy()
    assert (cnt.view(np.uint32) == ex["count"].view(np.uint32)).all()
    assert r["lsd"] == ex["local_sd"]
    hc = cov[:n].cpu().numpy()
    assert (hc == exp["cov"]).all()
    assert (state[:n].cpu().numpy() == exp["paths"][0]).all()
    assert (seg[:n].cpu().numpy() == exp["hmm_ids"][0]).all()
    rows = [f"chr20\t{s}\t{e}\t{O.format_f2(float(v))}\t{g}" for s, e, v, g in zip(out["start"][:n].cpu().numpy(), out["stop"][:n].cpu().numpy(), cnt, out["gc"][:n].cpu().numpy())]
    assert C20.digest(rows) == C20.EXPECTED["cleaned"]
    prow = [f"chr20\t{s}\t{e}\t{O.format_g15(float(v))}\t{i}" for s, e, v, i in zip(out["start"][:n].cpu().numpy(), out["stop"][:n].cpu().numpy(), hc, seg[:n].cpu().numpy())]
    assert C20.digest(prow) == C20.EXPECTED["hmm"]
    # CBS on the same cleaned coverage
    seg_len, nseg, stats = cv.cbs(cov[:n], r["off"], 0.01, 10000)
    g = seg_len.cpu().numpy()[:nseg[0]]
    assert nseg[0] == len(exp["seg_len"][0]) and (g == exp["seg_len"][0]).all()
    es = exp["cbs_stats"]
    assert stats[0] == es[0] and stats[2] == es[2] and stats[4] == es[4]
