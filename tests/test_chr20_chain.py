"""BASELINE configs[0]: chr20-only Germline-WGS 30x, CanvasBin -> CanvasClean -> CanvasPartition on the CPU (plumbing, no GPU), at the real chromosome length
L = 64 444 167.  The chain runs on the oracle; its result is pinned by digests of the three files' rows so that any change of the oracle (the checker of
every GPU test) shows up here, and tests/test_chr20_chain_gpu.py compares the HIP path with this same chain on the same bytes."""
import hashlib

import numpy as np

import oracle_flows as OF
from canvas_amd import synth

L_CHR20 = 64_444_167
SEED = 20260927 + 1                    # seeds = 20260927 + config#
RATE = 0.105                           # 30x


def chr20_inputs():
    thr = synth.poisson_thresholds(RATE)
    return synth.generate_chromosome(SEED, 19, L_CHR20, RATE, thr)          # chromosome index 19 = chr20


def digest(rows):
    return hashlib.sha256("\n".join(rows).encode()).hexdigest()[:16]


# digests of the committed run (this file's chain on the committed oracle): binned / cleaned / partitioned (PerSampleHMM) / partitioned (CBS)
EXPECTED = {"bin_size": 1037, "n_binned": 51467, "n_cleaned": 50281, "binned": "a20375db6c139f60", "cleaned": "2384a3b422cf8009", "hmm": "09890aeb02d4bb14", "cbs": "09890aeb02d4bb14"}


def run_chain():
    b, h, m = chr20_inputs()
    return OF.germline_single([b], [m], [h], np.array([1], np.uint8), ["chr20"])


def test_chr20_oracle_chain_properties_and_pinned_result():
    r = run_chain()
    B, ex = r["binned"], r["cleaned"]
    # CanvasBin: ~100 counts per bin by construction of the bin size; bins tile the chromosome in order
    assert 900 <= r["bin_size"] <= 1100
    assert 50_000 < len(B["chr"]) < 62_000
    assert (B["start"][1:] >= B["stop"][:-1]).all() and (B["stop"] > B["start"]).all()
    assert 90 < float(np.median(B["count"])) < 110
    # CanvasClean keeps most bins and leaves the median where it was (GC normalisation rescales to the global median)
    assert 0.85 * len(B["chr"]) < len(ex["chr"]) <= len(B["chr"])
    assert len(ex["chr"]) >= 50_000 and ex["local_sd"] > 0                  # the local-SD metric applies from 50000 bins on (CanvasClean.cs:483-486)
    # CanvasPartition: both methods find the planted copy-number segments; ids are a running counter from 0
    for ids in (r["hmm_ids"][0], r["cbs_ids"][0]):
        assert ids[0] == 0 and (np.diff(ids) >= 0).all() and (np.diff(ids) <= 1).all() and ids[-1] >= 3
    got = {"bin_size": int(r["bin_size"]), "n_binned": len(r["binned_rows"]), "n_cleaned": len(r["cleaned_rows"]), "binned": digest(r["binned_rows"]),
           "cleaned": digest(r["cleaned_rows"]), "hmm": digest(r["partitioned_hmm_rows"]), "cbs": digest(r["partitioned_cbs_rows"])}
    assert got == EXPECTED, got
