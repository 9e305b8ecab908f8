#!/usr/bin/env python3
"""Randomised parity soak on the GPU (not part of pytest): many seeds / sizes of Clean, F2, PerSampleHMM and segment ids against the oracle.
usage: tools/soak.py [minutes [seed]]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle_lib as O
from canvas_amd import Canvas, synth, CLEAN_GCNORM, CLEAN_FILTSIZE, CLEAN_OUTLIERS, CLEAN_LOCALSD

cv = Canvas(0); cv.profile_enable(True)
budget = float(sys.argv[1]) * 60 if len(sys.argv) > 1 else 120
t0 = time.time(); it = 0; fallbacks = 0; retries = 0; fb = {}
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
while time.time() - t0 < budget:
    seed = int(rng.randint(1, 2**31 - 1)); n = int(rng.choice([3_000, 30_000, 120_000, 600_000])); nchr = int(rng.choice([1, 3, 24]))
    bins = synth.generate_bins(seed, n, nchr=nchr)
    noise = rng.choice([0.0, 10.0, 40.0])
    if noise: bins["count"] = (bins["count"] + rng.normal(0, noise, len(bins["count"]))).clip(0).astype(np.float32)
    flags = int(rng.choice([CLEAN_GCNORM | CLEAN_FILTSIZE | CLEAN_OUTLIERS | CLEAN_LOCALSD, CLEAN_GCNORM, CLEAN_FILTSIZE | CLEAN_OUTLIERS, CLEAN_GCNORM | CLEAN_LOCALSD | CLEAN_FILTSIZE]))
    is_auto = synth.IS_AUTOSOME[:nchr]; is_y = np.zeros(nchr, np.uint8)
    ex = O.clean(bins["chr"], bins["start"], bins["stop"], bins["count"], bins["gc"], is_auto, is_y, flags)
    dev = {k: torch.from_numpy(np.ascontiguousarray(v)).to(cv.device) for k, v in bins.items()}
    n_out, lsd, info = cv.clean(dev, len(bins["chr"]), is_auto, flags)
    assert n_out == len(ex["chr"]) and lsd == ex["local_sd"], (seed, n, nchr, flags)
    got = dev["count"][:n_out].cpu().numpy()
    assert (got.view(np.uint32) == ex["count"].view(np.uint32)).all() and (dev["start"][:n_out].cpu().numpy() == ex["start"]).all(), (seed, n, nchr, flags)
    if n_out >= 5:
        cov = cv.quantize_f2(dev["count"], n_out)
        off = cv.chromosome_offsets(dev["chr"], n_out, nchr)
        cv.profile_get("viterbi_sequential", reset=True); cv.profile_get("viterbi_retry", reset=True)
        st = cv.hmm_per_sample(cov, off).cpu().numpy()
        f1 = cv.profile_get("viterbi_sequential")[1]; fallbacks += f1; retries += cv.profile_get("viterbi_retry")[1]
        if f1: fb[(n, nchr, float(noise))] = fb.get((n, nchr, float(noise)), 0) + 1
        hc = cov.cpu().numpy()
        per = [np.ascontiguousarray(hc[off[c]:off[c + 1]]) for c in range(nchr)]
        paths, ran = O.hmm_genome_per_sample(per, threads=8)
        for c in range(nchr):
            e = paths[c] if ran[c] else np.full(len(per[c]), -1, np.int32)
            assert (st[off[c]:off[c + 1]] == e).all(), ("hmm", seed, n, nchr, c)
    it += 1
print("fallback runs by (n, nchr, noise):", sorted(fb.items()))
print(f"soak: {it} random configurations bit-identical to the oracle in {time.time() - t0:.0f} s; second speculative attempts: {retries}, sequential Viterbi fallbacks: {fallbacks}")
