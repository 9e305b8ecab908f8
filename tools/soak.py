#!/usr/bin/env python3
"""Randomised parity soak on the GPU (not part of pytest): many seeds / sizes of Clean, F2, PerSampleHMM and segment ids against the oracle.
usage: tools/soak.py [minutes [seed]]"""
import os as _os; _os.environ.setdefault("CANVAS_TEST_HOOKS", "1")      # (the library reads its CANVAS_* switches only with this set)
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle_lib as O
from canvas_amd import Canvas, synth, CLEAN_GCNORM, CLEAN_FILTSIZE, CLEAN_OUTLIERS, CLEAN_LOCALSD

cv = Canvas(0); cv.profile_enable(True)
budget = float(sys.argv[1]) * 60 if len(sys.argv) > 1 else 120
t0 = time.time(); it = 0; fallbacks = 0; retries = 0; fb = {}; nbatch = 0; ncount = 0; ngconly = 0; by_noise = {}; by_disp = {}
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
while time.time() - t0 < budget:
    seed = int(rng.randint(1, 2**31 - 1)); n = int(rng.choice([3_000, 30_000, 120_000, 600_000])); nchr = int(rng.choice([1, 3, 24]))
    if os.environ.get("SOAK_N"): n = int(rng.choice([int(v) for v in os.environ["SOAK_N"].split(",")]))          # (hooks for hunting a rare case: fixed sizes / noise level)
    if os.environ.get("SOAK_NCHR"): nchr = int(os.environ["SOAK_NCHR"])
    bins = synth.generate_bins(seed, n, nchr=nchr)
    noise = rng.choice([0.0, 10.0, 40.0])
    if os.environ.get("SOAK_NOISE"): noise = float(os.environ["SOAK_NOISE"])
    if noise: bins["count"] = (bins["count"] + rng.normal(0, noise, len(bins["count"]))).clip(0).astype(np.float32)
    # two thirds of the samples carry two-decimal counts (what CanvasClean reads from a .binned file) at some level: the per-value counters decide their order statistics
    # when the level allows; the rest go through the radix selects
    shape = rng.choice(["as_is", "f2", "f2_scaled", "whole"])
    if shape == "whole":      # whole-number counts (what CanvasBin writes): with -g alone they take the three-launch stage of clean_gc_only.hpp
        bins["count"] = np.round(bins["count"] * float(rng.choice([1.0, 1.0, 3.0, 12.0]))).astype(np.float32)
    elif shape != "as_is":
        scale = 1.0 if shape == "f2" else float(rng.choice([0.05, 0.4, 1.7, 4.0]))
        bins["count"] = (np.round(bins["count"].astype(np.float64) * scale * 100.0) / 100.0).astype(np.float32)
    flags = int(rng.choice([CLEAN_GCNORM | CLEAN_FILTSIZE | CLEAN_OUTLIERS | CLEAN_LOCALSD, CLEAN_GCNORM, CLEAN_FILTSIZE | CLEAN_OUTLIERS, CLEAN_GCNORM | CLEAN_LOCALSD | CLEAN_FILTSIZE]))
    is_auto = synth.IS_AUTOSOME[:nchr]; is_y = np.zeros(nchr, np.uint8)
    ex = O.clean(bins["chr"], bins["start"], bins["stop"], bins["count"], bins["gc"], is_auto, is_y, flags)
    dev = {k: torch.from_numpy(np.ascontiguousarray(v)).to(cv.device) for k, v in bins.items()}
    if rng.rand() < 0.3:
        # the sample as one of a cohort of 2-4 different samples through canvas_clean_batch (one launch chain for all of them); the others are checked too
        others = []
        for _ in range(int(rng.randint(1, 4))):
            b2 = synth.generate_bins(int(rng.randint(1, 2**31 - 1)), int(rng.choice([3_000, 30_000, 120_000, 600_000])), nchr=nchr)
            others.append((b2, O.clean(b2["chr"], b2["start"], b2["stop"], b2["count"], b2["gc"], is_auto, is_y, flags)))
        devs = [dev] + [{k: torch.from_numpy(np.ascontiguousarray(v)).to(cv.device) for k, v in b2.items()} for b2, _ in others]
        pos = int(rng.randint(0, len(devs))); devs[0], devs[pos] = devs[pos], devs[0]
        exps = [ex] + [e for _, e in others]; exps[0], exps[pos] = exps[pos], exps[0]
        nouts, lsds, infos = cv.clean_batch(devs, [int(d["chr"].numel()) for d in devs], is_auto, flags)
        for d, e, no, ls in zip(devs, exps, nouts, lsds):
            assert int(no) == len(e["chr"]) and float(ls) == e["local_sd"], ("batch", seed, n, nchr, flags)
            assert (d["count"][:int(no)].cpu().numpy().view(np.uint32) == e["count"].view(np.uint32)).all() and (d["stop"][:int(no)].cpu().numpy() == e["stop"]).all(), ("batch", seed, n, nchr, flags)
        nbatch += 1
        n_out, lsd = int(nouts[pos]), float(lsds[pos])
        ncount += int(infos[pos][5]); ngconly += int(infos[pos][6])
    else:
        n_out, lsd, info = cv.clean(dev, len(bins["chr"]), is_auto, flags)
        ncount += int(info[5]); ngconly += int(info[6])
    assert n_out == len(ex["chr"]) and lsd == ex["local_sd"], (seed, n, nchr, flags)
    got = dev["count"][:n_out].cpu().numpy()
    assert (got.view(np.uint32) == ex["count"].view(np.uint32)).all() and (dev["start"][:n_out].cpu().numpy() == ex["start"]).all(), (seed, n, nchr, flags)
    if n_out >= 5:
        cov = cv.quantize_f2(dev["count"], n_out)
        off = cv.chromosome_offsets(dev["chr"], n_out, nchr)
        cv.profile_get("viterbi_sequential", reset=True); cv.profile_get("viterbi_retry", reset=True)
        st = cv.hmm_per_sample(cov, off).cpu().numpy()
        f1 = cv.profile_get("viterbi_sequential")[1]; fallbacks += f1; r1 = cv.profile_get("viterbi_retry")[1]; retries += r1
        rn = by_noise.setdefault(float(noise), [0, 0]); rn[0] += 1; rn[1] += 1 if r1 else 0
        if os.environ.get("SOAK_DISPERSION"):      # (how the need for a second attempt goes with the sample's relative dispersion: the quantity the first lead-in is chosen from)
            hq = np.percentile(cov.cpu().numpy(), [25, 50, 75]); rr = (hq[2] - hq[0]) / max(hq[1], 1e-9)
            key = min(20, int(rr / 0.05)); dv = by_disp.setdefault(key, [0, 0]); dv[0] += 1; dv[1] += 1 if r1 else 0
        if f1: fb[(n, nchr, float(noise))] = fb.get((n, nchr, float(noise)), 0) + 1
        hc = cov.cpu().numpy()
        if f1 and os.environ.get("SOAK_DUMP"):     # a sample whose speculative attempts all failed: the coverage PerSampleHMM saw, for a look at it on the CPU
            np.savez(os.path.join(os.environ["SOAK_DUMP"], "viterbi_fallback_%d.npz" % int(fallbacks)), cov=hc, off=np.asarray(off), seed=seed, n=n, nchr=nchr, noise=noise)
        per = [np.ascontiguousarray(hc[off[c]:off[c + 1]]) for c in range(nchr)]
        paths, ran = O.hmm_genome_per_sample(per, threads=8)
        for c in range(nchr):
            e = paths[c] if ran[c] else np.full(len(per[c]), -1, np.int32)
            assert (st[off[c]:off[c + 1]] == e).all(), ("hmm", seed, n, nchr, c)
    it += 1
print("fallback runs by (n, nchr, noise):", sorted(fb.items()))
print("PerSampleHMM calls that needed a second speculative attempt, by the noise added to the counts:", {k: "%d of %d" % (v[1], v[0]) for k, v in sorted(by_noise.items())})
if by_disp: print("... by iqr / median of the coverage (bins of 0.05):", {"%.2f" % (k * 0.05): "%d of %d" % (v[1], v[0]) for k, v in sorted(by_disp.items())})
print(f"soak: {it} random configurations bit-identical to the oracle in {time.time() - t0:.0f} s; {nbatch} of them inside a cohort call; {ncount} decided by the per-value counters; {ngconly} through the -g-only stage (clean_gc_only.hpp); second speculative attempts: {retries}, sequential Viterbi fallbacks: {fallbacks}")
