#!/usr/bin/env python3
"""GCContentWeighted binning alone on a few synthetic chromosomes (tumour 80x with fragment lengths): wall time per call + how many bins the interval decided.
usage: tools/gcw_probe.py [scale]"""
import sys, time
import numpy as np
import torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from canvas_amd import Canvas, synth
from canvas_amd.lib import synth_generate_device, synth_generate_sample_device

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
cv = Canvas(0); dev = cv.device
seed = 20260930
lens = [max(200_000, int(L * scale)) for L in synth.GRCH38]
thr_t = torch.from_numpy(synth.poisson_thresholds(0.28, purity=0.7).view(np.int32)).to(dev)
bases, masks, hits, fls = [], [], [], []
thr = None
for c, L in enumerate(lens):
    b, _, m, thr = synth_generate_device(seed, c, L, 0.28, dev, thr)
    h, f = synth_generate_sample_device(seed, seed + 1000, c, L, thr_t, dev, with_fraglen=True)
    bases.append(b); masks.append(m); hits.append(h); fls.append(f)
torch.cuda.synchronize()
cap = int(sum(lens) // 50) + 64
mk = lambda dt: torch.empty(cap, dtype=dt, device=dev)
out = dict(chr=mk(torch.int32), start=mk(torch.int32), stop=mk(torch.int32), gc=mk(torch.int32), count=mk(torch.float32))
la = np.array(lens, np.int64)
cv.profile_enable(1)
for rep in range(3):
    for n in ("gcw_read_gc", "gcw_weighted", "bin_summary"): cv.profile_get(n, reset=True)
    t0 = time.perf_counter()
    o, per, total, bs = cv.bin_sample_gcweighted(bases, masks, hits, fls, la, synth.IS_AUTOSOME, 100, -1, out=out)
    dt = time.perf_counter() - t0
    print("call %d: %.2f ms, %d bins (bin size %d), decided / replayed %s, read_gc %.2f ms, weighted %.2f ms, sweep %.2f ms" %
          (rep, dt * 1e3, total, bs, cv.bin_gcw_stats(), cv.profile_get("gcw_read_gc")[0], cv.profile_get("gcw_weighted")[0], cv.profile_get("bin_summary")[0]))
