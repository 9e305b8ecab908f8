#!/usr/bin/env python3
"""Randomised parity soak of CanvasBin -m GCContentWeighted on the GPU against the oracle (not part of pytest): fragment-size regimes on both sides of the k_read_gc3 / k_read_gc2
switch (mean fragment 100), lengths that are no multiple of anything, chromosomes shorter than the window, hits without a length, lengths without a hit, negative / clipped
lengths, saturated hit counts; every third configuration through the forced kernels (CANVAS_GCW_READ_GC2, CANVAS_GCW_SERIAL).  usage: tools/soak_gcw.py [minutes [seed]]"""
import os as _os; _os.environ.setdefault("CANVAS_TEST_HOOKS", "1")      # (the library reads its CANVAS_* switches only with this set)
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle_lib as O
from canvas_amd import Canvas, synth

cv = Canvas(0)
budget = float(sys.argv[1]) * 60 if len(sys.argv) > 1 else 120
pad = lambda a: np.concatenate([a, np.zeros((-len(a)) % 16, a.dtype)])
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cv.device)
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
t0 = time.time(); it = 0; kinds = {}
while time.time() - t0 < budget:
    mean = int(rng.choice([40, 97, 100, 101, 102, 110, 150, 350, 520, 900, 2500]))
    sd = float(rng.choice([0, 3, 0.2 * mean]))
    nchr = int(rng.choice([1, 2, 4]))
    lengths = [int(rng.choice([3 * mean - 7, 3 * mean + 2, 9000, 65_537, 300_000, 700_001])) + int(rng.randint(0, 3000)) for _ in range(nchr)]
    lengths = [max(L, 50) for L in lengths]
    if max(lengths) < 100_000: lengths[0] = 300_000 + int(rng.randint(0, 5000))        # one chromosome long enough to carry the statistics
    rate = float(rng.choice([0.05, 0.21, 0.3, 0.9]))
    seed = int(rng.randint(1, 2**31 - 1))
    thr = synth.poisson_thresholds(rate)
    data = [synth.generate_chromosome(seed, c, L, rate, thr) for c, L in enumerate(lengths)]
    data = [(b, h.copy(), m) for b, h, m in data]
    fl = []
    for b, h, m in data:
        f = np.where(h > 0, np.clip(rng.normal(mean, sd, len(h)), 1, 32767), 0).astype(np.int16)
        f[rng.rand(len(h)) < rng.choice([0.0, 0.2, 0.9])] = 0
        add = (rng.rand(len(h)) < 0.01) & (h == 0); f[add] = mean
        if rng.rand() < 0.5: k = int(rng.randint(0, max(1, len(h) - 60))); f[k:k + 50] = -int(rng.randint(1, 300))
        if rng.rand() < 0.5: k = int(rng.randint(0, max(1, len(h) - 120))); f[k:k + 100] = int(rng.choice([3 * mean - 1, 3 * mean, 3 * mean + 1, 32767]))
        if rng.rand() < 0.3: h[rng.randint(0, len(h), 200)] = 255
        fl.append(f)
    if not any((f > 0).any() for f in fl): continue
    bs = int(rng.choice([24, 100, 411, 3000]))
    try:
        exp, mfrag, w, _ = O.bin_gc_weighted([d[0] for d in data], [d[2] for d in data], [d[1] for d in data], fl, bs)
    except Exception as ex:
        continue          # (no fragment size: the reference aborts; the library's error path has its own test)
    ex = np.concatenate([e[3] for e in exp]).astype(np.float32)
    bases = [dev(pad(b)) for b, h, m in data]; hits = [dev(pad(h)) for b, h, m in data]; masks = [dev(m.view(np.int64)) for b, h, m in data]
    dfl = [dev(pad(f)) for f in fl]
    lens = np.array(lengths, np.int64)
    cap = int(lens.sum() // max(4, bs // 8)) + 64
    mk = lambda dt: torch.empty(cap, dtype=dt, device=cv.device)
    out = dict(chr=mk(torch.int32), start=mk(torch.int32), stop=mk(torch.int32), gc=mk(torch.int32), count=mk(torch.float32))
    kind = int(rng.choice([0, 0, 0, 1, 2]))
    for k in ("CANVAS_GCW_READ_GC2", "CANVAS_GCW_SERIAL"): os.environ.pop(k, None)
    if kind == 1: os.environ["CANVAS_GCW_READ_GC2"] = "1"
    if kind == 2: os.environ["CANVAS_GCW_SERIAL"] = "1"
    o, per, total, _ = cv.bin_sample_gcweighted(bases, masks, hits, dfl, lens, [1] * nchr, 100, bs, out=out)
    got = out["count"][:total].cpu().numpy()
    assert total == len(ex) and (got == ex).all(), (seed, mean, sd, lengths, rate, bs, kind, mfrag, np.nonzero(got[:len(ex)] != ex[:len(got)])[0][:5])
    assert (out["stop"][:total].cpu().numpy() == np.concatenate([e[1] for e in exp])).all(), (seed, mean, lengths, bs)
    key = ("gc3" if mfrag > 100 and kind != 1 else "gc2") + ("+serial" if kind == 2 else "")
    kinds[key] = kinds.get(key, 0) + 1
    it += 1
print("soak_gcw: %d configurations identical to the oracle in %.0f s %s" % (it, time.time() - t0, kinds), flush=True)
cv.close()
