#!/usr/bin/env python3
"""Randomised parity soak of CanvasBin on the GPU against the oracle (not part of pytest).  usage: tools/soak_bin.py [minutes [seed]]"""
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
pad = lambda a: np.concatenate([a, np.zeros((-len(a)) % 64, a.dtype)])
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cv.device)
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
t0 = time.time(); it = 0; npath = {}
while time.time() - t0 < budget:
    nchr = int(rng.choice([1, 2, 5]))
    lengths = [int(rng.choice([4096, 4097, 65_537, 300_000, 1_000_003, 2_500_000])) + int(rng.randint(0, 5000)) for _ in range(nchr)]
    rate = float(rng.choice([0.02, 0.105, 0.21, 0.9]))
    seed = int(rng.randint(1, 2**31 - 1))
    thr = synth.poisson_thresholds(rate)
    data = [synth.generate_chromosome(seed, c, L, rate, thr) for c, L in enumerate(lengths)]
    if rng.rand() < 0.3:      # leading stretch without possible positions / a saturated pile-up
        b, h, m = data[0]; h = h.copy(); h[len(h) // 3: len(h) // 3 + 50] = 255; data[0] = (b, h, m)
    if rng.rand() < 0.5:      # leading 'n' stretch of random length (pos0 inside / at / beyond a tile; possible positions in front of it stay set)
        c = int(rng.randint(0, nchr)); b, h, m = data[c]; b = b.copy()
        k = int(rng.choice([1, 63, 64, 100, 4095, 4096, 4097, 9000, len(b)])); b[:min(k, len(b))] = ord("n"); data[c] = (b, h, m)
    path = "CANVAS_BIN_SINGLE_READ" if rng.rand() < 0.6 else "CANVAS_BIN_TWO_PASS"
    os.environ.pop("CANVAS_BIN_SINGLE_READ", None); os.environ.pop("CANVAS_BIN_TWO_PASS", None); os.environ[path] = "1"
    npath[path] = npath.get(path, 0) + 1
    bases = [dev(pad(b)) for b, h, m in data]; hits = [dev(pad(h)) for b, h, m in data]; masks = [dev(m.view(np.int64)) for b, h, m in data]
    lens = np.array(lengths, np.int64)
    mode = int(rng.choice([0, 3]))
    if rng.rand() < 0.5:
        bs = int(rng.choice([1, 2, 7, 64, 475, 534, 1000, 4096, 50_000]))
    else:
        _, _, r = cv.bin_rates(hits, masks, lens)
        rr = [O.bin_rate(h, m) for b, h, m in data]
        assert list(r) == rr, (seed, lengths)
        bs = cv.bin_size_from_rates(r, 100); assert bs == O.bin_size(rr, 100)
        if bs <= 0: continue
    if mode == 0:                 # Binary mode: the hit array holds 0 / 1 (CanvasBin.cs:259-262)
        data = [(b, np.minimum(h, 1), m) for b, h, m in data]; hits = [dev(pad(h)) for b, h, m in data]
    if rng.rand() < 0.4:          # the packed planes (host packer for the reference planes, device packer for the hit planes, or the other way round)
        from canvas_amd.lib import pack_reference_host, pack_hits_host
        path = "packed"; npath[path] = npath.get(path, 0) + 1
        if rng.rand() < 0.5:
            dref, dpl, pos0, _ = cv.pack_genome_device(bases, masks, hits, lens)
        else:
            hr = [pack_reference_host(np.ascontiguousarray(b), np.ascontiguousarray(m).view(np.uint64), len(b), threads=int(rng.choice([1, 4]))) for b, h, m in data]
            dref = [dev(r.view(np.int64)) for r, _ in hr]; pos0 = np.array([p for _, p in hr], np.int64)
            dpl = [dev(pack_hits_host(np.ascontiguousarray(h), len(h))[0].view(np.int64)) for b, h, m in data]
        cap = int(sum(lengths) // bs) + 8
        mk = lambda dt: torch.empty(cap, dtype=dt, device=cv.device)
        out = dict(chr=mk(torch.int32), start=mk(torch.int32), stop=mk(torch.int32), gc=mk(torch.int32), count=mk(torch.float32))
        out, per, total, _ = cv.bin_sample_packed(dref, dpl, lens, pos0, [1] * nchr, 100, bs, mode, out=out)
    else:
        out, per, total = cv.bin_genome(bases, masks, hits, lens, bs, mode)
    cv.synchronize()
    exp = [O.bin_chromosome(b, m, h, bs, mode) for b, h, m in data]
    assert total == sum(len(e[0]) for e in exp), (seed, lengths, bs, mode, total, path)
    if total:
        for k, j in (("start", 0), ("stop", 1), ("gc", 2)):
            assert (out[k][:total].cpu().numpy() == np.concatenate([e[j] for e in exp])).all(), (k, seed, lengths, bs, mode, path)
        assert (out["count"][:total].cpu().numpy() == np.concatenate([e[3] for e in exp]).astype(np.float32)).all(), (seed, lengths, bs, mode)
    it += 1
print(f"soak_bin: {it} random configurations bit-identical to the oracle in {time.time() - t0:.0f} s; paths {npath}")
