#!/usr/bin/env python3
"""Randomised parity soak of canvas_wavelets on the GPU against the oracle (not part of pytest).  usage: tools/soak_wavelets.py [minutes [seed]]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle_lib as O
from canvas_amd import Canvas

cv = Canvas(0)
budget = float(sys.argv[1]) * 60 if len(sys.argv) > 1 else 120
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 17)
t0 = time.time(); it = 0; redone = 0; nbp = 0
while time.time() - t0 < budget:
    nchr = int(rng.choice([1, 2, 4]))
    per = []
    for _ in range(nchr):
        n = int(rng.choice([5, 11, 12, 200, 257, 313, 1000, 5000, 20000, 60000])) + int(rng.randint(0, 50))
        mean = float(rng.choice([5, 30, 100, 1000]))
        x = rng.poisson(mean, n).astype(np.float64)
        for _ in range(int(rng.randint(0, 8)) if n > 60 else 0):
            a = int(rng.randint(0, n - 20)); b = min(n, a + int(rng.choice([11, 25, 100, 700, n // 3 + 1])))
            x[a:b] = np.round(x[a:b] * float(rng.choice([0.0, 0.5, 1.5, 2.0, 3.0])))
        kind = rng.rand()
        if kind < 0.15: x = np.round(x * (1 + 0.1 * np.sin(np.arange(n) / float(rng.choice([50, 700])))))      # waviness
        elif kind < 0.25: x = np.round(x / 10) * 10                                                              # heavy ties
        elif kind < 0.30: x[:] = x[0]                                                                            # flat
        per.append(np.round(x * 100) / 100)
    germ = bool(rng.rand() < 0.5)
    window = int(rng.choice([11, 100, 1000, 20000, 100000]))
    kw = dict(is_germline=germ, window=window, mad_factor=float(rng.choice([2.0, 5.0])), thr_lower=float(rng.choice([0.05, 5.0])), min_size=int(rng.choice([10, 10, 4])))
    exp = O.wavelets_genome(per, **kw)
    cov = np.ascontiguousarray(np.concatenate(per)); off = np.concatenate([[0], np.cumsum([len(a) for a in per])]).astype(np.int64)
    kg = dict(kw); kg['threshold_lower'] = kg.pop('thr_lower')
    got = cv.wavelets(torch.from_numpy(cov).to(cv.device), off, **kg)
    for c in range(nchr):
        assert got[c].tolist() == exp[c].tolist(), (it, c, len(per[c]), kw)
    redone += int(cv.wavelets_stats()[1]); nbp += sum(len(e) for e in exp)
    it += 1
print(f"soak_wavelets: {it} random configurations identical to the oracle in {time.time() - t0:.0f} s; {nbp} breakpoints; nodes that needed the exact chain: {redone}")
