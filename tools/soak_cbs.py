#!/usr/bin/env python3
"""Randomised parity soak of CBS (device arc search + device permutation engine) against the oracle: segment lengths and RNG consumption.
usage: tools/soak_cbs.py [minutes [seed]]"""
import os as _os; _os.environ.setdefault("CANVAS_TEST_HOOKS", "1")      # (the library reads its CANVAS_* switches only with this set)
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle_lib as O
from canvas_amd import Canvas

cv = Canvas(0)
budget = float(sys.argv[1]) * 60 if len(sys.argv) > 1 else 120
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
t0 = time.time(); it = 0; devp = 0; rechecks = 0
while time.time() - t0 < budget:
    nchr = int(rng.choice([1, 2, 4]))
    parts = []
    for c in range(nchr):
        n = int(rng.choice([300, 1500, 5000, 12000, 30000]))
        sd = float(rng.choice([3.0, 10.0, 25.0]))
        x = rng.normal(100, sd, n)
        for _ in range(int(rng.randint(0, 5))):
            a = int(rng.randint(0, n)); ln = int(rng.choice([5, 40, 400, n // 3 + 1])); x[a:a + ln] += float(rng.choice([-1, 1])) * sd * float(rng.choice([0.15, 0.4, 1.0, 3.0]))
        if rng.rand() < 0.3: x = np.round(x)              # heavy quantisation: exact ties
        if rng.rand() < 0.3:                              # isolated spikes: a block's extremes next to each other, where the reference's arc search leaves arcs out (DESIGN, CBS)
            sp = rng.choice(n, max(1, n // int(rng.choice([50, 300, 2000]))), replace=False)
            x[sp] += rng.standard_cauchy(len(sp)) * sd * 5; x = np.clip(x, 0, 20000)
        parts.append(np.round(x, 2))
    cov = np.concatenate(parts)
    off = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int64)
    nperm = int(rng.choice([500, 2000, 10000]))
    undo = int(rng.choice([0, 0, 1, 2]))
    exp, est = O.cbs_genome(parts, 0.01, nperm, threads=8, undo=undo)
    seg_len, nseg, stats = cv.cbs(torch.from_numpy(cov).to(cv.device), off, 0.01, nperm, undo=undo)
    got = seg_len.cpu().numpy()
    for c in range(nchr):
        g = got[off[c]:off[c] + nseg[c]]
        assert nseg[c] == len(exp[c]) and (g == exp[c]).all(), (it, c, g[:8], exp[c][:8])
    assert stats[0] == est[0] and stats[2] == est[2] and stats[4] == est[4], (it, list(stats), list(est))
    d = cv.cbs_device_stats(); devp += int(d[0]); rechecks += int(d[2])
    assert int(d[5]) == 0, (it, "CANVAS_CBS_TEST_VERIFY: a device statistic or generator snapshot disagrees with the host", list(d))      # (only counted when the hook is set)
    it += 1
print(f"soak_cbs: {it} random configurations with identical segments and RNG consumption in {time.time() - t0:.0f} s; {devp} device permutations, {rechecks} exact re-evaluations")
