"""The device permutation engine on one segment (canvas_cbs_perm_probe): milliseconds of the generator and of the permutation + statistic kernel per batch,
and the first intervals against the oracle's XPerm + HTMaxP.  usage: python tools/perm_probe.py [kernel ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from canvas_amd.lib import Canvas
import oracle_lib as O

kernels = [int(a) for a in sys.argv[1:]] or [1]
cv = Canvas(0)
rng = np.random.default_rng(7)
for n in (20000, 67000, 131000, 234000):
    x = rng.standard_normal(n); x[n // 3: n // 3 + n // 5] += 0.15; x -= x.mean(); x = np.round(x, 4); x -= x.sum() / n
    tss = float(np.sum(x * x))
    for kernel in kernels:
        for nb in (256, 1024, 4096):
            if n * nb > (1 << 30): continue
            best = None
            for rep in range(3):
                lohi, ms = cv.cbs_perm_probe(x, 12345, nb, kernel, tss)
                if best is None or ms[2] < best[2]: best = ms.copy()
            chk = ""
            if nb == 256:
                bad = 0
                for b in range(4):
                    px = O.xperm(x, 12345, b); ex = O.htmaxp(px, tss)
                    if not (lohi[b, 0] <= ex <= lohi[b, 1]) or not (lohi[b, 1] - lohi[b, 0] <= 1e-6 * abs(ex)): bad += 1
                chk = f" first 4 intervals vs oracle: {'ok' if bad == 0 else 'BAD %d' % bad}"
            el = n * nb
            print(f"n {n} nb {nb} kernel {kernel}: generator {best[0]:.3f} + {best[1]:.3f} ms, permutation+statistic {best[2]:.3f} ms = {best[2] * 1e6 / el:.3f} ns/element ({el / best[2] / 1e6:.2f} G elements/s){chk}", flush=True)
cv.close()
