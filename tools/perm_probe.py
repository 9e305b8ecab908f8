"""The device permutation engine on one segment (canvas_cbs_perm_probe): milliseconds of the generator and of the permutation + statistic kernel per batch,
and the first intervals against the oracle's XPerm + HTMaxP.  usage: python tools/perm_probe.py [kernel ...]"""
import os as _os; _os.environ.setdefault("CANVAS_TEST_HOOKS", "1")      # (the library reads its CANVAS_* switches only with this set)
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from canvas_amd.lib import Canvas
import oracle_lib as O

kernels = [int(a) for a in sys.argv[1:]] or [1]
cv = Canvas(0)
rng = np.random.default_rng(7)
for n in (3000, 20000, 67000, 131000, 234000, 476000):
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
                bad = 0; wid = 0.0; gave_up = int(np.sum(~np.isfinite(lohi[:, 0])))
                for b in (0, 1, 2, 255):
                    px = O.xperm(x, 12345, b); ex = O.htmaxp(px, tss)
                    if not (lohi[b, 0] <= ex <= lohi[b, 1]): bad += 1
                    wid = max(wid, (lohi[b, 1] - lohi[b, 0]) / abs(ex))
                chk = f" intervals 0, 1, 2, 255 vs oracle: {'contain the exact value' if bad == 0 else 'BAD %d' % bad} (relative width {wid:.1e}; given up {gave_up})"
            el = n * nb
            print(f"n {n} nb {nb} kernel {kernel}: generator {best[0]:.3f} + {best[1]:.3f} ms, permutation+statistic {best[2]:.3f} ms = {best[2] * 1e6 / el:.3f} ns/element ({el / best[2] / 1e6:.2f} G elements/s){chk}", flush=True)
cv.close()
