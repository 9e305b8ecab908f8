import os as _os; _os.environ.setdefault("CANVAS_TEST_HOOKS", "1")      # (the library reads its CANVAS_* switches only with this set)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); 
import numpy as np
from canvas_amd.lib import Canvas
cv = Canvas(0); rng = np.random.default_rng(7)
for n in (67000, 234000):
    x = rng.standard_normal(n); x -= x.mean(); tss = float(np.sum(x * x))
    for nb in (512, 4096):
        lohi, ms = cv.cbs_perm_probe(x, 12345, nb, 2, tss); print(n, nb, ms, flush=True)
