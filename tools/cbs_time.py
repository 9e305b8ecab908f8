"""Wall time of canvas_cbs on a synthetic genome (tools only; set CANVAS_CBS_TIMING=1 / CANVAS_CBS_DEBUG_BATCHES=1 for the library's own breakdown).

usage: python tools/cbs_time.py [bins] [calls] [seed offset]
"""
import os as _os; _os.environ.setdefault("CANVAS_TEST_HOOKS", "1")      # (the library reads its CANVAS_* switches only with this set)
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from canvas_amd import synth
from canvas_amd.lib import Canvas

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_700_000
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 3
so = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cv = Canvas(0)
bins = synth.generate_bins(20260927 + so, n)
cov = np.round(bins["count"].astype(np.float64), 2)
off = np.concatenate([[0], np.cumsum(np.bincount(bins["chr"], minlength=24))]).astype(np.int64)
d = torch.from_numpy(cov).to(cv.device)
ts = []
for r in range(calls):
    t = time.perf_counter(); seg_len, nseg, stats = cv.cbs(d, off, 0.01, 10000); ts.append(time.perf_counter() - t)
print("cbs calls:", [round(x, 3) for x in ts], "segments", int(sum(nseg)), "perms", int(stats[2]), "device stats", cv.cbs_device_stats(), flush=True)
cv.close()
