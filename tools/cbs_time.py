import time, numpy as np, torch, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from canvas_amd import synth
from canvas_amd.lib import Canvas
cv = Canvas(0)
n = 4_700_000
bins = synth.generate_bins(20260927 + 2, n)
cov = np.round(bins["count"].astype(np.float64), 2)
off = np.concatenate([[0], np.cumsum(np.bincount(bins["chr"], minlength=24))]).astype(np.int64)
d = torch.from_numpy(cov).to(cv.device)
for undo in (0,):
    t = time.perf_counter()
    seg_len, nseg, stats = cv.cbs(d, off, 0.01, 10000, undo=undo)
    dt = time.perf_counter() - t
    print("cbs WGS-size", n, "bins:", round(dt, 3), "s; segments", int(sum(nseg)), "stats [tmaxo_calls, tmaxo_elems, perms, perm_elems, tpermp_draws, tailp_exits, gpu_searches, tie_replays] =", list(map(int, stats)))
print("device stats [dev perms, host perms, exact rechecks, batches, verified, violations]:", list(map(int, cv.cbs_device_stats())))
t = time.perf_counter(); seg_len, nseg, stats = cv.cbs(d, off, 0.01, 10000); print("second call", round(time.perf_counter() - t, 3), "s")
