import sys, time, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from canvas_amd import Canvas
cv = Canvas(0); cv.profile_enable(True)
rng = np.random.RandomState(3)
for n in (100_000, 400_000):
    x = np.round(rng.poisson(100, n) * 1.0); x[n//3:n//2] *= 1.5
    d = torch.from_numpy(x).to(cv.device); off = np.array([0, n], np.int64)
    cv.wavelets(d, off)
    cv.profile_get("wavelet_chain", reset=True)
    t = time.perf_counter(); bp = cv.wavelets(d, off); dt = time.perf_counter() - t
    ms, k = cv.profile_get("wavelet_chain"); st = cv.wavelets_stats()
    print("decisions [closed-form, undecided, exact chains, closed form on]:", cv.wavelets_decisions())
    print(f"n={n}: total {dt*1e3:.1f} ms, chain kernels {ms:.1f} ms over {k} launches, levels {st[0]}, first-level est {ms/max(1,k):.2f} ms avg; breakpoints {len(bp[0])}")
