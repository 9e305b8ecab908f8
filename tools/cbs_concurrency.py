"""Is canvas_cbs on the tumour / normal coverage bound by the device's throughput or by its own chains of dependent launches?  The same call from one context, then from two
contexts at once (two host threads): throughput-bound work takes twice as long, latency-bound work about the same.   usage: python tools/cbs_concurrency.py"""
import os as _os; _os.environ.setdefault("CANVAS_TEST_HOOKS", "1")
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from canvas_amd import Canvas, synth, CLEAN_GCNORM, CLEAN_FILTSIZE, CLEAN_OUTLIERS, CLEAN_LOCALSD
from canvas_amd.lib import synth_generate_device, synth_generate_sample_device
cv = Canvas(0); dev = cv.device
lens = np.array(synth.GRCH38, np.int64); seed = 20260930
thr = None; bases = []; masks = []
for c, L in enumerate(lens):
    b, h, m, thr = synth_generate_device(seed, c, int(L), 0.21, dev, thr); bases.append(b); masks.append(m)
thr_t = torch.from_numpy(synth.poisson_thresholds(0.28, purity=0.7).view(np.int32)).to(dev)
thr_n = torch.from_numpy(synth.poisson_thresholds(0.14, flat=True).view(np.int32)).to(dev)
hits_t, fl_t, hits_n = [], [], []
for c, L in enumerate(lens):
    h, f = synth_generate_sample_device(seed, seed + 1000, c, int(L), thr_t, dev, with_fraglen=True); hits_t.append(h); fl_t.append(f)
    h, _ = synth_generate_sample_device(seed, seed + 2000, c, int(L), thr_n, dev); hits_n.append(h)
torch.cuda.synchronize()
flags = CLEAN_GCNORM | CLEAN_FILTSIZE | CLEAN_OUTLIERS | CLEAN_LOCALSD
r = cv.tumor_normal_flow(bases, masks, hits_t, fl_t, hits_n, lens, synth.IS_AUTOSOME, flags, 0.01, 10000, keep=True)
cov, off = r["cov"], r["chr_offset"]
del bases, masks, hits_t, fl_t, hits_n; torch.cuda.empty_cache()
cvs = [cv, Canvas(0)]
def run(c, out, i):
    t = time.perf_counter(); c.cbs(cov, off, 0.01, 10000); out[i] = time.perf_counter() - t
for c in cvs: run(c, [0, 0], 0)          # warm both contexts
for rep in range(2):
    one = [0]; run(cvs[0], one, 0)
    two = [0, 0]; th = [threading.Thread(target=run, args=(cvs[i], two, i)) for i in range(2)]
    t = time.perf_counter(); [x.start() for x in th]; [x.join() for x in th]; wall = time.perf_counter() - t
    print(f"one call {one[0]:.3f} s; two calls at once {wall:.3f} s (each {two[0]:.3f}, {two[1]:.3f}): ratio {wall / one[0]:.2f}", flush=True)
# ---- the longest chromosome by itself, then the first 2 / 4 / 8: how much of the call's time is chromosome 1's own chain of dependent batches
for k in (1, 2, 4, 8, 24):
    offk = np.ascontiguousarray(off[:k + 1]); ck = cov[:int(offk[-1])]
    cvs[0].cbs(ck, offk, 0.01, 10000); t = time.perf_counter(); cvs[0].cbs(ck, offk, 0.01, 10000); print(f"first {k} chromosomes: {time.perf_counter() - t:.3f} s", flush=True)
