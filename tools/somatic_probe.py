"""BASELINE configs[4] flow twice (for rocprofv3): python tools/somatic_probe.py"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from canvas_amd import Canvas, synth, CLEAN_GCNORM, CLEAN_FILTSIZE, CLEAN_OUTLIERS, CLEAN_LOCALSD
from canvas_amd.lib import synth_generate_device, synth_generate_sample_device
cv = Canvas(0); dev = cv.device
lens = np.array(synth.GRCH38, np.int64); seed = 20260930
thr = None; bases = []; masks = []
for c, L in enumerate(lens):
    b, h, m, thr = synth_generate_device(seed, c, int(L), 0.21, dev, thr); bases.append(b); masks.append(m)
rt, rn = 0.28, 0.14
thr_t = torch.from_numpy(synth.poisson_thresholds(rt, purity=0.7).view(np.int32)).to(dev)
thr_n = torch.from_numpy(synth.poisson_thresholds(rn, flat=True).view(np.int32)).to(dev)
hits_t, fl_t, hits_n = [], [], []
for c, L in enumerate(lens):
    h, f = synth_generate_sample_device(seed, seed + 1000, c, int(L), thr_t, dev, with_fraglen=True); hits_t.append(h); fl_t.append(f)
    h, _ = synth_generate_sample_device(seed, seed + 2000, c, int(L), thr_n, dev); hits_n.append(h)
torch.cuda.synchronize()
flags = CLEAN_GCNORM | CLEAN_FILTSIZE | CLEAN_OUTLIERS | CLEAN_LOCALSD
for rep in range(2):
    t0 = time.perf_counter()
    r = cv.tumor_normal_flow(bases, masks, hits_t, fl_t, hits_n, lens, synth.IS_AUTOSOME, flags, 0.01, 10000)
    print(f"flow {time.perf_counter() - t0:.3f} s", r["stage_seconds"], "cbs stats", [int(x) for x in r["cbs_stats"]], "device stats", [int(x) for x in cv.cbs_device_stats()])
