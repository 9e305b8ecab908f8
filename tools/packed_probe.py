"""packed-planes pipeline in a loop (for rocprofv3 --kernel-trace --stats): python tools/packed_probe.py [steps] [scale]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from canvas_amd import Canvas, synth, CLEAN_GCNORM, CLEAN_FILTSIZE, CLEAN_OUTLIERS, CLEAN_LOCALSD
from canvas_amd.lib import synth_generate_device

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cv = Canvas(0)
dev = cv.device
lengths = [max(200_000, int(L * scale)) for L in synth.GRCH38]
lens = np.array(lengths, np.int64)
thr = None; bases = []; hits = []; masks = []
for c, L in enumerate(lengths):
    b, h, m, thr = synth_generate_device(20260930, c, L, 0.21, dev, thr)
    bases.append(b); hits.append(h); masks.append(m)
flags = CLEAN_GCNORM | CLEAN_FILTSIZE | CLEAN_OUTLIERS | CLEAN_LOCALSD
cap = int(lens.sum() // 100) + 16
mk = lambda dt: torch.empty(cap, dtype=dt, device=dev)
out = dict(chr=mk(torch.int32), start=mk(torch.int32), stop=mk(torch.int32), gc=mk(torch.int32), count=mk(torch.float32))
cov, st, seg = mk(torch.float64), mk(torch.int32), mk(torch.int32)
torch.cuda.synchronize()
dref, dpl, pos0, sat = cv.pack_genome_device(bases, masks, hits, lens)
del bases, hits, masks
r = cv.sample_pipeline(dref, None, dpl, lens, synth.IS_AUTOSOME, out, cov, st, seg, flags=flags, pos0=pos0)
cv.synchronize()
ts = []
for i in range(steps):
    t0 = time.perf_counter()
    cv.sample_pipeline(None, None, None, None, None, None, None, None, None, prepared=r["prepared"])
    cv.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("packed pass ms:", [round(t, 3) for t in ts], "bins", r["total"], "segments", r["nseg"])
