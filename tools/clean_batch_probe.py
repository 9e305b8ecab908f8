"""canvas_clean_batch on B copies of one WGS-size bin list (for rocprofv3): python tools/clean_batch_probe.py [B] [reps] [count scale]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from canvas_amd import Canvas, synth, CLEAN_GCNORM, CLEAN_FILTSIZE, CLEAN_OUTLIERS, CLEAN_LOCALSD
from canvas_amd.lib import synth_generate_device
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cv = Canvas(0); dev = cv.device
lengths = list(synth.GRCH38); lens = np.array(lengths, np.int64)
thr = None; bases = []; hits = []; masks = []
for c, L in enumerate(lengths):
    b, h, m, thr = synth_generate_device(20260930, c, L, 0.21, dev, thr)
    bases.append(b); hits.append(h); masks.append(m)
torch.cuda.synchronize()
cap = int(lens.sum() // 100) + 16
mk = lambda dt: torch.empty(cap, dtype=dt, device=dev)
out = dict(chr=mk(torch.int32), start=mk(torch.int32), stop=mk(torch.int32), gc=mk(torch.int32), count=mk(torch.float32))
_, per, total, bs = cv.bin_sample(bases, masks, hits, lens, synth.IS_AUTOSOME, 100, -1, 3, out=out)
binned = {k: v[:total].clone() for k, v in out.items()}
if len(sys.argv) > 3:          # two-decimal counts at another level (e.g. 0.37: the pseudo-counts of a tumour / normal ratio) instead of integer read counts
    binned["count"] = (torch.round(binned["count"].double() * float(sys.argv[3]) * 100.0) / 100.0).float()
flags = CLEAN_GCNORM | CLEAN_FILTSIZE | CLEAN_OUTLIERS | CLEAN_LOCALSD
for r in range(reps):
    copies = [{k: v.clone() for k, v in binned.items()} for _ in range(B)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nout, lsd, _ = cv.clean_batch(copies, [total] * B, synth.IS_AUTOSOME, flags)
    dt = time.perf_counter() - t0
    print(f"B={B}: {dt * 1e3:.3f} ms per batch, {dt / B * 1e3:.4f} ms per sample, bins per sample {total}, n_out {int(nout[0])}")
