"""canvas_clean on one WGS-size bin list, repeated (for rocprofv3 / timing): python tools/clean_probe.py [reps] [rate] [g]
prints the hipEvent time of the stage (the library's own clean_total scope) per call"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from canvas_amd import Canvas, synth, CLEAN_GCNORM, CLEAN_FILTSIZE, CLEAN_OUTLIERS, CLEAN_LOCALSD
from canvas_amd.lib import synth_generate_device
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
rate = float(sys.argv[2]) if len(sys.argv) > 2 else 0.21
cv = Canvas(0); dev = cv.device
lengths = list(synth.GRCH38); lens = np.array(lengths, np.int64)
thr = None; bases = []; hits = []; masks = []
for c, L in enumerate(lengths):
    b, h, m, thr = synth_generate_device(20260930, c, L, rate, dev, thr)
    bases.append(b); hits.append(h); masks.append(m)
torch.cuda.synchronize()
cap = int(lens.sum() // 100) + 16
mk = lambda dt: torch.empty(cap, dtype=dt, device=dev)
out = dict(chr=mk(torch.int32), start=mk(torch.int32), stop=mk(torch.int32), gc=mk(torch.int32), count=mk(torch.float32))
_, per, total, bs = cv.bin_sample(bases, masks, hits, lens, synth.IS_AUTOSOME, 100, -1, 3, out=out)
binned = {k: v[:total].clone() for k, v in out.items()}
del bases, hits, masks
flags = CLEAN_GCNORM | CLEAN_FILTSIZE | CLEAN_OUTLIERS | CLEAN_LOCALSD
if len(sys.argv) > 3 and sys.argv[3] == "g": flags = CLEAN_GCNORM          # BASELINE configs[1]: -g alone (clean_gc_only.hpp)
cv.profile_enable(True)
work = {k: v.clone() for k, v in binned.items()}
for r in range(reps):
    for k in work: work[k].copy_(binned[k])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_out, lsd, info = cv.clean(work, total, synth.IS_AUTOSOME, flags)
    dt = time.perf_counter() - t0
    ms, k = cv.profile_get("clean_total")
    print(f"{total} bins -> {n_out}, localSD {lsd:.6f}, info {list(info)}: host {dt * 1e3:.3f} ms, device scope {ms / max(1, k) * 1e3:.1f} us, 232 B/bin fraction of 8 TB/s: {232.0 * total / (ms / max(1, k) * 1e-3) / 8e12:.3f}")
