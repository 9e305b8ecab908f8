"""Two (or more) samples in flight on one GPU: one context + host thread per sample, passes back to back.  usage: tools/overlap_probe.py [samples] [steps] [packed]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from canvas_amd import Canvas, synth, CLEAN_GCNORM, CLEAN_FILTSIZE, CLEAN_OUTLIERS, CLEAN_LOCALSD
from canvas_amd.lib import synth_generate_device

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
packed = len(sys.argv) > 3 and sys.argv[3] == "packed"
dev = torch.device("cuda", 0)
lengths = list(synth.GRCH38); lens = np.array(lengths, np.int64)
flags = CLEAN_GCNORM | CLEAN_FILTSIZE | CLEAN_OUTLIERS | CLEAN_LOCALSD
cap = int(lens.sum() // 100) + 16
thr = None; bases = []; hits = []; masks = []
for c, L in enumerate(lengths):
    b, h, m, thr = synth_generate_device(20260930, c, L, 0.21, dev, thr)
    bases.append(b); hits.append(h); masks.append(m)
torch.cuda.synchronize()
cvs = [Canvas(0) for _ in range(S)]
if packed:
    dref, dpl, pos0, _ = cvs[0].pack_genome_device(bases, masks, hits, lens)
prep = []
for cv in cvs:
    mk = lambda dt: torch.empty(cap, dtype=dt, device=dev)
    out = dict(chr=mk(torch.int32), start=mk(torch.int32), stop=mk(torch.int32), gc=mk(torch.int32), count=mk(torch.float32))
    cov, st, seg = mk(torch.float64), mk(torch.int32), mk(torch.int32)
    if packed:
        r = cv.sample_pipeline(dref, None, dpl, lens, synth.IS_AUTOSOME, out, cov, st, seg, flags=flags, pos0=pos0)
    else:
        r = cv.sample_pipeline(bases, masks, hits, lens, synth.IS_AUTOSOME, out, cov, st, seg, flags=flags)
    cv.synchronize()
    prep.append((r["prepared"], out, seg, r, cov, st))

def run(i, n):
    cv = cvs[i]
    for _ in range(n):
        cv.sample_pipeline(None, None, None, None, None, None, None, None, None, prepared=prep[i][0])
    cv.synchronize()

for nthreads in ([1, S] if S > 1 else [1]):
    run(0, 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(i, steps)) for i in range(nthreads)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tot = prep[0][3]["total"] * steps * nthreads
    print(f"{'packed' if packed else 'bytes'} samples in flight {nthreads}: {dt / (steps * nthreads) * 1e3:.3f} ms per sample-pass, {tot / dt / 1e9:.3f} G bins/s")
same = all(torch.equal(prep[0][2][:prep[0][3]['n_out']], p[2][:p[3]['n_out']]) for p in prep[1:])
print("all contexts identical segment ids:", same)
