"""Many process starts, each a small Bin -> Clean -> {HMM, CBS, Wavelets} flow against the oracle (tools/start_child.py).  A result that a kernel writes straight into pinned host
memory was read before it had arrived about once in eight process starts in round 4 (silently wrong Wavelets breakpoints); every such result now carries a sequence word that the host
checks (common.hpp: cvx_mail_*), and this is the test that would see the class come back: it can only be seen across process starts."""
import os
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_thirty_two_process_starts_agree_with_the_oracle():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "start_stress.sh"), "32", "4"], capture_output=True, text=True, timeout=1500)
    sys.stdout.write(r.stdout[-6000:])
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert "32 process starts, 0 mismatches" in r.stdout


@pytest.mark.gpu
def test_stale_read_counter_counts_the_looks():
    import numpy as np
    from gpu_common import get_canvas
    cv = get_canvas()
    a0 = cv.stale_reads()
    import torch
    from canvas_amd import synth
    L = 1_000_000
    b, h, m = synth.generate_chromosome(5, 0, L, 0.21)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cv.device)
    pad = lambda a: np.concatenate([a, np.zeros((-len(a)) % 64, a.dtype)])
    cap = 20000
    out = {k: torch.zeros(cap, dtype=dt, device=cv.device) for k, dt in (("chr", torch.int32), ("start", torch.int32), ("stop", torch.int32), ("gc", torch.int32), ("count", torch.float32))}
    out, per, total, bs = cv.bin_sample([dev(pad(b))], [dev(m.view(np.int64))], [dev(pad(h))], np.array([L], np.int64), [1], 100, -1, 3, out=out)
    a1 = cv.stale_reads()
    assert a1[0] > a0[0], "the bin size and the totals come back through a pinned mailbox: the look must be counted"
    assert a1[1] >= a0[1]
