"""Host-to-device rate of the boundary's upload helper (canvas_memcpy_h2d) for pageable and pinned host buffers: the PCIe-inclusive
figure quoted in DESIGN.md (never part of bench.py's value, which starts with the inputs resident in HBM)."""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from canvas_amd import Canvas  # noqa: E402

cv = Canvas(0)
nbytes = 1 << 30
dev = torch.empty(nbytes, dtype=torch.uint8, device="cuda:0")
for name, host in (("pageable", torch.from_numpy(np.ones(nbytes, np.uint8))), ("pinned", torch.ones(nbytes, dtype=torch.uint8).pin_memory())):
    best = 0.0
    for _ in range(4):
        t0 = time.perf_counter()
        cv._check(cv.lib.canvas_memcpy_h2d(cv.ctx, C.c_void_p(dev.data_ptr()), C.c_void_p(host.data_ptr()), C.c_int64(nbytes)))
        cv.synchronize()
        best = max(best, nbytes / (time.perf_counter() - t0) / 1e9)
    print(f"h2d {name}: {best:.1f} GB/s")
