"""canvas_amd — MI355X-native read-depth hot path of Canvas (CanvasBin -> CanvasClean -> CanvasPartition).

The product is canvas_amd/libcanvas_hip.so (C ABI in include/canvas_hip.h, hand-written HIP for gfx950).  This package
is the thin host-side binding used by the tests, bench.py and the drop-in tool drivers; it has NO CPU fallback: if the
library is missing or no GPU is usable it raises."""
# CanvasPartition -m CBS keeps a dozen independent kernels in flight (permutation batches of the chromosomes, generator, arc searches); the HIP runtime maps its streams onto
# GPU_MAX_HW_QUEUES hardware queues (4 unless told otherwise) and never runs more kernels at once than that.  It reads the variable at the process's first HIP call, so it is
# the HOST APPLICATION's to set (bench.py, tests/conftest.py and the three executables do): importing this package does not touch the process environment.
from .lib import Canvas, CanvasError, load_library, MODE_BINARY, MODE_TDR, CLEAN_GCNORM, CLEAN_FILTSIZE, CLEAN_OUTLIERS, CLEAN_LOCALSD, CLEAN_LOESS  # noqa: F401
