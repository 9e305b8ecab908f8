"""canvas_amd — MI355X-native read-depth hot path of Canvas (CanvasBin -> CanvasClean -> CanvasPartition).

The product is canvas_amd/libcanvas_hip.so (C ABI in include/canvas_hip.h, hand-written HIP for gfx950).  This package
is the thin host-side binding used by the tests, bench.py and the drop-in tool drivers; it has NO CPU fallback: if the
library is missing or no GPU is usable it raises."""
from .lib import Canvas, CanvasError, load_library, MODE_BINARY, MODE_TDR, CLEAN_GCNORM, CLEAN_FILTSIZE, CLEAN_OUTLIERS, CLEAN_LOCALSD, CLEAN_LOESS  # noqa: F401
