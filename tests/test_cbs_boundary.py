"""The sequential stopping boundary of CBS (GetBoundary.cs:19-157) as the PRODUCT computes it — on its host thread pool, every scan predicted by bisection and then checked with the
scan's own evaluations — against the oracle's sequential restatement, entry for entry.  Host-only: runs without a GPU."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from canvas_amd.lib import load_library


# (the oracle wrapper holds 65 536 entries: floor(nperm alpha) <= 360)
@pytest.mark.parametrize("nperm,alpha", [(10000, 0.01), (10000, 0.02), (5000, 0.01), (2500, 0.02), (1000, 0.01), (500, 0.05), (200, 0.01)])
def test_boundary_table_equals_the_oracles(nperm, alpha):
    lib = load_library()
    lib.canvas_cbs_boundary.restype = C.c_int64
    out = np.zeros(1 << 17, np.uint32)
    n = lib.canvas_cbs_boundary(C.c_uint32(nperm), C.c_double(alpha), out.ctypes.data_as(C.c_void_p), C.c_int64(len(out)))
    exp = O.cbs_boundary(nperm, alpha)
    assert n == len(exp) and n == (int(np.floor(nperm * alpha)) + 1) * (int(np.floor(nperm * alpha)) + 2) // 2
    assert (out[:n] == exp).all()


def test_the_compiled_in_default_table_equals_a_fresh_computation(monkeypatch):
    """canvas_amd/csrc/cbs_boundary_default.hpp (tools/gen_boundary_table.py) holds the table for CanvasPartition's defaults (10000 permutations, alpha 0.01): what the library
    hands out by default must equal what it computes when told not to use the constant, and the oracle's."""
    lib = load_library()
    lib.canvas_cbs_boundary.restype = C.c_int64
    a = np.zeros(1 << 14, np.uint32); b = np.zeros(1 << 14, np.uint32)
    monkeypatch.delenv("CANVAS_CBS_NO_EMBEDDED_BOUNDARY", raising=False)
    na = lib.canvas_cbs_boundary(C.c_uint32(10000), C.c_double(0.01), a.ctypes.data_as(C.c_void_p), C.c_int64(len(a)))
    monkeypatch.setenv("CANVAS_CBS_NO_EMBEDDED_BOUNDARY", "1")
    nb = lib.canvas_cbs_boundary(C.c_uint32(10000), C.c_double(0.01), b.ctypes.data_as(C.c_void_p), C.c_int64(len(b)))
    assert na == nb == 5151 and (a[:na] == b[:nb]).all()
    assert (a[:na] == O.cbs_boundary(10000, 0.01)).all()
    src = open(__import__("os").path.join(__import__("os").path.dirname(__file__), "..", "canvas_amd", "csrc", "cbs_boundary_default.hpp")).read()
    assert "kCbsDefaultBoundary[5151]" in src
