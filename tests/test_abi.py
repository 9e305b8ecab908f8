"""CPU-side checks of the C-ABI library: it loads here (no GPU) and exports every symbol include/canvas_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from canvas_amd import build
    so, _ = build.build()
    return ctypes.CDLL(so)


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "canvas_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(canvas_\w+)\s*\(", hdr))
    assert len(names) >= 20
    from canvas_amd.lib import ABI_SYMBOLS
    assert names == set(ABI_SYMBOLS), names ^ set(ABI_SYMBOLS)
    for n in names:
        assert hasattr(lib, n), n


def test_nothing_undeclared_is_exported():
    """the other direction: every canvas_* symbol the library exports is declared in the header (no private entry points)"""
    import shutil
    import subprocess
    from canvas_amd import build
    so, _ = build.build()
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.run([nm, "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if line.split() and line.split()[-1].startswith("canvas_")}
    from canvas_amd.lib import ABI_SYMBOLS
    assert exported == set(ABI_SYMBOLS), exported ^ set(ABI_SYMBOLS)


def test_host_scalar_entry_points_without_gpu(lib):
    import numpy as np
    rates = np.array([0.2, 0.1, 0.4, 0.3], np.float64)
    lib.canvas_bin_size_from_rates.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32]
    assert lib.canvas_bin_size_from_rates(rates.ctypes.data, 4, 100) == int(100 / 0.25)
    lens = np.array([1000, 2500], np.int64)
    lib.canvas_bin_count_upper_bound.restype = ctypes.c_int64
    lib.canvas_bin_count_upper_bound.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32]
    assert lib.canvas_bin_count_upper_bound(2, lens.ctypes.data, 100) == 35
    lib.canvas_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.canvas_version()


def test_no_cpu_fallback():
    import torch
    from canvas_amd import Canvas, CanvasError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(CanvasError):
        Canvas(0)


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "canvas_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_lib" not in txt and "libcanvas_oracle" not in txt and "oracle/" not in txt.replace("the oracle", ""), f


def test_split_overlapping_segments_reference_cases(lib):
    """canvas_split_overlapping (host scalar entry point) against the reference's own 9 known-answer cases
    (CanvasTest/CanvasPartition/GenomeSegmentationResultsTests.cs:14-248)"""
    import json
    import numpy as np
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "split_overlapping_cases.json")))
    for c in cases:
        for chrom, exp in c["expected"].items():
            st = [np.array([s[0] for s in smp[chrom]], np.uint32) for smp in c["samples"]]
            en = [np.array([s[1] for s in smp[chrom]], np.uint32) for smp in c["samples"]]
            n = len(st)
            P = ctypes.c_void_p * n
            nseg = np.array([len(s) for s in st], np.int32)
            os_ = np.zeros(64, np.uint32); oe = np.zeros(64, np.uint32); nout = ctypes.c_int32(0)
            rc = lib.canvas_split_overlapping(n, P(*[a.ctypes.data for a in st]), P(*[a.ctypes.data for a in en]), nseg.ctypes.data_as(ctypes.c_void_p),
                                              os_.ctypes.data_as(ctypes.c_void_p), oe.ctypes.data_as(ctypes.c_void_p), 64, ctypes.byref(nout))
            assert rc == 0
            got = [[int(a), int(b)] for a, b in zip(os_[:nout.value], oe[:nout.value])]
            assert got == exp, c["name"]
