"""Builds the in-tree HIP libraries for gfx950 with hipcc (cross-compiles without a GPU).
  canvas_amd/libcanvas_hip.so    the product (C ABI of include/canvas_hip.h)
  canvas_amd/libcanvas_synth.so  synthetic-input generator used by bench/tests only
"""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unused-result"]
PRODUCT_SRC = ["ctx.hip", "bin.hip", "clean.hip", "hmm.hip", "cbs.hip", "comm.hip"]


def _hipcc():
    for c in ("hipcc", "/opt/rocm/bin/hipcc"):
        if subprocess.call(["which", c], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) == 0 or os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def _stale(out, srcs):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    deps = list(srcs) + glob.glob(os.path.join(CSRC, "*.hpp")) + [os.path.join(HERE, "..", "include", "canvas_hip.h")]
    return any(os.path.getmtime(s) > t for s in deps)


def build(force=False, verbose=False):
    hipcc = _hipcc()
    srcs = [os.path.join(CSRC, s) for s in PRODUCT_SRC if os.path.exists(os.path.join(CSRC, s))]
    out = os.path.join(HERE, "libcanvas_hip.so")
    if force or _stale(out, srcs):
        cmd = [hipcc] + FLAGS + ["-o", out] + srcs
        if any(s.endswith("comm.hip") for s in srcs):
            cmd += ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    out2 = os.path.join(HERE, "libcanvas_synth.so")
    s2 = [os.path.join(CSRC, "synth.hip")]
    if force or _stale(out2, s2):
        subprocess.check_call([hipcc] + FLAGS + ["-o", out2] + s2)
    return out, out2


if __name__ == "__main__":
    print(build(force=True, verbose=True))
