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
PRODUCT_SRC = ["ctx.hip", "prep.hip", "bin.hip", "clean.hip", "hmm.hip", "cbs.hip", "wavelets.hip", "normalize.hip", "pipeline.hip", "comm.hip"]


def _hipcc():
    for c in ("hipcc", "/opt/rocm/bin/hipcc"):
        if subprocess.call(["which", c], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) == 0 or os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def _torch_lib_dir():
    """When the library is hosted in a torch process it must use the SAME HIP runtime as torch (torch wheels bundle their own
    libamdhip64.so / librccl.so): link against those so that one runtime is loaded.  Without torch: /opt/rocm (hipcc default)."""
    try:
        import torch
        d = os.path.join(os.path.dirname(torch.__file__), "lib")
        if os.path.exists(os.path.join(d, "libamdhip64.so")):
            return d
    except Exception:
        pass
    return None


def _stale(out, srcs):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    deps = list(srcs) + glob.glob(os.path.join(CSRC, "*.hpp")) + [os.path.join(HERE, "..", "include", "canvas_hip.h")]
    return any(os.path.getmtime(s) > t for s in deps)


def _compile_and_link(hipcc, srcs, out, extra_libs, verbose):
    """hipcc -c per source, then an explicit link so that WE choose which libamdhip64 / librccl is recorded as DT_NEEDED
    (hipcc's own link step always resolves -lamdhip64 in /opt/rocm/lib first)."""
    cflags = [f for f in FLAGS if f != "-shared"]
    objs = []
    for sfile in srcs:
        o = os.path.join(CSRC, "." + os.path.basename(sfile) + ".o")
        cmd = [hipcc] + cflags + ["-c", sfile, "-o", o]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(o)
    tl = _torch_lib_dir()
    dirs = ([tl] if tl else []) + ["/opt/rocm/lib"]
    link = ["g++", "-shared", "-o", out] + objs
    for d in dirs:
        link += ["-L" + d, "-Wl,-rpath," + d]
    link += ["-lamdhip64"] + extra_libs + ["-lpthread", "-ldl"]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    for o in objs:
        os.remove(o)


def build(force=False, verbose=False):
    hipcc = _hipcc()
    srcs = [os.path.join(CSRC, s) for s in PRODUCT_SRC if os.path.exists(os.path.join(CSRC, s))]
    out = os.path.join(HERE, "libcanvas_hip.so")
    if force or _stale(out, srcs):
        _compile_and_link(hipcc, srcs, out, ["-lrccl"] if any(s.endswith("comm.hip") for s in srcs) else [], verbose)
    out2 = os.path.join(HERE, "libcanvas_synth.so")
    s2 = [os.path.join(CSRC, "synth.hip")]
    if force or _stale(out2, s2):
        _compile_and_link(hipcc, s2, out2, [], verbose)
    build_tools(force=force, verbose=verbose)
    return out, out2


def build_tools(force=False, verbose=False):
    """drop-in tool drivers (plain C++ over the C ABI + zlib): canvas_amd/bin/CanvasBin, CanvasClean, CanvasPartition"""
    tdir = os.path.join(HERE, "tools")
    bdir = os.path.join(HERE, "bin")
    os.makedirs(bdir, exist_ok=True)
    tl = _torch_lib_dir()
    outs = []
    for name, src in (("CanvasBin", "canvas_bin_main.cpp"), ("CanvasClean", "canvas_clean_main.cpp"), ("CanvasPartition", "canvas_partition_main.cpp")):
        out = os.path.join(bdir, name)
        srcs = [os.path.join(tdir, src), os.path.join(tdir, "tool_common.hpp")]
        if force or not os.path.exists(out) or any(os.path.getmtime(x) > os.path.getmtime(out) for x in srcs + [os.path.join(HERE, "libcanvas_hip.so")]):
            cmd = ["g++", "-O2", "-std=c++17", "-o", out, srcs[0], "-L" + HERE, "-lcanvas_hip", "-lz", "-Wl,-rpath," + HERE]
            for d in ([tl] if tl else []) + ["/opt/rocm/lib"]:
                cmd += ["-L" + d, "-Wl,-rpath," + d]
            cmd += ["-lamdhip64", "-lrccl"]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        outs.append(out)
    return outs


if __name__ == "__main__":
    print(build(force=True, verbose=True))
