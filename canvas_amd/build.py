"""Builds the in-tree HIP libraries for gfx950 with hipcc (cross-compiles without a GPU).
  canvas_amd/libcanvas_hip.so    the product (C ABI of include/canvas_hip.h)
  canvas_amd/libcanvas_synth.so  synthetic-input generator used by bench/tests only
"""
import glob
import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unused-result"]
PRODUCT_SRC = ["ctx.hip", "prep.hip", "bin.hip", "clean.hip", "hmm.hip", "cbs.hip", "wavelets.hip", "evenness.hip", "normalize.hip", "pipeline.hip", "sharded.hip", "comm.hip"]


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


HASH_MARKER = b"CANVAS_SRC_HASH="
LAST_BUILD_MODE = {}   # output path -> "compiled" | "reused (source hash matches)"; printed by __graft_entry__.smoke()


def source_hash(srcs):
    """sha256 over the flags and the contents of everything a library is compiled from (its .hip files, every csrc/*.hpp, the header).
    The hash is compiled INTO the library (-DCANVAS_SRC_HASH, see ctx.hip / synth.hip) and read back from the file's bytes, so a shipped
    .so that does not correspond to the sources next to it is rebuilt no matter what its mtime says."""
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    deps = sorted(srcs) + sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + [os.path.join(HERE, "..", "include", "canvas_hip.h")]
    for d in deps:
        h.update(os.path.basename(d).encode() + b"\0")
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:32]


def embedded_hash(path):
    """the source hash a built file carries (None if absent)"""
    if not os.path.exists(path):
        return None
    with open(path, "rb") as f:
        blob = f.read()
    i = blob.find(HASH_MARKER)
    if i < 0:
        return None
    j = i + len(HASH_MARKER)
    return blob[j:j + 32].decode("ascii", "replace")


def _stale(out, srcs):
    return embedded_hash(out) != source_hash(srcs)


def _compile_and_link(hipcc, srcs, out, extra_libs, verbose, src_hash=""):
    """hipcc -c per source, then an explicit link so that WE choose which libamdhip64 / librccl is recorded as DT_NEEDED
    (hipcc's own link step always resolves -lamdhip64 in /opt/rocm/lib first)."""
    cflags = [f for f in FLAGS if f != "-shared"] + ['-DCANVAS_SRC_HASH="%s"' % src_hash]
    # objects are cached per source under csrc/.obj (keyed by the hash of that source + every header + the flags, and by the library
    # hash for the one file that embeds it), and the sources compile in parallel: a one-file edit rebuilds in seconds
    from concurrent.futures import ThreadPoolExecutor
    odir = os.path.join(CSRC, ".obj")
    os.makedirs(odir, exist_ok=True)

    def one(sfile):
        embeds = b"CANVAS_SRC_HASH" in open(sfile, "rb").read()
        key = source_hash([sfile]) + ("-" + src_hash if embeds else "")
        o = os.path.join(odir, os.path.basename(sfile) + "." + key + ".o")
        if not os.path.exists(o):
            for stale in glob.glob(os.path.join(odir, os.path.basename(sfile) + ".*.o")):
                os.remove(stale)
            cmd = [hipcc] + cflags + ["-c", sfile, "-o", o + ".tmp.o"]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            os.replace(o + ".tmp.o", o)
        return o

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(one, srcs))
    tl = _torch_lib_dir()
    dirs = ([tl] if tl else []) + ["/opt/rocm/lib"]
    link = ["g++", "-shared", "-o", out] + objs
    for d in dirs:
        link += ["-L" + d, "-Wl,-rpath," + d]
    link += ["-lamdhip64"] + extra_libs + ["-lpthread", "-ldl"]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)


def build(force=False, verbose=False):
    hipcc = _hipcc()
    srcs = [os.path.join(CSRC, s) for s in PRODUCT_SRC if os.path.exists(os.path.join(CSRC, s))]
    out = os.path.join(HERE, "libcanvas_hip.so")
    if force or _stale(out, srcs):
        _compile_and_link(hipcc, srcs, out, ["-lrccl"] if any(s.endswith("comm.hip") for s in srcs) else [], verbose, source_hash(srcs))
        LAST_BUILD_MODE[out] = "compiled"
    else:
        LAST_BUILD_MODE[out] = "reused (embedded source hash matches the sources)"
    assert embedded_hash(out) == source_hash(srcs), "libcanvas_hip.so does not carry the hash of its sources"
    out2 = os.path.join(HERE, "libcanvas_synth.so")
    s2 = [os.path.join(CSRC, "synth.hip")]
    if force or _stale(out2, s2):
        _compile_and_link(hipcc, s2, out2, [], verbose, source_hash(s2))
        LAST_BUILD_MODE[out2] = "compiled"
    else:
        LAST_BUILD_MODE[out2] = "reused (embedded source hash matches the sources)"
    build_tools(force=force, verbose=verbose)
    return out, out2


def _tool_hash(srcs):
    h = hashlib.sha256()
    for d in list(srcs) + [os.path.join(HERE, "..", "include", "canvas_hip.h")]:
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:32]


def build_tools(force=False, verbose=False):
    """drop-in tool drivers (plain C++ over the C ABI + zlib): canvas_amd/bin/CanvasBin, CanvasClean, CanvasPartition"""
    tdir = os.path.join(HERE, "tools")
    bdir = os.path.join(HERE, "bin")
    os.makedirs(bdir, exist_ok=True)
    tl = _torch_lib_dir()
    outs = []
    for name, src in (("CanvasBin", "canvas_bin_main.cpp"), ("CanvasClean", "canvas_clean_main.cpp"), ("CanvasPartition", "canvas_partition_main.cpp")):
        out = os.path.join(bdir, name)
        srcs = [os.path.join(tdir, src), os.path.join(tdir, "tool_common.hpp"), os.path.join(tdir, "protobuf_dat.hpp"), os.path.join(tdir, "fast_io.hpp")]
        th = _tool_hash(srcs)
        if force or embedded_hash(out) != th:
            cmd = ["g++", "-O2", "-std=c++17", '-DCANVAS_SRC_HASH="%s"' % th, "-o", out, srcs[0], "-L" + HERE, "-lcanvas_hip", "-lz", "-Wl,-rpath," + HERE]
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
