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
LAST_BUILD_MODE = {}   # output path -> "compiled ..." | "relinked ..." | "prebuilt ..." (see _build_one); printed by __graft_entry__.build() / smoke()


def source_hash(srcs):
    """sha256 over the flags and the contents of everything a library is compiled from (its .hip files, every csrc/*.hpp, the header).
    The hash is compiled INTO the library (-DCANVAS_SRC_HASH, see ctx.hip / synth.hip) and read back from the file's bytes, so a shipped
    .so that does not correspond to the sources next to it is rebuilt no matter what its mtime says."""
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    deps = sorted(srcs) + sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + sorted(glob.glob(os.path.join(HERE, "..", "include", "*.h")))
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


def _boot_id():
    try:
        with open("/proc/sys/kernel/random/boot_id") as f:
            return f.read().strip()
    except OSError:
        import socket
        return socket.gethostname()


class _BuildLock:
    """one builder at a time per tree (concurrent smoke() ranks, pytest-xdist workers): flock on canvas_amd/.build.lock"""
    def __enter__(self):
        import fcntl
        self.f = open(os.path.join(HERE, ".build.lock"), "w")
        fcntl.flock(self.f, fcntl.LOCK_EX)
        return self

    def __exit__(self, *a):
        import fcntl
        fcntl.flock(self.f, fcntl.LOCK_UN)
        self.f.close()


def _compile_and_link(hipcc, srcs, out, extra_libs, verbose, src_hash="", use_cache=True):
    """hipcc -c per source, then an explicit link so that WE choose which libamdhip64 / librccl is recorded as DT_NEEDED
    (hipcc's own link step always resolves -lamdhip64 in /opt/rocm/lib first).  Returns (sources hipcc compiled now, sources total).

    use_cache=True: objects are cached per source under csrc/.obj (keyed by the hash of that source + every header + the flags, and by the
    library hash for the one file that embeds it) — a one-file edit rebuilds in seconds.  That directory is listed in .gpurunignore, so it never
    reaches a GPU box.  use_cache=False (build(force=True)): every source is compiled now into a private scratch directory and nothing is read
    from or left in the cache.  Either way the library is linked under a private name and moved into place with os.replace()."""
    import shutil
    import tempfile
    cflags = [f for f in FLAGS if f != "-shared"] + ['-DCANVAS_SRC_HASH="%s"' % src_hash]
    from concurrent.futures import ThreadPoolExecutor
    odir = os.path.join(CSRC, ".obj") if use_cache else tempfile.mkdtemp(prefix="canvas_obj_")
    os.makedirs(odir, exist_ok=True)
    compiled = []

    def one(sfile):
        embeds = b"CANVAS_SRC_HASH" in open(sfile, "rb").read()
        key = source_hash([sfile]) + ("-" + src_hash if embeds else "")
        o = os.path.join(odir, os.path.basename(sfile) + "." + key + ".o")
        if not os.path.exists(o):
            for stale in glob.glob(os.path.join(odir, os.path.basename(sfile) + ".*.o")):
                os.remove(stale)
            tmp = "%s.%d.tmp.o" % (o, os.getpid())
            cmd = [hipcc] + cflags + ["-c", sfile, "-o", tmp]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            os.replace(tmp, o)
            compiled.append(sfile)
        return o

    try:
        with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
            objs = list(ex.map(one, srcs))
        tl = _torch_lib_dir()
        dirs = ([tl] if tl else []) + ["/opt/rocm/lib"]
        tmp_out = "%s.%d.tmp" % (out, os.getpid())
        link = ["g++", "-shared", "-o", tmp_out] + objs
        for d in dirs:
            link += ["-L" + d, "-Wl,-rpath," + d]
        link += ["-lamdhip64"] + extra_libs + ["-lpthread", "-ldl"]
        if verbose:
            print(" ".join(link))
        subprocess.check_call(link)
        os.replace(tmp_out, out)       # a process that has the previous file mapped keeps its inode
    finally:
        if not use_cache:
            shutil.rmtree(odir, ignore_errors=True)
    return len(compiled), len(srcs)


def _record_built_here(out, src_hash):
    with open(out + ".built_here", "w") as f:
        f.write("%s %s\n" % (_boot_id(), src_hash))


def built_here(out, src_hash):
    """True when `out` was compiled from scratch (hipcc on every source) on THIS machine since it booted, from the sources whose hash is src_hash"""
    try:
        with open(out + ".built_here") as f:
            return f.read().split() == [_boot_id(), src_hash]
    except OSError:
        return False


def _build_one(hipcc, srcs, out, libs, force, verbose):
    """LAST_BUILD_MODE[out] says what happened, in words that mean what they say:
      compiled   hipcc compiled EVERY source of the library now, on this machine (no object cache involved)
      relinked   at least one object came from csrc/.obj (compiled earlier, possibly elsewhere); k of n sources were compiled now
      prebuilt   the file was there and carries the hash of the tree's sources; nothing ran"""
    h = source_hash(srcs)
    if force:
        if built_here(out, h) and embedded_hash(out) == h:      # another process of this box (a second smoke rank) has just done exactly this
            LAST_BUILD_MODE[out] = "compiled (on this machine by an earlier process: %s)" % os.path.basename(out + ".built_here")
            return
        k, n = _compile_and_link(hipcc, srcs, out, libs, verbose, h, use_cache=False)
        assert k == n
        _record_built_here(out, h)
        LAST_BUILD_MODE[out] = "compiled (hipcc ran on %d of %d sources on this machine)" % (k, n)
    elif _stale(out, srcs):
        k, n = _compile_and_link(hipcc, srcs, out, libs, verbose, h, use_cache=True)
        if k == n:
            _record_built_here(out, h)
        LAST_BUILD_MODE[out] = "compiled (hipcc ran on %d of %d sources on this machine)" % (k, n) if k == n else "relinked (%d of %d sources compiled now, the other objects from csrc/.obj)" % (k, n)
    else:
        LAST_BUILD_MODE[out] = "prebuilt (embedded source hash matches the sources; nothing compiled)"


def build(force=False, verbose=False):
    hipcc = _hipcc()
    srcs = [os.path.join(CSRC, s) for s in PRODUCT_SRC if os.path.exists(os.path.join(CSRC, s))]
    out = os.path.join(HERE, "libcanvas_hip.so")
    out2 = os.path.join(HERE, "libcanvas_synth.so")
    s2 = [os.path.join(CSRC, "synth.hip")]
    with _BuildLock():
        _build_one(hipcc, srcs, out, ["-lrccl"] if any(s.endswith("comm.hip") for s in srcs) else [], force, verbose)
        assert embedded_hash(out) == source_hash(srcs), "libcanvas_hip.so does not carry the hash of its sources"
        _build_one(hipcc, s2, out2, [], force, verbose)
        build_tools(force=force, verbose=verbose)
    return out, out2


def _tool_hash(srcs):
    h = hashlib.sha256()
    for d in list(srcs) + sorted(glob.glob(os.path.join(HERE, "..", "include", "*.h"))):
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
            tmp_out = "%s.%d.tmp" % (out, os.getpid())
            cmd = ["g++", "-O2", "-std=c++17", '-DCANVAS_SRC_HASH="%s"' % th, "-o", tmp_out, srcs[0], "-L" + HERE, "-lcanvas_hip", "-lz", "-Wl,-rpath," + HERE]
            for d in ([tl] if tl else []) + ["/opt/rocm/lib"]:
                cmd += ["-L" + d, "-Wl,-rpath," + d]
            cmd += ["-lamdhip64", "-lrccl"]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            os.replace(tmp_out, out)
        outs.append(out)
    return outs


if __name__ == "__main__":
    print(build(force=True, verbose=True))
