"""Builds libsrk.so (hand-written HIP kernels + the C ABI of include/srk.h) for gfx950.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container as well as on
the MI355X box.  Objects go to csrc/build/ (git-ignored), the shared library next to this file
(git-ignored, but it travels with the gpurun snapshot).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsrk.so")
SOURCES = ["api.hip", "elementwise.hip", "loss_optim.hip", "conv_generic.hip", "conv_mfma.hip", "conv_mfma_bf16.hip", "conv_bfd.hip", "conv_tapn.hip", "conv_rown.hip", "conv_bfw.hip", "conv_bfr.hip", "conv_rowsw.hip", "conv_res2.hip", "conv_c64.hip",
           "conv_wgrad_mfma.hip", "conv_wgrad_bf16.hip", "bn_linear.hip", "resize_pil.hip", "prepost.hip"]
HEADERS = ["srk_common.h", "conv_problem.h", "conv_tile.h", "pack_items.h", "conv_bfw.h", "bf16_frag.h", os.path.join("..", "..", "include", "srk.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mcode-object-version=5",
         "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]
# SRK_BUILD_EXPERIMENTS=1: compile the experiment switches in (SRK_DBG ablation bits, forced tiles / block shapes --
# csrc/srk_common.h SRK_EXP_INT); the release library carries their defaults as constants.  The stamp file records which
# flavour the objects in csrc/build are, so switching flavours rebuilds.
EXPERIMENTS = os.environ.get("SRK_BUILD_EXPERIMENTS", "0") == "1"
if EXPERIMENTS:
    FLAGS = FLAGS + ["-DSRK_EXPERIMENTS"]
STAMP = os.path.join(CSRC, "build", "flavour.txt")


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _mtime(p):
    return os.path.getmtime(p) if os.path.exists(p) else 0.0


def _flavour():
    return "experiments" if EXPERIMENTS else "release"


def _stamp_ok():
    try:
        with open(STAMP) as fh:
            return fh.read().strip() == _flavour()
    except OSError:
        # no stamp: a library somebody built elsewhere (it travelled with a gpurun snapshot) is taken as it is
        return not EXPERIMENTS


def needs_build():
    newest = max(_mtime(os.path.join(CSRC, f)) for f in SOURCES + HEADERS)
    return _mtime(LIB) < max(newest, _mtime(__file__)) or not _stamp_ok()


def build(force=False, verbose=True):
    """Compile every HIP source and link libsrk.so. Returns the library path."""
    if not force and not needs_build():
        return LIB
    bdir = os.path.join(CSRC, "build")
    os.makedirs(bdir, exist_ok=True)
    if not _stamp_ok():
        force = True
    hipcc = _hipcc()
    hdr_time = max(_mtime(os.path.join(CSRC, h)) for h in HEADERS)

    def compile_one(src):
        obj = os.path.join(bdir, src.replace(".hip", ".o"))
        if not force and _mtime(obj) > max(_mtime(os.path.join(CSRC, src)), hdr_time, _mtime(__file__)):
            return obj
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, " ".join(cmd), r.stderr))
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    tmp = LIB + ".tmp"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
    os.replace(tmp, LIB)
    with open(STAMP, "w") as fh:
        fh.write(_flavour() + "\n")
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
