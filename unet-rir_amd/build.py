"""Build the gfx950 shared library (C ABI of include/unetrir.h) in-tree with hipcc.

One object per translation unit (compiled in parallel, only those whose source or headers changed), then one link:
editing a single kernel rebuilds in seconds instead of a minute."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libunetrir.so")
SOURCES = ["igemm.hip", "api.hip", "elementwise.hip", "head.hip", "head_mfma.hip", "wgrad3x3.hip", "wgrad3x3r.hip", "wgrad3x3g.hip", "wgrad3x3d.hip", "igemm_bf16.hip", "conv3x3.hip", "conv3x3r.hip", "conv3x3g.hip", "conv3x3h.hip", "conv3x3s.hip", "conv3x3d.hip", "conv3x3p.hip", "upconv3x3.hip", "upconv3x3g.hip", "upconv3x3q.hip", "features.hip", "dense.hip", "stem3x3.hip", "pw1x1.hip", "igemm2_bf16.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wno-unused-value"]


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [
        os.path.join(os.path.dirname(HERE), "include", "unetrir.h")]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + _headers()
    return _stale(LIB, deps)


def build(force=False, verbose=False, extra_flags=(), lib=LIB, obj_dir=OBJ):
    """hipcc cross-compiles for gfx950 without a GPU present."""
    if not force and lib == LIB and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(obj_dir, exist_ok=True)
    hdrs = _headers()
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(obj_dir, s[:-4] + ".o")
        if force or _stale(obj, [src] + hdrs):
            jobs.append([hipcc] + FLAGS + list(extra_flags) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(run, jobs))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] +
        [os.path.join(obj_dir, s[:-4] + ".o") for s in SOURCES])
    return lib


ABL_LIB = os.path.join(HERE, "libunetrir_abl.so")


def build_ablations(verbose=False):
    """The timing-ablation build for scripts/ (kernels with a stage switched off; results invalid).  Not loaded by the package
    unless a script asks for it with _lib.use_library(build.ABL_LIB)."""
    return build(force=False, verbose=verbose, extra_flags=["-DUNETRIR_ABLATIONS"], lib=ABL_LIB, obj_dir=os.path.join(HERE, "build_abl"))


if __name__ == "__main__":
    if "--ablations" in sys.argv:
        print(build_ablations(verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
