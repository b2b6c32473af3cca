"""Build the gfx950 shared library (C ABI of include/unetrir.h) in-tree with hipcc."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libunetrir.so")
SOURCES = ["igemm.hip", "api.hip", "elementwise.hip", "head.hip", "head_mfma.hip", "wgrad3x3.hip", "wgrad3x3r.hip", "wgrad3x3g.hip", "igemm_bf16.hip", "conv3x3.hip", "conv3x3r.hip", "conv3x3g.hip", "conv3x3h.hip", "upconv3x3.hip", "upconv3x3g.hip", "features.hip", "dense.hip", "stem3x3.hip"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [
        os.path.join(os.path.dirname(HERE), "include", "unetrir.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """hipcc cross-compiles for gfx950 without a GPU present."""
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
           "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
