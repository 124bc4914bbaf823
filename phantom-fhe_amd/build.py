"""In-tree build of the HIP library (hipcc cross-compiles gfx950 without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libphantom_amd.so")
SOURCES = ["phantom_amd.hip", "numth.cpp"]
DEPS = SOURCES + ["modarith.cuh", "ntt_kernels.cuh", "rns_kernels.cuh", "keygen_kernels.cuh", "encoder_kernels.cuh", "numth.hpp",
                  os.path.join("..", "..", "include", "phantom_amd.h")]


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not _stale():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", *os.environ.get("PHA_EXTRA_DEFINES", "").split(),
           *[os.path.join(CSRC, s) for s in SOURCES], "-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    return OUT


if __name__ == "__main__":
    build(force=True, verbose=True)
