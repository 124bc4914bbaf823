"""In-tree build of the HIP library (hipcc cross-compiles gfx950 without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libphantom_amd.so")
SOURCES = ["phantom_amd.hip", "numth.cpp"]
DEPS = SOURCES + ["modarith.cuh", "ntt_kernels.cuh", "ntt_fused.cuh", "rns_kernels.cuh", "keygen_kernels.cuh", "encoder_kernels.cuh", "numth.hpp",
                  os.path.join("..", "..", "include", "phantom_amd.h")]


CMD_FILE = OUT + ".cmd"        # the command line the library was built with: other flags (PHA_EXTRA_DEFINES) = stale


def _command():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    # -amdgpu-mfma-vgpr-form: bconv_mfma_kernel's accumulators land in VGPRs, where the vector ALU recombines them (no
    # v_accvgpr_read per word; it is the only kernel with MFMA instructions)
    return [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form",
            *os.environ.get("PHA_EXTRA_DEFINES", "").split(),
            *[os.path.join(CSRC, s) for s in SOURCES], "-o", OUT]


def _stale(cmd):
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    if any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS):
        return True
    # a library without its .cmd file was built elsewhere (it travels to the GPU box without one only if deleted): a GPU
    # box has no reason to rebuild, so only a recorded, different command counts
    return os.path.exists(CMD_FILE) and open(CMD_FILE).read().strip() != " ".join(cmd)


def build(force=False, verbose=True):
    """Compile the library if a source is newer than it, the flags changed, or `force`; prints the hipcc command when it
    compiles (verbose) so that a log shows what the library under test was built from."""
    cmd = _command()
    if not force and not _stale(cmd):
        return OUT
    if verbose:
        print("[build] " + " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    with open(CMD_FILE, "w") as f:
        f.write(" ".join(cmd) + "\n")
    return OUT


if __name__ == "__main__":
    build(force=True, verbose=True)
