"""Builds libdrudetgnh_hip.so (the C-ABI of include/drude_tgnh.h) for gfx950 with hipcc.

In-tree build: the .so lands next to this file so that it travels with the repo
snapshot to the GPU box.  No runtime JIT (the reference JIT-compiles embedded
CUDA strings, platforms/cuda/src/CudaDrudeTGNHKernels.cpp:258-279; here the
kernels are compiled offline for one architecture).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdrudetgnh_hip.so")
SOURCES = ["tgnh_host.cpp", "tgnh_kernels.hip", "tgnh_harness.hip"]
HEADERS = ["tgnh_internal.h", "tgnh_chain_device.h", os.path.join("..", "..", "include", "drude_tgnh.h")]
ARCH = "gfx950"


def source_sha():
    """sha1 over the sources the library is built from, in a fixed order: what profiles/ and bench.py stamp their
    numbers with, so a profile is only ever quoted for the binary it was taken from."""
    import hashlib
    h = hashlib.sha1()
    for name in sorted(os.listdir(CSRC)):
        path = os.path.join(CSRC, name)
        if os.path.isfile(path) and name.endswith((".cpp", ".hip", ".h")):
            h.update(name.encode())
            h.update(open(path, "rb").read())
    h.update(open(os.path.join(HERE, "..", "include", "drude_tgnh.h"), "rb").read())
    return h.hexdigest()[:16]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile the library if it is missing or older than its sources."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function", "-x", "hip"]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-ldl", "-o", LIB]       # (RCCL is bound with dlopen at the first tgnh_rccl_* call: tgnh_host.cpp)
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
