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
SOURCES = ["tgnh_host.cpp", "tgnh_kernels.hip", "tgnh_gather.hip", "tgnh_harness.hip"]
HEADERS = ["tgnh_internal.h", "tgnh_chain_device.h", "tgnh_tile_device.h", os.path.join("..", "..", "include", "drude_tgnh.h")]
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


def _object_key(src, flags):
    """what an object file depends on: its source, every header of csrc/ and the ABI header, the compiler line"""
    import hashlib
    h = hashlib.sha1(" ".join(flags).encode())
    for path in [os.path.join(CSRC, src)] + [os.path.join(CSRC, x) for x in HEADERS]:
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def build(force=False, verbose=False):
    """Compile the library if it is missing or older than its sources: one hipcc per source file, side by side, objects kept
    under build/ by content hash (an edit of the host code does not recompile the kernels), then one link."""
    if not force and not _stale():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-x", "hip"]
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, f"{os.path.splitext(src)[0]}.{_object_key(src, flags)}.o")
        if force or not os.path.exists(obj):
            cmd = flags + ["-c", os.path.join(CSRC, src), "-o", obj + ".tmp"]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.run(cmd, check=True)
            os.replace(obj + ".tmp", obj)
        return obj

    with ThreadPoolExecutor(len(SOURCES)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    for name in os.listdir(objdir):                      # objects of earlier source states
        if name.endswith(".o") and os.path.join(objdir, name) not in objs:
            os.remove(os.path.join(objdir, name))
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC"] + objs + ["-ldl", "-o", LIB]   # (RCCL is bound with dlopen at the first tgnh_rccl_* call: tgnh_host.cpp)
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
