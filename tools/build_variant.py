#!/usr/bin/env python3
"""Builds a tuning variant of the HIP library: tools/build_variant.py OUT.so -DTGNH_SPT=4 -DTGNH_PREFETCH=0 ...
Select it at run time with TGNH_LIB=OUT.so (openmm_drudenose_amd/_lib.py)."""
import os, subprocess, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
csrc = os.path.join(root, "openmm_drudenose_amd", "csrc")
out, defs = sys.argv[1], sys.argv[2:]
sys.path.insert(0, root)
from openmm_drudenose_amd.build import SOURCES
from concurrent.futures import ThreadPoolExecutor
base = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip"] + defs
objs = [out + "." + os.path.splitext(s)[0] + ".o" for s in SOURCES]
with ThreadPoolExecutor(len(SOURCES)) as pool:      # one hipcc per source file, side by side (as build.py)
    list(pool.map(lambda so: subprocess.run(base + ["-c", os.path.join(csrc, so[0]), "-o", so[1]], check=True, stderr=subprocess.DEVNULL), zip(SOURCES, objs)))
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", out], check=True)
for o in objs:
    os.remove(o)
print(out)
