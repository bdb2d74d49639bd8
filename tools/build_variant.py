#!/usr/bin/env python3
"""Builds a tuning variant of the HIP library: tools/build_variant.py OUT.so -DTGNH_SPT=4 -DTGNH_PREFETCH=0 ...
Select it at run time with TGNH_LIB=OUT.so (openmm_drudenose_amd/_lib.py)."""
import os, subprocess, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
csrc = os.path.join(root, "openmm_drudenose_amd", "csrc")
out, defs = sys.argv[1], sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip",
       os.path.join(csrc, "tgnh_host.cpp"), os.path.join(csrc, "tgnh_kernels.hip"), os.path.join(csrc, "tgnh_harness.hip"), "-ldl", "-o", out] + defs
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
print(out)
