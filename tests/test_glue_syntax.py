"""The OpenMM-HIP glue through a C++ front end (CPU).  OpenMM is absent from this image, so the plugin cannot be BUILT here; but
its 200 lines can be PARSED: g++ -fsyntax-only against tests/cpp/openmm_shim -- declarations-only stand-ins for exactly the symbols
INTEGRATION.md section 6 lists ("NOT OpenMM, pins nothing": README.md there).  What this catches: typos, missing includes, calls that
do not fit the signatures the hand-over checklist assumes, and the C ABI's own header used from C++ (include/drude_tgnh.h is
the real one).  What it cannot tell: whether a real OpenMM >= 8.2 has those signatures -- SURVEY 8f-3 stays open until somebody
builds against one.  Reference counterparts: platforms/cuda/src/CudaDrudeTGNHKernelFactory.cpp:37-66, CudaDrudeTGNHKernels.cpp:284-408."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GLUE = os.path.join(ROOT, "openmm_drudenose_amd", "csrc", "openmm_glue", "platforms", "hip", "src", "HipDrudeTGNHKernels.cpp")


@pytest.mark.parametrize("defines", [[], ["-DDRUDETGNH_RESIDENT_STEP"], ["-DDRUDETGNH_TRUST_STATE_CHANGED"],
                                     ["-DDRUDETGNH_RESIDENT_STEP", "-DDRUDETGNH_TRUST_STATE_CHANGED"]])
def test_the_glue_parses(defines):
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror=return-type", "-I", os.path.join(ROOT, "tests", "cpp", "openmm_shim"),
           "-I", os.path.join(ROOT, "include")] + defines + [GLUE]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]


def test_the_shim_says_what_it_is():
    """every file of the stand-in directory carries the label (nobody should mistake it for OpenMM or for a reference build)"""
    shim = os.path.join(ROOT, "tests", "cpp", "openmm_shim")
    n = 0
    for d, _, files in os.walk(shim):
        for f in files:
            text = open(os.path.join(d, f)).read()
            assert "NOT OpenMM" in text, os.path.join(d, f)
            n += 1
    assert n >= 10
