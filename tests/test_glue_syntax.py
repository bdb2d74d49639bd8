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


GLUE_DIR = os.path.join(ROOT, "openmm_drudenose_amd", "csrc", "openmm_glue")
SHIM = ["-I", os.path.join(ROOT, "tests", "cpp", "openmm_shim"), "-I", os.path.join(ROOT, "include")]


def test_the_glue_with_the_thermostat_checkpoint_parses():
    """-DDRUDETGNH_THERMOSTAT_CHECKPOINT: the kernel registers a reader with the serialization proxy and takes a parked state"""
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror=return-type", "-DDRUDETGNH_THERMOSTAT_CHECKPOINT",
           "-I", os.path.join(GLUE_DIR, "serialization", "include")] + SHIM + [GLUE]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]


@pytest.mark.parametrize("source", ["DrudeTGNHIntegratorProxy.cpp", "DrudeTGNHSerializationProxyRegistration.cpp"])
def test_the_serialization_proxy_parses(source):
    """csrc/openmm_glue/serialization: the counterpart of the reference's serialization/src (DrudeTGNHIntegratorProxy.cpp:40-67,
    DrudeTGNHSerializationProxyRegistration.cpp:52-65) through the same front end"""
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror=return-type",
           "-I", os.path.join(GLUE_DIR, "serialization", "include")] + SHIM + [os.path.join(GLUE_DIR, "serialization", "src", source)]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]


def test_the_proxy_and_the_python_writer_agree_on_the_attributes():
    """The C++ proxy writes the reference's nine properties (serialization/src/DrudeTGNHIntegratorProxy.cpp:43-55: names and
    version) plus maxDrudeDistance, useCOMTempGroup and the TempGroups table -- exactly the attribute set of the Python writer
    (openmm_drudenose_amd/serialization.py), whose XML round trip tests/test_abi.py checks: files of one are read by the other."""
    import re
    import sys
    import xml.etree.ElementTree as ET
    sys.path.insert(0, ROOT)
    from openmm_drudenose_amd import DrudeTGNHIntegrator
    from openmm_drudenose_amd import serialization
    text = open(os.path.join(GLUE_DIR, "serialization", "src", "DrudeTGNHIntegratorProxy.cpp")).read()
    body = text[text.index("void DrudeTGNHIntegratorProxy::serialize"):text.index("DrudeTGNHThermostatState state;")]
    written = set(re.findall(r'node\.set(?:Int|Double)Property\("(\w+)"', body))
    tables = text[text.index("const RealProperty realProperties[]"):text.index("const int proxyVersion")]      # (the reference's nine go through two tables)
    written |= set(re.findall(r'\{"(\w+)", &DrudeTGNHIntegrator::get', tables))
    reference_nine = {"stepSize", "constraintTolerance", "temperature", "couplingTime", "drudeTemperature", "drudeCouplingTime",
                      "drudeStepsPerRealStep", "numNHChains", "useDrudeNHChains"}
    assert written == reference_nine | {"version", "maxDrudeDistance", "useCOMTempGroup"}, written
    it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001, 20, 3, True, True)
    it.addTempGroup(); it.addTempGroup()
    for g in (0, 1, 1):
        it.addParticleTempGroup(g)
    node = ET.fromstring(serialization.serialize(it))
    assert set(node.attrib) - {"type"} == written, (set(node.attrib), written)
    assert [c.tag for c in node] == ["TempGroups"] and '"TempGroups"' in body and 'setIntProperty("count"' in body
    # ... and what deserialize asks for is what serialize wrote
    back = text[text.index("void* DrudeTGNHIntegratorProxy::deserialize"):]
    asked = set(re.findall(r'node\.(?:get(?:Int|Double|Bool)Property|hasProperty)\("(\w+)"', back))
    assert asked == written, (asked, written)


def test_the_plugin_entry_points_come_out_unmangled_and_visible(tmp_path):
    """What OpenMM dlsym()s after dlopen of every library in lib/plugins (platforms/cuda/src/CudaDrudeTGNHKernelFactory.cpp:37-59):
    registerPlatforms, registerKernelFactories and the static-link entry registerDrudeTGNHHipKernelFactories must be C symbols with
    default visibility in the object the glue compiles to -- also when the plugin is built with -fvisibility=hidden, as OpenMM's
    own plugins are.  Compiled against the declaration shims (pins nothing about OpenMM; catches a linkage or visibility slip).
    The serialization library's registration entry point likewise."""
    obj = str(tmp_path / "glue.o")
    for src, want in ((GLUE, ["registerPlatforms", "registerKernelFactories", "registerDrudeTGNHHipKernelFactories"]),
                      (os.path.join(GLUE_DIR, "serialization", "src", "DrudeTGNHSerializationProxyRegistration.cpp"),
                       ["registerDrudeTGNHSerializationProxies"])):
        cmd = ["g++", "-std=c++17", "-c", "-fPIC", "-fvisibility=hidden", "-DTGNH_SHIM_EXPORT_DEFAULT",
               "-I", os.path.join(GLUE_DIR, "serialization", "include")] + SHIM + [src, "-o", obj]
        p = subprocess.run(cmd, capture_output=True, text=True)
        assert p.returncode == 0, p.stderr[-3000:]
        so = str(tmp_path / "glue.so")
        p = subprocess.run(["g++", "-shared", "-Wl,--unresolved-symbols=ignore-all", obj, "-o", so], capture_output=True, text=True)
        assert p.returncode == 0, p.stderr[-3000:]
        syms = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True).stdout
        exported = {ln.split()[-1]: ln.split()[-2] for ln in syms.splitlines() if len(ln.split()) >= 3}
        for name in want:
            assert exported.get(name) == "T", (name, syms)           # unmangled, in the dynamic symbol table, a function


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
