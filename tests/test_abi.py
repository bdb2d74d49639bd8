"""CPU tests: the C-ABI library loads and exports every symbol include/drude_tgnh.h declares,
argument validation fails loudly, and the host-side mirror keeps the reference's API behaviour.
No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from openmm_drudenose_amd import _lib, synth
from openmm_drudenose_amd.drudetgnhplugin import DrudeTGNHIntegrator, TgnhError
from openmm_drudenose_amd.system import shard_bounds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "drude_tgnh.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tgnh_[a-z_0-9]+)\s*\(", text)) - {"tgnh_allreduce_fn"})


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/drude_tgnh.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), "ctypes binding and header disagree"
    assert lib.tgnh_abi_version() == 1


def test_create_rejects_bad_desc():
    lib = _lib.load()
    d = _lib.TgnhDesc()
    h = C.c_void_p()
    assert lib.tgnh_create(C.byref(d), C.byref(h)) == _lib.ERR_ARG       # struct_size 0
    assert b"size mismatch" in lib.tgnh_last_error()
    d.struct_size = C.sizeof(d)
    assert lib.tgnh_create(C.byref(d), C.byref(h)) == _lib.ERR_ARG
    assert lib.tgnh_destroy(None) == _lib.ERR_ARG


def test_integrator_api_mirror():
    # python/drudetgnhplugin.i:62 default useDrudeNHChains=True; DrudeTGNHIntegrator.cpp:58 tolerance 1e-5
    it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001)
    assert it.getDrudeStepsPerRealStep() == 20 and it.getNumNHChains() == 1
    assert it.getUseDrudeNHChains() == 1 and it.getUseCOMTempGroup() == 1
    assert it.getConstraintTolerance() == 1e-5 and it.getMaxDrudeDistance() == 0
    with pytest.raises(TgnhError, match="Distance cannot be negative"):      # DrudeTGNHIntegrator.cpp:97-100
        it.setMaxDrudeDistance(-1.0)
    # DrudeTGNHIntegrator.cpp:61-70: addTempGroup returns the running index; addParticleTempGroup is range-checked
    assert it.getNumTempGroups() == 0
    with pytest.raises(TgnhError):
        it.addParticleTempGroup(0)
    assert it.addTempGroup() == 0 and it.addTempGroup() == 1
    assert it.addParticleTempGroup(1) == 0 and it.getParticleTempGroup(0) == 1
    it.setParticleTempGroup(0, 0)
    assert it.getParticleTempGroup(0) == 0
    with pytest.raises(TgnhError):
        it.setParticleTempGroup(3, 0)
    with pytest.raises(TgnhError, match="does not match"):                  # :133-134
        it._resolve_groups(5)
    with pytest.raises(TgnhError, match="not bound to a context"):          # :183-184
        it.step(1)
    it2 = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001)
    g, ng = it2._resolve_groups(7)                                          # :127-132 default: everything in group 0
    assert ng == 1 and np.array_equal(g, np.zeros(7, np.int32))


def test_synthetic_configs_have_the_documented_sizes():
    s, g, ng = synth.nacl()
    assert (s.num_particles, s.num_pairs, s.num_residues, ng) == (2500, 512, 512, 1)
    s, g, ng = synth.water_box(6400)
    assert (s.num_particles, s.num_pairs, ng) == (32000, 6400, 1)
    s, g, ng = synth.ionic_liquid(22)
    assert (s.num_particles, s.num_pairs, s.num_residues, ng) == (990, 330, 44, 2)
    s, g, ng = synth.mixed(60, 4)
    assert s.num_particles == 300 + 180 and ng == 4 and set(np.unique(g)) == {0, 1, 2, 3}
    assert np.all(g[s.pair_drude] == g[s.pair_parent])


def test_shard_bounds_cut_at_molecules():
    s, g, ng = synth.mixed(100, 7)
    for w in (2, 3, 4, 8):
        b = shard_bounds(s, w)
        assert b[0] == 0 and b[-1] == s.num_particles and all(x <= y for x, y in zip(b, b[1:]))
        for x in b[1:-1]:
            assert s.resid[x] != s.resid[x - 1]
        parts = [s.slice_molecules(lo, hi) for lo, hi in zip(b, b[1:])]
        assert sum(p.num_particles for p in parts) == s.num_particles
        assert sum(p.num_pairs for p in parts) == s.num_pairs


def test_xml_serialization_round_trip():
    """serialization/tests/TestSerializeDrudeTGNHIntegrator.cpp:45-69, same eight comparisons."""
    from openmm_drudenose_amd import serialization
    integ1 = DrudeTGNHIntegrator(301.1, 0.1, 10.5, 0.005, 0.001)
    xml = serialization.serialize(integ1)
    assert 'type="DrudeTGNHIntegrator"' in xml and 'version="1"' in xml
    integ2 = serialization.deserialize(xml)
    for getter in ("getTemperature", "getCouplingTime", "getDrudeTemperature", "getDrudeCouplingTime",
                   "getDrudeStepsPerRealStep", "getNumNHChains", "getUseDrudeNHChains", "getConstraintTolerance"):
        assert getattr(integ1, getter)() == getattr(integ2, getter)(), getter
    # what the reference's proxy drops survives here
    integ1.setMaxDrudeDistance(0.02); integ1.setUseCOMTempGroup(False)
    integ1.addTempGroup(); integ1.addTempGroup()
    for g in (0, 1, 1, 0):
        integ1.addParticleTempGroup(g)
    integ3 = serialization.deserialize(serialization.serialize(integ1))
    assert integ3.getMaxDrudeDistance() == 0.02 and integ3.getUseCOMTempGroup() == 0
    assert integ3.getNumTempGroups() == 2 and [integ3.getParticleTempGroup(i) for i in range(4)] == [0, 1, 1, 0]
    with pytest.raises(TgnhError, match="Unsupported version"):
        serialization.deserialize(xml.replace('version="1"', 'version="2"'))


def test_cpp_mirror_class(tmp_path):
    """include/DrudeTGNHIntegratorHip.hpp (C++ host-side mirror of the API class) against the library, host-only."""
    import subprocess
    _lib.load()
    exe = tmp_path / "test_mirror"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "test_mirror.cpp"), "-o", str(exe),
                    "-L", libdir, "-ldrudetgnh_hip", f"-Wl,-rpath,{libdir}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"],
                   check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


def test_plugin_tree_configures_without_openmm(tmp_path):
    """openmm_glue/CMakeLists.txt: without OPENMM_DIR the project still configures (the C-ABI library target only) and
    says so; with OpenMM it adds platforms/hip and python/ (not available in this image)."""
    import shutil
    import subprocess
    if shutil.which("cmake") is None or shutil.which("ninja") is None:
        pytest.skip("cmake / ninja not installed")
    src = os.path.join(ROOT, "openmm_drudenose_amd", "csrc", "openmm_glue")
    r = subprocess.run(["cmake", "-S", src, "-B", str(tmp_path / "b"), "-G", "Ninja"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "building libdrudetgnh_hip only" in r.stdout
    for f in ("platforms/hip/CMakeLists.txt", "platforms/hip/src/HipDrudeTGNHKernels.cpp", "python/drudetgnhplugin.i", "python/setup.py.in"):
        assert os.path.exists(os.path.join(src, f)), f
    # the SWIG interface declares the reference's surface: the class, its Python-side default, the OUTPUT typemap
    text = open(os.path.join(src, "python", "drudetgnhplugin.i")).read()
    for needle in ("%module drudetgnhplugin", "class DrudeTGNHIntegrator : public Integrator", "int useDrudeNHChains = True",
                   "int addParticleTempGroup(int tempGroup);", "%apply int& OUTPUT { int& tempGroup };", "virtual void step(int steps);"):
        assert needle in text, needle
