"""The OpenMM-HIP glue's REAL sources on the device.  OpenMM is absent from the image, so the plugin cannot be
built; but every OpenMM symbol the glue touches is declared in tests/cpp/openmm_shim, and in its functional form (shim_runtime.cpp:
a miniature runtime behind those signatures -- NOT OpenMM, it pins nothing about OpenMM's behaviour) the unmodified files
openmm_glue/platforms/hip/src/HipDrudeTGNHKernels.cpp and openmm_glue/serialization/src/*.cpp compile, link with
libdrudetgnh_hip.so and run: registerKernelFactories -> factory -> initialize -> execute per step -> computeKineticEnergy, in the
glue's four build variants, checked against the committed vectors, the oracle, and the C++ mirror of the sequence
(tests/cpp/test_glue_sequence.cpp); and the XML proxy with the thermostat checkpoint: serialize mid-run, deserialize, a new kernel
takes the parked state, the run continues bit for bit.  What this cannot show is that OpenMM's classes behave as the shim's do
(INTEGRATION.md section 6 lists the assumptions)."""
import os
import subprocess

import numpy as np
import pytest

from openmm_drudenose_amd import synth, _lib
from openmm_drudenose_amd.drudetgnhplugin import DrudeTGNHIntegrator
from helpers import make_oracle, rel_err
import test_glue_sequence_gpu as mirror

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GLUE = os.path.join(ROOT, "openmm_drudenose_amd", "csrc", "openmm_glue")
VARIANTS = {0: [], _lib.FLAG_RESIDENT_STEP: ["-DDRUDETGNH_RESIDENT_STEP"], _lib.FLAG_TRUST_STATE_CHANGED: ["-DDRUDETGNH_TRUST_STATE_CHANGED"],
            _lib.FLAG_TRUST_STATE_CHANGED | _lib.FLAG_RESIDENT_STEP: ["-DDRUDETGNH_RESIDENT_STEP", "-DDRUDETGNH_TRUST_STATE_CHANGED"]}


@pytest.fixture(scope="module")
def exes(tmp_path_factory):
    """the glue in its four build variants (CMake options of openmm_glue/platforms/hip), each with the thermostat checkpoint"""
    _lib.load()
    d = tmp_path_factory.mktemp("glue_linked")
    libdir = os.path.dirname(_lib.LIB_PATH)
    shim = os.path.join(ROOT, "tests", "cpp", "openmm_shim")
    out = {}
    for flags, defs in VARIANTS.items():
        exe = str(d / f"test_glue_linked_{flags}")
        subprocess.run(["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-DTGNH_SHIM_FUNCTIONAL", "-DDRUDETGNH_THERMOSTAT_CHECKPOINT"] + defs +
                       ["-I", shim, "-I", os.path.join(GLUE, "platforms", "hip", "src"), "-I", os.path.join(GLUE, "serialization", "include"),
                        "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                        os.path.join(ROOT, "tests", "cpp", "test_glue_linked.cpp"), os.path.join(shim, "shim_runtime.cpp"),
                        os.path.join(GLUE, "platforms", "hip", "src", "HipDrudeTGNHKernels.cpp"),
                        os.path.join(GLUE, "serialization", "src", "DrudeTGNHIntegratorProxy.cpp"),
                        os.path.join(GLUE, "serialization", "src", "DrudeTGNHSerializationProxyRegistration.cpp"),
                        "-o", exe, "-L", libdir, "-ldrudetgnh_hip", f"-Wl,-rpath,{libdir}", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-ldl"],
                       check=True)
        out[flags] = exe
    return out


def run_linked(exe, tmp_path, *a, split_at=0, **kw):
    """mirror.run_glue with the linked executable (same input files; `flags` is this executable's build variant, not an input)"""
    if not split_at:
        return mirror.run_glue(exe, tmp_path, *a, **kw)
    wrapper = tmp_path / "with_split.sh"
    wrapper.write_text(f'#!/bin/bash\nexec "{exe}" "$@" {split_at}\n')
    wrapper.chmod(0o755)
    return mirror.run_glue(str(wrapper), tmp_path, *a, **kw)


@pytest.mark.parametrize("flags", list(VARIANTS))
def test_the_linked_glue_fused_sequence_against_the_committed_vectors(exes, tmp_path, flags):
    """initialize() + 40 x execute() on the case nacl_tgnh of tests/golden/oracle_regression.npz (no constraints: tgnh_step_begin /
    calcForcesAndEnergy / tgnh_step_end), double precision; computeKineticEnergy(isKESumValid = true) = the cached half sum"""
    frozen = np.load(os.path.join(ROOT, "tests", "golden", "oracle_regression.npz"))
    s, g, ng = synth.nacl()
    pos, vel, eta_dot, ke = run_linked(exes[flags], tmp_path, s, g, ng, 40, False, 1, True, True, 0.001, 0.02, 1e-5, "double", flags)
    assert rel_err(pos[:64], frozen["nacl_tgnh/pos64"]) <= 1e-6
    assert rel_err(vel[:64], frozen["nacl_tgnh/vel64"]) <= 1e-6
    assert np.allclose(eta_dot, frozen["nacl_tgnh/etaDot"], rtol=1e-6, atol=1e-9)
    assert ke == pytest.approx(0.5 * frozen["nacl_tgnh/ke"][-1].sum(), rel=1e-6)


@pytest.mark.parametrize("flags", list(VARIANTS))
@pytest.mark.parametrize("name", ["rigid water", "ionic liquid"])
def test_the_linked_glue_split_sequence_with_call_outs_and_state_changes(exes, tmp_path, name, flags):
    """Constraints and virtual sites: execute()'s split branch around applyConstraints / computeVirtualSites / calcForcesAndEnergy /
    applyVelocityConstraints, a CMMotionRemover in the System (has_cm_motion_remover; not on the velocity-neutral list, so the
    TRUST variants must not trust) that changes the velocities between steps + stateChanged.  Against the oracle doing the same."""
    s, g, ng = synth.water_box(64, rigid=True) if name == "rigid water" else synth.ionic_liquid(12, constrained=True)
    tol, nsteps = 1e-10, 40
    chains = 1 if flags & _lib.FLAG_RESIDENT_STEP else 2
    pos, vel, eta_dot, ke = run_linked(exes[flags], tmp_path, s, g, ng, nsteps, True, chains, True, True, 0.001, 0.02, tol, "mixed", flags)
    it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001, 20, chains, True, True)
    it.setMaxDrudeDistance(0.02)
    s.has_cm_motion_remover = True                               # (the glue finds the CMMotionRemover among the System's forces: three degrees of freedom fewer)
    o = make_oracle(s, g, ng, "TGNH", it)
    x0 = s.positions.astype(np.float32).astype(np.float64)
    po, vo = s.positions.copy(), s.velocities.copy()
    m = s.mass
    massive = m > 0
    f = o.harness_force(po, x0, synth.K_DRUDE, synth.K_TETHER)
    for _ in range(nsteps):
        vo[massive] -= (m[massive, None] * vo[massive]).sum(0) / m[massive].sum()
        o.run_harness_constrained(po, vo, f, x0, synth.K_DRUDE, synth.K_TETHER, tol, 1)
    ep, ev = rel_err(pos, po), rel_err(vel, vo)
    print(f"linked glue, split path, {name}, variant {flags}: pos {ep:.2e} vel {ev:.2e}")
    assert ep <= 1e-6 and ev <= 1e-6
    assert np.allclose(eta_dot, o.chain(1), rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("flags", [_lib.FLAG_TRUST_STATE_CHANGED, _lib.FLAG_TRUST_STATE_CHANGED | _lib.FLAG_RESIDENT_STEP, 0])
def test_a_trusting_build_hears_of_velocities_the_user_sets(exes, tmp_path, flags):
    """-DDRUDETGNH_TRUST_STATE_CHANGED with a System of velocity-neutral forces only (a DrudeForce): the glue takes the flag, the begin
    half starts from the carried sums -- and when the user writes velocities between steps (Context::setVelocities -> stateChanged ->
    isKineticEnergySumValid() false) execute() must forward it (tgnh_state_changed), or the next chain runs on a stale sum (1e-3).
    Against the oracle with the same change of the velocities before every step."""
    s, g, ng = synth.nacl()
    nsteps = 30
    pos, vel, eta_dot, ke = run_linked(exes[flags], tmp_path, s, g, ng, nsteps, 2, 1, True, True, 0.001, 0.02, 1e-5, "double", flags)
    it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001, 20, 1, True, True)
    it.setMaxDrudeDistance(0.02)
    o = make_oracle(s, g, ng, "TGNH", it)
    x0 = s.positions.copy()
    po, vo = s.positions.copy(), s.velocities.copy()
    m = s.mass
    massive = m > 0
    f = o.harness_force(po, x0, synth.K_DRUDE, synth.K_TETHER)
    for _ in range(nsteps):
        vo[massive] -= (m[massive, None] * vo[massive]).sum(0) / m[massive].sum()
        vo *= 0.97                                               # (6 % of every kinetic energy: without the forwarding the velocities end 1e-3 off, checked once)
        o.run_harness(po, vo, f, x0, synth.K_DRUDE, synth.K_TETHER, 1)
    ep, ev = rel_err(pos, po), rel_err(vel, vo)
    print(f"linked glue, user-set velocities, variant {flags}: pos {ep:.2e} vel {ev:.2e}")
    assert ep <= 1e-6 and ev <= 1e-6
    assert np.allclose(eta_dot, o.chain(1), rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("flags", [0, _lib.FLAG_RESIDENT_STEP])      # (a TRUST build recomputes the sums on the first step of a new integrator -- its
@pytest.mark.parametrize("name", ["nacl", "rigid water"])              # isKineticEnergySumValid() starts false, as the API's -- where the old one carried them: equal to rounding only)
def test_checkpoint_through_the_xml_proxy_continues_bit_for_bit(exes, tmp_path, name, flags):
    """After 17 of 40 steps: DrudeTGNHIntegratorProxy::serialize (found in the registry the library's constructor filled) with the live
    kernel's thermostat and clock (DrudeTGNHThermostatStore), deserialize into a NEW integrator, the old kernel and integrator are
    destroyed, a new kernel's initialize() takes the parked state -- and steps 18-40 give the bits of the uninterrupted run."""
    if name == "nacl":
        s, g, ng = synth.nacl()
        args = (s, g, ng, 40, False, 3, True, True, 0.001, 0.02, 1e-5, "double", flags)
    else:
        s, g, ng = synth.water_box(64, rigid=True)
        args = (s, g, ng, 40, True, 1, True, True, 0.001, 0.02, 1e-10, "mixed", flags)
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    whole = run_linked(exes[flags], tmp_path / "a", *args)
    parts = run_linked(exes[flags], tmp_path / "b", *args, split_at=17)
    for a, b in zip(whole, parts):
        assert np.array_equal(np.asarray(a), np.asarray(b))
    # what the C++ proxy wrote at step 17, read by the Python reader of the same format (openmm_drudenose_amd/serialization.py)
    from openmm_drudenose_amd import serialization
    it, state = serialization.deserialize(open(tmp_path / "b" / "out.bin.xml").read(), with_thermostat=True)
    assert (it.getStepSize(), it.getMaxDrudeDistance(), it.getNumNHChains(), it.getUseDrudeNHChains(), it.getUseCOMTempGroup()) == (0.001, 0.02, args[5], 1, 1)
    assert it.getConstraintTolerance() == args[10] and it.getNumTempGroups() == ng
    assert [it.getParticleTempGroup(i) for i in range(s.num_particles)] == [int(x) for x in g]
    assert state["stepCount"] == 17 and state["time"] == pytest.approx(17 * 0.001, rel=1e-12)
    assert len(state["etaDot"]) == len(whole[2]) and len(state["eta"]) == len(state["etaDotDot"]) > 0 and np.isfinite(state["etaDot"]).all()   # (etaDot: one more entry per thermostat, Cu :229-231)
    # ... and written again by the Python writer: the same tree
    import xml.etree.ElementTree as ET
    again = ET.fromstring(serialization.serialize(it, thermostat=state))
    first = ET.fromstring(open(tmp_path / "b" / "out.bin.xml").read())
    flat = lambda e: [(x.tag, sorted((k, v if k in ("type", "stepCount", "count", "group") else float(v)) for k, v in x.attrib.items())) for x in e.iter()]   # noqa: E731
    assert flat(again) == flat(first)


def test_the_mirror_is_the_glue(exes, tmp_path):
    """tests/cpp/test_glue_sequence.cpp drives the C++ mirror of the glue's sequence (include/DrudeTGNHIntegratorHip.hpp); the linked glue
    on the same inputs gives the same trajectory -- to rounding, not to the bit: the mirror's call-outs run on the integrating handle
    (whose sweep direction they turn), the linked test's on a handle of their own."""
    libdir = os.path.dirname(_lib.LIB_PATH)
    mexe = str(tmp_path / "test_glue_sequence")
    subprocess.run(["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                    os.path.join(ROOT, "tests", "cpp", "test_glue_sequence.cpp"), "-o", mexe,
                    "-L", libdir, "-ldrudetgnh_hip", f"-Wl,-rpath,{libdir}", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    s, g, ng = synth.ionic_liquid(12, constrained=True)
    args = (s, g, ng, 40, False, 2, True, True, 0.001, 0.02, 1e-10, "mixed", 0)
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    a = mirror.run_glue(mexe, tmp_path / "a", *args)
    b = run_linked(exes[0], tmp_path / "b", *args)
    assert rel_err(b[0], a[0]) < 1e-11 and rel_err(b[1], a[1]) < 1e-9
    assert np.allclose(b[2], a[2], rtol=1e-8, atol=1e-11) and b[3] == pytest.approx(a[3], rel=1e-10)
