"""XML round trip of DrudeTGNHIntegrator and a thermostat-state checkpoint.

The XML carries the nine properties of the reference's proxy under the same names and version
(serialization/src/DrudeTGNHIntegratorProxy.cpp:43-55), in the shape OpenMM's XmlSerializer gives a
root node (`<Integrator type="DrudeTGNHIntegrator" version="1" .../>`), so files are interchangeable.
What the reference's proxy drops (maxDrudeDistance, useCOMTempGroup, temperature groups)
is written as extra attributes / children that an older reader ignores, and so are the thermostat
variables (eta, etaDot, etaDotDot) and the clock, which the reference cannot checkpoint at all -- all in
the node shapes of the plugin tree's C++ proxy (csrc/openmm_glue/serialization/src/DrudeTGNHIntegratorProxy.cpp:
`<TempGroups count><Particle group/>...`, `<ThermostatState time stepCount><eta><Value v/>...`), so a file
written by either is read by the other (tests/test_glue_linked_gpu.py feeds the C++ proxy's output to this reader).
"""
import xml.etree.ElementTree as ET

import numpy as np

from .drudetgnhplugin import DrudeTGNHIntegrator, TgnhError
from . import _lib

_DOUBLES = [("stepSize", "getStepSize"), ("constraintTolerance", "getConstraintTolerance"),
            ("temperature", "getTemperature"), ("couplingTime", "getCouplingTime"),
            ("drudeTemperature", "getDrudeTemperature"), ("drudeCouplingTime", "getDrudeCouplingTime")]
_INTS = [("drudeStepsPerRealStep", "getDrudeStepsPerRealStep"), ("numNHChains", "getNumNHChains"),
         ("useDrudeNHChains", "getUseDrudeNHChains")]


def serialize(integrator, root_name="Integrator", thermostat=None):
    """thermostat: a dict as save_thermostat() returns -> a ThermostatState child (eta, etaDot, etaDotDot, time, stepCount)"""
    node = ET.Element(root_name, {"type": "DrudeTGNHIntegrator", "version": "1"})
    for name, getter in _DOUBLES:
        node.set(name, repr(float(getattr(integrator, getter)())))
    for name, getter in _INTS:
        node.set(name, str(int(getattr(integrator, getter)())))
    # extensions (ignored by the reference's version-1 reader)
    node.set("maxDrudeDistance", repr(float(integrator.getMaxDrudeDistance())))
    node.set("useCOMTempGroup", str(int(integrator.getUseCOMTempGroup())))
    if integrator.getNumTempGroups() > 0:
        g = ET.SubElement(node, "TempGroups", {"count": str(integrator.getNumTempGroups())})
        for x in integrator._particleTempGroup:
            ET.SubElement(g, "Particle", {"group": str(int(x))})
    if thermostat is not None:
        t = ET.SubElement(node, "ThermostatState", {"time": repr(float(thermostat["time"])), "stepCount": str(int(thermostat["stepCount"]))})
        for key in ("eta", "etaDot", "etaDotDot"):
            a = ET.SubElement(t, key)
            for v in np.asarray(thermostat[key], np.float64):
                ET.SubElement(a, "Value", {"v": repr(float(v))})
    return ET.tostring(node, encoding="unicode")


def deserialize(text, with_thermostat=False):
    """-> the integrator; with_thermostat: (integrator, dict or None) -- the ThermostatState child as load_thermostat() takes it"""
    node = ET.fromstring(text)
    if node.get("type") != "DrudeTGNHIntegrator":
        raise TgnhError(_lib.ERR_ARG, "not a DrudeTGNHIntegrator node")
    if int(node.get("version")) != 1:                            # DrudeTGNHIntegratorProxy.cpp:58-59
        raise TgnhError(_lib.ERR_ARG, "Unsupported version number")
    it = DrudeTGNHIntegrator(float(node.get("temperature")), float(node.get("couplingTime")),
                             float(node.get("drudeTemperature")), float(node.get("drudeCouplingTime")),
                             float(node.get("stepSize")), int(node.get("drudeStepsPerRealStep")),
                             int(node.get("numNHChains")), bool(int(node.get("useDrudeNHChains"))))
    it.setConstraintTolerance(float(node.get("constraintTolerance")))
    if node.get("maxDrudeDistance") is not None:
        it.setMaxDrudeDistance(float(node.get("maxDrudeDistance")))
    if node.get("useCOMTempGroup") is not None:
        it.setUseCOMTempGroup(int(node.get("useCOMTempGroup")))
    g = node.find("TempGroups")
    if g is not None:
        for _ in range(int(g.get("count"))):
            it.addTempGroup()
        for x in (g.text or "").split():                         # (files written before round 5 held the table as text)
            it.addParticleTempGroup(int(x))
        for p in g.findall("Particle"):
            it.addParticleTempGroup(int(p.get("group")))
    if not with_thermostat:
        return it
    t = node.find("ThermostatState")
    state = None
    if t is not None:
        state = {"time": float(t.get("time")), "stepCount": int(t.get("stepCount"))}
        for key in ("eta", "etaDot", "etaDotDot"):
            a = t.find(key)
            state[key] = np.array([float(v.get("v")) for v in a.findall("Value")]) if a is not None else np.zeros(0)
    return it, state


def save_thermostat(ctx):
    """Thermostat variables + clock of a context: what a checkpoint must hold besides positions/velocities.
    (Between steps of a TGNH_FLAG_DEFER_SCALE context the chain already holds the coming step's first half: checkpoint
    a plain-variant context, or the deferred one before its first step.)"""
    t, k = ctx.time()
    return {"eta": ctx.thermostat_state(0), "etaDot": ctx.thermostat_state(1), "etaDotDot": ctx.thermostat_state(2),
            "etaMass": ctx.thermostat_state(3), "time": t, "stepCount": k}


def load_thermostat(ctx, state):
    for which, key in enumerate(("eta", "etaDot", "etaDotDot", "etaMass")):
        if key in state:
            ctx.set_thermostat_state(which, np.asarray(state[key], np.float64))
    if "time" in state and "stepCount" in state:
        ctx.set_time(float(state["time"]), int(state["stepCount"]))
