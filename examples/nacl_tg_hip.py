#!/usr/bin/env python3
"""The reference's example (example/nacl_tg.py: 1 M NaCl(aq) in SWM4-NDP water, DrudeTGNHIntegrator(300 K, 0.1 ps,
1 K, 0.1 ps, 1 fs, 20), hard wall 0.02 nm) on the MI355X HIP path.  OpenMM's force field, minimiser and reporters
are not here: the topology (492 waters + 10 Na+ + 10 Cl-, 2500 sites, 512 Drude pairs) is generated and the force
call-out is the harness spring force.  Needs a GPU.

Note on what it prints: the harness force makes every site an independent harmonic oscillator, the textbook case in
which a Nose-Hoover thermostat is not ergodic -- instantaneous temperatures swing widely and only their long-time
means approach the targets (three chain links help; the CPU oracle shows the same numbers).  With a real force
field the thermostats hold their targets: tests/test_reference_water*.py (216 interacting waters, within 1.4 %)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from drudetgnhplugin import DrudeTGNHIntegrator, HipContext           # the reference: from drudetgnhplugin import *
from openmm_drudenose_amd import synth

temperature, REALFREQ, DRUDEFREQ, timestep, numDrudeSteps = 300.0, 0.1, 0.1, 0.001, 20      # nacl_tg.py:12-19
integ = DrudeTGNHIntegrator(temperature, REALFREQ, 1.0, DRUDEFREQ, timestep, numDrudeSteps, 3)  # nacl_tg.py:21 (+ numNHChains = 3)
integ.setMaxDrudeDistance(0.02)                                                              # nacl_tg.py:22
system, group, ngroups = synth.nacl()
context = HipContext(system, integ, mode="TGNH", precision="mixed")                          # 'CudaPrecision': 'mixed', nacl_tg.py:60
print("Simulating...")
dof, nkt = context.dof()
mean, n = 0.0, 0
for block in range(1, 21):
    for _ in range(10):
        integ.step(100)                                                                      # nacl_tg.py:94-95
        mean, n = mean + context.compute_kinetic_energies() / dof / synth.KB, n + 1
    t, steps = context.time()
    print(f"step {steps:6d}  time {t:7.3f} ps  KE {integ.computeKineticEnergy():10.3f} kJ/mol  "
          f"<T>(group, COM, Drude) so far = {', '.join(f'{x:7.2f}' for x in mean / n)} K")
print("Done!")
