"""How far ONE ulp in one velocity component carries on the CPU oracle, by chain length (not a test: run by hand, `python
tests/ulp_sensitivity.py`; the numbers are in profiles/r04_fuzz_soak.md).  Two oracle runs of the same box, the second with
one massive slot's v_y moved to the next double: chains of four and six links far from equilibrium -- the synthetic boxes -- are
chaotic within a few hundred steps, which is why tools/fuzz_soak.py gates its walks against a twin instead of a constant."""
import sys, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import rel_err
from openmm_drudenose_amd import synth
from oracle import Oracle, MODE_TGNH
def run(s,g,ng,chains,nsteps,perturb):
    o=Oracle(s,g,ng,MODE_TGNH,300.0,0.1,1.0,0.005,0.001,20,chains,True,True,0.02)
    pos,vel=s.positions.copy(),s.velocities.copy()
    if perturb:
        i=np.flatnonzero(s.mass>0)[7]; vel[i,1]=np.nextafter(vel[i,1],np.inf)
    x0=s.positions.copy()
    f=o.harness_force(pos,x0,synth.K_DRUDE,synth.K_TETHER)
    o.run_harness(pos,vel,f,x0,synth.K_DRUDE,synth.K_TETHER,nsteps)
    return pos,vel,[o.chain(w) for w in (0,1)]
for name,mk in [("mixed-60-6",lambda:synth.mixed(60,6)),("water-400",lambda:synth.water_box(400)),("nacl",synth.nacl)]:
    s,g,ng=mk()
    for chains in (1,2,3,4,6):
        for nsteps in (300,700):
            a=run(s,g,ng,chains,nsteps,False); b=run(s,g,ng,chains,nsteps,True)
            th=max(float(np.max(np.abs(np.asarray(x)-np.asarray(y)))) for x,y in zip(a[2],b[2]))
            print(f"{name} C={chains} steps={nsteps}: pos {rel_err(a[0],b[0]):.2e} vel {rel_err(a[1],b[1]):.2e} thermostat abs {th:.2e}",flush=True)
