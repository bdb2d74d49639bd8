"""What can the statistical pin of the oracle see?  (CPU only; test infrastructure.)

The reference holds no golden vectors for the step path: its tests assert statistics (testWater: mean temperature
within 3 % / 2 %; testSinglePair; nothing else).  This script seeds plausible mis-restatements into a COPY of the oracle
(oracle/libtgnh_oracle_mut.so, `MUT(k)` in oracle/tgnh_oracle.c -- constant 0 in the oracle proper) and records which
check notices each:

  water    the reference's testWater protocol (tests/test_reference_water.py; TGNH mode: the CUDA test's 10 000 samples, 2 %)
  pair     the reference's (disabled) testSinglePair: <KE_internal> within 1 %
  energy   the Nose-Hoover-chain invariant this repository adds (tests/helpers.py::extended_energy)
  bridge   TGNH == dualNH in the degenerate configuration: the mutated mode against the pristine other one

    python tests/pin_sensitivity.py            # everything (~10 CPU-minutes, 6 processes) -> tests/golden/pin_sensitivity.json
    python tests/pin_sensitivity.py --table    # print the committed table as markdown

tests/test_pin_sensitivity.py re-runs the cheap detectors on every mutant and checks them against the committed file.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

from openmm_drudenose_amd import synth                                   # noqa: E402
from openmm_drudenose_amd.drudetgnhplugin import DrudeTGNHIntegrator    # noqa: E402
from oracle import Oracle, MODE_DUALNH, MODE_TGNH, water_forces         # noqa: E402
from oracle.binding import load_mutants                                  # noqa: E402

OUT = os.path.join(HERE, "golden", "pin_sensitivity.json")
MODES = {"dualNH": MODE_DUALNH, "TGNH": MODE_TGNH}

MUTANTS = {
    0: ("none: the oracle as it is", ("dualNH", "TGNH")),
    1: ("CMMotionRemover's 3 degrees of freedom not subtracted (Ref :158-165, Cu :204-212)", ("dualNH", "TGNH")),
    2: ("3 degrees of freedom too few in the (first) real thermostat (Ref :157, Cu :195)", ("dualNH", "TGNH")),
    3: ("red_g dropped: N kT and Q from dof_g instead of dof_g - sum 3 m_i/M_res (Cu :130-132, :219)", ("TGNH",)),
    4: ("Ref :495 with the first loop's stride: exp(-dtc/8 etaDot[i + numTempGroup]) instead of [i + 2]", ("dualNH",)),
    5: ("Ref :477 'corrected' to stride 2: with useDrudeNHChains = false the real chain is no longer damped by the Drude "
        "thermostat's etaDot (the indexing quirk not reproduced)", ("dualNH",)),
    6: ("drudekbT and realkbT swapped in the higher links (Ref :498, Cu :589)", ("dualNH", "TGNH")),
    7: ("Cu :583-585 without expfac: link 0's second quarter-kick undamped by link 1", ("TGNH",)),
    8: ("kinetic energies from velocities with the molecular COM left in (K :123-129 skipped)", ("TGNH",)),
    9: ("control: velocities rescaled by the KE factor exp(-dtc etaDot) instead of exp(-dtc/2 etaDot) (Ref :483, Cu :573)",
        ("dualNH", "TGNH")),
}


def oracle(system, integ, group, ngroups, mode, mutant):
    lib = load_mutants()
    lib.tgo_set_mutant(int(mutant))
    return Oracle.from_integrator(system, integ, group, ngroups, MODES[mode], lib=lib)


# ---------------------------------------------------------------------------------------------------------------
# detectors
# ---------------------------------------------------------------------------------------------------------------
def detect_water(mutant, mode):
    """TestReferenceDrudeTGNHIntegrator.cpp:111-192 (dualNH: 4000 samples, 3 %) / TestCudaDrudeTGNHIntegrator.cpp (TGNH:
    10 000 samples, 2 %, velocity constraints after the second half kick, KE = the cached sum of the last half step)."""
    import water_test_system as wts
    s = wts.build()
    it = wts.integrator()
    o = oracle(s, it, np.zeros(s.num_particles, np.int32), 1, mode, mutant)
    target, _ = wts.expected_temperature(s)
    num_dof = 3 * 3 * 216 - len(s.constraints) - 3 + 3 * 216               # the TEST's own dof count (test :186-188)
    mass, dt, tol = s.mass, it.getStepSize(), it.getConstraintTolerance()
    massive = mass > 0
    inv = np.where(massive, 1.0 / np.where(massive, mass, 1.0), 0.0)
    pos, vel = s.positions.copy(), s.velocities.copy()
    f, _ = water_forces(pos, wts.BOX)
    last_ke = [0.0]

    def step():
        nonlocal f
        vel[massive] -= (mass[massive, None] * vel[massive]).sum(0) / mass[massive].sum()      # CMMotionRemover
        o.propagate_nhc(vel)
        o.half_kick(vel, f)
        delta = np.where(massive[:, None], vel * dt, 0.0)
        o.shake_positions(pos, delta, tol)
        pos[massive] += delta[massive]
        vel[massive] = delta[massive] / dt
        o.hardwall(pos, vel)
        o.virtual_sites(pos)
        f, _ = water_forces(pos, wts.BOX)
        o.half_kick(vel, f)
        if mode == "TGNH":
            o.shake_velocities(pos, vel, tol)                                                  # Cu :391
        ke, _ = o.propagate_nhc(vel)
        last_ke[0] = 0.5 * float(np.sum(ke))                                                   # Cu :493-497 (cached KESum)

    nsamp, gate = (4000, 0.03) if mode == "dualNH" else (10000, 0.02)
    for _ in range(5000):
        step()
    temps = np.zeros(nsamp)
    for i in range(nsamp):
        step()
        if mode == "dualNH":                                                                   # Ref :70-98
            sv = vel + f * (0.5 * dt * inv)[:, None]
            o.shake_velocities(pos, sv, 1e-4)
            k = 0.5 * float((mass[:, None] * sv ** 2).sum())
        else:
            k = last_ke[0]
        temps[i] = k / (0.5 * num_dof * synth.KB)
    blocks = temps[:nsamp // 20 * 20].reshape(20, -1).mean(1)
    dev = temps.mean() / target - 1.0
    return {"deviation": round(float(dev), 5), "stderr": round(float(blocks.std(ddof=1) / np.sqrt(20) / target), 5),
            "gate": gate, "caught": bool(not np.isfinite(dev) or abs(dev) > gate)}


def detect_pair(mutant, mode):
    """TestReferenceDrudeTGNHIntegrator.cpp:54-109: <KE_internal> = 3/2 kT(10 K) within 1 % (its <KE_cm> assertion does not
    hold for the unmutated algorithm either -- tests/test_oracle.py -- and is recorded only)."""
    from helpers import ONE_4PI_EPS0
    s, g, ng = synth.single_pair()
    it = DrudeTGNHIntegrator(300.0, 0.1, 10.0, 0.005, 0.003, 20, 2, False)
    it.setMaxDrudeDistance(0.05)
    o = oracle(s, it, g, ng, mode, mutant)
    k = ONE_4PI_EPS0 * 1.5
    pos, vel, x0 = s.positions.copy(), s.velocities.copy(), s.positions.copy()
    f = o.harness_force(pos, x0, k, 0.0)
    o.run_harness(pos, vel, f, x0, k, 0.0, 1000)
    m1, m2 = 1.0, 0.1
    tot, red = m1 + m2, m1 * m2 / (m1 + m2)
    ke_cm = ke_int = 0.0
    nsamp = 10000
    for _ in range(nsamp):
        o.run_harness(pos, vel, f, x0, k, 0.0, 10)
        vcm = vel[0] * (m1 / tot) + vel[1] * (m2 / tot)
        ke_cm += 0.5 * tot * vcm.dot(vcm)
        vi = vel[0] - vel[1]
        ke_int += 0.5 * red * vi.dot(vi)
    r_int = ke_int / nsamp / (1.5 * synth.KB * 10.0)
    r_cm = ke_cm / nsamp / (1.5 * synth.KB * 300.0)
    return {"ke_int_ratio": round(float(r_int), 4), "ke_cm_ratio": round(float(r_cm), 3), "caught": bool(not np.isfinite(r_int) or abs(r_int - 1) > 0.01)}


def detect_energy(mutant, mode):
    """tests/test_oracle.py::test_extended_energy_is_conserved on the mutant (27 waters, numNHChains = 3,
    useDrudeNHChains = true -- the invariant is defined for the self-consistent chain layout only)."""
    from helpers import extended_energy
    s, g, ng = synth.water_box(27)
    if mode == "dualNH":
        g, ng = np.zeros_like(g), 1
    worst = []
    for dt in (0.0005, 0.00025):
        it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, dt, 20, 3, True, True)
        o = oracle(s, it, g, ng, mode, mutant)
        pos, vel, x0 = s.positions.copy(), s.velocities.copy(), s.positions.copy()
        f = o.harness_force(pos, x0, synth.K_DRUDE, synth.K_TETHER)
        normal = o.normal_particles()

        def energy():
            return extended_energy(s, normal, pos, vel, x0, o.dof()[1], o.chain(0), o.chain(1), o.chain(3), 3,
                                   synth.KB * 300.0, synth.KB * 1.0, mode)
        h0, _, _ = energy()
        dev = 0.0
        for _ in range(10):
            o.run_harness(pos, vel, f, x0, synth.K_DRUDE, synth.K_TETHER, int(round(0.02 / dt)))
            dev = max(dev, abs(energy()[0] - h0))
        worst.append(dev / h0)
    ok = bool(np.isfinite(worst).all()) and worst[0] < 2e-4 and worst[1] < 5e-5 and 2.5 < worst[0] / worst[1] < 7.0
    return {"dH_over_H0": [float(f"{w:.2e}") for w in worst], "caught": bool(not ok)}


def detect_bridge(mutant, mode):
    """tests/test_oracle.py::test_bridge_identity_tgnh_equals_dualnh with the mutated mode against the pristine other."""
    from helpers import oracle_run, rel_err
    s, g, ng = synth.water_box(27)
    it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001, 20, 3, True, False)
    it.setMaxDrudeDistance(0.02)
    other = "TGNH" if mode == "dualNH" else "dualNH"
    pm, vm = oracle_run(oracle(s, it, g, ng, mode, mutant), s, 100)
    po, vo = oracle_run(oracle(s, it, g, ng, other, 0), s, 100)
    e = max(rel_err(pm, po), rel_err(vm, vo))
    return {"max_rel_err": float(f"{e:.2e}"), "caught": bool(e > 1e-9)}


CHEAP = {"pair": detect_pair, "energy": detect_energy, "bridge": detect_bridge}


def _job(args):
    name, mutant, mode = args
    fn = detect_water if name == "water" else CHEAP[name]
    return name, mutant, mode, fn(mutant, mode)


def run_all(with_water=True, processes=6):
    import multiprocessing as mp
    jobs = []
    for k, (_, modes) in MUTANTS.items():
        for mode in modes:
            for name in (("water",) if with_water else ()) + tuple(CHEAP):
                jobs.append((name, k, mode))
    jobs.sort(key=lambda j: (j[0] != "water", j[2] != "TGNH"))              # the long ones first
    table = {str(k): {"what": MUTANTS[k][0], "modes": {m: {} for m in MUTANTS[k][1]}} for k in MUTANTS}
    with mp.get_context("spawn").Pool(processes) as pool:
        for name, k, mode, res in pool.imap_unordered(_job, jobs):
            table[str(k)]["modes"][mode][name] = res
            print(f"mutant {k} {mode:6s} {name:7s} {res}", flush=True)
    return table


def markdown(table):
    rows = ["| # | mis-restatement seeded into a copy of the oracle | mode | testWater (3 % / 2 %) | testSinglePair (1 %) | extended energy | bridge |",
            "|---|---|---|---|---|---|---|"]

    def cell(r, fmt):
        if r is None:
            return "—"
        return ("**caught** " if r["caught"] else "not seen ") + fmt(r)
    for k in sorted(table, key=int):
        for mode, d in table[k]["modes"].items():
            rows.append("| {} | {} | {} | {} | {} | {} | {} |".format(
                k, table[k]["what"], mode,
                cell(d.get("water"), lambda r: f"({r['deviation']:+.2%} ± {r['stderr']:.2%})"),
                cell(d.get("pair"), lambda r: f"(⟨KE_int⟩ {r['ke_int_ratio']:.3f}, ⟨KE_cm⟩ {r['ke_cm_ratio']:.2f})"),
                cell(d.get("energy"), lambda r: f"(dH/H₀ {r['dH_over_H0'][0]:.1e}, {r['dH_over_H0'][1]:.1e})"),
                cell(d.get("bridge"), lambda r: f"({r['max_rel_err']:.1e})")))
    return "\n".join(rows)


def summary(table):
    """Which seeded mis-restatements the reference's own checks (testWater, testSinglePair) notice, which only the
    cross-checks this repository adds (extended energy, bridge), and which nothing does."""
    seen_ref, seen_added, seeded = [], [], []
    for k in sorted(table, key=int):
        if k == "0":
            continue
        for mode, d in sorted(table[k]["modes"].items()):
            seeded.append([int(k), mode])
            if d["water"]["caught"] or d["pair"]["caught"]:
                seen_ref.append([int(k), mode])
            if d["energy"]["caught"] or d["bridge"]["caught"]:
                seen_added.append([int(k), mode])
    return {"seeded": seeded,
            "unseen_by_testWater": [x for x in seeded if not table[str(x[0])]["modes"][x[1]]["water"]["caught"]],
            "unseen_by_reference_checks": [x for x in seeded if x not in seen_ref],
            "unseen_by_all": [x for x in seeded if x not in seen_ref and x not in seen_added]}


SUMMARY = OUT.replace("pin_sensitivity.json", "pin_sensitivity_summary.json")

if __name__ == "__main__":
    if "--table" in sys.argv:
        print(markdown(json.load(open(OUT))))
    elif "--summary" in sys.argv:
        json.dump(summary(json.load(open(OUT))), open(SUMMARY, "w"), indent=1)
        print(open(SUMMARY).read())
    else:
        t = run_all(with_water="--cheap" not in sys.argv)
        if "--cheap" not in sys.argv:
            json.dump(t, open(OUT, "w"), indent=1, sort_keys=True)
            json.dump(summary(t), open(SUMMARY, "w"), indent=1)
        print(markdown(t))
