"""Where does testWater's temperature offset in TGNH mode come from?  (CPU only; test infrastructure.)

The CUDA platform's testWater (platforms/cuda/tests/TestCudaDrudeTGNHIntegrator.cpp:111-192) averages
`context.getState(Energy).getKineticEnergy()` = the cached KESum of the LAST thermostat half step
(CudaDrudeTGNHKernels.cpp:493-497, :654-658): 1/2 sum of the kinetic-energy bins BEFORE that half step's rescale,
and compares it with (numStandardDof T + numDrudeDof T_D)/numDof (test :186-190) at 2 %.

This script runs that protocol on the oracle (oracle/tgnh_oracle.c, TGNH mode) for several independent
trajectories (the lattice's molecules displaced rigidly by N(0, 1e-4 nm) per seed; seed 0 = the test's own lattice)
and records, per thermostat bin b in {group 0, molecular COM, Drude}:

    pre[b]   <KE_b> the second half step's chain starts from   (what KESum is made of, Cu :493-497)
    mid[b]   <s_b^2 KE_b> after that half step's rescale      = what the velocities carry at the end of the step
    post[b]  the same after the NEXT step's first half step   (the other end of the thermostat's full step)

each as a ratio to N_b kT_b / 2 (the thermostat's own target, Cu :219, :227-235), and the temperature the test's
formula gives for the sum over bins from each of the three -- with the test's numDof and with the thermostats' own
dof sum (they coincide for this system: asserted).

    python tests/water_decomposition.py [--seeds 8] [--samples 10000] [--mode TGNH|dualNH] -> tests/golden/water_decomposition.json
    python tests/water_decomposition.py --table      # the committed result as markdown
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
from oracle import Oracle, MODE_DUALNH, MODE_TGNH, water_forces         # noqa: E402
import water_test_system as wts                                          # noqa: E402

OUT = os.path.join(HERE, "golden", "water_decomposition.json")


def jittered_system(seed):
    """The test's lattice; seed > 0: every molecule displaced rigidly (constraints and the virtual site stay exact)."""
    s = wts.build()
    if seed:
        rng = np.random.default_rng(1000 + seed)
        shift = rng.normal(0.0, 1e-4, (s.num_particles // 5, 3))
        s.positions = s.positions + np.repeat(shift, 5, axis=0)
    return s


def run(seed, mode, samples, equil=5000, chains=None, label="as the test"):
    s = jittered_system(seed)
    it = wts.integrator()
    if chains is not None:
        it.setNumNHChains(chains)
    o = Oracle.from_integrator(s, it, np.zeros(s.num_particles, np.int32), 1, MODE_TGNH if mode == "TGNH" else MODE_DUALNH)
    dof, nkt = o.dof()
    target, num_dof = wts.expected_temperature(s)
    mass, dt, tol = s.mass, it.getStepSize(), it.getConstraintTolerance()
    massive = mass > 0
    inv = np.where(massive, 1.0 / np.where(massive, mass, 1.0), 0.0)
    pos, vel = s.positions.copy(), s.velocities.copy()
    f, _ = water_forces(pos, wts.BOX)
    nt = o.num_thermostats()
    rec = {"first_pre": np.zeros(nt), "first_post": np.zeros(nt), "pre": np.zeros(nt), "mid": np.zeros(nt)}

    def step():
        nonlocal f
        vel[massive] -= (mass[massive, None] * vel[massive]).sum(0) / mass[massive].sum()      # CMMotionRemover (API :186)
        ke, sc = o.propagate_nhc(vel)                                                          # Cu :336
        rec["first_pre"], rec["first_post"] = ke, ke * sc * sc
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
        ke, sc = o.propagate_nhc(vel)                                                          # Cu :394
        rec["pre"], rec["mid"] = ke, ke * sc * sc                                              # KESum = 1/2 sum(pre), Cu :493-497

    for _ in range(equil):
        step()
    acc = {k: np.zeros((samples, nt)) for k in ("pre", "mid", "post")}
    shifted = np.zeros(samples)
    plain = np.zeros(samples)
    # the thermostats' own equation of motion, averaged over the sampling window (Cu :566-592 in continuous time):
    #   Q_0 d(etaDot_0)/dt = (KE - N kT) - Q_0 etaDot_0 etaDot_1
    #   =>  <KE - N kT> = Q_0 <etaDot_0 etaDot_1> + Q_0 [etaDot_0(end) - etaDot_0(start)] / T
    C = it.getNumNHChains()
    link0, link1, q0 = link_indices(mode, nt, C, o)
    ed_prod = np.zeros(nt)
    ed0_start = o.chain(1)[link0].copy()
    for i in range(samples):
        step()
        ed = o.chain(1)
        ed_prod += ed[link0] * (ed[link1] if not isinstance(link1, tuple) else 0.5 * (ed[link1[0]] + ed[link1[1]]))
        acc["pre"][i], acc["mid"][i] = rec["pre"], rec["mid"]
        if i:
            acc["post"][i - 1] = rec["first_post"]            # the first half step of THIS step closes the last one's full thermostat step
        plain[i] = 0.5 * float((mass[:, None] * vel ** 2).sum())
        sv = vel + f * (0.5 * dt * inv)[:, None]                # Ref :70-98 (the Reference platform's query)
        o.shake_velocities(pos, sv, 1e-4)
        shifted[i] = 0.5 * float((mass[:, None] * sv ** 2).sum())
    acc["post"][samples - 1] = acc["post"][samples - 2]
    half_nkt = 0.5 * nkt
    ed0_end = o.chain(1)[link0]
    coupling = q0 * ed_prod / samples                        # Q_0 <etaDot_0 etaDot_1>
    drift = q0 * (ed0_end - ed0_start) / (samples * dt)      # Q_0 d<etaDot_0>/dt over the window
    out = {"seed": seed, "mode": mode, "label": label, "chains": C, "equil": equil,
           "samples": samples, "dt": dt, "target": target, "num_dof_test": int(num_dof),
           "chain_coupling_over_nkt": [float(coupling[b] / nkt[b]) if nkt[b] > 0 else None for b in range(nt)],
           "chain_drift_over_nkt": [float(drift[b] / nkt[b]) if nkt[b] > 0 else None for b in range(nt)],
           "dof_thermostats": [float(x) for x in dof], "nkt": [float(x) for x in nkt]}
    live = half_nkt > 0
    for k in ("pre", "mid", "post"):
        m = acc[k].mean(0) * 0.5
        out[k + "_ratio"] = [float(m[b] / half_nkt[b]) if live[b] else None for b in range(nt)]
        t = acc[k].sum(1) * 0.5 / (0.5 * num_dof * synth.KB)
        out["T_" + k] = float(t.mean())
        blocks = t[:samples // 20 * 20].reshape(20, -1).mean(1)
        out["T_" + k + "_stderr"] = float(blocks.std(ddof=1) / np.sqrt(20))
    out["T_plain_velocities"] = float(plain.mean() / (0.5 * num_dof * synth.KB))       # 1/2 sum m v^2 at the end of the step
    out["T_shifted_projected"] = float(shifted.mean() / (0.5 * num_dof * synth.KB))    # Ref :70-98
    out["T_own_dof_pre"] = float(acc["pre"].sum(1).mean() * 0.5 / (0.5 * dof.sum() * synth.KB))
    return out


def link_indices(mode, nt, C, o):
    """Where links 0 and 1 of every thermostat sit in the oracle's etaDot vector, and Q of link 0.
    TGNH: rows [thermostat][C + 1] (Cu :94-97).  dualNH: the Reference platform's interleaved vector (Ref :186-217):
    real link i at 2 i, Drude link i at 2 i + 1 with useDrudeNHChains; without (the test's setting) [real0, drude0, real1, ...],
    where the loop that damps link 0 (Ref :476-481) reads etaDot[i + 1]: real0 is damped by drude0, drude0 by real1."""
    mass = o.chain(3)
    if mode == "TGNH":
        l0 = np.arange(nt) * (C + 1)
        return l0, l0 + 1, mass[np.arange(nt) * C]
    # nt = 2 (real, Drude).  The descending loop damps link i with etaDot[i + 1] (Ref :477, numTempGroup = 1), the ascending one
    # with etaDot[i + 2] (Ref :495): half of each
    return np.array([0, 1]), (np.array([1, 2]), np.array([2, 3])), mass[:2]


def _job(args):
    return run(*args)


def pooled(rows, key):
    v = np.array([r[key] for r in rows])
    return float(v.mean()), float(v.std(ddof=1) / np.sqrt(len(v))) if len(v) > 1 else float("nan")


def table(res):
    lines = []
    for key, rows in res["runs"].items():
        tgt = rows[0]["target"]
        mode = rows[0]["mode"]
        lines.append(f"**{mode}, {rows[0]['label']}** (numNHChains = {rows[0]['chains']}, {rows[0]['equil']} equilibration steps), "
                     f"{len(rows)} independent trajectories x {rows[0]['samples']} samples, dt = {rows[0]['dt']} ps; "
                     f"expected {tgt:.2f} K (test :186-190)\n")
        lines.append("| kinetic energy taken | <T> by the test's formula | vs expected | pooled standard error |")
        lines.append("|---|---|---|---|")
        names = {"T_pre": "cached KESum: bins BEFORE the second half step's rescale (Cu :493-497, :654-658 -- what the CUDA test reads)",
                 "T_mid": "after that rescale = 1/2 sum m v^2 of the stored velocities, thermostat partition",
                 "T_post": "after the next step's first half step (other end of the thermostat's full step)",
                 "T_plain_velocities": "1/2 sum m v^2 at the end of the step (OpenMM's computeKineticEnergy(0), Cu :656)",
                 "T_shifted_projected": "shifted by dt/2 along the forces and projected (Ref :70-98 -- what the Reference test reads)"}
        for k, label in names.items():
            m, se = pooled(rows, k)
            lines.append(f"| {label} | {m:.2f} K | {m / tgt - 1:+.2%} | {se / tgt:.2%} |")
        lines.append("")
        nt = len(rows[0]["pre_ratio"])
        bins = ["real"] + [f"bin {b}" for b in range(1, nt - 1)] + ["Drude"] if mode == "dualNH" else ["group 0", "molecular COM", "Drude"]
        lines.append("| thermostat bin | N kT/2 share of the total | <KE> / (N kT/2): before the rescale | after | after the next first half "
                     "| mean of before/after/next - 1 | Q0 <etaDot0 etaDot1> / N kT | Q0 d<etaDot0>/dt / N kT |")
        lines.append("|---|---|---|---|---|---|---|---|")
        nkt = np.array(rows[0]["nkt"])
        for b in range(nt):
            if rows[0]["pre_ratio"][b] is None:
                continue
            r = [np.mean([row[k][b] for row in rows]) for k in ("pre_ratio", "mid_ratio", "post_ratio")]
            cpl = np.mean([row["chain_coupling_over_nkt"][b] for row in rows])
            drf = np.mean([row["chain_drift_over_nkt"][b] for row in rows])
            lines.append(f"| {bins[b]} | {nkt[b] / nkt.sum():.3%} | {r[0]:.4f} | {r[1]:.4f} | {r[2]:.4f} | {(r[0] + 2 * r[1] + r[2]) / 4 - 1:+.4f} | {cpl:+.4f} | {drf:+.4f} |")
        lines.append("")
    return "\n".join(lines)


def main():
    if "--table" in sys.argv:
        print(table(json.load(open(OUT))))
        return
    import multiprocessing as mp
    arg = lambda name, d: type(d)(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else d   # noqa: E731
    seeds, samples = arg("--seeds", 8), arg("--samples", 10000)
    # the protocol as the reference runs it, and two controls: one-link chains (a plain Nose-Hoover thermostat is an integral
    # controller: <KE> = N kT whatever the heating), and a ten times longer equilibration (the lattice has condensed further)
    variants = [("as the test", 5000, None), ("control: one-link chains", 5000, 1), ("control: 50 000 equilibration steps", 50000, None)]
    if "--quick" in sys.argv:
        variants = variants[:1]
    modes = [arg("--mode", "")] if "--mode" in sys.argv else ["TGNH", "dualNH"]
    jobs = [(s, m, samples if m == "TGNH" else min(samples, 4000), eq, ch, lab)
            for lab, eq, ch in variants for m in modes for s in range(seeds)]
    jobs.sort(key=lambda j: -j[3])                                # the long ones first
    with mp.Pool(min(8, len(jobs))) as pool:
        rows = pool.map(_job, jobs, chunksize=1)
    res = {"what": __doc__.split("\n")[0], "runs": {}}
    for lab, eq, ch in variants:
        for m in modes:
            res["runs"][f"{m} / {lab}"] = sorted([r for r in rows if r["mode"] == m and r["label"] == lab], key=lambda r: r["seed"])
    json.dump(res, open(OUT, "w"), indent=1)
    print(table(res))


if __name__ == "__main__":
    main()
