#!/usr/bin/env python3
"""Per-launch table of the CONSTRAINED step (the split path around OpenMM's constraint call-outs, Cu :336-406): every launch of a
step with its algorithmic bytes and rate, the integrator's own launches apart from the harness' stand-ins for the call-outs
(SHAKE on posDelta, the velocity stage, virtual sites, the force).  On the GPU box, from the repo root:

    python3 tools/constrained_table.py r05                  -> gpurun_out/profiles_out/r05_constrained.md

For each system (BASELINE config 3 as worded: the ionic liquid at 100 k slots with its 33 330 X-H constraints; rigid SWM4 water
at 2 M slots = BASELINE config 5's size with the reference test's own constraints and M site) and each pass structure (flags 0
= what the OpenMM glue runs; TGNH_FLAG_RESIDENT_STEP = each thermostat half as one launch) a child process steps the system
under `rocprofv3 --kernel-trace --stats`; this process never touches the GPU.
"""
import csv
import glob
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

SYSTEMS = {
    "C3-il-100k-shake": ("BASELINE config 3 as worded: [BMIM][BF4] 99 990 slots, 2 temperature groups, 33 330 X-H constraints", "ionic_liquid", 2222),
    "water-2M-rigid": ("rigid SWM4 water, 2 000 000 slots (config 5's size), 3 constraints + M site per molecule (TestReference...:145-148)", "water", 400000),
}
STEPS = 200


def child(sysname, variant):
    import torch
    from openmm_drudenose_amd import synth, DrudeTGNHIntegrator, HipContext
    from openmm_drudenose_amd.drudetgnhplugin import FLAG_RESIDENT_STEP, FLAG_TRUST_STATE_CHANGED
    kind, n = SYSTEMS[sysname][1:]
    s, g, ng = synth.ionic_liquid(n, constrained=True) if kind == "ionic_liquid" else synth.water_box(n, rigid=True)
    it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001, 20, 1, True, True)
    it.setMaxDrudeDistance(0.02)
    for _ in range(ng):
        it.addTempGroup()
    it._particleTempGroup = g.astype("int32")
    flags = {"plain": 0, "plain-resident": FLAG_RESIDENT_STEP, "plain-trust": FLAG_TRUST_STATE_CHANGED}[variant]
    ctx = HipContext(s, it, mode="TGNH", precision="mixed", flags=flags)
    assert ctx.constrained
    ctx.step(20)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    ctx.step(STEPS)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"RESULT {sysname} {variant} slots={s.num_particles} steps_per_s={STEPS / dt:.1f} status={ctx.status_flags()}", flush=True)
    ctx.close()


def bytes_per_slot(name):
    """algorithmic state bytes per slot of one launch (mixed precision: V 32, F 24, X 16 + 16, posDelta 32), by kernel name"""
    V, F, X, PD = 32, 24, 32, 32
    m = re.search(r"(tile|wke)_kernel<(\d), (\d+), ", name)
    if m:
        ops, b = int(m.group(3)), 0
        if ops & 64 or ops & 2: b += F                       # (pre)kick: F r
        if ops & (1 | 2 | 32 | 64) and not ops & 128: b += 2 * V     # velocities read and written
        elif ops & (8 | 1 | 2 | 64): b += V                  # read only (KE; an unstored kick)
        if ops & 4: b += 2 * X                               # drift: X r/w
        if ops & 16: b += PD                                 # posDelta w
        if ops & 32: b += 2 * X + PD                         # move: X r/w, posDelta r (its velocities: counted above)
        return b
    m = re.search(r"step_kernel<(\d), (\d), (\d)>", name)
    if m:
        return {0: 3 * V + 2 * F + 2 * X, 1: 3 * V + F + 2 * X, 2: 3 * V + 2 * F, 3: 3 * V + F + PD, 4: 3 * V}[int(m.group(3))]
    return None


def main():
    tag = sys.argv[1]
    from profile_round import pretty
    from openmm_drudenose_amd import build as hip_build
    prof = os.path.join(ROOT, "gpurun_out", "profiles_out")
    os.makedirs(prof, exist_ok=True)
    out = [f"# The constrained step, launch by launch ({tag}; csrc_sha `{hip_build.source_sha()}`)\n",
           "`tools/constrained_table.py`: `rocprofv3 --kernel-trace --stats` over %d steps of each system and pass structure, mixed "
           "precision, one-link chains, hard wall 0.02 nm, eager launches.  Algorithmic MB = state arrays only (SURVEY 8d: V 32, F 24, "
           "X 16 + 16, posDelta 32 B per slot); harness kernels are this repository's stand-ins for OpenMM's call-outs "
           "(applyConstraints Cu :363, applyVelocityConstraints :391, computeVirtualSites :377, calcForcesAndEnergy :380) and are "
           "listed apart.\n" % STEPS]
    for sysname, (title, _, _) in SYSTEMS.items():
        for variant in ("plain", "plain-resident"):
            scratch = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_constrained", f"{sysname}_{variant}")
            shutil.rmtree(scratch, ignore_errors=True)
            os.makedirs(scratch)
            cmd = ["rocprofv3", "--kernel-trace", "--stats", "-d", scratch, "--output-format", "csv", "--",
                   "python3", os.path.abspath(__file__), "--child", sysname, variant]
            print("+", " ".join(cmd), flush=True)
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=open(os.path.join(scratch, "err.log"), "w"), env=dict(os.environ, TMPDIR="/tmp"), cwd=ROOT)
            res = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("RESULT")]
            if p.returncode != 0 or not res:
                raise SystemExit(f"child failed ({p.returncode}): {scratch}/err.log")
            slots = int(re.search(r"slots=(\d+)", res[0]).group(1))
            rate = float(re.search(r"steps_per_s=([\d.]+)", res[0]).group(1))
            stats = glob.glob(os.path.join(scratch, "**", "*_kernel_stats.csv"), recursive=True)[0]
            rows = [r for r in csv.DictReader(open(stats)) if "tgnh" in r["Name"]]
            total = STEPS + 20
            own, harness = [], []
            for r in rows:
                calls, avg = int(r["Calls"]), float(r["AverageNs"]) / 1e3
                per_step = calls / total
                if per_step < 0.4:                           # create-time / one-off launches
                    continue
                b = bytes_per_slot(r["Name"])
                (own if b is not None or "chain" in r["Name"] or "rowsum" in r["Name"] else harness).append((pretty(r["Name"]), per_step, avg, b))
            out.append(f"\n## {title} -- `{variant}` ({'flags 0: what the OpenMM glue runs' if variant == 'plain' else 'TGNH_FLAG_RESIDENT_STEP: each thermostat half one launch'}), "
                       f"{rate:.0f} steps/s under the profiler\n")
            out.append("| launch | per step | avg us | algorithmic MB | GB/s | of 8 TB/s |\n|---|---|---|---|---|---|\n")
            t_own = t_h = mb_own = 0.0
            for name, per, avg, b in sorted(own, key=lambda x: -x[1] * x[2]):
                if b:
                    mb = b * slots / 1e6
                    out.append(f"| `{name}` | {per:.2f} | {avg:.1f} | {mb:.1f} | {mb / avg * 1e3:.0f} | {mb / avg * 1e3 / 8000:.2f} |\n")
                    mb_own += mb * per
                else:
                    out.append(f"| `{name}` | {per:.2f} | {avg:.1f} | -- | -- | latency |\n")
                t_own += per * avg
            out.append(f"| **the integrator's own launches** | | **{t_own:.1f} per step** | {mb_own:.1f} | {mb_own / t_own * 1e3:.0f} | **{mb_own / t_own * 1e3 / 8000:.2f}** |\n")
            for name, per, avg, b in sorted(harness, key=lambda x: -x[1] * x[2]):
                out.append(f"| harness: `{name}` | {per:.2f} | {avg:.1f} | -- | -- | call-out stand-in |\n")
                t_h += per * avg
            out.append(f"| harness call-outs together | | {t_h:.1f} per step | | | |\n")
    path = os.path.join(prof, f"{tag}_constrained.md")
    open(path, "w").write("".join(out))
    print(open(path).read())


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "--child":
        child(sys.argv[2], sys.argv[3])
    else:
        main()
