#!/usr/bin/env python3
"""Takes the round's profile evidence in one go, on the GPU box, and stamps it with the hash of the kernel sources:

    python3 tools/profile_round.py r02 [--molecules M] [--variant V]        (run from the repo root; ~2 GPU-minutes)
    cp gpurun_out/profiles_out/r02_* profiles/                              (afterwards, where the repository is tracked)

  1. rocprofv3 --kernel-trace --stats   of `bench.py` (the driver's command shape)   -> <tag>_bench.json,
                                                                                         <tag>_kernel_stats.csv, <tag>_summary.md
  2. rocprofv3 --pmc FETCH_SIZE  and  --pmc WRITE_SIZE  (separate passes, kernel trace only; MI355X_MICROARCH.md "HBM":
     on gfx950 FETCH_SIZE counts half of a wide streaming read, so bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB)
                                                                                      -> <tag>_pmc_traffic.json
  3. rocprofv3 --pmc <SQ counters> (LDS bank conflicts, wait / issue cycles)        -> <tag>_pmc_sq.json

Every file carries `csrc_sha` (openmm_drudenose_amd/build.py::source_sha) and the step variant; bench.py quotes
`roofline.traffic` from <tag>_pmc_traffic.json only when both match the binary and the variant it is running.
This process never touches the GPU: every measurement is a child `rocprofv3 ... -- python3 bench.py ...`.
"""
import argparse
import collections
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OPS = {64: "prekick", 1: "rescale", 2: "kick", 4: "drift", 8: "KE", 16: "posDelta", 32: "move", 128: "unstored"}
PREC = ["single", "mixed", "double"]
STEP_KINDS = ["deferred step", "plain begin half", "plain end half", "split begin half", "split end half"]
SQ = ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
      "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"]


def pretty(n):
    m = re.search(r"tile_kernel<(\d), (\d+), (\d)(, (true|false))?>", n)
    if m:
        ops = int(m.group(2))
        return "tile<%s,%s>%s" % (PREC[int(m.group(1))], "+".join(v for k, v in OPS.items() if ops & k), " (chains of 2-4 links)" if m.group(5) == "true" else "")
    m = re.search(r"wke_kernel<(\d), (\d+), (\d)>", n)
    if m:
        ops = int(m.group(2))
        return "wke<%s,%s>" % (PREC[int(m.group(1))], "+".join(v for k, v in OPS.items() if ops & k))
    m = re.search(r"wstep_kernel<(\d), (\d), (true|false)>", n)
    if m:
        return "wstep_kernel<%s,deferred step>%s" % (PREC[int(m.group(1))], " (chains of 2-4 links)" if m.group(3) == "true" else "")
    m = re.search(r"step_kernel<(\d), (\d), (\d)>", n)
    if m:
        return "step_kernel<%s,%s>" % (PREC[int(m.group(1))], STEP_KINDS[int(m.group(3))])
    return re.sub(r"\(.*", "", n).replace("void ", "").replace("tgnh::", "")[:60]


def run(cmd, log):
    env = dict(os.environ, TMPDIR="/tmp")
    print("+", " ".join(cmd), flush=True)
    with open(log, "w") as f:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=f, env=env, cwd=ROOT)
    if p.returncode != 0:
        raise SystemExit(f"failed ({p.returncode}): see {log}")
    return p.stdout.decode()


def counters(outdir):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "tgnh" in r["Kernel_Name"]:
                acc[pretty(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: (sum(v) / len(v), len(v)) for c, v in cs.items()} for k, cs in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--molecules", type=int, default=1_000_000)
    ap.add_argument("--variant", default="auto")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--skip-sq", action="store_true")
    a = ap.parse_args()
    from openmm_drudenose_amd import build as hip_build
    sha = hip_build.source_sha()
    scratch = os.path.join(ROOT, "gpurun_out", f"prof_{a.tag}")
    shutil.rmtree(scratch, ignore_errors=True)
    os.makedirs(scratch)
    # on the GPU box only gpurun_out/ travels back: the files are written there and copied into profiles/ (tracked) afterwards
    prof = os.path.join(ROOT, "gpurun_out", "profiles_out")
    os.makedirs(prof, exist_ok=True)
    bench = ["python3", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", str(a.steps), "--warmup", str(a.warmup),
             "--molecules", str(a.molecules), "--variant", a.variant]
    lean = bench + ["--no-extra", "--no-cpu-baseline"]

    # 1. kernel trace + stats of the bench command itself
    out = run(["rocprofv3", "--kernel-trace", "--stats", "-d", os.path.join(scratch, "trace"), "--output-format", "csv", "--"] + bench,
              os.path.join(scratch, "trace.err"))
    line = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    variant = line["config"]["variant"]
    slots = line["config"]["slots_per_gpu"]
    stamp = {"csrc_sha": sha, "variant": variant, "slots": slots, "command": " ".join(bench[1:])}
    line["profile_stamp"] = stamp
    json.dump(line, open(os.path.join(prof, f"{a.tag}_bench.json"), "w"), indent=1)
    stats = glob.glob(os.path.join(scratch, "trace", "**", "*_kernel_stats.csv"), recursive=True)[0]
    shutil.copy(stats, os.path.join(prof, f"{a.tag}_kernel_stats.csv"))
    rows = list(csv.DictReader(open(stats)))
    prec = line["config"]["precision"]
    if line["roofline"]["kernel"] in ("step_kernel", "wstep_kernel"):        # (bench.py names the kernel that ran: tgnh_get_resident_kernel)
        dom = f"{line['roofline']['kernel']}<{prec},deferred step>" if variant == "resident" else f"step_kernel<{prec},plain begin half>"
    elif variant == "plain-gather":                          # the gather path (tgnh_gather.hip): its update kernel, all four launches per step of it together
        dom = f"gather_update_kernel<{PREC.index(prec)}>"
    else:
        dom = f"tile<{prec},{'prekick+' if variant == 'defer' else ''}rescale+kick+drift>"
    with open(os.path.join(prof, f"{a.tag}_summary.md"), "w") as f:
        f.write(f"`rocprofv3 --kernel-trace --stats -- {' '.join(bench)}`  \ncsrc_sha `{sha}`, variant `{variant}`, {slots} slots; "
                f"bench line: {line['value']} steps/s, roofline.avg_launch_us {line['roofline']['avg_launch_us']} (HIP events)\n\n")
        f.write("| kernel | calls | avg us | min us | max us | % of GPU time |\n|---|---|---|---|---|---|\n")
        for r in rows:
            if float(r["Percentage"]) < 0.05:
                continue
            f.write(f"| `{pretty(r['Name'])}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | {int(r['MinNs']) / 1e3:.2f} | "
                    f"{int(r['MaxNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} |\n")
    print(open(os.path.join(prof, f"{a.tag}_summary.md")).read())

    # 2. HBM traffic: two PMC passes of the same (lean) command
    res = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(scratch, c)
        run(["rocprofv3", "--pmc", c, "--kernel-trace", "-d", d, "--output-format", "csv", "--"] + lean, os.path.join(scratch, c + ".err"))
        for k, cs in counters(d).items():
            res.setdefault(k, {})[c + "_KiB"] = round(cs[c][0], 1)
            res[k]["launches"] = cs[c][1]
    for k, v in res.items():
        v["hbm_bytes_per_launch"] = int((2 * v.get("FETCH_SIZE_KiB", 0) + v.get("WRITE_SIZE_KiB", 0)) * 1024)
    res["dominant"] = dict(res[dom], kernel=dom)
    json.dump(dict(stamp, note="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes with --kernel-trace only; raw counters in KiB; "
                               "gfx950: FETCH_SIZE counts half of a wide streaming read (MI355X_MICROARCH.md, HBM) -> "
                               "hbm_bytes = (2 FETCH_SIZE + WRITE_SIZE) * 1024; Infinity-Cache hits are counted, not excluded",
                   kernels=res), open(os.path.join(prof, f"{a.tag}_pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(res["dominant"]))

    # 3. SQ counters (one pass, 8 slots)
    if not a.skip_sq:
        d = os.path.join(scratch, "sq")
        run(["rocprofv3", "--pmc"] + SQ + ["--kernel-trace", "-d", d, "--output-format", "csv", "--"] + lean, os.path.join(scratch, "sq.err"))
        sq = {}
        for k, cs in counters(d).items():
            e = {c: round(cs[c][0], 1) for c in cs}
            if e.get("SQ_LDS_IDX_ACTIVE"):
                e["lds_bank_conflict_frac"] = round(e.get("SQ_LDS_BANK_CONFLICT", 0.0) / e["SQ_LDS_IDX_ACTIVE"], 4)
            if e.get("SQ_WAVE_CYCLES"):
                e["wait_any_frac"] = round(e.get("SQ_WAIT_ANY", 0.0) / e["SQ_WAVE_CYCLES"], 4)
                e["active_valu_frac"] = round(e.get("SQ_ACTIVE_INST_VALU", 0.0) / e["SQ_WAVE_CYCLES"], 4)
            sq[k] = e
        json.dump(dict(stamp, note="rocprofv3 --pmc " + " ".join(SQ) + " (one pass, --kernel-trace only), per-launch averages", kernels=sq),
                  open(os.path.join(prof, f"{a.tag}_pmc_sq.json"), "w"), indent=1)
        print(json.dumps({k: {c: v for c, v in e.items() if c.endswith("frac")} for k, e in sq.items()}, indent=1))


if __name__ == "__main__":
    main()
