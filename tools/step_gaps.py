#!/usr/bin/env python3
"""Where a step's time goes BETWEEN its kernels: kernel durations and the gaps between consecutive kernels of a hipGraph-replayed
step loop, by the profiler's own clocks (rocprofv3 --kernel-trace; HIP events bracket a launch and include the dispatch).

    python3 tools/step_gaps.py <tag> [--molecules 125000] [--variant resident] [--dist]     (GPU box, repo root)
        -> gpurun_out/profiles_out/<tag>_step_gaps_<molecules>_<variant>[_rccl].md

--dist: TGNH_FORCE_DIST=1 -- the RCCL launch structure on one rank (row sum, ncclAllReduce, the chain inside the next launch).
This process never touches the GPU: the measurement is a child `rocprofv3 ... -- python3 bench.py ...`."""
import argparse
import collections
import csv
import glob
import os
import shutil
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--molecules", type=int, default=125000)
    ap.add_argument("--variant", default="resident")
    ap.add_argument("--dist", action="store_true")
    ap.add_argument("--parse-only", action="store_true", help="parse the trace an earlier run left under gpurun_out/")
    a = ap.parse_args()
    from profile_round import pretty
    name = f"{a.tag}_step_gaps_{a.molecules}_{a.variant}{'_rccl' if a.dist else ''}"
    scratch = os.path.join(ROOT, "gpurun_out", "prof_" + name)
    import json
    if a.parse_only:                                     # the trace a GPU run left (gpurun merges gpurun_out/ back): parse it here
        bench = json.load(open(os.path.join(scratch, "bench_line.json")))
    else:
        shutil.rmtree(scratch, ignore_errors=True)
        os.makedirs(scratch)
        cmd = ["rocprofv3", "--kernel-trace", "-d", scratch, "--output-format", "csv", "--", "python3", os.path.join(ROOT, "bench.py"),
               "--molecules", str(a.molecules), "--variant", a.variant, "--graph", "on", "--steps", "2000", "--warmup", "100", "--no-extra", "--no-cpu-baseline"]
        env = dict(os.environ, TMPDIR="/tmp")
        if a.dist:
            env["TGNH_FORCE_DIST"] = "1"
        print("+", " ".join(cmd), flush=True)
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=open(os.path.join(scratch, "err.log"), "w"), env=env, cwd=ROOT)
        if p.returncode != 0:
            raise SystemExit(f"failed ({p.returncode}): {scratch}/err.log")
        line = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][-1]
        open(os.path.join(scratch, "bench_line.json"), "w").write(line)
        bench = json.loads(line)
    rows = []
    for f in glob.glob(os.path.join(scratch, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # the timed region: the longest run of launches whose names repeat with the period of one step (the graph replays of the
    # step loop; warm-up, the integrator-only leg and the instrumented repeat have other periods or are shorter)
    names = [pretty(n) for _, _, n in rows]
    best = (0, 0, 1)
    for per in range(1, 9):
        i = 0
        while i < len(names) - per:
            j = i
            while j < len(names) - per and names[j] == names[j + per]:
                j += 1
            if j - i > best[0]:
                best = (j - i, i, per)
            i = j + 1
        if best[0] > 1500:
            break
    length, start, period = best
    start, stop = start + length // 10, start + length - length // 10
    dur = collections.defaultdict(list)
    gap = collections.defaultdict(list)
    step = []
    for i in range(start, stop):
        s0, e0, _ = rows[i]
        dur[names[i]].append((e0 - s0) / 1e3)
        gap[f"{names[i]} -> {names[i + 1]}"].append((rows[i + 1][0] - e0) / 1e3)
        step.append((rows[i + period][0] - s0) / 1e3)
    out = [f"# Kernels and the gaps between them: {a.molecules} molecules, `{a.variant}`{', RCCL launch structure on one rank (TGNH_FORCE_DIST=1)' if a.dist else ''}, hipGraph replay ({a.tag})\n\n",
           f"`rocprofv3 --kernel-trace` over `bench.py --molecules {a.molecules} --variant {a.variant} --graph on --steps 2000`; the bench line of the same run: "
           f"{bench['value']} steps/s ({1e6 / bench['value']:.2f} us per step; under the profiler), kernel `{bench['roofline']['kernel']}`.  "
           f"Times by the profiler's clocks, medians over {len(step)} launches of the steady state; one step = {period} launches.\n\n",
           "| | median us | p10 | p90 |\n|---|---|---|---|\n"]
    q = lambda v, f: statistics.quantiles(v, n=10)[f]
    tot_k = tot_g = 0.0
    for k, v in dur.items():
        out.append(f"| kernel `{k}` | {statistics.median(v):.2f} | {q(v, 0):.2f} | {q(v, 8):.2f} |\n")
        tot_k += statistics.median(v)
    for k, v in gap.items():
        out.append(f"| gap {k} | {statistics.median(v):.2f} | {q(v, 0):.2f} | {q(v, 8):.2f} |\n")
        tot_g += statistics.median(v)
    out.append(f"| **step, start to start** | **{statistics.median(step):.2f}** | {q(step, 0):.2f} | {q(step, 8):.2f} |\n")
    out.append(f"| = kernels {tot_k:.2f} + gaps {tot_g:.2f} | | | |\n")
    prof = os.path.join(ROOT, "gpurun_out", "profiles_out")
    os.makedirs(prof, exist_ok=True)
    open(os.path.join(prof, name + ".md"), "w").write("".join(out))
    print("".join(out))


if __name__ == "__main__":
    main()
