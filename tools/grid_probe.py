#!/usr/bin/env python3
"""Persistent-grid probe at shard sizes: tools/grid_probe.py  (TGNH_GRID override; hipGraph replay, no communication)."""
import json, os, subprocess, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
cases = {125000: [0, 611, 814, 1221, 1024, 768], 250000: [0, 814, 1221, 1024, 2048], 500000: [0, 977, 1024, 1221, 1628]}
for mol, grids in cases.items():
    for g in grids:
        e = dict(os.environ)
        if g: e["TGNH_GRID"] = str(g)
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-extra", "--steps", "2000", "--warmup", "200",
                            "--graph", "on", "--molecules", str(mol), *sys.argv[1:]], env=e, capture_output=True, text=True)
        if r.returncode: print(mol, g, "FAILED", r.stderr[-300:]); continue
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print(f"{mol:7d} grid {g or 'auto':>5} {d['value']:8.1f} steps/s | " + " | ".join(f"{n} {v['avg_us']:.1f}" for n, v in d["kernels"].items()), flush=True)
