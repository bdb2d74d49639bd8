#!/usr/bin/env python3
"""Tile-cap probe at shard sizes: tools/cap_probe.py  (TGNH_TILE_CAP override; hipGraph replay, no communication)."""
import json, os, subprocess, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
cases = {125000: [0, 410, 415, 275, 280, 210], 250000: [0, 410, 415, 330, 275], 500000: [0, 470, 410], 1000000: [0, 505, 470]}
for mol, caps in cases.items():
    for c in caps:
        e = dict(os.environ)
        if c: e["TGNH_TILE_CAP"] = str(c)
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-extra", "--steps", "2000" if mol < 1000000 else "500", "--warmup", "200",
                            "--graph", "on", "--molecules", str(mol), *sys.argv[1:]], env=e, capture_output=True, text=True)
        if r.returncode: print(mol, c, "FAILED", r.stderr[-300:]); continue
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print(f"{mol:7d} cap {c or 512:>5} {d['value']:8.1f} steps/s | " + " | ".join(f"{n} {v['avg_us']:.1f}" for n, v in d["kernels"].items()), flush=True)
