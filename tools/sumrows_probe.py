#!/usr/bin/env python3
"""Where does summing the partial rows inside the rescale launch stop paying?  tools/sumrows_probe.py  (TGNH_INLINE_SUM_ROWS)"""
import json, os, subprocess, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for mol in (6400, 25000, 50000, 75000, 100000, 125000):
    for rep in range(2):
        for rows in ("0", "4096"):
            e = dict(os.environ, TGNH_INLINE_SUM_ROWS=rows)
            r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-extra", "--steps", "3000", "--warmup", "200",
                                "--graph", "on", "--molecules", str(mol)], env=e, capture_output=True, text=True)
            if r.returncode: print(mol, rows, "FAILED", r.stderr[-300:]); continue
            d = json.loads(r.stdout.strip().splitlines()[-1])
            print(f"{mol:7d} ({(mol * 5 + 509) // 510:5d} rows) in-launch sum {'on ' if rows != '0' else 'off'} {d['value']:8.1f} steps/s | " + " | ".join(f"{n} {v['avg_us']:.1f}" for n, v in d["kernels"].items()), flush=True)
