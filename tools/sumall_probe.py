#!/usr/bin/env python3
"""Row sum by all four wavefronts of the rescale launch (no chain launch) vs the sum launch: tools/sumall_probe.py"""
import json, os, subprocess, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for mol in (125000, 250000, 500000, 1000000):
    for rep in range(2):
        for mode in ("0", "1"):
            e = dict(os.environ, TGNH_INLINE_SUM_ALL=mode)
            big = mol >= 500000
            r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-extra", "--steps", "600" if big else "3000", "--warmup", "100",
                                "--graph", "off" if big else "on", "--molecules", str(mol)], env=e, capture_output=True, text=True)
            if r.returncode: print(mol, mode, "FAILED", r.stderr[-300:]); continue
            d = json.loads(r.stdout.strip().splitlines()[-1])
            print(f"{mol:7d} all-wave sum {'on ' if mode == '1' else 'off'} {d['value']:8.1f} steps/s | " + " | ".join(f"{n} {v['avg_us']:.1f}" for n, v in d["kernels"].items()), flush=True)
