#!/usr/bin/env python3
"""GPU-side A/B of library variants on ONE box: tools/ab.py [bench args...]  (variants = build_variants/*.so + the main lib)."""
import json, os, subprocess, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
variants = {"main": None}
vdir = os.path.join(root, "build_variants")
if os.path.isdir(vdir):
    for f in sorted(os.listdir(vdir)):
        if f.endswith(".so"): variants[f[4:-3]] = os.path.join(vdir, f)
for rep in range(2):
    for name, lib in variants.items():
        e = dict(os.environ)
        if lib: e["TGNH_LIB"] = lib
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-extra", "--steps", "300", "--graph", "off", *sys.argv[1:]],
                           env=e, capture_output=True, text=True)
        if r.returncode: print(name, "FAILED", r.stderr[-300:]); continue
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print(f"{name:10s} {d['value']:8.1f} steps/s | " + " | ".join(f"{n} {v['avg_us']:.1f}" for n, v in d["kernels"].items()), flush=True)
