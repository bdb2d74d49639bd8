#!/usr/bin/env python3
"""GPU-side tuning sweep: runs bench.py for each (library variant, env) and prints per-kernel microseconds."""
import json, os, subprocess, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
def run(tag, env, extra=()):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-extra", "--steps", "300",
                        "--warmup", "30", *extra], env=e, capture_output=True, text=True)
    if r.returncode != 0:
        print(tag, "FAILED", r.stderr[-400:]); return
    d = json.loads(r.stdout.strip().splitlines()[-1])
    k = d["kernels"]
    print(f"{tag:34s} {d['value']:8.1f} steps/s | " + " | ".join(f"{n} {v['avg_us']:.1f}" for n, v in k.items()), flush=True)
variants = {"spt2": None}
vdir = os.path.join(root, "build_variants")
if os.path.isdir(vdir):
    for f in sorted(os.listdir(vdir)):
        if f.endswith(".so"): variants[f[4:-3]] = os.path.join(vdir, f)
which = sys.argv[1:] or list(variants)
for name in which:
    lib = variants[name]
    base = {"TGNH_LIB": lib} if lib else {}
    for prec in ("mixed", "single"):
        run(f"{name} {prec} auto-grid", base, ("--precision", prec))
    for g in ("512", "1024", "2048"):
        run(f"{name} mixed grid={g}", dict(base, TGNH_GRID=g), ("--precision", "mixed"))
