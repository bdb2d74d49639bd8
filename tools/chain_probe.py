#!/usr/bin/env python3
"""GPU-side probe: chain kernel time vs drudeStepsPerRealStep and chain length (fixed cost vs per-iteration cost)."""
import json, os, subprocess, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for chains in (1, 3):
    for S in (1, 5, 20, 80):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-extra", "--steps", "200",
                            "--warmup", "20", "--molecules", "100000", "--drude-steps", str(S), "--chains", str(chains)],
                           capture_output=True, text=True)
        if r.returncode: print("FAILED", r.stderr[-300:]); continue
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print(f"chains {chains} S {S:3d}: chain {d['kernels']['chain']['avg_us']:.2f} us | " +
              " | ".join(f"{n} {v['avg_us']:.1f}" for n, v in d['kernels'].items() if n != 'chain'), flush=True)
