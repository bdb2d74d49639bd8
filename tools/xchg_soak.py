#!/usr/bin/env python3
"""Soak of the two-process mailbox exchange on one GPU: tools/xchg_soak.py [steps] [waters]  (both ranks on cuda:0;
bit-identical thermostats and no time-out after many thousand exchanges; boxes small enough that the two processes'
grids fit the GPU together)."""
import os, subprocess, sys, tempfile
import numpy as np
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
steps = sys.argv[1] if len(sys.argv) > 1 else "20000"
waters = sys.argv[2] if len(sys.argv) > 2 else "20000"
lag = sys.argv[3] if len(sys.argv) > 3 else "0"          # seconds rank 1 idles every 50 steps (rank 0 waits in its kernels)
for variant in ("2", "0"):
    out = tempfile.mkdtemp()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", TGNH_XW_WATERS=waters, TGNH_XW_PAIRS="500", TGNH_XW_LAG=lag)
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "xchg_worker.py"), str(r), "2", out, steps, variant], env=env) for r in range(2)]
    rcs = [p.wait() for p in procs]
    flags = [int(np.load(os.path.join(out, f"flags{r}.npy"))[0]) for r in range(2)]
    eta = [np.load(os.path.join(out, f"eta{r}.npy")) for r in range(2)]
    vel = [np.load(os.path.join(out, f"vel{r}.npy")) for r in range(2)]
    print(f"variant {variant}: rc {rcs} flags {flags} thermostats identical {np.array_equal(eta[0], eta[1])} finite {all(np.isfinite(v).all() for v in vel)} etaDot {eta[0][len(eta[0]) // 2:][:4]}", flush=True)
