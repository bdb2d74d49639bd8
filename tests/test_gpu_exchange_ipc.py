"""The mailbox exchange across PROCESSES: two ranks, one process each, both on this one GPU, every rank's mailbox
mapped into the other by a hipIpc handle -- what `bench.py --gpus N` does with one GPU per rank.  (Real multi-GPU
runs need the driver's 8-GPU node.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

from openmm_drudenose_amd import synth
from openmm_drudenose_amd.drudetgnhplugin import HipContext, FLAG_DEFER_SCALE
from helpers import rel_err

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
NSTEPS = 120


@pytest.mark.parametrize("variant,lag", [(FLAG_DEFER_SCALE, 0.0), (0, 0.0), (FLAG_DEFER_SCALE, 0.03)])
def test_two_processes_exchange_by_ipc_mailboxes(tmp_path, variant, lag):
    """lag > 0: rank 1 idles that long after every 50 steps, so rank 0's launches really wait in their spin loops (and
    run ahead by the one exchange the two mailbox parities allow)."""
    sys.path.insert(0, HERE)
    from xchg_worker import make_integrator
    out = str(tmp_path)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", TGNH_XW_LAG=str(lag))
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "xchg_worker.py"), str(r), "2", out, str(NSTEPS), str(variant)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{logs[r][-3000:]}"
    s, g, ng = synth.mixed(400, 30)
    ref = HipContext(s, make_integrator(g, ng), mode="TGNH", precision="double", flags=variant)
    ref.step(NSTEPS)
    flags = [int(np.load(os.path.join(out, f"flags{r}.npy"))[0]) for r in range(2)]
    assert all(f & 4 == 0 for f in flags), f"exchange time-out, flags {flags}"
    pos = np.concatenate([np.load(os.path.join(out, f"pos{r}.npy")) for r in range(2)])
    vel = np.concatenate([np.load(os.path.join(out, f"vel{r}.npy")) for r in range(2)])
    assert rel_err(pos, ref.getPositions()) < 1e-12 and rel_err(vel, ref.getVelocities()) < 1e-10
    eta = [np.load(os.path.join(out, f"eta{r}.npy")) for r in range(2)]
    assert np.array_equal(eta[0], eta[1])                       # rank-order sums: bit-identical thermostats
    assert np.allclose(eta[0], np.concatenate([ref.thermostat_state(0), ref.thermostat_state(1)]), rtol=1e-9, atol=1e-13)
    ref.close()
