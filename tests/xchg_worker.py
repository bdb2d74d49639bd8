"""One rank of the two-process mailbox-exchange test (tests/test_gpu_exchange_ipc.py): both ranks on cuda:0, the
mailboxes mapped into each other by hipIpc handles passed through files in `out`.
usage: python xchg_worker.py RANK WORLD OUT_DIR NSTEPS VARIANT"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from openmm_drudenose_amd import synth                                                  # noqa: E402
from openmm_drudenose_amd.drudetgnhplugin import DrudeTGNHIntegrator, HipContext      # noqa: E402
from openmm_drudenose_amd.system import shard_bounds                                    # noqa: E402


def make_integrator(g, ng, chains=1):
    it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001, 20, chains, True, True)
    it.setMaxDrudeDistance(0.02)
    for _ in range(ng):
        it.addTempGroup()
    for x in g:
        it.addParticleTempGroup(int(x))
    return it


def wait_for(path, timeout=120.0):
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > timeout:
            raise TimeoutError(path)
        time.sleep(0.01)
    return path


def publish(path, arr):
    np.save(path + ".tmp.npy", arr)
    os.replace(path + ".tmp.npy", path)


def main():
    rank, world, out, nsteps, variant = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    n_water, n_pairs = int(os.environ.get("TGNH_XW_WATERS", "400")), int(os.environ.get("TGNH_XW_PAIRS", "30"))   # soak runs: bigger boxes
    s, g, ng = synth.mixed(n_water, n_pairs)
    b = shard_bounds(s, world)
    loc, lg = s.slice_molecules(b[rank], b[rank + 1]), g[b[rank]:b[rank + 1]]
    ctx = HipContext(loc, make_integrator(lg, ng), mode="TGNH", precision="double", flags=variant)
    handle, _ = ctx.exchange_create(world, rank)
    publish(os.path.join(out, f"handle{rank}.npy"), np.frombuffer(handle, np.uint8))
    publish(os.path.join(out, f"dof{rank}.npy"), ctx.local_dof_terms())
    handles = [np.load(wait_for(os.path.join(out, f"handle{r}.npy"))).tobytes() for r in range(world)]
    total = sum(np.load(wait_for(os.path.join(out, f"dof{r}.npy"))) for r in range(world))
    ctx.set_global_dof_terms(total)
    ctx.exchange_attach(handles)
    publish(os.path.join(out, f"ready{rank}.npy"), np.zeros(1))
    for r in range(world):
        wait_for(os.path.join(out, f"ready{r}.npy"))
    lag = float(os.environ.get("TGNH_XW_LAG", "0"))          # soak runs: odd ranks fall behind by `lag` s every 50 steps
    if lag > 0:
        for done in range(0, nsteps, 50):
            ctx.step(min(50, nsteps - done))
            if rank % 2:
                ctx.torch.cuda.synchronize()
                time.sleep(lag)
    else:
        ctx.step(nsteps)
    ctx.torch.cuda.synchronize()
    flags = ctx.check()
    publish(os.path.join(out, f"pos{rank}.npy"), ctx.getPositions())
    publish(os.path.join(out, f"vel{rank}.npy"), ctx.getVelocities())
    publish(os.path.join(out, f"eta{rank}.npy"), np.concatenate([ctx.thermostat_state(0), ctx.thermostat_state(1)]))
    publish(os.path.join(out, f"flags{rank}.npy"), np.array([flags]))
    # unmap the peers' mailboxes; free the own one only when every peer has unmapped it
    ctx.exchange_detach()
    publish(os.path.join(out, f"done{rank}.npy"), np.zeros(1))
    for r in range(world):
        wait_for(os.path.join(out, f"done{r}.npy"))
    ctx.close()


if __name__ == "__main__":
    main()
