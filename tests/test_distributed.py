"""world_size-2 gloo test (CPU) of the particle-sharded algorithm: whole-molecule slabs, additive dof terms,
one all-reduce of the (G+2) kinetic-energy sums per thermostat half step, chain replicated on every rank.
The per-particle arithmetic here is the oracle's (no GPU in this container); what is under test is that the
sharded schedule the HIP path uses reproduces the unsharded trajectory, and the library's own dof bookkeeping
across ranks (host-only handles)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from openmm_drudenose_amd import synth, DrudeTGNHIntegrator, HostTopology
from openmm_drudenose_amd.system import shard_bounds
from oracle import Oracle, MODE_TGNH

NSTEPS = 25


def _integ(g, ng):
    it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001, 20, 2, True, True)
    it.setMaxDrudeDistance(0.02)
    for _ in range(ng):
        it.addTempGroup()
    for x in g:
        it.addParticleTempGroup(int(x))
    return it


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s, g, ng = synth.mixed(90, 7)
        b = shard_bounds(s, world)
        lo, hi = b[rank], b[rank + 1]
        loc, lg = s.slice_molecules(lo, hi), g[lo:hi]
        # (1) the library's dof terms are additive over ranks
        full = HostTopology(s, _integ(g, ng))
        part = HostTopology(loc, _integ(lg, ng))
        t = torch.from_numpy(part.local_dof_terms())
        dist.all_reduce(t)
        assert np.allclose(t.numpy(), full.local_dof_terms(), rtol=1e-13), (t, full.local_dof_terms())
        part.set_global_dof_terms(t.numpy())
        assert np.allclose(part.dof()[0], full.dof()[0], rtol=1e-13) and np.allclose(part.dof()[1], full.dof()[1], rtol=1e-13)
        assert np.allclose(part.thermostat_state(3), full.thermostat_state(3), rtol=1e-13)
        # (2) sharded schedule: local per-particle work, all-reduced KE, replicated chain
        it = _integ(lg, ng)
        o_loc = Oracle.from_integrator(loc, it, lg, ng, MODE_TGNH)            # per-particle ops on the slab
        o_glob = Oracle.from_integrator(s, _integ(g, ng), g, ng, MODE_TGNH)    # holds the replicated thermostat
        pos, vel, x0 = loc.positions.copy(), loc.velocities.copy(), loc.positions.copy()
        f = o_loc.harness_force(pos, x0, synth.K_DRUDE, synth.K_TETHER)

        def thermostat_half():
            ke = torch.from_numpy(o_loc.kinetic_energies(vel))
            dist.all_reduce(ke)                                               # the only exchange of the path
            sc = o_glob.chain_only(ke.numpy())
            o_loc.scale_velocities(vel, sc)
        for _ in range(NSTEPS):
            thermostat_half()
            o_loc.half_kick(vel, f); o_loc.drift(pos, vel); o_loc.hardwall(pos, vel)
            f = o_loc.harness_force(pos, x0, synth.K_DRUDE, synth.K_TETHER)
            o_loc.half_kick(vel, f)
            thermostat_half()
        np.save(os.path.join(out, f"pos{rank}.npy"), pos)
        np.save(os.path.join(out, f"vel{rank}.npy"), vel)
        np.save(os.path.join(out, f"eta{rank}.npy"), o_glob.chain(1))
    finally:
        dist.destroy_process_group()


def test_sharded_schedule_reproduces_unsharded_run(tmp_path):
    world = 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    s, g, ng = synth.mixed(90, 7)
    o = Oracle.from_integrator(s, _integ(g, ng), g, ng, MODE_TGNH)
    pos, vel, x0 = s.positions.copy(), s.velocities.copy(), s.positions.copy()
    f = o.harness_force(pos, x0, synth.K_DRUDE, synth.K_TETHER)
    o.run_harness(pos, vel, f, x0, synth.K_DRUDE, synth.K_TETHER, NSTEPS)
    pos_s = np.concatenate([np.load(tmp_path / f"pos{r}.npy") for r in range(world)])
    vel_s = np.concatenate([np.load(tmp_path / f"vel{r}.npy") for r in range(world)])
    assert np.abs(pos_s - pos).max() / np.abs(pos).max() < 1e-12
    assert np.abs(vel_s - vel).max() / np.abs(vel).max() < 1e-10
    e0, e1 = np.load(tmp_path / "eta0.npy"), np.load(tmp_path / "eta1.npy")
    assert np.array_equal(e0, e1)                               # replicated chain: identical on every rank
    assert np.allclose(e0, o.chain(1), rtol=1e-9, atol=1e-13)
