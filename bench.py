#!/usr/bin/env python3
"""bench.py -- integrator steps/s at 1 M Drude pairs (BASELINE.json metric) on N MI355X GPUs.

A "step" is one full DrudeTGNHIntegrator time step (thermostat half step, rescale, half kick,
drift, [force call-out], half kick, thermostat half step, rescale) over the whole synthetic
SWM4-NDP water box of 1,000,000 molecules (5,000,000 particle slots, 1,000,000 Drude pairs),
state resident in HBM.  The force call-out (OpenMM's calcForcesAndEnergy in a real context) is
the harness spring kernel and is INSIDE the timed region.  N > 1 shards whole molecules over the
ranks (strong scaling: the 1 M-pair system is fixed) with one all-reduce of the per-thermostat
kinetic-energy sums per thermostat half step.  The headline value of a sharded run uses RCCL
(the library's own ncclAllReduce on its device buffer, captured into the step's hipGraph; torch.distributed's
all_reduce through the hook where that cannot be had), as
BASELINE.json's north_star names it; the library's mailbox exchange (stores into every peer's mailbox
over xGMI, waited for inside the rescale launch) is measured right after it and reported beside it as
`extra.mailbox`, together with its validation verdict against the RCCL run.

`python bench.py --gpus N` from a plain shell starts its own N ranks (one per GPU) through
torch.distributed.run before anything touches a GPU; under a launcher (WORLD_SIZE set) it is one of the ranks.

Prints ONE JSON line on rank 0.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Rehearsal knobs (not for measurements): TGNH_BENCH_BACKEND=gloo with TGNH_BENCH_DEVICE=0 runs several ranks of a
# sharded bench on ONE GPU, to exercise the multi-process flow (sharding, mailbox attach, lockstep graph replays).
BACKEND = os.environ.get("TGNH_BENCH_BACKEND", "nccl")
CDEV = "cuda" if BACKEND == "nccl" else "cpu"          # where the small tensors of the setup collectives live

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md; ~6.3 TB/s measured achievable)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=1000)
    p.add_argument("--warmup", type=int, default=100)
    p.add_argument("--molecules", type=int, default=1_000_000, help="SWM4 molecules = Drude pairs (metric: 1,000,000)")
    p.add_argument("--precision", default="mixed", choices=["single", "mixed", "double"])
    p.add_argument("--mode", default="TGNH", choices=["TGNH", "dualNH"])
    p.add_argument("--variant", default="auto", choices=["auto", "plain", "plain-trust", "plain-resident", "plain-resident-trust", "defer", "resident", "plain-gather"],
                   help="defer = end-of-step rescale and second half kick folded into the next step's first pass; resident = "
                        "defer with the whole step in ONE launch whose work-groups meet on the device (step_kernel; with the "
                        "RCCL hook it steps the defer way); plain = the reference's pass structure; plain-trust = that structure with "
                        "TGNH_FLAG_TRUST_STATE_CHANGED (the begin half starts its chain from the kinetic energies the last end half "
                        "left: no KE pass; an OPTION of the OpenMM glue, off by default: it needs an accessor the reference's API class "
                        "does not have and a System of known Force types only, INTEGRATION.md section 3); "
                        "plain-resident = the plain structure with each thermostat half one step_kernel launch; plain-resident-trust = both; "
                        "plain-gather = the reference's structure on the GATHER path (TGNH_FLAG_GATHER: the reference's un-fused kernels by global "
                        "index, what the library takes for topologies its tiles cannot hold; for comparison); "
                        "auto = resident (single precision from 3 M slots per GPU: defer) (DESIGN.md)")
    p.add_argument("--chains", type=int, default=1)
    p.add_argument("--drude-steps", type=int, default=20, help="drudeStepsPerRealStep (reference default 20)")
    p.add_argument("--hardwall", type=float, default=0.02, help="maxDrudeDistance nm (example/nacl_tg.py:22); 0 = off")
    p.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg (rank 0, N=1)")
    p.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                   help="replay the step loop from a hipGraph (auto: on when sharded over >1 GPU, where launches are short)")
    p.add_argument("--graph-steps", type=int, default=50,
                   help="time steps per captured graph (at most --steps).  A graph launch costs ~8 us of idle GPU between two replays "
                        "(profiles/r05_step_gaps_*.md): 0.8 us per step at 10 steps per graph -- 2 %% of a 40 us step at the 8-GPU shard size --, 0.17 at 50")
    p.add_argument("--exchange", default="rccl", choices=["rccl", "mailbox"],
                   help="KE all-reduce of the HEADLINE sharded run: rccl = torch.distributed all_reduce (north_star); "
                        "mailbox = tgnh_exchange_* (stores over xGMI).  The other one is measured as an extra leg")
    p.add_argument("--dry-launch", action="store_true",
                   help="launcher check only: the ranks rendezvous (gloo), count themselves and rank 0 prints the count; no GPU work")
    p.add_argument("--side-leg", action="store_true",
                   help="internal: this process is a child rank that measures the non-headline exchange of a sharded run "
                        "(--exchange names it) and prints its record; started by the ranks of the main run, see side_leg_in_children")
    p.add_argument("--side-leg-timeout", type=float, default=180.0, help="seconds the main run gives its side-leg children")
    p.add_argument("--probe-rccl-site", action="store_true",
                   help="internal: this process is a child rank that tries the library's own RCCL site (tgnh_rccl_init: a communicator "
                        "of its own, ncclAllReduce enqueued by the library) on a small sharded box and prints whether it worked; started "
                        "by the ranks of a sharded run before they commit to it, see probe_rccl_site_in_children")
    p.add_argument("--probe-timeout", type=float, default=90.0, help="seconds the main run gives the RCCL-site probe")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-extra", action="store_true", help="skip the extra single-precision / variant legs")
    return p.parse_args()


# Whether the ranks of a sharded run may use the library's own RCCL site (tgnh_rccl_init): None = not probed (one rank: a one-rank
# communicator is what the GPU suite tests), True / False = what probe_rccl_site_in_children found on this node.
RCCL_SITE_OK = None


def build_context(args, system, group, ngroups, rank, world, precision, variant, lattice_sites=True):
    import torch
    import torch.distributed as dist
    from openmm_drudenose_amd import DrudeTGNHIntegrator, HipContext
    from openmm_drudenose_amd.drudetgnhplugin import FLAG_DEFER_SCALE, FLAG_RESIDENT_STEP, FLAG_TRUST_STATE_CHANGED, FLAG_GATHER
    from openmm_drudenose_amd.system import shard_bounds
    it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001, args.drude_steps, args.chains, True, True)
    it.setMaxDrudeDistance(args.hardwall)
    flags = {"plain": 0, "plain-trust": FLAG_TRUST_STATE_CHANGED, "plain-resident": FLAG_RESIDENT_STEP,
             "plain-resident-trust": FLAG_RESIDENT_STEP | FLAG_TRUST_STATE_CHANGED, "defer": FLAG_DEFER_SCALE,
             "resident": FLAG_DEFER_SCALE | FLAG_RESIDENT_STEP, "plain-gather": FLAG_GATHER}[variant]
    local, lgroup = system, group
    if world > 1:
        b = shard_bounds(system, world)
        local = system.slice_molecules(b[rank], b[rank + 1])
        lgroup = group[b[rank]:b[rank + 1]]
    if args.mode == "TGNH":
        for _ in range(ngroups):
            it.addTempGroup()
        # (an array, not a list of five million Python ints: a list that long is traversed by every full pass of Python's
        # cyclic garbage collector -- 17-22 ms each, see timed_run)
        it._particleTempGroup = np.ascontiguousarray(lgroup, np.int32)
    dev = torch.cuda.current_device()
    kw = {}
    if world > 1 or os.environ.get("TGNH_FORCE_DIST") == "1":
        def dof_sum(terms):
            t = torch.tensor(terms, dtype=torch.float64, device=CDEV)
            dist.all_reduce(t)
            return t.cpu().numpy()
        kw = dict(allreduce=lambda t: dist.all_reduce(t), global_dof_sum=dof_sum)
    ctx = HipContext(local, it, mode=args.mode, precision=precision, device=dev, flags=flags, lattice_sites=lattice_sites, **kw)
    ctx.exchange = "rccl" if kw else None
    ctx.rccl_site = None
    if kw:
        # The collective is the library's own ncclAllReduce (tgnh_rccl_init: its communicator, set up from an id that travels
        # through the process group); the torch.distributed hook set above stays only if that cannot be had -- a rehearsal
        # with several ranks on one device (RCCL refuses a device twice) or a gloo group.
        native = "TGNH_BENCH_DEVICE" not in os.environ and os.environ.get("TGNH_BENCH_BACKEND", "nccl") == "nccl" \
            and os.environ.get("TGNH_BENCH_RCCL", "library") == "library" and (RCCL_SITE_OK is True or (world == 1 and RCCL_SITE_OK is None))
        ctx.rccl_site = "library (ncclAllReduce enqueued by libdrudetgnh_hip)" if native and ctx.rccl_init_over(dist, rank, world) \
            else "torch.distributed.all_reduce through the tgnh_set_allreduce hook"
    if world > 1 and "TGNH_BENCH_DEVICE" in os.environ:
        ctx.set_resident_share(world)      # rehearsal: the ranks share ONE device, so each gets 1/world of its work-group slots
    return ctx


def attach_mailbox(ctx, rank, world):
    """Every rank's mailbox mapped into every other rank (HipContext.attach_exchange_over: IPC handles through the
    process group).  All ranks end up with the same answer: True = attached everywhere, False = nobody uses it."""
    import torch.distributed as dist
    if not ctx.attach_exchange_over(dist, rank, world, tensor_device=CDEV):
        e = getattr(ctx, "exchange_error", None)
        if e is not None:
            print(f"[bench] rank {rank}: mailbox exchange not available ({type(e).__name__}: {e})", file=sys.stderr)
        return False
    ctx.exchange = "mailbox"
    return True


def close_sharded(ctx):
    """unmap the peers' mailboxes, wait for everybody, then free (a mailbox must outlive its mappings)"""
    import torch
    import torch.distributed as dist
    torch.cuda.synchronize()
    if ctx.exchange == "mailbox":
        ctx.exchange_detach()
    if ctx.exchange is not None:
        dist.barrier()
    ctx.close()


def all_ranks_agree(ok):
    import torch
    import torch.distributed as dist
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=CDEV)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item()) == 1


def validate_mailbox(args, rank, world):
    """A small sharded box stepped twice, with the RCCL hook and with the mailbox exchange: the thermostats must agree
    (different summation order: 1e-9), be bit-identical over the ranks with the mailbox, and no wait may time out."""
    import copy
    import torch
    import torch.distributed as dist
    from openmm_drudenose_amd import synth
    small = copy.copy(args)
    system, group, ngroups = synth.water_box(3000 * world)
    res = {}
    for which in ("rccl", "mailbox"):
        ctx = build_context(small, system, group, ngroups, rank, world, args.precision, args.variant)
        attached = which == "rccl" or attach_mailbox(ctx, rank, world)
        if not attached:
            close_sharded(ctx)
            return False
        ctx.step(60)
        torch.cuda.synchronize()
        eta = torch.from_numpy(np.concatenate([ctx.thermostat_state(0), ctx.thermostat_state(1)])).to(CDEV)
        flags = ctx.status_flags()
        if which == "mailbox":
            every = [torch.empty_like(eta) for _ in range(world)]
            dist.all_gather(every, eta)
            same = all(torch.equal(every[0], e) for e in every)
            if not all_ranks_agree(same and (flags & 4) == 0):
                print(f"[bench] rank {rank}: mailbox validation failed (identical over ranks: {same}, flags {flags})", file=sys.stderr)
                close_sharded(ctx)
                return False
        res[which] = eta.cpu().numpy()
        close_sharded(ctx)
    close = np.allclose(res["mailbox"], res["rccl"], rtol=1e-9, atol=1e-13)
    if not all_ranks_agree(close):
        print(f"[bench] rank {rank}: mailbox and rccl thermostats differ", file=sys.stderr)
        return False
    return True


def per_rank_exchange_report(ctx, exchange, world):
    """Where a sharded leg spends its step, rank by rank (collective): the dominant launch's average duration on every rank
    and, for the mailbox exchange, how long work-group 0 of every rank waited for its peers' sums (device clock, whole leg:
    warm-up, timed region and the instrumented repeat).  A rank that waits long waits for a slower peer or for the link."""
    import torch.distributed as dist
    from openmm_drudenose_amd import _lib
    mine = {}
    for kid in (_lib.KID_STEP, _lib.KID_SKD, _lib.KID_KICK_KE, _lib.KID_CHAIN):
        row = ctx.leg["kernels"].get(_lib.KERNEL_NAMES[kid])
        if row:
            mine[_lib.KERNEL_NAMES[kid]] = row["avg_us"]
    if exchange == "mailbox":
        mean, mx, n = ctx.exchange_wait_stats()
        mine["wait_us_mean"], mine["wait_us_max"], mine["exchanges"] = round(mean, 2), round(mx, 2), n
    every = [None] * world
    dist.all_gather_object(every, mine)
    rep = {"per_rank_kernel_us": [{k: v for k, v in e.items() if not k.startswith("wait") and k != "exchanges"} for e in every]}
    if exchange == "mailbox":
        rep["wait_us_mean"] = [e["wait_us_mean"] for e in every]
        rep["wait_us_max"] = [e["wait_us_max"] for e in every]
        rep["exchanges"] = every[0]["exchanges"]
    return rep


def dominant_kid(variant):
    """the launch that carries most of a step: step_kernel (resident) or the fused rescale + half kick + drift pass"""
    from openmm_drudenose_amd import _lib
    return _lib.KID_STEP if variant in ("resident", "plain-resident") else _lib.KID_SKD      # (plain-resident-trust: the begin half is the tile launch)


def timed_run(ctx, steps, warmup, world, graph_steps=0, dom_kid=0):
    """W untimed warm-up steps, then exactly `steps` steps between barrier + synchronize pairs; max over ranks.
    graph_steps > 0: the steps are replays of a hipGraph holding graph_steps steps (+ an eager remainder).
    Leaves ctx.leg = what the JSON line reports for this leg, including the sum of the per-kernel times of one step
    (instrumented repeat) and a `suspect` mark when the timed region was more than 1.3x that sum."""
    import torch
    import torch.distributed as dist
    ctx.timing(True)               # creates the HIP-event pool now, outside the timed region
    ctx.timing(False)
    ctx.step(warmup)
    torch.cuda.synchronize()
    replay = None
    if graph_steps > 0:
        try:
            replay = ctx.capture_steps(graph_steps)
        except Exception as e:                         # capture unsupported here: stay eager
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            replay = None
        if world > 1:                                  # every rank replays, or none does: their exchanges must pair up
            t = torch.tensor([1 if replay is not None else 0], dtype=torch.int32, device=CDEV)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)   # (also a barrier: captures take different times, mailbox waits are bounded)
            if int(t.item()) == 0:
                replay = None
        torch.cuda.synchronize()
        if replay is not None:
            replay()                                   # first replay = the steps the capture recorded; untimed
            torch.cuda.synchronize()
    ctx.graph_used = replay is not None
    if replay is None:
        ctx.timing(2 + dom_kid)    # HIP events around the dominant kernel only: 2 records per step
    # Python's cyclic garbage collector stays out of the timed region (as in `timeit`): a full pass landing inside it is a
    # host-side stall of milliseconds to tens of milliseconds -- that was the leg the driver's round-1 record had at
    # 1.95 ms/step against 0.42 ms of kernels (20 steps enqueued in 18 ms instead of 0.5; it comes and goes with where the
    # collector's counters stand)
    gc.collect()
    gc.disable()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if replay is None:
        ctx.step(steps)
    else:
        for _ in range(steps // graph_steps):
            replay()
        ctx.step(steps % graph_steps)
    t_enq = time.perf_counter() - t0                   # host time to enqueue the region (eager: launch-bound if close to dt)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=CDEV)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    dom = None
    if replay is None:
        ctx.timing(False)
        dom = ctx.timing_read(dom_kid)                 # (total ms, launches) of the dominant kernel inside the timed region
    # full per-kernel table: an instrumented repeat of the same steps right after the timed region (event records
    # around all kernels cost ~20 us per step, which is why they are not inside it)
    nrep = max(1, min(steps, 200))
    ctx.timing(True)
    ctx.step(nrep)
    torch.cuda.synchronize()
    ctx.timing(False)
    ctx.dominant_in_timed_region = dom
    rows = kernel_table(ctx)
    sum_us = sum(v["avg_us"] * v["launches"] for v in rows.values()) / nrep
    ms = dt / steps * 1e3
    ctx.leg = {"steps_per_s": round(steps / dt, 2), "ms_per_step": round(ms, 4), "instrumented_steps": nrep,
               "sum_kernels_us_per_step": round(sum_us, 2), "host_enqueue_ms": round(t_enq * 1e3, 3),
               "suspect": bool(sum_us > 0 and ms * 1e3 > 1.3 * sum_us), "kernels": rows}
    if dom and dom[1]:
        ctx.leg["dominant_in_timed_region"] = {"avg_us": round(dom[0] / dom[1] * 1e3, 3), "launches": dom[1]}
    return dt


def integrator_only_run(ctx, steps, world, graph_steps):
    """The integrator's own launches as a TIMED REGION: the same barrier + synchronize bracket as timed_run around steps WITHOUT
    the force call-out (tgnh_run_steps).  The force buffer is zeroed first -- left as the last call-out filled it, a frozen spring
    force drives every Drude particle through its hard wall within ten steps and the thermostats out of every polynomial's
    range -- so these steps are free flight under the thermostats: the same launches on the same arrays, not a trajectory.
    At most 60 steps (a free Drude pair reaches its hard wall after ~80), a multiple of the graph's.  Call last: the state is
    spent afterwards."""
    import torch
    import torch.distributed as dist
    n = max(1, min(steps, 60))
    ctx.flush()                                        # (a deferred variant owes velm a half kick of the OLD force buffer)
    ctx.force.zero_()
    replay = None
    if graph_steps > 0 and ctx.graph_used:
        g = min(graph_steps, n)
        n = n // g * g
        try:
            replay = ctx.capture_steps(g, forces=False)
        except Exception as e:
            print(f"[bench] integrator-only capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
        if world > 1:
            t = torch.tensor([1 if replay is not None else 0], dtype=torch.int32, device=CDEV)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            if int(t.item()) == 0:
                replay = None
        torch.cuda.synchronize()
        if replay is not None:
            replay()
            torch.cuda.synchronize()
    else:
        ctx.step_without_forces(2)                     # (the transition out of the steps with a call-out)
    gc.collect()
    gc.disable()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if replay is None:
        ctx.step_without_forces(n)
    else:
        for _ in range(n // g):
            replay()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=CDEV)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return {"value": round(n / dt, 1), "unit": "steps/s", "steps": n, "us_per_step": round(dt / n * 1e6, 2), "hipgraph": replay is not None,
            "how": "timed region (barrier + synchronize on both sides, max over ranks) around tgnh_run_steps: step_begin + step_end with "
                   "NO force call-out launch, force buffer zeroed (free flight under the thermostats: the same launches on the same "
                   "arrays, not a trajectory); right after the headline leg, same context"}


def device_copy_gbps():
    """Achievable HBM bandwidth of THIS device: a 1 GiB float4-style device-to-device copy (read + write bytes / time).
    MI355X boxes differ by ~10 % on streaming kernels; this puts the roofline fraction next to what a copy reaches.
    Runs after every timed leg (its 2 GiB go through torch's caching allocator)."""
    import torch
    n = 256 * 1024 * 1024
    a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * 4 * n * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9


def kernel_table(ctx):
    from openmm_drudenose_amd import _lib
    rows = {}
    for kid, name in _lib.KERNEL_NAMES.items():
        ms, n = ctx.timing_read(kid)
        if n:
            b = ctx.algorithmic_bytes(kid)
            avg_us = ms / n * 1e3
            rows[name] = {"launches": n, "avg_us": round(avg_us, 3), "algorithmic_MB": round(b / 1e6, 3),
                          "GBps": round(b / (avg_us * 1e-6) / 1e9, 1) if b else None}
    return rows


def csrc_sha():
    """Hash of the kernel sources the loaded library was built from (profiles are stamped with the same hash)."""
    from openmm_drudenose_amd import build as hip_build
    return hip_build.source_sha()


def pmc_traffic(precision, slots, variant):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
    --pmc WRITE_SIZE in separate runs, gfx950 correction applied; tools/profile_round.py writes
    profiles/rNN_pmc_traffic.json and stamps it with the hash of csrc/).  PMC counters cannot be collected from
    inside this process, so the figure is the profiled one -- quoted only when it was taken from THIS binary
    (same source hash), the same workload and the same step variant; otherwise None."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")), reverse=True)
    for path in paths:                                   # the newest round's file that was taken from this binary and workload
        try:
            d = json.load(open(path))
            if d.get("slots") == slots and d.get("csrc_sha") == csrc_sha() and d.get("variant") == variant:
                return d["kernels"]["dominant"]["hbm_bytes_per_launch"], path
        except Exception:
            continue
    return None, (paths[0] if paths else None)


def cpu_baseline(args, system, group, ngroups):
    """The CPU oracle (the restatement of the reference's algorithm, 1 thread -- the reference platform is
    single-threaded scalar code) on the same system and harness force, a bounded number of steps: in dualNH mode
    (platforms/reference's algorithm: the stated baseline, `value`) and in TGNH mode (platforms/cuda's algorithm, the
    one the GPU headline runs)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from openmm_drudenose_amd import synth
    from oracle import Oracle, MODE_TGNH, MODE_DUALNH

    def one(mode_name, budget):
        mode = MODE_TGNH if mode_name == "TGNH" else MODE_DUALNH
        g = group if mode_name == "TGNH" else np.zeros_like(group)
        o = Oracle(system, g, ngroups if mode_name == "TGNH" else 1, mode, 300.0, 0.1, 1.0, 0.005, 0.001, args.drude_steps,
                   args.chains, True, True, args.hardwall)
        pos, vel, x0 = system.positions.copy(), system.velocities.copy(), system.positions.copy()
        f = o.harness_force(pos, x0, synth.K_DRUDE, synth.K_TETHER)
        t0 = time.perf_counter()
        o.run_harness(pos, vel, f, x0, synth.K_DRUDE, synth.K_TETHER, 2)
        per = (time.perf_counter() - t0) / 2
        n = int(max(3, min(200, budget / max(per, 1e-6))))
        # three chunks, the fastest one's rate: one core of a shared host (256 hardware threads, other tenants) reads 25 % low now
        # and then for seconds at a time (10.8 and 7.7 steps/s in two runs a minute apart on one box, round 4)
        best, done = 0.0, 0
        for k in range(3):
            m = n // 3 + (1 if k < n % 3 else 0)
            if m == 0:
                continue
            t0 = time.perf_counter()
            o.run_harness(pos, vel, f, x0, synth.K_DRUDE, synth.K_TETHER, m)
            best = max(best, m / (time.perf_counter() - t0))
            done += m
        return done, best

    nd, vd = one("dualNH", args.cpu_seconds / 2)
    nt, vt = one("TGNH", args.cpu_seconds / 2)
    return {"value": round(vd, 4), "unit": "steps/s", "cores": 1, "kind": "port",
            "dualnh": round(vd, 4), "tgnh": round(vt, 4),
            "sample": f"{nd} steps (dualNH = platforms/reference's algorithm, `value`) and {nt} steps (TGNH = platforms/cuda's "
                      f"algorithm) of the same {system.num_pairs}-pair system ({system.num_particles} slots), "
                      f"oracle/tgnh_oracle.c, fp64, gcc -O2, 1 thread, harness force included; each in three chunks, the fastest chunk's rate",
            "host_cpus": os.cpu_count()}


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args, argv):
    """`bench.py --gpus N` from a plain shell: start N fresh ranks (one per GPU) under torch.distributed.run, relay the
    one JSON line of rank 0 and exit with the launcher's code.  This process never touches a GPU (and has not imported
    torch), so the ranks are ordinary children -- nothing that has initialised HIP is re-executed."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    print("[bench] self-launch: " + " ".join(cmd), file=sys.stderr)
    p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    lines = [ln for ln in p.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not lines:
        print(f"[bench] the {args.gpus}-rank run failed (exit code {p.returncode}, {len(lines)} JSON lines)", file=sys.stderr)
        raise SystemExit(p.returncode or 1)
    print(lines[-1], flush=True)
    raise SystemExit(0)


def side_leg_in_children(args, other, rank, world, argv):
    """The exchange beside the headline one is measured by CHILD processes, one per rank, with a process group of their
    own: whatever happens to them -- a mapping that fails, a wait that never ends, a device fault that aborts the
    process -- the headline record of this run is already in hand and still gets printed.  (The mailbox exchange maps
    every rank's mailbox into every peer with hipIpc and stores across xGMI; on a one-GPU box only its one-device form
    can be rehearsed.)  The parents have closed their contexts and only wait; the children are ordinary child
    processes (nothing that has initialised HIP is re-executed).  Returns rank 0's record of the leg."""
    import subprocess
    import torch.distributed as dist
    port = [free_port() if rank == 0 else None]
    dist.broadcast_object_list(port, src=0)
    # (under torchrun the ranks' rendezvous is the agent's store; the children make their own on the new port)
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port[0]))
    keep = child_argv(argv, ("--side-leg", "--no-extra", "--probe-rccl-site"), ("--exchange",))
    cmd = [sys.executable, os.path.abspath(__file__)] + keep + ["--exchange", other, "--side-leg", "--no-cpu-baseline"]
    info = {"measured_by": "child ranks with a process group of their own"}
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, timeout=args.side_leg_timeout)
        lines = [ln for ln in p.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
        if p.returncode == 0 and lines:
            info.update(json.loads(lines[-1]))
        elif p.returncode != 0:
            info["failed"] = f"child rank {rank} ended with exit code {p.returncode}"
    except subprocess.TimeoutExpired:
        info["failed"] = f"child rank {rank} did not finish within {args.side_leg_timeout:.0f} s and was stopped"
    # one verdict for the run: rank 0 holds the record; any rank's failure marks it
    fails = [None] * world
    dist.all_gather_object(fails, info.get("failed"))
    bad = [f for f in fails if f]
    if bad:
        info = {k: v for k, v in info.items() if k in ("measured_by", "validated_against_rccl")}
        info["failed"] = "; ".join(bad)
        info["steps_per_s"] = None
    return info


def child_argv(argv, drop_flags, drop_valued):
    """the main run's own arguments minus what a child sets itself"""
    keep, skip = [], False
    for a in argv:
        if skip:
            skip = False
            continue
        if a in drop_flags:
            continue
        if a in drop_valued:
            skip = True
            continue
        if any(a.startswith(v + "=") for v in drop_valued):
            continue
        keep.append(a)
    return keep


def probe_rccl_site_in_children(args, rank, world, argv):
    """Before the ranks of a sharded run commit to the library's own RCCL site, child processes try it: ncclCommInitRank is
    collective, so a rank on which it fails or blocks would leave its peers inside it for good, with the headline record lost.
    One child per rank, a process group of their own, a small sharded box stepped eagerly and from a hipGraph through
    tgnh_rccl_init; the parents only wait (bounded).  Every rank returns the same verdict: True = the site works across
    these devices, False = the headline uses torch.distributed's all_reduce through the hook (the same RCCL, torch's site)."""
    import subprocess
    import torch.distributed as dist
    port = [free_port() if rank == 0 else None]
    dist.broadcast_object_list(port, src=0)
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port[0]))
    cmd = [sys.executable, os.path.abspath(__file__)] + child_argv(argv, ("--side-leg", "--no-extra", "--probe-rccl-site"), ("--exchange",)) \
        + ["--probe-rccl-site", "--no-cpu-baseline", "--no-extra"]
    ok, why = False, None
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, timeout=args.probe_timeout)
        lines = [ln for ln in p.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
        if p.returncode == 0 and (rank != 0 or (lines and json.loads(lines[-1]).get("rccl_site_ok") is True)):
            ok = True
        else:
            why = f"child rank {rank} ended with exit code {p.returncode}" + ("" if rank != 0 or lines else ", no verdict")
    except subprocess.TimeoutExpired:
        why = f"child rank {rank} did not finish within {args.probe_timeout:.0f} s and was stopped"
    votes = [None] * world
    dist.all_gather_object(votes, (ok, why))
    bad = [w for o, w in votes if not o]
    if bad and rank == 0:
        print("[bench] the library's RCCL site did not pass its probe (" + "; ".join(str(b) for b in bad) + "): torch.distributed's all_reduce through the hook", file=sys.stderr)
    return not bad


def rccl_site_probe(args, system_unused, rank, world):
    """the child's side of probe_rccl_site_in_children"""
    import torch
    import torch.distributed as dist
    from openmm_drudenose_amd import synth
    global RCCL_SITE_OK
    RCCL_SITE_OK = True                                # (this IS the probe: take the site)
    system, group, ngroups = synth.water_box(2500 * world)
    ctx = build_context(args, system, group, ngroups, rank, world, args.precision, "defer")
    good = ctx.rccl_site is not None and ctx.rccl_site.startswith("library")
    if good:
        ctx.step(20)
        replay = ctx.capture_steps(5)
        for _ in range(4):
            replay()
        torch.cuda.synchronize()
        eta = torch.from_numpy(np.concatenate([ctx.thermostat_state(0), ctx.thermostat_state(1)])).to(CDEV)
        every = [torch.empty_like(eta) for _ in range(world)]
        dist.all_gather(every, eta)
        good = bool(torch.isfinite(eta).all()) and all(torch.equal(every[0], e) for e in every) and ctx.check() == 0
    good = all_ranks_agree(good)
    close_sharded(ctx)
    return good


def dry_launch(args, world, rank):
    """Launcher check (CPU, gloo): the ranks meet, count themselves, rank 0 prints the count."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    dist.init_process_group("gloo")
    t = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(t)
    met = int(t.item())
    ok = met == args.gpus == dist.get_world_size()
    if rank == 0:
        print(json.dumps({"launcher_check": ok, "n_gpus": dist.get_world_size(), "ranks_met": met, "gpus_asked": args.gpus}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    raise SystemExit(0 if ok else 3)


def step_model_bytes(num_slots, precision, variant):
    """SURVEY 8d state-array model of one step for the pass structure actually run: plain = the reference's passes (SURVEY's
    7V + 2F + 2X as 6V + 3F + 2X since round 4: the end half's kick+KE pass stores nothing and its rescale launch forms the kicked
    velocities again); defer = rescale+kick+drift (2V + F + 2X) and kick+KE without a velocity write (V + F)."""
    V = 16 if precision == "single" else 32
    X = 16 if precision == "single" else 32
    F = 24
    per = {"plain": 6 * V + 3 * F + 2 * X, "plain-trust": 5 * V + 3 * F + 2 * X, "plain-resident": 6 * V + 3 * F + 2 * X,
           "plain-resident-trust": 5 * V + 3 * F + 2 * X, "defer": 3 * V + 2 * F + 2 * X, "plain-gather": 7 * V + 2 * F + 2 * X,
           "resident": 3 * V + 2 * F + 2 * X}[variant]
    return num_slots * per


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        self_launch(args, sys.argv[1:])                # does not return
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.dry_launch:
        dry_launch(args, world, rank)
    # Exactly ONE line may reach stdout (the JSON).  Libraries print there too (RCCL writes its version banner to
    # stdout at communicator creation), so fd 1 is pointed at stderr for the whole run and the JSON goes to a
    # private duplicate of the original stdout.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    local_rank = int(os.environ.get("TGNH_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(local_rank)
    # TGNH_FORCE_DIST=1 exercises the sharded code path (process group, dof all-reduce, KE all-reduce hook) on one rank
    use_dist = world > 1 or os.environ.get("TGNH_FORCE_DIST") == "1"
    ranks_met = 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if BACKEND == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(BACKEND)
        t = torch.ones(1, dtype=torch.int64, device=CDEV)
        dist.all_reduce(t)                              # the ranks that actually met, counted over the collective itself
        ranks_met = int(t.item())
        if ranks_met != args.gpus or dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but {ranks_met} ranks met in the process group")

    global RCCL_SITE_OK
    if args.probe_rccl_site:                           # a child rank of probe_rccl_site_in_children: try, report, done
        good = rccl_site_probe(args, None, rank, world)
        if rank == 0:
            os.write(real_stdout, (json.dumps({"rccl_site_ok": bool(good)}) + "\n").encode())
        dist.barrier()
        dist.destroy_process_group()
        return

    from openmm_drudenose_amd import synth, _lib
    system, group, ngroups = synth.water_box(args.molecules)
    if args.variant == "auto":
        # One launch per step (wstep_kernel on a water box): +4-10 % from 2.5 M slots per GPU down, i.e. for the shards of a
        # 2-, 4- or 8-GPU run (profiles/r03_scaling_ceiling.md), and since the wave tiles of identical molecules stopped reading
        # per-slot words also +3 % at 5 M slots (interleaved on one box: 3 866 / 3 869 / 3 875 against 3 745 / 3 740 / 3 727 for the
        # three-launch structure, tools/micro/variant_ab.sh).  A handle that cannot hold the step resident falls back by itself
        # (`variant_ran`).  Single precision at the metric size stays with three launches (6 331 against 5 671 steps/s on one box:
        # its passes are short enough for the one-launch step's serial section to show).
        args.variant = "defer" if args.precision == "single" and system.num_particles / world >= 3_000_000 else "resident"

    use_graph = args.graph == "on" or (args.graph == "auto" and world > 1)
    gsteps = max(1, min(args.graph_steps, args.steps)) if use_graph else 0

    def run_leg(exchange):
        """one timed run of the main configuration with the given exchange -> (ctx, seconds) ; ctx still open"""
        ctx = build_context(args, system, group, ngroups, rank, world, args.precision, args.variant)
        if exchange == "mailbox" and not attach_mailbox(ctx, rank, world):
            close_sharded(ctx)
            return None, None
        # (rehearsals over gloo: its all_reduce cannot be captured into a hipGraph, RCCL's can)
        g = 0 if (exchange == "rccl" and BACKEND != "nccl") else gsteps
        dt = timed_run(ctx, args.steps, args.warmup, world, g, dominant_kid(args.variant))
        return ctx, dt

    def side_leg(other):
        """the exchange that is not the headline one, same system, same steps -> its record"""
        info = {}
        ok = True
        if other == "mailbox":
            ok = validate_mailbox(args, rank, world)
            info["validated_against_rccl"] = ok
        if ok:
            c2, d2 = run_leg(other)
            if c2 is not None:
                timed_out = other == "mailbox" and not all_ranks_agree((c2.status_flags() & 4) == 0)
                info.update({"steps_per_s": None if timed_out else round(args.steps / d2, 3),
                             "ms_per_step": round(d2 / args.steps * 1e3, 5), "hipgraph": c2.graph_used,
                             "step_kernel_ran": _lib.KERNEL_NAMES[_lib.KID_STEP] in c2.leg["kernels"],
                             "timed_out": timed_out, "sum_kernels_us_per_step": c2.leg["sum_kernels_us_per_step"]})
                info.update(per_rank_exchange_report(c2, other, world))
                close_sharded(c2)
            else:
                info["attached"] = False
        return info

    # (TGNH_BENCH_PROBE=1: the probe also on one rank, for tests/test_bench_sharded_gpu.py -- a one-GPU box cannot hold two RCCL ranks)
    if use_dist and (world > 1 or os.environ.get("TGNH_BENCH_PROBE") == "1") and BACKEND == "nccl" and "TGNH_BENCH_DEVICE" not in os.environ \
            and os.environ.get("TGNH_BENCH_RCCL", "library") == "library":
        # (a side-leg child measuring the RCCL exchange probes like the main run would; one measuring the mailboxes never takes the site)
        RCCL_SITE_OK = probe_rccl_site_in_children(args, rank, world, sys.argv[1:]) if args.exchange == "rccl" else False
    if args.side_leg:                                  # a child rank of side_leg_in_children: measure, report, done
        info = side_leg(args.exchange)
        if rank == 0:
            os.write(real_stdout, (json.dumps(info) + "\n").encode())
        dist.barrier()
        dist.destroy_process_group()
        return

    headline_exchange = args.exchange if use_dist else None
    mailbox_info = None
    if headline_exchange == "mailbox":
        ok = validate_mailbox(args, rank, world)
        mailbox_info = {"validated_against_rccl": ok}
        if not ok:
            raise SystemExit("bench.py: --exchange mailbox, but the mailbox exchange did not validate against rccl on this node")
    ctx, dt = run_leg(headline_exchange)
    if ctx is None:
        raise SystemExit("bench.py: the mailbox exchange could not be attached")
    if ctx.exchange == "mailbox" and not all_ranks_agree((ctx.status_flags() & 4) == 0):
        raise SystemExit("bench.py: the mailbox exchange timed out inside the timed run")
    graph_used = ctx.graph_used
    step_kernel_name = ctx.resident_kernel() or "step_kernel"
    exchange_used = ctx.exchange
    rccl_site = ctx.rccl_site
    headline_ranks = per_rank_exchange_report(ctx, exchange_used, world) if use_dist else None
    leg = ctx.leg
    rows = leg["kernels"]
    assert ctx.check() == 0
    # dominant kernel: step_kernel (resident variant: the whole step but the force call-out) or the fused rescale +
    # half kick + drift (+ hard wall) pass; a handle that cannot run step_kernel (RCCL hook) falls back to the passes
    dkid = dominant_kid(args.variant)
    if _lib.KERNEL_NAMES[dkid] not in rows:
        dkid = _lib.KID_SKD
    dom_name = _lib.KERNEL_NAMES[dkid]
    bytes_dom = ctx.algorithmic_bytes(dkid)
    if dkid == dominant_kid(args.variant) and ctx.dominant_in_timed_region and ctx.dominant_in_timed_region[1]:
        ms, n = ctx.dominant_in_timed_region
        dom = {"avg_us": round(ms / n * 1e3, 3), "launches": n, "where": "HIP events inside the timed region"}
    else:
        dom = dict(rows.get(dom_name), where="HIP events in the instrumented repeat right after the timed region "
                                             "(the region itself was a hipGraph replay, or ran another launch structure than asked for)")
    achieved = bytes_dom / (dom["avg_us"] * 1e-6) / 1e9
    local_slots = ctx.n
    harness_force = ctx.sites_kind()                   # how the call-out inside `value` found its tether sites: lattice | packed | x0
    int_only = integrator_only_run(ctx, args.steps, world, gsteps)
    close_sharded(ctx)

    extra = {}
    if use_dist and world > 1 and not args.no_extra:
        # the other exchange beside the headline one (the mailboxes beside an RCCL headline, or the reverse), measured by
        # child ranks so that nothing it does can take the headline record with it
        other = "mailbox" if headline_exchange == "rccl" else "rccl"
        torch.cuda.empty_cache()                       # the children run on this rank's GPU
        extra[other] = side_leg_in_children(args, other, rank, world, sys.argv[1:])
    if mailbox_info:
        extra["mailbox"] = dict(extra.get("mailbox", {}), **mailbox_info)
    if world == 1 and not args.no_extra:
        for prec, var in ((args.precision, "plain"), (args.precision, "plain-trust"), (args.precision, "plain-resident"), (args.precision, "plain-resident-trust"), (args.precision, "defer"), (args.precision, "resident"),
                          (args.precision, "plain-gather"),       # the reference's structure on the gather path (TGNH_FLAG_GATHER): what a topology the tiles cannot hold gets
                          ("single", "defer" if system.num_particles / world >= 3_000_000 else args.variant)):
            if (prec, var) == (args.precision, args.variant):
                continue
            c2 = build_context(args, system, group, ngroups, rank, world, prec, var)
            timed_run(c2, args.steps, args.warmup, world, 0, dominant_kid(var))
            extra[f"{prec}/{var}"] = c2.leg
            c2.close()
        if harness_force == "lattice":           # the headline again with the harness force reading PACKED sites (any box, not
            # only one molecule on a cubic lattice: what rounds 1-3 timed): `value` moves with the call-out stand-in, the integrator does not
            c2 = build_context(args, system, group, ngroups, rank, world, args.precision, args.variant, lattice_sites=False)
            timed_run(c2, args.steps, args.warmup, world, gsteps, dominant_kid(args.variant))
            extra[f"{args.precision}/{args.variant}/packed-sites"] = dict(c2.leg, harness_force=c2.sites_kind())
            c2.close()
        if args.mode == "TGNH":                  # the other semantic mode (platforms/reference's algorithm), same workload
            import copy
            a2 = copy.copy(args)
            a2.mode = "dualNH"
            c2 = build_context(a2, system, group, ngroups, rank, world, args.precision, args.variant)
            timed_run(c2, args.steps, args.warmup, world, 0, dominant_kid(args.variant))
            extra[f"dualNH/{args.precision}/{args.variant}"] = c2.leg
            c2.close()
    copy_gbps = device_copy_gbps() if rank == 0 else None

    if rank == 0:
        b_step = step_model_bytes(system.num_particles, args.precision, args.variant)
        traffic, traffic_path = pmc_traffic(args.precision, local_slots, args.variant) if world == 1 else (None, None)
        out = {
            "metric": "integrator steps/sec at 1M Drude pairs",
            "value": round(args.steps / dt, 3), "unit": "steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if args.precision == "single" else "f64",
            "data": "synthetic",
            "config": {
                "workload": f"SWM4-NDP water box, {args.molecules} molecules = {system.num_particles} particle slots, "
                            f"{system.num_pairs} Drude pairs, 1 temperature group (+ molecular-COM and Drude thermostats), "
                            f"{args.mode} mode, {args.precision} precision, numNHChains={args.chains}, hard wall "
                            f"{args.hardwall} nm, harness force call-out inside the timed region (harness_force: {harness_force})",
                "harness_force": harness_force,
                "precision": args.precision, "variant": args.variant,
                "variant_ran": args.variant if dkid == dominant_kid(args.variant) else {"resident": "defer", "plain-resident": "plain", "plain-resident-trust": "plain-trust"}.get(args.variant, args.variant),
                "hipgraph": graph_used,
                "parallelism": f"particle-sharded x{world} (whole molecules), one KE all-reduce per step",
                "exchange": exchange_used, "rccl_ranks": ranks_met if use_dist else None,
                "rccl_site": rccl_site if exchange_used == "rccl" else None,
                "rccl_site_probe": RCCL_SITE_OK,          # what child ranks found when they tried the library's site first (None: not probed)
                "per_rank": headline_ranks,
                "slots_per_gpu": local_slots,
                "model_bytes_per_step": b_step,
                "step_GBps_vs_model": round(b_step / (dt / args.steps) / 1e9 / world, 1),
                "sum_kernels_us_per_step": leg["sum_kernels_us_per_step"], "suspect": leg["suspect"],
                "csrc_sha": csrc_sha(),
            },
            "roofline": {"bound": "hbm", "kernel": step_kernel_name if dkid == _lib.KID_STEP else "tile_kernel<scale+kick+drift>",
                         "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic,
                         "traffic_source": f"{os.path.relpath(traffic_path, ROOT)} (separate rocprofv3 --pmc passes, bytes per launch; "
                                           "null unless taken from this binary, workload and variant)" if traffic_path else None,
                         "algorithmic_bytes_per_launch": bytes_dom, "avg_launch_us": dom["avg_us"],
                         "launches_timed": dom["launches"], "timing": dom["where"],
                         "device_copy_GBps": round(copy_gbps, 1),
                         "frac_of_device_copy": round(achieved / copy_gbps, 4)},
            "kernels": rows,
        }
        # `value` times whole steps, the harness force call-out included (in a real context that slot is OpenMM's
        # calcForcesAndEnergy).  The integrator's own launches alone (SURVEY 8d reports the force kernel separately):
        own = sum(v["avg_us"] * v["launches"] for k, v in rows.items() if k != "harness force") / leg["instrumented_steps"]
        out["integrator_only"] = dict(int_only, vs_model_roofline_steps_per_s=round(HBM_PEAK_GBS * 1e9 / b_step * world, 1))
        if own > 0:
            out["integrator_only"]["sum_of_kernels"] = {"steps_per_s": round(1e6 / own, 1), "us_per_step": round(own, 2),
                                                        "how": "sum of the durations of the integrator's own launches per step of the real trajectory "
                                                               "(instrumented repeat), force call-out excluded; derived, not a timed region"}
        if extra:
            out["extra"] = extra
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, system, group, ngroups)
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
