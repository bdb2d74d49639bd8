#!/usr/bin/env python3
"""GPU-side measurement probes behind the tuning logs in profiles/ (one file, sub-commands; run on an MI355X box):

  probe.py knob NAME=v1,v2,... [--molecules a,b] [bench args]   bench.py under a tuning build for every value of one environment
                                                             knob (TGNH_TILE_CAP, TGNH_GRID, TGNH_INLINE_CHAIN, TGNH_INLINE_SUM_ROWS,
                                                             TGNH_INLINE_SUM_ALL, ...): needs TGNH_LIB = a -DTGNH_TUNING build
  probe.py variants [bench args]                             A/B of the library builds in build_variants/ against the main one
  probe.py chain                                             chain cost vs drudeStepsPerRealStep and chain length
  probe.py copy                                              device copy time vs footprint (what a streaming launch can cost at best)
  probe.py stream                                            c = a + b over 160 MB arrays vs the distance between their bases
  probe.py placement [molecules] [pools]                     does a launch's speed go with where the driver put the buffers?
  probe.py drift [molecules]                                 does the dominant launch drift in time (clock ramp, idle gaps)?
  probe.py soak [steps] [waters] [lag]                       two-process mailbox exchange on one GPU, many thousand exchanges
  probe.py ke [molecules]                                    the KE pass back to back vs behind a kernel that has just written 120-1000 MB
  probe.py resident-soak                                     step_kernel for 10^5 launches at three sizes: no meeting may time out
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
BENCH = os.path.join(ROOT, "bench.py")


def bench(env, args):
    r = subprocess.run([sys.executable, BENCH, "--no-cpu-baseline", "--no-extra", *args], env=env, capture_output=True, text=True)
    if r.returncode:
        return None, r.stderr[-300:]
    return json.loads(r.stdout.strip().splitlines()[-1]), None


def show(tag, d, err):
    if d is None:
        print(tag, "FAILED", err, flush=True)
    else:
        print(f"{tag:34s} {d['value']:9.1f} steps/s | " + " | ".join(f"{n} {v['avg_us']:.1f}" for n, v in d["kernels"].items()), flush=True)


def knob(argv):
    name, values = argv[0].split("=")
    rest = argv[1:]
    mols = [1000000]
    if "--molecules" in rest:
        i = rest.index("--molecules")
        mols = [int(x) for x in rest[i + 1].split(",")]
        rest = rest[:i] + rest[i + 2:]
    if "TGNH_LIB" not in os.environ:
        raise SystemExit("set TGNH_LIB to a -DTGNH_TUNING build (tools/build_variant.py): the product library reads no knobs")
    for mol in mols:
        big = mol >= 500000
        for rep in range(2):
            for v in values.split(","):
                e = dict(os.environ)
                if v != "default":
                    e[name] = v
                d, err = bench(e, ["--molecules", str(mol), "--steps", "600" if big else "2000", "--warmup", "100",
                                   "--graph", "off" if big else "on", *rest])
                show(f"{mol:8d} {name}={v}", d, err)


def variants(argv):
    libs = {"main": None}
    vdir = os.path.join(ROOT, "build_variants")
    if os.path.isdir(vdir):
        for f in sorted(os.listdir(vdir)):
            if f.endswith(".so"):
                libs[f[:-3]] = os.path.join(vdir, f)
    for rep in range(2):
        for name, lib in libs.items():
            e = dict(os.environ)
            if lib:
                e["TGNH_LIB"] = lib
            d, err = bench(e, ["--steps", "300", "--graph", "off", *argv])
            show(name, d, err)


def chain(argv):
    for chains in (1, 3):
        for S in (1, 5, 20, 80):
            d, err = bench(dict(os.environ), ["--steps", "200", "--warmup", "20", "--molecules", "100000", "--variant", "defer",
                                              "--drude-steps", str(S), "--chains", str(chains)])
            show(f"chains {chains} S {S:3d}", d, err)


def copy(argv):
    import torch
    dev = torch.device("cuda:0")
    for mb in (8, 16, 32, 48, 64, 96, 128, 256, 512, 1024):
        n = mb * (1 << 20) // 2          # mb = bytes read + bytes written
        a = torch.empty(n, dtype=torch.uint8, device=dev)
        b = torch.empty_like(a)
        for _ in range(20):
            b.copy_(a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(50):
                b.copy_(a)
        g.replay()
        torch.cuda.synchronize()
        e0.record(); g.replay(); g.replay(); e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 100
        print(f"{mb:5d} MB moved (r+w): {us:7.1f} us  {mb * 1.048576 / us:6.2f} TB/s", flush=True)


def stream(argv):
    import torch
    dev = torch.device("cuda:0")
    MiB = 1 << 20
    n = 40_000_000                      # float32 elements: 160 MB per array
    pool = torch.zeros(6 * 1024 * MiB, dtype=torch.uint8, device=dev)
    base = (-pool.data_ptr()) % (2 * MiB)

    def view(off):
        return pool[base + off: base + off + 4 * n].view(torch.float32)

    def timeit(a, b, c, reps=30):
        for _ in range(5):
            torch.add(a, b, out=c)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            torch.add(a, b, out=c)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    size = 4 * n
    span = ((size + 2 * MiB - 1) // (2 * MiB)) * 2 * MiB
    a, c = view(0), view(4096 * MiB)
    for d in [span, span + 2 * MiB, 160 * MiB, 256 * MiB, 258 * MiB, 512 * MiB, 514 * MiB, 1024 * MiB, 1026 * MiB, 2048 * MiB, 3072 * MiB]:
        us = timeit(a, view(d), c)
        print(f"b at +{d / MiB:7.1f} MiB: {us:7.1f} us  {3 * size / us / 1e6:6.2f} TB/s", flush=True)


def _context(mol):
    from openmm_drudenose_amd import synth
    from openmm_drudenose_amd.drudetgnhplugin import DrudeTGNHIntegrator, HipContext, FLAG_DEFER_SCALE
    system, group, ngroups = synth.water_box(mol)
    it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001, 20, 1, True, True)
    it.setMaxDrudeDistance(0.02)
    return HipContext(system, it, mode="TGNH", precision="mixed", flags=FLAG_DEFER_SCALE)


def placement(argv):
    """All state arrays carved from a fresh pool (earlier pools stay allocated, so every pool is different physical memory);
    the streaming launches timed on each, then again over the same pools: does the figure stick to the pool?"""
    import torch
    from openmm_drudenose_amd import _lib
    from openmm_drudenose_amd.drudetgnhplugin import _check
    mol = int(argv[0]) if argv else 1000000
    npools = int(argv[1]) if len(argv) > 1 else 10
    ctx = _context(mol)
    names = ["velm", "force", "posq", "posq_corr", "x0", "pos_delta"]
    state = {n: getattr(ctx, n).clone() for n in names}
    MB2 = 2 << 20
    slots, off = {}, 0
    for n in names:
        slots[n] = off
        off += (state[n].numel() * state[n].element_size() + MB2 - 1) // MB2 * MB2

    def use(pool):
        base = (-pool.data_ptr()) % MB2
        for n in names:
            o = state[n]
            v = pool[base + slots[n]: base + slots[n] + o.numel() * o.element_size()].view(o.dtype).view(o.shape)
            v.copy_(o)
            setattr(ctx, n, v)
        _check(ctx.lib.tgnh_flush(ctx.h, ctx._stream()))
        _check(ctx.lib.tgnh_bind_buffers(ctx.h, ctx.posq.data_ptr(), ctx.posq_corr.data_ptr(), ctx.velm.data_ptr(),
                                         ctx.force.data_ptr(), ctx.pos_delta.data_ptr()))

    def measure(tag):
        ctx.step(40)
        torch.cuda.synchronize()
        ctx.timing(True)
        ctx.step(200)
        torch.cuda.synchronize()
        ctx.timing(False)
        ks = {kid: ctx.timing_read(kid) for kid in (_lib.KID_SKD, _lib.KID_KICK_KE, _lib.KID_FORCE)}
        _check(ctx.lib.tgnh_flush(ctx.h, ctx._stream()))
        for n in names:
            state[n] = getattr(ctx, n).clone()
        print(f"{tag}: " + " | ".join(f"{_lib.KERNEL_NAMES[k]} {ms / n * 1e3:7.2f}" for k, (ms, n) in ks.items()), flush=True)
    pools = []
    for i in range(npools):
        pools.append(torch.zeros(off + MB2, dtype=torch.uint8, device=ctx.dev))
        use(pools[-1])
        measure(f"pool {i:2d} @ {pools[-1].data_ptr():#x}")
    for i in range(npools):
        use(pools[i])
        measure(f"again {i:2d} @ {pools[i].data_ptr():#x}")


def drift(argv):
    import torch
    ctx = _context(int(argv[0]) if argv else 1000000)

    def window(tag, steps=200):
        ctx.timing(2 + 0)
        t0 = time.perf_counter()
        ctx.step(steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ctx.timing(False)
        ms, n = ctx.timing_read(0)
        print(f"{tag}: {steps / dt:8.1f} steps/s | scale+kick+drift {ms / n * 1e3:7.2f} us", flush=True)
    for i in range(12):
        window(f"back to back {i:2d}")
    for gap in (0.5, 1.0, 2.0, 5.0, 5.0):
        time.sleep(gap)
        window(f"after {gap:3.1f} s idle, first 100 steps", 100)
        window("                  next 200 steps      ")


def soak(argv):
    import tempfile
    import numpy as np
    steps = argv[0] if argv else "20000"
    waters = argv[1] if len(argv) > 1 else "20000"
    lag = argv[2] if len(argv) > 2 else "0"          # seconds rank 1 idles every 50 steps (rank 0 waits in its kernels)
    for variant in ("2", "0"):
        out = tempfile.mkdtemp()
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", TGNH_XW_WATERS=waters, TGNH_XW_PAIRS="500", TGNH_XW_LAG=lag)
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "xchg_worker.py"), str(r), "2", out, steps, variant], env=env)
                 for r in range(2)]
        rcs = [p.wait() for p in procs]
        flags = [int(np.load(os.path.join(out, f"flags{r}.npy"))[0]) for r in range(2)]
        eta = [np.load(os.path.join(out, f"eta{r}.npy")) for r in range(2)]
        vel = [np.load(os.path.join(out, f"vel{r}.npy")) for r in range(2)]
        print(f"variant {variant}: rc {rcs} flags {flags} thermostats identical {np.array_equal(eta[0], eta[1])} "
              f"finite {all(np.isfinite(v).all() for v in vel)}", flush=True)


def ke(argv):
    """The KE pass alone, back to back, against the same launch behind a kernel that has just WRITTEN 120 / 480 MB (what the
    harness force and the rescale+kick+drift launch leave behind in a step): does the pass run slower in a step because of what
    it does, or because the memory system is still writing back the launch before it?  (HIP events per launch.)"""
    import torch
    from openmm_drudenose_amd import _lib
    ctx = _context(int(argv[0]) if argv else 1000000)
    ctx.step(5)
    _check_flush(ctx)
    n = ctx.n
    mb = n * (32 + 4) / 1e6                       # velm + index word

    def run(tag, fill_mb, reps=40):
        buf = torch.empty(int(fill_mb * 1e6) // 8, dtype=torch.float64, device=ctx.dev) if fill_mb else None
        for _ in range(5):
            ctx.compute_kinetic_energies()
        torch.cuda.synchronize()
        ctx.timing(2 + _lib.KID_KE)
        for i in range(reps):
            if buf is not None:
                buf.fill_(float(i))
            _lib_call(ctx)
        torch.cuda.synchronize()
        ctx.timing(False)
        ms, k = ctx.timing_read(_lib.KID_KE)
        us = ms / k * 1e3
        print(f"{tag:58s} {us:7.1f} us  {mb / us:5.2f} TB/s (velm + index word, {mb:.0f} MB)", flush=True)
    run("KE pass, back to back", 0)
    run("KE pass behind a 120 MB fill (the harness force's stores)", 120)
    run("KE pass behind a 320 MB fill (rescale+kick+drift's stores)", 320)
    run("KE pass behind a 1 GB fill", 1000)
    run("KE pass, back to back (again)", 0)


def _check_flush(ctx):
    from openmm_drudenose_amd.drudetgnhplugin import _check
    _check(ctx.lib.tgnh_flush(ctx.h, ctx._stream()))


def _lib_call(ctx):
    from openmm_drudenose_amd.drudetgnhplugin import _check
    _check(ctx.lib.tgnh_compute_kinetic_energies(ctx.h, ctx._stream()))


def resident_soak(argv):
    import numpy as np
    import torch
    from openmm_drudenose_amd import synth
    from openmm_drudenose_amd.drudetgnhplugin import DrudeTGNHIntegrator, HipContext, FLAG_DEFER_SCALE, FLAG_RESIDENT_STEP
    for mol, steps in ((125000, 150000), (20000, 300000), (1000000, 40000)):     # (the metric size: bench.py's default variant since round 3)
        s, g, ng = synth.water_box(mol)
        it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001, 20, 1, True, True)
        it.setMaxDrudeDistance(0.02)
        ctx = HipContext(s, it, mode="TGNH", precision="mixed", flags=FLAG_DEFER_SCALE | FLAG_RESIDENT_STEP)
        ctx.step(20)
        torch.cuda.synchronize()
        rep = ctx.capture_steps(10)
        rep()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps // 10):
            rep()
            if i % 5000 == 4999:
                torch.cuda.synchronize()
                print(mol, "steps", (i + 1) * 10, "status word", ctx.status_flags(), flush=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{mol} molecules: {steps} steps in {dt:.1f} s = {steps / dt:.0f} steps/s, status word {ctx.status_flags()}, "
              f"velocities finite {np.isfinite(ctx.getVelocities()).all()}", flush=True)
        t0 = time.perf_counter()
        ctx.step(steps // 4)                                 # ... and launched eagerly, as bench.py does
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{mol} molecules, eager: {steps // 4} steps in {dt:.1f} s = {steps // 4 / dt:.0f} steps/s, status word {ctx.status_flags()}, check {ctx.check()}, "
              f"velocities finite {np.isfinite(ctx.getVelocities()).all()}", flush=True)
        ctx.close()


if __name__ == "__main__":
    cmds = {"ke": ke, "resident-soak": resident_soak, "knob": knob, "variants": variants, "chain": chain, "copy": copy, "stream": stream, "placement": placement,
            "drift": drift, "soak": soak}
    if len(sys.argv) < 2 or sys.argv[1] not in cmds:
        raise SystemExit(__doc__)
    cmds[sys.argv[1]](sys.argv[2:])
