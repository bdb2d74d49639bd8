"""bench.py's sharded flow, rehearsed with two ranks on the ONE GPU of the box (gloo for the setup collectives and the
all-reduce hook; the ranks share the device's work-group slots): its own launcher, whole-molecule shards, the headline leg
with the collective hook, and -- measured by child ranks with a process group of their own, so that nothing it does can
cost the headline record -- the mailbox exchange validated against the hook.  What cannot be rehearsed here is the link
itself (stores crossing xGMI, RCCL between devices)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _bench(argv, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(TGNH_BENCH_BACKEND="gloo", TGNH_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.update(env)
    return subprocess.run([sys.executable, BENCH] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=600)


@pytest.mark.gpu
def test_two_rank_rehearsal_reports_both_exchanges():
    p = _bench(["--gpus", "2", "--molecules", "20000", "--steps", "100", "--warmup", "10"])
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["value"] > 0
    assert j["config"]["exchange"] == "rccl" and j["config"]["rccl_ranks"] == 2       # the headline is the collective hook
    assert j["config"]["slots_per_gpu"] == 50000
    # the call-out stand-in inside `value` is named, and the integrator's own launches are a timed region of their own
    assert j["config"]["harness_force"] in ("lattice", "packed", "x0") and j["config"]["harness_force"] in j["config"]["workload"]
    io = j["integrator_only"]
    assert io["value"] > j["value"] and io["steps"] >= 1 and "timed region" in io["how"] and io["sum_of_kernels"]["steps_per_s"] > 0
    mb = j["extra"]["mailbox"]
    assert "failed" not in mb, mb
    assert mb["validated_against_rccl"] is True and mb["timed_out"] is False
    assert mb["steps_per_s"] > 0 and mb["step_kernel_ran"] is True
    assert "child ranks" in mb["measured_by"]


@pytest.mark.gpu
def test_a_side_leg_that_dies_does_not_take_the_headline_with_it():
    # the children are told to stop at once (their time limit is zero seconds): the run still ends with its headline line
    p = _bench(["--gpus", "2", "--molecules", "20000", "--steps", "50", "--warmup", "5", "--side-leg-timeout", "0.001"])
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    j = json.loads([ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][0])
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["config"]["exchange"] == "rccl"
    mb = j["extra"]["mailbox"]
    assert mb["steps_per_s"] is None and "stopped" in mb["failed"]


def _bench_nccl(argv, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(TGNH_FORCE_DIST="1", TGNH_BENCH_PROBE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT="29533")
    e.update(env)
    return subprocess.run([sys.executable, BENCH] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=600)


@pytest.mark.gpu
def test_the_rccl_site_is_probed_by_children_before_the_ranks_commit_to_it():
    """ncclCommInitRank is collective: a rank on which the library's own RCCL site fails or blocks would leave its peers inside it.
    A sharded bench.py therefore lets CHILD ranks try the site first (a small box, eagerly and from a hipGraph, thermostats compared
    over the ranks) and takes it only when they all came back in time.  One rank here (a one-GPU box cannot hold two RCCL ranks:
    TGNH_BENCH_PROBE=1 runs the same plumbing): the probe passes and the headline uses the library's site; with no time given to the
    children it fails and the headline still comes out, through torch.distributed's all_reduce."""
    p = _bench_nccl(["--molecules", "20000", "--steps", "50", "--warmup", "5", "--no-extra", "--no-cpu-baseline"])
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    j = json.loads([ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][0])
    assert j["config"]["rccl_site_probe"] is True and j["config"]["rccl_site"].startswith("library") and j["value"] > 0
    p = _bench_nccl(["--molecules", "20000", "--steps", "50", "--warmup", "5", "--no-extra", "--no-cpu-baseline", "--probe-timeout", "0.001"],
                    MASTER_PORT="29534")
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    j = json.loads([ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][0])
    assert j["config"]["rccl_site_probe"] is False and j["config"]["rccl_site"].startswith("torch.distributed") and j["value"] > 0
