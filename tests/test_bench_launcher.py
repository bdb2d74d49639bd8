"""bench.py's own launcher: `python bench.py --gpus N` from a plain shell must start N ranks itself (the driver's
command shape), and a rank count that differs from --gpus must end non-zero instead of silently measuring one GPU.
CPU only: --dry-launch makes the ranks rendezvous over gloo, count themselves and stop before any GPU work."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(argv, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env)
    return subprocess.run([sys.executable, BENCH] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=300)


def test_plain_shell_invocation_starts_its_own_ranks():
    p = _run(["--gpus", "2", "--dry-launch"])
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()          # exactly one JSON line is relayed
    j = json.loads(lines[0])
    assert j == {"launcher_check": True, "n_gpus": 2, "ranks_met": 2, "gpus_asked": 2}


def test_rank_count_mismatch_is_an_error():
    # under a launcher that started a different number of ranks than --gpus asks for: refuse, non-zero
    p = _run(["--gpus", "2", "--dry-launch"], WORLD_SIZE="3", RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    assert p.returncode != 0
    assert b"WORLD_SIZE=3" in p.stderr
    # one rank, --gpus 1: no launcher involved
    p = _run(["--gpus", "1", "--dry-launch"])
    assert p.returncode == 0 and json.loads(p.stdout.decode().splitlines()[-1])["n_gpus"] == 1


def test_step_model_bytes_follow_the_variant():
    sys.path.insert(0, ROOT)
    import bench
    n = 5_000_000
    # SURVEY 8d counts the reference's passes as 7V + 2F + 2X (336 / 192 B per slot); since round 4 the end half's kick+KE pass
    # stores nothing and its rescale launch reads the forces again: 6V + 3F + 2X, what the launches of that structure move
    assert bench.step_model_bytes(n, "mixed", "plain") == n * 328
    assert bench.step_model_bytes(n, "single", "plain") == n * 200
    assert bench.step_model_bytes(n, "mixed", "plain-trust") == n * 296    # ... without the begin half's KE pass
    assert bench.step_model_bytes(n, "mixed", "defer") == n * 208       # 3V + 2F + 2X
    assert bench.step_model_bytes(n, "mixed", "plain-resident") == n * 328   # 6V + 3F + 2X
