"""`python bench.py --gpus N` launches its own ranks (the driver's scaling command is the N = 1 command with another --gpus).
CPU-only: the spawn path is exercised with --launcher-selftest (gloo), the device check with the real path on a box without
N GPUs."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, timeout=240):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=timeout, env=env)


def test_self_launch_spawns_the_ranks_and_rank_zero_prints_one_line():
    r = _run("--gpus", "2", "--launcher-selftest")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d == {"launcher_selftest": True, "n_gpus": 2, "world": 2, "sum": 3.0, "self_launched": True}


def test_more_gpus_than_devices_fails_with_a_clear_message():
    import torch
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    r = _run("--gpus", str(n_dev + 2), "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert f"needs {n_dev + 2} devices" in r.stderr, r.stderr[-2000:]
