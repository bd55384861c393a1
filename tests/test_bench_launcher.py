"""bench.py's own launcher: `--gpus N` without a launcher around it starts N ranks (one `torch.distributed.run`
child, before anything touches a GPU); under a launcher whose world size differs it exits non-zero.  The ranks here
only rendezvous over gloo on the CPU (--launcher-selftest)."""

import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_gpus_2_starts_two_ranks():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launcher-selftest"], env=_env(), capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    got = json.loads(line)
    assert got["n_gpus"] == 2 and got["ranks"] == [0, 1]


def test_world_size_mismatch_is_an_error():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launcher-selftest"], env=_env(WORLD_SIZE="1", RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 2 and "WORLD_SIZE=1" in p.stderr
    p = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--launcher-selftest"], env=_env(WORLD_SIZE="2", RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 2


def test_single_rank_needs_no_launcher():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--launcher-selftest"], env=_env(), capture_output=True,
                       text=True, timeout=120)
    assert p.returncode == 0 and json.loads(p.stdout.strip().splitlines()[-1])["n_gpus"] == 1
