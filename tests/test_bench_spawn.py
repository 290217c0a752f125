"""`python bench.py --gpus N` must start its N ranks itself (the flag used to be parsed and never read, so an 8-GPU run
launched as plain `python bench.py --gpus 8` would have printed a one-GPU line).  No GPU here, so two things are checked:
  * the launcher itself -- two ranks started by bench.py meet over gloo, rank 0's input streams reach both, results come back, ONE JSON
    line with n_gpus = 2 leaves rank 0 (--spawn-selftest: shard.py's two collectives, no filter);
  * the real job on a GPU-less box reaches "not enough devices" from BOTH children and the launcher's exit code is not 0.
The launcher that runs the filters on real GPUs is exercised by tests/test_gpu_nccl.py (-m gpu)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_gpus_2_spawns_two_ranks_that_meet_over_gloo():
    run = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--spawn-selftest"], cwd=ROOT, env=_env(), capture_output=True, text=True,
                         timeout=300)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, run.stdout  # ONE JSON line, from rank 0
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["self_spawned"] is True
    assert line["ranks_seen"] == [0, 1] and line["filters"] == 4 and line["events"] == 33


def test_gpus_2_without_gpus_fails_loudly_on_both_ranks():
    import torch

    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs are visible: the job would simply run")
    run = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "22", "--warmup", "11"], cwd=ROOT, env=_env(), capture_output=True,
                         text=True, timeout=300)
    assert run.returncode != 0
    assert "rank 0 of 2: not enough devices" in run.stderr and "rank 1 of 2: not enough devices" in run.stderr, run.stderr[-3000:]
    assert not [ln for ln in run.stdout.splitlines() if ln.strip().startswith("{")]  # and no line that could be mistaken for a result


def test_gpus_flag_must_agree_with_the_launcher():
    env = dict(_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    run = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--spawn-selftest"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert run.returncode == 2 and "WORLD_SIZE is 1" in run.stderr
