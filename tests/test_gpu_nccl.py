"""The RCCL code path of bench.py on ONE GPU: launched the way the driver launches an N-GPU
run (python -m torch.distributed.run, RANK / WORLD_SIZE / MASTER_* in the environment), so that init_process_group("nccl"), the scatter
of the input streams, the gather of the results and the all_reduce of the timing all execute -- with a single rank, which is all that
can be proven before a multi-GPU node exists.  The line it prints must agree with the same job run without torch.distributed."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--gpus", "1", "--steps", "220", "--warmup", "110", "--no-batch64", "--no-cpu-baseline", "--no-traffic", "--no-parity",
        "--no-steady-state"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _last_json_line(out):
    for ln in reversed(out.strip().splitlines()):
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            return json.loads(ln)
    raise AssertionError("no JSON line in:\n" + out[-2000:])


def test_bench_runs_its_nccl_path_with_one_rank():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    plain = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + ARGS, cwd=ROOT, env=env, capture_output=True, text=True,
                           timeout=600)
    assert plain.returncode == 0, plain.stderr[-3000:]
    ref = _last_json_line(plain.stdout)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + ARGS
    run = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-3000:]
    line = _last_json_line(run.stdout)
    assert line["n_gpus"] == 1 and line["device_error_flag"] == 0
    assert line["steps"] == 220 and line["config"]["filters_total"] == 1
    assert "RCCL" in line["config"]["parallelism"]
    # same workload, same kernels: the rate may only differ by run-to-run noise, and the filter must have produced the same numbers
    assert abs(line["value"] / ref["value"] - 1.0) < 0.25, (line["value"], ref["value"])
    assert line["sigma_fro_filter0"] == ref["sigma_fro_filter0"]
    assert line["roofline"]["kernel"] == ref["roofline"]["kernel"]
