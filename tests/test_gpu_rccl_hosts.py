"""The two C++ multi-GPU hosts (eqf_vio_amd/cpp/example_batch_rccl.cpp, example_tiled_rccl.cpp: one process per GPU, RCCL, no MPI, no Python)
with ONE rank on the GPU box -- all that can be run before a node exists: the launcher (--spawn), the ncclUniqueId rendezvous through a file,
communicator construction and splitting, the scatter / gather groups, the broadcast callback and its self-check all execute; the filters'
results are compared with the Python binding / the oracle on the same inputs.  Their behaviour on a box with too few GPUs is in
tests/test_rccl_hosts_cpu.py."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "eqf_vio_amd", "cpp")


def _json_line(out):
    for ln in reversed(out.strip().splitlines()):
        if ln.strip().startswith("{"):
            return json.loads(ln)
    raise AssertionError("no JSON line in:\n" + out[-2000:])


def _write_streams(path, streams):
    from eqf_vio_amd import shard

    imu, vst, bear = shard.pack(streams)
    ids = streams[0].ids.astype(np.int32)
    with open(path, "wb") as f:
        np.array([imu.shape[0], vst.shape[0], len(ids), len(streams)], dtype=np.int32).tofile(f)
        imu.tofile(f)
        vst.tofile(f)
        ids.tofile(f)
        bear.tofile(f)
    return imu, vst, ids, bear


@pytest.mark.parametrize("launch", ["plain", "spawn"])
def test_batch_host_matches_the_python_binding(tmp_path, launch):
    """Four filters of N = 24 on the synthetic streams of SURVEY.md 8(d), seeds 1234 + b: the C++ host (rank 0 reads the file, scatter, replay,
    gather) against FilterBatch on the same arrays -- same library, same kernels, same inputs: the same numbers."""
    from eqf_vio_amd import binding, synth

    exe = os.path.join(CPP, "eqf_example_batch_rccl")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    N, B = 24, 4
    streams = [synth.make_stream(N, seed=1234 + b, duration=0.42) for b in range(B)]
    sp, op = str(tmp_path / "streams.bin"), str(tmp_path / "res.bin")
    imu, vst, ids, bear = _write_streams(sp, streams)
    cmd = ([exe, "--spawn", "1"] if launch == "spawn" else [exe]) + ["--streams", sp, "--out", op]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr[-3000:] + run.stdout[-1000:]
    line = _json_line(run.stdout)
    assert line["n_gpus"] == 1 and line["filters_total"] == B and line["landmarks"] == N and line["device_error_flag"] == 0
    res = np.fromfile(op, dtype=np.float64).reshape(B, 8)
    fb = binding.FilterBatch(synth.template_settings_dict(), capacity=N, batch=B)
    fb.stream_upload(imu, vst, ids, bear)
    k = 0
    for f in range(vst.shape[0]):  # the schedule of main.cpp:111-170, as the C++ host replays it
        while k < imu.shape[0] and imu[k, 0, 0] < vst[f, 0]:
            fb.stream_imu(k)
            k += 1
        fb.stream_vision(f)
    assert fb.device_error() == 0
    for b in range(B):
        e = fb.state_estimate(b)
        assert np.array_equal(res[b, :4], e["q"]) and np.array_equal(res[b, 4:7], e["x"]), b
        assert abs(res[b, 7] / np.linalg.norm(fb.sigma(b)) - 1.0) < 1e-13, b  # (the two sums of squares add in different orders)
    assert line["steps"] > 0 and line["value"] > 0


def test_tiled_host_matches_the_oracle(oracle_lib):
    """The partitioned filter driven from C++ with its RCCL callback in place (1 x 1 grid: the library does not exchange, the host still builds
    and checks every communicator) on a closed-form stream with landmarks leaving and coming back, against the oracle on the same inputs."""
    from eqf_vio_amd import synth

    exe = os.path.join(CPP, "eqf_example_tiled_rccl")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    N, frames, bl = 120, 5, 16
    run = subprocess.run([exe, "--spawn", "1", str(N), str(frames), str(bl)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr[-3000:] + run.stdout[-1000:]
    head = next(ln for ln in run.stdout.splitlines() if ln.startswith("t="))
    nums = [float(x) for x in re.findall(r"[-+]?\d+\.\d+(?:e[-+]?\d+)?", head)]
    t, pos, q, fro = nums[0], nums[1:4], nums[4:8], nums[8]
    line = _json_line(run.stdout)
    assert line["grid"] == "1 x 1" and line["landmarks"] == N
    i = np.arange(N)
    lm = np.stack([2 * np.sin(1.3 * i), 2 * np.cos(0.7 * i), 5 + np.sin(0.37 * i)], axis=1)
    y = lm / np.linalg.norm(lm, axis=1, keepdims=True)
    fo = oracle_lib.OracleFilter(synth.template_settings_dict())
    k = 0
    for f in range(frames):
        stamp = 0.05 * f + 0.0025
        while 0.005 * k < stamp:
            w = [0.02 * np.sin(0.015 * k), 0.015 * np.cos(0.01 * k), 0.01 * np.sin(0.0075 * k)]
            a = [9.81 + 0.05 * np.sin(0.02 * k), 0.04 * np.cos(0.015 * k), 0.03 * np.sin(0.0125 * k)]
            fo.processIMUData(0.005 * k, w, a)
            k += 1
        vis = (f + i) % 97 != 0
        fo.processVisionData(stamp, i[vis].astype(np.int32), y[vis])
    e = fo.stateEstimate()
    assert abs(t - fo.getTime()) < 1e-9
    assert np.abs(np.array(pos) - e["x"]).max() < 2e-9 and np.abs(np.array(q) - e["q"]).max() < 2e-9  # printed with 9 digits
    assert abs(fro / np.linalg.norm(fo.stateCovariance()) - 1) < 1e-8


def test_bench_gpus_1_prints_one_line():
    """`python bench.py --gpus 1`, the shape of the driver's command: one process, n_gpus 1, launcher named in the line."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "44", "--warmup", "22", "--no-batch64",
                          "--no-cpu-baseline", "--no-traffic", "--no-parity", "--no-steady-state", "--no-batch8", "--no-n1000", "--no-tiled",
                          "--no-churn"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["launcher"] == "single process" and line["device_error_flag"] == 0
