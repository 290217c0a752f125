"""The C++ multi-GPU hosts on a box without (enough) GPUs: `--spawn 2` must start two ranks and BOTH must stop with "not enough devices"
before any communicator is built -- one process per GPU is a precondition, not something to discover as a hang.  (With GPUs present the
hosts are run by tests/test_gpu_rccl_hosts.py.)"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "eqf_vio_amd", "cpp")


@pytest.mark.parametrize("exe,args", [("eqf_example_batch_rccl", ["4", "6", "3"]), ("eqf_example_tiled_rccl", ["40", "2", "8"])])
def test_spawn_2_without_two_gpus_fails_loudly_on_both_ranks(exe, args):
    import torch

    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs are visible: the job would simply run")
    path = os.path.join(CPP, exe)
    if not os.path.exists(path):  # (a checkout that has not been through __graft_entry__.build() yet)
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "eqf_vio_amd", "csrc"), "-j3"])
        subprocess.check_call(["make", "-C", CPP])
    run = subprocess.run([path, "--spawn", "2"] + args, capture_output=True, text=True, timeout=120)
    assert run.returncode == 3
    assert "rank 0 of 2: not enough devices" in run.stderr and "rank 1 of 2: not enough devices" in run.stderr, run.stderr
