"""world_size-2 gloo test of the only multi-GPU path: sharding independent filters over ranks
(scatter of pre-generated streams, gather of results; no data-path collective).  CPU only: the per-filter work
is done by the fp64 oracle here, standing in for the HIP filter, which needs a GPU."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, N, B, nev, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from eqf_vio_amd import shard, synth
    from oracle import binding as ob

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    imu, vst, bear, events = shard.scatter_streams(dist, rank, world, N, B, nev)
    # every rank must have received exactly the streams of ITS filters (seed = 1234 + global index)
    for b in range(B):
        ref = synth.make_stream(N, seed=1234 + rank * B + b, duration=max(1.0, (nev + 40) / 220.0 + 0.1))
        assert np.array_equal(imu[:, b], ref.imu)
        assert np.array_equal(vst[:, b], ref.vision_stamps)
        assert np.array_equal(bear[:, b], ref.bearings)
    res = np.zeros((B, 8))
    for b in range(B):
        f = ob.OracleFilter(synth.template_settings_dict())
        for kind, k in events:
            if kind == "imu":
                f.processIMUData(imu[k, b, 0], imu[k, b, 1:4], imu[k, b, 4:7])
            else:
                f.processVisionData(vst[k, b], np.arange(N, dtype=np.int32), bear[k, b])
        e = f.stateEstimate()
        res[b, :3], res[b, 3:7], res[b, 7] = e["x"], e["q"], np.linalg.norm(f.stateCovariance())
    allres = shard.gather_results(dist, rank, world, res)
    if rank == 0:
        np.save(os.path.join(out_dir, "gathered.npy"), allres)
    else:
        assert allres is None
    np.save(os.path.join(out_dir, f"local_{rank}.npy"), res)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_scatter_gather(tmp_path):
    world, N, B, nev = 2, 4, 2, 60
    mp.spawn(_worker, args=(world, _free_port(), N, B, nev, str(tmp_path)), nprocs=world, join=True)
    gathered = np.load(tmp_path / "gathered.npy")
    assert gathered.shape == (world * B, 8)
    for r in range(world):
        assert np.array_equal(gathered[r * B:(r + 1) * B], np.load(tmp_path / f"local_{r}.npy"))
    # different seeds -> different filters
    assert not np.allclose(gathered[0], gathered[1])
