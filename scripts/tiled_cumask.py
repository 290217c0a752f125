#!/usr/bin/env python
"""Verdict r3 item 4(a): the 2 x 4 schedule of the partitioned filter (BASELINE cfg 5) measured WITHOUT a node -- 8 ranks share ONE MI355X,
each confined to 32 of its 256 CUs (hipExtStreamCreateWithCUMask), the grid's broadcasts over gloo on device tensors.  What this can
show: how much of a frame each rank spends in its own kernels against the 1 x 1 grid on all 256 CUs (schedule efficiency of the
compute side).  What it cannot show: RCCL over xGMI -- gloo stages every broadcast through host memory, so the time spent INSIDE the
collectives is reported separately and is not representative.
    python scripts/tiled_cumask.py [N=4000] [block=250] [frames=2] [Pr=2] [Pc=4]"""
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")


def worker(rank, world, port, Pr, Pc, N, bl, frames, out_dir):
    import numpy as np
    import torch
    import torch.distributed as dist

    from eqf_vio_amd import synth, tiled

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist_ = None
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist_ = dist
    torch.cuda.set_device(0)
    cus = 256 // world
    d = synth.template_settings_dict()
    be = tiled.HipBackend(d, capacity=N, device_index=0, reserve_cus=8 if world == 1 else 4, cu_range=None if world == 1 else (rank * cus, cus))
    grid = tiled.ProcessGrid(dist_, Pr, Pc, device=be.device)
    # time inside the grid's collectives (host wall clock around a synchronised call: the schedule is perturbed a little, the split is honest)
    coll = {"t": 0.0, "n": 0, "bytes": 0}

    def timed(fn):
        def w(t, *a):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = fn(t, *a)
            torch.cuda.synchronize()
            coll["t"] += time.perf_counter() - t0
            coll["n"] += 1
            coll["bytes"] += t.numel() * t.element_size()
            return r
        return w

    if world > 1:
        grid.bcast_row = timed(grid.bcast_row)
        grid.bcast_col = timed(grid.bcast_col)
    tf = tiled.TiledFilter(grid, be, bl)
    tf.overlap_chains = world == 1  # (two chains' collectives from two streams over gloo: serialised anyway)
    st = synth.make_stream(N, seed=1234, duration=(frames + 2) / 20.0 + 0.011)
    ev = list(st.events())
    first_vis = next(i for i, (kind, _) in enumerate(ev) if kind == "vision")
    warm, timed_ev = ev[: first_vis + 1], ev[first_vis + 1: first_vis + 1 + 11 * frames]

    def run(events):
        for kind, k in events:
            if kind == "imu":
                r = st.imu[k]
                tf.processIMUData(r[0], r[1:4], r[4:7])
            else:
                tf.processVisionData(st.vision_stamps[k], st.ids, st.bearings[k])

    tf.check_every = 0
    run(warm)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    coll.update(t=0.0, n=0, bytes=0)
    t0 = time.perf_counter()
    run(timed_ev)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    res = {"rank": rank, "frame_ms": dt * 1e3 / frames, "in_collectives_ms": coll["t"] * 1e3 / frames, "collectives": coll["n"] // frames,
           "MB_moved": coll["bytes"] / frames / 1e6, "err": be.device_error(), "sigma_fro_local": float(torch.linalg.norm(tf.Sll).item())}
    json.dump(res, open(os.path.join(out_dir, f"r{rank}.json"), "w"))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    import tempfile

    import torch.multiprocessing as mp

    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    bl = int(sys.argv[2]) if len(sys.argv) > 2 else 250
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    Pr = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    Pc = int(sys.argv[5]) if len(sys.argv) > 5 else 4
    for (pr, pc) in ((1, 1), (Pr, Pc)):
        world = pr * pc
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        out = tempfile.mkdtemp()
        if world == 1:
            worker(0, 1, port, 1, 1, N, bl, frames, out)
        else:
            mp.spawn(worker, args=(world, port, pr, pc, N, bl, frames, out), nprocs=world, join=True)
        rs = [json.load(open(os.path.join(out, f"r{r}.json"))) for r in range(world)]
        fm = max(r["frame_ms"] for r in rs)
        print(f"# grid {pr} x {pc}, N = {N}, block {bl}, {frames} frames of 10 IMU + 1 vision call; every rank on {256 // world} CUs of ONE MI355X")
        for r in rs:
            print("  rank %d: frame %.1f ms, of which inside gloo broadcasts %.1f ms (%d calls, %.0f MB through host memory) -> own kernels + host %.1f ms   err %d"
                  % (r["rank"], r["frame_ms"], r["in_collectives_ms"], r["collectives"], r["MB_moved"], r["frame_ms"] - r["in_collectives_ms"], r["err"]))
        comp = max(r["frame_ms"] - r["in_collectives_ms"] for r in rs)
        print(f"  frame (max over ranks) {fm:.1f} ms; compute side (frame minus time inside broadcasts, max over ranks) {comp:.1f} ms")
        if world == 1:
            base = fm
        else:
            print(f"  schedule efficiency of the compute side: {base:.1f} ms on 256 CUs against {comp:.1f} ms on {256 // world} CUs per rank x {world} ranks = {base / comp:.2f}"
                  f"  (1.00 = the 1 x 1 grid's time; each rank has 1/{world} of the chip, so 1.00 is also the ideal)")
