#!/usr/bin/env python
"""Verdict r3 item 4(a): the 2 x 4 schedule of the partitioned filter (BASELINE cfg 5) measured WITHOUT a node -- 8 ranks share ONE MI355X,
each confined to 32 of its 256 CUs (hipExtStreamCreateWithCUMask), the grid's broadcasts over gloo on device tensors.  What this can
show: how much of a frame each rank spends in its own kernels against the 1 x 1 grid on all 256 CUs (schedule efficiency of the
compute side).  What it cannot show: RCCL over xGMI -- gloo stages every broadcast through host memory, so the time spent INSIDE the
collectives is reported separately and is not representative.
    python scripts/tiled_cumask.py [N=4000] [block=250] [frames=2] [Pr=2] [Pc=4]
CUMASK_MODE=plain leaves the broadcasts untimed (for a rocprofv3 kernel trace of the ranks: scripts/cumask_kstats.py), CUMASK_ONLY=1x1|grid
runs one of the two grids."""
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

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
    import tiled_reference as tiled_ref  # (the Python twin of the host loop: it takes a CU slice per rank, HipBackend's cu_range)
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
    grid = tiled_ref.ProcessGrid(dist_, Pr, Pc, device=be.device)
    # time inside the grid's collectives (host wall clock around a synchronised call: the schedule is perturbed a little, the split is honest)
    coll = {"wait": 0.0, "xfer": 0.0, "n": 0, "bytes": 0}

    def timed(fn, group):
        # a barrier over the collective's own group first: what it takes is the time this rank waits for the others to ARRIVE (imbalance,
        # the serial diagonal blocks: part of the schedule); what the broadcast takes after it is transfer through host memory (gloo's, not
        # xGMI's).  Host wall clock around synchronised calls: the schedule loses its host/device overlap, the split is honest.
        def w(t, *a):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dist.barrier(group=group())
            t1 = time.perf_counter()
            r = fn(t, *a)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            coll["wait"] += t1 - t0
            coll["xfer"] += t2 - t1
            coll["n"] += 1
            coll["bytes"] += t.numel() * t.element_size()
            return r
        return w

    if world > 1 and os.environ.get("CUMASK_MODE", "split") == "split":
        grid.bcast_row = timed(grid.bcast_row, lambda: grid.row_group)
        grid.bcast_col = timed(grid.bcast_col, lambda: grid.col_group)
    tf = tiled_ref.TiledFilter(grid, be, bl)
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
    coll.update(wait=0.0, xfer=0.0, n=0, bytes=0)
    t0 = time.perf_counter()
    run(timed_ev)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    res = {"rank": rank, "frame_ms": dt * 1e3 / frames, "wait_ms": coll["wait"] * 1e3 / frames, "xfer_ms": coll["xfer"] * 1e3 / frames, "collectives": coll["n"] // frames,
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
    only = os.environ.get("CUMASK_ONLY", "both")  # "1x1", "grid" or "both" (a rocprofv3 run wants one grid per invocation)
    fro = {}
    for (pr, pc) in ((1, 1), (Pr, Pc)):
        world = pr * pc
        if (only == "1x1" and world > 1) or (only == "grid" and world == 1):
            continue
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
        fro[world] = sum(r["sigma_fro_local"] ** 2 for r in rs) ** 0.5
        print(f"# grid {pr} x {pc}, N = {N}, block {bl}, {frames} frames of 10 IMU + 1 vision call; every rank on {256 // world} CUs of ONE MI355X")
        for r in rs:
            print("  rank %d: frame %.1f ms = own kernels + host %.1f + waiting for the group to arrive %.1f + inside gloo broadcasts %.1f (%d calls, %.0f MB through host memory)   err %d"
                  % (r["rank"], r["frame_ms"], r["frame_ms"] - r["wait_ms"] - r["xfer_ms"], r["wait_ms"], r["xfer_ms"], r["collectives"], r["MB_moved"], r["err"]))
        busy = max(r["frame_ms"] - r["wait_ms"] - r["xfer_ms"] for r in rs)
        path = max(r["frame_ms"] - r["xfer_ms"] for r in rs)
        print(f"  frame (max over ranks) {fm:.1f} ms; busiest rank's own work {busy:.1f} ms; frame minus the host-staged transfers (= the schedule's critical path with free links) {path:.1f} ms")
        if world == 1:
            base = fm
        elif 1 in fro:
            print(f"  ||Sigma||_F after the run: 1 x 1 grid {fro[1]:.12e}, {pr} x {pc} grid (root of the ranks' sum of squares) {fro[world]:.12e}, relative difference {abs(fro[world] - fro[1]) / fro[1]:.2e}")
            print(f"  1 x 1 grid on 256 CUs: {base:.1f} ms.  {pr} x {pc} grid, {256 // world} CUs per rank: critical path {path:.1f} ms = {base / path:.2f} of the ideal "
                  f"(each rank has 1/{world} of the chip, so the 1 x 1 time IS the ideal), busiest rank {busy:.1f} ms = {base / busy:.2f}")
            print(f"  => on {world} whole GPUs the same schedule's compute side would take about {path / world:.1f} ms per frame if kernels scaled with CUs "
                  f"({busy / world:.1f} ms for the busiest rank's own work); the rest of a real frame is RCCL over xGMI, which this run cannot show")
