#!/usr/bin/env python
"""cfg 5 (one filter of N landmarks, 1 x 1 grid) with the update's products on the integer matrix pipe: frame time and per-phase GPU time for
chain_slices / downdate_slices settings, against the fp64 run of the same frames (round 6; GPU box).
    python scripts/chain_slices_probe.py [N=4000] [bl=250] [frames=3] [settings = "0,0 0,6 5,6 5,0 6,6"]   (chain,downdate[,trsm_leaf[,downdate_early]] per setting)"""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from eqf_vio_amd import synth, tiled  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
bl = int(sys.argv[2]) if len(sys.argv) > 2 else 250
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 3
settings = [tuple(int(x) for x in s.split(",")) for s in (sys.argv[4] if len(sys.argv) > 4 else "0,0 0,6 5,6 5,0 6,6").split()]  # chain,downdate[,trsm_leaf[,downdate_early %]]
st = synth.make_stream(N, seed=1234, duration=(frames + 2) / 20.0 + 0.011)
ev = list(st.events())
first_vis = next(i for i, (kind, _) in enumerate(ev) if kind == "vision")
warm, timed, more = ev[: first_vis + 1], ev[first_vis + 1: first_vis + 1 + 11 * frames], ev[first_vis + 1 + 11 * frames: first_vis + 1 + 11 * (frames + 1)]
d = synth.template_settings_dict()
ref = None
for (chain, dd, *rest) in settings:
    be = tiled.HipBackend(d, capacity=N)
    tf = tiled.TiledFilter(tiled.ProcessGrid(None, 1, 1, device=be.device), be, bl)
    tf.check_every = 0
    tf.chain_slices, tf.downdate_slices = chain, dd
    if rest:
        tf._opt("trsm_leaf", rest[0])
    if len(rest) > 1:
        tf._opt("downdate_early", rest[1])
    if len(rest) > 2:
        tf._opt("solve_inverse", rest[2])

    def run(events):
        for kind, k in events:
            if kind == "imu":
                r = st.imu[k]
                tf.processIMUData(r[0], r[1:4], r[4:7])
            else:
                tf.processVisionData(st.vision_stamps[k], st.ids, st.bearings[k])

    tf.phase_ms = None
    run(warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(timed)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tf.check()
    sll = tf.Sll.clone()
    if ref is None and chain == 0 and dd == 0:
        ref = sll
    diff = float((torch.linalg.norm(sll - ref) / torch.linalg.norm(ref)).item()) if ref is not None else float("nan")
    tf.phase_ms = {}
    run(more)
    tf.collect_phases()
    ph = {k: round(v, 2) for k, v in tf.phase_ms.items()}
    print(f"N={N} bl={bl} chain_slices={chain} downdate_slices={dd}{' trsm_leaf=%d' % rest[0] if rest else ''}{' downdate_early=%d' % rest[1] if len(rest) > 1 else ''}{' solve_inverse=%d' % rest[2] if len(rest) > 2 else ''}: {dt * 1e3 / frames:7.2f} ms/frame  {len(timed) / dt:7.1f} steps/s   "
          f"Sigma vs fp64 {diff:.2e}   phases {ph}   err {be.device_error()}", flush=True)
    tf.close()
    be.close()
    del tf, be, sll
    gc.collect()
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    print(f"    (device memory in use after closing the handle: {(total - free) / 2**30:.1f} GiB)", flush=True)
