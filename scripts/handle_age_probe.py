#!/usr/bin/env python
"""Does a process get slower with every partitioned filter it has created and closed?  (round 6; GPU box.)  A plain fp64 product on torch
memory (eqf_tile_gemm_tn on torch's stream, addresses unchanged over the run) and on freshly allocated torch memory, timed before and after each
create / run / close cycle of a cfg 5 filter; and the filter's own frame time per cycle.   python scripts/handle_age_probe.py [cycles=5] [N=4000]"""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from eqf_vio_amd import synth, tiled  # noqa: E402

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 5
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
bl = 250
dev = torch.device("cuda", 0)
be0 = tiled.HipBackend({}, capacity=8)
k, m = 750, 8192
A = torch.randn(k, m, dtype=torch.float64, device=dev)
C = torch.zeros(m, m, dtype=torch.float64, device=dev)


def gemm_ms(Cm, Am):
    be0.gemm_tn(Cm, Am, Am, -1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        be0.gemm_tn(Cm, Am, Am, -1.0)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5


def fresh_ms():
    A2 = torch.randn(k, m, dtype=torch.float64, device=dev)
    C2 = torch.zeros(m, m, dtype=torch.float64, device=dev)
    t = gemm_ms(C2, A2)
    del A2, C2
    torch.cuda.empty_cache()
    return t


flops = 2.0 * k * m * m


t = gemm_ms(C, A)
print(f"before any filter: product on resident torch memory {t:.3f} ms ({flops / t / 1e9:.1f} TFLOP/s), on fresh memory {fresh_ms():.3f} ms", flush=True)
st = synth.make_stream(N, seed=1234, duration=4 / 20.0 + 0.011)
ev = list(st.events())
first_vis = next(i for i, (kind, _) in enumerate(ev) if kind == "vision")
warm, timed = ev[: first_vis + 1], ev[first_vis + 1: first_vis + 1 + 22]
d = synth.template_settings_dict()
for c in range(cycles):
    be = tiled.HipBackend(d, capacity=N)
    tf = tiled.TiledFilter(tiled.ProcessGrid(None, 1, 1, device=be.device), be, bl)
    tf.check_every = 0
    if os.environ.get("PROBE_PANEL_AHEAD"):
        tf._opt("panel_ahead", 1)
    if os.environ.get("PROBE_NO_OVERLAP"):
        tf.overlap_chains = False

    def run(events):
        for kind, kk in events:
            if kind == "imu":
                r = st.imu[kk]
                tf.processIMUData(r[0], r[1:4], r[4:7])
            else:
                tf.processVisionData(st.vision_stamps[kk], st.ids, st.bearings[kk])

    run(warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(timed)
    t_sub = (time.perf_counter() - t0) / 2
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    tf.close()
    be.close()
    del tf, be
    gc.collect()
    t = gemm_ms(C, A)
    print(f"cycle {c}: filter {dt * 1e3:7.2f} ms/frame (host returned after {t_sub * 1e3:7.2f});  afterwards the product on resident torch memory {t:.3f} ms ({flops / t / 1e9:.1f} TFLOP/s), on fresh memory {fresh_ms():.3f} ms",
          flush=True)
