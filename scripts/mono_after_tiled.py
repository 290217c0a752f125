#!/usr/bin/env python
"""The monolithic single-GPU filter at N = 4000 (eqf_process_*) timed after k partitioned handles have been created, run and closed in the same
process, with or without the integer-pipe options (round 6; GPU box).   python scripts/mono_after_tiled.py [handles=4] [chain_slices=0] [downdate_slices=0]"""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from eqf_vio_amd import binding, synth, tiled  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
chain = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dd = int(sys.argv[3]) if len(sys.argv) > 3 else 0
N, bl = 4000, 250
st = synth.make_stream(N, seed=1234, duration=6 / 20.0 + 0.011)
ev = list(st.events())
first_vis = next(i for i, (kind, _) in enumerate(ev) if kind == "vision")
warm, timed = ev[: first_vis + 1], ev[first_vis + 1: first_vis + 1 + 33]
d = synth.template_settings_dict()


def run(f_imu, f_vis, events):
    for kind, k in events:
        if kind == "imu":
            r = st.imu[k]
            f_imu(r[0], r[1:4], r[4:7])
        else:
            f_vis(st.vision_stamps[k], st.ids, st.bearings[k])


for c in range(K):
    be = tiled.HipBackend(d, capacity=N)
    tf = tiled.TiledFilter(tiled.ProcessGrid(None, 1, 1, device=be.device), be, bl)
    tf.check_every = 0
    tf.chain_slices, tf.downdate_slices = chain, dd
    run(tf.processIMUData, tf.processVisionData, warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(tf.processIMUData, tf.processVisionData, timed)
    torch.cuda.synchronize()
    print(f"partitioned handle {c}: {(time.perf_counter() - t0) * 1e3 / 3:.2f} ms/frame", flush=True)
    if os.environ.get("PROBE_PHASES"):  # (bench.py's per-phase pass: event brackets with timing on the handle's streams)
        tf.phase_ms = {}
        run(tf.processIMUData, tf.processVisionData, ev[first_vis + 34: first_vis + 45])
        tf.collect_phases()
    if os.environ.get("PROBE_CLONE"):
        keep = tf.Sll.clone()
    tf.close()
    be.close()
    del tf, be
    gc.collect()
fb = binding.FilterBatch(d, capacity=N, batch=1)
run(lambda s_, w_, a_: fb.process_imu([s_], w_, a_), lambda s_, i_, y_: fb.process_vision([s_], i_, y_), warm)
fb.synchronize()
t0 = time.perf_counter()
run(lambda s_, w_, a_: fb.process_imu([s_], w_, a_), lambda s_, i_, y_: fb.process_vision([s_], i_, y_), timed)
fb.synchronize()
print(f"monolithic filter after {K} partitioned handles (chain_slices={chain}, downdate_slices={dd}): {(time.perf_counter() - t0) * 1e3 / 3:.2f} ms/frame", flush=True)
