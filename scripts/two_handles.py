"""Dev tool (verdict r3 item 3): B filters of N = 200 on ONE GPU as G handles of B / G filters each, driven from one thread -- each handle
has its own stream, so handle A's IMU burst + prep can run under handle B's latency-bound update kernel.  stagger: handle g runs g *
(11 / G) events ahead of handle 0 (so the handles' vision frames are spread over a frame period); 0: lockstep.
    python scripts/two_handles.py [B=8] [steps=880]"""
import os
import sys
import time

for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from eqf_vio_amd import binding, shard, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 880
N, warm = 200, 110
imu, vst, bear, events = shard.scatter_streams(None, 0, 1, N, B, steps + warm + 22, device=None)
ids = np.arange(N, dtype=np.int32)
d = synth.template_settings_dict()


def run(G, stagger):
    per = B // G
    hs = []
    for g in range(G):
        fb = binding.FilterBatch(d, capacity=N, batch=per, device=0)
        sl = slice(g * per, (g + 1) * per)
        fb.stream_upload(imu[:, sl], vst[:, sl], ids, bear[:, sl])
        hs.append(fb)
    off = [int(round(g * 11.0 / G)) if stagger else 0 for g in range(G)]

    def play(a, b):
        for i in range(a, b):
            for g, fb in enumerate(hs):
                kind, k = events[i + off[g]]
                (fb.stream_imu if kind == "imu" else fb.stream_vision)(k)

    # (a staggered handle starts `off` events into the stream: bring it there first)
    for g, fb in enumerate(hs):
        for i in range(off[g]):
            kind, k = events[i]
            (fb.stream_imu if kind == "imu" else fb.stream_vision)(k)
    play(0, warm)
    for fb in hs:
        fb.synchronize()
    t0 = time.perf_counter()
    play(warm, warm + steps)
    for fb in hs:
        fb.synchronize()
    dt = time.perf_counter() - t0
    err = [fb.device_error() for fb in hs]
    del hs
    return steps * B / dt, err


run(1, 0)
for G in [g for g in (1, 2, 4, 8) if B % g == 0 and g <= B]:
    for stagger in ((0,) if G == 1 else (0, 1)):
        v = max(run(G, stagger) for _ in range(3))
        print(f"B={B} as {G} handle(s) x {B // G} filters, stagger={stagger}: {v[0]:9.0f} steps/s  err {v[1]}", flush=True)
