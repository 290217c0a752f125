#!/usr/bin/env python
"""8 filters of N = 200 on one GPU as ONE handle against G handles of 8 / G filters each, the handles' events issued alternately from one host
thread (every handle has its own stream: one group's latency-chain update next to another group's burst).  Round 4 measured two handles
slower (NOTES R4.5); asked again with round 6's kernels.   python scripts/split_batch_probe.py [total=8] [steps=880]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from eqf_vio_amd import binding, shard, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 880
warmup, N = 110, 200
for G in (1, 2, 4, 1, 2):
    per = B // G
    fbs, evs = [], None
    for g in range(G):
        imu, vst, bear, events = shard.scatter_streams(None, 0, 1, N, per, steps + warmup, device=None)
        fb = binding.FilterBatch(synth.template_settings_dict(), capacity=N, batch=per, device=0)
        fb.stream_upload(imu, vst, np.arange(N, dtype=np.int32), bear)
        fbs.append(fb)
        evs = events

    def run(events, stagger):
        # stagger: group g runs g * stagger events behind group 0, so that its update falls next to another group's burst
        n = len(events)
        for i in range(n + stagger * (G - 1)):
            for g, fb in enumerate(fbs):
                j = i - g * stagger
                if 0 <= j < n:
                    kind, k = events[j]
                    (fb.stream_imu if kind == "imu" else fb.stream_vision)(k)

    for stagger in ((0,) if G == 1 else (0, 5)):
        for fb in fbs:
            fb.reset() if hasattr(fb, "reset") else None
        run(evs[:warmup], stagger)
        for fb in fbs:
            fb.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(evs[warmup:], stagger)
        for fb in fbs:
            fb.synchronize()
        dt = time.perf_counter() - t0
        err = [fb.device_error() for fb in fbs]
        print(f"{B} filters as {G} handle(s) of {per}, stagger {stagger}: {B * steps / dt / 1e3:7.1f} k steps/s  ({dt * 1e6 / (steps / 11):6.1f} us per frame)  err {err}", flush=True)
        break_after_first = True
        if break_after_first and stagger == 0 and G > 1:
            continue
    del fbs
