"""The driver's 20-step shape from the host's side: time of every API call of the timed region and of the final synchronise.
Usage: short_run_trace.py [steps] [warmup]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eqf_vio_amd import binding, synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
warmup = int(sys.argv[2]) if len(sys.argv) > 2 else 5
N = 200
st = synth.make_stream(N, seed=1234, duration=1.0)
ev = list(st.events())[: steps + warmup]
for rep in range(3):
    fb = binding.FilterBatch(synth.template_settings_dict(), capacity=N, batch=1)
    fb.stream_upload(st.imu[:, None], st.vision_stamps[:, None], st.ids, st.bearings[:, None])
    for kind, k in ev[:warmup]:
        (fb.stream_imu if kind == "imu" else fb.stream_vision)(k)
    fb.synchronize()
    t = [time.perf_counter()]
    for kind, k in ev[warmup:]:
        (fb.stream_imu if kind == "imu" else fb.stream_vision)(k)
        t.append(time.perf_counter())
    fb.synchronize()
    t.append(time.perf_counter())
    d = np.diff(np.array(t)) * 1e6
    print("rep", rep, "total %.1f us = %.0f steps/s;" % ((t[-1] - t[0]) * 1e6, steps / (t[-1] - t[0])),
          " ".join("%s%.1f" % ("V" if kind == "vision" else "i", x) for (kind, _), x in zip(ev[warmup:], d[:-1])), "| sync %.1f" % d[-1])
    del fb
