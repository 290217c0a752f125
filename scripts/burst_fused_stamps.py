"""Wall-clock stamps (100 MHz, common to all CUs) of a burst: builder workgroup 0 per wave and tick, block workgroup (tile) 9 per wave and
step -- library built with -DEQF_BURST_STAMPS (load it with EQF_VIO_AMD_LIB=...).  Times in microseconds from the builder's first stamp."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eqf_vio_amd import binding as hip, synth
N = 200
st = synth.make_stream(N, duration=0.3)
fb = hip.FilterBatch(synth.template_settings_dict(), capacity=N, batch=1)
fb.stream_upload(st.imu, st.vision_stamps, st.ids, st.bearings)
ev = list(st.events())
for kind, k in ev[:50]:
    (fb.stream_imu if kind == "imu" else fb.stream_vision)(k)
fb.synchronize()
out = (C.c_longlong * 640)()
hip.lib().eqf_debug_burst_stamps(out)
a = np.array(out[:]).reshape(8, 20, 4).astype(np.float64) / 100.0
t0 = a[:, 0, 0].min()
names = ["panel", "Lw", "Sbb", "cam+store", "recurrence", "F", "lift", "D,Lv"]
print("builder workgroup 0, tick:   " + " ".join("%6d" % t for t in range(14)))
for w in range(8):
    print("%-10s begin " % names[w] + " ".join("%6.2f" % (a[w, t, 0] - t0) for t in range(14)))
    print("%-10s busy  " % names[w] + " ".join("%6.2f" % (a[w, t, 1] - a[w, t, 0]) for t in range(14)))
print("cam+store: older stores done + published | records stored | cam done   (from the tick's begin)")
print("                 " + " ".join("%4.2f|%4.2f|%4.2f" % (a[3, t, 2] - a[3, t, 0], a[3, t, 3] - a[3, t, 0], a[3, t, 1] - a[3, t, 0]) for t in range(3, 13)))
r = (C.c_longlong * 256)()
hip.lib().eqf_debug_ring_stamps(r)
r = np.array(r[:]).reshape(4, 64).astype(np.float64) / 100.0
for w in range(4):
    print("block tile 9 wave %d: starts %.2f;  per step: data there | arithmetic done" % (w, r[w, 63] - t0))
    print("    " + "  ".join("%.2f|%.2f" % (r[w, 2 + 3 * s] - t0, r[w, 3 + 3 * s] - t0) for s in range(12)))
