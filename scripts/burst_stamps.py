"""Per-wave, per-tick stamps of k_burst_build (library built with -DEQF_BURST_STAMPS): where a tick's time goes.  Units: 10 ns (the 100 MHz
wall clock, the same on every CU; scripts/burst_fused_stamps.py prints the fused launch in microseconds).  Run with EQF_BURST_FUSED=0 for the two launches."""
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
a = np.array(out[:]).reshape(8, 20, 4)
t0 = a[:, 0, 0].min()
names = ["panel0", "panel1", "panel2", "panel3", "glob+Sbb", "common+F", "lift", "blocks"]
print("tick:      " + " ".join("%6d" % t for t in range(14)))
for w in range(8):
    print("%-9s b " % names[w] + " ".join("%6d" % (a[w, t, 0] - t0) for t in range(14)))
    print("%-9s d " % names[w] + " ".join("%6d" % (a[w, t, 1] - a[w, t, 0]) for t in range(14)))
for w in (4, 5):
    print("%-9s first part " % names[w] + " ".join("%6d" % (a[w, t, 2] - a[w, t, 0]) for t in range(2, 12)))
# the block kernel (k_burst_riccati_ring), one workgroup: per wave and step  math | hand-over | wait at the barrier
r = (C.c_longlong * 256)()
hip.lib().eqf_debug_ring_stamps(r)
r = np.array(r[:]).reshape(4, 64)
for w in range(4):
    t0 = r[w, 63]
    print("ring wave %d: prologue %d (+barrier %d)" % (w, r[w, 0] - t0, r[w, 1] - r[w, 0]), " steps:",
          " ".join("%d|%d|%d" % (r[w, 2 + 3 * s] - (r[w, 1] if s == 0 else r[w, 4 + 3 * (s - 1)]), r[w, 3 + 3 * s] - r[w, 2 + 3 * s], r[w, 4 + 3 * s] - r[w, 3 + 3 * s]) for s in range(12)))
