"""Per-workgroup durations of the k_chol_step64 launches of one update (library built with -DEQF_CHOL_WG_STAMPS)."""
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
t = (C.c_longlong * (16 * 256 * 2))()
info = (C.c_int * (16 * 256 * 4))()
hip.lib().eqf_debug_chol_wg(t, info)
t = np.array(t[:]).reshape(16, 256, 2)
info = np.array(info[:]).reshape(16, 256, 4)
for K in range(11):
    m = t[K, :, 0] != 0
    if not m.any():
        continue
    d = t[K, m, 1] - t[K, m, 0]
    inf = info[K, m]
    order = np.argsort(-d)[:6]
    print("K=%d  workgroups %d  longest:" % (K, m.sum()), ", ".join("%d (chain %d %s R=%d C=%d)" % (d[o], inf[o, 0], "rhs" if inf[o, 1] else "A", inf[o, 2], inf[o, 3]) for o in order),
          " median %d" % np.median(d))
