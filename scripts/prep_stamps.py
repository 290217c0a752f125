"""Per-workgroup durations of k_update_prep64 (library built with -DEQF_PREP_STAMPS)."""
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
out = (C.c_longlong * 1024)()
hip.lib().eqf_debug_prep_stamps(out)
a = np.array(out[:]).reshape(512, 2)
n = int((a[:, 0] != 0).sum())
d = (a[:n, 1] - a[:n, 0])
print("workgroups:", n)
print("durations :", d.tolist())
