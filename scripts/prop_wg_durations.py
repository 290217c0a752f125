import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eqf_vio_amd import binding as hip, synth
N = 200
st = synth.make_stream(N, duration=0.3)
fb = hip.FilterBatch(synth.template_settings_dict(), capacity=N, batch=1)
fb.stream_upload(st.imu, st.vision_stamps, st.ids, st.bearings)
ev = list(st.events())
for kind, k in ev[:38]:
    (fb.stream_imu if kind == "imu" else fb.stream_vision)(k)
fb.synchronize()
out = (C.c_longlong * 1024)()
hip.lib().eqf_debug_prop_stamps(out)
a = np.array(out[:]).reshape(512, 2)[:201]
d = a[:, 1] - a[:, 0]
print("tiles   : median", int(np.median(d[:169])), "max", d[:169].max(), "at", int(d[:169].argmax()))
print("base blk:", d[169], " state:", d[170])
print("row tails:", d[171:184])
print("col tails:", d[184:197])
print("landmark:", d[197:201])
