"""Dev tool: cycle stamps inside k_propagate (library built with -DEQF_PROP_STAMPS)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eqf_vio_amd import binding as hip, synth

N = 200
st = synth.make_stream(N, duration=0.3)
fb = hip.FilterBatch(synth.template_settings_dict(), capacity=N, batch=1)
fb.stream_upload(st.imu, st.vision_stamps, st.ids, st.bearings)
ev = list(st.events())
out = (C.c_longlong * 32)()
for kind, k in ev[:40]:
    if kind == "imu":
        fb.stream_imu(k)
    else:
        fb.stream_vision(k)
fb.synchronize()
hip.lib().eqf_debug_prop_stamps(out)
a = np.array(out[:]).reshape(4, 8)
names = ["wg0 wave0", "wg0 wave2", "diag(1,1) wave0", "diag(1,1) wave2"]
for i, n in enumerate(names):
    t = a[i]
    print(f"{n:18s} chain/loads done +{t[1]-t[0]:6d}  barrier +{t[2]-t[0]:6d}  blocks done +{t[3]-t[0]:6d}  end +{t[4]-t[0]:6d}  stepGlobal +{t[5]-t[0]:6d}")
