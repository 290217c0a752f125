"""Stamps of the diagonal workgroup (S-chain) of every k_chol_step64 launch of one update (library built with
-DEQF_STEP64_STAMPS): loads | panel solves | first 16 columns updated | the four factorisation stages (pivot chain P, tile update U)."""
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
hip.lib().eqf_debug_step64_stamps(out)
a = np.array(out[:]).reshape(64, 16)
for K in range(6):
    r = a[K]
    print("K=%d  loads %d  solves %d  upd0 %d  | stages (P,U): %s | end %d  total %d" % (
        K, r[1] - r[0], r[2] - r[1], r[3] - r[2], " ".join("%d,%d" % (r[8 + 2 * j] - (r[3] if j == 0 else r[8 + 2 * j - 1]), r[9 + 2 * j] - r[8 + 2 * j]) for j in range(4)),
        r[4] - r[15], r[4] - r[0]))
r = a[40]
print("last rhs workgroup (K = nb-1): loads %d  solve %d  stores+reductions %d  innovation lift %d  total %d" % (r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[4] - r[0]))
