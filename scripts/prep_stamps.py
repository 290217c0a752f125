"""Per-workgroup durations of k_update_prep64, filter 0 of the batch (library built with -DEQF_PREP_STAMPS).
Usage: prep_stamps.py [B] [N] [cs_in_burst]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eqf_vio_amd import binding as hip, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
cs = int(sys.argv[3]) if len(sys.argv) > 3 else 1
sts = [synth.make_stream(N, seed=100 + b, duration=0.16) for b in range(B)]
fb = hip.FilterBatch(synth.template_settings_dict(), capacity=N, batch=B)
fb.debug_option("cs_in_burst", cs)
fb.stream_upload(np.stack([s.imu for s in sts], axis=1), np.stack([s.vision_stamps for s in sts], axis=1), sts[0].ids,
                 np.stack([s.bearings for s in sts], axis=1))
for kind, k in list(sts[0].events())[:34]:
    (fb.stream_imu if kind == "imu" else fb.stream_vision)(k)
fb.synchronize()
out = (C.c_longlong * 1024)()
hip.lib().eqf_debug_prep_stamps(out)
a = np.array(out[:]).reshape(512, 2)
n = int((a[:, 0] != 0).sum())
t0 = a[:n, 0].min()
print("B=%d N=%d cs_in_burst=%d: %d workgroups of filter 0; (start, duration) in us at 100 MHz" % (B, N, cs, n))
print(" ".join("%.1f+%.1f" % ((a[i, 0] - t0) / 100.0, (a[i, 1] - a[i, 0]) / 100.0) for i in range(n)))
