"""Where the slot-time of k_chol_resident goes (library built with -DEQF_WAIT_STATS): per role class the number of workgroups, their summed
lifetimes and the part of it spent waiting for a hand-off flag (hoWait, the downdate's gate), over the updates of a short run.
Usage: wait_stats.py [B] [N]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eqf_vio_amd import binding as hip, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
sts = [synth.make_stream(N, seed=100 + b, duration=0.31) for b in range(B)]
fb = hip.FilterBatch(synth.template_settings_dict(), capacity=N, batch=B)
fb.stream_upload(np.stack([s.imu for s in sts], axis=1), np.stack([s.vision_stamps for s in sts], axis=1), sts[0].ids, np.stack([s.bearings for s in sts], axis=1))
ev = list(sts[0].events())
out = (C.c_ulonglong * 48)()
nupd = 0
for i, (kind, k) in enumerate(ev):
    (fb.stream_imu if kind == "imu" else fb.stream_vision)(k)
    if kind == "vision":
        fb.synchronize()
        nupd += 1
        if nupd == 2:  # (the first updates: landmark initialisation, cold code)
            hip.lib().eqf_debug_wait_stats(out, 1)
            nupd0 = nupd
fb.synchronize()
hip.lib().eqf_debug_wait_stats(out, 0)
a = np.array(out[:], dtype=np.float64).reshape(16, 3)
n = nupd - nupd0
names = {0: "S row heads", 1: "S interior tiles", 2: "S right-hand sides", 3: "S first block", 4: "prep roles", 5: "downdate tiles", 7: "(left before a class was set)",
         8: "E row heads", 9: "E interior tiles", 10: "E right-hand sides (lift)", 11: "E first block"}
tot = a[:, 1].sum()
print("B=%d N=%d, %d updates; per update: workgroups, slot-time (us), of it waiting for flags (us), share of all slot-time" % (B, N, n))
for c in range(16):
    if a[c, 2] > 0:
        print("  %-30s %8.0f %12.1f %12.1f   %5.1f %%   (%.1f us per workgroup, %.0f %% waiting)" % (names.get(c, str(c)), a[c, 2] / n, a[c, 1] / 100 / n, a[c, 0] / 100 / n,
              100 * a[c, 1] / tot, a[c, 1] / a[c, 2] / 100, 100 * a[c, 0] / max(a[c, 1], 1)))
print("  all: %.1f us of slot-time per update, %.1f waiting (%.0f %%)" % (tot / 100 / n, a[:, 0].sum() / 100 / n, 100 * a[:, 0].sum() / tot))
