"""Phases of the row-head workgroups of k_chol_resident (library built with -DEQF_RES_STAMPS): wall-clock stamps (100 MHz) of
the E-chain and S-chain row heads of the LAST update of a short N = 200 run, relative to the first stamp."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from eqf_vio_amd import binding, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1  # filters in the handle (stamps are those of filter 0)
st = synth.make_stream(N, duration=0.3)
fb = binding.FilterBatch(synth.template_settings_dict(), capacity=N, batch=B)
fb.stream_upload(st.imu, st.vision_stamps, st.ids, st.bearings)
for kind, k in st.events():
    (fb.stream_imu if kind == "imu" else fb.stream_vision)(k)
fb.synchronize()
out = (C.c_longlong * (2 * 16 * 16))()
assert binding.lib().eqf_debug_res_stamps(out) == 0
t = np.array(out, dtype=np.int64).reshape(2, 16, 16)
names = ["tiles loaded", "panels applied", "D flag seen", "D loaded + tile in LDS", "solved", "L published", "first column in LDS", "factored", "D published"]
for ch, nm in ((1, "E-chain"), (0, "S-chain")):
    rows = [R for R in range(1, 16) if t[ch, R, 8] > 0]
    if not rows:
        continue
    t0 = t[ch, rows[0], 0]
    print(nm, "(us since the first row head started; phases:", ", ".join(names), ")")
    for R in rows:
        print(f"  H({R:2d})", " ".join(f"{(t[ch, R, i] - t0) / 100.0:7.2f}" for i in range(9)), "  | last panel wait: from",
              " ".join(f"{(t[ch, R, i] - t0) / 100.0:7.2f}" if t[ch, R, i] > 0 else "      -" for i in (9, 10)), " T(R,R-2) published",
              f"{(t[ch, R, 11] - t0) / 100.0:7.2f}" if t[ch, R, 11] > 0 else "-", " | L_{R,R-1} published, late block loaded, stage-2 flag seen, stages 0-2 loaded:",
              " ".join(f"{(t[ch, R, i] - t0) / 100.0:7.2f}" if t[ch, R, i] > 0 else "      -" for i in (12, 13, 14, 15)))

w = t[1, 15]
if w[5] > 0:
    t0 = t[1, [R for R in range(1, 15) if t[1, R, 8] > 0][0], 0]
    print("last right-hand-side workgroup of the E-chain (lift): ready, D flag seen, solved, published, sums collected, lift done")
    print("   ", " ".join(f"{(w[i] - t0) / 100.0:7.2f}" for i in range(6)))

x = t[0, 15]
if x[0] > 0:
    t0 = t[1, [R for R in range(1, 15) if t[1, R, 8] > 0][0], 0]
    print("downdate tiles 0 / 27 / 54 as workgroups of their own (grid larger than the chip): dispatched, last Y tile seen, done")
    for k in range(3):
        print("   ", " ".join(f"{(x[3 * k + i] - t0) / 100.0:7.2f}" for i in range(3)))

# factor64's own stamps of every wave (library built with -DEQF_RES_STAMPS -DEQF_F64_STAMPS): shader cycles since wave 0 entered stage 0
if hasattr(binding.lib(), "eqf_debug_res_f64_stamps"):
    o2 = (C.c_longlong * (2 * 16 * 128))()
    assert binding.lib().eqf_debug_res_f64_stamps(o2) == 0
    f = np.array(o2, dtype=np.int64).reshape(2, 16, 4, 4, 8)
    for ch, nm in ((1, "E-chain"), (0, "S-chain")):
        for R in (2, 5):
            if f[ch, R, 0, 0, 0] == 0:
                continue
            t0 = f[ch, R, 0, 0, 0]
            print(f"factor64 inside {nm} H({R}): per wave and stage: start | P done | A passed | U done | B passed | rows loaded | pivots done  (cycles)")
            for w in range(4):
                for j in range(4):
                    print(f"   wave {w} stage {j}:", " ".join(f"{(f[ch, R, w, j, k] - t0) if f[ch, R, w, j, k] else -1:7d}" for k in range(7)))
