#!/usr/bin/env python
"""Soak of the 2-D partitioned filter (1 x 1 grid) with landmark churn and the outlier gate: N landmarks in the pool, every frame 1 % of
them out of view (a rotating window, so every landmark leaves and comes back as a new one), gate at the reference default 0.01, IMU bursts
on.  Every 10th frame: ids and Sigma against the single-GPU product path (its compacting churn), symmetry, smallest eigenvalue (N <= 1500),
number of inactive slots.   python scripts/tiled_soak.py [N] [seconds] [block]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from eqf_vio_amd import binding, synth, tiled  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
bl = int(sys.argv[3]) if len(sys.argv) > 3 else 125
d = synth.template_settings_dict()
d["outlierThreshold"] = 0.01
st = synth.make_stream(N, duration=seconds + 0.011)
be = tiled.HipBackend(d, capacity=N)
tf = tiled.TiledFilter(tiled.ProcessGrid(None, 1, 1, device=be.device), be, bl)
fg = binding.FilterBatch(d, capacity=N, batch=1)
turn, f, worst = max(N // 100, 1), 0, 0.0
t0 = time.time()
print(f"# N = {N}, {seconds} s of stream, blocks of {bl}, {turn} landmarks leave / enter per frame, gate 0.01")
for kind, k in st.events():
    if kind == "imu":
        r = st.imu[k]
        tf.processIMUData(r[0], r[1:4], r[4:7])
        fg.process_imu([r[0]], r[1:4], r[4:7])
        continue
    vis = np.ones(N, dtype=bool)
    vis[(np.arange(turn) + f * turn) % N] = False
    f += 1
    ids, y = st.ids[vis], st.bearings[k][vis]
    assert tf.processVisionData(st.vision_stamps[k], ids, y) == 0
    fg.process_vision([st.vision_stamps[k]], ids, y)
    if f % 10 == 0 or f == 1:
        assert np.array_equal(tf.ids, fg.ids())
        S, Sg = tf.stateCovariance(), fg.sigma()
        rel = float(np.linalg.norm(S - Sg) / np.linalg.norm(Sg))
        worst = max(worst, rel)
        sym = float(np.abs(S - S.T).max() / np.abs(S).max())
        eig = float(np.linalg.eigvalsh(0.5 * (S + S.T))[0]) if N <= 1500 else float("nan")
        holes = int((~tf.taken[: tf.nslots]).sum())
        e1, e2 = tf.stateEstimate(), fg.state_estimate()
        print(f"frame {f:4d}: landmarks {len(tf.ids):5d}  slots {tf.nslots:5d}  holes {holes:4d}  rel |S - S_single_gpu| {rel:.2e}  asym {sym:.1e}  "
              f"min eig {eig:.3e}  |x - x_sg| {np.abs(e1['x'] - e2['x']).max():.1e}  churn so far {tf.churn_stats}")
print(f"# worst rel difference {worst:.2e}; device error flags: tiled {be.device_error()}, single-GPU {fg.device_error()}; {time.time() - t0:.1f} s wall")
