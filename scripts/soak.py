"""Dev tool: long run (default 60 s of data) with landmark churn and the outlier gate on, per-call API; checks the device
error flag, symmetry / positive definiteness of Sigma and that the pose error stays small."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eqf_vio_amd import binding as hip, synth

dur = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
NP = 260
st = synth.make_stream(NP, duration=dur)
F = st.bearings.shape[0]
rng = np.random.default_rng(11)
start = rng.integers(-F // 4, F, size=NP)
length = rng.integers(F // 8, F // 2, size=NP)
start[:60] = -1
length[:60] = 10 * F
d = synth.template_settings_dict()
d["outlierThreshold"] = 0.05
fb = hip.FilterBatch(d, capacity=NP, batch=1)
t0 = time.perf_counter()
n = 0
worst_pos = 0.0
for kind, k in st.events():
    if kind == "imu":
        r = st.imu[k]
        fb.process_imu([r[0]], r[1:4], r[4:7])
    else:
        vis = np.where((start <= k) & (k < start + length))[0]
        fb.process_vision([st.vision_stamps[k]], st.ids[vis].astype(np.int32), st.bearings[k, vis].copy())
        if k % 100 == 99:
            e = fb.state_estimate()
            # the filter's frame is anchored at its first pose: compare displacements
            err = np.linalg.norm((e["x"] - e0) - (st.true_pos[k] - p0)) if "e0" in dir() else 0.0
            worst_pos = max(worst_pos, err)
        if k == 20:
            e0, p0 = fb.state_estimate()["x"].copy(), st.true_pos[k].copy()
    n += 1
fb.synchronize()
dt = time.perf_counter() - t0
S = fb.sigma()
w = np.linalg.eigvalsh(0.5 * (S + S.T))
print(f"{n} calls in {dt:.2f} s = {n/dt:.0f} steps/s; N at end {fb.num_landmarks()}; device error {fb.device_error()}")
print(f"Sigma: asymmetry {np.abs(S - S.T).max():.2e}, min eigenvalue {w.min():.3e}, max {w.max():.3e}")
print(f"worst displacement error vs ground truth (sampled): {worst_pos:.3e} m")
