"""Dev tool: cost of landmark churn.  A pool of 260 landmarks, ~200 visible, a few entering / leaving every frame
(VIOFilter.cpp:345-443 path: removeOldLandmarks, addNewLandmarks with median depth); per-call API (host buffers)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eqf_vio_amd import binding as hip, synth

NP, dur = 260, 3.0
st = synth.make_stream(NP, duration=dur)
F = st.bearings.shape[0]
rng = np.random.default_rng(3)
# visibility windows: each landmark visible for a long window; about 5 changes per frame
start = rng.integers(-F, F, size=NP)
length = rng.integers(F // 2, F, size=NP)
start[:120] = -1
length[:120] = 10 * F
d = synth.template_settings_dict()
def run(mode):
    dd = dict(d)
    if "gate" in mode:
        dd["outlierThreshold"] = float(os.environ.get("GATE_THR", "0.01"))  # the reference default is 0.01 (VIOFilterSettings.h)
    fb = hip.FilterBatch(dd, capacity=NP, batch=1)
    ev = list(st.events())
    nvis, nch = 0, 0
    prev = None
    t0 = time.perf_counter()
    for kind, k in ev:
        if kind == "imu":
            r = st.imu[k]
            fb.process_imu([r[0]], r[1:4], r[4:7])
        else:
            vis = np.arange(200) if mode.startswith("fixed") else np.where((start <= k) & (k < start + length))[0]
            if prev is not None:
                nch += len(set(vis) ^ set(prev))
            prev = vis
            fb.process_vision([st.vision_stamps[k]], st.ids[vis].astype(np.int32), st.bearings[k, vis].copy())
            nvis += 1
    fb.synchronize()
    dt = time.perf_counter() - t0
    return dt, len(ev), fb.num_landmarks(), nch / max(nvis - 1, 1), fb.device_error()


if os.environ.get("CHURN_MODE"):  # one mode, once after a warm-up (for a rocprofv3 kernel summary: scripts/churn_profile.sh)
    run("fixed")
    dt, n, N, ch, err = run(os.environ["CHURN_MODE"])
    print(f"{os.environ['CHURN_MODE']:20s}: {n} calls, {dt*1e3:.1f} ms = {n/dt:.0f} steps/s, N at end {N}, landmark changes per frame {ch:.1f}, device error {err}")
    sys.exit(0)
run("fixed")  # warm-up: module load, first launches
for mode in ("fixed", "fixed+outlier-gate", "churn", "churn+outlier-gate"):
    res = [run(mode) for _ in range(3)]
    dt, n, N, ch, err = min(res)
    print(f"{mode:20s}: {n} calls, best of 3 {dt*1e3:.1f} ms = {n/dt:.0f} steps/s, N at end {N}, landmark changes per frame {ch:.1f}, device error {err}")
