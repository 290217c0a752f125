"""Shared helpers of the parity tests: golden-fixture loading and running a filter over a fixture."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BOOL_KEYS = ("useInnovationLift", "useDiscreteInnovationLift", "useDiscreteVelocityLift", "fastRiccati")


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    settings = {str(k): float(v) for k, v in zip(d["setting_names"], d["setting_values"])}
    for k in BOOL_KEYS:
        settings[k] = bool(settings[k])
    settings["cameraOffset_x"] = d["cameraOffset_x"]
    settings["cameraOffset_q"] = d["cameraOffset_q"]
    return d, settings


def events_of(imu, vision_stamps):
    """Event interleave of the reference's runner (eqf_vio/src/main.cpp:113): IMU while imu.stamp < meas.stamp."""
    k = f = 0
    while k < len(imu) and f < len(vision_stamps):
        if imu[k, 0] < vision_stamps[f]:
            yield ("imu", k)
            k += 1
        else:
            yield ("vision", f)
            f += 1


def frame_record(q, x, v, bias, S, n):
    return np.concatenate([q, x, v, bias, [np.linalg.norm(S), n]])


def run_oracle_on_golden(ob, d, settings, capture=(), structured=False):
    """C++ oracle over a golden fixture -> per-frame records, final filter, captured internals."""
    f = ob.OracleFilter(settings, structured=structured)
    frames, internals = [], {}
    for kind, k in events_of(d["imu"], d["vision_stamps"]):
        if kind == "imu":
            r = d["imu"][k]
            f.processIMUData(r[0], r[1:4], r[4:7])
        else:
            nb = int(d["meas_nb"][k])
            f.processVisionData(d["vision_stamps"][k], d["meas_ids"][k, :nb], d["meas_y"][k, :nb])
            e = f.stateEstimate()
            frames.append(frame_record(e["q"], e["x"], e["v"], f.bias(), f.stateCovariance(), f.N))
            if k in capture:
                internals[k] = f.last_update()
    return np.array(frames), f, internals


def run_hip_on_golden(binding, d, settings, capacity, capture=(), precision=0):
    fb = binding.FilterBatch(settings, capacity=capacity, batch=1, precision=precision)
    frames, internals = [], {}
    for kind, k in events_of(d["imu"], d["vision_stamps"]):
        if kind == "imu":
            r = d["imu"][k]
            fb.process_imu([r[0]], r[1:4], r[4:7])
        else:
            nb = int(d["meas_nb"][k])
            fb.process_vision([d["vision_stamps"][k]], d["meas_ids"][k, :nb], d["meas_y"][k, :nb])
            e = fb.state_estimate()
            frames.append(frame_record(e["q"], e["x"], e["v"], fb.bias(), fb.sigma(), fb.num_landmarks()))
            if k in capture:
                internals[k] = fb.last_update()
    return np.array(frames), fb, internals


def rel_fro(A, B):
    return np.linalg.norm(A - B) / np.linalg.norm(B)


def drive_pair(stream, a_imu, a_vis, b_imu, b_vis, on_frame=None):
    """Feed one synthetic stream to two filters (callables taking the event's arrays); on_frame(k) after each vision call."""
    for kind, k in stream.events():
        if kind == "imu":
            r = stream.imu[k]
            a_imu(r[0], r[1:4], r[4:7])
            b_imu(r[0], r[1:4], r[4:7])
        else:
            a_vis(stream.vision_stamps[k], stream.ids, stream.bearings[k])
            b_vis(stream.vision_stamps[k], stream.ids, stream.bearings[k])
            if on_frame is not None:
                on_frame(k)


def check_large_golden(d, f, est, bias, S, last=None, what=""):
    """One vision frame of a tests/golden/large_N*.npz fixture (structured fp64 oracle, tests/golden/make_golden_large.py) against a filter's
    outputs: pose / velocity / bias, |Sigma|_F, trace, the 11 x 11 base block, 600 sampled entries of Sigma, and -- when given -- delta / gamma /
    Gamma of that update (norms + first 64 entries).  Returns the worst relative deviation of the covariance quantities."""
    ref = d["frames"][f]
    assert np.abs(est["q"] - ref[0:4]).max() < 1e-8 and np.abs(est["x"] - ref[4:7]).max() < 1e-8, (what, f, "pose")
    assert np.abs(est["v"] - ref[7:10]).max() < 1e-8 and np.abs(bias - ref[10:16]).max() < 1e-8, (what, f, "velocity / bias")
    fro, tr = float(np.linalg.norm(S)), float(np.trace(S))
    w = max(abs(fro / ref[16] - 1.0), abs(tr / ref[17] - 1.0))
    smp, want = S[d["sample_rows"], d["sample_cols"]], d["sigma_samples"][f]
    w = max(w, float(np.abs(smp - want).max() / np.abs(want).max()))
    w = max(w, float(np.linalg.norm(S[:11, :11] - d["sigma_base"][f]) / np.linalg.norm(d["sigma_base"][f])))
    assert w < 1e-8, (what, f, w)
    if last is not None and f > 0:  # (the first frame's residual is zero: its landmarks were initialised on their bearings)
        lu = d["last_update"][f]
        for i, (key, tol) in enumerate((("delta", 1e-9), ("gamma", 1e-7), ("Gamma", 1e-7))):
            v = last[key]
            assert abs(np.linalg.norm(v) / lu[i] - 1.0) < tol, (what, f, key)
            assert np.abs(v[:64] - lu[3 + 64 * i: 3 + 64 * (i + 1)]).max() < tol * max(lu[i], 1.0), (what, f, key)
    return w
