"""Generates the golden fixtures under tests/golden/ with the fp64 numpy restatement (oracle/eqf_numpy.py).

Run here (dev container) only:  python tests/golden/make_golden.py
The reference itself cannot be built in this image (Eigen 3 / yaml-cpp absent), so these vectors come from the
restatement that the property tests of tests/test_oracle_properties.py pin; they freeze its behaviour so that
the C++ oracle and the HIP path are compared against committed numbers, not against each other only.
Each .npz holds the inputs (settings, IMU records, per-frame ids + bearings) and the expected outputs
(per-vision-frame pose / velocity / bias / |Sigma|_F / landmark ids, final Sigma, and the update internals
delta / gamma / Gamma of selected frames).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from eqf_vio_amd import synth  # noqa: E402
from oracle import eqf_numpy as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def run(N, duration, overrides, churn, name, seed=1234):
    st = synth.make_stream(N, seed=seed, duration=duration)
    d = synth.template_settings_dict()
    d.update(overrides)
    dd = dict(d)
    cx, cq = dd.pop("cameraOffset_x"), dd.pop("cameraOffset_q")
    s = O.Settings(**dd)
    s.cameraOffset = O.SE3(cq, cx)
    f = O.VIOFilter(s)
    meas = synth.churn_measurements(st, outlier_frames=(5, 9)) if churn else [(st.ids, st.bearings[k]) for k in range(len(st.vision_stamps))]
    frames = []
    internals = {}
    for kind, k in st.events():
        if kind == "imu":
            r = st.imu[k]
            f.processIMUData(O.IMUVelocity(r[0], r[1:4], r[4:7]))
        else:
            ids, y = meas[k]
            f.processVisionData(st.vision_stamps[k], ids, y)
            e = f.stateEstimate()
            frames.append(np.concatenate([e.pose.q, e.pose.x, e.velocity, f.inputBias, [np.linalg.norm(f.Sigma), len(f.xi0.ids)]]))
            if k in (1, 4, len(st.vision_stamps) - 1) and f.last:
                internals[f"delta_{k}"] = f.last["delta"]
                internals[f"gamma_{k}"] = f.last["gamma"]
                internals[f"Gamma_{k}"] = f.last["Gamma"] if f.last["Gamma"] is not None else np.zeros(0)
                internals[f"ids_{k}"] = f.xi0.ids.copy()
    nbmax = max(len(i) for i, _ in meas)
    mids = -np.ones((len(meas), nbmax), dtype=np.int32)
    my = np.zeros((len(meas), nbmax, 3))
    mnb = np.zeros(len(meas), dtype=np.int32)
    for k, (i, y) in enumerate(meas):
        mnb[k] = len(i)
        mids[k, : len(i)] = i
        my[k, : len(i)] = y
    keys = sorted(k for k in d if not k.startswith("cameraOffset"))
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"),
        setting_names=np.array(keys), setting_values=np.array([float(d[k]) for k in keys]),
        cameraOffset_x=d["cameraOffset_x"], cameraOffset_q=d["cameraOffset_q"],
        imu=st.imu, vision_stamps=st.vision_stamps, meas_nb=mnb, meas_ids=mids, meas_y=my,
        frames=np.array(frames), final_sigma=f.Sigma, final_ids=f.xi0.ids, final_p0=f.xi0.p,
        final_Qq=np.array([Q.q for Q in f.X.Q]).reshape(-1, 4), final_Qa=np.array([Q.a for Q in f.X.Q]),
        **internals,
    )
    print(name, "frames", len(frames), "final N", len(f.xi0.ids), "|Sigma|", np.linalg.norm(f.Sigma))


if __name__ == "__main__":
    run(5, 0.8, {}, False, "stream_N5")
    run(25, 1.0, {}, False, "stream_N25")
    run(12, 1.0, {"outlierThreshold": 0.01}, True, "churn_N12")
    run(6, 0.6, {"useDiscreteVelocityLift": False, "useDiscreteInnovationLift": False}, False, "flags_continuous_N6")
    run(6, 0.6, {"useInnovationLift": False, "fastRiccati": True}, False, "flags_nolift_fast_N6")
