"""Generates tests/golden/large_N2000.npz and large_N4000.npz: the BASELINE sizes no `-m gpu` test can afford an oracle run for
(the structured fp64 oracle needs minutes of one core per frame at N = 4000), frozen as DATA.

Run here (dev container) only:  python tests/golden/make_golden_large.py [N ...]
Source of the numbers: oracle/eqf_oracle.cpp, structured backend (pinned against the dense restatement of the reference's operation
sequence by tests/test_oracle_structured.py).  Each file holds the inputs -- template settings, IMU records, vision stamps, ids, bearings of
the synthetic stream (eqf_vio_amd/synth.py, seed 1234) -- and, after every vision update: pose / velocity / bias, |Sigma|_F, trace(Sigma),
the 11 x 11 base block, `n_samples` entries of Sigma at fixed pseudo-random positions (rows, cols stored), the norms and the first 64 entries of
delta / gamma / Gamma (VIOFilter.cpp:264-297).  Consumers: tests/test_gpu_tiled.py (partitioned filter, N = 4000), tests/test_gpu_configs.py
(monolithic path, N = 2000 / 4000).
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from eqf_vio_amd import synth  # noqa: E402
from oracle import binding as ob  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = {2000: 0.16, 4000: 0.11}  # N -> seconds of stream (three / two vision updates)
N_SAMPLES = 600


def run(N, duration, seed=1234):
    st = synth.make_stream(N, seed=seed, duration=duration)
    d = synth.template_settings_dict()
    fo = ob.OracleFilter(d, structured=True)
    n = 11 + 3 * N
    rng = np.random.default_rng(2024 + N)
    rows = rng.integers(0, n, size=N_SAMPLES).astype(np.int32)
    cols = rng.integers(0, n, size=N_SAMPLES).astype(np.int32)
    rows[:64] = cols[:64] = rng.integers(0, n, size=64)  # (some diagonal entries)
    rows[64:128] = rng.integers(0, 11, size=64)          # (base rows against landmark columns)
    frames, samples, base, lu = [], [], [], []
    t0 = time.time()
    for kind, k in st.events():
        if kind == "imu":
            r = st.imu[k]
            fo.processIMUData(r[0], r[1:4], r[4:7])
        else:
            fo.processVisionData(st.vision_stamps[k], st.ids, st.bearings[k])
            S = fo.stateCovariance()
            e = fo.stateEstimate()
            frames.append(np.concatenate([e["q"], e["x"], e["v"], fo.bias(), [np.linalg.norm(S), np.trace(S)]]))
            samples.append(S[rows, cols].copy())
            base.append(S[:11, :11].copy())
            L = fo.last_update()
            lu.append(np.concatenate([[np.linalg.norm(L["delta"]), np.linalg.norm(L["gamma"]), np.linalg.norm(L["Gamma"])],
                                      L["delta"][:64], L["gamma"][:64], L["Gamma"][:64]]))
            print(f"N={N} vision frame {k}: |S|_F = {np.linalg.norm(S):.6e}  ({time.time() - t0:.0f} s)", flush=True)
            del S
    keys = sorted(k for k in d if not k.startswith("cameraOffset"))
    nf = len(frames)
    np.savez_compressed(
        os.path.join(HERE, f"large_N{N}.npz"),
        setting_names=np.array(keys), setting_values=np.array([float(d[k]) for k in keys]),
        cameraOffset_x=np.asarray(d["cameraOffset_x"], dtype=np.float64), cameraOffset_q=np.asarray(d["cameraOffset_q"], dtype=np.float64),
        imu=st.imu, vision_stamps=st.vision_stamps[:nf], ids=st.ids.astype(np.int32), bearings=st.bearings[:nf],
        frames=np.array(frames), sample_rows=rows, sample_cols=cols, sigma_samples=np.array(samples), sigma_base=np.array(base),
        last_update=np.array(lu), seed=np.array([seed]), duration=np.array([duration]),
    )


if __name__ == "__main__":
    for N in ([int(a) for a in sys.argv[1:]] or sorted(CASES)):
        run(N, CASES[N])
