"""Dev tool: HIP path vs C++ oracle over many landmark counts / seeds (block-boundary cases of the 64-wide chains included)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eqf_vio_amd import binding as hip, synth
from oracle import binding as ob

Ns = [1, 2, 3, 7, 19, 20, 21, 22, 31, 32, 33, 42, 43, 63, 64, 65, 85, 86, 96, 97, 106, 107, 127, 128, 129, 150]
worst = 0.0
for idx, N in enumerate(Ns):
    seed = 100 + idx
    st = synth.make_stream(N, seed=seed, duration=0.35)
    d = synth.template_settings_dict()
    if N < 2:
        d["useInnovationLift"] = False  # bundleLift is rank deficient with one landmark
    fo = ob.OracleFilter(d)
    fg = hip.FilterBatch(d, capacity=N, batch=1)
    w = 0.0
    for kind, k in st.events():
        if kind == "imu":
            r = st.imu[k]
            fo.processIMUData(r[0], r[1:4], r[4:7])
            fg.process_imu([r[0]], r[1:4], r[4:7])
        else:
            fo.processVisionData(st.vision_stamps[k], st.ids, st.bearings[k])
            fg.process_vision([st.vision_stamps[k]], st.ids, st.bearings[k])
            So, Sg = fo.stateCovariance(), fg.sigma()
            w = max(w, np.linalg.norm(Sg - So) / np.linalg.norm(So))
    eo, eg = fo.stateEstimate(), fg.state_estimate()
    pe = np.abs(eo["x"] - eg["x"]).max()
    print(f"N={N:4d} seed={seed} worst relS={w:.2e} pos diff={pe:.2e} err={fg.device_error()}", flush=True)
    worst = max(worst, w)
    assert w < 1e-7 and pe < 1e-8 and fg.device_error() == 0, N
print("worst", worst)
