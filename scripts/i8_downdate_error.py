import sys, numpy as np
sys.path.insert(0, '.')
from eqf_vio_amd import binding, synth, tiled
def run(N, bl, dur, slices):
    d = synth.template_settings_dict()
    st = synth.make_stream(N, duration=dur)
    be = tiled.HipBackend(d, capacity=N)
    tf = tiled.TiledFilter(tiled.ProcessGrid(None, 1, 1, device=be.device), be, bl)
    tf.downdate_slices = slices
    fg = binding.FilterBatch(d, capacity=N, batch=1)
    errs = []
    for kind, k in st.events():
        if kind == "imu":
            r = st.imu[k]; tf.processIMUData(r[0], r[1:4], r[4:7]); fg.process_imu([r[0]], r[1:4], r[4:7])
        else:
            fg.process_vision([st.vision_stamps[k]], st.ids, st.bearings[k]); tf.processVisionData(st.vision_stamps[k], st.ids, st.bearings[k])
            S1, S0 = tf.stateCovariance(), fg.sigma()
            errs.append(float(np.linalg.norm(S1 - S0) / np.linalg.norm(S0)))
    print("per frame:", " ".join("%.1e" % e for e in errs), flush=True)
    print(f"N={N} bl={bl} slices={slices}: worst {max(errs):.2e} at frame {int(np.argmax(errs))}, last {errs[-1]:.2e}, frames {len(errs)}, err flag {be.device_error()}", flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "long":  # does the margin at N = 4000 hold past the fifth frame (the worst one at the smaller sizes)?
    run(4000, 250, 0.46, 6)
    run(1000, 125, 1.0, 6)
    sys.exit(0)
for S in (5, 6, 7):
    run(200, 64, 2.0, S)
for S in (5, 6, 7):
    run(1000, 125, 0.26, S)
for S in (6, 7):
    run(4000, 250, 0.16, S)
