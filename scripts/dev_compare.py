"""Dev tool: step the HIP filter and the C++ oracle through the same synthetic stream and print deviations."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eqf_vio_amd import synth, binding
from oracle import binding as ob

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dur = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
prec = int(sys.argv[3]) if len(sys.argv) > 3 else 0
structured = len(sys.argv) > 4 and sys.argv[4] == "structured"  # the structured CPU backend: affordable at N >= 1000
st = synth.make_stream(N, duration=dur)
d = synth.template_settings_dict()
fo = ob.OracleFilter(d, structured=structured)
fg = binding.FilterBatch(d, capacity=max(N, 16), batch=1, precision=prec)
print(binding.lib().eqf_version().decode())
nev = 0
worst = 0
worst_at = None
worst_pos = worst_att = 0.0
nvis = 0
for kind, k in st.events():
    if kind == "imu":
        r = st.imu[k]
        fo.processIMUData(r[0], r[1:4], r[4:7])
        fg.process_imu([r[0]], r[1:4], r[4:7])
    else:
        fo.processVisionData(st.vision_stamps[k], st.ids, st.bearings[k])
        stt = fg.process_vision([st.vision_stamps[k]], st.ids, st.bearings[k])
    nev += 1
    nvis += kind == 'vision'
    if kind == "vision" or nev <= 3 or nev % 50 == 0:
        So = fo.stateCovariance(); Sg = fg.sigma()
        eo = fo.stateEstimate(); eg = fg.state_estimate()
        rel = np.linalg.norm(Sg - So) / np.linalg.norm(So) if So.shape == Sg.shape else float('nan')
        if rel == rel and rel > worst:
            worst, worst_at = rel, (nev, nvis)
        worst_pos = max(worst_pos, float(np.abs(eg['x']-eo['x']).max()))
        qd = float(min(np.abs(eg['q']-eo['q']).max(), np.abs(eg['q']+eo['q']).max()))
        worst_att = max(worst_att, 2.0 * qd)  # small-angle: |dq| ~ angle / 2
        msg = f"{nev:4d} {kind:6s} N={fg.num_landmarks()} |S|={np.linalg.norm(So):9.3e} relS={rel:8.2e} pos={np.abs(eg['x']-eo['x']).max():8.2e} q={np.abs(eg['q']-eo['q']).max():8.2e} v={np.abs(eg['v']-eo['v']).max():8.2e}"
        if fg.num_landmarks() and eo['p'].shape == eg['p'].shape:
            msg += f" p={np.abs(eg['p']-eo['p']).max():8.2e}"
        msg += f" bias={np.abs(fg.bias()-fo.bias()).max():8.2e} err={fg.device_error()}"
        if kind == "vision":
            lo = fo.last_update(); lg = fg.last_update()
            if lo is not None:
                msg += f" | delta={np.abs(lg['delta']-lo['delta']).max():8.2e} gamma={np.abs(lg['gamma']-lo['gamma']).max():8.2e} Gamma={np.abs(lg['Gamma']-lo['Gamma']).max():8.2e}"
        print(msg, flush=True)
print(f"worst relS over {nvis} vision updates: {worst:.3e} at event {worst_at[0]} (vision frame {worst_at[1]})   (north_star tolerance 1e-4)")
print(f"worst pose difference to the oracle: position {worst_pos:.3e} m, attitude {worst_att:.3e} rad   (SURVEY 8d gate: 1e-4 / 1e-4)")
