import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eqf_vio_amd import binding, synth, shard
N=200
streams, events = shard.build_streams(N, 1, 0, 2420)
imu, vst, bear = shard.pack(streams)
fb = binding.FilterBatch(synth.template_settings_dict(), capacity=N, batch=1)
fb.stream_upload(imu, vst, np.arange(N,dtype=np.int32), bear)
for kind,k in events[:220]: (fb.stream_imu if kind=='imu' else fb.stream_vision)(k)
fb.synchronize(); fb.profile_enable(True)
for kind,k in events[220:]: (fb.stream_imu if kind=='imu' else fb.stream_vision)(k)
print({k:(v[0], round(v[1]*1e3/max(v[0],1),2)) for k,v in fb.profile().items() if v[0]})
