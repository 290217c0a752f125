import sys, os, time, threading, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eqf_vio_amd import binding, synth, shard
N=200
streams, events = shard.build_streams(N, 1, 0, 2200)
imu, vst, bear = shard.pack(streams)
fb = binding.FilterBatch(synth.template_settings_dict(), capacity=N, batch=1)
fb.stream_upload(imu, vst, np.arange(N,dtype=np.int32), bear)
stop=False
def smi():
    while not stop:
        out=subprocess.run("rocm-smi --showclocks --showpower --showuse 2>&1 | grep -E 'sclk|Power|GPU use'", shell=True, capture_output=True, text=True).stdout
        print(" | ".join(l.split(':',1)[1].strip() for l in out.strip().splitlines()), flush=True)
        time.sleep(1.0)
t=threading.Thread(target=smi); t.start()
t0=time.time(); n=0
while time.time()-t0<6:
    fb.reset()
    for kind,k in events:
        (fb.stream_imu if kind=='imu' else fb.stream_vision)(k)
    fb.synchronize(); n+=len(events)
print("steps/s", n/(time.time()-t0))
stop=True; t.join()
if len(sys.argv)>1:
    print(subprocess.run(sys.argv[1], shell=True, capture_output=True, text=True))
    stop=False; t=threading.Thread(target=smi); t.start()
    t0=time.time(); n=0
    while time.time()-t0<5:
        fb.reset()
        for kind,k in events:
            (fb.stream_imu if kind=='imu' else fb.stream_vision)(k)
        fb.synchronize(); n+=len(events)
    print("steps/s after", n/(time.time()-t0))
    stop=True; t.join()
