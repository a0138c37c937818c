import sys, time, numpy as np
sys.path.insert(0,'.')
from ffmpeg_ffv2_amd import FFV2Encoder, frames as synth
for (W,H,fmt,P,depth,qp) in [(1920,1080,"yuv444p",3,8,16),(1280,720,"yuv444p10le",3,10,64),(960,540,"gray",1,8,4)]:
    enc=FFV2Encoder(W,H,fmt,device=0,max_batch=2)
    fr=np.stack([synth.noise(n,P,H,W,depth) for n in range(2)])
    dev=enc.upload(fr)
    host=enc.encode_batch_to_host(dev,qp=qp)
    enc.set_device_coder(True)
    t0=time.time(); got=enc.encode_batch_to_host(dev,qp=qp); dt=time.time()-t0
    enc.set_device_coder(False)
    print(W,H,fmt,qp, [len(p) for p in host], got==host, "%.2f s"%dt, flush=True)
    assert got==host
    enc.close()
print("ok")
