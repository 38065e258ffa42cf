import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import compeg_amd as ca
from tools import synth
gpu = ca.Gpu.open(0)
for samp in ((2,1),(2,2),(1,1),(1,2)):
    for (w,h,ri) in ((3840,2160,4),(1920,1080,4),(3840,2160,240 if samp[0]==2 else 480)):
        j = synth.make_jpeg(w,h,seed=7,quality=85,ri=ri,sampling=samp)
        img = ca.ImageData(j, allow_sampling=True)
        b = ca.Batch(gpu); b.upload([img])
        for _ in range(3): b.decode(); b.wait()
        b.timing(reset=True); ts=[]
        for _ in range(10):
            b.decode(); b.wait(); ts.append(b.timing(reset=True)[1]*1000)
        print(f"{samp[0]}x{samp[1]} {w}x{h} DRI={ri}: kernel {b.last_kernel()} {np.median(ts):.1f} us", flush=True)
