import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import compeg_amd as ca
from tools import synth
gpu = ca.Gpu.open(0)
for (w, h, ri, samp) in ((1912, 1088, 8, (1, 2)), (1920, 1088, 8, (1, 2)), (1912, 1088, 9, (1, 2)), (1912, 1088, 8, (1, 1)), (1912, 1088, 16, (1, 1))):
    frames = [synth.make_jpeg(w, h, seed=70 + i, quality=85, ri=ri, sampling=samp) for i in range(8)]
    imgs = [ca.ImageData(j, allow_sampling=True) for j in frames]
    b = ca.Batch(gpu); b.upload([imgs[i % 8] for i in range(256)])
    for _ in range(3): b.decode(); b.wait()
    b.timing(reset=True); ts = []
    for _ in range(8):
        b.decode(); b.wait(); ts.append(b.timing(reset=True)[1] * 1000)
    print(f"{samp} {w}x{h} DRI={ri}: {b.last_kernel()} {np.median(ts):.1f} us", flush=True)
