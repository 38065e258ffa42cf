import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import compeg_amd as ca
from tools import synth
gpu = ca.Gpu.open(0)
for (w, h) in ((1080, 1920), (1088, 1920), (1072, 1920), (1920, 1080), (360, 640), (368, 640)):
    frames = [synth.make_jpeg(w, h, seed=70 + i, quality=85, ri=4) for i in range(8)]
    imgs = [ca.ImageData(j) for j in frames]
    n = 256
    b = ca.Batch(gpu); b.upload([imgs[i % 8] for i in range(n)])
    for _ in range(3): b.decode(); b.wait()
    b.timing(reset=True); ts = []
    for _ in range(8):
        b.decode(); b.wait(); ts.append(b.timing(reset=True)[1] * 1000)
    print(f"{w}x{h} x{n}: {b.last_kernel()} {np.median(ts):.1f} us  {n*w*h/np.median(ts)/1e3:.1f} Gpx/s", flush=True)
