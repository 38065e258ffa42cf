import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import compeg_amd as ca
from tools import synth
gpu = ca.Gpu.open(0)
frames = [synth.make_jpeg(3840, 2160, seed=70 + i, quality=85, ri=4) for i in range(8)]
imgs = [ca.ImageData(j) for j in frames]
out = []
for n in (1, 2, 3, 4, 5, 6, 8, 12, 16, 32, 64):
    b = ca.Batch(gpu); b.upload([imgs[i % 8] for i in range(n)])
    for _ in range(5): b.decode(); b.wait()
    b.timing(reset=True); ts = []
    for _ in range(12):
        b.decode(); b.wait(); ts.append(b.timing(reset=True)[1] * 1000)
    out.append(f"{n}:{b.last_kernel()} {np.median(ts):.1f}")
print("  ".join(out), flush=True)
