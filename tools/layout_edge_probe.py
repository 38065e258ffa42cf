"""What MCUs cut by the bottom edge cost the layouts' kernels: 1920x1080 (67.5 MCU rows of 16 pixels) against 1920x1088."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compeg_amd as ca
from tools import synth
gpu = ca.Gpu.open(0)
for samp in ((2, 2), (1, 2), (2, 1), (1, 1)):
    for n in (1, 64, 256):
        row = []
        for (w, h) in ((1920, 1080), (1920, 1088), (1912, 1088)):
            frames = [synth.make_jpeg(w, h, seed=70 + i, quality=85, ri=4, sampling=samp) for i in range(min(n, 8))]
            imgs = [ca.ImageData(j, allow_sampling=True) for j in frames]
            b = ca.Batch(gpu); b.upload([imgs[i % len(imgs)] for i in range(n)])
            for _ in range(3): b.decode(); b.wait()
            b.timing(reset=True); ts = []
            for _ in range(8):
                b.decode(); b.wait(); ts.append(b.timing(reset=True)[1] * 1000)
            row.append(f"{w}x{h} {b.last_kernel()} {np.median(ts):.1f} us")
        print(f"{samp[0]}x{samp[1]} x{n}: " + "; ".join(row), flush=True)
