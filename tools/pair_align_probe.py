"""Pairs of 8-pixel MCUs (even restart intervals: 64-byte rows) against single MCUs (odd ones: 32-byte rows) where an MCU row
holds an odd number of MCUs -- every second MCU row's pairs then lie across two 64-byte segments."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compeg_amd as ca
from tools import synth
gpu = ca.Gpu.open(0)
for samp in ((1, 1), (1, 2)):
    for (w, h) in ((1920, 1088), (1912, 1088), (1080, 1920)):
        row = []
        for ri in (4, 5, 8, 9):
            frames = [synth.make_jpeg(w, h, seed=70 + i, quality=85, ri=ri, sampling=samp) for i in range(8)]
            imgs = [ca.ImageData(j, allow_sampling=True) for j in frames]
            n = 256
            b = ca.Batch(gpu); b.upload([imgs[i % len(imgs)] for i in range(n)])
            for _ in range(3): b.decode(); b.wait()
            b.timing(reset=True); ts = []
            for _ in range(8):
                b.decode(); b.wait(); ts.append(b.timing(reset=True)[1] * 1000)
            row.append(f"DRI={ri} {b.last_kernel()} {np.median(ts):.1f} us")
        print(f"{samp[0]}x{samp[1]} {w}x{h} x256: " + "; ".join(row), flush=True)
