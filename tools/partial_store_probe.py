"""One configuration of tools/pair_align_probe.py, for the counters: python tools/partial_store_probe.py W H DRI [hs vs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compeg_amd as ca
from tools import synth
w, h, ri = (int(x) for x in sys.argv[1:4])
samp = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (1, 1)
gpu = ca.Gpu.open(0)
frames = [synth.make_jpeg(w, h, seed=70 + i, quality=85, ri=ri, sampling=samp) for i in range(8)]
imgs = [ca.ImageData(j, allow_sampling=True) for j in frames]
b = ca.Batch(gpu); b.upload([imgs[i % 8] for i in range(256)])
for _ in range(4):
    b.decode(); b.wait()
print(w, h, ri, samp, b.last_kernel(), 256 * w * h * 4 / 1e9, "GB of pixels a launch")
