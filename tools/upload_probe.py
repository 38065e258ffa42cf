"""PCIe/host-inclusive rate of a batch: upload (host or device preprocessing) + decode of 256 4K frames."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compeg_amd
from tools import synth
from concurrent.futures import ThreadPoolExecutor

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
with ThreadPoolExecutor(16) as ex:
    jpegs = list(ex.map(lambda i: synth.make_jpeg(3840, 2160, seed=0xC0FFEE + i, quality=85, ri=4), range(min(n, 64))))
images = [compeg_amd.ImageData(jpegs[i % len(jpegs)]) for i in range(n)]
gpu = compeg_amd.Gpu.open()
px = n * 3840 * 2160
for mode, name in ((0, "host preprocessing, 16 threads"), (1, "device preprocessing")):
    b = compeg_amd.Batch(gpu)
    b.set_device_preprocess(mode)
    ts = []
    for it in range(4):
        t0 = time.perf_counter()
        b.upload(images, host_threads=16)
        t1 = time.perf_counter()
        b.decode()
        b.wait()
        t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1))
    up, dec = min(ts[1:])
    print(f"{name}: upload {up*1e3:.1f} ms + decode {dec*1e3:.2f} ms for {n} frames = {px/(up+dec)/1e9:.1f} Gpixel/s "
          f"({sum(len(j) for j in jpegs)/len(jpegs)*n/1e6:.0f} MB of JPEG)", flush=True)
