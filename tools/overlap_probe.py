"""Two batches in flight on two streams against one batch on one: does the walk of one decode run under the second kernel
of another?  (laboratory question behind the pipelined form of the walk + lane-per-MCU route)
    python tools/overlap_probe.py [WxH:ri:n[:q]] ..."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compeg_amd as ca
from tools import synth

cfgs = sys.argv[1:] or ["960x720:60:256", "960x720:60:1024", "960x720:10:256", "3840x2160:240:16"]
gpu = ca.Gpu.open(0)
reps = int(os.environ.get("PROBE_REPS", "20"))
for cfg in cfgs:
    parts = cfg.split(":")
    w, h = (int(v) for v in parts[0].split("x"))
    ri, n = int(parts[1]), int(parts[2])
    q = int(parts[3]) if len(parts) > 3 else 85
    frames = [synth.make_jpeg(w, h, seed=4000 + i + ri, quality=q, ri=ri) for i in range(min(8, n))]
    images = [ca.ImageData(f) for f in frames]
    batches = [ca.Batch(gpu) for _ in range(2)]
    for b in batches:
        b.upload([images[i % len(images)] for i in range(n)])
        b.set_timing(False)
    streams = [torch.cuda.Stream() for _ in range(2)]
    res = {}
    for name, plan in (("one batch, one stream", [(0, 0)] * 2), ("two batches, one stream", [(0, 0), (1, 0)]),
                       ("two batches, two streams", [(0, 0), (1, 1)])):
        for _ in range(2):
            for bi, si in plan:
                batches[bi].decode(streams[si].cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            for bi, si in plan:
                batches[bi].decode(streams[si].cuda_stream)
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / (reps * len(plan)) * 1e6
    mpix = n * w * h / 1e6
    print(f"{cfg:22s} kernel {batches[0].last_kernel():10s} us per decode: " +
          ", ".join(f"{k} {v:8.1f} ({mpix / v * 1e3:6.1f} Gpx/s)" for k, v in res.items()), flush=True)
