"""The walk + lane-per-MCU route against the other kernels, batch by batch: bit-exact against the oracle (sampled
images), which kernel ran, kernel time by HIP events (median).  Run once per COMPEG_WALK setting (laboratory library):
    COMPEG_LIB=compeg_amd/libcompeg_hip_lab.so COMPEG_WALK=1 python tools/walk_probe.py [configs]
configs: WxH:ri:n[:q[:kind]] ...  (default: a sweep)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compeg_amd as ca
from oracle import oracle as orc
from tools import synth

DEFAULT = ["960x720:60:256", "960x720:10:256", "960x720:60:1024", "960x720:4:256", "960x720:30:64", "3840x2160:4:4", "3840x2160:4:8",
           "3840x2160:4:16", "3840x2160:4:32", "1920x1080:4:64", "1000x990:7:40", "3840x2160:240:16"]
cfgs = sys.argv[1:] or DEFAULT
gpu = ca.Gpu.open(0)
distinct = int(os.environ.get("PROBE_DISTINCT", "8"))
check = int(os.environ.get("PROBE_CHECK", "3"))
for cfg in cfgs:
    parts = cfg.split(":")
    w, h = (int(v) for v in parts[0].split("x"))
    ri, n = int(parts[1]), int(parts[2])
    q = int(parts[3]) if len(parts) > 3 else 85
    kind = int(parts[4]) if len(parts) > 4 else 0
    frames = [synth.make_jpeg(w, h, seed=4000 + i + ri, kind=kind, quality=q, ri=ri) for i in range(min(distinct, n))]
    images = [ca.ImageData(f) for f in frames]
    b = ca.Batch(gpu)
    b.upload([images[i % len(images)] for i in range(n)])
    for _ in range(3):
        b.decode(); b.wait()
    b.timing(reset=True)
    ts, firsts = [], []
    for _ in range(int(os.environ.get("PROBE_REPS", "12"))):
        b.decode(); b.wait()
        _, total, first, _ = b.timing(reset=True)
        ts.append(total * 1000.0)
        firsts.append(first * 1000.0)
    bad = 0
    for i in sorted({0, n - 1, n // 2, min(n - 1, len(frames) - 1)})[:check + 1]:
        want = orc.ImageData(frames[i % len(frames)]).decode()
        got = b.read_output(i)
        if not np.array_equal(got, want):
            bad += 1
            diff = (got != want).any(axis=2)
            ys, xs = np.nonzero(diff)
            print(f"   image {i}: {int(diff.sum())} pixels differ, first at x={xs[0]} y={ys[0]}", flush=True)
    mpix = n * w * h / 1e6
    frac = b.algorithmic_bytes() / (np.median(ts) * 1e-6) / 8e12
    print(f"{cfg:24s} kernel {b.last_kernel():12s} {'OK ' if not bad else 'BAD'} median {np.median(ts):9.1f} us min {min(ts):9.1f} us "
          f"(first kernel {np.median(firsts):8.1f}) {mpix / np.median(ts) * 1e3:8.1f} Gpx/s frac {frac:.3f}", flush=True)
