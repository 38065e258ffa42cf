"""How often does a single frame with restart intervals of 41..128 MCUs hit the cooperative kernel's slow path (a
speculative walk that is given up and repeated by the serial decoder)?  Kernel time by events for N seeds.
    COMPEG_LIB=compeg_amd/libcompeg_hip_lab.so COMPEG_WALK=0 python tools/coop_cliff_probe.py [WxH:ri[:q]] [seeds]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compeg_amd as ca
from tools import synth

cfg = (sys.argv[1] if len(sys.argv) > 1 else "1920x1080:120").split(":")
w, h = (int(v) for v in cfg[0].split("x"))
ri = int(cfg[1])
q = int(cfg[2]) if len(cfg) > 2 else 85
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 24
gpu = ca.Gpu.open(0)
out = []
for sd in range(seeds):
    j = synth.make_jpeg(w, h, seed=4120 + sd, kind=0, quality=q, ri=ri)
    b = ca.Batch(gpu)
    if os.environ.get("PROBE_DEVICE"):
        b.set_device_preprocess(1)   # (the scan kernels preprocess the raw segment: windows planned from estimates)
    b.upload([ca.ImageData(j)])
    for _ in range(2):
        b.decode(); b.wait()
    b.timing(reset=True)
    ts = []
    for _ in range(5):
        b.decode(); b.wait()
        ts.append(b.timing(reset=True)[1] * 1000.0)
    out.append((4120 + sd, b.last_kernel(), float(np.median(ts))))
ts = [t for _, _, t in out]
print(f"{cfg[0]} DRI={ri} q{q}{' device-preprocessed' if os.environ.get('PROBE_DEVICE') else ''}: kernel {out[0][1]}, us per frame over {seeds} seeds: median {np.median(ts):.0f}, min {min(ts):.0f}, max {max(ts):.0f}; "
      f"above 1.5 x the median: {[(s, round(t)) for s, _, t in out if t > 1.5 * np.median(ts)]}")
