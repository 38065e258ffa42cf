"""One frame per restart interval through the cooperative kernel: bit-exact against the oracle, which kernel ran,
kernel time by HIP events (median of a few decodes of a one-image batch).
    python tools/coop_dri_probe.py [WxH] [dri,dri,...] [quality]"""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import compeg_amd as ca
from oracle import oracle as orc
from tools import synth

wh = sys.argv[1] if len(sys.argv) > 1 else "960x720"
w, h = (int(v) for v in wh.split("x"))
dris = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1,2,3,4,5,7,8,10,16,30,60,64,65,120,240").split(",")]
q = int(sys.argv[3]) if len(sys.argv) > 3 else 85
gpu = ca.Gpu.open(0)
for ri in dris:
    jpeg = synth.make_jpeg(w, h, seed=100 + ri, kind=0, quality=q, ri=ri)
    want = orc.ImageData(jpeg).decode()
    dec = ca.Decoder(gpu)
    data = ca.ImageData(jpeg)
    dec.decode_blocking(data)
    got = dec.read_texture(w, h)
    ok = np.array_equal(got, want)
    b = ca.Batch(gpu)
    b.upload([data])
    for _ in range(3):
        b.decode(); b.wait()
    b.timing(reset=True)
    ts = []
    for _ in range(15):
        b.decode(); b.wait()
        n, total, _, _ = b.timing(reset=True)
        ts.append(total * 1000.0)
    okb = np.array_equal(b.read_output(0), want)
    try:
        kernels = f"{dec.last_kernel():9s}/{b.last_kernel():9s}"
    except AttributeError:   # (an older build of the library, COMPEG_LIB=...)
        kernels = "?"
    print(f"{wh} dri {ri:4d} intervals {data.parallelism():6d} kernel {kernels} "
          f"{'OK ' if ok and okb else 'BAD'} median {np.median(ts):7.1f} us  min {min(ts):7.1f} us", flush=True)
