"""20 blocking decodes of one 4K frame through a Decoder, then 20 back-to-back decodes of a one-image Batch: what
tools/prof_single_frame.sh traces (every HIP call, kernel and copy of a single-frame decode)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compeg_amd as ca
from tools import synth

w, h, ri = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "3840x2160x4").split("x"))
jpeg = synth.make_jpeg(w, h, seed=0xC0FFEE, kind=0, quality=85, ri=ri)
gpu = ca.Gpu.open(0)
img = ca.ImageData(jpeg, copy=False)
dec = ca.Decoder(gpu)
for _ in range(30):
    dec.decode_blocking(img)
time.sleep(0.002)
t = time.perf_counter()
for _ in range(20):
    dec.decode_blocking(img)
print("blocking decode: %.1f us each; stages %s" % ((time.perf_counter() - t) / 20 * 1e6, dec.last_stage_times()))
b = ca.Batch(gpu)
b.upload([img])
for _ in range(30):
    b.decode()
b.wait()
time.sleep(0.002)
t = time.perf_counter()
for _ in range(20):
    b.decode()
b.wait()
print("one-image batch, back to back: %.1f us each" % ((time.perf_counter() - t) / 20 * 1e6))
if hasattr(b, "set_timing"):
    b.set_timing(False)
    for _ in range(30):
        b.decode()
    b.wait()
    time.sleep(0.002)
    t = time.perf_counter()
    for _ in range(20):
        b.decode()
    b.wait()
    print("one-image batch, back to back, no timing events: %.1f us each" % ((time.perf_counter() - t) / 20 * 1e6))
