"""Diagnostic: per-phase cycle shares inside decode_fused_422_kernel.  Needs the library built
with CXXFLAGS+=-DCG_STAMPS (tools/run_stamps.sh); never quote this build's run time."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["COMPEG_STAMPS"] = "1"
import compeg_amd as ca
from compeg_amd._lib import lib
from tools import synth

lib.compeg_debug_read_dc.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
lib.compeg_debug_read_batch_dc.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
gpu = ca.Gpu.open(0)
jpeg = synth.make_jpeg(3840, 2160, seed=0xC0FFEE)
img = ca.ImageData(jpeg)
dec = ca.Decoder(gpu)
for _ in range(3):
    dec.decode_blocking(img)
waves = (img.parallelism() + 63) // 64
buf = np.zeros((waves, 4), dtype=np.uint64)
assert lib.compeg_debug_read_dc(dec._h, buf.ctypes.data, buf.nbytes) == 0
tot = buf.sum(axis=1)
print("single frame: waves", waves, "cycles/wave mean %.0f max %.0f" % (tot.mean(), tot.max()))
for name, col in zip(("dc", "ac", "idct+slot", "composite"), buf.T):
    print("  %-10s %8.0f cycles/wave  %5.1f %%" % (name, col.mean(), 100 * col.sum() / tot.sum()))
n = 64
batch = ca.Batch(gpu)
batch.upload([img] * n)
for _ in range(2):
    batch.decode()
batch.wait()
dus = 16200 * 16
b = np.zeros(n * dus * 4 // 8, dtype=np.uint64)
assert lib.compeg_debug_read_batch_dc(batch._h, b.ctypes.data, b.nbytes) == 0
b = b.reshape(n, dus * 4 // 8)[:, : waves * 4].reshape(n * waves, 4)
tot = b.sum(axis=1)
print("batch of %d: cycles/wave mean %.0f max %.0f" % (n, tot.mean(), tot.max()))
for name, col in zip(("dc", "ac", "idct+slot", "composite"), b.T):
    print("  %-10s %8.0f cycles/wave  %5.1f %%" % (name, col.mean(), 100 * col.sum() / tot.sum()))
# imbalance inside workgroups: a block's LDS is held until its slowest wave is done
wpb = int(os.environ.get("COMPEG_WPB", "12"))
per_img = tot.reshape(n, waves)
loss = []
for img_t in per_img:
    nb = (waves + wpb - 1) // wpb
    pad = np.concatenate([img_t, np.zeros(nb * wpb - waves)])
    blk = pad.reshape(nb, wpb)
    loss.append(blk.max(axis=1).sum() * wpb / max(img_t.sum(), 1))
print("block time / mean wave time (12-wave blocks): %.3f" % np.mean(loss))
