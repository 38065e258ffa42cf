"""Diagnostic: wall-clock stamps of decode_team_422_kernel's workgroups on one 4K frame.  Needs the
library built with -DCG_STAMPS (tools/run_team_stamps.sh); never quote this build's run time."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["COMPEG_TEAM"] = "1"
import compeg_amd as ca
from compeg_amd._lib import lib
from tools import synth

lib.compeg_debug_read_dc.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
gpu = ca.Gpu.open(0)
jpeg = synth.make_jpeg(3840, 2160, seed=0xC0FFEE)
img = ca.ImageData(jpeg)
dec = ca.Decoder(gpu)
for _ in range(5):
    dec.decode_blocking(img)
groups = (img.parallelism() + 63) // 64
buf = np.zeros((groups, 8), dtype=np.uint64)
assert lib.compeg_debug_read_dc(dec._h, buf.ctypes.data, buf.nbytes) == 0
t = buf[:, :6].astype(np.float64) / 100.0    # us
t0 = t[:, 0].min()
print("workgroups", groups)
print("entry        : first %.1f last %.1f us" % (0.0, t[:, 0].max() - t0))
print("prologue     : mean %.1f max %.1f us" % ((t[:, 1] - t[:, 0]).mean(), (t[:, 1] - t[:, 0]).max()))
print("decoder      : mean %.1f max %.1f us after prologue; %.0f cycles mean -> %.2f GHz" % (
    (t[:, 2] - t[:, 1]).mean(), (t[:, 2] - t[:, 1]).max(), buf[:, 6].mean(),
    buf[:, 6].mean() / ((t[:, 2] - t[:, 1]).mean() * 1e3)))
for k in range(3):
    print("transformer %d: ends %.1f us (mean) after the decoder" % (k, (t[:, 3 + k] - t[:, 2]).mean()))
print("last end     : %.1f us after the first entry" % (t[:, 2:6].max() - t0))
