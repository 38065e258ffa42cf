"""Diagnostic (library built with -DCG_STAMPS): cycles per AC-loop iteration and the share spent at
the LDS wait, for the fused kernel and for the two-kernel pipeline at several occupancies."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compeg_amd as ca
from compeg_amd._lib import lib
from tools import synth

lib.compeg_debug_ac_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
gpu = ca.Gpu.open(0)
imgs = [ca.ImageData(synth.make_jpeg(3840, 2160, seed=100 + i)) for i in range(4)]
n = int(os.environ.get("FRAMES", "64"))
batch = ca.Batch(gpu)
batch.upload([imgs[i % 4] for i in range(n)])
out = (C.c_ulonglong * 4)()
lib.compeg_debug_ac_stamps(out, 1)
for _ in range(2):
    batch.decode()
batch.wait()
lib.compeg_debug_ac_stamps(out, 1)
cyc, wait, its, calls = [int(v) for v in out]
print("pipeline=%s wpb=%s pad=%s: %.0f cycles/iteration, %.0f of them at the LDS wait, %.1f iterations/DU, loop %.0f cycles/DU" % (
    os.environ.get("COMPEG_PIPELINE", "fused"), os.environ.get("COMPEG_WPB", "-"), os.environ.get("COMPEG_LDS_PAD", "-"),
    cyc / max(its, 1), wait / max(its, 1), its / max(calls, 1), cyc / max(calls, 1)))
