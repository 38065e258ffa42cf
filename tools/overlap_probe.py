"""Experiment: do the two kernels of the two-kernel pipeline overlap when two batches run on two
streams (entropy of one batch beside IDCT+composite of the other)?  Compares one 128-frame batch on
one stream with two 64-frame batches on two streams, for the pipeline chosen by COMPEG_PIPELINE."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compeg_amd as ca
from tools import synth

torch.cuda.set_device(0)
gpu = ca.Gpu.open(0)
jpegs = [synth.make_jpeg(3840, 2160, seed=500 + i) for i in range(16)]
images = [ca.ImageData(j) for j in jpegs]
n = 128


def run(batches, streams, iters=6):
    for b, s in zip(batches, streams):
        b.decode(s.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        for b, s in zip(batches, streams):
            b.decode(s.cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters / n * 1e6


one = ca.Batch(gpu)
one.upload([images[i % 16] for i in range(n)])
s0 = torch.cuda.Stream()
print("pipeline=%s wpb=%s: one stream, %d frames: %.2f us/frame" % (
    os.environ.get("COMPEG_PIPELINE", "fused"), os.environ.get("COMPEG_WPB", "-"), n, run([one], [s0])))
del one
halves = [ca.Batch(gpu), ca.Batch(gpu)]
for h in halves:
    h.upload([images[i % 16] for i in range(n // 2)])
ss = [torch.cuda.Stream(), torch.cuda.Stream()]
print("   two streams, 2 x %d frames: %.2f us/frame" % (n // 2, run(halves, ss)))
quarters = [ca.Batch(gpu) for _ in range(4)]
for q in quarters:
    q.upload([images[i % 16] for i in range(n // 4)])
s4 = [torch.cuda.Stream() for _ in range(4)]
print("   four streams, 4 x %d frames: %.2f us/frame" % (n // 4, run(quarters, s4)))
