"""Diagnostic: per-phase cycles inside decode_coop_422_kernel.  Needs a library built with -DCG_COOP_STAMPS
(tools/build_variant.sh coopstamps "-DCG_COOP_STAMPS", COMPEG_LIB=gpurun_ab/lib_coopstamps.so); never quote
this build's run time."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compeg_amd as ca
from compeg_amd._lib import lib
from tools import synth

lib.compeg_debug_read_dc.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
gpu = ca.Gpu.open(0)
w, h, ri = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (3840, 2160, 4)))
jpeg = synth.make_jpeg(w, h, seed=0xC0FFEE, ri=ri, quality=int(os.environ.get('PROBE_QUALITY', '85')))
img = ca.ImageData(jpeg)
dec = ca.Decoder(gpu)
for _ in range(3):
    dec.decode_blocking(img)
team = os.environ.get("COMPEG_COOP_TEAM", "1") != "0"
if team:
    per_team = max(1, 256 // (4 * ri))          # (coop_shape: a team of four waves takes 4 x 64 data units' worth of intervals)
    waves = (img.parallelism() + per_team - 1) // per_team * 4
else:
    ipw = max(1, 64 // (4 * ri))
    waves = (img.parallelism() + ipw - 1) // ipw
full = np.zeros((waves, 16), dtype=np.uint64)
assert lib.compeg_debug_read_dc(dec._h, full.ctypes.data, full.nbytes) == 0
buf = full[:, :8]
wall = full[:, 8:10].astype(np.int64)
ok = wall[:, 0] > 0
t0 = wall[ok, 0].min()
print("device clock (10 ns ticks): first wave starts at 0; starts p50 %d max %d; ends p50 %d p99 %d max %d" % (
    np.median(wall[ok, 0] - t0), (wall[ok, 0] - t0).max(), np.median(wall[ok, 1] - t0), np.percentile(wall[ok, 1] - t0, 99),
    (wall[ok, 1] - t0).max()))
if team:
    walker = buf[:, 1] > 0  # (the waves that spent time in the walk loop)
    for name, sel in (("walkers", walker), ("others", ~walker)):
        sub = buf[sel & ok]
        print(name, "cycles/wave by phase:", " ".join("%s %.0f" % (n, c) for n, c in zip(
            ("setup", "chase", "validate", "decode", "fixups", "dc+idct", "composite", "wait"), sub.mean(axis=0))))
    # the teams that end last: their walker's phases
    end = wall[:, 1] - t0
    late = np.argsort(end)[-64:]
    late_walkers = [i for i in late if walker[i]]
    if late_walkers:
        sub = buf[late_walkers]
        print("the %d walkers among the 64 waves that end last (ends %d..%d): %s" % (len(late_walkers), end[late].min(), end[late].max(),
              " ".join("%s %.0f" % (n, c) for n, c in zip(("setup", "chase", "validate", "decode", "fixups", "dc+idct", "composite", "wait"), sub.mean(axis=0)))))
    ex = full[walker & ok, 10:15].astype(np.float64)
    print("staging (cycles from the wave's start): window known %.0f, copies done %.0f, behind the barrier %.0f" % (
        ex[:, 0].mean(), ex[:, 1].mean(), ex[:, 2].mean()))
    if ex[:, 1].sum() > 0:
        print("walk loop alone: cycles/wave %.0f, steps/wave %.1f, cycles/step %.0f, times entered %.1f; before it %.0f, entries to state words %.0f" % (
            ex[:, 0].mean(), ex[:, 1].mean(), ex[:, 0].sum() / ex[:, 1].sum(), ex[:, 2].mean(), ex[:, 3].mean(), ex[:, 4].mean()))
tot = buf.sum(axis=1)
print("waves", waves, "cycles/wave mean %.0f max %.0f p99 %.0f" % (tot.mean(), tot.max(), np.percentile(tot, 99)))
names = ("setup", "chase", "validate", "decode", "fixups", "dc+idct", "composite", "wait")
for name, col in zip(names, buf.T):
    print("  %-10s %8.0f cycles/wave (max %8.0f)  %5.1f %%" % (name, col.mean(), col.max(), 100 * col.sum() / max(tot.sum(), 1)))
# per-wave walk steps from the emulation (tools/probes/wave_steps_<shift>.txt, if present): cycles per step
shift = os.environ.get("COMPEG_COOP_SPEC_SHIFT", "0")
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "wave_steps_%s.txt" % shift)
if os.path.exists(path) and (w, h, ri) == (3840, 2160, 4) and not team:
    steps = np.zeros(waves)
    for line in open(path):
        wv, rnd, n = (int(v) for v in line.split())
        steps[wv] += n
    per = buf[:, 1] / np.maximum(steps, 1)
    print("walk: steps/wave mean %.1f; cycles per step mean %.0f p10 %.0f p90 %.0f" % (steps.mean(), per.mean(), np.percentile(per, 10), np.percentile(per, 90)))
    if os.environ.get("COOP_STAMPS_DUMP"):
        print("cycles per step of the first 64 waves:", " ".join("%d" % v for v in per[:64]))
        print("setup cycles of the first 64 waves:  ", " ".join("%d" % v for v in buf[:64, 0]))
        # by position inside the workgroup (8 waves each)
        for k in range(8):
            print("wave %d of its workgroup: cycles per step mean %.0f" % (k, per[k::8].mean()))
