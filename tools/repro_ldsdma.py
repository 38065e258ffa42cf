"""Replays the kind of batch round 3's GPU fuzz found wrong with LDS-DMA staged rows (profiles/r03/NOTES.md: seed 9902,
batch 235 -- 1463 slots of four distinct 640x360 frames, DRI = 7, 5 to 8 bit per pixel, on the streamed-window kernel):
every output of a few decodes against the oracle, for the library in COMPEG_LIB (tools/repro_ldsdma.sh runs it for the
laboratory build with the LDS-DMA arm and for one without)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compeg_amd as ca
import oracle.oracle as orc
from tools import synth

gpu = ca.Gpu.open()
total_bad = total = 0
for (q, kind, ri) in ((85, 1, 7), (70, 1, 7), (95, 1, 7), (85, 1, 10)):
    frames = [synth.make_jpeg(640, 360, seed=9902 + 17 * i + q, kind=kind, quality=q, ri=ri) for i in range(4)]
    wants = [orc.ImageData(f).decode() for f in frames]
    bpp = 8 * sum(len(f) for f in frames) / 4 / (640 * 360)
    images = [ca.ImageData(f) for f in frames]
    b = ca.Batch(gpu)
    b.upload([images[i % 4] for i in range(1463)])
    bad = n = 0
    for rep in range(3):
        b.decode()
        b.wait()
        for i in range(1463):
            n += 1
            got = b.read_output(i)
            if not np.array_equal(got, wants[i % 4]):
                bad += 1
                diff = (got != wants[i % 4]).any(axis=2)
                ys, xs = np.nonzero(diff)
                # (an interval is ri MCUs of 16 x 8 pixels, 64 of them a wave's unit: which intervals differ)
                mcus = sorted({(int(y) // 8) * 40 + int(x) // 16 for y, x in zip(ys[::7], xs[::7])})
                ivs = sorted({m // ri for m in mcus})
                print(f"   wrong: decode {rep} slot {i} (frame {i % 4}): {int(diff.sum())} pixels, rows {ys.min()}..{ys.max()}, columns {xs.min()}..{xs.max()}; "
                      f"intervals {ivs[0]}..{ivs[-1]} ({len(ivs)} of them, units {sorted({v // 64 for v in ivs})}), first MCU inside its interval {mcus[0] % ri}", flush=True)
    print(f"q{q} kind {kind} DRI {ri}: {bpp:.1f} bit per pixel, kernel {b.last_kernel()}, {n} outputs compared, {bad} wrong", flush=True)
    total_bad += bad
    total += n
print(f"repro_ldsdma: {total} outputs, {total_bad} wrong ({os.environ.get('COMPEG_LIB', 'shipped library')})")
