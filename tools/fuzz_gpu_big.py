"""Randomized check on the GPU with frames of 720p to 4K+ (tools/fuzz_gpu.py stays below 2600 x 1300): random
quality, restart interval 1..8, content kind, bit flips in every other scan, both entropy modes -- single blocking
decodes with the scan preprocessed on 8 host threads and by the device kernels -- against the oracle.
tests/test_gpu_parity.py runs a short seed of it.  Found in round 2: the threaded staging copy of the decoder's
device path lost the last bytes of a segment for some lengths (ScanBuffer::copy).
    python tools/fuzz_gpu_big.py [seed] [iterations]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compeg_amd as ca
import oracle.oracle as orc
from tools import synth

SIZES = [(3840, 2160), (1920, 1080), (2560, 1440), (1280, 720), (3840, 2160), (4096, 2304)]


def run(seed=20261004, iters=36, log=print, dris=(1, 2, 4, 4, 4, 8)):
    """Returns (decodes compared, mismatches).  dris: the restart intervals drawn from."""
    rng = np.random.default_rng(seed)
    gpu = ca.Gpu.open()
    decs = []
    for device, threads in ((False, 8), (True, 4)):
        d = ca.Decoder(gpu)
        d.set_device_preprocess(device)
        d.set_scan_threads(threads)
        decs.append((device, d))
    bad = n = 0
    t0 = time.time()
    for it in range(iters):
        w, h = SIZES[it % len(SIZES)]
        q = int(rng.choice([50, 85, 95]))
        ri = int(rng.choice(list(dris)))
        kind = int(rng.integers(0, 3))
        j = bytearray(synth.make_jpeg(w, h, seed=int(rng.integers(1, 1 << 30)), kind=kind, quality=q, ri=ri))
        if it % 2:
            scan_at = j.find(b"\xff\xda") + 14
            for _ in range(int(rng.integers(1, 60))):
                pos = int(rng.integers(scan_at, len(j) - 2))
                if j[pos] != 0xFF and j[pos - 1] != 0xFF:     # keep the marker structure intact
                    j[pos] ^= 1 << int(rng.integers(0, 8))
                    if j[pos] == 0xFF:
                        j[pos] = 0xFE
        j = bytes(j)
        std = bool((it // 2) % 2)
        try:
            want = orc.ImageData(j, standard_entropy=std).decode()
        except orc.OracleError:
            continue
        img = ca.ImageData(j, standard_entropy=std)
        # a reused decoder keeps the texels no MCU covers (a truncated last restart interval) from earlier
        # images, like the reference's reused texture; the oracle starts from zeros: leave those MCUs out
        wm, hm = (w + 15) // 16, (h + 7) // 8
        mask = np.ones((h, w), dtype=bool)
        for m in range((wm * hm // ri) * ri, wm * hm):
            mask[(m // wm) * 8:(m // wm + 1) * 8, (m % wm) * 16:(m % wm + 1) * 16] = False
        for device, d in decs:
            d.decode_blocking(img)
            got = d.read_texture(w, h)
            n += 1
            if not np.array_equal(got[mask], want[mask]):
                bad += 1
                log("MISMATCH", it, w, h, q, ri, kind, std, "device preprocess" if device else "host preprocess",
                    int(((got != want).any(axis=2) & mask).sum()), "pixels")
        if it % 6 == 5:
            log("iteration", it, "decodes", n, "bad", bad, "%.0f s" % (time.time() - t0))
    return n, bad


if __name__ == "__main__":
    n, bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 20261004, int(sys.argv[2]) if len(sys.argv) > 2 else 36)
    print("big-frame fuzz: decodes", n, "mismatches", bad)
    sys.exit(1 if bad else 0)
