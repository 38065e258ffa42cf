"""Development aid: decodes a few images many times through the Decoder (cooperative kernel where it applies) and
reports every run that differs from the oracle -- to tell deterministic differences from intermittent ones.
    python tools/coop_stress.py [repeats]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compeg_amd as ca
import oracle.oracle as orc
from tools import synth


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    gpu = ca.Gpu.open()
    cases = [(1920, 1080, 77, 0, 85, 4, True), (1920, 1080, 77, 0, 85, 4, False), (1920, 1080, 77, 0, 85, 8, True),
             (3840, 2160, 9, 0, 85, 4, False), (640, 360, 8, 1, 95, 4, False), (1920, 1080, 5, 0, 95, 2, True)]
    total_bad = 0
    for (w, h, seed, kind, q, ri, std) in cases:
        j = synth.make_jpeg(w, h, seed=seed, kind=kind, quality=q, ri=ri)
        want = orc.ImageData(j, standard_entropy=std).decode()
        img = ca.ImageData(j, standard_entropy=std)
        seen = {}
        for r in range(reps):
            dec = ca.Decoder(gpu) if r % 2 == 0 else dec
            dec.decode_blocking(img)
            got = dec.read_texture(w, h)
            if not np.array_equal(got, want):
                diff = (got != want).any(axis=2)
                ys, xs = np.nonzero(diff)
                key = (int(diff.sum()), int(xs[0]), int(ys[0]))
                seen[key] = seen.get(key, 0) + 1
        total_bad += sum(seen.values())
        print((w, h, seed, kind, q, ri, std), "bad runs:", sum(seen.values()), "of", reps, seen, flush=True)
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main())
