"""Randomized check on the GPU (tests/test_gpu_parity.py runs a bounded seed of it; longer runs by hand): random geometry, quality, restart interval,
sampling, bit flips in the scan, both entropy modes -- single decodes through every way of getting the scan to the
card (host preprocessor on 1 / 4 threads, device scan kernels; blocking and non-blocking), and batches (host and
device preprocessing) -- against the oracle.
    python tools/fuzz_gpu.py [seed] [iterations]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compeg_amd as ca
import oracle.oracle as orc
from tools import synth


def run(seed=4321, iters=200, budget_s=None, log=print):
    """Returns (decodes compared, mismatches, inputs the oracle rejects); stops early after budget_s seconds."""
    import time
    t_end = time.monotonic() + budget_s if budget_s else None
    rng = np.random.default_rng(seed)
    gpu = ca.Gpu.open()
    decs = []
    for device, threads in ((False, 1), (False, 4), (True, 4)):
        d = ca.Decoder(gpu)
        d.set_device_preprocess(device)
        d.set_scan_threads(threads)
        decs.append((f"device={device} threads={threads}", d))
    bad = runs = skipped = 0
    pool = []
    for it in range(iters):
        if t_end and time.monotonic() > t_end:
            iters = it
            break
        big = it % 10 == 0                      # now and then a scan large enough for the threaded host path
        w = int(rng.integers(600, 2600)) if big else int(rng.integers(16, 420))
        h = int(rng.integers(300, 1300)) if big else int(rng.integers(8, 200))
        if os.environ.get('FUZZ_NARROW'):   # (one to three MCUs across, many MCU rows)
            w = int(rng.integers(1, 50)); h = int(rng.integers(8, 900))
        kind = int(rng.integers(0, 3))
        q = int(rng.choice([30, 60, 85, 95, 100]))
        ri = int(rng.integers(0, 9))
        if it % 3 == 2:                         # every restart interval is somebody's: longer ones, odd ones
            ri = int(rng.choice([10, 12, 16, 17, 30, 40, 41, 60, 64, 65, 120, 240]))
        ext = it % 4 == 3
        sampling = [(2, 1), (1, 1), (1, 2), (2, 2)][int(rng.integers(0, 4))] if ext else (2, 1)
        std = bool(it % 2)
        j = bytearray(synth.make_jpeg(w, h, seed=int(rng.integers(1, 1 << 30)), kind=kind, quality=q, ri=ri,
                                      sampling=sampling))
        if it % 3 != 0:
            scan_at = j.find(b"\xff\xda") + 14
            for _ in range(int(rng.integers(1, 40))):
                pos = int(rng.integers(scan_at, len(j) - 2))
                if j[pos] != 0xFF and j[pos - 1] != 0xFF:
                    j[pos] ^= 1 << int(rng.integers(0, 8))
                    if j[pos] == 0xFF:
                        j[pos] = 0xFE
        j = bytes(j)
        try:
            want = orc.ImageData(j, allow_sampling=ext, standard_entropy=std).decode()
        except orc.OracleError:
            skipped += 1
            continue
        img = ca.ImageData(j, allow_sampling=ext, standard_entropy=std)
        # a reused decoder keeps the texels no MCU covers (a truncated last restart interval) from earlier
        # images, like the reference's reused texture; the oracle starts from zeros: leave those MCUs out
        mw, mh = 8 * sampling[0], 8 * sampling[1]
        wm, hm = (w + mw - 1) // mw, (h + mh - 1) // mh
        covered = (wm * hm // ri) * ri if ri else wm * hm
        mask = np.ones((h, w), dtype=bool)
        for m in range(covered, wm * hm):
            mask[(m // wm) * mh:(m // wm + 1) * mh, (m % wm) * mw:(m % wm + 1) * mw] = False
        for name, d in decs:
            for blocking in (True, False):
                if blocking:
                    d.decode_blocking(img)
                else:
                    d.start_decode(img).wait()
                got = d.read_texture(img.width(), img.height())
                runs += 1
                if not np.array_equal(got[mask], want[mask]):
                    bad += 1
                    log("MISMATCH", it, w, h, kind, q, ri, sampling, std, name, "blocking" if blocking else "async", flush=True)
                    open("/tmp/bad_gpu_%d.jpg" % it, "wb").write(j)
        pool.append((img, want))
        if len(pool) == 24 or it == iters - 1:
            for mode in (0, 1):
                b = ca.Batch(gpu)
                if all(im.width() for im, _ in pool):
                    try:
                        b.set_device_preprocess(mode)
                        b.upload([im for im, _ in pool])
                    except ca.Error as e:       # device preprocessing of batches is 4:2:2 only
                        if mode == 1:
                            continue
                        raise
                    b.decode()
                    b.wait()
                    for i, (_, wnt) in enumerate(pool):
                        runs += 1
                        if not np.array_equal(b.read_output(i), wnt):
                            bad += 1
                            log("MISMATCH in batch, preprocess mode", mode, "entry", i, flush=True)
            pool = []
        if it % 20 == 0:
            log("iteration", it, "runs", runs, "bad", bad, flush=True)
    if pool:  # (a run cut short by its budget: the images still waiting for their batch)
        b = ca.Batch(gpu)
        b.upload([im for im, _ in pool])
        b.decode()
        b.wait()
        for i, (_, wnt) in enumerate(pool):
            runs += 1
            if not np.array_equal(b.read_output(i), wnt):
                bad += 1
                log("MISMATCH in the final batch, entry", i, flush=True)
    log("runs", runs, "bad", bad, "skipped (oracle rejects)", skipped)
    return runs, bad, skipped


def main():
    runs, bad, _ = run(int(sys.argv[1]) if len(sys.argv) > 1 else 4321, int(sys.argv[2]) if len(sys.argv) > 2 else 200)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
