"""Randomized check of the batch paths on the GPU: batches of 1..6 images of 360p to 4K (same or mixed sizes, restart
intervals, qualities; bit flips in some scans), uploaded as parsed images and as JPEG bytes, scans preprocessed on the
host or by the device kernels (once / on every decode), decoded twice -- every output against the oracle.  Small
batches take the cooperative kernel where all images qualify, larger ones the paired / fused kernels; every fourth
batch of equal frames is repeated into a long uniform one (resident waves that walk over several units).
    python tools/fuzz_gpu_batch.py [seed] [batches]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compeg_amd as ca
import oracle.oracle as orc
from tools import synth

SIZES = [(640, 360), (1280, 720), (1920, 1080), (2560, 1440), (3840, 2160), (250, 70), (1000, 600)]
if os.environ.get('FUZZ_NARROW'):   # (frames one to five MCUs across, odd MCU counts a row, portrait shapes)
    SIZES = [(8, 600), (16, 900), (24, 333), (40, 1000), (72, 640), (360, 640), (1080, 1920), (1912, 1088), (1000, 600)]


def run(seed=77, batches=40, log=print):
    """Returns (outputs compared, mismatches)."""
    rng = np.random.default_rng(seed)
    gpu = ca.Gpu.open()
    bad = n = 0
    kernels = {}
    t0 = time.time()
    for it in range(batches):
        count = int(rng.integers(1, 7))
        same = bool(rng.integers(0, 2))
        w0, h0 = SIZES[int(rng.integers(0, len(SIZES)))]
        ri0 = int(rng.choice([1, 2, 4, 4, 8, 16, 3, 10, 30, 7, 120]))
        # one batch in four is of an extension layout (the fused layout kernels, whole windows or streamed), one in
        # eight mixes layouts (the two-kernel route)
        ext = int(rng.integers(0, 8))
        smp0 = [(1, 1), (1, 2), (2, 2)][ext % 3] if ext < 2 else (2, 1)
        items = []
        for i in range(count):
            w, h = (w0, h0) if same else SIZES[int(rng.integers(0, len(SIZES)))]
            ri = ri0 if same or rng.integers(0, 2) else int(rng.choice([1, 2, 4, 8]))
            smp = [(2, 1), (1, 1), (1, 2), (2, 2)][int(rng.integers(0, 4))] if ext == 2 else smp0
            j = bytearray(synth.make_jpeg(w, h, seed=int(rng.integers(1, 1 << 30)), kind=int(rng.integers(0, 3)),
                                          quality=int(rng.choice([50, 85, 95])), ri=ri, sampling=smp))
            if rng.integers(0, 4) == 0:
                scan_at = j.find(b"\xff\xda") + 14
                for _ in range(int(rng.integers(1, 30))):
                    pos = int(rng.integers(scan_at, len(j) - 2))
                    if j[pos] != 0xFF and j[pos - 1] != 0xFF:
                        j[pos] ^= 1 << int(rng.integers(0, 8))
                        if j[pos] == 0xFF:
                            j[pos] = 0xFE
            j = bytes(j)
            try:
                items.append((j, orc.ImageData(j, allow_sampling=True).decode()))
            except orc.OracleError:
                pass
        if not items:
            continue
        if same and it % 4 == 3:
            # a long uniform batch: the same few frames over and over until the launch has more units of 64 intervals
            # than the chip holds waves, by a random fraction of a round -- the throughput kernel's resident waves then
            # take a second, third ... unit each (quantisers and content differ per slot, sizes and tables do not)
            par = ca.ImageData(items[0][0], allow_sampling=True).parallelism()
            waves = (par + 63) // 64
            slots = int((3072 * float(rng.uniform(1.02, 2.6))) // waves) + 1
            if slots * w0 * h0 * 4 < (3 << 30):
                items = [items[int(rng.integers(0, len(items)))] for _ in range(slots)]
        mode = int(rng.integers(0, 3))
        as_bytes = bool(rng.integers(0, 2))
        where = int(rng.integers(0, 3)) if as_bytes else 0   # the bytes: pageable, page-locked (the copy-free road), every other one
        threads = int(rng.choice([1, 4, 8]))
        only = os.environ.get("FUZZ_ONLY")   # (replaying one batch of a seed: everything drawn, nothing else run)
        if only is not None and it != int(only):
            continue
        if only is not None and os.environ.get("FUZZ_DUMP"):
            import pickle
            uniq = {}
            for j, _ in items:
                uniq.setdefault(j, len(uniq))
            with open(os.environ["FUZZ_DUMP"], "wb") as f:
                pickle.dump({"jpegs": list(uniq), "order": [uniq[j] for j, _ in items], "mode": mode, "as_bytes": as_bytes, "where": where,
                             "threads": threads}, f)
        if only is not None:
            log("batch", it, "images", len(items), "first", items[0][1].shape, "mode", mode, "as bytes", as_bytes, "where", where, "threads", threads,
                "ri0", ri0, "ext", ext, "same", same)
        batch = ca.Batch(gpu)
        batch.set_device_preprocess(mode)
        pinned = None
        if as_bytes:
            srcs = [j for j, _ in items]
            if where:
                pinned = ca.HostBuffer(sum(len(j) + 64 for j in srcs))
                views = pinned.place(srcs)
                srcs = views if where == 1 else [v if i % 2 else j for i, (v, j) in enumerate(zip(views, srcs))]
            batch.upload_jpegs(srcs, host_threads=threads, allow_sampling=True)
        else:
            batch.upload([ca.ImageData(j, allow_sampling=True) for j, _ in items], host_threads=threads)
        for rep in range(2):
            batch.decode()
            batch.wait()
            kernels[batch.last_kernel()] = kernels.get(batch.last_kernel(), 0) + 1
            for i, (_, want) in enumerate(items):
                got = batch.read_output(i)
                n += 1
                if not np.array_equal(got, want):
                    bad += 1
                    log("MISMATCH batch", it, "image", i, "of", len(items), "mode", mode, ("bytes", "pinned bytes", "mixed bytes")[where] if as_bytes else "parsed", "rep", rep,
                        want.shape, int((got != want).any(axis=2).sum()), "pixels")
        del batch
        if pinned is not None:
            pinned.close()
        if it % 10 == 9:
            log("batch", it, "outputs", n, "bad", bad, "%.0f s" % (time.time() - t0), "decodes by kernel", kernels)
    return n, bad


if __name__ == "__main__":
    n, bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 77, int(sys.argv[2]) if len(sys.argv) > 2 else 40)
    print("batch fuzz: outputs", n, "mismatches", bad)
    sys.exit(1 if bad else 0)
