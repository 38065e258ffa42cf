"""Randomized check of the walk + lane-per-MCU route on the GPU: batches sized so that the dispatch takes it (few or long
restart intervals, at most a wave of them per SIMD) -- frames of one stream and mixed sizes, restart intervals from 8 MCUs
to a whole image (no DRI), qualities 50 to 95, smooth / noisy / sparse content, bit flips in some scans, both entropy
modes, host and device preprocessing, launches of part of a batch -- every checked output against the oracle, which
kernel ran counted.
    python tools/fuzz_gpu_walk.py [seed] [batches]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compeg_amd as ca
import oracle.oracle as orc
from tools import synth

SIZES = [(640, 360), (960, 720), (1280, 720), (1920, 1080), (250, 70), (1000, 600), (1016, 990), (3840, 2160)]


def run(seed=4242, batches=60, log=print):
    rng = np.random.default_rng(seed)
    gpu = ca.Gpu.open()
    bad = n = 0
    kernels = {}
    t0 = time.time()
    for it in range(batches):
        same = bool(rng.integers(0, 3))
        w0, h0 = SIZES[int(rng.integers(0, len(SIZES)))]
        ri0 = int(rng.choice([8, 10, 16, 30, 45, 60, 120, 240, 0, 0]))
        std = bool(rng.integers(0, 5) == 0)
        distinct = int(rng.integers(2, 6))
        frames = []
        for i in range(distinct):
            w, h = (w0, h0) if same else SIZES[int(rng.integers(0, len(SIZES) - 1))]
            ri = ri0 if same or rng.integers(0, 2) else int(rng.choice([8, 30, 60]))
            j = bytearray(synth.make_jpeg(w, h, seed=int(rng.integers(1, 1 << 30)), kind=int(rng.choice([0, 0, 0, 1, 2])),
                                          quality=int(rng.choice([50, 70, 85, 85, 95])), ri=ri))
            if rng.integers(0, 4) == 0:
                scan_at = j.find(b"\xff\xda") + 14
                for _ in range(int(rng.integers(1, 30))):
                    pos = int(rng.integers(scan_at, len(j) - 2))
                    if j[pos] != 0xFF and j[pos - 1] != 0xFF:
                        j[pos] ^= 1 << int(rng.integers(0, 8))
                        if j[pos] == 0xFF:
                            j[pos] = 0xFE
            if os.environ.get("FUZZ_ONES") is not None and rng.integers(0, 3) == 0:
                # runs of one bits (0xFF 0x00: eight of them): bits that are no Huffman code, at DC codes and AC codes alike
                scan_at = j.find(b"\xff\xda") + 14
                for _ in range(int(rng.integers(3, 40))):
                    pos = int(rng.integers(scan_at, max(scan_at + 1, len(j) - 12)))
                    if j[pos - 1] == 0xFF or j[pos] == 0xFF:
                        continue
                    run_ = bytes([0xFF, 0x00] * int(rng.integers(1, 3)) + [0xFE] * int(rng.integers(0, 2)))
                    if 0xFF in j[pos + len(run_):pos + len(run_) + 1]:
                        continue
                    j[pos:pos + len(run_)] = run_
            j = bytes(j)
            try:
                frames.append((j, orc.ImageData(j, standard_entropy=std).decode()))
            except orc.OracleError:
                pass
        if not frames:
            continue
        # as many slots as keep the launch at a wave of intervals per SIMD or fewer (what the route is for), at least a few
        mcus = max(((orc.ImageData(j).width() + 15) // 16) * ((orc.ImageData(j).height() + 7) // 8) for j, _ in frames)
        intervals = max(1, mcus // max(ri0, 1)) if ri0 else 1
        slots = int(np.clip(rng.integers(4, 400), 2, max(2, 60000 // max(1, (intervals + 63) // 64 * 64) * 1)))
        slots = min(slots, 2 if w0 >= 3840 and not ri0 else (24 if w0 >= 3840 else 400))
        order = [int(rng.integers(0, len(frames))) for _ in range(slots)]
        images = [ca.ImageData(frames[k][0], standard_entropy=std) for k in range(len(frames))]
        mode = int(rng.integers(0, 3))
        only = os.environ.get("FUZZ_ONLY")   # (replay of one batch: the others only draw their random numbers)
        if only is not None and int(only) != it:
            if rng.integers(0, 5) == 0 and slots > 3:
                rng.integers(2, slots)
            for rep in range(2):
                rng.integers(0, slots), rng.integers(0, slots)
            continue
        if only is not None:
            log(f"batch {it}: {len(frames)} frames {[(orc.ImageData(j).width(), orc.ImageData(j).height(), len(j)) for j, _ in frames]} ri {ri0} slots {slots} mode {mode} std {std} same {same}")
            if os.environ.get("FUZZ_DUMP"):
                for k, (j, _) in enumerate(frames):
                    open(os.path.join(os.environ["FUZZ_DUMP"], f"batch{it}_frame{k}.jpg"), "wb").write(j)
                open(os.path.join(os.environ["FUZZ_DUMP"], f"batch{it}_order.txt"), "w").write(" ".join(map(str, order)))
        b = ca.Batch(gpu)
        b.set_device_preprocess(mode)
        b.upload([images[k] for k in order])
        if rng.integers(0, 5) == 0 and slots > 3:
            ch = int(rng.integers(2, slots))
            b.set_chunk(ch)
            if only is not None:
                log(f"  chunk {ch}")
        for rep in range(2):
            b.decode()
            b.wait()
            kernels[b.last_kernel()] = kernels.get(b.last_kernel(), 0) + 1
            picks = sorted({0, slots - 1, int(rng.integers(0, slots)), int(rng.integers(0, slots))} |
                           {order.index(k) for k in set(order)})   # (... and one slot of every distinct frame)
            for i in (range(slots) if only is not None else picks):
                n += 1
                if not np.array_equal(b.read_output(i), frames[order[i]][1]):
                    bad += 1
                    log(f"MISMATCH batch {it} slot {i} rep {rep} kernel {b.last_kernel()} seed {seed} ri {ri0} same {same} std {std} mode {mode}")
        if it % 10 == 9:
            log(f"  {it + 1} batches, {n} outputs compared, {bad} mismatches, kernels {kernels}, {time.time() - t0:.0f} s")
    log(f"fuzz_gpu_walk seed {seed}: {batches} batches, {n} outputs compared, {bad} mismatches; kernels {kernels}; {time.time() - t0:.0f} s")
    return n, bad


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 4242
    batches = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    n, bad = run(seed, batches)
    sys.exit(1 if bad else 0)
