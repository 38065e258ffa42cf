"""Corrupt frames without restart markers, or with few: a bit flipped early in the scan and the reference decodes garbage
for the rest of the interval -- long runs of improbable states (codes that are none, a reader that runs dry, ZRLs that
overrun) through the walk + lane-per-MCU route and the cooperative kernel.  Every frame alone in a decoder and three to
a batch, against the oracle.
    python tools/fuzz_gpu_garbage.py [seed] [frames]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compeg_amd as ca
import oracle.oracle as orc
from tools import synth

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = np.random.default_rng(seed)
gpu = ca.Gpu.open(0)
dec = ca.Decoder(gpu)
bad = n = 0
kernels = {}
t0 = time.time()
for it in range(frames):
    w, h = [(1016, 990), (640, 360), (960, 720), (1280, 720)][int(rng.integers(0, 4))]
    ri = int(rng.choice([0, 0, 0, 300, 120, 64, 30, 10]))
    quirky = os.environ.get("FUZZ_QUIRKS") is not None   # valid streams of the kind the reader's quirks live on: noise at q97-100
    if quirky:
        ri = int(rng.choice([0, 300, 240, 120, 60, 30, 16, 10, 7, 4, 2, 1]))
    j = bytearray(synth.make_jpeg(w, h, seed=int(rng.integers(1, 1 << 30)), kind=int(rng.choice([1, 1, 0, 2])) if not quirky else 1,
                                  quality=int(rng.choice([70, 85, 95, 95, 100])) if not quirky else int(rng.choice([97, 99, 100, 100])), ri=ri))
    at = j.find(b"\xff\xda") + 14
    for _ in range(int(rng.integers(1, 12)) if not quirky else 0):
        pos = int(rng.integers(at, at + max(16, (len(j) - at) // int(rng.choice([1, 4, 16])))))
        pos = min(pos, len(j) - 3)
        if j[pos] != 0xFF and j[pos - 1] != 0xFF:
            j[pos] ^= 1 << int(rng.integers(0, 8))
            if j[pos] == 0xFF:
                j[pos] = 0xFE
    if os.environ.get("FUZZ_ONES") is not None:
        # runs of one bits (0xFF 0x00: eight of them): bits that are no Huffman code, at DC codes and AC codes alike
        for _ in range(int(rng.integers(3, 40))):
            pos = int(rng.integers(at, max(at + 1, len(j) - 12)))
            if j[pos - 1] == 0xFF or j[pos] == 0xFF:
                continue
            run_ = bytes([0xFF, 0x00] * int(rng.integers(1, 3)) + [0xFE] * int(rng.integers(0, 2)))
            if 0xFF in j[pos + len(run_):pos + len(run_) + 1]:
                continue
            j[pos:pos + len(run_)] = run_
    j = bytes(j)
    std = bool(rng.integers(0, 6) == 0)
    try:
        want = orc.ImageData(j, standard_entropy=std).decode()
    except orc.OracleError:
        continue
    img = ca.ImageData(j, standard_entropy=std)
    d = ca.Decoder(gpu)   # (a texture of its own: a corrupt scan may leave texels unwritten)
    d.decode_blocking(img)
    kernels[d.last_kernel()] = kernels.get(d.last_kernel(), 0) + 1
    n += 1
    if not np.array_equal(d.read_texture(w, h), want):
        bad += 1
        print(f"MISMATCH decoder frame {it} {w}x{h} ri {ri} std {std} kernel {d.last_kernel()} seed {seed}", flush=True)
        open(f"/tmp/bad_garbage_{seed}_{it}.jpg", "wb").write(j)
    b = ca.Batch(gpu)
    b.set_device_preprocess(int(rng.integers(0, 3)))
    b.upload([img, img, img])
    b.decode(); b.wait()
    kernels[b.last_kernel()] = kernels.get(b.last_kernel(), 0) + 1
    for i in (0, 2):
        n += 1
        if not np.array_equal(b.read_output(i), want):
            bad += 1
            print(f"MISMATCH batch frame {it} slot {i} {w}x{h} ri {ri} std {std} kernel {b.last_kernel()} seed {seed}", flush=True)
print(f"fuzz_gpu_garbage seed {seed}: {frames} frames, {n} outputs compared, {bad} mismatches; kernels {kernels}; {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
