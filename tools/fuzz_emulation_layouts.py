"""One-off randomized check of the extension layouts' fused kernels under the emulator (ASan/UBSan): random geometry -- odd and
even MCU counts a row --, sampling, restart interval (odd ones: pairs with the last MCU alone, or the single form), bit flips,
outputs tight and as the runtime allocates them, whole windows and streamed ones -- against the oracle.
    python tools/fuzz_emulation_layouts.py [seed] [iterations]   (tests/emul/emul_runner must be built)"""
import os, subprocess, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import synth
import oracle.oracle as orc
RUN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "emul", "emul_runner")
tmp = tempfile.mkdtemp()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 4321)
n = bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 120):
    samp = [(1, 1), (1, 2), (2, 2)][int(rng.integers(0, 3))]
    w = int(rng.integers(8, 420)); h = int(rng.integers(8, 150)); ri = int(rng.integers(1, 12))
    if it % 5 == 4:   # (one to three MCUs across)
        w = int(rng.integers(1, 40))
    j = bytearray(synth.make_jpeg(w, h, seed=int(rng.integers(1, 1 << 30)), kind=int(rng.integers(0, 3)), quality=int(rng.choice([50, 85, 95])), ri=ri, sampling=samp))
    if it % 3 == 0:
        at = j.find(b"\xff\xda") + 14
        for _ in range(int(rng.integers(1, 20))):
            pos = int(rng.integers(at, len(j) - 2))
            if j[pos] != 0xFF and j[pos - 1] != 0xFF:
                j[pos] ^= 1 << int(rng.integers(0, 8))
                if j[pos] == 0xFF: j[pos] = 0xFE
    j = bytes(j)
    try:
        want = orc.ImageData(j, allow_sampling=True).decode()
    except orc.OracleError:
        continue
    p = os.path.join(tmp, "in.jpg"); open(p, "wb").write(j)
    for padded in ("0", "1"):
        for singles in (False, True):
            if singles and not (samp[0] == 1 and ri % 2 and ri > 1):
                continue
            env = dict(os.environ, EMUL_FUSED="6", EMUL_PADDED=padded, EMUL_COOP_PASSES="1", EMUL_STREAM_STAGE="8", EMUL_WALK_CHUNK="1")
            rows = int(rng.choice([0, 0, 4, 24]))
            if rows: env["EMUL_STREAM_ROWS"] = str(rows)
            if singles: env["EMUL_SINGLES"] = "1"
            r = subprocess.run([RUN, p, tmp + "/rgba", tmp + "/ac", tmp + "/dc", str(int(rng.integers(1, 4))), "300", "12288"], capture_output=True, text=True, env=env, timeout=600)
            n += 1
            ok = r.returncode == 0
            if ok:
                _, ww, hh, _ = r.stdout.split()
                ok = np.array_equal(np.fromfile(tmp + "/rgba", dtype=np.uint8).reshape(int(hh), int(ww), 4), want)
            if not ok:
                bad += 1; print("MISMATCH", it, samp, w, h, ri, padded, singles, rows, r.stderr[-300:]); open("/tmp/bad_layout_%d.jpg" % it, "wb").write(j)
print("runs", n, "bad", bad)
