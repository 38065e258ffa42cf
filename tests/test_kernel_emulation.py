"""The gfx950 kernel bodies (compeg_amd/csrc/kernels_body.h) compiled for the host with
ASan + UBSan and run lane by lane (tests/emul).  GPU sanitizers are not available on the
target pool, so this is where out-of-bounds LDS / global accesses and undefined shifts in
the kernel code are caught, including on corrupt streams and hostile tables.  Results must
equal the oracle bit for bit.  This is test infrastructure, not a CPU fallback."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, read_golden
from oracle import oracle as orc
from tools import synth

EMUL_DIR = os.path.join(ROOT, "tests", "emul")
RUNNER = os.path.join(EMUL_DIR, "emul_runner")


@pytest.fixture(scope="module")
def runner():
    # (one build at a time: under pytest-xdist every worker comes here, and two makes writing one binary leave a
    # truncated file behind)
    import fcntl
    with open(os.path.join(EMUL_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        subprocess.check_call(["make", "-C", EMUL_DIR, "-s"])
    return RUNNER


STATS = {}   # rare-path counters of the emulated kernels, summed over every run of this module


def _run(runner, tmp_path, jpeg, fused, waves=1, window=2048, l2=12288, standard=False, coop_passes=1, stage=8, below=None,
         layout_rows=None, chunk=1, padded=False, singles=False):
    p = tmp_path / "in.jpg"
    p.write_bytes(jpeg)
    env = dict(os.environ)
    env.pop("EMUL_FUSED", None)
    env.pop("EMUL_STANDARD", None)
    # a walk covers the whole restart intervals of 4 x 64 data units (a team; small intervals through the walk tables),
    # of 2 x 64 (dense streams) or of 64 (also: a lone wave's geometry, every interval walked speculatively)
    env["EMUL_COOP_PASSES"] = str(coop_passes)
    env["EMUL_STREAM_STAGE"] = str(stage)   # (streamed window: behind which data units of an MCU the rows are staged ...
    env.pop("EMUL_STREAM_BELOW", None)
    env.pop("EMUL_STREAM_ROWS", None)
    env["EMUL_WALK_CHUNK"] = str(chunk)   # (fused = 8: MCUs a lane walks between two looks at what it found)
    env.pop("EMUL_SINGLES", None)
    if singles:
        env["EMUL_SINGLES"] = "1"   # (fused = 6, 8-pixel MCUs: every odd restart interval through the single-MCU form -- the library's for intervals of one MCU)
    env["EMUL_PADDED"] = "1" if padded else "0"   # (the output as the runtime allocates it: rows of whole MCUs, 16 pixels each way)
    if layout_rows is not None:
        env["EMUL_STREAM_ROWS"] = str(layout_rows)   # (fused = 6: the layout kernels' streamed form)
    if below is not None:
        env["EMUL_STREAM_BELOW"] = str(below)   # ... when some lane has fewer words than this left; default: always)
    if standard:
        env["EMUL_STANDARD"] = "1"
    if fused:
        # 1 = fused kernel, 2 = paired-wave kernel, 3 = entropy + IDCT kernels, 5 = cooperative kernel,
        # 8 = the walk + lane-per-MCU route (walk_mcus_422_kernel, then decode_fused_422_mcu_rec_kernel; window: the walk's rows)
        env["EMUL_FUSED"] = str(int(fused))
    r = subprocess.run([runner, str(p), str(tmp_path / "rgba"), str(tmp_path / "ac"), str(tmp_path / "dc"),
                        str(waves), str(window), str(l2)], capture_output=True, text=True, env=env, timeout=600)
    if fused == 5 and "does not qualify for the cooperative kernel" in r.stdout:
        return None   # (no restart interval of 1..256 MCUs, or tables the direct tables cannot hold)
    if fused == 8 and "does not qualify for the walk" in r.stdout:
        return None   # (tables the direct tables cannot hold, a DC category above 15)
    assert r.returncode == 0, r.stdout + r.stderr[-2000:]
    _, w, h, _ = r.stdout.split()
    for line in r.stderr.splitlines():
        if line.startswith("stats ") or line.startswith("coop ") or line.startswith("coopdead ") or line.startswith("walk "):
            for kv in line.split()[1:]:
                k, v = kv.split("=")
                STATS[k] = STATS.get(k, 0) + int(v)
    return np.fromfile(tmp_path / "rgba", dtype=np.uint8).reshape(int(h), int(w), 4)


def _check(runner, tmp_path, jpeg, **kw):
    want = orc.ImageData(jpeg).decode()
    for fused in (1, 2, 3, 0):   # fused kernel, paired-wave kernel, two-kernel pipeline, reference-style split kernels
        got = _run(runner, tmp_path, jpeg, fused, **kw)
        assert np.array_equal(got, want), f"fused={fused}: {(got != want).any(axis=2).sum()} pixels differ"
    # the fused kernel with its window in the streamed form: rows enough for an MCU; far too few (lanes keep running
    # out of them inside data units and come back with the next rows)
    for rows in (16, 3):
        got = _run(runner, tmp_path, jpeg, 7, **dict(kw, window=rows))
        assert np.array_equal(got, want), f"streamed window, {rows} rows: {(got != want).any(axis=2).sum()} pixels differ"
    # the cooperative kernel, with the window the runtime would plan and with the test's (possibly cut short: the
    # walks that leave it hand their interval to the serial decoder)
    for window in sorted({0, kw.get("window", 0)}):
        for passes in (4, 2, 1):
            got = _run(runner, tmp_path, jpeg, 5, window=window, coop_passes=passes)
            if got is not None:
                assert np.array_equal(got, want), (f"cooperative kernel, window {window}, {passes} round(s) per walk: "
                                                   f"{(got != want).any(axis=2).sum()} pixels differ")
    # the walk + lane-per-MCU route: rows enough for several MCUs; for about one (lanes keep running out and take the
    # slow road through the wave's window); fewer than the walk reads at a step (everything by the slow road)
    for rows, below, chunk in ((40, 20, 1), (12, 3, 2), (2, 1, 1), (64, 30, 5)):
        got = _run(runner, tmp_path, jpeg, 8, window=rows, below=below, chunk=chunk)
        if got is not None:
            assert np.array_equal(got, want), f"walk + lane-per-MCU route, {rows} rows: {(got != want).any(axis=2).sum()} pixels differ"


CASES = [
    (64, 8, 0, 100, 1, 1), (320, 200, 0, 85, 4, 2), (128, 64, 1, 95, 1, 3), (250, 70, 0, 50, 3, 4),
    (33, 17, 0, 90, 1, 5), (256, 64, 2, 85, 7, 6), (96, 48, 1, 100, 0, 7),
    (500, 40, 0, 85, 1, 8),   # DRI = 1 over several waves, right edge cut: the rows that leave wave-wide
]


@pytest.mark.parametrize("w,h,kind,q,ri,seed", CASES)
def test_emulated_kernels_match_oracle(runner, tmp_path, w, h, kind, q, ri, seed):
    _check(runner, tmp_path, synth.make_jpeg(w, h, seed=seed, kind=kind, quality=q, ri=ri))


# (device_types.h: coop_shape) a lane per interval through the walk tables up to 40 MCUs; whole intervals laid end to
# end in the rounds, straddling them unless 4 * DRI divides 64; speculative walks beyond 40; one interval per team in
# more than four rounds, lists in their own area, beyond 64
@pytest.mark.parametrize("ri", [3, 5, 6, 7, 10, 13, 16, 17, 30, 40, 41, 60, 64, 65, 100, 120, 240, 256])
def test_emulated_cooperative_kernel_any_restart_interval(runner, tmp_path, ri):
    for (w, h, kind, q, seed) in [(320, 64, 0, 85, 1), (256, 48, 1, 95, 2)] + ([(960, 96, 0, 85, 3)] if ri > 60 else []):
        jpeg = synth.make_jpeg(w, h, seed=seed + ri, kind=kind, quality=q, ri=ri)
        want = orc.ImageData(jpeg).decode()
        for passes in (4, 2, 1):
            for window in (0, 200):   # as planned; cut short (walks leave it: the serial decoder inside the kernel)
                got = _run(runner, tmp_path, jpeg, 5, window=window, coop_passes=passes)
                assert got is not None
                assert np.array_equal(got, want), (ri, w, h, passes, window, int((got != want).any(axis=2).sum()))
    # corrupt streams at this restart interval
    rng = np.random.default_rng(ri)
    base = synth.make_jpeg(320, 64, seed=9, kind=0, quality=80, ri=ri)
    scan_at = base.find(b"\xff\xda") + 14
    for it in range(3):
        j = bytearray(base)
        for _ in range(int(rng.integers(1, 12))):
            pos = int(rng.integers(scan_at, len(j) - 2))
            if j[pos] != 0xFF and j[pos - 1] != 0xFF:
                j[pos] ^= 1 << int(rng.integers(0, 8))
                if j[pos] == 0xFF:
                    j[pos] = 0xFE
        want = orc.ImageData(bytes(j)).decode()
        for passes in (4, 1):
            got = _run(runner, tmp_path, bytes(j), 5, window=0, coop_passes=passes)
            assert got is not None and np.array_equal(got, want), (ri, it, passes)


@pytest.mark.parametrize("ri", [1, 2, 4, 7, 10, 16, 30, 120, 0])
def test_emulated_streamed_window_any_restart_interval(runner, tmp_path, ri):
    """decode_fused_422_stream_kernel's body: every lane's rows staged MCU by MCU, whatever the interval's length;
    lanes that reach their last row inside a data unit finish it with the reference reader and come back."""
    STATS.clear()
    for (w, h, kind, q, seed) in [(320, 64, 0, 85, 1), (256, 48, 1, 95, 2), (200, 40, 2, 100, 3)]:
        jpeg = synth.make_jpeg(w, h, seed=seed + ri, kind=kind, quality=q, ri=ri)
        want = orc.ImageData(jpeg).decode()
        for rows, stage, below in ((2, 8, None), (5, 0xa, 3), (12, 0xf, 6), (40, 8, 20), (3, 0xf, 0), (24, 8, 10)):
            got = _run(runner, tmp_path, jpeg, 7, window=rows, stage=stage, below=below)
            assert np.array_equal(got, want), (ri, w, h, rows, stage, below, int((got != want).any(axis=2).sum()))
    assert STATS.get("left_window", 0) > 0 and STATS.get("fast_dus", 0) > 0, STATS
    # corrupt streams
    rng = np.random.default_rng(100 + ri)
    base = synth.make_jpeg(320, 64, seed=9, kind=0, quality=80, ri=ri)
    scan_at = base.find(b"\xff\xda") + 14
    for it in range(4):
        j = bytearray(base)
        for _ in range(int(rng.integers(1, 12))):
            pos = int(rng.integers(scan_at, len(j) - 2))
            if j[pos] != 0xFF and j[pos - 1] != 0xFF:
                j[pos] ^= 1 << int(rng.integers(0, 8))
                if j[pos] == 0xFF:
                    j[pos] = 0xFE
        want = orc.ImageData(bytes(j)).decode()
        for rows in (3, 16):
            got = _run(runner, tmp_path, bytes(j), 7, window=rows)
            assert np.array_equal(got, want), (ri, it, rows)


@pytest.mark.parametrize("ri", [1, 2, 4, 7, 10, 16, 30, 60, 120, 0])
def test_emulated_walk_route_any_restart_interval(runner, tmp_path, ri):
    """walk_mcus_422_kernel's body (the cooperative kernel's walk loop over streamed rows, an MCU at a time, the DC
    differences and quirk Q1's test behind it, the slow road through the wave's window) and the fused kernel begun from
    the walk's records: any restart interval, rows from plenty to none, corrupt streams."""
    for (w, h, kind, q, seed) in [(320, 64, 0, 85, 1), (256, 48, 1, 95, 2), (200, 40, 2, 100, 3), (960, 24, 0, 85, 4)]:
        jpeg = synth.make_jpeg(w, h, seed=seed + ri, kind=kind, quality=q, ri=ri)
        want = orc.ImageData(jpeg).decode()
        for rows, below, chunk in ((64, 24, 1), (24, 10, 1), (9, 4, 1), (3, 1, 1), (64, 30, 3), (100, 50, 16), (30, 10, 4)):
            got = _run(runner, tmp_path, jpeg, 8, window=rows, below=below, chunk=chunk)
            assert got is not None
            assert np.array_equal(got, want), (ri, w, h, rows, below, int((got != want).any(axis=2).sum()))
    rng = np.random.default_rng(300 + ri)
    base = synth.make_jpeg(320, 64, seed=9, kind=0, quality=80, ri=ri)
    scan_at = base.find(b"\xff\xda") + 14
    for it in range(4):
        j = bytearray(base)
        for _ in range(int(rng.integers(1, 12))):
            pos = int(rng.integers(scan_at, len(j) - 2))
            if j[pos] != 0xFF and j[pos - 1] != 0xFF:
                j[pos] ^= 1 << int(rng.integers(0, 8))
                if j[pos] == 0xFF:
                    j[pos] = 0xFE
        want = orc.ImageData(bytes(j)).decode()
        for rows, chunk in ((40, 1), (8, 1), (60, 4)):
            got = _run(runner, tmp_path, bytes(j), 8, window=rows, below=rows // 2, chunk=chunk)
            assert got is not None and np.array_equal(got, want), (ri, it, rows, chunk)


@pytest.mark.parametrize("waves,window,l2", [(4, 64, 512), (2, 80, 0), (1, 4, 0), (3, 300, 1024), (2, 2048, 3000),
                                              (2, 80, 12288), (1, 70, 12288), (3, 300, 12288)])
def test_emulated_kernels_small_lds_budgets(runner, tmp_path, waves, window, l2):
    """Windows / LUT staging cut short by the LDS budget: the global-memory paths."""
    _check(runner, tmp_path, synth.make_jpeg(320, 64, seed=11, kind=1, quality=92, ri=2),
           waves=waves, window=window, l2=l2)


def test_emulated_reference_fixtures(runner, tmp_path):
    for name in ("64x8-Ri-1.jpg", "64x8-Ri-2.jpg"):
        _check(runner, tmp_path, read_golden("refs", name))


def test_emulated_corrupt_entropy_data(runner, tmp_path):
    """Bit flips inside the scan: decoding runs off the rails (long codes, huge runs, reads
    past the interval and past the end of the scan) but must stay memory-safe and must still
    equal the oracle's restatement of the reference semantics."""
    rng = np.random.default_rng(5)
    base = synth.make_jpeg(192, 48, seed=21, kind=0, quality=75, ri=2)
    scan_at = base.find(b"\xff\xda") + 14
    for it in range(12):
        j = bytearray(base)
        for _ in range(int(rng.integers(1, 30))):
            pos = int(rng.integers(scan_at, len(j) - 2))
            if j[pos] != 0xFF and j[pos - 1] != 0xFF:   # keep the marker structure intact
                j[pos] ^= 1 << int(rng.integers(0, 8))
                if j[pos] == 0xFF:
                    j[pos] = 0xFE
        _check(runner, tmp_path, bytes(j), window=64 if it % 2 else 2048)


def test_emulated_hostile_huffman_tables(runner, tmp_path):
    """DC categories above 15 and other values a baseline encoder never emits: shift counts
    wrap modulo 32 exactly like the WGSL reference (quirk list in SURVEY.md)."""
    rng = np.random.default_rng(8)
    base = synth.make_jpeg(128, 32, seed=31, kind=1, quality=90, ri=1)
    i = base.find(b"\xff\xc4")
    for it in range(10):
        counts = np.zeros(16, dtype=np.uint8)
        counts[1] = 2
        counts[2] = 3
        counts[4] = int(rng.integers(1, 4))
        counts[8] = int(rng.integers(0, 6))
        counts[15] = int(rng.integers(0, 30))
        nsym = int(counts.sum())
        syms = rng.integers(0, 256, nsym, dtype=np.uint8)   # DC tables with categories up to 255
        tcth = [0x00, 0x01, 0x10, 0x11][it % 4]
        seg = bytes([0xFF, 0xC4]) + (2 + 17 + nsym).to_bytes(2, "big") + bytes([tcth]) + counts.tobytes() + syms.tobytes()
        # put the hostile table AFTER the regular ones so that it wins
        sos = base.find(b"\xff\xda")
        _check(runner, tmp_path, base[:sos] + seg + base[sos:], window=128)


def test_emulated_cooperative_kernel_on_the_gpu_suite_inputs(runner, tmp_path):
    """The corrupt / hostile inputs of tests/test_gpu_parity.py through the emulated cooperative kernel (one of
    them -- an underflow of the reference reader at a DC code whose cut-off bits select a category above 15 --
    was first caught on the GPU: such an interval must go to the serial decoder, not to the zero-stream shortcut)."""
    import test_gpu_parity as gp
    checked = 0
    for j in gp._corrupt_variants(12) + gp._hostile_table_variants(10):
        try:
            want = orc.ImageData(j).decode()
        except orc.OracleError:
            continue
        for passes in (4, 1):
            got = _run(runner, tmp_path, j, 5, window=0, coop_passes=passes)
            if got is not None:
                assert np.array_equal(got, want), f"{passes} round(s) per walk: {(got != want).any(axis=2).sum()} pixels differ"
                checked += passes == 4
    assert checked >= 10


@pytest.mark.parametrize("sampling", [(2, 1), (2, 2), (1, 1), (1, 2)])
def test_emulated_outputs_of_whole_mcus(runner, tmp_path, sampling):
    """The output as the runtime allocates it -- rows of whole MCUs, 16 pixels each way: an MCU (group) the image's edge
    cuts is stored whole, its outside into the padding (and, in the layouts' kernels, one that is cut at a 16-byte
    piece's end but not whole goes through the quad with its limits) -- ragged sizes, even and odd restart intervals,
    tight buffers beside them; under ASan: nothing is written behind the allocation."""
    # (264, 200, 195 pixels across: an odd number of 8-pixel MCUs a row -- with an even restart interval the pairs of the
    # 4:4:4 / 4:4:0 kernels have their second MCU at the next MCU row's beginning once in two rows: each half its own place)
    for (w, h, ri) in ((250, 70, 3), (250, 70, 2), (33, 17, 1), (264, 120, 4), (1080 // 4, 104, 6), (200, 64, 2), (195, 50, 4), (264, 41, 2), (200, 64, 3), (195, 50, 5), (264, 41, 7),
                       (8, 93, 4), (8, 40, 3), (16, 30, 2)):   # (one MCU across: a pair is two MCU rows -- the emulation fuzz's find)
        jpeg = synth.make_jpeg(w, h, seed=50 + w + ri, kind=1, quality=85, ri=ri, sampling=sampling)
        want = orc.ImageData(jpeg, allow_sampling=True).decode()
        for padded in (True, False):
            got = _run(runner, tmp_path, jpeg, 1 if sampling == (2, 1) else 6, padded=padded)
            assert got is not None and np.array_equal(got, want), (sampling, w, h, ri, padded)


def test_emulated_walk_route_dc_code_cut_by_the_readers_buffer(runner, tmp_path):
    """tests/golden/route/cut_dc_code.jpg: at one data unit's start the stream's bits are no DC code, the six bits the
    reference's reader has left (zeros behind them: it is not topped up in front of DC codes) are -- of more bits than it
    has: it runs dry (quirk Q1).  The route's walk has to notice that the two readings differ (`slow_cut`) and hand
    the MCU to the slow road; rows enough for the lean walk to get there, and few (the slow road all the way)."""
    jpeg = read_golden("route", "cut_dc_code.jpg")
    want = orc.ImageData(jpeg).decode()
    before = STATS.get("slow_cut", 0)
    for rows, chunk in ((64, 6), (16, 1)):
        got = _run(runner, tmp_path, jpeg, 8, window=rows, chunk=chunk)
        assert got is not None and np.array_equal(got, want), (rows, chunk)
    assert STATS.get("slow_cut", 0) > before
    assert STATS.get("dead_mcus", 0) > 0


def test_emulated_speculative_walks_need_no_serial_decoder(runner, tmp_path):
    """Frames on which the cooperative kernel's speculative walks (more than 40 MCUs an interval) used to give an interval
    up -- handed to the serial decoder: a millisecond on the GPU for a frame of 250 us -- now walk all of them: the last
    walker of an interval stands up to 140 bits behind the interval's end when it begins its last data units (room for
    that in the window: kCoopEndSlack), and an image's last counted interval ends where the next start position says
    when the scan goes on behind it (an image whose MCUs the interval does not divide).  Counted by the emulator
    (`coop ... serial=`), bit-exact as ever.  (profiles/r04/NOTES.md, tools/coop_cliff_probe.py)"""
    for (w, h, ri, seed) in ((1920, 1080, 120, 4127), (1920, 1080, 60, 4121), (960, 720, 250, 4121)):
        jpeg = synth.make_jpeg(w, h, seed=seed, kind=0, quality=85, ri=ri)
        before = STATS.get("serial", 0), STATS.get("intervals", 0)
        got = _run(runner, tmp_path, jpeg, 5, window=0, coop_passes=4)
        assert got is not None and np.array_equal(got, orc.ImageData(jpeg).decode()), (w, h, ri, seed)
        assert STATS.get("intervals", 0) > before[1]
        assert STATS.get("serial", 0) == before[0], (w, h, ri, seed, "an interval went to the serial decoder")


def test_emulated_count_mismatch_and_truncated_interval(runner, tmp_path):
    j = bytearray(synth.make_jpeg(128, 32, seed=40, ri=2))
    i = j.find(b"\xff\xdd")
    j[i + 4:i + 6] = (3).to_bytes(2, "big")     # DRI says 3, stream has markers every 2 MCUs
    _check(runner, tmp_path, bytes(j))


def test_emulated_long_codes_and_dense_blocks(runner, tmp_path):
    """Noise at quality 100: AC codes of up to 16 bits (escapes from the direct tables), data
    units that end at position 63 without EOB and large DC differences -- the data units after
    which the reference reader reaches a DC code with few buffered bits (quirk Q1)."""
    for seed in (51, 52):
        _check(runner, tmp_path, synth.make_jpeg(96, 32, seed=seed, kind=1, quality=100, ri=3))


def test_emulated_reference_reader_underflow_on_valid_streams(runner, tmp_path):
    """Quirk Q1 on VALID streams: noise at quality 100 whose data units now and then end on a long code right
    in front of a large DC difference, at a bit alignment that leaves the reference's un-refilled reader short
    of bits (found by search; about one image in fifty of this kind has such a data unit).  Everything behind
    the underflow decodes from zeros in the reference.  In the cooperative kernel that is a dead data unit
    (DC difference from what is left of the reader, AC levels from the host's zero-stream record) and
    zero-stream data units behind it; in the other kernels the lane's switch to the exact reader."""
    for (w, h, seed) in [(256, 64, 77), (512, 128, 86), (512, 128, 163)]:
        before = STATS.get("dead", 0), STATS.get("zero", 0), STATS.get("left_underflow", 0), STATS.get("slow_q1", 0)
        jpeg = synth.make_jpeg(w, h, seed=seed, kind=1, quality=100, ri=4)
        _check(runner, tmp_path, jpeg)
        assert STATS.get("dead", 0) > before[0] and STATS.get("left_underflow", 0) > before[2]
        # (the walk + lane-per-MCU route with rows enough for these MCUs: its own test at the DC codes finds it)
        assert np.array_equal(_run(runner, tmp_path, jpeg, 8, window=160, below=60), orc.ImageData(jpeg).decode())
    assert STATS.get("slow_q1", 0) > 0
    assert STATS.get("zero", 0) > 0
    # long restart intervals: MCUs behind the underflow inside the interval (records that say "run dry")
    before = STATS.get("dead_mcus", 0)
    for (w, h, seed) in [(256, 64, 77), (512, 128, 86)]:
        jpeg = synth.make_jpeg(w, h, seed=seed, kind=1, quality=100, ri=w // 16)
        want = orc.ImageData(jpeg).decode()
        for rows, chunk in ((160, 1), (20, 1), (200, 3)):
            assert np.array_equal(_run(runner, tmp_path, jpeg, 8, window=rows, below=rows // 2, chunk=chunk), want)
    assert STATS.get("dead_mcus", 0) > before


def test_emulated_q1_underflow_in_every_quarter(runner, tmp_path):
    """The tiles of tests/test_gpu_parity.py::Q1_TILES: the emulated team kernel (decoding by quarters) meets the
    reference reader's underflow in each of the four quarters of an interval, and equals the oracle."""
    import test_gpu_parity as gp
    jpeg, tiles = gp.q1_frame(1024, 64, every=5)
    want = orc.ImageData(jpeg).decode()
    before = [STATS.get(f"q{q}", 0) for q in range(4)]
    got = _run(runner, tmp_path, jpeg, 5, window=0, coop_passes=4)
    assert np.array_equal(got, want)
    assert all(STATS.get(f"q{q}", 0) > before[q] for q in range(4)), STATS
    for fused in (1, 2):
        assert np.array_equal(_run(runner, tmp_path, jpeg, fused), want)
    assert np.array_equal(_run(runner, tmp_path, jpeg, 8, window=40, below=20), want)
    assert np.array_equal(_run(runner, tmp_path, jpeg, 8, window=120, below=50, chunk=4), want)


@pytest.mark.parametrize("sampling", [(1, 1), (2, 1), (1, 2), (2, 2)])
def test_emulated_extension_layouts(runner, tmp_path, sampling):
    """4:4:4, 4:2:2, 4:4:0, 4:2:0 through the extension pipeline (entropy records, IDCT in place,
    generic composite) against the oracle with the same extension switched on."""
    # (8-pixel MCUs in pairs, also across the end of an MCU row -- 250 and 33 pixels are 32 and 5 MCUs --, the last MCU of
    # an odd restart interval alone; intervals of one MCU singly)
    for (w, h, kind, q, ri, seed) in [(96, 48, 0, 90, 2, 91), (250, 70, 1, 75, 3, 92), (33, 17, 2, 85, 1, 93), (250, 70, 0, 85, 4, 94),
                                      (33, 17, 1, 90, 2, 95), (40, 24, 0, 85, 6, 96)]:
        jpeg = synth.make_jpeg(w, h, seed=seed, kind=kind, quality=q, ri=ri, sampling=sampling)
        want = orc.ImageData(jpeg, allow_sampling=True).decode()
        got = _run(runner, tmp_path, jpeg, 4)
        assert np.array_equal(got, want), f"{sampling} {w}x{h}: {(got != want).any(axis=2).sum()} pixels differ"
        if sampling != (2, 1):   # the fused kernels of the extension layouts (decode_fused_444 / _440 / _420_kernel)
            got = _run(runner, tmp_path, jpeg, 6, waves=3, window=300)
            assert np.array_equal(got, want), f"fused {sampling} {w}x{h}: {(got != want).any(axis=2).sum()} pixels differ"
            if ri % 2:   # (pairs with the interval's last MCU alone above; here the single-MCU form)
                got = _run(runner, tmp_path, jpeg, 6, waves=3, window=300, singles=True)
                assert np.array_equal(got, want), f"fused, single MCUs {sampling} {w}x{h}: {(got != want).any(axis=2).sum()} pixels differ"
            # ... and their streamed form (the odd restart intervals too: the body is the same, only the GPU library
            # has no kernel of it for them): rows enough; too few; staged behind every data unit
            for rows, stage, below in ((24, 8, 24), (3, 8, 2), (6, 0xf, 3)):
                got = _run(runner, tmp_path, jpeg, 6, waves=2, layout_rows=rows, stage=stage, below=below)
                assert np.array_equal(got, want), f"streamed {sampling} {w}x{h} rows {rows}: {(got != want).any(axis=2).sum()} pixels differ"


def test_emulated_standard_entropy_extension(runner, tmp_path):
    """COMPEG_PARSE_STANDARD_ENTROPY: refill in front of DC codes, ZRL = 16 -- every pipeline against the
    oracle with the same switch, on content where the reference's reader underflows (noise at
    quality 100) and with cut-short windows (exact-mode path)."""
    for (w, h, kind, q, ri, seed, window) in [(96, 32, 1, 100, 3, 51, 2048), (320, 64, 1, 92, 2, 11, 80),
                                              (250, 70, 0, 50, 3, 4, 2048), (320, 200, 0, 95, 4, 5, 2048),
                                              (96, 32, 1, 100, 4, 51, 2048)]:
        jpeg = synth.make_jpeg(w, h, seed=seed, kind=kind, quality=q, ri=ri)
        want = orc.ImageData(jpeg, standard_entropy=True).decode()
        if kind == 0:
            assert not np.array_equal(want, orc.ImageData(jpeg).decode())   # the switch matters on this input
        for fused in (1, 2, 3, 0, 5, 54, 8):
            got = _run(runner, tmp_path, jpeg, fused % 10, window=40 if fused == 8 else window, standard=True, coop_passes=4 if fused == 54 else 1)
            if got is None:
                continue   # (tables the direct tables cannot hold)
            assert np.array_equal(got, want), f"fused={fused} {w}x{h}: {(got != want).any(axis=2).sum()} pixels differ"


def test_emulated_rare_paths_were_reached():
    """Runs last in this module: the cases above must have exercised every branch of the fast
    entropy path (counters come from the emulator build, -DCG_EMUL_STATS)."""
    print("emulation path counters:", STATS)
    if not STATS or os.environ.get("PYTEST_XDIST_WORKER"):
        pytest.skip("the cases of this module ran in other processes (pytest -n): their counters are not here")
    assert STATS.get("fast_dus", 0) > 10000
    assert STATS.get("exact_dus", 0) > 100
    for key in ("left_window", "left_underflow", "dc_cut", "escapes"):
        assert STATS.get(key, 0) > 0, (key, STATS)
    # ... and of the cooperative kernel: intervals settled in one round and in several, lanes that walked on,
    # reference-reader underflows (dead data units, zero-stream data units behind them), serial hand-overs
    assert STATS.get("intervals", 0) > 1000
    for key in ("continued", "dead", "zero", "serial"):
        assert STATS.get(key, 0) > 0, (key, STATS)
    # ... and of the walk + lane-per-MCU route: MCUs through the walk tables, by the slow road (quirk Q1, rows run out),
    # MCUs behind an underflow, long codes inside the loop
    for key in ("lean_mcus", "slow_mcus", "slow_q1", "slow_rows", "dead_mcus", "long_codes", "restages"):
        assert STATS.get(key, 0) > 0, (key, STATS)


def test_threaded_host_scan_under_the_thread_sanitizer(runner):
    """compeg_amd/csrc/scan.cpp with helper threads: same buffers as the one-thread loop, no data race."""
    exe = os.path.join(EMUL_DIR, "scan_threads_check")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "bad 0" in r.stdout and "ThreadSanitizer" not in r.stderr
