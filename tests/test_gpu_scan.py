"""Device-side scan preprocessing against the oracle's restatement of
src/scan.rs: byte-identical words and start positions, same error text."""
import numpy as np
import pytest

from conftest import read_golden
from oracle import oracle as orc
from tools import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ca():
    import compeg_amd
    return compeg_amd


@pytest.fixture(scope="module")
def gpu(ca):
    return ca.Gpu.open(0)


def _both(ca, gpu, data, expected):
    a, b = ca.ScanBuffer(), orc.ScanBuffer()
    ea = eb = None
    try:
        a.process_on_gpu(gpu, data, expected)
    except ca.Error as e:
        ea = str(e)
        assert e.code == ca.E_COUNT_MISMATCH
    try:
        b.process(data, expected)
    except orc.OracleError as e:
        eb = str(e)
    assert ea == eb
    assert a.start_positions() == b.start_positions()
    assert a.processed_scan_data() == b.processed_scan_data()


def test_reference_kats_on_gpu(ca, gpu):
    # ref src/scan.rs:151-180
    _both(ca, gpu, bytes([0x12, 0x34, 0x56, 0x78]), 1)
    _both(ca, gpu, bytes([0xFF, 0xD0, 0xFF, 0xD0]), 3)
    _both(ca, gpu, bytes([0xFF, 0x00, 0x44, 0x55, 0xFF, 0xD0, 0x34]), 2)
    _both(ca, gpu, bytes([0x11, 0xFF, 0xD0, 0x11, 0xFF, 0xD0, 0x11]), 3)
    _both(ca, gpu, bytes([0x11, 0xFF, 0xD0, 0x11, 0xFF, 0xD0, 0x11]), 1)   # the mismatch error text
    _both(ca, gpu, b"", 0)
    _both(ca, gpu, b"\xff", 1)
    _both(ca, gpu, b"\xff\xff\xff", 2)
    _both(ca, gpu, b"\xff\xff\x00\xff", 2)


def test_bench_data_on_gpu(ca, gpu):
    data = read_golden("scan", "scan.dat")       # ref benches/bench.rs: 42 876 intervals
    _both(ca, gpu, data, 42876)
    _both(ca, gpu, data, 42875)                  # count mismatch, wrapped slot indices
    _both(ca, gpu, data, 1000)
    _both(ca, gpu, data, 50000)


def test_fuzz_on_gpu(ca, gpu):
    rng = np.random.default_rng(11)
    for it in range(150):
        n = int(rng.integers(0, 20000))
        data = rng.integers(0, 256, n, dtype=np.uint8)
        data[rng.random(n) < 0.08] = 0xFF         # FF pairs, FF FF, FF runs across tile borders
        data[rng.random(n) < 0.05] = 0
        if it % 5 == 0 and n > 5000:
            s = int(rng.integers(0, n - 300))
            data[s:s + int(rng.integers(2, 300))] = 0xFF
        _both(ca, gpu, data.tobytes(), int(rng.integers(0, 400)))


def test_long_ff_run_falls_back_to_host(ca, gpu):
    data = bytes([1, 2, 3]) + b"\xff" * 70000 + bytes([0, 5, 6])
    _both(ca, gpu, data, 35001)


def test_real_scans_on_gpu(ca, gpu):
    for w, h, kind, q, ri in ((640, 360, 0, 85, 4), (1920, 1080, 1, 95, 1), (3840, 2160, 0, 85, 4)):
        jpeg = synth.make_jpeg(w, h, seed=w, kind=kind, quality=q, ri=ri)
        img = orc.ImageData(jpeg)
        _both(ca, gpu, img.scan_data(), img.parallelism())


@pytest.mark.parametrize("mode", [1, 2])
def test_batch_with_device_preprocess(ca, gpu, mode):
    jpegs = [synth.make_jpeg(w, h, seed=80 + i, kind=k, quality=q, ri=ri)
             for i, (w, h, k, q, ri) in enumerate([(640, 360, 0, 85, 4), (320, 240, 1, 95, 1),
                                                   (1280, 720, 0, 70, 8), (64, 8, 0, 100, 2),
                                                   (250, 70, 2, 85, 3), (1920, 1080, 0, 85, 4)])]
    images = [ca.ImageData(j) for j in jpegs]
    batch = ca.Batch(gpu)
    batch.set_device_preprocess(mode)
    batch.upload(images)
    assert batch.host_fallbacks() == 0
    for _ in range(2):
        batch.decode()
    batch.wait()
    for i, j in enumerate(jpegs):
        want = orc.ImageData(j).decode()
        got = batch.read_output(i)
        assert np.array_equal(got, want), f"image {i}"


def test_decoder_with_device_preprocess(ca, gpu):
    dec = ca.Decoder(gpu)
    dec.set_device_preprocess(True)
    for w, h, kind, q, ri, seed in ((640, 360, 0, 85, 4, 1), (250, 70, 0, 50, 3, 2), (1920, 1080, 1, 95, 1, 3),
                                    (64, 8, 0, 100, 1, 4), (3840, 2160, 0, 85, 4, 5)):
        jpeg = synth.make_jpeg(w, h, seed=seed, kind=kind, quality=q, ri=ri)
        data = ca.ImageData(jpeg)
        dec.decode_blocking(data)
        got = dec.read_texture(w, h)
        assert np.array_equal(got, orc.ImageData(jpeg).decode()), (w, h)
    # count mismatch is still only a warning
    j = bytearray(synth.make_jpeg(128, 32, seed=40, ri=2))
    i = j.find(b"\xff\xdd")
    j[i + 4:i + 6] = (4).to_bytes(2, "big")
    dec.decode_blocking(ca.ImageData(bytes(j)))
    assert dec.last_warning().startswith("restart interval count mismatch: counted")
    assert np.array_equal(dec.read_texture(128, 32), orc.ImageData(bytes(j)).decode())


def test_decoder_device_preprocess_hands_pathological_scans_to_the_host(ca, gpu):
    """decode_blocking runs the scan kernels without a read-back in the middle and looks at their verdict after
    the decode: a scan they give up on (an FF run beyond their look-back bound) is decoded again through the
    host preprocessor, and the next ordinary image takes the device path again."""
    j = synth.make_jpeg(256, 64, seed=3, ri=2)
    sd = orc.ImageData(j).scan_data()
    o = j.find(sd[:16])
    half = len(sd) // 2
    dec = ca.Decoder(gpu)
    dec.set_device_preprocess(True)
    # runs below the bound (the kernels handle them), just above it, and of a megabyte (every chunk of which
    # would otherwise walk back to the bound: the walk is short, so this takes milliseconds)
    for run in (301, 513, 70001, (1 << 20) + 1):
        bad = j[:o] + sd[:half] + b"\xff" * run + b"\x00" + sd[half:] + j[o + len(sd):]
        for jpeg in (j, bad, j, bad):
            dec.decode_blocking(ca.ImageData(jpeg))
            assert np.array_equal(dec.read_texture(256, 64), orc.ImageData(jpeg).decode()), run
        # the non-blocking entry point keeps the read-back in the middle: same results
        for jpeg in (bad, j):
            op = dec.start_decode(ca.ImageData(jpeg))
            op.wait()
            assert np.array_equal(dec.read_texture(256, 64), orc.ImageData(jpeg).decode()), run


def _jpeg_set():
    jpegs = [synth.make_jpeg(w, h, seed=180 + i, kind=k, quality=q, ri=ri)
             for i, (w, h, k, q, ri) in enumerate([(640, 360, 0, 85, 4), (320, 240, 1, 95, 1), (1280, 720, 0, 70, 10),
                                                   (64, 8, 0, 100, 2), (250, 70, 2, 85, 3), (1920, 1080, 0, 85, 4),
                                                   (960, 720, 0, 85, 120)])]
    jpegs.append(read_golden("parser", "mjpeg.jpg"))
    return jpegs


@pytest.mark.parametrize("mode", [1, 2])
def test_batch_fed_from_pinned_bytes_reads_headers_only(ca, gpu, mode):
    """compeg_batch_upload_jpegs with device preprocessing and the JPEG bytes in page-locked memory
    (compeg_host_alloc / compeg_host_register): the host parses headers, the segments go up from where they lie --
    the borrowed-bytes road of src/lib.rs:577-595,397-407 -- and the scan kernels check what the skipped walk
    would have found.  Same pixels as the oracle; pageable bytes and a mix of both take the staged road."""
    jpegs = _jpeg_set()
    wants = [orc.ImageData(j).decode() for j in jpegs]
    pinned = ca.HostBuffer(sum(len(j) + 64 for j in jpegs))
    views = pinned.place(jpegs)
    registered = np.frombuffer(bytearray(b"".join(jpegs)), dtype=np.uint8).copy()   # a caller-owned buffer, page-locked in place
    ca.host_register(registered)
    offs = np.cumsum([0] + [len(j) for j in jpegs])
    reg_views = [registered[offs[i]:offs[i + 1]] for i in range(len(jpegs))]
    try:
        for name, src in (("pinned", views), ("registered", reg_views), ("pageable", jpegs),
                          ("mixed", [v if i % 2 else j for i, (v, j) in enumerate(zip(views, jpegs))])):
            batch = ca.Batch(gpu)
            batch.set_device_preprocess(mode)
            for _ in range(2):                       # (the second upload reuses arenas and the output)
                batch.upload_jpegs(src, host_threads=4)
                assert batch.host_fallbacks() == 0, name
                batch.decode()
                batch.wait()
                for i, want in enumerate(wants):
                    assert np.array_equal(batch.read_output(i), want), (name, i)
    finally:
        ca.host_unregister(registered)
        pinned.close()


def test_deferred_segment_end_is_checked_on_the_device(ca, gpu):
    """The copy-free road takes a segment to run up to the file's final EOI.  Files where the reference's parser ends
    it earlier (src/file.rs:163-201: any marker but RSTn) -- here: bytes behind the real EOI that themselves end in
    FF D9 -- are caught by the scan kernels and parsed again in full; files the front-end rejects fail the call
    with the reference's message."""
    good = synth.make_jpeg(640, 360, seed=7, quality=85, ri=4)
    tricky = good + b"trailing bytes \x12\x34" + b"\xff\xd9"
    want_good, want_tricky = orc.ImageData(good).decode(), orc.ImageData(tricky).decode()
    assert np.array_equal(want_good, want_tricky)          # (the reference stops at the first EOI)
    pinned = ca.HostBuffer(4 * len(tricky) + 1024)
    views = pinned.place([good, tricky, tricky, good])
    try:
        batch = ca.Batch(gpu)
        batch.set_device_preprocess(1)
        batch.upload_jpegs(views, host_threads=2)
        assert batch.host_fallbacks() == 2
        batch.decode()
        batch.wait()
        for i in range(4):
            assert np.array_equal(batch.read_output(i), want_good), i
        bad = bytearray(good)
        bad[good.find(b"\xff\xc0") + 1] = 0xC2                  # SOF2: rejected like ImageData::new does
        with pytest.raises(ca.Error) as e:
            batch.upload_jpegs(pinned.place([good, bytes(bad)]), host_threads=2)
        assert str(e.value).startswith("image 1: ")
        with pytest.raises(orc.OracleError) as eo:
            orc.ImageData(bytes(bad))
        assert str(e.value) == "image 1: " + str(eo.value)
    finally:
        pinned.close()


def test_upload_in_two_steps_keeps_the_link_busy(ca, gpu):
    """compeg_batch_upload_jpegs_begin / _end: the second batch's transfers are queued while the first one's arrive;
    decode() ends a begun upload by itself; a new upload on a batch ends the begun one first.  Same pixels."""
    jpegs = _jpeg_set()
    wants = [orc.ImageData(j).decode() for j in jpegs]
    pinned = ca.HostBuffer(2 * sum(len(j) + 64 for j in jpegs))
    views_a = ca.JpegList(pinned.place(jpegs + jpegs[::-1])[:len(jpegs)])
    views_b = ca.JpegList(pinned.place(jpegs + jpegs[::-1])[len(jpegs):])
    g2 = ca.Gpu.open(0)
    a, b = ca.Batch(gpu), ca.Batch(g2)
    try:
        for batch in (a, b):
            batch.set_device_preprocess(1)
        for rep in range(3):
            a.upload_jpegs_begin(views_a, host_threads=4)
            b.upload_jpegs_begin(views_b, host_threads=4)         # queued behind a's
            a.upload_end()
            a.decode()
            b.decode()                                            # (ends b's upload itself)
            a.wait()
            b.wait()
            for i, want in enumerate(wants):
                assert np.array_equal(a.read_output(i), want), (rep, i)
                assert np.array_equal(b.read_output(len(jpegs) - 1 - i), want), (rep, i)
        a.upload_jpegs_begin(views_a, host_threads=2)
        a.upload_jpegs(views_b, host_threads=2)                   # a whole new upload: the begun one is ended first
        a.decode()
        a.wait()
        for i, want in enumerate(wants):
            assert np.array_equal(a.read_output(len(jpegs) - 1 - i), want), i
        # pageable bytes and host preprocessing: _begin does the whole upload, _end has nothing left to do
        c = ca.Batch(gpu)
        c.upload_jpegs_begin(jpegs, host_threads=2)
        c.upload_end()
        c.decode()
        c.wait()
        assert np.array_equal(c.read_output(0), wants[0])
    finally:
        del a, b
        pinned.close()


@pytest.mark.parametrize("mode", [1, 2])
def test_device_preprocessing_of_extension_layouts(ca, gpu, mode):
    """Rows f1 x f3: batches of 4:4:4 / 4:4:0 / 4:2:0 images with the scan kernels' preprocessing -- parsed images
    (compeg_batch_upload), JPEG bytes in pageable and page-locked memory (headers only on the host) -- through the
    layouts' fused kernels, their streamed form, and, layouts mixed with 4:2:2, the two-kernel route.  Same pixels as
    the oracle with the same extension; stuffed bytes and ragged edges among the images."""
    for sampling in ((1, 1), (1, 2), (2, 2)):
        cases = [(640, 360, 0, 85, 4, 311), (250, 70, 1, 75, 3, 312), (33, 17, 2, 85, 1, 313), (1000, 600, 1, 95, 8, 314)]
        jpegs = [synth.make_jpeg(w, h, seed=s, kind=k, quality=q, ri=ri, sampling=sampling) for (w, h, k, q, ri, s) in cases]
        wants = [orc.ImageData(j, allow_sampling=True).decode() for j in jpegs]
        same = [jpegs[i % 4] for i in range(40)]
        batch = ca.Batch(gpu)
        batch.set_device_preprocess(mode)
        batch.upload([ca.ImageData(j, allow_sampling=True) for j in same])
        assert batch.host_fallbacks() == 0
        batch.decode()
        batch.wait()
        # (the scan kernels report window spans less exactly than the host: the dense image may ask for streamed windows)
        assert batch.last_kernel() in ("fused_layout", "fused_stream")
        for i in range(40):
            assert np.array_equal(batch.read_output(i), wants[i % 4]), (sampling, i)
        pinned = ca.HostBuffer(sum(len(j) + 64 for j in same))
        try:
            for name, src in (("pageable", same), ("pinned", pinned.place(same))):
                batch.upload_jpegs(src, host_threads=4, allow_sampling=True)
                assert batch.host_fallbacks() == 0, name
                batch.decode()
                batch.wait()
                assert batch.last_kernel() in ("fused_layout", "fused_stream")
                for i in range(40):
                    assert np.array_equal(batch.read_output(i), wants[i % 4]), (sampling, name, i)
        finally:
            pinned.close()
        # long restart intervals: the layout's streamed form
        frames = [synth.make_jpeg(960, 720, seed=720 + i, kind=0, quality=85, ri=16, sampling=sampling) for i in range(8)]
        uniform = [frames[i % 8] for i in range(240)]
        batch.upload_jpegs(uniform, host_threads=8, allow_sampling=True)
        batch.decode()
        batch.wait()
        assert batch.last_kernel() == "fused_stream"
        for i in (0, 1, 7, 100, 239):
            assert np.array_equal(batch.read_output(i), orc.ImageData(uniform[i], allow_sampling=True).decode()), (sampling, i)
        # layouts mixed in one batch
        mixed = jpegs[:3] + [synth.make_jpeg(320, 240, seed=315, ri=2)]
        batch.upload_jpegs(mixed, host_threads=2, allow_sampling=True)
        batch.decode()
        batch.wait()
        assert batch.last_kernel() == "generic"
        for i, j in enumerate(mixed):
            assert np.array_equal(batch.read_output(i), orc.ImageData(j, allow_sampling=True).decode()), (sampling, i)
    # without the flag the front-end rejects the layout, as the reference does (src/lib.rs:650)
    with pytest.raises(ca.Error):
        batch.upload_jpegs([synth.make_jpeg(64, 64, seed=1, ri=2, sampling=(2, 2))], host_threads=1)
