"""CPU-side tests of the product's host logic (no GPU): the C ABI surface, the JPEG front-end,
the Huffman LUT builder and the scan preprocessor of libcompeg_hip.so against the oracle."""
import os
import re

import numpy as np
import pytest

import compeg_amd as ca
from conftest import GOLDEN, ROOT, read_golden
from oracle import oracle as orc
from tools import synth


# ---- C ABI surface ---------------------------------------------------------------------------

def _declared_functions():
    text = open(os.path.join(ROOT, "include", "compeg_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(compeg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import ctypes
    lib = ctypes.CDLL(ca.LIB_PATH)
    names = _declared_functions()
    assert len(names) >= 45
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_python_mirror_binds_every_declared_symbol():
    from compeg_amd import _lib
    assert sorted(_lib.lib._signatures) == _declared_functions()


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ca.Error) as e:
        ca.Gpu.open()
    assert e.value.code == ca.E_HIP and "no CPU fallback" in str(e.value)


def test_product_never_loads_the_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "compeg_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")) or f == "Makefile":
                assert "oracle" not in open(os.path.join(root, f)).read().lower().replace(
                    "cpu oracle", "").replace("the oracle", ""), f
    import subprocess
    out = subprocess.run(["ldd", ca.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out


# ---- ImageData parity ----------------------------------------------------------------------

def _same_image(jpeg):
    """Both front-ends must agree: same bytes when accepted, same message when rejected
    ("panic:" marks inputs on which the reference aborts; we return an error instead)."""
    try:
        want = orc.ImageData(jpeg)
    except orc.OracleError as oe:
        with pytest.raises(ca.Error) as pe:
            ca.ImageData(jpeg)
        if not str(oe).startswith("panic:"):
            assert str(pe.value) == str(oe)
        return False
    got = ca.ImageData(jpeg)
    assert (got.width(), got.height(), got.parallelism()) == \
        (want.width(), want.height(), want.parallelism())
    assert got.metadata() == want.metadata()
    assert got.huffman_l1() == want.l1()
    assert got.huffman_l2() == want.l2()
    assert got.scan_range() == want.scan_range()
    return True


def _golden_jpegs():
    out = []
    for sub in ("parser", "refs"):
        d = os.path.join(GOLDEN, sub)
        out += [(sub, f) for f in sorted(os.listdir(d)) if f.endswith(".jpg")]
    return out


@pytest.mark.parametrize("sub,name", _golden_jpegs())
def test_front_end_on_reference_fixtures(sub, name):
    _same_image(read_golden(sub, name))


def test_front_end_error_codes():
    with pytest.raises(ca.Error) as e:
        ca.ImageData(read_golden("parser", "progressive3.jpg"))
    assert e.value.code == ca.E_UNSUPPORTED
    with pytest.raises(ca.Error) as e:
        ca.ImageData(b"\xff\xd8\xff")
    assert e.value.code == ca.E_MALFORMED
    assert str(e.value) == "reached end of data while decoding JPEG stream"


def test_front_end_on_synthetic_variants():
    for flags in (0, synth.NO_DHT, synth.JFIF, synth.NO_EOI):
        for sampling in ((2, 1), (1, 1), (2, 2)):
            for ri in (0, 1, 4, 7):
                _same_image(synth.make_jpeg(72, 40, seed=ri, sampling=sampling, ri=ri, flags=flags))


def test_front_end_fuzz_mutations():
    rng = np.random.default_rng(1234)
    seeds = [synth.make_jpeg(64, 32, seed=1, ri=2), synth.make_jpeg(48, 16, seed=2, ri=1, flags=synth.NO_DHT),
             read_golden("refs", "64x8-Ri-2.jpg"), read_golden("parser", "restarts.jpg")]
    accepted = 0
    for it in range(3000):
        base = bytearray(seeds[it % len(seeds)])
        header_end = base.find(b"\xff\xda") + 14
        for _ in range(int(rng.integers(1, 4))):
            kind = rng.integers(0, 4)
            if len(base) < 4:
                break
            hdr = max(3, min(header_end, len(base)))
            if kind == 0:      # flip a header byte
                base[int(rng.integers(2, hdr))] = int(rng.integers(0, 256))
            elif kind == 1:    # flip any byte
                base[int(rng.integers(0, len(base)))] = int(rng.integers(0, 256))
            elif kind == 2:    # truncate
                del base[int(rng.integers(3, len(base))):]
            else:              # insert a byte in the header
                base.insert(int(rng.integers(2, hdr)), int(rng.integers(0, 256)))
        accepted += bool(_same_image(bytes(base)))
    assert accepted > 100  # the corpus exercises both outcomes


def test_huffman_tables_fuzz():
    """Random code-length histograms through a DHT: valid prefix codes give identical LUTs,
    over-subscribed ones are rejected by both."""
    rng = np.random.default_rng(99)
    base = synth.make_jpeg(32, 16, seed=3, ri=1)
    i = base.find(b"\xff\xc4")
    ok = bad = 0
    for it in range(400):
        counts = np.zeros(16, dtype=np.uint8)
        budget = 1.0
        overfull_at = int(rng.integers(1, 17)) if it % 3 == 0 else 0
        for length in range(1, 17):
            room = int(budget * (1 << length) + 1e-9)
            n = 0
            if length == overfull_at and room < 255:
                n = room + 1 + int(rng.integers(0, 2))   # one code too many: not a prefix code
            elif room > 0 and rng.random() < 0.6:
                n = int(rng.integers(0, min(room, 40) + 1))
            counts[length - 1] = min(n, 255)
            budget -= min(n, 255) / (1 << length)
            if budget < 0:
                break
        nsym = int(counts.sum())
        if nsym == 0:
            continue
        syms = rng.integers(0, 256, nsym, dtype=np.uint8)
        tcth = int(rng.choice([0x00, 0x10, 0x01, 0x11]))
        seg = bytes([0xFF, 0xC4]) + (2 + 17 + nsym).to_bytes(2, "big") + bytes([tcth]) + counts.tobytes() + syms.tobytes()
        jpeg = base[:i] + seg + base[i:]
        if _same_image(jpeg):
            ok += 1
        else:
            bad += 1
    assert ok > 50 and bad > 5


# ---- ScanBuffer parity ---------------------------------------------------------------------

def _scan_both(data, expected):
    a, b = ca.ScanBuffer(), orc.ScanBuffer()
    ea = eb = None
    try:
        a.process(data, expected)
    except ca.Error as e:
        ea = str(e)
        assert e.code == ca.E_COUNT_MISMATCH
    try:
        b.process(data, expected)
    except orc.OracleError as e:
        eb = str(e)
    assert ea == eb
    assert a.processed_scan_data() == b.processed_scan_data()
    assert a.start_positions() == b.start_positions()
    return a


def test_scanbuffer_reference_kats():
    # ref src/scan.rs:151-180
    sb = _scan_both(bytes([0x12, 0x34, 0x56, 0x78]), 1)
    assert sb.processed_scan_data() == bytes([0x12, 0x34, 0x56, 0x78])
    sb = _scan_both(bytes([0xFF, 0xD0, 0xFF, 0xD0]), 3)
    assert sb.processed_scan_data() == b"" and sb.start_positions() == bytes(12)
    sb = _scan_both(bytes([0xFF, 0x00, 0x44, 0x55, 0xFF, 0xD0, 0x34]), 2)
    assert sb.processed_scan_data() == bytes([0xFF, 0x44, 0x55, 0x00, 0x34, 0, 0, 0])
    sb = _scan_both(bytes([0x11, 0xFF, 0xD0, 0x11, 0xFF, 0xD0, 0x11]), 3)
    assert sb.processed_scan_data() == bytes([0x11, 0, 0, 0, 0x11, 0, 0, 0, 0x11, 0, 0, 0])
    with pytest.raises(ca.Error) as e:
        ca.ScanBuffer().process(bytes([0x11, 0xFF, 0xD0, 0x11, 0xFF, 0xD0, 0x11]), 1)
    assert str(e.value) == "restart interval count mismatch: counted 3, expected 1"


def test_scanbuffer_bench_data():
    data = read_golden("scan", "scan.dat")   # ref benches/bench.rs:9-19
    sb = _scan_both(data, 42876)
    assert len(sb.start_positions()) == 42876 * 4
    _scan_both(data, 42875)                  # mismatch: identical truncated buffers too
    _scan_both(data, 50000)


def test_scanbuffer_fuzz():
    rng = np.random.default_rng(7)
    for it in range(400):
        n = int(rng.integers(0, 300))
        data = rng.integers(0, 256, n, dtype=np.uint8)
        mask = rng.random(n) < 0.15           # plenty of FFs, incl. FF FF and a trailing FF
        data[mask] = 0xFF
        zero = rng.random(n) < 0.1
        data[zero] = 0
        expected = int(rng.integers(0, 12))
        _scan_both(data.tobytes(), expected)
    # long plain runs (the vector copy of the host loop): sparse FFs, FFs on and around every
    # 16/32-byte boundary, FF pairs cut by the end of the data
    for it in range(120):
        n = int(rng.integers(1, 4000))
        data = rng.integers(0, 255, n, dtype=np.uint8)       # no FF yet
        for pos in rng.integers(0, n, int(rng.integers(0, 12))):
            data[pos] = 0xFF
            if pos + 1 < n and rng.random() < 0.5:
                data[pos + 1] = 0
        if it % 3 == 0:
            edge = int(rng.integers(1, max(2, n // 16))) * 16 + int(rng.integers(-2, 2))
            if 0 <= edge < n:
                data[edge] = 0xFF
        if it % 5 == 0:
            data[n - 1] = 0xFF
        if it % 7 == 0 and n > 2:
            data[n - 2] = 0xFF
        _scan_both(data.tobytes(), int(rng.integers(0, 14)))
    _scan_both(b"", 0)
    _scan_both(b"\xff", 1)
    _scan_both(b"\xff\xff\xff", 2)


def test_scanbuffer_with_threads_gives_the_same_buffers():
    """Extension: several threads share one process() call (pieces preprocessed privately, then put in place)."""
    rng = np.random.default_rng(23)
    cases = []
    for it in range(10):
        n = int(rng.integers(200_000, 700_000))
        data = rng.integers(0, 256, n, dtype=np.uint8)
        if it % 2:
            data[rng.random(n) < 0.02] = 0xFF                 # FF runs, markers every few dozen bytes
        data[rng.random(n) < 0.01] = 0
        for cut in (n // 2, n // 3, 2 * (n // 3), n // 4, 3 * (n // 4)):   # FF pairs and FF runs across the cuts
            k = int(rng.integers(0, 6))
            data[max(0, cut - k):cut + int(rng.integers(0, 6))] = 0xFF
        cases.append((data.tobytes(), int(rng.integers(1, 40000))))
    cases.append((read_golden("scan", "scan.dat"), 42876))     # ref benches/bench.rs:9-19
    cases.append((bytes([0x55]) * 300_000, 1))                  # no marker at all: every piece is all head
    cases.append((bytes([0xFF, 0xD0]) * 150_000, 150_001))      # nothing but markers
    for threads in (2, 3, 4):
        sb = ca.ScanBuffer()
        sb.set_threads(threads)
        for data, expected in cases:
            ref = orc.ScanBuffer()
            try:
                ref.process(data, expected)
            except orc.OracleError:
                pass
            try:
                sb.process(data, expected)
            except ca.Error as e:
                assert e.code == ca.E_COUNT_MISMATCH
            assert sb.processed_scan_data() == ref.processed_scan_data(), (threads, len(data))
            assert sb.start_positions() == ref.start_positions(), (threads, len(data))
        sb.process(bytes([1, 2, 0xFF, 0xD0, 3]), 2)             # small segments stay on the calling thread
        assert sb.processed_scan_data() == bytes([1, 2, 0, 0, 3, 0, 0, 0])


def test_scanbuffer_thread_count_is_validated_and_can_change():
    sb = ca.ScanBuffer()
    for bad in (0, 17, 1000):
        with pytest.raises(ca.Error) as e:
            sb.set_threads(bad)
        assert e.value.code == ca.E_INVALID_ARG
    data = read_golden("scan", "scan.dat")
    ref = orc.ScanBuffer()
    ref.process(data, 42876)
    for threads in (3, 1, 2, 2):               # helpers are replaced / dropped / kept
        sb.set_threads(threads)
        sb.process(data, 42876)
        assert sb.processed_scan_data() == ref.processed_scan_data()
        assert sb.start_positions() == ref.start_positions()


def test_scanbuffer_reuse_gives_fresh_buffer_output():
    """Quirk Q7: we always produce what the reference produces on a fresh buffer."""
    sb = ca.ScanBuffer()
    sb.process(bytes([0xAA] * 64), 1)
    sb.process(bytes([0x11, 0xFF, 0xD0, 0x22]), 2)
    assert sb.processed_scan_data() == bytes([0x11, 0, 0, 0, 0x22, 0, 0, 0])


def _c_prototypes(header):
    """name -> (return type, [parameter types]) of every function include/compeg_hip.h declares, C spelling
    normalised (no parameter names, single spaces, '*' detached)."""
    import re
    text = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    text = re.sub(r'extern\s+"C"\s*\{', "", text)
    text = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", "", text, flags=re.S)   # struct bodies are checked apart
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(compeg_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        ret, name, params = m.group(1), m.group(2), m.group(3)

        def norm(t, is_param):
            t = re.sub(r"(\w+)\s*\[\s*\d*\s*\]\s*$", r"* \1", t.strip())   # an array parameter is a pointer
            t = re.sub(r"\s+", " ", t.replace("*", " * ")).strip()
            if is_param:   # drop the parameter's name: the last identifier, unless it is the type itself
                toks = t.split(" ")
                if len(toks) > 1 and re.fullmatch(r"[A-Za-z_]\w*", toks[-1]) and toks[-1] not in ("int", "char", "void"):
                    toks = toks[:-1]
                t = " ".join(toks)
            return t
        ps = [norm(x, True) for x in params.split(",")] if params.strip() not in ("", "void") else []
        protos[name] = (norm(ret, False), ps)
    return protos


_SCALARS = {"int": "c_int", "unsigned": "c_uint", "unsigned int": "c_uint", "uint32_t": "u32", "uint8_t": "u8",
            "uint16_t": "u16", "uint64_t": "u64", "int32_t": "i32", "int16_t": "i16", "int64_t": "i64", "size_t": "usize", "double": "c_double", "char": "c_char",
            "void": "c_void"}


def _rust_type_of(c_type):
    """The Rust FFI type a C type (as _c_prototypes spells it) must be bound with."""
    toks = c_type.split(" ")
    # split into base (with its const) and the pointer levels, each with an optional const behind the star
    base, levels, i = [], [], 0
    while i < len(toks) and toks[i] != "*":
        base.append(toks[i])
        i += 1
    while i < len(toks):
        assert toks[i] == "*", c_type
        const_ptr = i + 1 < len(toks) and toks[i + 1] == "const"
        levels.append(const_ptr)
        i += 2 if const_ptr else 1
    base_const = "const" in base
    name = " ".join(t for t in base if t != "const")
    rust = _SCALARS.get(name, name)            # opaque handle types keep their name
    pointee_const = base_const
    for const_ptr in levels:
        rust = ("*const " if pointee_const else "*mut ") + rust
        pointee_const = const_ptr
    return rust


def _rust_prototypes(ffi):
    import re
    protos = {}
    for m in re.finditer(r"pub fn (compeg_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", ffi, flags=re.S):
        params = [re.sub(r"\s+", " ", x.split(":", 1)[1]).strip() for x in m.group(2).split(",") if ":" in x]
        protos[m.group(1)] = ((m.group(3) or "()").strip(), params)
    return protos


def test_rust_binding_matches_the_header_signature_by_signature():
    """integration/rust/compeg-hip/src/ffi.rs (the reference-side binding; no Rust toolchain
    here to compile it) against include/compeg_hip.h: the same entry points, and for each one the same arity,
    every parameter's C type bound with the Rust FFI type it has to be bound with (integer width and
    signedness, pointer depth, const-ness at every level, opaque handle names), the same return type; the
    same constants; the plain-data struct field by field."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "compeg_hip.h")).read()
    ffi = open(os.path.join(root, "integration", "rust", "compeg-hip", "src", "ffi.rs")).read()
    c, r = _c_prototypes(header), _rust_prototypes(ffi)
    assert len(c) >= 50 and set(c) == set(r), (sorted(set(c) - set(r)), sorted(set(r) - set(c)))
    for name, (ret, params) in c.items():
        rret, rparams = r[name]
        assert len(params) == len(rparams), (name, params, rparams)
        for i, (ct, rt) in enumerate(zip(params, rparams)):
            assert _rust_type_of(ct) == rt, f"{name}: parameter {i} is `{ct}` in C, `{rt}` in Rust (expected `{_rust_type_of(ct)}`)"
        want_ret = "()" if ret == "void" else _rust_type_of(ret)
        assert want_ret == rret, f"{name}: returns `{ret}` in C, `{rret}` in Rust"
    # constants
    c_consts = {k: int(v, 0) for k, v in re.findall(r"#define\s+(COMPEG_[A-Z0-9_]+)\s+\(?(-?(?:0x)?[0-9a-fA-F]+)u?\)?", header)}
    r_consts = {k: int(v, 0) for k, v in re.findall(r"pub const (COMPEG_[A-Z0-9_]+):\s*\w+\s*=\s*(-?(?:0x)?[0-9a-fA-F]+)\s*;", ffi)}
    for k in ("COMPEG_OK", "COMPEG_E_INVALID_ARG", "COMPEG_E_UNSUPPORTED", "COMPEG_E_MALFORMED", "COMPEG_E_COUNT_MISMATCH",
              "COMPEG_E_HIP", "COMPEG_PARSE_ANY_LUMA_SAMPLING", "COMPEG_PARSE_STANDARD_ENTROPY"):
        assert k in c_consts and r_consts.get(k) == c_consts[k], (k, c_consts.get(k), r_consts.get(k))
    # the one plain-data struct
    body = re.search(r"typedef\s+struct\s+compeg_stage_times\s*\{(.*?)\}", re.sub(r"/\*.*?\*/", "", header, flags=re.S), flags=re.S).group(1)
    c_fields = re.findall(r"(\w+)\s+(\w+)\s*;", body)
    r_body = re.search(r"#\[repr\(C\)\][^{]*pub struct compeg_stage_times\s*\{(.*?)\}", ffi, flags=re.S).group(1)
    r_fields = re.findall(r"pub (\w+):\s*(\w+)", r_body)
    assert [(n, _SCALARS[t]) for t, n in c_fields] == r_fields, (c_fields, r_fields)


def test_type_mapping_of_the_signature_check_itself():
    assert _rust_type_of("const compeg_image * const *") == "*const *const compeg_image"
    assert _rust_type_of("compeg_gpu * *") == "*mut *mut compeg_gpu"
    assert _rust_type_of("const char *") == "*const c_char"
    assert _rust_type_of("void *") == "*mut c_void"
    assert _rust_type_of("uint32_t") == "u32" and _rust_type_of("size_t *") == "*mut usize"


def test_a_plain_c_consumer_of_the_header(tmp_path):
    """A C99 program (tests/c_consumer/consumer.c) compiled with gcc -Wall -Wextra -Werror against
    include/compeg_hip.h and linked with libcompeg_hip.so -- a non-Python FFI user of the boundary, built on
    every run: parses a reference fixture (same numbers as the Python binding), gets the reference's error
    texts through status code + compeg_last_error, runs the ScanBuffer known-answer vector (src/scan.rs:151-159)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "consumer")
    libdir = os.path.dirname(ca.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "c_consumer", "consumer.c"), "-o", exe,
                           "-L", libdir, "-l:libcompeg_hip.so", "-Wl,-rpath," + libdir])
    fixture = os.path.join(root, "tests", "golden", "parser", "mjpeg.jpg")
    r = subprocess.run([exe, fixture], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = dict(l.split(" ", 1) for l in r.stdout.splitlines())
    img = ca.ImageData(open(fixture, "rb").read())
    assert lines["version"] == ca.version()
    assert lines["image"] == f"{img.width()} {img.height()} {img.parallelism()}" == "960 720 540"
    off, length = img.scan_range()
    assert lines["scan"] == f"{off} {length} l2 {len(img.huffman_l2())}"
    acc = 0
    for b in img.metadata():
        acc = (acc * 131 + b) & 0xffffffffffffffff
    assert int(lines["metadata"]) == acc & 0xffffffff
    E_MALFORMED, E_INVALID_ARG, E_COUNT_MISMATCH = ca.E_MALFORMED, ca.E_INVALID_ARG, ca.E_COUNT_MISMATCH
    assert lines["junk"] == f"{E_MALFORMED} JPEG image does not start with SOI marker"
    assert lines["null"] == str(E_INVALID_ARG)
    assert lines["scanbuffer"] == "ff 44 55 00 34 00 00 00 | 00 00 00 00 01 00 00 00"
    assert lines["mismatch"] == f"{E_COUNT_MISMATCH} restart interval count mismatch: counted 2, expected 1"
    assert lines["stage_times"] == f"24 {E_INVALID_ARG}"


def test_sampling_extension_front_end():
    """COMPEG_PARSE_ANY_LUMA_SAMPLING: without the flag the reference's rejection and
    message; with it the same metadata block / LUTs as the oracle with its own extension switched on.
    Luma samplings outside 1..2 x 1..2 and sub-sampled-the-other-way chroma stay rejected."""
    for sampling in ((1, 1), (1, 2), (2, 2)):
        jpeg = synth.make_jpeg(100, 52, seed=9, sampling=sampling, ri=3)
        with pytest.raises(ca.Error) as e:
            ca.ImageData(jpeg)
        assert str(e.value) == f"invalid sampling factors {sampling[0]}x{sampling[1]} for Y component (expected 2x1)"
        got, want = ca.ImageData(jpeg, allow_sampling=True), orc.ImageData(jpeg, allow_sampling=True)
        assert (got.width(), got.height(), got.parallelism()) == (want.width(), want.height(), want.parallelism())
        assert got.metadata() == want.metadata()
        assert got.huffman_l1() == want.l1() and got.huffman_l2() == want.l2()
    jpeg = bytearray(synth.make_jpeg(64, 32, seed=1, sampling=(2, 1), ri=2))
    sof = jpeg.find(b"\xff\xc0")
    jpeg[sof + 11] = 0x41                      # Y sampled 4x1
    for image_data in (ca.ImageData, orc.ImageData):
        with pytest.raises((ca.Error, orc.OracleError)) as e:
            image_data(bytes(jpeg), allow_sampling=True)
        assert "invalid sampling factors 4x1 for Y component" in str(e.value)
    jpeg[sof + 11] = 0x22
    jpeg[sof + 14] = 0x21                      # Cb sampled 2x1
    for image_data in (ca.ImageData, orc.ImageData):
        with pytest.raises((ca.Error, orc.OracleError)) as e:
            image_data(bytes(jpeg), allow_sampling=True)
        assert "invalid U/V sampling factors" in str(e.value)
