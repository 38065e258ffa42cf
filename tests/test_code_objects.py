"""What the compiler made of the decode kernels (no GPU needed): their LDS traffic has to be ds_* instructions and
their registers must not spill.  An integer round trip of an LDS pointer is enough for the compiler to forget the
address space: every access behind it then becomes a flat one -- correct, several times slower, and invisible in
the results (round 2 lost 3 us of a 45 us launch that way until a profile showed the scratch and flat counts)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "compeg_amd", "libcompeg_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
KERNELS = ("decode_fused_422_kernel", "decode_fused_422_mcu_kernel", "decode_fused_422_mcu_rec_kernel", "walk_mcus_422_kernel", "decode_fused_422_stream_kernel", "decode_fused_444_stream_kernel", "decode_fused_440_stream_kernel", "decode_fused_420_stream_kernel", "decode_fused_444_kernel", "decode_fused_440_kernel", "decode_fused_420_kernel",
           "decode_fused_444_single_kernel", "decode_fused_440_single_kernel",
           "decode_pair_422_kernel", "decode_coop_team_422_kernel",
           "entropy_kernel", "idct_composite_kernel")


def _code_objects(tmp_path):
    if not (os.path.exists(LIB) and os.path.exists(os.path.join(LLVM, "llvm-objdump"))):
        pytest.skip("library or llvm-objdump not here")
    lib = shutil.copy(LIB, tmp_path / "lib.so")
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", lib], check=True, capture_output=True, cwd=tmp_path)
    return [str(p) for p in tmp_path.iterdir() if "gfx950" in p.name]


def test_decode_kernels_use_lds_instructions_and_no_scratch(tmp_path):
    seen = set()
    for co in _code_objects(tmp_path):
        asm = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], check=True, capture_output=True, text=True).stdout
        # per symbol: the text between "<name>:" labels
        for m in re.finditer(r"^[0-9a-f]+ <(\S+)>:\n(.*?)(?=^[0-9a-f]+ <\S+>:|\Z)", asm, re.S | re.M):
            name, body = m.group(1), m.group(2)
            kernel = next((k for k in KERNELS if re.search(r"\d+" + k + "E", name)), None)
            if not kernel:
                continue
            seen.add(kernel)
            flat = len(re.findall(r"\bflat_(load|store|atomic)", body))
            scratch = len(re.findall(r"\bscratch_(load|store)", body))
            assert flat == 0, f"{kernel}: {flat} flat memory instructions (an LDS or global pointer lost its address space)"
            assert scratch == 0, f"{kernel}: {scratch} scratch instructions (register spills)"
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
        for block in notes.split("- .agpr_count")[1:]:
            name = re.search(r"\.name:\s+(\S+)", block)
            size = re.search(r"\.private_segment_fixed_size:\s+(\d+)", block)
            if name and size and any(re.search(r"\d+" + k + "E", name.group(1)) for k in KERNELS):
                assert int(size.group(1)) == 0, f"{name.group(1)}: {size.group(1)} bytes of private memory per lane"
    assert seen == set(KERNELS), f"kernels not found in the code objects: {set(KERNELS) - seen}"


def test_no_lane_exchange_under_a_narrowed_exec_mask(tmp_path):
    """`cond ? quad_lane<j>(v) : 0` -- the compiler is free to run the DPP move under the EXEC mask of the lanes where
    `cond` holds; the lanes it reads from, off there, read as 0 (round 4: the pairs' second halves, selected by
    lane & 2, lost every offset that came from their quad's lanes 0 and 1 -- on the GPU only, the emulator has no EXEC).
    The kernels exchange in every lane and mask the value; here: no DPP instruction between a saveexec and the
    instruction that restores EXEC, in any kernel of the library."""
    for co in _code_objects(tmp_path):
        asm = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
        narrowed, found = None, []
        for line in asm.splitlines():
            if "saveexec" in line:
                narrowed = line.strip()
            elif re.search(r"s_or_b64 exec, exec|s_mov_b64 exec|s_endpgm|^[0-9a-f]+ <\S+>:", line):
                narrowed = None
            elif narrowed and re.search(r"_dpp\b|ds_swizzle|ds_bpermute|ds_permute|v_permlane", line):
                found.append(line.strip())
        assert not found, f"lane exchanges under a narrowed EXEC mask: {found[:4]}"


def test_shipped_library_has_no_laboratory_switches():
    """The experiment knobs (compeg_amd/csrc/lab.h) exist in the laboratory build only: the shipped library's
    strings name no environment variable but the four it documents, and the knock-out arms of the kernel bodies
    (CG_EXP) are compiled out of it -- its code objects equal those of a build that never heard of CG_EXP."""
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    names = set(re.findall(rb"COMPEG_[A-Z_0-9]+", open(LIB, "rb").read()))
    allowed = {b"COMPEG_TRACE", b"COMPEG_TRACE_BATCH", b"COMPEG_VERBOSE", b"COMPEG_SCAN_THREADS"}
    # (error-code and macro names of the header may appear in messages: they are not environment variables)
    env_like = {n for n in names if not n.startswith((b"COMPEG_E_", b"COMPEG_OK", b"COMPEG_PARSE_", b"COMPEG_KERNEL_", b"COMPEG_HIP_H",
                                                      b"COMPEG_METADATA", b"COMPEG_HUFFMAN"))}
    assert env_like <= allowed, f"experiment switches in the shipped library: {sorted(env_like - allowed)}"
    lab = os.path.join(os.path.dirname(LIB), "libcompeg_hip_lab.so")
    if os.path.exists(lab):
        lab_names = set(re.findall(rb"COMPEG_[A-Z_0-9]+", open(lab, "rb").read()))
        assert {b"COMPEG_PIPELINE", b"COMPEG_COOP", b"COMPEG_RESIDENT"} <= lab_names   # (the check above can see such names)
    src = open(os.path.join(ROOT, "compeg_amd", "csrc", "kernels_body.h")).read()
    assert "#if !defined(COMPEG_LAB)\n#undef CG_EXP" in src   # release builds: every knock-out arm off, whatever the flags
