"""Static guard on the shipped device code (VERDICT / ADVICE round 2): kernels built WITH packed-f32 VALU instructions
(v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 / v_pk_mov_b32) were not bitwise repeatable at two waves per SIMD
(DESIGN.md section 9).  The Makefile disables the target feature; this test disassembles every gfx950 code object inside
libphnn_mpc.so and asserts that none of those opcodes is present -- so a toolchain bump, a dropped flag or a new
translation unit cannot re-introduce them silently.  It also pins the matrix instructions the design rests on.
"""
import os
import re
import struct
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "phnn_mpc_amd", "csrc", "libphnn_mpc.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(path):
    """gfx950 ELF images inside the library's .hip_fatbin section (one clang offload bundle per translation unit)."""
    blob = open(path, "rb").read()
    out, pos = [], blob.find(MAGIC)
    while pos >= 0:
        n = struct.unpack_from("<Q", blob, pos + len(MAGIC))[0]
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size > 0:
                out.append(blob[pos + off:pos + off + size])
        pos = blob.find(MAGIC, pos + len(MAGIC))
    return out


@pytest.fixture(scope="module")
def opcodes():
    if not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        pytest.skip("llvm-objdump not available")
    objs = _code_objects(LIB)
    assert len(objs) >= 4, f"expected the four translation units' gfx950 code objects, found {len(objs)}"
    counts = {}
    with tempfile.TemporaryDirectory() as tmp:
        for k, img in enumerate(objs):
            f = os.path.join(tmp, f"co{k}.elf")
            open(f, "wb").write(img)
            txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", f], capture_output=True,
                                 text=True, check=True).stdout
            for m in re.finditer(r"^\s+([sv]_[a-z0-9_]+|ds_[a-z0-9_]+|global_[a-z0-9_]+|scratch_[a-z0-9_]+|buffer_[a-z0-9_]+)\b", txt, re.M):
                counts[m.group(1)] = counts.get(m.group(1), 0) + 1
    return counts


def test_no_packed_f32_valu_in_device_code(opcodes):
    packed = {op: n for op, n in opcodes.items() if re.match(r"v_pk_(fma|mul|add)_f32|v_pk_mov_b32", op)}
    assert not packed, f"packed-f32 VALU instructions in the shipped device code (Makefile: NOPK flag lost?): {packed}"


def test_matrix_instructions_the_design_rests_on_are_present(opcodes):
    for op in ("v_mfma_f32_16x16x32_f16", "v_mfma_f32_16x16x32_bf16", "v_mfma_f32_16x16x4_f32", "v_mfma_f32_4x4x1_16b_f32",
               "ds_read_b64_tr_b16"):
        assert opcodes.get(op, 0) > 0, f"{op} missing from the device code"
