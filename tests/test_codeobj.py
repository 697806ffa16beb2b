"""Static checks on the gfx950 code objects of the built library (tools/check_codeobj.py) -- no GPU needed.

VERDICT r03 item 1: the round-3 HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION came from a FLAT access whose 64-bit base the
compiler had decremented below the start of the private aperture (profiles/r04_aperture_fault_isa.txt).
These tests keep that pattern, and any stack under-accounting, out of what ships:
  * no register pair that is decremented by a constant addresses a flat access in any function of libmlhip.so;
  * every kernel's .private_segment_fixed_size covers its own frame plus its deepest chain of callee frames, no kernel
    uses a dynamic stack, every call target resolves inside its code object;
  * the checker itself still recognises the faulting form (an excerpt of the round-3 callee's ISA is the fixture).
"""
import gzip
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_codeobj  # noqa: E402

LIB = os.path.join(ROOT, "mathlib_amd", "libmlhip.so")


def _have_llvm():
    return os.path.exists(os.path.join(check_codeobj.LLVM_BIN, "llvm-objdump"))


@pytest.mark.skipif(not _have_llvm(), reason="llvm-objdump of the ROCm toolchain not found")
def test_shipped_code_objects_are_clean():
    if not os.path.exists(LIB):
        from mathlib_amd import build

        build.build(verbose=False)
    findings, stats = check_codeobj.check_file(LIB, jobs=4)
    assert stats["code_objects"] == 9, stats  # one per translation unit with kernels (api.hip has none)
    assert stats["kernels"] > 150 and stats["kernels_with_calls"] > 50 and stats["flat_insns"] > 1000, stats
    assert not findings, "\n".join(findings[:20])


def test_checker_recognises_the_round3_fault():
    """The ISA of the faulting helper (profiles/r04_aperture_fault_callee.s.gz, the compiler's own output of the build that
    faulted): the backward pass reaches tab[7 - j] as (v[210:211] - 96 j) + offset:672.."""
    path = os.path.join(ROOT, "profiles", "r04_aperture_fault_callee.s.gz")
    f = check_codeobj.Func("jac_small_multiples_ool", 0)
    with gzip.open(path, "rt") as fh:
        for i, line in enumerate(fh, 1):
            t = line.split(";")[0].strip()
            if t and not t.endswith(":") and not t.startswith("."):
                f.lines.append((i, t))
    found = check_codeobj.check_flat_bias(f)
    assert found and all("v[210:211]" in x for x in found), found[:3]
    assert any("offset:672" in x for x in found)
    # and a forward walk (base incremented by +0x60, the helper's first loop) is not flagged
    assert not any("v[86:87]" in x for x in found)


def test_scratch_accounting_detects_a_short_descriptor():
    k = check_codeobj.Func("kern", 0x100)
    k.lines = [(0x100, "s_movk_i32 s32, 0x480"), (0x104, "s_getpc_b64 s[2:3]"), (0x108, "s_add_u32 s2, s2, 0xf8"),
               (0x110, "s_addc_u32 s3, s3, 0"), (0x114, "s_swappc_b64 s[30:31], s[2:3]"), (0x118, "s_endpgm")]
    c = check_codeobj.Func("callee", 0x200)
    c.lines = [(0x200, "s_mov_b32 s33, s32"), (0x204, "s_addk_i32 s32, 0x390"), (0x208, "s_setpc_b64 s[30:31]")]
    funcs = {"kern": k, "callee": c}
    kernels = {"kern": {"scratch": 2064, "dynamic": False}}
    check_codeobj.analyse_frames(funcs, kernels)
    assert k.frame == 0x480 and c.frame == 0x390 and k.calls == {"callee"} and not k.unresolved
    assert check_codeobj.deepest(funcs, "kern", {}) == 2064
