#!/usr/bin/env python3
"""Static checks on the gfx950 code objects inside libmlhip.so (or any .o / .so with a .hip_fatbin section).

Why this exists ("the aperture violation of round 3"): a FLAT instruction picks its aperture (private /
LDS / global) from the high bits of VADDR alone, *before* the instruction's immediate offset is added.  When the
compiler's loop strength reduction walks a by-reference array of the CALLER'S FRAME backwards, it keeps a 64-bit base
that it decrements per iteration and folds a positive constant into `offset:`; a frame object that sits closer to the
start of the private aperture than that constant makes the base underflow below the aperture (the low word wraps, the
high word becomes aperture_hi - 1), the access is routed as a global one to an illegal address, and the queue aborts
with HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION.  Nothing in the source is out of bounds.

Two checks, both on the disassembly of the shipped code objects (llvm-objdump; a few seconds per translation unit):

  1. biased flat pointers -- inside a function, a VGPR pair that is (a) the target of a 64-bit addition of a NEGATIVE
     constant (v_lshl_add_u64 with an SGPR pair holding a negative literal, or v_add_co / v_addc with negative
     literals) and (b) the address of a flat_load / flat_store / flat_atomic.  None may exist.
  2. scratch accounting -- for every kernel, .private_segment_fixed_size of its descriptor covers its own frame plus the
     deepest chain of callee frames (frames read from the prologues, the call graph from the s_getpc / s_swappc pairs),
     no kernel uses a dynamic stack, and every call target resolves to a function of the same code object.

  3. host / device agreement -- every kernel the host side of the library can launch (its handle symbols, named like the
     kernels) exists in one of the code objects.  hipcc compiles a translation unit twice, device pass first: a header
     edited between the two passes of a running build gives a library whose launches abort with "Cannot find Symbol"
     (it happened in round 4; the build is never edited under any more, and this check catches it if it is).

Usage: tools/check_codeobj.py [path ...]   (default: mathlib_amd/libmlhip.so); exit status 1 on any finding.
Used by tests/test_codeobj.py.
"""
from __future__ import annotations

import os
import re
import shutil
import struct
import subprocess
import sys
import tempfile

LLVM_BIN = os.environ.get("MLHIP_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _tool(name: str) -> str:
    p = os.path.join(LLVM_BIN, name)
    return p if os.path.exists(p) else name


def fatbin_section(path: str) -> bytes:
    """Bytes of the .hip_fatbin section of an ELF object / shared library."""
    with tempfile.NamedTemporaryFile(suffix=".fatbin") as t:
        r = subprocess.run([_tool("llvm-objcopy"), "--dump-section", ".hip_fatbin=" + t.name, path, os.devnull],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("llvm-objcopy failed on %s: %s" % (path, r.stderr))
        with open(t.name, "rb") as f:
            return f.read()


def code_objects(fatbin: bytes, arch: str = "gfx950") -> list[bytes]:
    """Every device code object for `arch` in a (possibly concatenated) clang offload bundle."""
    out = []
    pos = 0
    while True:
        start = fatbin.find(MAGIC, pos)
        if start < 0:
            break
        p = start + len(MAGIC)
        (n,) = struct.unpack_from("<Q", fatbin, p)
        p += 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", fatbin, p)
            p += 24
            triple = fatbin[p : p + tl].decode()
            p += tl
            if triple.startswith("hip") and arch in triple and size:
                out.append(fatbin[start + off : start + off + size])
        pos = p
    return out


class Func:
    def __init__(self, name: str, addr: int):
        self.name, self.addr = name, addr
        self.lines: list[tuple[int, str]] = []  # (address, instruction text)
        self.frame = 0  # bytes of private memory this function itself needs
        self.calls: set[str] = set()
        self.unresolved: list[str] = []


_FUNC = re.compile(r"^([0-9a-f]+) <(.+)>:$")
_INSN = re.compile(r"^\s+(\S.*?)\s*// ([0-9A-F]+):")


def disassemble(co: bytes) -> tuple[dict[str, Func], dict[str, dict]]:
    """(functions by name, kernel metadata by name) of one code object."""
    with tempfile.NamedTemporaryFile(suffix=".co") as t:
        t.write(co)
        t.flush()
        dis = subprocess.run([_tool("llvm-objdump"), "-d", "--mcpu=gfx950", t.name], capture_output=True, text=True)
        if dis.returncode != 0:
            raise RuntimeError("llvm-objdump failed: " + dis.stderr)
        notes = subprocess.run([_tool("llvm-readelf"), "--notes", t.name], capture_output=True, text=True).stdout
    funcs: dict[str, Func] = {}
    cur = None
    for line in dis.stdout.splitlines():
        m = _FUNC.match(line)
        if m:
            cur = Func(m.group(2), int(m.group(1), 16))
            funcs[cur.name] = cur
            continue
        if cur is None:
            continue
        m = _INSN.match(line)
        if m:
            cur.lines.append((int(m.group(2), 16), m.group(1)))
    kernels: dict[str, dict] = {}
    name = None
    for line in notes.splitlines():
        s = line.strip()
        if s.startswith(".name:"):
            name = s.split(":", 1)[1].strip()
            kernels.setdefault(name, {})
        elif name and s.startswith(".private_segment_fixed_size:"):
            kernels[name]["scratch"] = int(s.split(":")[1])
        elif name and s.startswith(".uses_dynamic_stack:"):
            kernels[name]["dynamic"] = s.split(":")[1].strip() == "true"
        elif name and s.startswith(".symbol:"):
            kernels[name]["symbol"] = s.split(":", 1)[1].strip()
    return funcs, {k: v for k, v in kernels.items() if "scratch" in v}


def _lit(tok: str) -> int | None:
    tok = tok.strip().rstrip(",")
    try:
        return int(tok, 0)
    except ValueError:
        return None


def _neg32(v: int | None) -> bool:
    return v is not None and (v < 0 or v >= 0x80000000)


_PAIR = re.compile(r"^([sv])\[(\d+):(\d+)\]$")


def check_flat_bias(f: Func) -> list[str]:
    """Finding strings for check 1 (module docstring) in one function.  Flow-insensitive on purpose: a register pair that
    is decremented ANYWHERE in the function must not address a flat access anywhere in it (a loop's decrement sits below
    the accesses it feeds)."""
    neg_sgpr: dict[int, bool] = {}  # SGPR -> holds a literal with the sign bit set (tracked in program order)
    biased: dict[int, str] = {}  # low VGPR of a decremented pair -> the instruction that did it
    pending_lo: tuple[int, str] | None = None  # v_add_co_u32 with a negative literal: (dst vgpr, text)
    for addr, text in f.lines:
        op, _, rest = text.partition(" ")
        args = [a.strip() for a in rest.split(",")] if rest else []
        if op in ("s_mov_b32", "s_movk_i32") and len(args) == 2 and args[0].startswith("s") and args[0][1:].isdigit():
            v = _lit(args[1])
            if op == "s_movk_i32" and v is not None and v >= 0x8000:
                v -= 0x10000  # sign-extended 16-bit literal
            neg_sgpr[int(args[0][1:])] = _neg32(v)
        elif op == "s_mov_b64" and len(args) == 2:
            m = _PAIR.match(args[0])
            v = _lit(args[1])
            if m and m.group(1) == "s":
                neg = v is not None and (v < 0 or v >= 1 << 63)
                neg_sgpr[int(m.group(2))] = neg
                neg_sgpr[int(m.group(3))] = neg
        elif op.startswith("s_") and args and args[0].startswith("s"):
            m = _PAIR.match(args[0])
            if m:
                for r in range(int(m.group(2)), int(m.group(3)) + 1):
                    neg_sgpr[r] = False
            elif args[0][1:].isdigit():
                neg_sgpr[int(args[0][1:])] = False
        if op == "v_lshl_add_u64" and len(args) == 4:
            d, s = _PAIR.match(args[0]), _PAIR.match(args[3])
            if d and d.group(1) == "v":
                lo = int(d.group(2))
                if s and s.group(1) == "s" and neg_sgpr.get(int(s.group(3)), False):
                    biased[lo] = "%X: %s" % (addr, text)
                elif _lit(args[3]) is not None and _lit(args[3]) < 0:
                    biased[lo] = "%X: %s" % (addr, text)
        elif op.startswith("v_add_co_u32") and len(args) >= 4:
            lits = [_lit(a) for a in args[2:]]
            if any(_neg32(v) for v in lits) and args[0].startswith("v") and args[0][1:].isdigit():
                pending_lo = (int(args[0][1:]), "%X: %s" % (addr, text))
            else:
                pending_lo = None
        elif op.startswith("v_addc_co_u32") and pending_lo is not None:
            lits = [_lit(a) for a in args[2:]]
            if any(v == -1 or v == 0xFFFFFFFF for v in lits if v is not None) and args[0].startswith("v"):
                biased[pending_lo[0]] = pending_lo[1]
            pending_lo = None
    out = []
    if not biased:
        return out
    for addr, text in f.lines:
        op, _, rest = text.partition(" ")
        if not op.startswith(("flat_load", "flat_store", "flat_atomic")):
            continue
        args = [a.strip() for a in rest.split(",")]
        # flat_load vdst, v[a:b] [offset:N]   /   flat_store v[a:b], vdata [offset:N]   /   flat_atomic [vdst,] v[a:b], vdata
        cand = args[0] if op.startswith("flat_store") else (args[1] if len(args) > 1 else "")
        if op.startswith("flat_atomic") and _PAIR.match(args[0].split(" ")[0]) and not ("glc" in text or "sc0" in text):
            cand = args[0]  # no return value: the address comes first
        m = _PAIR.match(cand.split(" ")[0])
        if m and m.group(1) == "v" and int(m.group(3)) == int(m.group(2)) + 1 and int(m.group(2)) in biased:
            out.append("%s: flat access %X: %s -- its address pair is decremented at %s" %
                       (f.name, addr, text, biased[int(m.group(2))]))
    return out


def analyse_frames(funcs: dict[str, Func], kernels: dict[str, dict]) -> None:
    """Fills Func.frame / calls / unresolved from the prologues and the call sequences."""
    by_addr = {f.addr: f.name for f in funcs.values()}
    for f in funcs.values():
        is_kernel = f.name in kernels
        getpc: dict[int, int] = {}  # low SGPR of a pair -> resolved target address
        pend: dict[int, int] = {}  # low SGPR -> address of the instruction after s_getpc (base of the relative add)
        seen_sp = False
        for i, (addr, text) in enumerate(f.lines):
            op, _, rest = text.partition(" ")
            args = [a.strip() for a in rest.split(",")] if rest else []
            if not seen_sp and args and args[0] == "s32":
                if is_kernel and op in ("s_movk_i32", "s_mov_b32"):
                    v = _lit(args[1])
                    if v is not None:
                        f.frame, seen_sp = v, True
                elif not is_kernel and op in ("s_addk_i32", "s_add_i32", "s_add_u32"):
                    v = _lit(args[-1])
                    if v is not None and v > 0:
                        f.frame, seen_sp = v, True
            if op == "s_getpc_b64":
                m = _PAIR.match(args[0])
                if m:
                    pend[int(m.group(2))] = f.lines[i + 1][0] if i + 1 < len(f.lines) else addr + 4
            elif op == "s_add_u32" and len(args) == 3 and args[0] == args[1] and args[0][1:].isdigit():
                r = int(args[0][1:])
                v = _lit(args[2])
                if r in pend and v is not None:
                    if v >= 0x80000000:
                        v -= 1 << 32
                    getpc[r] = pend.pop(r) + v
            elif op == "s_swappc_b64":
                m = _PAIR.match(args[1])
                tgt = getpc.get(int(m.group(2))) if m else None
                if tgt is not None and tgt in by_addr:
                    f.calls.add(by_addr[tgt])
                else:
                    f.unresolved.append("%X: %s" % (addr, text))
        if is_kernel and not seen_sp:
            # a kernel that makes no call never sets s32; its frame is whatever its descriptor says
            f.frame = kernels[f.name]["scratch"] if not f.calls else 0


def deepest(funcs: dict[str, Func], name: str, memo: dict[str, int], stack: tuple[str, ...] = ()) -> int:
    if name in memo:
        return memo[name]
    if name in stack:
        raise RuntimeError("recursion through " + name)
    f = funcs[name]
    d = f.frame + max([deepest(funcs, c, memo, stack + (name,)) for c in f.calls] or [0])
    memo[name] = d
    return d


def check_code_object(co: bytes) -> tuple[list[str], dict]:
    funcs, kernels = disassemble(co)
    analyse_frames(funcs, kernels)
    findings = []
    stats = {"functions": len(funcs), "kernels": len(kernels), "kernels_with_calls": 0, "flat_insns": 0, "max_scratch": 0,
             "kernel_names": sorted(kernels)}
    memo: dict[str, int] = {}
    for f in funcs.values():
        findings += check_flat_bias(f)
        stats["flat_insns"] += sum(1 for _, t in f.lines if t.startswith("flat_"))
        for u in f.unresolved:
            findings.append("%s: call target not resolved at %s" % (f.name, u))
    for k, meta in kernels.items():
        if k not in funcs:
            findings.append("kernel %s has a descriptor but no code" % k)
            continue
        if meta.get("dynamic"):
            findings.append("kernel %s uses a dynamic stack" % k)
        if funcs[k].calls:
            stats["kernels_with_calls"] += 1
        need = deepest(funcs, k, memo)
        stats["max_scratch"] = max(stats["max_scratch"], meta["scratch"])
        if meta["scratch"] < need:
            findings.append("kernel %s: descriptor scratch %d B < own frame %d B + callee chain = %d B" %
                            (k, meta["scratch"], funcs[k].frame, need))
    return findings, stats


def _check_one(co: bytes):
    return check_code_object(co)


def host_kernel_handles(path: str) -> set[str]:
    """Mangled names of the kernels the host code of `path` registers: for every `__device_stub__` function clang emits a
    data symbol (the launch handle) that carries the kernel's own mangled name."""
    out = subprocess.run([shutil.which("nm") or _tool("llvm-nm"), path], capture_output=True, text=True)
    if out.returncode != 0:
        return set()
    data, stubs = set(), 0
    for ln in out.stdout.splitlines():
        parts = ln.split()
        if len(parts) < 2:
            continue
        typ, name = parts[-2], parts[-1]
        if "__device_stub__" in name:
            stubs += 1
        elif typ in "VvBbDd" and name.startswith("_Z") and re.search(r"\d+k_[a-z0-9_]+", name):
            data.add(name)
    return data if stubs else set()


def check_file(path: str, jobs: int = 1) -> tuple[list[str], dict]:
    total = {"code_objects": 0, "functions": 0, "kernels": 0, "kernels_with_calls": 0, "flat_insns": 0, "max_scratch": 0}
    findings = []
    cos = code_objects(fatbin_section(path))
    if jobs > 1 and len(cos) > 1:
        from concurrent.futures import ProcessPoolExecutor

        with ProcessPoolExecutor(max_workers=min(jobs, len(cos))) as ex:
            results = list(ex.map(_check_one, cos))
    else:
        results = [check_code_object(co) for co in cos]
    device_kernels = set()
    for _, st in results:
        device_kernels |= set(st.pop("kernel_names"))
    handles = host_kernel_handles(path)
    total["host_kernel_handles"] = len(handles)
    for h in sorted(handles - device_kernels):
        findings.append("host code can launch %s but no code object defines it (sources edited during the build?)" % h)
    for f, s in results:
        findings += f
        total["code_objects"] += 1
        for k in ("functions", "kernels", "kernels_with_calls", "flat_insns"):
            total[k] += s[k]
        total["max_scratch"] = max(total["max_scratch"], s["max_scratch"])
    return findings, total


def main(argv: list[str]) -> int:
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    paths = argv[1:] or [os.path.join(here, "mathlib_amd", "libmlhip.so")]
    rc = 0
    for p in paths:
        findings, stats = check_file(p, jobs=int(os.environ.get("MLHIP_CHECK_JOBS", "4")))
        print("%s: %s" % (p, stats))
        for f in findings:
            print("  FINDING: " + f)
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv))
