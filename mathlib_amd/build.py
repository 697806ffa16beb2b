"""Builds mathlib_amd/libmlhip.so (HIP, gfx950 only) in-tree: one hipcc job per translation unit, in
parallel, then one link.  `python -m mathlib_amd.build` or `__graft_entry__.build()`.

`python -m mathlib_amd.build --alt` builds the TEST library mathlib_amd/libmlhip_alt.so beside it (-DMLHIP_BUILD_ALT=1):
the same code plus the second implementations the parity tests compare the default kernels with (mlhip_internal.h);
`MLHIP_LIB=mathlib_amd/libmlhip_alt.so pytest -m gpu` runs the suite on it, the product library skips those cases.

Objects are cached under mathlib_amd/csrc/_obj (_obj_alt) and rebuilt when any source/header is newer.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libmlhip.so")
ARCH = "gfx950"

UNITS = [
    "api.hip",
    "tu_msm_bn254.hip",
    "tu_msm_bls381.hip",
    "tu_msm_bls377.hip",
    "tu_pairing_bn254.hip",
    "tu_pairing_bls381.hip",
    "tu_pairing_bls377.hip",
    "tu_codec_bn254.hip",
    "tu_codec_bls381.hip",
    "tu_codec_bls377.hip",
]
# -fvisibility=hidden: the dynamic symbol table is the MLHIP_API functions of include/mlhip.h and nothing else
# (tests/test_abi.py); kernels keep the protected visibility the HIP runtime needs inside the code objects.
FLAGS = ["-O3", "-std=c++17", "--offload-arch=" + ARCH, "-fPIC", "-fno-gpu-rdc", "-fvisibility=hidden",
         "-fvisibility-inlines-hidden"] + os.environ.get("MLHIP_EXTRA_HIPCC_FLAGS", "").split()


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _deps_mtime() -> float:
    m = 0.0
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in os.listdir(root):
            if f.endswith((".h", ".hip", ".inc")):
                m = max(m, os.path.getmtime(os.path.join(root, f)))
    return m


def source_hash() -> str:
    """sha256 over the kernel sources (csrc/*.h, *.hip, *.inc and include/*.h, by name): profiles taken on a GPU box carry
    it, so that bench.py can tell whether a committed PMC pass was made with the code it is running."""
    import hashlib

    h = hashlib.sha256()
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in sorted(os.listdir(root)):
            if f.endswith((".h", ".hpp", ".hip", ".inc")):
                h.update(f.encode())
                with open(os.path.join(root, f), "rb") as fh:
                    h.update(fh.read())
    return h.hexdigest()[:16]


def abi_functions() -> list[str]:
    """The entry points include/mlhip.h declares (MLHIP_API ...): the library's whole dynamic symbol table."""
    import re

    with open(os.path.join(os.path.dirname(HERE), "include", "mlhip.h")) as f:
        return sorted(set(re.findall(r"^MLHIP_API\s+[^;(]*?\b(mlhip_[a-z0-9_]+)\s*\(", f.read(), re.M)))


def _version_script() -> str:
    """Linker version script naming exactly the header's functions: everything else -- the kernels' host-side handles,
    libstdc++ template instantiations, the mlhip_tu_* / mlhip_rt entry points between translation units -- stays local."""
    path = os.path.join(OBJ, "mlhip.map")
    text = "MLHIP_1 {\n  global:\n" + "".join("    %s;\n" % n for n in abi_functions()) + "  local:\n    *;\n};\n"
    if not os.path.exists(path) or open(path).read() != text:
        with open(path, "w") as f:
            f.write(text)
    return path


def _compile(unit: str, newest: float, verbose: bool, alt: bool = False) -> str:
    src = os.path.join(CSRC, unit)
    obj = os.path.join(OBJ + ("_alt" if alt else ""), unit.replace(".hip", ".o"))
    if os.path.exists(obj) and os.path.getmtime(obj) >= newest:
        return obj
    cmd = [_hipcc()] + FLAGS + (["-DMLHIP_BUILD_ALT=1"] if alt else []) + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    # hipcc compiles the unit twice (device pass, then host pass): a header saved between the two gives an object whose host
    # code launches kernels its code object does not have ("Cannot find Symbol" at the first launch; it happened in round 4).
    # The object is therefore stamped with the time the compile STARTED -- any dependency saved after that makes it stale --
    # and a compile that was overtaken by an edit is repeated at once.
    for _ in range(3):
        import time

        t0 = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (unit, r.stdout, r.stderr))
        os.utime(obj, (t0, t0))
        if _deps_mtime() <= t0:
            break
    return obj


LIB_ALT = os.path.join(HERE, "libmlhip_alt.so")


def build(verbose: bool = True, jobs: int | None = None, alt: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(OBJ + ("_alt" if alt else ""), exist_ok=True)
    newest = _deps_mtime()
    jobs = jobs or min(len(UNITS), max(1, (os.cpu_count() or 2)))
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda u: _compile(u, newest, verbose, alt), UNITS))
    LIB = LIB_ALT if alt else globals()["LIB"]
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(o) for o in objs):
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-Wl,--version-script=" + _version_script(), "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    # which sources this library was built from (mathlib_amd/_lib.py: alt_available -- the tests use the test build only when
    # it matches the tree)
    with open(LIB + ".srchash", "w") as f:
        f.write(source_hash() + "\n")
    return LIB


if __name__ == "__main__":
    if "--source-hash" in sys.argv:
        print(source_hash())
    else:
        print(build(verbose="-q" not in sys.argv, alt="--alt" in sys.argv))
