import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
CURVE_TAGS = {"BN254": "bn254", "BLS12-381": "bls12_381", "BLS12-377": "bls12_377"}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def load_golden(curve_name: str) -> dict:
    with open(os.path.join(GOLDEN, CURVE_TAGS[curve_name] + ".json")) as f:
        return json.load(f)


def load_msm1000(curve_name: str, fp_bytes: int):
    with open(os.path.join(GOLDEN, CURVE_TAGS[curve_name] + "_msm1000.bin"), "rb") as f:
        blob = f.read()
    n = 1000
    ps = 2 * fp_bytes
    return blob[: n * ps], blob[n * ps : n * ps + n * 32], blob[n * ps + n * 32 :]


@pytest.fixture(scope="session")
def hostmath():
    """g++ build of the kernels' __host__ __device__ headers (test artifact, see tests/hostmath)."""
    import ctypes

    d = os.path.join(ROOT, "tests", "hostmath")
    so = os.path.join(d, "libhostmath.so")
    src = os.path.join(d, "hostmath.cpp")
    csrc = os.path.join(ROOT, "mathlib_amd", "csrc")
    newest = max([os.path.getmtime(src)] + [os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc) if f.endswith(".h")])
    if not os.path.exists(so) or os.path.getmtime(so) < newest:
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-DMLHIP_HOST_USE_DEVICE_PATH", "-o", so, src])
    return ctypes.CDLL(so)


# switches that select a second implementation which only the test build contains (python -m mathlib_amd.build --alt,
# MLHIP_LIB=mathlib_amd/libmlhip_alt.so); include/mlhip.h lists them under "2nd impl"
ALT_SWITCHES = ("MLHIP_ACC32", "MLHIP_REDUCE32", "MLHIP_REDUCE_ONE_LANE", "MLHIP_G2_KC", "MLHIP_PAIRING_ONE_LANE", "MLHIP_PAIRING_SAT",
                "MLHIP_SCALAR_MUL_ONE_LANE")


def alt_build() -> bool:
    """Can this test run a second implementation?  Either the loaded library is the test build (mlhip_version() bit 16), or
    an up-to-date libmlhip_alt.so lies next to the product library and the `monkeypatch` fixture below will route the test
    through it when it turns such a switch on."""
    from mathlib_amd import _lib

    return bool(_lib.load().mlhip_version() & 0x10000) or _lib.alt_available()


@pytest.fixture
def monkeypatch(monkeypatch):
    """pytest's monkeypatch, except that switching on a second implementation the product library does not contain routes
    the REST OF THE TEST through the test build (mathlib_amd/_lib.py: use_alt; back to the product at teardown) -- or skips
    the test when there is no up-to-date test build (the product ignores those switches: the test would silently re-run the
    default path).  Tests that reach such a switch only after creating library objects guard it with alt_build() and create
    the objects afterwards: a handle never crosses from one library to the other."""
    from mathlib_amd import _lib

    plain = monkeypatch.setenv
    switched = []

    def setenv(name, value, *a, **k):
        if name in ALT_SWITCHES and str(value) == "1" and not (_lib.load().mlhip_version() & 0x10000):
            if not _lib.alt_available():
                pytest.skip("%s=1 needs the test build (python -m mathlib_amd.build --alt)" % name)
            _lib.use_alt(True)
            switched.append(name)
        return plain(name, value, *a, **k)

    monkeypatch.setenv = setenv
    yield monkeypatch
    if switched:
        _lib.load().mlhip_release_cache()
        _lib.use_alt(False)


@pytest.fixture(scope="session")
def mlhip():
    from mathlib_amd import _lib

    return _lib
