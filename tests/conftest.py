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


@pytest.fixture(scope="session")
def mlhip():
    from mathlib_amd import _lib

    return _lib
