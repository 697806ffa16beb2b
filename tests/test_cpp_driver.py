"""The C++ mirror of the driver interface (include/mlhip_driver.hpp) and its test program
(tests/cpp/driver_test.cpp, a restatement of math_test.go's helpers).  The host mode needs no GPU."""
import os
import subprocess

import pytest

from conftest import ROOT, load_golden
from oracle import pyref as R

BIN = os.path.join(ROOT, "tests", "cpp", "driver_test")
NAMES = ["BN254", "BLS12-381", "BLS12-377"]


def _build():
    src = os.path.join(ROOT, "tests", "cpp", "driver_test.cpp")
    hdr = os.path.join(ROOT, "include", "mlhip_driver.hpp")
    lib = os.path.join(ROOT, "mathlib_amd", "libmlhip.so")
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < max(os.path.getmtime(src), os.path.getmtime(hdr), os.path.getmtime(lib)):
        subprocess.check_call(
            ["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), src, "-o", BIN,
             "-L", os.path.join(ROOT, "mathlib_amd"), "-lmlhip", "-Wl,-rpath," + os.path.join(ROOT, "mathlib_amd")]
        )
    return BIN


def _lines(out):
    d = {}
    for ln in out.splitlines():
        parts = ln.split()
        if len(parts) == 3 and parts[0] in NAMES:
            d[(parts[0], parts[1])] = parts[2]
    return d


def test_cpp_driver_host_logic():
    out = subprocess.run([_build(), "host"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "RESULT OK" in out.stdout, out.stdout + out.stderr
    d = _lines(out.stdout)
    for name in NAMES:
        cp = R.CURVES[name]
        assert d[(name, "gen_g1_compressed")] == R.g1_wire_compressed(cp, cp.g1).hex()
        assert d[(name, "gen_g1_bytes")] == R.g1_wire_uncompressed(cp, cp.g1).hex()
        assert d[(name, "inf_g1_compressed")] == R.g1_wire_compressed(cp, None).hex()


@pytest.mark.gpu
def test_cpp_driver_reference_tests_on_gpu():
    g = load_golden("BLS12-377")
    co = g["g2_gen_coords"]
    out = subprocess.run([_build(), "gpu", co[0][0], co[0][1], co[1][0], co[1][1]], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "RESULT OK" in out.stdout, out.stdout + out.stderr
    d = _lines(out.stdout)
    for name in NAMES:
        cp = R.CURVES[name]
        assert d[(name, "msm_40G_compressed")] == R.g1_wire_compressed(cp, R.g1_mul(cp, cp.g1, 40)).hex()
        assert d[(name, "gen_gt_bytes")] == load_golden(name)["gen_gt_wire"]
        assert d[(name, "gen_g2_compressed")] == R.g2_wire_compressed(cp, R.g2_generator(cp)).hex()
