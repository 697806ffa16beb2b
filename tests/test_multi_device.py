"""Several devices in one process through the C ABI (include/mlhip.h: mlhip_init / mlhip_msm_multi /
mlhip_bases_create_multi; SURVEY.md 8e "one process, one host thread per device, the C ABI takes a device list").
The reference call this serves is math.Curve.MultiScalarMul (math.go:960-969) on host slices.  A one-GPU box
rehearses it with device lists that name device 0 more than once: the sharding, the per-shard host threads, the
per-device plan pool and the host-side combination are the code that runs on eight devices; only the device indices
differ."""
import ctypes
import os
import subprocess
import sys

import pytest

from conftest import ROOT, load_golden, load_msm1000


def _devs(*d):
    return (ctypes.c_int * len(d))(*d), len(d)


def test_device_list_host_logic(mlhip):
    """no GPU needed: argument checking, the list round trip, thread pinning"""
    lib = mlhip.load()
    d1 = 0 if mlhip.device_count() == 1 else 1  # with devices present, indices at or above their count are rejected
    try:
        arr, k = _devs(0, 0, d1)
        assert lib.mlhip_init(arr, k) == 0
        assert mlhip.get_devices() == [0, 0, d1]
        bad, k = _devs(0, -2)
        assert lib.mlhip_init(bad, k) == -1 and b"out of range" in lib.mlhip_last_error()
        big, k = _devs(0, 64)  # indices are 0 .. 63: the per-device tables are indexed by them, nothing is masked
        assert lib.mlhip_init(big, k) == -1 and b"out of range" in lib.mlhip_last_error()
        assert lib.mlhip_init(None, 3) == -1
        assert lib.mlhip_init(arr, 65) == -1
        assert mlhip.get_devices() == [0, 0, d1]  # a rejected list changes nothing
        out = ctypes.create_string_buffer(96)
        assert lib.mlhip_msm_multi(1, 1, None, 2, b"x", b"y", 0, 1, 0, out) == -1
        assert lib.mlhip_msm_multi(1, 3, arr, 2, b"x", b"y", 0, 1, 0, out) == -1
        assert lib.mlhip_msm_multi(9, 1, arr, 2, b"x", b"y", 0, 1, 0, out) == -1
        # n = 0: the identity, no device touched (MultiExp on empty slices, bls12-381.go:777)
        out = ctypes.create_string_buffer(b"\xff" * 96, 96)
        assert lib.mlhip_msm_multi(1, 1, arr, 2, None, None, 0, 0, 0, out) == 0 and out.raw == bytes(96)
        assert lib.mlhip_set_device(-1) == 0 and lib.mlhip_set_device(-2) == -1 and lib.mlhip_set_device(64) == -1
        big, k2 = _devs(0, 1023)
        assert lib.mlhip_msm_multi(1, 1, big, k2, b"x", b"y", 0, 2, 0, out) == -1 and b"out of range" in lib.mlhip_last_error()
    finally:
        lib.mlhip_shutdown()
    assert mlhip.get_devices() == []


def test_device_list_from_the_environment():
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from mathlib_amd import _lib\n"
        "print(_lib.get_devices())\n" % ROOT
    )
    for env, want in (("0,2,1", "[0, 2, 1]"), ("", "[]"), ("3", "[3]")):
        e = dict(os.environ, MLHIP_DEVICES=env)
        out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, check=True).stdout
        assert out.strip().splitlines()[-1] == want, (env, out)


def test_malformed_environment_list_is_an_error_not_device_0():
    """MLHIP_DEVICES that does not parse: mlhip_get_devices and every compute call fail with MLHIP_EINVAL (they used to
    run on device 0 without a word); mlhip_shutdown makes the library read the variable again."""
    code = (
        "import sys, os, ctypes; sys.path.insert(0, %r)\n"
        "from mathlib_amd import _lib\n"
        "lib = _lib.load()\n"
        "buf = (ctypes.c_int * 4)()\n"
        "print(lib.mlhip_get_devices(buf, 4), lib.mlhip_last_error().decode())\n"
        "out = ctypes.create_string_buffer(96)\n"
        "rc = lib.mlhip_msm_g1(1, bytes(96), bytes(32), 0, 1, 0, out)\n"
        "print(rc, lib.mlhip_last_error().decode())\n"
        "os.environ['MLHIP_DEVICES'] = '1,0'\n"
        "lib.mlhip_shutdown()\n"
        "print(_lib.get_devices())\n" % ROOT
    )
    for env in ("0,x", "0,64", "-1", "0;1"):
        e = dict(os.environ, MLHIP_DEVICES=env)
        out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, check=True).stdout.strip().splitlines()
        assert out[-3].startswith("-1 ") and "malformed" in out[-3], (env, out)
        # without a GPU the missing device is reported first; with one, the malformed list
        assert out[-2].startswith("-1 ") and "malformed" in out[-2] or out[-2].startswith("-2 "), (env, out)
        assert out[-1] == "[1, 0]", (env, out)


def test_multi_without_a_gpu_fails_loudly(mlhip):
    import torch

    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present")
    lib = mlhip.load()
    pts, scs, _ = load_msm1000("BLS12-381", 48)
    arr, k = _devs(0, 0)
    out = ctypes.create_string_buffer(96)
    assert lib.mlhip_msm_multi(1, 1, arr, k, pts, scs, 0, 1000, 0, out) == mlhip.ENODEVICE
    assert b"shard" in lib.mlhip_last_error() and b"no HIP device" in lib.mlhip_last_error()


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("curve", ["BN254", "BLS12-381", "BLS12-377"])
def test_msm_multi_matches_golden(mlhip, curve):
    lib = mlhip.load()
    g = load_golden(curve)
    cid = g["curve_id"]
    fpb, g1b, g2b, _ = mlhip.sizes(cid)
    pts, scs, exp = load_msm1000(curve, fpb)
    for devs in ((0,), (0, 0), (0, 0, 0), (0,) * 7):
        arr, k = _devs(*devs)
        for c in (0, 16):
            out = ctypes.create_string_buffer(g1b)
            mlhip.check(lib.mlhip_msm_multi(cid, 1, arr, k, pts, scs, 0, 1000, c, out))
            assert out.raw == exp, (curve, devs, c)
    # more devices than pairs; edge cases of the golden file through two shards
    arr, k = _devs(0, 0, 0, 0)
    for case in g["msm_g1"]:
        p = b"".join(bytes.fromhex(x) for x in case["points"])
        sc = b"".join(bytes.fromhex(x) for x in case["scalars"])
        out = ctypes.create_string_buffer(g1b)
        mlhip.check(lib.mlhip_msm_multi(cid, 1, arr, k, p, sc, 0, len(case["points"]), 0, out))
        assert out.raw == bytes.fromhex(case["expected"]), (curve, case["name"])
    for case in g["msm_g2"]:
        p = b"".join(bytes.fromhex(x) for x in case["points"])
        sc = b"".join(bytes.fromhex(x) for x in case["scalars"])
        out = ctypes.create_string_buffer(g2b)
        mlhip.check(lib.mlhip_msm_multi(cid, 2, arr, 2, p, sc, 0, len(case["points"]), 0, out))
        assert out.raw == bytes.fromhex(case["expected"]), (curve, case["name"])


@pytest.mark.gpu
def test_process_device_list_spreads_the_reference_shaped_calls(mlhip, monkeypatch):
    """mlhip_init + the plain entry points a Go MultiScalarMul / PairingBatch binds: above the thresholds the call is
    sharded over the list ({0, 0} here), below them and on pinned threads it is not; same bytes either way, and the
    same as the C oracle at config 4's shapes in miniature (G1 and G2 over shared scalars)."""
    import numpy as np

    from oracle import cref

    code = r"""
import ctypes, sys
sys.path.insert(0, %r)
import numpy as np
from mathlib_amd import _lib
from oracle import cref
lib = _lib.load()
cid = 1
n = 70001
sc = np.random.default_rng(11).integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
g1 = cref.gen_points(cid, 1, 31, 7, n)
g2 = cref.gen_points(cid, 2, 37, 9, n // 8)
want1 = cref.msm(cid, 1, g1, sc, n, False, 0, 8)
want2 = cref.msm(cid, 2, g2, sc[: n // 8].copy(), n // 8, False, 0, 8)
_lib.init_devices([0, 0])
assert _lib.get_devices() == [0, 0]
out = ctypes.create_string_buffer(96)
_lib.check(lib.mlhip_msm_g1(cid, g1, sc.tobytes(), 0, n, 16, out))
assert out.raw == want1, "sharded G1"
out2 = ctypes.create_string_buffer(192)
_lib.check(lib.mlhip_msm_g2(cid, g2, sc.tobytes(), 0, n // 8, 0, out2))
assert out2.raw == want2, "sharded G2"
# resident bases spread over the list; a call with fewer scalars than bases (second shard partly / not used)
h = ctypes.c_void_p()
_lib.check(lib.mlhip_bases_create(cid, 1, g1, n, 16, ctypes.byref(h)))
for m in (n, n // 2 + 5, 100):
    _lib.check(lib.mlhip_bases_msm(h, sc.tobytes(), 0, m, out))
    assert out.raw == cref.msm(cid, 1, g1[: m * 96], sc[:m].copy(), m, False, 0, 8), ("bases", m)
_lib.check(lib.mlhip_bases_destroy(h))
# pairings: independent elements, split over the list
m = 300
gt = ctypes.create_string_buffer(576 * m)
_lib.check(lib.mlhip_pairing_batch(cid, g1[: m * 96], g2[: m * 192], m, gt))
assert gt.raw == cref.pairing_batch(cid, g1[: m * 96], g2[: m * 192], m, 8), "sharded pairing batch"
ml = ctypes.create_string_buffer(576 * (m // 2))
_lib.check(lib.mlhip_miller_loop(cid, g1[: m * 96], g2[: m * 192], 2, m // 2, ml))
fe = ctypes.create_string_buffer(576 * (m // 2))
_lib.check(lib.mlhip_final_exp(cid, ml, m // 2, fe))
want = cref.final_exp(cid, cref.miller_loop(cid, g1[: m * 96], g2[: m * 192], 2, m // 2, 8), m // 2, 8)
assert fe.raw == want, "sharded Pairing2 + FExp"
# a pinned thread is never spread
_lib.check(lib.mlhip_set_device(0))
_lib.check(lib.mlhip_msm_g1(cid, g1, sc.tobytes(), 0, n, 16, out))
assert out.raw == want1
lib.mlhip_shutdown()
print("ok")
""" % ROOT
    env = dict(os.environ, MLHIP_MULTI_MIN="1000", MLHIP_MULTI_MIN_PAIRINGS="64")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.gpu
def test_bases_multi_explicit_list(mlhip):
    lib = mlhip.load()
    curve = "BLS12-381"
    g = load_golden(curve)
    cid = g["curve_id"]
    fpb, g1b, _, _ = mlhip.sizes(cid)
    pts, scs, exp = load_msm1000(curve, fpb)
    arr, k = _devs(0, 0, 0)
    h = ctypes.c_void_p()
    mlhip.check(lib.mlhip_bases_create_multi(cid, 1, arr, k, pts, 1000, 0, ctypes.byref(h)))
    out = ctypes.create_string_buffer(g1b)
    for _ in range(3):
        mlhip.check(lib.mlhip_bases_msm(h, scs, 0, 1000, out))
        assert out.raw == exp
    assert lib.mlhip_bases_msm(h, scs, 0, 1001, out) == -1
    mlhip.check(lib.mlhip_bases_destroy(h))
