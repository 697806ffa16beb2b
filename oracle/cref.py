"""oracle/cref.py -- TEST INFRASTRUCTURE: ctypes access to oracle/libcref.so (the C restatement).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
from __future__ import annotations

import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libcref.so")
_lib = None


def build() -> str:
    subprocess.check_call(["make", "-s", "-C", HERE])
    return LIB


def load() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        srcs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith((".c", ".h"))]
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(s) for s in srcs):
            build()
        _lib = ctypes.CDLL(LIB)
        vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        _lib.cref_fp_mul.argtypes = [ci, vp, vp, vp]
        _lib.cref_msm_g1.argtypes = [ci, vp, vp, ci, sz, ci, ci, vp]
        _lib.cref_msm_g2.argtypes = [ci, vp, vp, ci, sz, ci, ci, vp]
        _lib.cref_miller_loop.argtypes = [ci, vp, vp, sz, sz, vp, ci]
        _lib.cref_final_exp.argtypes = [ci, vp, sz, vp, ci]
        _lib.cref_pairing_batch.argtypes = [ci, vp, vp, sz, vp, ci]
        _lib.cref_gt_mul.argtypes = [ci, vp, vp, sz, vp]
        _lib.cref_gen_g1.argtypes = [ci, vp, vp, sz, vp]
        _lib.cref_gen_g2.argtypes = [ci, vp, vp, sz, vp]
        _lib.cref_g1_mul.argtypes = [ci, vp, vp, ci, vp]
        _lib.cref_g2_mul.argtypes = [ci, vp, vp, ci, vp]
    return _lib


FP_BYTES = {0: 32, 1: 48, 2: 48}


def _buf(b):
    """accept bytes / bytearray / numpy array; return a ctypes-passable pointer and keep-alive"""
    if isinstance(b, (bytes, bytearray)):
        return b
    return b.ctypes.data_as(ctypes.c_void_p)  # numpy


def msm(curve: int, group: int, points, scalars, n: int, mont: bool = False, c: int = 0, threads: int = 1) -> bytes:
    out = ctypes.create_string_buffer(FP_BYTES[curve] * 2 * group)
    fn = load().cref_msm_g1 if group == 1 else load().cref_msm_g2
    assert fn(curve, _buf(points), _buf(scalars), 1 if mont else 0, n, c, threads, out) == 0
    return out.raw


def pairing_batch(curve: int, g1, g2, n: int, threads: int = 1) -> bytes:
    out = ctypes.create_string_buffer(FP_BYTES[curve] * 12 * n)
    assert load().cref_pairing_batch(curve, _buf(g1), _buf(g2), n, out, threads) == 0
    return out.raw


def miller_loop(curve: int, g1, g2, ppp: int, n: int, threads: int = 1) -> bytes:
    out = ctypes.create_string_buffer(FP_BYTES[curve] * 12 * n)
    assert load().cref_miller_loop(curve, _buf(g1), _buf(g2), ppp, n, out, threads) == 0
    return out.raw


def final_exp(curve: int, gt, n: int, threads: int = 1) -> bytes:
    out = ctypes.create_string_buffer(FP_BYTES[curve] * 12 * n)
    assert load().cref_final_exp(curve, _buf(gt), n, out, threads) == 0
    return out.raw


def gt_mul(curve: int, a, b, n: int) -> bytes:
    out = ctypes.create_string_buffer(FP_BYTES[curve] * 12 * n)
    assert load().cref_gt_mul(curve, _buf(a), _buf(b), n, out) == 0
    return out.raw


def gen_points(curve: int, group: int, k0: int, k1: int, n: int) -> bytes:
    """P_i = [k0]G + i*[k1]G in the C-ABI layout (distinct points of the r-torsion)."""
    out = ctypes.create_string_buffer(FP_BYTES[curve] * 2 * group * n)
    fn = load().cref_gen_g1 if group == 1 else load().cref_gen_g2
    assert fn(curve, k0.to_bytes(32, "little"), k1.to_bytes(32, "little"), n, out) == 0
    return out.raw


def point_mul(curve: int, group: int, p: bytes, k: int, mont: bool = False) -> bytes:
    out = ctypes.create_string_buffer(FP_BYTES[curve] * 2 * group)
    fn = load().cref_g1_mul if group == 1 else load().cref_g2_mul
    assert fn(curve, p, (k % (1 << 256)).to_bytes(32, "little"), 1 if mont else 0, out) == 0
    return out.raw


def fp_mul(curve: int, a: bytes, b: bytes) -> bytes:
    out = ctypes.create_string_buffer(FP_BYTES[curve])
    assert load().cref_fp_mul(curve, a, b, out) == 0
    return out.raw
