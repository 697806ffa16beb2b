#!/usr/bin/env python3
"""Throughput of the wire-format kernels on one MI355X, inputs resident in HBM: encode n random
points, decode them back with and without the subgroup check, verify the round trip bit for bit."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden  # noqa: E402
from mathlib_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
gen = torch.Generator(device=dev)
gen.manual_seed(11)


def rnd(n):
    return torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, generator=gen, device=dev).view(torch.uint8).reshape(n, 32).contiguous()


def timed(fn, reps=3):
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return best


def run(name, group, n):
    g = load_golden(name)
    cid = g["curve_id"]
    fpb, g1b, g2b, gtb = _lib.sizes(cid)
    sz = g1b if group == 1 else g2b
    base = torch.frombuffer(bytearray(bytes.fromhex(g["g1_gen" if group == 1 else "g2_gen"])), dtype=torch.uint8).to(dev)
    P = torch.empty(n * sz, dtype=torch.uint8, device=dev)
    _lib.check(lib.mlhip_scalar_mul_device(cid, group, base.data_ptr(), 0, rnd(n).data_ptr(), 0, n, P.data_ptr(), st))
    W = torch.empty(n * sz // 2, dtype=torch.uint8, device=dev)
    Q = torch.empty_like(P)
    S = torch.empty(n, dtype=torch.uint8, device=dev)
    enc = lib.mlhip_g1_to_bytes_device if group == 1 else lib.mlhip_g2_to_bytes_device
    dec = lib.mlhip_g1_from_bytes_device if group == 1 else lib.mlhip_g2_from_bytes_device
    t_enc = timed(lambda: _lib.check(enc(cid, P.data_ptr(), n, 1, W.data_ptr(), st)))
    res = {}
    for sg in (0, 1):
        Q.zero_()
        res[sg] = timed(lambda: _lib.check(dec(cid, W.data_ptr(), n, 1, sg, Q.data_ptr(), S.data_ptr(), st)))
        ok = bool(torch.equal(P, Q)) and int(S.max()) == 0
        assert ok, "round trip mismatch"
    print("%s G%d n=2^%d: compress %.3f ms (%.2e pts/s)  decompress %.3f ms (%.2e pts/s)  decompress+subgroup %.3f ms (%.2e pts/s)  round trip OK" % (
        name, group, n.bit_length() - 1, t_enc * 1e3, n / t_enc, res[0] * 1e3, n / res[0], res[1] * 1e3, n / res[1]), flush=True)


for nm in ("BLS12-381", "BN254", "BLS12-377"):
    run(nm, 1, 1 << 18)
    run(nm, 2, 1 << 16)
