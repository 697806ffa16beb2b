#!/usr/bin/env python3
"""Per-GPU shapes of BASELINE.json's multi-GPU configs on one MI355X: config 4 (BLS12-381 2^24 G1+G2 over 8 GPUs
= 2^21 per GPU; also the whole 2^24 G1 on one GPU) and config 5 (BLS12-377 2^22 over 8 GPUs = 2^19 per GPU).
Each result is checked by the split-sum property MSM(all) == MSM(first part) + MSM(rest)."""
import ctypes
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
gen.manual_seed(5)


def rnd(n):
    return torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, generator=gen, device=dev).view(torch.uint8).reshape(n, 32).contiguous()


def run(name, group, n, c):
    g = load_golden(name)
    cid = g["curve_id"]
    fpb, g1b, g2b, gtb = _lib.sizes(cid)
    sz = g1b if group == 1 else g2b
    base = torch.frombuffer(bytearray(bytes.fromhex(g["g1_gen" if group == 1 else "g2_gen"])), dtype=torch.uint8).to(dev)
    P = torch.empty(n * sz, dtype=torch.uint8, device=dev)
    _lib.check(lib.mlhip_scalar_mul_device(cid, group, base.data_ptr(), 0, rnd(n).data_ptr(), 0, n, P.data_ptr(), st))
    S = rnd(n)
    torch.cuda.synchronize()
    plan = _lib.MsmPlan(cid, group, n, c)
    plan.set_profiling(True)
    best, out = None, None
    for _ in range(3):
        t0 = time.perf_counter()
        out = plan.run(P.data_ptr(), S.data_ptr(), n, False, st)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    ph = {k: round(v, 3) for k, v in plan.timings().items()}
    h = n // 3 + 17
    a = plan.run(P.data_ptr(), S.data_ptr(), h, False, st)
    b = plan.run(P.data_ptr() + h * sz, S.data_ptr() + h * 32, n - h, False, st)
    tot = ctypes.create_string_buffer(sz)
    _lib.check((lib.mlhip_g1_sum if group == 1 else lib.mlhip_g2_sum)(cid, a + b, 2, tot))
    print("%s G%d n=2^%d c=%d: %.3f ms -> %.3e scalar-muls/s  split-sum check %s  phases=%s" % (
        name, group, n.bit_length() - 1, c, best * 1e3, n / best, "OK" if tot.raw == out else "MISMATCH", ph), flush=True)
    plan.close()


run("BLS12-381", 1, 1 << 21, 16)
run("BLS12-381", 2, 1 << 21, 16)
run("BLS12-381", 1, 1 << 24, 16)
run("BLS12-377", 1, 1 << 19, 16)
run("BLS12-377", 1, 1 << 22, 16)
run("BN254", 1, 1000, 0)
