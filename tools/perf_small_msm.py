#!/usr/bin/env python3
"""Latency of small MSMs (resident inputs, one plan): where the fixed cost goes.  BLS12-381 and BN254 G1."""
import os
import sys
import time

import torch

sys.path.insert(0, os.getcwd())
sys.path.insert(0, "tests")
from conftest import load_golden  # noqa: E402
from mathlib_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
gen = torch.Generator(device=dev)
gen.manual_seed(4)


def rnd(k):
    return torch.randint(-(1 << 63), (1 << 63) - 1, (k, 4), dtype=torch.int64, generator=gen, device=dev).view(torch.uint8).reshape(k, 32).contiguous()


for name in ("BLS12-381", "BN254"):
    g = load_golden(name)
    cid = g["curve_id"]
    fpb, g1b, g2b, gtb = _lib.sizes(cid)
    base = torch.frombuffer(bytearray(bytes.fromhex(g["g1_gen"])), dtype=torch.uint8).to(dev)
    for lg in (5, 8, 10, 12, 14, 16):
        n = 1 << lg
        P = torch.empty(n * g1b, dtype=torch.uint8, device=dev)
        _lib.check(lib.mlhip_scalar_mul_device(cid, 1, base.data_ptr(), 0, rnd(n).data_ptr(), 0, n, P.data_ptr(), st))
        S = rnd(n)
        plan = _lib.MsmPlan(cid, 1, n, 0)
        plan.set_profiling(True)
        best = 1e9
        for rep in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            plan.run(P.data_ptr(), S.data_ptr(), n, False, st)
            best = min(best, time.perf_counter() - t0)
        print("%s G1 MSM n=2^%d (window picked by the library): %.3f ms  phases=%s" % (
            name, lg, best * 1e3, {k: round(v, 3) for k, v in plan.timings().items()}), flush=True)
