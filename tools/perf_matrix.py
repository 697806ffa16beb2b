#!/usr/bin/env python3
"""Timing matrix on one MI355X for the configurations bench.py does not headline: G1/G2 MSM and the pairing
batch on all three curves (inputs made on the device by the batched scalar-mul kernel).  Prints one line each."""
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


def points(cid, group, n, base_raw):
    fpb, g1b, g2b, gtb = _lib.sizes(cid)
    sz = g1b if group == 1 else g2b
    base = torch.frombuffer(bytearray(base_raw), dtype=torch.uint8).to(dev)
    out = torch.empty(n * sz, dtype=torch.uint8, device=dev)
    _lib.check(lib.mlhip_scalar_mul_device(cid, group, base.data_ptr(), 0, rnd(n).data_ptr(), 0, n, out.data_ptr(), st))
    torch.cuda.synchronize()
    return out


for name in ("BLS12-381", "BN254", "BLS12-377"):
    g = load_golden(name)
    cid = g["curve_id"]
    fpb, g1b, g2b, gtb = _lib.sizes(cid)
    g1 = bytes.fromhex(g["g1_gen"])
    g2 = bytes.fromhex(g["g2_gen"])
    for group, n, c in ((1, 1 << 20, 16), (1, 1 << 22, 16), (2, 1 << 20, 16)) if name == "BLS12-381" else ((1, 1 << 20, 16), (2, 1 << 18, 16)):
        t0 = time.time()
        P = points(cid, group, n, g1 if group == 1 else g2)
        tgen = time.time() - t0
        S = rnd(n)
        plan = _lib.MsmPlan(cid, group, n, c)
        plan.set_profiling(True)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            plan.run(P.data_ptr(), S.data_ptr(), n, False, st)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        print("%s G%d MSM n=2^%d c=%d: %.3f ms -> %.3e scalar-muls/s  phases=%s  (input gen %.2fs)" % (
            name, group, n.bit_length() - 1, c, best * 1e3, n / best, {k: round(v, 3) for k, v in plan.timings().items()}, tgen), flush=True)
        plan.close()
        del P, S
    npair = 1 << 16
    P = points(cid, 1, npair, g1)
    Q = points(cid, 2, npair, g2)
    out = torch.empty(npair * gtb, dtype=torch.uint8, device=dev)
    for what, fn in (("miller", lambda: lib.mlhip_miller_loop_device(cid, P.data_ptr(), Q.data_ptr(), 1, npair, out.data_ptr(), st)),
                     ("final_exp", lambda: lib.mlhip_final_exp_device(cid, out.data_ptr(), npair, out.data_ptr(), st)),
                     ("pairing", lambda: lib.mlhip_pairing_batch_device(cid, P.data_ptr(), Q.data_ptr(), npair, out.data_ptr(), st))):
        best = None
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _lib.check(fn())
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        print("%s %s batch=%d: %.3f ms -> %.3e /s" % (name, what, npair, best * 1e3, npair / best), flush=True)
    del P, Q, out
