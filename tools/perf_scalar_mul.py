#!/usr/bin/env python3
"""Batched single-scalar multiplication (G1.Mul / G2.Mul for many scalars), one MI355X.
One common base (stride 0): the fixed-base table path against the double-and-add kernel (MLHIP_FIXED_BASE_MIN=0);
a base per scalar (stride 1): always double-and-add."""
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
gen.manual_seed(1)


def rnd(k):
    return torch.randint(-(1 << 63), (1 << 63) - 1, (k, 4), dtype=torch.int64, generator=gen, device=dev).view(torch.uint8).reshape(k, 32).contiguous()


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


for name in ("BLS12-381", "BN254", "BLS12-377"):
    g = load_golden(name)
    cid = g["curve_id"]
    fpb, g1b, g2b, gtb = _lib.sizes(cid)
    for group, size, key, lgs in ((1, g1b, "g1_gen", (12, 17, 20)), (2, g2b, "g2_gen", (12, 17, 20))):
        base = torch.frombuffer(bytearray(bytes.fromhex(g[key])), dtype=torch.uint8).to(dev)
        for lg in lgs:
            n = 1 << lg
            S = rnd(n)
            P = torch.empty(n * size, dtype=torch.uint8, device=dev)
            Q = torch.empty(n * size, dtype=torch.uint8, device=dev)
            os.environ.pop("MLHIP_FIXED_BASE_MIN", None)
            run = lambda: _lib.check(lib.mlhip_scalar_mul_device(cid, group, base.data_ptr(), 0, S.data_ptr(), 0, n, P.data_ptr(), st))  # noqa: E731
            os.environ["MLHIP_FB_CACHE"] = "0"
            t_build = timed(run)
            os.environ.pop("MLHIP_FB_CACHE", None)
            run()
            t_tab = timed(run)
            os.environ["MLHIP_FIXED_BASE_MIN"] = "0"
            t_dbl = timed(lambda: _lib.check(lib.mlhip_scalar_mul_device(cid, group, base.data_ptr(), 0, S.data_ptr(), 0, n, Q.data_ptr(), st)), reps=2)
            print("%s G%d one base, 2^%d scalars: table path %.2f ms with the table built in the call, %.2f ms (%.3e /s) with the table of an earlier call | double-and-add %.2f ms | same bytes: %s" % (
                name, group, lg, t_build * 1e3, t_tab * 1e3, n / t_tab, t_dbl * 1e3, bool(torch.equal(P, Q))), flush=True)
            os.environ.pop("MLHIP_FIXED_BASE_MIN", None)
            if lg == 20 and name == "BLS12-381":  # the window width (MLHIP_FB_WINDOW; default 12)
                line = []
                for w in ((8, 10, 11, 12, 13) if group == 1 else (8, 10, 11, 12)):
                    os.environ["MLHIP_FB_WINDOW"] = str(w)
                    os.environ["MLHIP_FB_CACHE"] = "0"
                    tb = timed(run)
                    os.environ.pop("MLHIP_FB_CACHE", None)
                    run()
                    tc = timed(run)
                    line.append("w=%d: %.2f / %.2f ms" % (w, tb * 1e3, tc * 1e3))
                    assert torch.equal(P, Q)
                os.environ.pop("MLHIP_FB_WINDOW", None)
                print("    window widths (built in the call / kept): " + "   ".join(line), flush=True)
    os.environ.pop("MLHIP_FIXED_BASE_MIN", None)
