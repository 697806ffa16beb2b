#!/usr/bin/env python3
"""BLS12-377 G1 MSM through one resident plan with and without the caller's subgroup promise (twisted Edwards bucket
sums, ed28.h, against XYZZ), alternating on one box, phases printed.  Usage: perf_edwards.py [log2 n] [window_c]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden  # noqa: E402
from mathlib_amd import _lib  # noqa: E402

n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 22)
c = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib = _lib.load()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
gen = torch.Generator(device=dev)
gen.manual_seed(11)
g = load_golden("BLS12-377")
cid = g["curve_id"]
fpb, g1b, g2b, gtb = _lib.sizes(cid)


def rnd(k):
    return torch.randint(-(1 << 63), (1 << 63) - 1, (k, 4), dtype=torch.int64, generator=gen, device=dev).view(torch.uint8).reshape(k, 32).contiguous()


base = torch.frombuffer(bytearray(bytes.fromhex(g["g1_gen"])), dtype=torch.uint8).to(dev)
P = torch.empty(n * g1b, dtype=torch.uint8, device=dev)
_lib.check(lib.mlhip_scalar_mul_device(cid, 1, base.data_ptr(), 0, rnd(n).data_ptr(), 0, n, P.data_ptr(), st))
S = rnd(n)
torch.cuda.synchronize()
plan = _lib.MsmPlan(cid, 1, n, c)
plan.set_profiling(True)
print("BLS12-377 G1, n = 2^%d, window %s" % (n.bit_length() - 1, plan.window()), flush=True)
ref = None
for rep in range(3):
    for trust in (False, True):
        plan.assume_srs(trust)
        plan.run(P.data_ptr(), S.data_ptr(), n, False, st)
        ts, ph = [], None
        for _ in range(8):
            t0 = time.perf_counter()
            out = plan.run(P.data_ptr(), S.data_ptr(), n, False, st)
            ts.append((time.perf_counter() - t0) * 1e3)
            ph = plan.timings()
        if ref is None:
            ref = out
        ts.sort()
        print("assume_srs=%-5s edwards=%d  median %.3f ms  min %.3f  | sort %.3f accumulate %.3f reduce %.3f host tail %.3f tiles %d | same result: %s" % (
            trust, ph["edwards"], ts[len(ts) // 2], ts[0], ph["sort"], ph["accumulate"], ph["reduce"], ph["host_tail"], ph["tiles"], out == ref), flush=True)
