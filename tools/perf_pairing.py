#!/usr/bin/env python3
"""Pairing-only timing (one curve, batch 65 536) for profiling the lane-pair kernels under rocprofv3."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden  # noqa: E402
from mathlib_amd import _lib  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "BLS12-381"
npair = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 16
lib = _lib.load()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
gen = torch.Generator(device=dev)
gen.manual_seed(11)
g = load_golden(name)
cid = g["curve_id"]
fpb, g1b, g2b, gtb = _lib.sizes(cid)


def rnd(n):
    return torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, generator=gen, device=dev).view(torch.uint8).reshape(n, 32).contiguous()


def points(group, n):
    sz = g1b if group == 1 else g2b
    base = torch.frombuffer(bytearray(bytes.fromhex(g["g1_gen" if group == 1 else "g2_gen"])), dtype=torch.uint8).to(dev)
    out = torch.empty(n * sz, dtype=torch.uint8, device=dev)
    _lib.check(lib.mlhip_scalar_mul_device(cid, group, base.data_ptr(), 0, rnd(n).data_ptr(), 0, n, out.data_ptr(), st))
    torch.cuda.synchronize()
    return out


P, Q = points(1, npair), points(2, npair)
out = torch.empty(npair * gtb, dtype=torch.uint8, device=dev)
for what, fn in (("miller", lambda: lib.mlhip_miller_loop_device(cid, P.data_ptr(), Q.data_ptr(), 1, npair, out.data_ptr(), st)),
                 ("final_exp", lambda: lib.mlhip_final_exp_device(cid, out.data_ptr(), npair, out.data_ptr(), st)),
                 ("pairing", lambda: lib.mlhip_pairing_batch_device(cid, P.data_ptr(), Q.data_ptr(), npair, out.data_ptr(), st))):
    best = None
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.check(fn())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    print("%s %s batch=%d: %.3f ms -> %.3e /s" % (name, what, npair, best * 1e3, npair / best), flush=True)
