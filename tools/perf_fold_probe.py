#!/usr/bin/env python3
"""Probe for window-shifted base tables (one bucket set for all digits of a scalar, the points of digit j read from a
table of 2^off(j) P_i): what would the accumulation cost when its W * n gathers go to a table of W * n rows (1.5-1.9 GB
for 2^20 bases) instead of re-reading 2^20 rows (117 MB, Infinity Cache) W times?  Emulated with the kernels as they are:
an MSM over W * 2^20 DISTINCT points whose scalars have one non-zero digit (window 0), so that the accumulation kernel
does exactly the additions, the bucket lengths and the gathers of the folded form.  The other windows' buckets are empty;
their cost is measured by the all-zero-scalar run and subtracted."""
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
sys.path.insert(0, "tests")
from conftest import load_golden  # noqa: E402
from mathlib_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
gen = torch.Generator(device=dev)
gen.manual_seed(23)
os.environ["MLHIP_TILE_LOG2"] = "0"  # one pass over all points


def rnd(k):
    return torch.randint(-(1 << 63), (1 << 63) - 1, (k, 4), dtype=torch.int64, generator=gen, device=dev).view(torch.uint8).reshape(k, 32).contiguous()


def small(k, bits):
    s = torch.zeros((k, 4), dtype=torch.int64, device=dev)
    if bits:
        s[:, 0] = torch.randint(1, 1 << bits, (k,), dtype=torch.int64, generator=gen, device=dev)
    return s.view(torch.uint8).reshape(k, 32).contiguous()


g = load_golden("BLS12-381")
cid = g["curve_id"]
fpb, g1b, g2b, gtb = _lib.sizes(cid)
base = torch.frombuffer(bytearray(bytes.fromhex(g["g1_gen"])), dtype=torch.uint8).to(dev)
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n0 = 1 << lg


def points(n):
    P = torch.empty(n * g1b, dtype=torch.uint8, device=dev)
    _lib.check(lib.mlhip_scalar_mul_device(cid, 1, base.data_ptr(), 0, rnd(n).data_ptr(), 0, n, P.data_ptr(), st))
    torch.cuda.synchronize()
    return P


def run(tag, P, S, n, c, adds):
    plan = _lib.MsmPlan(cid, 1, n, c)
    plan.set_profiling(True)
    best = None
    for rep in range(4):
        plan.run(P.data_ptr(), S.data_ptr(), n, False, st)
        t = plan.timings()
        if best is None or t["accumulate"] < best["accumulate"]:
            best = t
    plan.close()
    print("%-46s accumulate %.3f ms (%.4f ns/add)  sort %.2f  reduce %.3f  host tail %.3f" % (
        tag, best["accumulate"], best["accumulate"] * 1e6 / adds if adds else 0.0, best["digits"] + best["sort"], best["reduce"], best["host_tail"]), flush=True)
    return best["accumulate"]


P = points(n0)
run("today: c=16, 2^%d bases, 255-bit scalars" % lg, P, rnd(n0), n0, 16, 16 * n0)
del P
torch.cuda.empty_cache()
for c, W in ((20, 13), (19, 14), (18, 15)):
    n = W * n0
    P = points(n)
    t_empty = run("c=%d, %d x 2^%d rows, all scalars zero" % (c, W, lg), P, small(n, 0), n, c, 0)
    t = run("c=%d, %d x 2^%d rows, one digit each" % (c, W, lg), P, small(n, c - 1), n, c, n)
    # the folded form has 2^(c-1) buckets, all of them populated: of the W 2^(c-1) buckets here, (W - 1) 2^(c-1) are empty
    print("   folded estimate: %.3f ms (accumulate minus %d/%d of the empty-bucket run)" % (t - t_empty * (W - 1) / W, W - 1, W), flush=True)
    del P
    torch.cuda.empty_cache()
