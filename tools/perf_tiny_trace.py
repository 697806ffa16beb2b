#!/usr/bin/env python3
"""Twenty device-resident MSMs of n points (default 32) on one plan: for `rocprofv3 --kernel-trace --stats` -- how much of a
small MSM's device time is kernels and how much is the gaps between its ~20 launches."""
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
sys.path.insert(0, "tests")
from conftest import load_golden  # noqa: E402
from mathlib_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
lib = _lib.load()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
g = load_golden("BLS12-381")
cid = g["curve_id"]
fpb, g1b, g2b, gtb = _lib.sizes(cid)
gen = torch.Generator(device=dev)
gen.manual_seed(3)
rnd = lambda k: torch.randint(-(1 << 63), (1 << 63) - 1, (k, 4), dtype=torch.int64, generator=gen, device=dev).view(torch.uint8).reshape(k, 32).contiguous()  # noqa: E731
base = torch.frombuffer(bytearray(bytes.fromhex(g["g1_gen"])), dtype=torch.uint8).to(dev)
P = torch.empty(n * g1b, dtype=torch.uint8, device=dev)
_lib.check(lib.mlhip_scalar_mul_device(cid, 1, base.data_ptr(), 0, rnd(n).data_ptr(), 0, n, P.data_ptr(), st))
S = rnd(n)
plan = _lib.MsmPlan(cid, 1, n, 0)
plan.set_profiling(True)
tot = 0.0
for rep in range(20):
    plan.run(P.data_ptr(), S.data_ptr(), n, False, st)
    tot += plan.timings()["device_total"]
print("n=%d: device_total %.3f ms per MSM (HIP events)" % (n, tot / 20))
