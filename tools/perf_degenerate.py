#!/usr/bin/env python3
"""MSM on degenerate scalar distributions (resident inputs, BLS12-381 G1, n = 2^20, c = 16): all scalars equal, all
ones (a plain sum of points), small scalars; against the uniform case."""
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
gen.manual_seed(8)


def rnd(k):
    return torch.randint(-(1 << 63), (1 << 63) - 1, (k, 4), dtype=torch.int64, generator=gen, device=dev).view(torch.uint8).reshape(k, 32).contiguous()


g = load_golden("BLS12-381")
cid = g["curve_id"]
fpb, g1b, g2b, gtb = _lib.sizes(cid)
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
base = torch.frombuffer(bytearray(bytes.fromhex(g["g1_gen"])), dtype=torch.uint8).to(dev)
P = torch.empty(n * g1b, dtype=torch.uint8, device=dev)
_lib.check(lib.mlhip_scalar_mul_device(cid, 1, base.data_ptr(), 0, rnd(n).data_ptr(), 0, n, P.data_ptr(), st))
cases = {"uniform": rnd(n)}
one = torch.zeros((n, 32), dtype=torch.uint8, device=dev)
one[:, 0] = 1
cases["all ones (sum of points)"] = one
cases["all equal"] = rnd(1).repeat(n, 1).contiguous()
small = rnd(n).clone()
small[:, 4:] = 0
cases["below 2^32"] = small
two = rnd(n).clone()
two[:, 1:] = 0
two[:, 0] &= 1
cases["bits (0/1)"] = two
group = 2 if (len(sys.argv) > 2 and sys.argv[2] == "g2") else 1
if group == 2:
    base2 = torch.frombuffer(bytearray(bytes.fromhex(g["g2_gen"])), dtype=torch.uint8).to(dev)
    P = torch.empty(n * g2b, dtype=torch.uint8, device=dev)
    _lib.check(lib.mlhip_scalar_mul_device(cid, 2, base2.data_ptr(), 0, rnd(n).data_ptr(), 0, n, P.data_ptr(), st))
    print("G2:")
plan = _lib.MsmPlan(cid, group, n, 16)
plan.set_profiling(True)
for name, S in cases.items():
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan.run(P.data_ptr(), S.data_ptr(), n, False, st)
        best = min(best, time.perf_counter() - t0)
    print("%-26s %.2f ms  phases=%s" % (name, best * 1e3, {k: round(v, 2) for k, v in plan.timings().items()}), flush=True)
