#!/usr/bin/env python3
"""K MSMs of 2^20 BLS12-381 G1 pairs over ONE resident-bases handle with shifted-base tables, scalars resident
(mlhip_bases_msm_device): the workload the rocprofv3 passes of the table path trace (kernel trace / --pmc FETCH_SIZE /
--pmc WRITE_SIZE, each its own run).  Usage: run_tables_steps.py [steps = 8] [log2 n = 20]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
sys.path.insert(0, "tests")
from conftest import load_golden  # noqa: E402
from mathlib_amd import _lib  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lg = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = 1 << lg
lib = _lib.load()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
gen = torch.Generator(device=dev)
gen.manual_seed(31)
g = load_golden("BLS12-381")
cid = g["curve_id"]
fpb, g1b, g2b, gtb = _lib.sizes(cid)


def rnd(k):
    return torch.randint(-(1 << 63), (1 << 63) - 1, (k, 4), dtype=torch.int64, generator=gen, device=dev).view(torch.uint8).reshape(k, 32).contiguous()


base = torch.frombuffer(bytearray(bytes.fromhex(g["g1_gen"])), dtype=torch.uint8).to(dev)
P = torch.empty(n * g1b, dtype=torch.uint8, device=dev)
_lib.check(lib.mlhip_scalar_mul_device(cid, 1, base.data_ptr(), 0, rnd(n).data_ptr(), 0, n, P.data_ptr(), st))
S = rnd(n)
torch.cuda.synchronize()
h = ctypes.c_void_p()
_lib.check(lib.mlhip_bases_create_device(cid, 1, P.data_ptr(), n, 0, ctypes.byref(h)))
assert _lib.plan_timings(lib, lib.mlhip_bases_plan(h))["tables"] == 1.0
out = ctypes.create_string_buffer(g1b)
for _ in range(steps):
    _lib.check(lib.mlhip_bases_msm_device(h, S.data_ptr(), 0, n, st, out))
_lib.check(lib.mlhip_bases_destroy(h))
print("done", steps, "MSMs of 2^%d pairs over shifted-base tables" % lg)
