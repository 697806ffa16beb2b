#!/usr/bin/env python3
"""A/B on one box: the same 2^20-point MSM (G1; MLHIP_PERF_GROUP=2: G2) through two plans created under different
environment settings (e.g. MLHIP_ACC32=1 vs default), alternating, phases printed.  Usage: perf_ab.py [curve] [log2 n] [ENVVAR]"""
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
n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
var = sys.argv[3] if len(sys.argv) > 3 else "MLHIP_ACC32"
group = int(os.environ.get("MLHIP_PERF_GROUP", "1"))
lib = _lib.load()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
gen = torch.Generator(device=dev)
gen.manual_seed(11)
g = load_golden(name)
cid = g["curve_id"]
fpb, g1b, g2b, gtb = _lib.sizes(cid)


def rnd(k):
    return torch.randint(-(1 << 63), (1 << 63) - 1, (k, 4), dtype=torch.int64, generator=gen, device=dev).view(torch.uint8).reshape(k, 32).contiguous()


base = torch.frombuffer(bytearray(bytes.fromhex(g["g1_gen" if group == 1 else "g2_gen"])), dtype=torch.uint8).to(dev)
P = torch.empty(n * (g1b if group == 1 else g2b), dtype=torch.uint8, device=dev)
_lib.check(lib.mlhip_scalar_mul_device(cid, group, base.data_ptr(), 0, rnd(n).data_ptr(), 0, n, P.data_ptr(), st))
S = rnd(n)
torch.cuda.synchronize()
plans = {}
os.environ[var] = "1"
plans[var + "=1"] = _lib.MsmPlan(cid, group, n, 16)
del os.environ[var]
plans["default"] = _lib.MsmPlan(cid, group, n, 16)
res = {}
for k, pl in plans.items():
    pl.set_profiling(True)
    res[k] = pl.run(P.data_ptr(), S.data_ptr(), n, False, st)
print("results equal:", len(set(res.values())) == 1)
for rep in range(4):
    for k, pl in plans.items():
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            pl.run(P.data_ptr(), S.data_ptr(), n, False, st)
            ts.append(time.perf_counter() - t0)
        print("%-16s best %.3f ms  median %.3f ms  phases=%s" % (k, min(ts) * 1e3, sorted(ts)[2] * 1e3, {a: round(b, 3) for a, b in pl.timings().items()}), flush=True)
