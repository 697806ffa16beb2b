#!/usr/bin/env python3
"""Best Pippenger window per problem size on this GPU (feeds pick_window in api.hip): resident inputs, best of 3."""
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
group = int(sys.argv[2]) if len(sys.argv) > 2 else 1
step = int(sys.argv[3]) if len(sys.argv) > 3 else 2  # sizes 2^lg0, 2^(lg0 + step), ...
lg0 = int(sys.argv[4]) if len(sys.argv) > 4 else 4
lib = _lib.load()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
gen = torch.Generator(device=dev)
gen.manual_seed(11)
g = load_golden(name)
cid = g["curve_id"]
fpb, g1b, g2b, gtb = _lib.sizes(cid)
sz = g1b if group == 1 else g2b


def rnd(k):
    return torch.randint(-(1 << 63), (1 << 63) - 1, (k, 4), dtype=torch.int64, generator=gen, device=dev).view(torch.uint8).reshape(k, 32).contiguous()


nmax = 1 << (22 if group == 1 else 20)
base = torch.frombuffer(bytearray(bytes.fromhex(g["g1_gen" if group == 1 else "g2_gen"])), dtype=torch.uint8).to(dev)
P = torch.empty(nmax * sz, dtype=torch.uint8, device=dev)
_lib.check(lib.mlhip_scalar_mul_device(cid, group, base.data_ptr(), 0, rnd(nmax).data_ptr(), 0, nmax, P.data_ptr(), st))
S = rnd(nmax)
torch.cuda.synchronize()
lg = lg0
while (1 << lg) <= nmax:
    n = 1 << lg
    res = {}
    for c in range(4, 21):
        if c > lg + 6:
            break
        try:
            plan = _lib.MsmPlan(cid, group, n, c)
        except Exception:
            continue
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            plan.run(P.data_ptr(), S.data_ptr(), n, False, st)
            best = min(best, time.perf_counter() - t0)
        plan.close()
        res[c] = best * 1e3
    bc = min(res, key=res.get)
    print("%s G%d n=2^%d: best c=%d (%.3f ms)  " % (name, group, lg, bc, res[bc]) + " ".join("c%d=%.3f" % (c, t) for c, t in sorted(res.items()) if t < 1.5 * res[bc]), flush=True)
    lg += step
