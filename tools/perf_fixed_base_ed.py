#!/usr/bin/env python3
"""BLS12-377 G1, 2^20 products [s_i]G of ONE base: the fixed-base table path on the twisted Edwards additions (round 4:
k_fixed_base_ed, the base's subgroup membership taken from the table build's extra entry [r]P) against the XYZZ additions
(MLHIP_EDWARDS=0), alternating on one box, table built in the call / kept from the call before.  Same bytes required."""
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
g = load_golden("BLS12-377")
cid = g["curve_id"]
fpb, g1b, g2b, gtb = _lib.sizes(cid)
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
S = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, generator=gen, device=dev).view(torch.uint8).reshape(n, 32).contiguous()
base = torch.frombuffer(bytearray(bytes.fromhex(g["g1_gen"])), dtype=torch.uint8).to(dev)
outs = {}
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rnd in range(3):
    for ed in ("1", "0"):
        os.environ["MLHIP_EDWARDS"] = ed
        P = torch.empty(n * g1b, dtype=torch.uint8, device=dev)
        res = {}
        for cache in ("0", "1"):
            if cache == "0":
                os.environ["MLHIP_FB_CACHE"] = "0"
            else:
                os.environ.pop("MLHIP_FB_CACHE", None)
            ts = []
            for rep in range(4):
                ev0.record()
                _lib.check(lib.mlhip_scalar_mul_device(cid, 1, base.data_ptr(), 0, S.data_ptr(), 0, n, P.data_ptr(), st))
                ev1.record()
                torch.cuda.synchronize()
                ts.append(ev0.elapsed_time(ev1))
            res[cache] = min(ts[1:])
        outs[ed] = P.clone()
        print("round %d MLHIP_EDWARDS=%s: 2^%d products, table built in the call %.3f ms, table kept %.3f ms" % (rnd, ed, n.bit_length() - 1, res["0"], res["1"]), flush=True)
print("same bytes on both paths:", bool(torch.equal(outs["0"], outs["1"])))
