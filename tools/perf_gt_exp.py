#!/usr/bin/env python3
"""Batched Gt.Exp: 65 536 exponentiations by 255-bit scalars, lane-pair kernel vs one-lane kernel."""
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
gen.manual_seed(2)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for name in ("BLS12-381", "BN254", "BLS12-377"):
    g = load_golden(name)
    cid = g["curve_id"]
    fpb, g1b, g2b, gtb = _lib.sizes(cid)
    base = torch.frombuffer(bytearray(bytes.fromhex(g["pairing"][1]["fexp"])), dtype=torch.uint8).to(dev)
    IN = base.repeat(n).contiguous()
    S = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, generator=gen, device=dev).view(torch.uint8).reshape(n, 32).contiguous()
    outs = {}
    for mode in ("0", "1"):
        os.environ["MLHIP_PAIRING_ONE_LANE"] = mode
        OUT = torch.empty(n * gtb, dtype=torch.uint8, device=dev)
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _lib.check(lib.mlhip_gt_exp_device(cid, IN.data_ptr(), S.data_ptr(), 0, n, OUT.data_ptr(), st))
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        outs[mode] = (best, OUT)
    print("%s Gt.Exp x %d: lane pairs %.1f ms (%.3e /s) | one lane %.1f ms | same bytes: %s" % (
        name, n, outs["0"][0] * 1e3, n / outs["0"][0], outs["1"][0] * 1e3, bool(torch.equal(outs["0"][1], outs["1"][1]))), flush=True)
os.environ.pop("MLHIP_PAIRING_ONE_LANE", None)
