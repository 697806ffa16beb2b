#!/usr/bin/env python3
"""Device-resident MSMs cut into tiles (msm_plan.h: plan_stream / resident_tiles): accumulation time against the tile size.
BLS12-381 G1 and G2, n = 2^22 .. 2^24, MLHIP_TILE_LOG2 = 0 (one pass) / 20 / 21 / 22 and the library's own choice."""
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
gen.manual_seed(11)


def rnd(k):
    return torch.randint(-(1 << 63), (1 << 63) - 1, (k, 4), dtype=torch.int64, generator=gen, device=dev).view(torch.uint8).reshape(k, 32).contiguous()


g = load_golden("BLS12-381")
cid = g["curve_id"]
fpb, g1b, g2b, gtb = _lib.sizes(cid)
lgs = [int(a) for a in sys.argv[1:]] or [22, 23, 24]
for group, sz, name in ((1, g1b, "G1"), (2, g2b, "G2")):
    base = torch.frombuffer(bytearray(bytes.fromhex(g["g1_gen" if group == 1 else "g2_gen"])), dtype=torch.uint8).to(dev)
    for lg in lgs:
        n = 1 << lg
        P = torch.empty(n * sz, dtype=torch.uint8, device=dev)
        _lib.check(lib.mlhip_scalar_mul_device(cid, group, base.data_ptr(), 0, rnd(n).data_ptr(), 0, n, P.data_ptr(), st))
        S = rnd(n)
        plan = _lib.MsmPlan(cid, group, n, 16)
        plan.set_profiling(True)
        ref = None
        for tile in ("0", "20", "21", "22", None):
            if tile is None:
                os.environ.pop("MLHIP_TILE_LOG2", None)
            else:
                os.environ["MLHIP_TILE_LOG2"] = tile
            best = None
            for rep in range(3):
                out = plan.run(P.data_ptr(), S.data_ptr(), n, False, st)
                t = plan.timings()
                if best is None or t["device_total"] < best["device_total"]:
                    best = t
            ref = ref or out
            print("BLS12-381 %s n=2^%d tile=%s: device %.2f ms  sort %.2f  accumulate %.2f (%.4f ns/add)  reduce %.2f  same=%s" % (
                name, lg, "library" if tile is None else ("off" if tile == "0" else "2^" + tile), best["device_total"],
                best["digits"] + best["sort"], best["accumulate"], best["accumulate"] * 1e6 / (n * 16), best["reduce"], out == ref), flush=True)
        plan.close()
        del P, S
        torch.cuda.empty_cache()
