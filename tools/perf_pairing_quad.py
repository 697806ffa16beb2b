#!/usr/bin/env python3
"""One pairing per quad of lanes against one per lane pair (MLHIP_PAIRING_QUAD=1 / 0), alternating on one box: fused
pairing, Miller loop, final exponentiation and Gt.Exp at batch sizes from one wave to a full chip.  Usage:
perf_pairing_quad.py [curve = BLS12-377]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.getcwd())
sys.path.insert(0, "tests")
from conftest import load_golden  # noqa: E402
from mathlib_amd import _lib  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "BLS12-377"
lib = _lib.load()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
g = load_golden(name)
cid = g["curve_id"]
fpb, g1b, g2b, gtb = _lib.sizes(cid)
gen = torch.Generator(device=dev)
gen.manual_seed(5)
nmax = 1 << 16
S = torch.randint(-(1 << 63), (1 << 63) - 1, (nmax, 4), dtype=torch.int64, generator=gen, device=dev).view(torch.uint8).reshape(nmax, 32).contiguous()
P = torch.empty(nmax * g1b, dtype=torch.uint8, device=dev)
Q = torch.empty(nmax * g2b, dtype=torch.uint8, device=dev)
for grp, key, dst in ((1, "g1_gen", P), (2, "g2_gen", Q)):
    base = torch.frombuffer(bytearray(bytes.fromhex(g[key])), dtype=torch.uint8).to(dev)
    _lib.check(lib.mlhip_scalar_mul_device(cid, grp, base.data_ptr(), 0, S.data_ptr(), 0, nmax, dst.data_ptr(), st))
torch.cuda.synchronize()
ML = torch.empty(nmax * gtb, dtype=torch.uint8, device=dev)
OUT = torch.empty(nmax * gtb, dtype=torch.uint8, device=dev)
os.environ["MLHIP_PAIRING_QUAD"] = "0"
_lib.check(lib.mlhip_miller_loop_device(cid, P.data_ptr(), Q.data_ptr(), 1, nmax, ML.data_ptr(), st))
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ops = {
    "pairing": lambda n: lib.mlhip_pairing_batch_device(cid, P.data_ptr(), Q.data_ptr(), n, OUT.data_ptr(), st),
    "miller": lambda n: lib.mlhip_miller_loop_device(cid, P.data_ptr(), Q.data_ptr(), 1, n, OUT.data_ptr(), st),
    "final_exp": lambda n: lib.mlhip_final_exp_device(cid, ML.data_ptr(), n, OUT.data_ptr(), st),
    "gt_exp": lambda n: lib.mlhip_gt_exp_device(cid, ML.data_ptr(), S.data_ptr(), 0, n, OUT.data_ptr(), st),
}
ref = {}
for n in (1, 64, 1024, 4096, 16384, 32768, 65536):
    for op, fn in ops.items():
        res = {}
        for rnd in range(2):
            for quad in ("1", "0"):
                os.environ["MLHIP_PAIRING_QUAD"] = quad
                ts = []
                for rep in range(3):
                    ev0.record()
                    _lib.check(fn(n))
                    ev1.record()
                    torch.cuda.synchronize()
                    ts.append(ev0.elapsed_time(ev1))
                res.setdefault(quad, []).append(min(ts[1:]))
                key = (op, n)
                got = OUT[: n * gtb].clone()
                if op != "miller":  # raw Miller values are not canonical: compared only through the other ops
                    if key in ref:
                        assert torch.equal(ref[key], got), (op, n, quad)
                    ref[key] = got
        print("%s %-9s n=%-6d quads %.3f ms   lane pairs %.3f ms" % (name, op, n, statistics.mean(res["1"]), statistics.mean(res["0"])), flush=True)
print("every quad result equals the lane pairs' (pairing, final_exp, gt_exp)")
