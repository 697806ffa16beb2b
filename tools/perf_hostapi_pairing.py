#!/usr/bin/env python3
"""Wall time of the host-buffer pairing entry point (upload + kernel + download per call) beside the device-pointer
one, BLS12-381, for a few batch sizes."""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden  # noqa: E402
from mathlib_amd import _lib  # noqa: E402

lib = _lib.load()
g = load_golden("BLS12-381")
cid = g["curve_id"]
fpb, g1b, g2b, gtb = _lib.sizes(cid)
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
gen = torch.Generator(device=dev)
gen.manual_seed(5)


def rnd(k):
    return torch.randint(-(1 << 63), (1 << 63) - 1, (k, 4), dtype=torch.int64, generator=gen, device=dev).view(torch.uint8).reshape(k, 32).contiguous()


nmax = 65536
b1 = torch.frombuffer(bytearray(bytes.fromhex(g["g1_gen"])), dtype=torch.uint8).to(dev)
b2 = torch.frombuffer(bytearray(bytes.fromhex(g["g2_gen"])), dtype=torch.uint8).to(dev)
P1 = torch.empty(nmax * g1b, dtype=torch.uint8, device=dev)
P2 = torch.empty(nmax * g2b, dtype=torch.uint8, device=dev)
_lib.check(lib.mlhip_scalar_mul_device(cid, 1, b1.data_ptr(), 0, rnd(nmax).data_ptr(), 0, nmax, P1.data_ptr(), st))
_lib.check(lib.mlhip_scalar_mul_device(cid, 2, b2.data_ptr(), 0, rnd(nmax).data_ptr(), 0, nmax, P2.data_ptr(), st))
torch.cuda.synchronize()
h1 = P1.cpu().numpy().tobytes()
h2 = P2.cpu().numpy().tobytes()
OUT = torch.empty(nmax * gtb, dtype=torch.uint8, device=dev)
for n in (256, 4096, 65536):
    out = ctypes.create_string_buffer(gtb * n)
    host = []
    for rep in range(5):
        t0 = time.perf_counter()
        _lib.check(lib.mlhip_pairing_batch(cid, h1, h2, n, out))
        host.append((time.perf_counter() - t0) * 1e3)
    devt = []
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.check(lib.mlhip_pairing_batch_device(cid, P1.data_ptr(), P2.data_ptr(), n, OUT.data_ptr(), st))
        torch.cuda.synchronize()
        devt.append((time.perf_counter() - t0) * 1e3)
    same = OUT[: n * gtb].cpu().numpy().tobytes() == out.raw
    print("pairing batch n=%6d: host-buffer call %s ms | device pointers %s ms | same bytes: %s" % (
        n, ", ".join("%.2f" % x for x in host), ", ".join("%.2f" % x for x in devt), same), flush=True)
