#!/usr/bin/env python3
"""Resident bases with and without shifted-base tables (msm_fold.h), same box, alternating: mlhip_bases_msm (scalars from
host memory: protocol (b)) and mlhip_bases_msm_device (scalars resident), phases from the handle's plan, table build time.
Usage: perf_fold.py [curve] [log2 n] [digit widths ...]   (default BLS12-381 20 20); MLHIP_PERF_GROUP=2 for G2"""
import ctypes
import os
import statistics
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
sys.path.insert(0, "tests")
from conftest import load_golden  # noqa: E402
from mathlib_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
gen = torch.Generator(device=dev)
gen.manual_seed(29)
curve = sys.argv[1] if len(sys.argv) > 1 else "BLS12-381"
lg = int(sys.argv[2]) if len(sys.argv) > 2 else 20
widths = [int(a) for a in sys.argv[3:]] or [20]
group = int(os.environ.get("MLHIP_PERF_GROUP", "1"))
n = 1 << lg
g = load_golden(curve)
cid = g["curve_id"]
fpb, g1b, g2b, gtb = _lib.sizes(cid)


def rnd(k):
    return torch.randint(-(1 << 63), (1 << 63) - 1, (k, 4), dtype=torch.int64, generator=gen, device=dev).view(torch.uint8).reshape(k, 32).contiguous()


base = torch.frombuffer(bytearray(bytes.fromhex(g["g1_gen" if group == 1 else "g2_gen"])), dtype=torch.uint8).to(dev)
g1b = g1b if group == 1 else g2b
P = torch.empty(n * g1b, dtype=torch.uint8, device=dev)
_lib.check(lib.mlhip_scalar_mul_device(cid, group, base.data_ptr(), 0, rnd(n).data_ptr(), 0, n, P.data_ptr(), st))
torch.cuda.synchronize()
hp = P.cpu().numpy().tobytes()
S = rnd(n)
hs = np.ascontiguousarray(S.cpu().numpy())
out = ctypes.create_string_buffer(g1b)


pc = int(os.environ.get("MLHIP_PERF_PLAIN_C", "16"))


def make(tables, c):
    os.environ["MLHIP_BASES_TABLES"] = "1" if tables else "0"
    if tables:
        os.environ["MLHIP_FOLD_WINDOW"] = str(c)
    h = ctypes.c_void_p()
    t0 = time.perf_counter()
    _lib.check(lib.mlhip_bases_create(cid, group, hp, n, 0 if tables else c, ctypes.byref(h)))
    dt = (time.perf_counter() - t0) * 1e3
    _lib.check(lib.mlhip_msm_plan_set_profiling(lib.mlhip_bases_plan(h), 1))
    return h, dt


handles = [("plain c=%d" % pc, *make(False, pc))] + [("tables c=%d" % c, *make(True, c)) for c in widths]
ref = None
res = {name: {"b": [], "dev": [], "ph": None} for name, _, _ in handles}
for rep in range(7):
    for name, h, _ in handles:
        t0 = time.perf_counter()
        _lib.check(lib.mlhip_bases_msm(h, hs.ctypes.data, 0, n, out))
        tb = (time.perf_counter() - t0) * 1e3
        ref = ref or out.raw
        assert out.raw == ref, name
        t0 = time.perf_counter()
        _lib.check(lib.mlhip_bases_msm_device(h, S.data_ptr(), 0, n, st, out))
        td = (time.perf_counter() - t0) * 1e3
        assert out.raw == ref, name
        if rep >= 2:
            res[name]["b"].append(tb)
            res[name]["dev"].append(td)
            res[name]["ph"] = _lib.plan_timings(lib, lib.mlhip_bases_plan(h))
for name, h, dt in handles:
    r = res[name]
    ph = r["ph"]
    print("%s G%d 2^%d %-14s create %7.1f ms | host scalars (b) median %.3f min %.3f | resident scalars median %.3f min %.3f | digits+sort %.3f accumulate %.3f reduce %.3f host tail %.3f (digits per scalar %d)" % (
        curve, group, lg, name, dt, statistics.median(r["b"]), min(r["b"]), statistics.median(r["dev"]), min(r["dev"]),
        ph["digits"] + ph["sort"], ph["accumulate"], ph["reduce"], ph["host_tail"], ph["digits_per_scalar"]), flush=True)
    _lib.check(lib.mlhip_bases_destroy(h))
