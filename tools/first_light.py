#!/usr/bin/env python3
"""First-light timing probe on one MI355X (not the bench): phase timings of the G1 MSM at a few
sizes, batched pairing rate, raw Fp multiply rate.  Inputs: the 1000 golden points tiled (so a
size-independent check exists: MSM(tiled points, s) == MSM(1000 points, per-point scalar sums))."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, load_msm1000  # noqa: E402
from mathlib_amd import _lib  # noqa: E402

lib = _lib.load()
curve = "BLS12-381"
g = load_golden(curve)
cid = g["curve_id"]
r = int(g["r"], 16)
fpb, g1b, g2b, gtb = _lib.sizes(cid)
pts, _, _ = load_msm1000(curve, fpb)
pts_np = np.frombuffer(pts, dtype=np.uint8).reshape(1000, g1b)
st = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(1)


def run(n, c, reps=3):
    idx = np.arange(n) % 1000
    P = torch.from_numpy(np.ascontiguousarray(pts_np[idx])).cuda()
    sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    sc[:, 3] &= (1 << 62) - 1  # < 2^254 < r
    S = torch.from_numpy(sc.view(np.uint8).reshape(n, 32)).cuda()
    plan = _lib.MsmPlan(cid, _lib.GROUP_G1, n, c)
    plan.set_profiling(True)
    out = None
    for k in range(reps):
        t0 = time.time()
        out = plan.run(P.data_ptr(), S.data_ptr(), n, False, st)
        dt = time.time() - t0
        print("n=2^%d c=%d rep%d wall=%.3f ms %s" % (int(np.log2(n)), c, k, dt * 1e3, {k_: round(v, 3) for k_, v in plan.timings().items()}), flush=True)
    # linearity check against the 1000-point path
    ints = [int.from_bytes(sc[i].tobytes(), "little") for i in range(n)] if n <= (1 << 18) else None
    if ints is not None:
        sums = [0] * 1000
        for i, v in enumerate(ints):
            sums[i % 1000] = (sums[i % 1000] + v) % r
        ref = ctypes.create_string_buffer(g1b)
        _lib.check(lib.mlhip_msm_g1(cid, pts, b"".join(s.to_bytes(32, "little") for s in sums), 0, 1000, 10, ref))
        print("   linearity check vs 1000-point MSM:", "OK" if ref.raw == out else "MISMATCH", flush=True)
    plan.close()


for n, c in ((1 << 14, 12), (1 << 16, 14), (1 << 18, 16), (1 << 20, 16)):
    run(n, c)

# pairing batch
pg = g["pairing"]
for n in (256, 4096, 16384):
    g1 = torch.from_numpy(np.frombuffer(b"".join(bytes.fromhex(pg[i % 4]["g1"]) for i in range(n)), dtype=np.uint8).copy()).cuda()
    g2 = torch.from_numpy(np.frombuffer(b"".join(bytes.fromhex(pg[i % 4]["g2"]) for i in range(n)), dtype=np.uint8).copy()).cuda()
    out = torch.empty(n * gtb, dtype=torch.uint8, device="cuda")
    for k in range(2):
        torch.cuda.synchronize()
        t0 = time.time()
        _lib.check(lib.mlhip_pairing_batch_device(cid, g1.data_ptr(), g2.data_ptr(), n, out.data_ptr(), st))
        torch.cuda.synchronize()
        dt = time.time() - t0
        print("pairing batch n=%d rep%d %.3f ms -> %.0f pairings/s" % (n, k, dt * 1e3, n / dt), flush=True)
    o = out.cpu().numpy().tobytes()
    ok = all(o[i * gtb : (i + 1) * gtb] == bytes.fromhex(pg[i % 4]["fexp"]) for i in range(0, n, max(1, n // 64)))
    print("   parity vs golden:", "OK" if ok else "MISMATCH", flush=True)

# fp_mul rate
n = 1 << 22
a = torch.from_numpy(np.frombuffer(bytes.fromhex(g["fp_mul"][0]["a"]) * n, dtype=np.uint8).copy()).cuda()
b = torch.from_numpy(np.frombuffer(bytes.fromhex(g["fp_mul"][0]["b"]) * n, dtype=np.uint8).copy()).cuda()
o = torch.empty_like(a)
for rep in (1, 64):
    for k in range(2):
        torch.cuda.synchronize()
        t0 = time.time()
        _lib.check(lib.mlhip_fp_mul_device(cid, a.data_ptr(), b.data_ptr(), n, rep, o.data_ptr(), st))
        torch.cuda.synchronize()
        dt = time.time() - t0
        print("fp_mul n=%d repeat=%d: %.3f ms -> %.3e Fp mul/s" % (n, rep, dt * 1e3, n * rep / dt), flush=True)
