#!/usr/bin/env python3
"""Wall time of the reference-shaped entry point mlhip_msm_g1 (host buffers in, affine point out): upload +
plan + kernels + host tail per call, n = 2^20 BLS12-381.  This is the PCIe-inclusive number of DESIGN.md section 6."""
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
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
g = load_golden(os.environ.get("MLHIP_PERF_CURVE", "BLS12-381"))
cid = g["curve_id"]
fpb, g1b, g2b, gtb = _lib.sizes(cid)
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
gen = torch.Generator(device=dev)
gen.manual_seed(3)


def rnd(k):
    return torch.randint(-(1 << 63), (1 << 63) - 1, (k, 4), dtype=torch.int64, generator=gen, device=dev).view(torch.uint8).reshape(k, 32).contiguous()


base = torch.frombuffer(bytearray(bytes.fromhex(g["g1_gen"])), dtype=torch.uint8).to(dev)
P = torch.empty(n * g1b, dtype=torch.uint8, device=dev)
_lib.check(lib.mlhip_scalar_mul_device(cid, 1, base.data_ptr(), 0, rnd(n).data_ptr(), 0, n, P.data_ptr(), st))
S = rnd(n)
torch.cuda.synchronize()
hp = P.cpu().numpy().tobytes()
hs = S.cpu().numpy().tobytes()
out = ctypes.create_string_buffer(g1b)
plan = _lib.MsmPlan(cid, 1, n, 16)
ref = plan.run(P.data_ptr(), S.data_ptr(), n, False, st)
# segments: "" = the library's own choice, 0 = one upload and one pass, K = streamed in K segments
for segs in ("", "0", "2", "4", "8", "16"):
    if segs:
        os.environ["MLHIP_STREAM_SEGMENTS"] = segs
    else:
        os.environ.pop("MLHIP_STREAM_SEGMENTS", None)
    res = []
    for rep in range(7):
        t0 = time.perf_counter()
        _lib.check(lib.mlhip_msm_g1(cid, hp, hs, 0, n, 16, out))
        res.append((time.perf_counter() - t0) * 1e3)
    print("mlhip_msm_g1 host-buffer call, n=2^%d, segments=%-7s: %s ms; matches resident-plan result: %s" % (
        n.bit_length() - 1, segs or "default", ", ".join("%.2f" % x for x in res), out.raw == ref), flush=True)

# resident bases: only the scalars travel
handle = ctypes.c_void_p()
_lib.check(lib.mlhip_bases_create(cid, 1, hp, n, 16, ctypes.byref(handle)))
for segs in ("", "0", "2", "4", "8"):
    if segs:
        os.environ["MLHIP_STREAM_SEGMENTS"] = segs
    else:
        os.environ.pop("MLHIP_STREAM_SEGMENTS", None)
    res = []
    for rep in range(7):
        t0 = time.perf_counter()
        _lib.check(lib.mlhip_bases_msm(handle, hs, 0, n, out))
        res.append((time.perf_counter() - t0) * 1e3)
    print("mlhip_bases_msm, n=2^%d, segments=%-7s: %s ms; matches: %s" % (
        n.bit_length() - 1, segs or "default", ", ".join("%.2f" % x for x in res), out.raw == ref), flush=True)
_lib.check(lib.mlhip_bases_destroy(handle))

# several host threads calling at once (goroutines on a shared SRS): wall time per MSM
import threading

os.environ.pop("MLHIP_STREAM_SEGMENTS", None)
for nthreads in (1, 2, 4):
    outs = [ctypes.create_string_buffer(g1b) for _ in range(nthreads)]
    reps = 6

    def work(i):
        for _ in range(reps):
            _lib.check(lib.mlhip_msm_g1(cid, hp, hs, 0, n, 16, outs[i]))

    for warm in range(2):
        ths = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt = (time.perf_counter() - t0) * 1e3
    print("mlhip_msm_g1 from %d host threads, n=2^%d: %.2f ms per MSM (%d calls in %.1f ms); all match: %s" % (
        nthreads, n.bit_length() - 1, dt / (reps * nthreads), reps * nthreads, dt, all(o.raw == ref for o in outs)), flush=True)

# G2 host-buffer call (BLS12-381: streamed like G1)
if len(sys.argv) > 2 and sys.argv[2] == "g2":
    base2 = torch.frombuffer(bytearray(bytes.fromhex(g["g2_gen"])), dtype=torch.uint8).to(dev)
    P2 = torch.empty(n * g2b, dtype=torch.uint8, device=dev)
    _lib.check(lib.mlhip_scalar_mul_device(cid, 2, base2.data_ptr(), 0, rnd(n).data_ptr(), 0, n, P2.data_ptr(), st))
    torch.cuda.synchronize()
    hp2 = P2.cpu().numpy().tobytes()
    out2 = ctypes.create_string_buffer(g2b)
    plan2 = _lib.MsmPlan(cid, 2, n, 16)
    ref2 = plan2.run(P2.data_ptr(), S.data_ptr(), n, False, st)
    for segs in ("", "0", "2", "4", "8", "16"):
        if segs:
            os.environ["MLHIP_STREAM_SEGMENTS"] = segs
        else:
            os.environ.pop("MLHIP_STREAM_SEGMENTS", None)
        res = []
        for rep in range(5):
            t0 = time.perf_counter()
            _lib.check(lib.mlhip_msm_g2(cid, hp2, hs, 0, n, 16, out2))
            res.append((time.perf_counter() - t0) * 1e3)
        print("mlhip_msm_g2 host-buffer call, n=2^%d, segments=%-7s: %s ms; matches resident-plan result: %s" % (
            n.bit_length() - 1, segs or "default", ", ".join("%.2f" % x for x in res), out2.raw == ref2), flush=True)
