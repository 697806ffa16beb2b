#!/usr/bin/env python3
"""Same-box A/B of the segment schedules of a host-buffer G1 MSM (msm_plan.h: stream_schedule): protocol (c) mlhip_msm_g1
(points and scalars from host memory) and protocol (b) mlhip_bases_msm (resident bases, scalars from host memory), beside
the resident MSM (a) of the same inputs.  MLHIP_STREAM_SEGMENTS=K = K equal segments (rounds 1-3), MLHIP_STREAM_SCHEDULE =
explicit weights, neither = the library's schedule.  Usage: perf_hostapi_schedule.py [log2 n = 20] [curve]"""
import ctypes
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden  # noqa: E402
from mathlib_amd import _lib  # noqa: E402

lib = _lib.load()
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log_n
g = load_golden(sys.argv[2] if len(sys.argv) > 2 else "BLS12-381")
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
plan = _lib.MsmPlan(cid, 1, n, 16 if log_n <= 21 else 0)
plan.set_profiling(True)
res = []
for _ in range(7):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ref = plan.run(P.data_ptr(), S.data_ptr(), n, False, st)
    res.append((time.perf_counter() - t0) * 1e3)
print("(a) resident plan, n=2^%d: median %.3f ms (%s); phases %s" % (log_n, statistics.median(res[2:]), ", ".join("%.2f" % x for x in res),
      {k: round(v, 3) for k, v in plan.timings().items()}), flush=True)
c = plan.window()[0]


def setenv(kind, val):
    os.environ.pop("MLHIP_STREAM_SEGMENTS", None)
    os.environ.pop("MLHIP_STREAM_SCHEDULE", None)
    if kind == "equal":
        os.environ["MLHIP_STREAM_SEGMENTS"] = val
    elif kind == "sched":
        os.environ["MLHIP_STREAM_SCHEDULE"] = val


variants_c = [("default", ""), ("equal", "4"), ("equal", "0"), ("equal", "5"), ("equal", "6"), ("equal", "8"), ("sched", "1,2,2,3,4,4"),
              ("sched", "5,6,7,7,7"), ("sched", "3,4,4,5"), ("sched", "4,5,5,5,6,7"), ("sched", "2,3,5,6"), ("sched", "1,1,2,3,4,5")]
variants_b = [("default", ""), ("equal", "4"), ("equal", "0"), ("sched", "3,13"), ("sched", "1,4,11"), ("sched", "4,12"), ("sched", "2,6,8")]
if len(sys.argv) > 3:  # a third argument keeps only the default and the round 1-3 form (quick A/B at other sizes)
    variants_c = variants_c[:3]
    variants_b = variants_b[:3]
rounds = 3  # alternate the variants: box drift shows up as spread inside a variant, not as a difference between them
acc = {("c",) + v: [] for v in variants_c}
acc.update({("b",) + v: [] for v in variants_b})
handle = ctypes.c_void_p()
_lib.check(lib.mlhip_bases_create(cid, 1, hp, n, c, ctypes.byref(handle)))
_lib.check(lib.mlhip_bases_msm(handle, hs, 0, n, out))  # the first call leaves the converted copy of the bases
ok = True
for r in range(rounds):
    for v in variants_c:
        setenv(*v)
        for rep in range(5):
            t0 = time.perf_counter()
            _lib.check(lib.mlhip_msm_g1(cid, hp, hs, 0, n, c, out))
            if rep >= 2:
                acc[("c",) + v].append((time.perf_counter() - t0) * 1e3)
        ok = ok and out.raw == ref
    for v in variants_b:
        setenv(*v)
        for rep in range(5):
            t0 = time.perf_counter()
            _lib.check(lib.mlhip_bases_msm(handle, hs, 0, n, out))
            if rep >= 2:
                acc[("b",) + v].append((time.perf_counter() - t0) * 1e3)
        ok = ok and out.raw == ref
_lib.check(lib.mlhip_bases_destroy(handle))
for k, ts in acc.items():
    print("(%s) %-8s %-16s median %.3f  min %.3f  max %.3f ms (%d calls)" % (k[0], k[1], k[2], statistics.median(ts), min(ts), max(ts), len(ts)), flush=True)
print("every result equals the resident plan's:", ok)
