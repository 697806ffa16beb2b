#!/usr/bin/env python3
"""Where MinDeviceMSM (go/driver/hip/hip.go) should sit: the reference-shaped host-slice call on the device
(mlhip_msm_g1: upload, kernels, host tail) against the C restatement (oracle/cref -- the stated stand-in for the
reference's CPU driver, which cannot be built here; gnark's MultiExp has ADX assembly and uses every core) with 1, 8 and
all threads of the box.  Run on the GPU box: python tools/perf_min_device_msm.py"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mathlib_amd import _lib  # noqa: E402
from oracle import cref  # noqa: E402

lib = _lib.load()
cores = len(os.sched_getaffinity(0))
for cid, name in ((1, "BLS12-381"), (0, "BN254")):
    _, g1b, _, _ = _lib.sizes(cid)
    for lg in (1, 3, 5, 6, 7, 8, 9, 10, 12, 14):
        n = 1 << lg
        pts = cref.gen_points(cid, 1, 5, 7, n)
        sc = np.random.default_rng(1).integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
        scb = sc.tobytes()
        row = []
        for threads in (1, 8, min(cores, 64)):
            best = 1e9
            for _ in range(5):
                t = time.perf_counter()
                ref = cref.msm(cid, 1, pts, sc, n, False, 0, threads)
                best = min(best, time.perf_counter() - t)
            row.append("%d thr %.3f ms" % (threads, best * 1e3))
        out = ctypes.create_string_buffer(g1b)
        best = 1e9
        for _ in range(8):
            t = time.perf_counter()
            _lib.check(lib.mlhip_msm_g1(cid, pts, scb, 0, n, 0, out))
            best = min(best, time.perf_counter() - t)
        assert out.raw == ref
        print("%s G1 MSM n=2^%-2d  device (host slices) %.3f ms | oracle/cref %s" % (name, lg, best * 1e3, ", ".join(row)), flush=True)
