#!/usr/bin/env python3
"""Where MinDeviceMSM (go/driver/hip/hip.go) should sit: one CPU thread of the C restatement (oracle/cref -- the stated
stand-in for the reference's CPU driver, which cannot be built here) against the device time of the same host-slice MSM
(profiles/r01_perf_small_msm.txt / tools/perf_small_msm.py).  CPU-only; run anywhere."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cref  # noqa: E402

for cid, name in ((1, "BLS12-381"), (0, "BN254"), (2, "BLS12-377")):
    for lg in (1, 3, 5, 6, 7, 8, 9, 10, 12):
        n = 1 << lg
        pts = cref.gen_points(cid, 1, 5, 7, n)
        sc = np.random.default_rng(1).integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
        best = 1e9
        for _ in range(5):
            t = time.perf_counter()
            cref.msm(cid, 1, pts, sc, n, False, 0, 1)
            best = min(best, time.perf_counter() - t)
        print("%s G1 MSM n=2^%d, oracle/cref, 1 thread: %.3f ms" % (name, lg, best * 1e3), flush=True)
