#!/usr/bin/env python3
"""Latency of single calls through the reference-shaped API (one MSM of 2 points = G1.Mul2, one Pairing, one FExp,
one Pairing2): the GPU path is a throughput device -- these are the numbers an integrator needs to decide which
calls to leave on the CPU driver."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden  # noqa: E402
from mathlib_amd.driver import Curve  # noqa: E402

for cid, name in ((1, "BLS12-381"), (0, "BN254")):
    c = Curve(cid)
    g = load_golden(name)
    co = g["g2_gen_coords"]
    g2 = c.NewG2FromCoords((int(co[0][0]), int(co[0][1])), (int(co[1][0]), int(co[1][1])))
    g1 = c.GenG1()
    a, b = c.NewZrFromInt(123456789), c.NewZrFromInt(987654321)

    def t(fn, reps=5):
        fn()
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            best = min(best, time.perf_counter() - t0)
        return best * 1e3

    print("%s  G1.Mul %.2f ms | MultiScalarMul(2) %.2f ms | Pairing %.2f ms | FExp %.2f ms | Pairing2 %.2f ms | FExp(Pairing) fused batch of 1: %.2f ms" % (
        name, t(lambda: g1.Mul(a)), t(lambda: c.MultiScalarMul([g1, g1], [a, b])), t(lambda: c.Pairing(g2, g1)),
        t(lambda: c.FExp(c.Pairing(g2, g1))) , t(lambda: c.Pairing2(g2, g2, g1, g1)), t(lambda: c.PairingBatch([g2], [g1]))), flush=True)
