#!/usr/bin/env python3
"""Wall time of mlhip_pairing_product (prod_i e(P_i, Q_i) with one shared final exponentiation), BLS12-381, host buffers."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.getcwd())
sys.path.insert(0, "tests")
from conftest import load_golden  # noqa: E402
from mathlib_amd import _lib  # noqa: E402
from oracle import cref  # noqa: E402

lib = _lib.load()
g = load_golden("BLS12-381")
cid = g["curve_id"]
fpb, g1b, g2b, gtb = _lib.sizes(cid)
for n in (7, 1000, 16384, 1 << 18):
    m = min(n, 4096)  # the oracle generates 4096 distinct pairs; larger products repeat them
    p1 = cref.gen_points(cid, 1, 11, 22, m) * (n // m)
    p2 = cref.gen_points(cid, 2, 33, 44, m) * (n // m)
    out = ctypes.create_string_buffer(gtb)
    ts = []
    for rep in range(4):
        t0 = time.perf_counter()
        _lib.check(lib.mlhip_pairing_product(cid, p1, p2, n, out))
        ts.append((time.perf_counter() - t0) * 1e3)
    print("pairing_product n=%d: %s ms" % (n, ", ".join("%.2f" % t for t in ts)), flush=True)
