#!/usr/bin/env python3
"""Randomised soak against the C oracle: MSMs (all curves, G1 and G2, random n / window / scalar widths, duplicated
and negated points, infinities), fixed-base batched Mul (random window width, kept / rebuilt tables), device-resident plans (one pass / tiles / the shared-scalar G1+G2 launch) and pairing batches, for a given number of
seconds.  Prints a progress line every
20 s; exits non-zero on the first mismatch."""
import ctypes
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden  # noqa: E402
from mathlib_amd import _lib  # noqa: E402
from oracle import cref  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lib = _lib.load()
rnd = random.Random(seed)
names = ["BN254", "BLS12-381", "BLS12-377"]
t0 = last = time.time()
done = {"msm": 0, "streamed": 0, "pairing": 0}
while time.time() - t0 < budget:
    name = rnd.choice(names)
    g = load_golden(name)
    cid = g["curve_id"]
    fpb, g1b, g2b, gtb = _lib.sizes(cid)
    r_order = int(g["r"], 16)
    if rnd.random() < 0.12:
        # one base, many scalars (mlhip_scalar_mul, stride 0): the fixed-base table path with a random window width, the table
        # of an earlier call kept or rebuilt, one of three bases per curve and group so that calls hit and miss the kept table
        group = 2 if rnd.random() < 0.35 else 1
        sz = g1b if group == 1 else g2b
        gen = bytes.fromhex(g["g1_gen" if group == 1 else "g2_gen"])
        base = gen if rnd.random() < 0.5 else cref.point_mul(cid, group, gen, rnd.choice([2, 0xABCDEF]))
        n = rnd.choice([1, 2, 63, 64, 65, 300, 1000, 4097])
        bits = rnd.choice([8, 64, 200, 253, 256])
        vals = [rnd.getrandbits(bits) % r_order for _ in range(n)]
        vals[rnd.randrange(n)] = rnd.choice([0, 1, r_order - 1, r_order // 2 + 1])
        os.environ["MLHIP_FIXED_BASE_MIN"] = "1"
        w = rnd.choice([None, None, 4, 7, 8, 10, 11, 13, 14])
        if w:
            os.environ["MLHIP_FB_WINDOW"] = str(w)
        if rnd.random() < 0.25:
            os.environ["MLHIP_FB_CACHE"] = "0"
        if rnd.random() < 0.3:  # BLS12-377 G1: the XYZZ products instead of the Edwards ones (round 4), on the same kept table
            os.environ["MLHIP_EDWARDS"] = "0"
        out = ctypes.create_string_buffer(sz * n)
        _lib.check(lib.mlhip_scalar_mul(cid, group, base, 0, b"".join(v.to_bytes(32, "little") for v in vals), 0, n, out))
        for k in ("MLHIP_FIXED_BASE_MIN", "MLHIP_FB_WINDOW", "MLHIP_FB_CACHE", "MLHIP_EDWARDS"):
            os.environ.pop(k, None)
        for i in {0, n - 1, rnd.randrange(n), rnd.randrange(n)}:
            if out.raw[i * sz : (i + 1) * sz] != cref.point_mul(cid, group, base, vals[i]):
                print("MISMATCH fixed-base mul", name, group, n, i, "window", w, "seed", seed, flush=True)
                sys.exit(1)
        done["fixed_base"] = done.get("fixed_base", 0) + 1
    elif rnd.random() < 0.8:
        group = 2 if rnd.random() < 0.25 else 1
        sz = g1b if group == 1 else g2b
        n = rnd.choice([1, 2, 5, 33, 100, 257, 1000, 1025, 3000, 5000, 20000, 70000, 300000]) if rnd.random() < 0.7 else rnd.randrange(1, 40000)
        if group == 2:
            n = min(n, 8000)
        c = rnd.choice([0, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 20])  # every width: windows differ by <= 1 bit
        pts = bytearray(cref.gen_points(cid, group, rnd.getrandbits(40), rnd.getrandbits(40), n))
        bits = rnd.choice([8, 32, 128, 200, 253])
        vals = [rnd.getrandbits(bits) % r_order for _ in range(n)]
        for k in range(n // 4):
            i, j = rnd.randrange(n), rnd.randrange(n)
            pts[j * sz : (j + 1) * sz] = pts[i * sz : (i + 1) * sz]
            vals[j] = vals[i] if k % 2 == 0 else (r_order - vals[i]) % r_order
        for k in range(n // 50):
            j = rnd.randrange(n)
            pts[j * sz : (j + 1) * sz] = bytes(sz)
        sc = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in vals), dtype=np.uint64).reshape(n, 4).copy()
        pts = bytes(pts)
        exp = cref.msm(cid, group, pts, sc, n, False, 0, 16)
        out = ctypes.create_string_buffer(sz)
        fn = lib.mlhip_msm_g1 if group == 1 else lib.mlhip_msm_g2
        # half of the calls stream the pairs in a random number of segments (the library reads the switch per call)
        segs = rnd.choice([0, 0, 0, 2, 3, 5, 16])
        os.environ["MLHIP_STREAM_SEGMENTS"] = str(segs)
        # round 4: segments of unequal length (msm_plan.h: stream_schedule; it takes precedence over the count above)
        os.environ.pop("MLHIP_STREAM_SCHEDULE", None)
        if rnd.random() < 0.3:
            os.environ["MLHIP_STREAM_SCHEDULE"] = ",".join(str(rnd.randrange(1, 9)) for _ in range(rnd.randrange(2, 8)))
        # round-3 paths, read per launch: the coarse scatter staged in LDS (default) or one store per entry; the next
        # tile's sort on a second stream (default) or in line
        os.environ["MLHIP_SCATTER_STAGED"] = rnd.choice(["1", "1", "0"])
        os.environ["MLHIP_SORT_AHEAD"] = rnd.choice(["1", "1", "0"])
        mode = rnd.random()
        if mode < 0.2:  # device-resident inputs through a plan: one pass or tiles (MLHIP_TILE_LOG2 is read per launch)
            import torch

            dev = torch.device("cuda", 0)
            st = torch.cuda.current_stream().cuda_stream
            tile = rnd.choice([0, 0, 7, 8, 9, 10, 11, 12])
            os.environ["MLHIP_TILE_LOG2"] = str(tile)
            dp = torch.frombuffer(bytearray(pts), dtype=torch.uint8).to(dev)
            ds = torch.frombuffer(bytearray(sc.tobytes()), dtype=torch.uint8).to(dev)
            cc = c if c else 12
            plan = _lib.MsmPlan(cid, group, n, cc)
            # BLS12-377 G1: half of the plans carry the SRS promise (the points are multiples of the generator): bucket sums
            # and reduction in twisted Edwards coordinates when the bucket set is large enough, twice through the same
            # plan so that the second launch runs on the kept conversion
            srs = name == "BLS12-377" and group == 1 and rnd.random() < 0.5
            if srs:
                plan.assume_srs(True)
                plan.set_profiling(True)
            if srs or rnd.random() < 0.5:
                got = plan.run(dp.data_ptr(), ds.data_ptr(), n, False, st)
                if srs:
                    if plan.timings().get("edwards") == 1.0:
                        done["edwards"] = done.get("edwards", 0) + 1
                    if plan.run(dp.data_ptr(), ds.data_ptr(), n, False, st) != got:
                        print("MISMATCH second launch on the kept Edwards conversion", n, cc, bits, "tile", tile, "seed", seed, flush=True)
                        sys.exit(1)
            else:  # the G1 and the G2 MSM of the same scalars, sorted once
                og = 3 - group
                osz = g1b if og == 1 else g2b
                opts = cref.gen_points(cid, og, rnd.getrandbits(40), rnd.getrandbits(40), n)
                dop = torch.frombuffer(bytearray(opts), dtype=torch.uint8).to(dev)
                other = _lib.MsmPlan(cid, og, n, cc if rnd.random() < 0.8 else max(4, cc - 1))
                a, b = (plan, other) if group == 1 else (other, plan)
                pa, pb = (dp, dop) if group == 1 else (dop, dp)
                a.launch_shared(b, pa.data_ptr(), pb.data_ptr(), ds.data_ptr(), n, False, st)
                ra, rb = a.finish(), b.finish()
                got, got_other = (ra, rb) if group == 1 else (rb, ra)
                if got_other != cref.msm(cid, og, opts, sc, n, False, 0, 16):
                    print("MISMATCH shared msm (other group)", name, og, n, cc, bits, "tile", tile, "seed", seed, flush=True)
                    sys.exit(1)
                other.close()
                done["shared"] = done.get("shared", 0) + 1
            plan.close()
            os.environ.pop("MLHIP_TILE_LOG2", None)
            out = ctypes.create_string_buffer(got, sz)
            done["resident"] = done.get("resident", 0) + 1
            if tile:
                done["tiled"] = done.get("tiled", 0) + 1
        elif mode < 0.32:  # the device-list entry point: 1-4 shards, all on device 0
            nd = rnd.randrange(1, 5)
            devs = (ctypes.c_int * nd)(*([0] * nd))
            _lib.check(lib.mlhip_msm_multi(cid, group, devs, nd, pts, sc.tobytes(), 0, n, c, out))
            done["multi"] = done.get("multi", 0) + 1
        elif mode < 0.50:  # a resident table (BLS12-377 G1: checked on the device, then Edwards bucket sums)
            # half of them with shifted-base tables (msm_fold.h) of a random digit width and tile length, G1 and G2
            shifted = rnd.random() < 0.5
            fw, ft = rnd.choice([5, 8, 11, 13, 16, 17, 19, 20]), rnd.choice([4, 6, 9, 12, 20])
            if shifted:
                while -(-n // (1 << ft)) > 24:  # at most MLHIP_MAX_SEGMENTS tiles
                    ft += 1
                os.environ.update({"MLHIP_BASES_TABLES": "1", "MLHIP_FOLD_WINDOW": str(fw), "MLHIP_FOLD_TILE_LOG2": str(ft)})
            else:
                os.environ["MLHIP_BASES_TABLES"] = "0"
            h = ctypes.c_void_p()
            _lib.check(lib.mlhip_bases_create(cid, group, pts, n, 0 if shifted else c, ctypes.byref(h)))
            for k2 in ("MLHIP_BASES_TABLES", "MLHIP_FOLD_WINDOW", "MLHIP_FOLD_TILE_LOG2"):
                os.environ.pop(k2, None)
            if shifted and _lib.plan_timings(lib, lib.mlhip_bases_plan(h)).get("tables") != 1.0:
                print("NO TABLES on a forced handle", name, group, n, fw, ft, "seed", seed, flush=True)
                sys.exit(1)
            k = rnd.randrange(1, n + 1)
            _lib.check(lib.mlhip_bases_msm(h, sc.tobytes(), 0, k, out))
            if k != n and out.raw != cref.msm(cid, group, pts, sc, k, False, 0, 16):
                print("MISMATCH bases prefix", name, group, n, k, c, bits, "shifted", shifted, fw, ft, "seed", seed, flush=True)
                sys.exit(1)
            if rnd.random() < 0.5:
                _lib.check(lib.mlhip_bases_msm(h, sc.tobytes(), 0, n, out))
            else:  # scalars already on the device
                import torch

                ds = torch.frombuffer(bytearray(sc.tobytes()), dtype=torch.uint8).to(torch.device("cuda", 0))
                _lib.check(lib.mlhip_bases_msm_device(h, ds.data_ptr(), 0, n, torch.cuda.current_stream().cuda_stream, out))
            if name == "BLS12-377" and group == 1:
                done["tables_checked"] = done.get("tables_checked", 0) + lib.mlhip_bases_checked_subgroup(h)
            _lib.check(lib.mlhip_bases_destroy(h))
            done["bases"] = done.get("bases", 0) + 1
            if shifted:
                done["shifted_tables"] = done.get("shifted_tables", 0) + 1
        else:
            _lib.check(fn(cid, pts, sc.tobytes(), 0, n, c, out))
        if out.raw != exp:
            print("MISMATCH msm", name, group, n, c, bits, "segments", segs, "seed", seed, flush=True)
            sys.exit(1)
        done["msm"] += 1
        done["streamed"] += 1 if segs else 0
    else:
        n = rnd.choice([1, 3, 64, 65, 500, 1500])
        p1 = bytearray(cref.gen_points(cid, 1, rnd.getrandbits(40), rnd.getrandbits(40), n))
        p2 = bytearray(cref.gen_points(cid, 2, rnd.getrandbits(40), rnd.getrandbits(40), n))
        for k in range(n // 40 + (1 if rnd.random() < 0.3 else 0)):  # points at infinity on either side
            j = rnd.randrange(n)
            if k % 2:
                p1[j * g1b : (j + 1) * g1b] = bytes(g1b)
            else:
                p2[j * g2b : (j + 1) * g2b] = bytes(g2b)
        p1, p2 = bytes(p1), bytes(p2)
        exp = cref.pairing_batch(cid, p1, p2, n, 16)
        out = ctypes.create_string_buffer(gtb * n)
        quad = rnd.choice([None, "0", "1"])  # BLS12-381: one pairing per quad of lanes (default at these sizes) or per lane pair
        if quad is None:
            os.environ.pop("MLHIP_PAIRING_QUAD", None)
        else:
            os.environ["MLHIP_PAIRING_QUAD"] = quad
        _lib.check(lib.mlhip_pairing_batch(cid, p1, p2, n, out))
        if out.raw != exp:
            print("MISMATCH pairing", name, n, "seed", seed, flush=True)
            sys.exit(1)
        done["pairing"] += 1
    if time.time() - last > 20:
        last = time.time()
        print("soak %.0fs: %s" % (time.time() - t0, done), flush=True)
print("soak OK after %.0fs: %s (seed %d)" % (time.time() - t0, done, seed), flush=True)
