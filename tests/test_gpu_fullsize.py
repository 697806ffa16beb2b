"""Full-size checks at BASELINE.json's sizes through size-independent properties (plus, where the host has
the cores for it, the C oracle on the whole input)."""
import ctypes

import pytest

pytestmark = pytest.mark.gpu


def _setup(mlhip, n, seed=1):
    import torch

    from mathlib_amd.driver import Curve

    lib = mlhip.load()
    cid = mlhip.CURVE_BLS12_381
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    k = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, generator=gen, device=dev).view(torch.uint8).reshape(n, 32)
    s = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, generator=gen, device=dev).view(torch.uint8).reshape(n, 32)
    base = torch.frombuffer(bytearray(Curve(cid).GenG1().raw), dtype=torch.uint8).to(dev)
    pts = torch.empty(n * 96, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    mlhip.check(lib.mlhip_scalar_mul_device(cid, 1, base.data_ptr(), 0, k.data_ptr(), 0, n, pts.data_ptr(), st))
    torch.cuda.synchronize()
    return lib, cid, pts, s, k, st


def test_msm_2_20_c16_linearity_and_oracle(mlhip):
    """BASELINE configs[1]: BLS12-381 2^20-point G1 MSM, c = 16.
    (1) MSM(P, s) with P_i = [k_i]G equals [sum k_i s_i]G  (checked with one n=1 scalar mul);
    (2) halves add up: MSM(P, s) == MSM(P[:h], s[:h]) + MSM(P[h:], s[h:]);
    (3) the C oracle on the whole input (threads = host cores)."""
    import os

    import numpy as np
    import torch

    from oracle import cref

    n = 1 << 20
    lib, cid, pts, s, k, st = _setup(mlhip, n)
    plan = mlhip.MsmPlan(cid, 1, n, 16)
    full = plan.run(pts.data_ptr(), s.data_ptr(), n, False, st)
    h = n // 2 + 12345
    a = plan.run(pts.data_ptr(), s.data_ptr(), h, False, st)
    b = plan.run(pts.data_ptr() + h * 96, s.data_ptr() + h * 32, n - h, False, st)
    out = ctypes.create_string_buffer(96)
    mlhip.check(lib.mlhip_g1_sum(cid, a + b, 2, out))
    assert out.raw == full
    # (1): scalar sum over the integers mod r, then one scalar multiplication of the generator
    r = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    kk = k.cpu().numpy().view(np.uint64).reshape(n, 4)
    ss = s.cpu().numpy().view(np.uint64).reshape(n, 4)
    to_int = lambda row: int(row[0]) | int(row[1]) << 64 | int(row[2]) << 128 | int(row[3]) << 192  # noqa: E731
    step = 4096  # exact big-int dot product in chunks
    tot = 0
    for lo in range(0, n, step):
        tot += sum((to_int(kk[i]) % r) * (to_int(ss[i]) % r) for i in range(lo, min(n, lo + step)))
    from mathlib_amd.driver import Curve

    cv = Curve(cid)
    assert cv.GenG1().Mul(cv.NewZrFromInt(tot % r)).raw == full
    # (3)
    threads = max(1, min(64, len(os.sched_getaffinity(0))))
    assert cref.msm(cid, 1, pts.cpu().numpy(), ss, n, False, 16, threads) == full


@pytest.mark.parametrize("shape", ["below_2_32_dups", "ones", "two_values"])
def test_msm_2_20_skewed_scalars_vs_oracle(mlhip, shape):
    """BASELINE configs[1], second distribution: all scalars below 2^32 with 1 % duplicated pairs (the carry bucket of
    the third window holds half of the entries), all scalars one (a plain sum of points), and two scalar values only
    -- the coarse bins and buckets far above the mean that the multi-workgroup sort and the sliced long-bucket sums
    exist for.  Checked against the C oracle on the whole input, resident plan and host-buffer (streamed) call."""
    import os

    import numpy as np
    import torch

    from oracle import cref

    n = 1 << 20
    lib, cid, pts, s, k, st = _setup(mlhip, n, seed=5)
    sc = s.clone().view(torch.int64).reshape(n, 4)
    if shape == "below_2_32_dups":
        sc[:, 1:] = 0
        sc[:, 0] &= 0xFFFFFFFF
        p2 = pts.clone().reshape(n, 96)
        p2[::100] = p2[0]
        sc[::100] = sc[0]
        pts = p2.reshape(-1).contiguous()
    elif shape == "ones":
        sc[:] = 0
        sc[:, 0] = 1
    else:
        sc[::2] = sc[0]
        sc[1::2] = sc[1]
    sc8 = sc.view(torch.uint8).reshape(n, 32).contiguous()
    plan = mlhip.MsmPlan(cid, 1, n, 16)
    got = plan.run(pts.data_ptr(), sc8.data_ptr(), n, False, st)
    threads = max(1, min(64, len(os.sched_getaffinity(0))))
    hp = pts.cpu().numpy()
    hs = sc8.cpu().numpy().view(np.uint64).reshape(n, 4)
    exp = cref.msm(cid, 1, hp, hs, n, False, 16, threads)
    assert got == exp
    out = ctypes.create_string_buffer(96)
    mlhip.check(lib.mlhip_msm_g1(cid, hp.tobytes(), hs.tobytes(), 0, n, 16, out))  # streamed in four segments
    assert out.raw == exp


def test_pairing_batch_65536_properties(mlhip):
    """BASELINE configs[2]: 65 536 pairings e([k_i]G1, [s_i]G2) + FExp.
    (1) EVERY output is covered by one identity: prod_i out_i^(w_i) == GenGt^(sum_i w_i k_i s_i mod r) with random 64-bit
        weights w_i (a wrong, missing or permuted element changes the left side) -- the left side through
        mlhip_gt_exp_device and a tree of mlhip_gt_mul_device, the right side by the oracle (pyref: one pairing, one
        exponentiation); the unweighted product is checked the same way;
    (2) a strided sample of 33 outputs byte for byte against the C oracle;
    (3) the whole batch against a second run split in two launches (determinism / indexing)."""
    import numpy as np
    import torch

    from oracle import cref
    from oracle import pyref as R

    n = 1 << 16
    lib, cid, pts, s, k, st = _setup(mlhip, n, seed=3)
    from mathlib_amd.driver import Curve

    g2 = torch.frombuffer(bytearray(Curve(cid).GenG2().raw), dtype=torch.uint8).cuda()
    q = torch.empty(n * 192, dtype=torch.uint8, device="cuda")
    mlhip.check(lib.mlhip_scalar_mul_device(cid, 2, g2.data_ptr(), 0, s.data_ptr(), 0, n, q.data_ptr(), st))
    out = torch.empty(n * 576, dtype=torch.uint8, device="cuda")
    mlhip.check(lib.mlhip_pairing_batch_device(cid, pts.data_ptr(), q.data_ptr(), n, out.data_ptr(), st))
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    hp, hq = pts.cpu().numpy(), q.cpu().numpy()

    # (1) all elements
    cp = R.BLS12_381
    T = R.tower(cp)
    gen_gt = R.pairing(cp, cp.g1, R.g2_generator(cp))
    to_int = lambda row: int(row[0]) | int(row[1]) << 64 | int(row[2]) << 128 | int(row[3]) << 192  # noqa: E731
    kk = k.cpu().numpy().view(np.uint64).reshape(n, 4)
    ss = s.cpu().numpy().view(np.uint64).reshape(n, 4)
    ks = [(to_int(kk[i]) % cp.r) * (to_int(ss[i]) % cp.r) % cp.r for i in range(n)]
    w = np.zeros((n, 4), dtype=np.uint64)
    w[:, 0] = np.random.default_rng(65536).integers(1, 1 << 63, size=n, dtype=np.uint64)

    def tree_product(buf):
        m = n
        while m > 1:
            half = m // 2
            mlhip.check(lib.mlhip_gt_mul_device(cid, buf.data_ptr(), buf.data_ptr() + (m - half) * 576, half, buf.data_ptr(), st))
            m -= half
        torch.cuda.synchronize()
        return bytes(buf[:576].cpu().numpy().tobytes())

    assert tree_product(out.clone()) == R.gt_to_mont_bytes(cp, T.f12_pow(gen_gt, sum(ks) % cp.r))
    dw = torch.from_numpy(w.view(np.uint8).reshape(-1).copy()).cuda()
    powered = torch.empty_like(out)
    mlhip.check(lib.mlhip_gt_exp_device(cid, out.data_ptr(), dw.data_ptr(), 0, n, powered.data_ptr(), st))
    weighted = sum(int(w[i, 0]) * ks[i] for i in range(n)) % cp.r
    assert tree_product(powered) == R.gt_to_mont_bytes(cp, T.f12_pow(gen_gt, weighted))

    # (2) sample against the C oracle
    idx = list(range(0, n, n // 32)) + [n - 1]
    g1s = b"".join(hp[i * 96 : (i + 1) * 96].tobytes() for i in idx)
    g2s = b"".join(hq[i * 192 : (i + 1) * 192].tobytes() for i in idx)
    ref = cref.pairing_batch(cid, g1s, g2s, len(idx), 8)
    for j, i in enumerate(idx):
        assert o[i * 576 : (i + 1) * 576].tobytes() == ref[j * 576 : (j + 1) * 576], i
    # (3) split run
    out2 = torch.empty_like(out)
    h = n // 2 + 77
    mlhip.check(lib.mlhip_pairing_batch_device(cid, pts.data_ptr(), q.data_ptr(), h, out2.data_ptr(), st))
    mlhip.check(lib.mlhip_pairing_batch_device(cid, pts.data_ptr() + h * 96, q.data_ptr() + h * 192, n - h, out2.data_ptr() + h * 576, st))
    torch.cuda.synchronize()
    assert torch.equal(out, out2)


@pytest.mark.parametrize("curve_name,group,log_n", [("BLS12-381", 2, 18), ("BLS12-377", 1, 19), ("BN254", 1, 20),
                                                   ("BLS12-381", 1, 21), ("BLS12-381", 2, 21), ("BLS12-377", 1, 22)])
def test_msm_other_config_shapes(mlhip, curve_name, group, log_n):
    """BASELINE configs[3] and [4] shapes: the per-GPU shard of config 4 (BLS12-381 2^21 G1 and 2^21 G2), the shard of
    config 5 (BLS12-377 G1 2^19) and its whole 2^22-point form on one GPU, G2 at 2^18 and BN254 2^20: split-sum
    property and the C oracle on the whole input."""
    import os

    import numpy as np
    import torch

    from conftest import load_golden
    from oracle import cref

    g = load_golden(curve_name)
    cid = g["curve_id"]
    lib = mlhip.load()
    fpb, g1b, g2b, gtb = mlhip.sizes(cid)
    sz = g1b if group == 1 else g2b
    n = 1 << log_n
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(100 + cid + group)
    rnd = lambda m: torch.randint(-(1 << 63), (1 << 63) - 1, (m, 4), dtype=torch.int64, generator=gen, device=dev).view(torch.uint8).reshape(m, 32).contiguous()  # noqa: E731
    base = torch.frombuffer(bytearray(bytes.fromhex(g["g1_gen" if group == 1 else "g2_gen"])), dtype=torch.uint8).to(dev)
    pts = torch.empty(n * sz, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    mlhip.check(lib.mlhip_scalar_mul_device(cid, group, base.data_ptr(), 0, rnd(n).data_ptr(), 0, n, pts.data_ptr(), st))
    s = rnd(n)
    torch.cuda.synchronize()
    plan = mlhip.MsmPlan(cid, group, n, 16)
    full = plan.run(pts.data_ptr(), s.data_ptr(), n, False, st)
    h = n // 3 + 777
    a = plan.run(pts.data_ptr(), s.data_ptr(), h, False, st)
    b = plan.run(pts.data_ptr() + h * sz, s.data_ptr() + h * 32, n - h, False, st)
    out = ctypes.create_string_buffer(sz)
    mlhip.check((lib.mlhip_g1_sum if group == 1 else lib.mlhip_g2_sum)(cid, a + b, 2, out))
    assert out.raw == full
    threads = max(1, min(64, len(os.sched_getaffinity(0))))
    ss = s.cpu().numpy().view(np.uint64).reshape(n, 4)
    assert cref.msm(cid, group, pts.cpu().numpy(), ss, n, False, 16, threads) == full
    if curve_name == "BLS12-377" and group == 1:
        # BASELINE configs[4] over a fixed SRS: the same 2^22 pairs with the caller's promise -- bucket sums and reduction
        # in twisted Edwards coordinates, tiles of 2^20 points, then a second launch on the kept conversion, and a prefix
        plan.set_profiling(True)
        plan.assume_srs(True)
        assert plan.run(pts.data_ptr(), s.data_ptr(), n, False, st) == full
        t = plan.timings()
        assert t["edwards"] == 1.0 and t["tiles"] == (n >> 20 if n >= 1 << 21 else 1), t
        assert plan.run(pts.data_ptr(), s.data_ptr(), n, False, st) == full
        assert plan.run(pts.data_ptr(), s.data_ptr(), h, False, st) == a
        plan.assume_srs(False)
        assert plan.run(pts.data_ptr(), s.data_ptr(), n, False, st) == full
        assert plan.timings()["edwards"] == 0.0
    plan.close()


def test_shared_scalars_config4_shard(mlhip, monkeypatch):
    """BASELINE configs[3] as bench.py runs it: the G1 and the G2 MSM of one scalar vector through
    mlhip_msm_launch_shared at the 8-GPU shard size (2^21 pairs), one pass and cut into four tiles -- against the two
    independent plan runs (which test_msm_other_config_shapes checks against the C oracle on the whole input) and, for
    the G1 half, against the C oracle directly."""
    import os

    import numpy as np
    import torch

    from conftest import load_golden
    from oracle import cref

    g = load_golden("BLS12-381")
    cid = g["curve_id"]
    lib = mlhip.load()
    _, g1b, g2b, _ = mlhip.sizes(cid)
    n = 1 << 21
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(2104)
    rnd = lambda m: torch.randint(-(1 << 63), (1 << 63) - 1, (m, 4), dtype=torch.int64, generator=gen, device=dev).view(torch.uint8).reshape(m, 32).contiguous()  # noqa: E731
    st = torch.cuda.current_stream().cuda_stream
    pts = {}
    for group, sz, key in ((1, g1b, "g1_gen"), (2, g2b, "g2_gen")):
        base = torch.frombuffer(bytearray(bytes.fromhex(g[key])), dtype=torch.uint8).to(dev)
        pts[group] = torch.empty(n * sz, dtype=torch.uint8, device=dev)
        mlhip.check(lib.mlhip_scalar_mul_device(cid, group, base.data_ptr(), 0, rnd(n).data_ptr(), 0, n, pts[group].data_ptr(), st))
    s = rnd(n)
    torch.cuda.synchronize()
    a, b = mlhip.MsmPlan(cid, 1, n, 16), mlhip.MsmPlan(cid, 2, n, 16)
    want1 = a.run(pts[1].data_ptr(), s.data_ptr(), n, False, st)
    want2 = b.run(pts[2].data_ptr(), s.data_ptr(), n, False, st)
    for tile in (None, "19"):
        if tile:
            monkeypatch.setenv("MLHIP_TILE_LOG2", tile)
        a.launch_shared(b, pts[1].data_ptr(), pts[2].data_ptr(), s.data_ptr(), n, False, st)
        assert (a.finish(), b.finish()) == (want1, want2), tile
    threads = max(1, min(64, len(os.sched_getaffinity(0))))
    ss = s.cpu().numpy().view(np.uint64).reshape(n, 4)
    assert cref.msm(cid, 1, pts[1].cpu().numpy(), ss, n, False, 16, threads) == want1
    a.close()
    b.close()


def test_config4_whole_n1(mlhip):
    """BASELINE configs[3] exactly as `bench.py --config 4` runs it at N = 1: BLS12-381 2^24 pairs, G1 and G2 MSM of ONE
    scalar vector through mlhip_msm_launch_shared (16 tiles of 2^20 pairs, each sorted once for both groups).
    (1) == mlhip_g1_sum / mlhip_g2_sum of the 8 contiguous 2^21-pair shard results (independent plan runs: the size
        and path test_msm_other_config_shapes pins to the C oracle, and what 8 ranks would each compute);
    (2) linearity, independent of any MSM path: P_i = [k_i]G1, Q_i = [k'_i]G2  =>  results are [sum k_i s_i]G1 and
        [sum k'_i s_i]G2, the integer dot products taken exactly on 16-bit limbs (2^24 products < 2^32 sum below 2^56)."""
    import torch

    from conftest import load_golden
    from mathlib_amd.driver import Curve

    g = load_golden("BLS12-381")
    cid = g["curve_id"]
    lib = mlhip.load()
    _, g1b, g2b, _ = mlhip.sizes(cid)
    log_n, shards = 24, 8
    n = 1 << log_n
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(2404)
    rnd = lambda m: torch.randint(-(1 << 63), (1 << 63) - 1, (m, 4), dtype=torch.int64, generator=gen, device=dev).view(torch.uint8).reshape(m, 32).contiguous()  # noqa: E731
    st = torch.cuda.current_stream().cuda_stream
    s = rnd(n)

    def limbs16(x):  # n x 32 bytes -> n x 16 int64 limbs of 16 bits, little endian
        return x.view(torch.int16).reshape(-1, 16).to(torch.int64) & 0xFFFF

    def dot(k):  # exact integer sum_i k_i * s_i
        kl, sl = limbs16(k), limbs16(s)
        tot = 0
        for a in range(16):
            col = (kl[:, a : a + 1] * sl).sum(0).cpu().tolist()
            tot += sum(int(v) << (16 * (a + b)) for b, v in enumerate(col))
        return tot

    pts, want = {}, {}
    r = int(g["r"], 16) if isinstance(g.get("r"), str) else 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    cv = Curve(cid)
    for group, sz, key in ((1, g1b, "g1_gen"), (2, g2b, "g2_gen")):
        base = torch.frombuffer(bytearray(bytes.fromhex(g[key])), dtype=torch.uint8).to(dev)
        pts[group] = torch.empty(n * sz, dtype=torch.uint8, device=dev)
        k = rnd(n)
        mlhip.check(lib.mlhip_scalar_mul_device(cid, group, base.data_ptr(), 0, k.data_ptr(), 0, n, pts[group].data_ptr(), st))
        torch.cuda.synchronize()
        z = cv.NewZrFromInt(dot(k) % r)
        want[group] = (cv.GenG1() if group == 1 else cv.GenG2()).Mul(z).raw
        del k
    a, b = mlhip.MsmPlan(cid, 1, n, 16), mlhip.MsmPlan(cid, 2, n, 16)
    a.launch_shared(b, pts[1].data_ptr(), pts[2].data_ptr(), s.data_ptr(), n, False, st)
    whole = {1: a.finish(), 2: b.finish()}
    a.close()
    b.close()
    lib.mlhip_release_cache()
    per = n // shards
    for group, sz in ((1, g1b), (2, g2b)):
        plan = mlhip.MsmPlan(cid, group, per, 16)
        parts = b"".join(plan.run(pts[group].data_ptr() + i * per * sz, s.data_ptr() + i * per * 32, per, False, st) for i in range(shards))
        plan.close()
        out = ctypes.create_string_buffer(sz)
        mlhip.check((lib.mlhip_g1_sum if group == 1 else lib.mlhip_g2_sum)(cid, parts, shards, out))
        assert out.raw == whole[group], group
        assert whole[group] == want[group], group


@pytest.mark.parametrize("shape", ["uniform", "below_2_32_dups", "two_values"])
def test_msm_2_20_over_shifted_base_tables_vs_oracle(mlhip, shape):
    """BASELINE configs[1]'s size over a resident-bases handle with the library's geometry (window_c = 0: shifted-base tables,
    13 digits of 20 bits into one bucket set; msm_fold.h): the C oracle on the whole input, for uniform scalars, for the
    skewed distribution (all below 2^32, 1 % duplicated pairs: every high digit is zero and the low ones collide) and for
    two scalar values only (13 x 2 buckets hold everything: the sliced long-bucket sums from the carry-free rows); scalars
    from the host (streamed in two segments) and already on the device; the handle made from device points."""
    import os

    import numpy as np
    import torch

    from oracle import cref

    n = 1 << 20
    lib, cid, pts, s, k, st = _setup(mlhip, n, seed=9)
    sc = s.clone().view(torch.int64).reshape(n, 4)
    if shape == "below_2_32_dups":
        sc[:, 1:] = 0
        sc[:, 0] &= 0xFFFFFFFF
        p2 = pts.clone().reshape(n, 96)
        p2[::100] = p2[0]
        sc[::100] = sc[0]
        pts = p2.reshape(-1).contiguous()
    elif shape == "two_values":
        sc[::2] = sc[0]
        sc[1::2] = sc[1]
    sc8 = sc.view(torch.uint8).reshape(n, 32).contiguous()
    hp = pts.cpu().numpy()
    hs = sc8.cpu().numpy().view(np.uint64).reshape(n, 4)
    threads = max(1, min(64, len(os.sched_getaffinity(0))))
    exp = cref.msm(cid, 1, hp, hs, n, False, 16, threads)
    h = ctypes.c_void_p()
    mlhip.check(lib.mlhip_bases_create_device(cid, 1, pts.data_ptr(), n, 0, ctypes.byref(h)))
    t = mlhip.plan_timings(lib, lib.mlhip_bases_plan(h))
    assert t["tables"] == 1.0 and t["window_c"] == 20 and t["digits_per_scalar"] == 13, t
    out = ctypes.create_string_buffer(96)
    mlhip.check(lib.mlhip_bases_msm_device(h, sc8.data_ptr(), 0, n, st, out))
    assert out.raw == exp, shape
    mlhip.check(lib.mlhip_bases_msm(h, hs.tobytes(), 0, n, out))
    assert out.raw == exp, shape
    half = n // 2 + 777
    mlhip.check(lib.mlhip_bases_msm(h, hs.tobytes(), 0, half, out))
    assert out.raw == cref.msm(cid, 1, hp, hs, half, False, 16, threads), shape
    mlhip.check(lib.mlhip_bases_destroy(h))


def test_msm_2_21_over_shifted_base_tables_two_tiles(mlhip):
    """2^21 + 5 bases: three tiles of the table (2^20, 2^20, 5), the bucket state carried from tile to tile; the halves add up
    and the whole equals the plain plan's result on the same device inputs (itself checked against the oracle at this size
    by the tests above)."""
    n = (1 << 21) + 5
    lib, cid, pts, s, k, st = _setup(mlhip, n, seed=10)
    plan = mlhip.MsmPlan(cid, 1, n, 16)
    want = plan.run(pts.data_ptr(), s.data_ptr(), n, False, st)
    plan.close()
    h = ctypes.c_void_p()
    mlhip.check(lib.mlhip_bases_create_device(cid, 1, pts.data_ptr(), n, 0, ctypes.byref(h)))
    assert mlhip.plan_timings(lib, lib.mlhip_bases_plan(h))["tables"] == 1.0
    out = ctypes.create_string_buffer(96)
    mlhip.check(lib.mlhip_bases_msm_device(h, s.data_ptr(), 0, n, st, out))
    assert out.raw == want
    mlhip.check(lib.mlhip_bases_msm(h, s.cpu().numpy().tobytes(), 0, n, out))
    assert out.raw == want
    mlhip.check(lib.mlhip_bases_destroy(h))
