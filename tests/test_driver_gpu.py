"""The reference's own acceptance tests for a driver (TestCurves, math_test.go:852-877), restated against
the HIP backend through the host mirror of the driver interface (mathlib_amd/driver.py).  Helper names
follow math_test.go; inputs are seeded (the reference uses crypto/rand)."""
import random

import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

CURVES = [(0, "BN254"), (1, "BLS12-381"), (2, "BLS12-377")]


@pytest.fixture(scope="module", params=CURVES, ids=[c[1] for c in CURVES])
def curve(request, mlhip):
    from mathlib_amd.driver import Curve

    assert mlhip.device_count() >= 1
    cid, name = request.param
    cv = Curve(cid)
    g = load_golden(name)
    co = g["g2_gen_coords"]
    cv._golden = g
    cv._gen_g2 = cv.NewG2FromCoords((int(co[0][0]), int(co[0][1])), (int(co[1][0]), int(co[1][1])))
    rnd = random.Random(20251003 + cid)
    cv._rng = lambda n: rnd.randrange(n)
    return cv


def test_runMultiScalarMul(curve):
    """math_test.go:323-346: MultiScalarMul == sum_i g1s[i].Mul(zrs[i]), n = 10"""
    c = curve
    g1s, zrs = [], []
    for _ in range(10):
        g1s.append(c.GenG1().Mul(c.NewRandomZr(c._rng)))
        zrs.append(c.NewRandomZr(c._rng))
    msm = c.MultiScalarMul(g1s, zrs)
    acc = c.NewG1()
    for p, z in zip(g1s, zrs):
        acc.Add(p.Mul(z))
    assert msm.Equals(acc)
    assert msm.Compressed() == acc.Compressed() and msm.Bytes() == acc.Bytes()  # Test381Compat-style byte check
    # length mismatch: gnark's error is dropped by the driver -> identity (bls12-381.go:777)
    assert c.MultiScalarMul(g1s[:3], zrs[:5]).IsInfinity()
    assert c.MultiScalarMul([], []).IsInfinity()


def test_runG1Test_mul_add_sub(curve):
    """math_test.go:272-321 (group-law sanity through the MSM path): [a]G + [b]G == [a+b]G, Mul2, Sub"""
    c = curve
    a, b = c.NewRandomZr(c._rng), c.NewRandomZr(c._rng)
    g = c.GenG1()
    s = g.Mul(a)
    s.Add(g.Mul(b))
    assert s.Equals(g.Mul(a.Plus(b)))
    assert g.Mul2(a, g.Mul(b), b).Equals(g.Mul(a.Plus(b.Mul(b))))
    d = g.Mul(a)
    d.Sub(g.Mul(a))
    assert d.IsInfinity()
    assert g.Mul(c.GroupOrder).IsInfinity()
    assert g.Mul(c.NewZrFromInt(-1)).Equals(g.Mul(c.NewZrFromInt(c.r - 1)))  # negative BaseZr scalars


def test_runPairingTest(curve):
    """math_test.go:423-455: bilinearity and Pairing2 == Pairing * Pairing, compared after FExp"""
    c = curve
    r1, r2 = c.NewRandomZr(c._rng), c.NewRandomZr(c._rng)
    g1, g2 = c.GenG1(), c._gen_g2
    a = c.FExp(c.Pairing(g2.Mul(r1), g1.Mul(r2)))
    b = c.FExp(c.Pairing(g2.Mul(r1.Mul(r2)), g1))
    assert a.Equals(b) and a.Bytes() == b.Bytes()
    p = c.Pairing(g2.Mul(r1), g1.Mul(r2))
    q = c.Pairing(g2.Mul(r2), g1.Mul(r1))
    pq = c.FExp(p)
    pq.Mul(c.FExp(q))
    p2 = c.FExp(c.Pairing2(g2.Mul(r1), g2.Mul(r2), g1.Mul(r2), g1.Mul(r1)))
    assert p2.Equals(pq)


def test_runGtTest_generator(curve):
    """math_test.go:457-470: FExp(Pairing(GenG2, GenG1)) == GenGt; wire bytes as the oracle's GT.Bytes()"""
    c = curve
    gengt = c.FExp(c.Pairing(c._gen_g2, c.GenG1()))
    assert gengt.raw == bytes.fromhex(c._golden["pairing"][0]["fexp"])
    assert gengt.Bytes() == bytes.fromhex(c._golden["gen_gt_wire"])
    assert not gengt.IsUnity()
    # a pair with the identity gives unity after FExp
    assert c.FExp(c.Pairing(c._gen_g2, c.NewG1())).IsUnity()


def test_PairingBatch_equals_elementwise(curve):
    """additive API: PairingBatch(g2s, g1s)[i] == FExp(Pairing(g2s[i], g1s[i]))"""
    c = curve
    g1s = [c.GenG1().Mul(c.NewRandomZr(c._rng)) for _ in range(5)] + [c.NewG1()]
    g2s = [c._gen_g2.Mul(c.NewRandomZr(c._rng)) for _ in range(5)] + [c._gen_g2]
    out = c.PairingBatch(g2s, g1s)
    for i in range(6):
        assert out[i].Equals(c.FExp(c.Pairing(g2s[i], g1s[i])))
    assert out[5].IsUnity()


def test_MultiScalarMulG1G2(curve):
    """the additive shared-scalar call equals the two reference-shaped calls, including the length rules"""
    c = curve
    n = 37
    g1s = [c.GenG1().Mul(c.NewRandomZr(c._rng)) for _ in range(n - 1)] + [c.NewG1()]
    g2s = [c._gen_g2.Mul(c.NewRandomZr(c._rng)) for _ in range(n)]
    zrs = [c.NewRandomZr(c._rng) for _ in range(n)]
    r1, r2 = c.MultiScalarMulG1G2(g1s, g2s, zrs)
    assert r1.Equals(c.MultiScalarMul(g1s, zrs)) and r2.Equals(c.MultiScalarMulG2(g2s, zrs))
    e1, e2 = c.MultiScalarMulG1G2(g1s[:3], g2s[:3], zrs[:5])  # more scalars than points: identity, like MultiScalarMul
    assert e1.IsInfinity() and e2.Equals(c.NewG2())
    with pytest.raises(IndexError):
        c.MultiScalarMulG1G2(g1s, g2s, zrs[:2])


def test_MultiScalarMulG2(curve):
    c = curve
    g2s = [c._gen_g2.Mul(c.NewRandomZr(c._rng)) for _ in range(4)]
    zrs = [c.NewRandomZr(c._rng) for _ in range(4)]
    acc = c.NewG2()
    for p, z in zip(g2s, zrs):
        acc.Add(p.Mul(z))
    assert c.MultiScalarMulG2(g2s, zrs).Equals(acc)


@pytest.mark.parametrize("one_lane", ["0", "1"])
def test_scalar_mul_batch_kernel(curve, mlhip, one_lane, monkeypatch):
    """batched G1.Mul / G2.Mul (mlhip_scalar_mul) against the n = 1 MSM path and the oracle; G2 on the lane-pair kernel
    (default) and on the one-lane kernel"""
    import ctypes

    monkeypatch.setenv("MLHIP_SCALAR_MUL_ONE_LANE", one_lane)

    from oracle import cref

    c = curve
    lib = mlhip.load()
    n = 200
    ks = [c._rng(c.r) for _ in range(n - 3)] + [0, 1, c.r - 1]
    sc = b"".join(k.to_bytes(32, "little") for k in ks)
    out = ctypes.create_string_buffer(c.g1_bytes * n)
    mlhip.check(lib.mlhip_scalar_mul(c.id, 1, c.GenG1().raw, 0, sc, 0, n, out))
    for i in (0, 1, 57, n - 3, n - 2, n - 1):
        assert out.raw[i * c.g1_bytes : (i + 1) * c.g1_bytes] == cref.point_mul(c.id, 1, c.GenG1().raw, ks[i])
    out2 = ctypes.create_string_buffer(c.g2_bytes * 8)
    mlhip.check(lib.mlhip_scalar_mul(c.id, 2, c._gen_g2.raw, 0, sc[: 8 * 32], 0, 8, out2))
    for i in range(8):
        assert out2.raw[i * c.g2_bytes : (i + 1) * c.g2_bytes] == cref.point_mul(c.id, 2, c._gen_g2.raw, ks[i])
    # per-point bases (stride 1)
    out3 = ctypes.create_string_buffer(c.g1_bytes * 16)
    mlhip.check(lib.mlhip_scalar_mul(c.id, 1, out.raw[: 16 * c.g1_bytes], 1, sc[32 : 17 * 32], 0, 16, out3))
    for i in range(16):
        assert out3.raw[i * c.g1_bytes : (i + 1) * c.g1_bytes] == cref.point_mul(c.id, 1, out.raw[i * c.g1_bytes : (i + 1) * c.g1_bytes], ks[i + 1])


def test_fixed_base_table_path(curve, mlhip, monkeypatch):
    """One base, many scalars: the table path (normally from 2^12 scalars on) forced at n = 300, against the oracle and
    the double-and-add kernel -- scalars with zero bytes, 0, 1, r - 1, a base outside the r-torsion subgroup of G1
    (the table must not reduce its multiples mod r) and the point at infinity as base."""
    import ctypes

    from oracle import cref
    from oracle import pyref as R

    c = curve
    lib = mlhip.load()
    n = 300
    ks = [c._rng(c.r) for _ in range(n - 6)] + [0, 1, c.r - 1, 0xFF00FF << 40, 1 << 248, (1 << 200) + 5]
    ks = [k % c.r for k in ks]
    sc = b"".join(k.to_bytes(32, "little") for k in ks)
    for group, base, size in ((1, c.GenG1().raw, c.g1_bytes), (2, c._gen_g2.raw, c.g2_bytes)):
        outs = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("MLHIP_FIXED_BASE_MIN", mode)
            out = ctypes.create_string_buffer(size * n)
            mlhip.check(lib.mlhip_scalar_mul(c.id, group, base, 0, sc, 0, n, out))
            outs[mode] = out.raw
        assert outs["1"] == outs["0"]
        for i in (0, 3, n - 6, n - 5, n - 4, n - 3, n - 2, n - 1):
            assert outs["1"][i * size : (i + 1) * size] == cref.point_mul(c.id, group, base, ks[i])
        if group == 1:
            # BLS12-377: a base in the prime-order subgroup (the table build computes [r]P as one more entry) takes the
            # 7-product twisted Edwards additions (k_fixed_base_ed, round 4); MLHIP_EDWARDS=0 keeps the XYZZ products --
            # same bytes, and the cached table serves both (no-op on the other curves)
            monkeypatch.setenv("MLHIP_FIXED_BASE_MIN", "1")
            for ed in ("0", "1", "0"):
                monkeypatch.setenv("MLHIP_EDWARDS", ed)
                out = ctypes.create_string_buffer(size * n)
                mlhip.check(lib.mlhip_scalar_mul(c.id, group, base, 0, sc, 0, n, out))
                assert out.raw == outs["1"], ("MLHIP_EDWARDS", ed)
            monkeypatch.delenv("MLHIP_EDWARDS")
        from conftest import alt_build

        if group == 2 and alt_build():  # the table path on the boundary-form lane-pair kernel (test build; the default is carry-free)
            monkeypatch.setenv("MLHIP_FIXED_BASE_MIN", "1")
            monkeypatch.setenv("MLHIP_ACC32", "1")
            out = ctypes.create_string_buffer(size * n)
            mlhip.check(lib.mlhip_scalar_mul(c.id, group, base, 0, sc, 0, n, out))
            monkeypatch.delenv("MLHIP_ACC32")
            assert out.raw == outs["1"]
    # a curve point outside G1 (cofactor > 1 on the BLS curves): x = 1, 2, ... until x^3 + b is a square
    cp = next(v for v in R.CURVES.values() if v.curve_id == c.id)
    if cp.family == "BLS12":
        x = 1
        while True:
            y = R.fp_sqrt((x * x * x + cp.b) % cp.p, cp.p)
            if y is not None and R.g1_mul_unreduced(cp, (x, y), cp.r) is not None:
                break
            x += 1
        base = R.g1_to_mont_bytes(cp, (x, y))
        monkeypatch.setenv("MLHIP_FIXED_BASE_MIN", "1")
        out = ctypes.create_string_buffer(c.g1_bytes * n)
        mlhip.check(lib.mlhip_scalar_mul(c.id, 1, base, 0, sc, 0, n, out))
        for i in (0, n - 4, n - 2, n - 1):
            assert out.raw[i * c.g1_bytes : (i + 1) * c.g1_bytes] == R.g1_to_mont_bytes(cp, R.g1_mul_unreduced(cp, (x, y), ks[i]))
    # infinity as base
    monkeypatch.setenv("MLHIP_FIXED_BASE_MIN", "1")
    out = ctypes.create_string_buffer(c.g1_bytes * n)
    mlhip.check(lib.mlhip_scalar_mul(c.id, 1, bytes(c.g1_bytes), 0, sc, 0, n, out))
    assert out.raw == bytes(c.g1_bytes * n)
    # the table stays on the device with its base as the key: the same base again (no build), another base (rebuilt), the
    # first one again, the other group in between; every window width; MLHIP_FB_CACHE=0 = a build per call
    ks2 = ks + [(1 << 255) % c.r, c.r // 2, c.r // 2 + 1, (1 << 252) - 1] + [((1 << w) - 1) << (5 * w) for w in (8, 11, 12, 13)]
    sc2 = b"".join(k.to_bytes(32, "little") for k in ks2)
    n2 = len(ks2)
    other = cref.point_mul(c.id, 1, c.GenG1().raw, 12345)
    want = {}
    monkeypatch.setenv("MLHIP_FIXED_BASE_MIN", "0")
    for name, group, base, size in (("g", 1, c.GenG1().raw, c.g1_bytes), ("h", 1, other, c.g1_bytes), ("g2", 2, c._gen_g2.raw, c.g2_bytes)):
        out = ctypes.create_string_buffer(size * n2)
        mlhip.check(lib.mlhip_scalar_mul(c.id, group, base, 0, sc2, 0, n2, out))
        want[name] = (group, base, size, out.raw)
    for i in (n - 1, n2 - 8, n2 - 7, n2 - 6, n2 - 1):
        assert want["g"][3][i * c.g1_bytes : (i + 1) * c.g1_bytes] == cref.point_mul(c.id, 1, c.GenG1().raw, ks2[i])
    monkeypatch.setenv("MLHIP_FIXED_BASE_MIN", "1")
    for width, cache in ((None, None), ("4", None), ("8", None), ("13", None), ("14", None), (None, "0")):
        if width:
            monkeypatch.setenv("MLHIP_FB_WINDOW", width)
        if cache:
            monkeypatch.setenv("MLHIP_FB_CACHE", cache)
        for name in ("g", "g", "h", "g", "g2", "g2", "g", "h", "h"):
            group, base, size, expect = want[name]
            out = ctypes.create_string_buffer(size * n2)
            mlhip.check(lib.mlhip_scalar_mul(c.id, group, base, 0, sc2, 0, n2, out))
            assert out.raw == expect, (name, width, cache)
        monkeypatch.delenv("MLHIP_FB_WINDOW", raising=False)
        monkeypatch.delenv("MLHIP_FB_CACHE", raising=False)


def test_batched_mul_and_exp_mirrors(curve):
    """The additive batch entry points of the host mirror (MulBatch, BaseMulBatch, ExpBatch: SURVEY 8f rows 2 and 3) agree
    with the single calls of the reference interface they batch, element by element -- G1 and G2, infinity and r among them."""
    c = curve
    n = 9
    zs = [c.NewRandomZr(c._rng) for _ in range(n - 2)] + [c.NewZrFromInt(0), c.NewZrFromInt(1)]
    p1 = [c.GenG1().Mul(c.NewRandomZr(c._rng)) for _ in range(n - 1)] + [c.NewG1()]
    p2 = [c._gen_g2.Mul(c.NewRandomZr(c._rng)) for _ in range(n - 1)] + [c.NewG2()]  # (the fixture's generator: the mirror has none for BLS12-377)
    for pts in (p1, p2):
        got = c.MulBatch(pts, zs)
        assert len(got) == n and all(got[i].Equals(pts[i].Mul(zs[i])) for i in range(n))
        fixed = c.BaseMulBatch(pts[2], zs)
        assert len(fixed) == n and all(fixed[i].Equals(pts[2].Mul(zs[i])) for i in range(n))
        assert [x.raw for x in c.BaseMulBatch(pts[2], zs)] == [x.raw for x in fixed]
    gt = c.FExp(c.Pairing(c._gen_g2, c.GenG1()))
    gts = [gt.Exp(z) for z in zs[:4]]
    ex = c.ExpBatch(gts, zs[4:8])
    assert all(ex[i].Equals(gts[i].Exp(zs[4 + i])) for i in range(4))
    assert c.MulBatch([], []) == [] and c.BaseMulBatch(c.GenG1(), []) == [] and c.ExpBatch([], []) == []
    with pytest.raises(ValueError):
        c.MulBatch(p1, zs[:3])


def test_runPowTest_gt_exp(curve):
    """math_test.go:390-421: e(g2, g1)^r == e(g2^r, g1) == e(g2, g1^r), through Gt.Exp"""
    c = curve
    r = c.NewRandomZr(c._rng)
    g1, g2 = c.GenG1(), c._gen_g2
    a = c.FExp(c.Pairing(g2, g1)).Exp(r)
    b = c.FExp(c.Pairing(g2.Mul(r), g1))
    d = c.FExp(c.Pairing(g2, g1.Mul(r)))
    assert a.Equals(b) and a.Equals(d)
    # Exp on a raw (not final-exponentiated) Miller value, then FExp: still the same element
    raw = c.Pairing(g2, g1).Exp(r)
    assert c.FExp(raw).Equals(a)
    assert c.FExp(c.Pairing(g2, g1)).Exp(c.GroupOrder).IsUnity()
    assert c.FExp(c.Pairing(g2, g1)).Exp(c.NewZrFromInt(0)).IsUnity()


@pytest.mark.parametrize("one_lane", ["0", "1", "pairs"])
def test_gt_exp_batch_vs_oracle(curve, mlhip, one_lane, monkeypatch):
    """Gt.Exp batch on the default kernel (BLS12-381: one exponentiation per quad of lanes at this size; the other curves:
    lane pairs), on the one-lane kernel and with the quads switched off (BLS12-381: lane pairs), against the Python
    tower."""
    import ctypes

    if one_lane == "pairs":
        monkeypatch.setenv("MLHIP_PAIRING_QUAD", "0")
    else:
        monkeypatch.setenv("MLHIP_PAIRING_ONE_LANE", one_lane)

    from oracle import pyref as R

    c = curve
    cp = R.CURVES_BY_ID[c.id]
    T = R.tower(cp)
    g = c._golden
    f = R.gt_from_mont_bytes(cp, bytes.fromhex(g["pairing"][1]["fexp"]))
    ks = [0, 1, 2, cp.r - 1, c._rng(cp.r), c._rng(1 << 64)]
    out = ctypes.create_string_buffer(c.gt_bytes * len(ks))
    mlhip.check(mlhip.load().mlhip_gt_exp(c.id, bytes.fromhex(g["pairing"][1]["fexp"]) * len(ks), b"".join(k.to_bytes(32, "little") for k in ks), 0, len(ks), out))
    for i, k in enumerate(ks):
        assert out.raw[i * c.gt_bytes : (i + 1) * c.gt_bytes] == R.gt_to_mont_bytes(cp, T.f12_pow(f, k)), k


def test_PairingProduct_shared_final_exp(curve):
    """prod_i e(P_i, Q_i) with one FExp == product of the individually exponentiated pairings (17 pairs:
    exercises the odd-sized tree), the empty product is one, and e(P,Q) e(-P,Q) == 1."""
    c = curve
    g1s = [c.GenG1().Mul(c.NewRandomZr(c._rng)) for _ in range(17)]
    g2s = [c._gen_g2.Mul(c.NewRandomZr(c._rng)) for _ in range(17)]
    prod = c.PairingProduct(g2s, g1s)
    acc = None
    for gt in c.PairingBatch(g2s, g1s):
        if acc is None:
            acc = gt
        else:
            acc.Mul(gt)
    assert prod.Equals(acc)
    assert c.PairingProduct([], []).IsUnity()
    n = g1s[0].Copy()
    n.Neg()
    assert c.PairingProduct([g2s[0], g2s[0]], [g1s[0], n]).IsUnity()


def test_runToFroBytesTest_and_compressed(curve):
    """math_test.go:511-589: Bytes()/Compressed() -> NewG1FromBytes/NewG1FromCompressed round trips; bad bytes raise"""
    c = curve
    p = c.GenG1().Mul(c.NewRandomZr(c._rng))
    assert c.NewG1FromBytes(p.Bytes()).Equals(p)
    assert c.NewG1FromCompressed(p.Compressed()).Equals(p)
    assert c.NewG1FromCompressed(c.NewG1().Compressed()).IsInfinity()
    bad = bytearray(p.Bytes())
    bad[-1] ^= 1
    with pytest.raises(ValueError):
        c.NewG1FromBytes(bytes(bad))
    with pytest.raises(ValueError):
        c.NewG1FromCompressed(p.Compressed()[:-1])


def test_g2_wire_round_trips(curve):
    c = curve
    q = c._gen_g2.Mul(c.NewRandomZr(c._rng))
    assert c.NewG2FromBytes(q.Bytes()).Equals(q)
    assert c.NewG2FromCompressed(q.Compressed()).Equals(q)
    assert c.NewG2FromCompressed(c.NewG2().Compressed()).IsInfinity()
    bad = bytearray(q.Bytes())
    bad[-1] ^= 1
    with pytest.raises(ValueError):
        c.NewG2FromBytes(bytes(bad))


@pytest.mark.parametrize("tables", ["0", "1"])
@pytest.mark.parametrize("segments", ["0", "3"])
def test_resident_bases_match_multiscalarmul(curve, segments, tables, monkeypatch):
    """the upload-once point table gives the same element as MultiScalarMul, for the full table and
    for a prefix of it, on repeated calls -- in one pass and with the scalars streamed in three segments; over the plain
    table and over shifted-base tables (forced: a table of 40 bases does not get them by itself)."""
    monkeypatch.setenv("MLHIP_STREAM_SEGMENTS", segments)
    monkeypatch.setenv("MLHIP_BASES_TABLES", tables)
    c = curve
    g = c.GenG1()
    pts = [g.Mul(c.NewRandomZr(c._rng)) for _ in range(40)]
    bases = c.NewBases(pts)
    from mathlib_amd import _lib

    assert bases.ShiftedTables() == (tables == "1")
    # the table is checked on the device where a curve has the twisted Edwards model that needs it (BLS12-377)
    assert bases.CheckedSubgroup() == (c.id == _lib.CURVE_BLS12_377)
    for n in (40, 17, 1, 40):
        sc = [c.NewRandomZr(c._rng) for _ in range(n)]
        assert bases.MultiScalarMul(sc).Equals(c.MultiScalarMul(pts[:n], sc))
    assert bases.MultiScalarMul([]).IsInfinity()
    with pytest.raises(IndexError):
        bases.MultiScalarMul([c.NewRandomZr(c._rng)] * 41)
    bases.Close()
