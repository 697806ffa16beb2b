"""GPU parity tests: the HIP path, called through the C ABI (include/mlhip.h), against the committed
golden vectors (tests/golden, produced by oracle/pyref.py) -- bit-exact, this is integer work.
Mirrors the reference's own checks: MSM == sum of scalar multiples (math_test.go:323-346),
FExp(Pairing(..)) equalities (math_test.go:423-470), serialized-byte comparison (math_test.go:879-911).
"""
import ctypes
import os

import pytest

from conftest import load_golden, load_msm1000

pytestmark = pytest.mark.gpu

CURVES = ["BN254", "BLS12-381", "BLS12-377"]


def _h(s):
    return bytes.fromhex(s)


@pytest.fixture(scope="module")
def lib(mlhip):
    l = mlhip.load()
    assert mlhip.device_count() >= 1, "no GPU visible: the product path has no CPU fallback"
    return l


@pytest.mark.parametrize("curve", CURVES)
def test_fp_mul_kernel_known_answers(lib, mlhip, curve):
    import torch

    g = load_golden(curve)
    cid = g["curve_id"]
    a = b"".join(_h(c["a"]) for c in g["fp_mul"])
    b = b"".join(_h(c["b"]) for c in g["fp_mul"])
    exp = b"".join(_h(c["ab"]) for c in g["fp_mul"])
    n = len(g["fp_mul"])
    da = torch.frombuffer(bytearray(a), dtype=torch.uint8).cuda()
    db = torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()
    out = torch.zeros(len(exp), dtype=torch.uint8, device="cuda")
    mlhip.check(lib.mlhip_fp_mul_device(cid, da.data_ptr(), db.data_ptr(), n, 1, out.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert bytes(out.cpu().numpy().tobytes()) == exp


def test_fp_mul_reference_held_products(lib, mlhip):
    """The field-multiplication kernel on the only products the REFERENCE itself fixes: the Montgomery-form SWU
    constants of driver/kilic/custom.go:38-42 -- mont_mul(z, zInv) = p - r1 (zInv holds -1/z, z = 11, r1 at
    custom.go:29) and mont_mul(minusBOverA, a) = p - b -- plus 1 * 1 = 1 on r1 and x * 1 = x on custom.go:329-336."""
    import torch

    from test_oracle_pinned import reference_held_products

    table = reference_held_products()
    da = torch.frombuffer(bytearray(b"".join(t[0] for t in table)), dtype=torch.uint8).cuda()
    db = torch.frombuffer(bytearray(b"".join(t[1] for t in table)), dtype=torch.uint8).cuda()
    out = torch.zeros(48 * len(table), dtype=torch.uint8, device="cuda")
    mlhip.check(lib.mlhip_fp_mul_device(mlhip.CURVE_BLS12_381, da.data_ptr(), db.data_ptr(), len(table), 1, out.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert bytes(out.cpu().numpy().tobytes()) == b"".join(t[2] for t in table)


@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("window_c", [0, 4, 7, 12, 16])
def test_msm_g1_golden_cases(lib, mlhip, curve, window_c):
    g = load_golden(curve)
    cid = g["curve_id"]
    _, g1b, _, _ = mlhip.sizes(cid)
    for case in g["msm_g1"]:
        pts = b"".join(_h(p) for p in case["points"])
        scs = b"".join(_h(s) for s in case["scalars"])
        out = ctypes.create_string_buffer(g1b)
        mlhip.check(lib.mlhip_msm_g1(cid, pts, scs, 0, len(case["points"]), window_c, out))
        assert out.raw == _h(case["expected"]), (curve, case["name"], window_c)


@pytest.mark.parametrize("curve", CURVES)
def test_msm_g1_montgomery_scalars(lib, mlhip, curve):
    """fr.Element (Montgomery) scalars, the form the gurvy BLS12-381 driver passes (bls12-381.go:772)."""
    g = load_golden(curve)
    cid = g["curve_id"]
    r = int(g["r"], 16)
    _, g1b, _, _ = mlhip.sizes(cid)
    case = next(c for c in g["msm_g1"] if c["name"] == "n10_random")
    pts = b"".join(_h(p) for p in case["points"])
    scs = b"".join((int(s) % r * (1 << 256) % r).to_bytes(32, "little") for s in case["scalars_int"])
    out = ctypes.create_string_buffer(g1b)
    mlhip.check(lib.mlhip_msm_g1(cid, pts, scs, 1, 10, 0, out))
    assert out.raw == _h(case["expected"])


@pytest.mark.parametrize("curve", CURVES)
def test_msm_g1_empty(lib, mlhip, curve):
    g = load_golden(curve)
    _, g1b, _, _ = mlhip.sizes(g["curve_id"])
    out = ctypes.create_string_buffer(b"\xff" * g1b, g1b)
    mlhip.check(lib.mlhip_msm_g1(g["curve_id"], None, None, 0, 0, 0, out))
    assert out.raw == bytes(g1b)


@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("window_c", [0, 10, 16])
def test_msm_g1_1000_points(lib, mlhip, curve, window_c):
    """BASELINE config 1 shape (1k-point G1 MSM), on every curve."""
    g = load_golden(curve)
    fpb, g1b, _, _ = mlhip.sizes(g["curve_id"])
    pts, scs, exp = load_msm1000(curve, fpb)
    out = ctypes.create_string_buffer(g1b)
    mlhip.check(lib.mlhip_msm_g1(g["curve_id"], pts, scs, 0, 1000, window_c, out))
    assert out.raw == exp


@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("window_c", [0, 8, 16])
def test_msm_g2_golden_cases(lib, mlhip, curve, window_c):
    g = load_golden(curve)
    cid = g["curve_id"]
    _, _, g2b, _ = mlhip.sizes(cid)
    for case in g["msm_g2"]:
        pts = b"".join(_h(p) for p in case["points"])
        scs = b"".join(_h(s) for s in case["scalars"])
        out = ctypes.create_string_buffer(g2b)
        mlhip.check(lib.mlhip_msm_g2(cid, pts, scs, 0, len(case["points"]), window_c, out))
        assert out.raw == _h(case["expected"]), (curve, case["name"], window_c)


@pytest.mark.parametrize("curve", CURVES)
def test_pairing_batch_golden(lib, mlhip, curve):
    g = load_golden(curve)
    cid = g["curve_id"]
    _, _, _, gtb = mlhip.sizes(cid)
    n = len(g["pairing"])
    g1 = b"".join(_h(c["g1"]) for c in g["pairing"])
    g2 = b"".join(_h(c["g2"]) for c in g["pairing"])
    out = ctypes.create_string_buffer(gtb * n)
    mlhip.check(lib.mlhip_pairing_batch(cid, g1, g2, n, out))
    for i, c in enumerate(g["pairing"]):
        assert out.raw[i * gtb : (i + 1) * gtb] == _h(c["fexp"]), (curve, i)


@pytest.mark.parametrize("curve", CURVES)
def test_miller_then_fexp_and_pairing2(lib, mlhip, curve):
    g = load_golden(curve)
    cid = g["curve_id"]
    _, _, _, gtb = mlhip.sizes(cid)
    # Pairing then FExp == golden (raw Miller values are not canonical, only the FExp output is)
    c = g["pairing"][1]
    ml = ctypes.create_string_buffer(gtb)
    fe = ctypes.create_string_buffer(gtb)
    mlhip.check(lib.mlhip_miller_loop(cid, _h(c["g1"]), _h(c["g2"]), 1, 1, ml))
    mlhip.check(lib.mlhip_final_exp(cid, ml.raw, 1, fe))
    assert fe.raw == _h(c["fexp"])
    # Pairing2: shared Miller loop over two pairs
    p2 = g["pairing2"]
    mlhip.check(lib.mlhip_miller_loop(cid, b"".join(_h(x) for x in p2["g1"]), b"".join(_h(x) for x in p2["g2"]), 2, 1, ml))
    mlhip.check(lib.mlhip_final_exp(cid, ml.raw, 1, fe))
    assert fe.raw == _h(p2["fexp"])
    # FExp on the oracle's raw Miller value
    io = g["fexp_io"]
    mlhip.check(lib.mlhip_final_exp(cid, _h(io["input"]), 1, fe))
    assert fe.raw == _h(io["output"])
    # bilinearity vector: e([a]G1, [b]G2) == e(G1,G2)^(ab)
    bl = g["bilinear"]
    mlhip.check(lib.mlhip_pairing_batch(cid, _h(bl["g1"]), _h(bl["g2"]), 1, fe))
    assert fe.raw == _h(bl["fexp"])
    # a pair holding infinity contributes one
    fpb, g1b, g2b, _ = mlhip.sizes(cid)
    mlhip.check(lib.mlhip_miller_loop(cid, bytes(g1b), _h(c["g2"]), 1, 1, ml))
    one = _h(g["fp_mul"][0]["a"])  # placeholder to get length
    assert ml.raw[fpb:] == bytes(gtb - fpb) and any(ml.raw[:fpb])


@pytest.mark.parametrize("curve", CURVES)
def test_gt_mul_matches_pairing2(lib, mlhip, curve):
    """Pairing2 == Pairing * Pairing after FExp (math_test.go:436-446)."""
    g = load_golden(curve)
    cid = g["curve_id"]
    _, _, _, gtb = mlhip.sizes(cid)
    a, b = g["pairing"][1], g["pairing"][2]
    out = ctypes.create_string_buffer(gtb)
    mlhip.check(lib.mlhip_gt_mul(cid, _h(a["fexp"]), _h(b["fexp"]), 1, out))
    assert out.raw == _h(g["pairing2"]["fexp"])


# ---------------------------------------------------------------------------------------------
# mid-size parity against the C restatement (oracle/cref, itself pinned to the golden vectors)
# ---------------------------------------------------------------------------------------------
def _rand_scalars(n, seed, bits=254):
    import numpy as np

    rng = np.random.default_rng(seed)
    sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * 2 + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    top = bits - 192
    sc[:, 3] &= np.uint64((1 << top) - 1)
    return sc


@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("n,window_c", [(3000, 0), (1 << 14, 13), (1 << 14, 16)])
def test_msm_g1_random_vs_cref(lib, mlhip, curve, n, window_c):
    from oracle import cref

    g = load_golden(curve)
    cid = g["curve_id"]
    _, g1b, _, _ = mlhip.sizes(cid)
    pts = cref.gen_points(cid, 1, 0x1234567 + n, 0x89ABCDEF01, n)
    sc = _rand_scalars(n, n + cid, 252)
    exp = cref.msm(cid, 1, pts, sc, n, False, 0, 8)
    out = ctypes.create_string_buffer(g1b)
    mlhip.check(lib.mlhip_msm_g1(cid, pts, sc.tobytes(), 0, n, window_c, out))
    assert out.raw == exp


@pytest.mark.parametrize("curve", CURVES)
def test_msm_g2_random_vs_cref(lib, mlhip, curve):
    from oracle import cref

    g = load_golden(curve)
    cid = g["curve_id"]
    _, _, g2b, _ = mlhip.sizes(cid)
    n = 5000
    pts = cref.gen_points(cid, 2, 77777, 31337, n)
    sc = _rand_scalars(n, 99 + cid, 252)
    exp = cref.msm(cid, 2, pts, sc, n, False, 0, 8)
    out = ctypes.create_string_buffer(g2b)
    mlhip.check(lib.mlhip_msm_g2(cid, pts, sc.tobytes(), 0, n, 12, out))
    assert out.raw == exp


@pytest.mark.parametrize("curve", CURVES)
def test_msm_g2_streamed_segments(lib, mlhip, curve, monkeypatch):
    """The host-buffer G2 MSM streamed in segments (BLS12-381; the other curves have no carry-free G2 kernel and run
    in one pass whatever the switch says): duplicates, negated duplicates, infinities and one hot bucket that is long
    in every segment; same bytes as the C oracle and as the one-pass run."""
    import numpy as np
    from oracle import cref

    g = load_golden(curve)
    cid = g["curve_id"]
    _, _, g2b, _ = mlhip.sizes(cid)
    n = 4000
    pts = bytearray(cref.gen_points(cid, 2, 9091, 4243, n))
    for i in range(0, n, 64):  # duplicated points (doubling inside a bucket when the scalars agree)
        pts[i * g2b : (i + 1) * g2b] = pts[:g2b]
    for i in range(33, n, 500):  # points at infinity
        pts[i * g2b : (i + 1) * g2b] = bytes(g2b)
    pts = bytes(pts)
    cases = {"random": _rand_scalars(n, 404 + cid, 252)}
    sc = _rand_scalars(n, 405 + cid, 252)
    sc[::64] = sc[0]
    cases["equal_scalars_on_equal_points"] = sc
    cases["all_equal"] = np.tile(_rand_scalars(1, 406 + cid, 252), (n, 1))
    for name, sc in cases.items():
        exp = cref.msm(cid, 2, pts, sc, n, False, 0, 8)
        for segs in ("0", "2", "7"):
            monkeypatch.setenv("MLHIP_STREAM_SEGMENTS", segs)
            for c in (8, 16):
                out = ctypes.create_string_buffer(g2b)
                mlhip.check(lib.mlhip_msm_g2(cid, pts, sc.tobytes(), 0, n, c, out))
                assert out.raw == exp, (curve, name, segs, c)


@pytest.mark.parametrize("segments", [0, 3])
@pytest.mark.parametrize("curve", CURVES)
def test_msm_g1_skewed_distributions(lib, mlhip, curve, segments, monkeypatch):
    """BASELINE.md section 3 extra distributions: small scalars, duplicated points, zero scalars, and one
    hot bucket (all scalars equal) which exercises the workgroup-per-bucket path.  segments = 3 streams the same
    inputs in three segments (the hot bucket is then long in every segment and is added into the kept state)."""
    import numpy as np
    from oracle import cref

    monkeypatch.setenv("MLHIP_STREAM_SEGMENTS", str(segments))

    g = load_golden(curve)
    cid = g["curve_id"]
    _, g1b, _, _ = mlhip.sizes(cid)
    n = 6000
    pts = bytearray(cref.gen_points(cid, 1, 424242, 1717, n))
    for i in range(0, n, 100):  # 1 % duplicated points
        pts[i * g1b : (i + 1) * g1b] = pts[:g1b]
    for i in range(50, n, 1000):  # points at infinity
        pts[i * g1b : (i + 1) * g1b] = bytes(g1b)
    pts = bytes(pts)
    cases = {}
    sc = _rand_scalars(n, 5, 252)
    sc[:, 1:] = 0
    sc[:, 0] &= np.uint64(0xFFFFFFFF)
    cases["lt_2_32"] = sc
    sc = _rand_scalars(n, 6, 252)
    sc[::7] = 0
    cases["zeros"] = sc
    sc = np.tile(_rand_scalars(1, 7, 252), (n, 1))
    cases["all_equal"] = sc
    for name, sc in cases.items():
        exp = cref.msm(cid, 1, pts, sc, n, False, 0, 8)
        for c in (8, 16):
            out = ctypes.create_string_buffer(g1b)
            mlhip.check(lib.mlhip_msm_g1(cid, pts, sc.tobytes(), 0, n, c, out))
            assert out.raw == exp, (curve, name, c)


@pytest.mark.parametrize("curve", CURVES)
def test_msm_g1_segment_schedules(lib, mlhip, curve, monkeypatch):
    """msm_plan.h: stream_schedule (round 4) -- a host-buffer G1 MSM streamed in segments of unequal length, for points
    and scalars from the host (mlhip_msm_g1: the sort of a segment starts on its scalars, under the upload of its points)
    and for resident bases (mlhip_bases_msm).  Growing, shrinking and single-weight schedules, sixteen segments, weights
    that round to empty segments, a long bucket in every segment, infinities; the same bytes as the oracle every time."""
    import numpy as np
    from oracle import cref

    g = load_golden(curve)
    cid = g["curve_id"]
    _, g1b, _, _ = mlhip.sizes(cid)
    n = 9000
    pts = bytearray(cref.gen_points(cid, 1, 60606, 707, n))
    for i in range(17, n, 900):
        pts[i * g1b : (i + 1) * g1b] = bytes(g1b)
    pts = bytes(pts)
    cases = {"uniform": _rand_scalars(n, 808 + cid, 252)}
    sk = _rand_scalars(n, 809 + cid, 252)
    sk[::3] = sk[0]  # one long bucket per window, present in every segment
    cases["skewed"] = sk
    handle = ctypes.c_void_p()
    mlhip.check(lib.mlhip_bases_create(cid, 1, pts, n, 12, ctypes.byref(handle)))
    try:
        for name, sc in cases.items():
            exp = cref.msm(cid, 1, pts, sc, n, False, 0, 8)
            exp_head = cref.msm(cid, 1, pts, sc, 5000, False, 0, 8)
            raw = sc.tobytes()
            for sched in ("1,1,2,3,4,5", "3,13", "5,1,1", "7", "1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1", "1,1000,1", "0,2,x,3"):
                monkeypatch.setenv("MLHIP_STREAM_SCHEDULE", sched)
                out = ctypes.create_string_buffer(g1b)
                mlhip.check(lib.mlhip_msm_g1(cid, pts, raw, 0, n, 12, out))
                assert out.raw == exp, (curve, name, sched, "msm_g1")
                mlhip.check(lib.mlhip_bases_msm(handle, raw, 0, n, out))
                assert out.raw == exp, (curve, name, sched, "bases_msm")
                mlhip.check(lib.mlhip_bases_msm(handle, raw, 0, 5000, out))  # fewer scalars than bases
                assert out.raw == exp_head, (curve, name, sched, "bases_msm prefix")
            monkeypatch.delenv("MLHIP_STREAM_SCHEDULE")
    finally:
        mlhip.check(lib.mlhip_bases_destroy(handle))


@pytest.mark.parametrize("curve", CURVES)
def test_msm_long_buckets_in_slices(lib, mlhip, curve, monkeypatch):
    """Buckets far above the slice length (4096 entries): all scalars one (a plain sum of points: one bucket of n
    entries), scalars below 2^16 with the carry bucket of the next window, G2 with equal scalars -- in one pass and
    streamed in segments (each segment's long bucket is folded into the kept state)."""
    import numpy as np
    from oracle import cref

    g = load_golden(curve)
    cid = g["curve_id"]
    _, g1b, g2b, _ = mlhip.sizes(cid)
    n = 40000
    pts = cref.gen_points(cid, 1, 31415, 9265, n)
    ones = np.zeros((n, 4), dtype=np.uint64)
    ones[:, 0] = 1
    small = _rand_scalars(n, 77, 252)
    small[:, 1:] = 0
    small[:, 0] &= np.uint64(0xFFFF)
    for name, sc in (("ones", ones), ("below_2_16", small)):
        exp = cref.msm(cid, 1, pts, sc, n, False, 0, 8)
        for segs in ("0", "3"):
            monkeypatch.setenv("MLHIP_STREAM_SEGMENTS", segs)
            out = ctypes.create_string_buffer(g1b)
            mlhip.check(lib.mlhip_msm_g1(cid, pts, sc.tobytes(), 0, n, 16, out))
            assert out.raw == exp, (curve, name, segs)
    n2 = 10000
    pts2 = cref.gen_points(cid, 2, 2718, 2818, n2)
    sc2 = np.tile(_rand_scalars(1, 78, 252), (n2, 1))
    exp2 = cref.msm(cid, 2, pts2, sc2, n2, False, 0, 8)
    for segs in ("0", "2"):
        monkeypatch.setenv("MLHIP_STREAM_SEGMENTS", segs)
        out = ctypes.create_string_buffer(g2b)
        mlhip.check(lib.mlhip_msm_g2(cid, pts2, sc2.tobytes(), 0, n2, 16, out))
        assert out.raw == exp2, (curve, "g2_all_equal", segs)


@pytest.mark.parametrize("curve", CURVES)
def test_pairing_batch_random_vs_cref(lib, mlhip, curve):
    from oracle import cref

    g = load_golden(curve)
    cid = g["curve_id"]
    fpb, g1b, g2b, gtb = mlhip.sizes(cid)
    n = 200
    g1 = bytearray(cref.gen_points(cid, 1, 5555, 77, n))
    g2 = bytearray(cref.gen_points(cid, 2, 6666, 99, n))
    g1[3 * g1b : 4 * g1b] = bytes(g1b)  # pairs holding infinity -> 1
    g2[9 * g2b : 10 * g2b] = bytes(g2b)
    exp = cref.pairing_batch(cid, bytes(g1), bytes(g2), n, 8)
    out = ctypes.create_string_buffer(gtb * n)
    mlhip.check(lib.mlhip_pairing_batch(cid, bytes(g1), bytes(g2), n, out))
    assert out.raw == exp
    # Miller loop with 3 pairs per product, compared after FExp
    ml = ctypes.create_string_buffer(gtb * 60)
    mlhip.check(lib.mlhip_miller_loop(cid, bytes(g1[: 180 * g1b]), bytes(g2[: 180 * g2b]), 3, 60, ml))
    fe = ctypes.create_string_buffer(gtb * 60)
    mlhip.check(lib.mlhip_final_exp(cid, ml.raw, 60, fe))
    assert fe.raw == cref.final_exp(cid, cref.miller_loop(cid, bytes(g1), bytes(g2), 3, 60, 8), 60, 8)


# ---------------------------------------------------------------------------------------------
# wire format: bulk NewG1FromBytes / NewG1FromCompressed and Bytes / Compressed
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("curve", CURVES)
def test_g1_wire_codec_vs_oracle(lib, mlhip, curve):
    from oracle import pyref as R

    cp = R.CURVES[curve]
    cid = cp.curve_id
    n = cp.fp_bytes
    d = R.Drbg("gpu/codec/" + curve)
    pts = [R.random_g1(cp, d) for _ in range(40)] + [None, cp.g1, R.g1_neg(cp, cp.g1)]
    for comp, enc in ((1, R.g1_wire_compressed), (0, R.g1_wire_uncompressed)):
        wire = b"".join(enc(cp, p) for p in pts)
        out = ctypes.create_string_buffer(2 * n * len(pts))
        st = ctypes.create_string_buffer(len(pts))
        mlhip.check(lib.mlhip_g1_from_bytes(cid, wire, len(pts), comp, 1, out, st))
        assert st.raw == bytes(len(pts))
        assert out.raw == b"".join(R.g1_to_mont_bytes(cp, p) for p in pts)
        back = ctypes.create_string_buffer(len(wire))
        mlhip.check(lib.mlhip_g1_to_bytes(cid, out.raw, len(pts), comp, back))
        assert back.raw == wire  # runToFroBytesTest / runToFroCompressedTest (math_test.go:511-589)
    # invalid encodings: every status must agree with the oracle's SetBytes restatement
    bad = []
    x = 1
    while len(bad) < 4:  # x^3 + b a non-residue
        x += 1
        if R.fp_sqrt((x**3 + cp.b) % cp.p, cp.p) is None:
            w = bytearray(x.to_bytes(n, "big"))
            w[0] |= 0x80
            bad.append(bytes(w))
    w = bytearray(cp.p.to_bytes(n, "big"))  # coordinate >= p
    w[0] |= 0x80
    bad.append(bytes(w))
    w = bytearray(R.g1_wire_compressed(cp, None))  # infinity with stray bits
    w[7] = 3
    bad.append(bytes(w))
    if cp.family == "BLS12":  # on the curve but outside the r-torsion subgroup
        x = 2
        while True:
            y = R.fp_sqrt((x**3 + cp.b) % cp.p, cp.p)
            if y is not None and R.g1_mul_unreduced(cp, (x, y), cp.r) is not None:
                break
            x += 1
        bad.append(R.g1_wire_compressed(cp, (x, y)))
    bad.append(R.g1_wire_compressed(cp, pts[0]))  # a good one in between
    st = ctypes.create_string_buffer(len(bad))
    out = ctypes.create_string_buffer(2 * n * len(bad))
    for mode in (1, 2):  # 1: endomorphism test (BLS12), 2: the plain [r]P ladder -- both exact
        mlhip.check(lib.mlhip_g1_from_bytes(cid, b"".join(bad), len(bad), 1, mode, out, st))
        assert list(st.raw) == [R.g1_from_wire(cp, w)[1] for w in bad], mode
        assert out.raw[-2 * n :] == R.g1_to_mont_bytes(cp, pts[0])
    w = bytearray(R.g1_wire_uncompressed(cp, pts[1]))  # uncompressed, off the curve
    w[-1] ^= 1
    st1 = ctypes.create_string_buffer(1)
    mlhip.check(lib.mlhip_g1_from_bytes(cid, bytes(w), 1, 0, 1, ctypes.create_string_buffer(2 * n), st1))
    assert st1.raw[0] == 2


@pytest.mark.parametrize("curve", CURVES)
def test_g2_wire_codec_vs_oracle(lib, mlhip, curve):
    from oracle import pyref as R

    cp = R.CURVES[curve]
    T = R.tower(cp)
    cid = cp.curve_id
    n = cp.fp_bytes
    d = R.Drbg("gpu/codec2/" + curve)
    g2 = R.g2_generator(cp)
    pts = [R.random_g2(cp, d) for _ in range(12)] + [None, g2, R.g2_neg(cp, g2)]
    for comp, enc in ((1, R.g2_wire_compressed), (0, R.g2_wire_uncompressed)):
        wire = b"".join(enc(cp, p) for p in pts)
        out = ctypes.create_string_buffer(4 * n * len(pts))
        st = ctypes.create_string_buffer(len(pts))
        mlhip.check(lib.mlhip_g2_from_bytes(cid, wire, len(pts), comp, 1, out, st))
        assert st.raw == bytes(len(pts))
        assert out.raw == b"".join(R.g2_to_mont_bytes(cp, p) for p in pts)
        back = ctypes.create_string_buffer(len(wire))
        mlhip.check(lib.mlhip_g2_to_bytes(cid, out.raw, len(pts), comp, back))
        assert back.raw == wire
    bad = []
    k = 1
    while len(bad) < 3:
        k += 1
        x = (k, 0) if len(bad) == 0 else (k, 1)
        if T.f2_sqrt(T.f2_add(T.f2_mul(T.f2_sqr(x), x), R.twist_b(cp))) is None:
            w = bytearray(x[1].to_bytes(n, "big") + x[0].to_bytes(n, "big"))
            w[0] |= 0x80
            bad.append(bytes(w))
    w = bytearray((1).to_bytes(n, "big") + cp.p.to_bytes(n, "big"))
    w[0] |= 0x80
    bad.append(bytes(w))
    w = bytearray(R.g2_wire_compressed(cp, None))
    w[n + 3] = 1
    bad.append(bytes(w))
    Qx = R._g2_some_point(cp, 3)
    bad.append(R.g2_wire_compressed(cp, Qx))
    bad.append(R.g2_wire_compressed(cp, pts[0]))
    st = ctypes.create_string_buffer(len(bad))
    out = ctypes.create_string_buffer(4 * n * len(bad))
    want = [R.g2_from_wire(cp, w)[1] for w in bad]
    assert want[-2] == 3 and want[-1] == 0
    for mode in (1, 2):  # 1: psi(Q) = [x]Q on the BLS12 curves, 2: the plain [r]Q ladder
        mlhip.check(lib.mlhip_g2_from_bytes(cid, b"".join(bad), len(bad), 1, mode, out, st))
        assert list(st.raw) == want, mode
    assert out.raw[-4 * n :] == R.g2_to_mont_bytes(cp, pts[0])
    mlhip.check(lib.mlhip_g2_from_bytes(cid, b"".join(bad), len(bad), 1, 0, out, st))  # subgroup check off
    assert st.raw[-2] == 0 and out.raw[-8 * n : -4 * n] == R.g2_to_mont_bytes(cp, Qx)


# ---------------------------------------------------------------------------------------------
# alternate code paths selected by environment switches (read when a plan is created / a batch is launched):
# every one must give the same bytes as the default path
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("switch", ["MLHIP_LEGACY_SORT", "MLHIP_ACC32", "MLHIP_REDUCE_ONE_LANE", "MLHIP_REDUCE32", "MLHIP_NO_QUAD_ACC",
                                    "MLHIP_NO_PLAN_CACHE", "MLHIP_ACC_BLOCK=256", "MLHIP_RED_BLOCK=64", "MLHIP_CHUNK_LOG2=3",
                                    "MLHIP_STREAM_SEGMENTS=2", "MLHIP_STREAM_SEGMENTS=5", "MLHIP_REDUCE32+MLHIP_STREAM_SEGMENTS=3",
                                    "MLHIP_SCATTER_STAGED=0", "MLHIP_HOST_THREADS=1"])
def test_msm_alternate_paths(lib, mlhip, curve, switch, monkeypatch):
    g = load_golden(curve)
    cid = g["curve_id"]
    fpb, g1b, _, _ = mlhip.sizes(cid)
    pts, scs, exp = load_msm1000(curve, fpb)
    for sw in switch.split("+"):
        name, _, value = sw.partition("=")
        monkeypatch.setenv(name, value or "1")
    monkeypatch.setenv("MLHIP_NO_PLAN_CACHE", "1")  # the knobs are read when a plan is created: no pooled plan
    for window_c in (9, 16):
        out = ctypes.create_string_buffer(g1b)
        mlhip.check(lib.mlhip_msm_g1(cid, pts, scs, 0, 1000, window_c, out))
        assert out.raw == exp, (curve, switch, window_c)
    for case in g["msm_g1"]:  # the edge cases: infinity inputs, P and -P, duplicates, zero scalars ...
        p = b"".join(_h(x) for x in case["points"])
        sc = b"".join(_h(x) for x in case["scalars"])
        out = ctypes.create_string_buffer(g1b)
        mlhip.check(lib.mlhip_msm_g1(cid, p, sc, 0, len(case["points"]), 5, out))
        assert out.raw == _h(case["expected"]), (curve, switch, case["name"])


@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("switches", [("MLHIP_ACC32",), ("MLHIP_ACC32", "MLHIP_STREAM_SEGMENTS=2"), ("MLHIP_LEGACY_SORT",),
                                      ("MLHIP_REDUCE32",), ("MLHIP_REDUCE32", "MLHIP_STREAM_SEGMENTS=3"),
                                      ("MLHIP_G2_KC",), ("MLHIP_G2_KC", "MLHIP_STREAM_SEGMENTS=3"),
                                      ("MLHIP_G2_KC", "MLHIP_REDUCE32", "MLHIP_STREAM_SEGMENTS=2")])
def test_msm_g2_alternate_paths(lib, mlhip, curve, switches, monkeypatch):
    """G2 on the boundary-form accumulation (MLHIP_ACC32=1), alone and together with a forced segment count: with no
    carry-free copy of the points a BLS12-381 G2 plan cannot stream and must fall back to one pass, not fail.
    MLHIP_G2_KC=1: the carry-free accumulation with the lane pairs split by coordinate (ec28_kc.h, BLS12-381 only)."""
    from oracle import cref

    g = load_golden(curve)
    cid = g["curve_id"]
    _, _, g2b, _ = mlhip.sizes(cid)
    monkeypatch.setenv("MLHIP_NO_PLAN_CACHE", "1")  # the knobs are read when a plan is created
    for sw in switches:
        name, _, value = sw.partition("=")
        monkeypatch.setenv(name, value or "1")
    for case in g["msm_g2"]:
        pts = b"".join(_h(x) for x in case["points"])
        sc = b"".join(_h(x) for x in case["scalars"])
        out = ctypes.create_string_buffer(g2b)
        mlhip.check(lib.mlhip_msm_g2(cid, pts, sc, 0, len(case["points"]), 5, out))
        assert out.raw == _h(case["expected"]), (curve, switches, case["name"])
    n = 3000
    pts = cref.gen_points(cid, 2, 5151, 777, n)
    sc = _rand_scalars(n, 909 + cid, 252)
    exp = cref.msm(cid, 2, pts, sc, n, False, 0, 8)
    for c in (9, 16):
        out = ctypes.create_string_buffer(g2b)
        mlhip.check(lib.mlhip_msm_g2(cid, pts, sc.tobytes(), 0, n, c, out))
        assert out.raw == exp, (curve, switches, c)


@pytest.mark.parametrize("curve", CURVES)
def test_pairing_one_lane_kernels_agree(lib, mlhip, curve, monkeypatch):
    g = load_golden(curve)
    cid = g["curve_id"]
    _, g1b, g2b, gtb = mlhip.sizes(cid)
    cases = g["pairing"]
    p1 = b"".join(_h(c["g1"]) for c in cases)
    p2 = b"".join(_h(c["g2"]) for c in cases)
    exp = b"".join(_h(c["fexp"]) for c in cases)
    monkeypatch.setenv("MLHIP_PAIRING_ONE_LANE", "1")
    out = ctypes.create_string_buffer(gtb * len(cases))
    mlhip.check(lib.mlhip_pairing_batch(cid, p1, p2, len(cases), out))
    assert out.raw == exp


def test_pairing_saturated_lane_pair_kernels_agree(lib, mlhip, monkeypatch):
    """BLS12-381 runs the carry-free lane-pair kernels (k_pairing_lp28) by default; MLHIP_PAIRING_SAT=1 selects the
    saturated lane-pair kernels of round 1 (what BN254 and BLS12-377 run): Miller loop, final exponentiation, fused
    pairing and Pairing2 must give the same bytes on both, on the goldens and on a random batch with infinities."""
    from oracle import cref

    g = load_golden("BLS12-381")
    cid = g["curve_id"]
    _, g1b, g2b, gtb = mlhip.sizes(cid)
    n = 130  # ragged: the last wave is partly idle
    p1 = bytearray(cref.gen_points(cid, 1, 808, 17, n))
    p2 = bytearray(cref.gen_points(cid, 2, 909, 19, n))
    p1[5 * g1b : 6 * g1b] = bytes(g1b)
    p2[9 * g2b : 10 * g2b] = bytes(g2b)
    p1, p2 = bytes(p1), bytes(p2)
    want = cref.pairing_batch(cid, p1, p2, n, 8)
    res = {}
    from conftest import alt_build

    sats = ("0", "1") if alt_build() else ("0",)  # the saturated kernels are in the test build only
    for sat in sats:
        monkeypatch.setenv("MLHIP_PAIRING_SAT", sat)
        out = ctypes.create_string_buffer(gtb * n)
        mlhip.check(lib.mlhip_pairing_batch(cid, p1, p2, n, out))
        assert out.raw == want, sat
        ml = ctypes.create_string_buffer(gtb * (n // 2))
        mlhip.check(lib.mlhip_miller_loop(cid, p1, p2, 2, n // 2, ml))
        fe = ctypes.create_string_buffer(gtb * (n // 2))
        mlhip.check(lib.mlhip_final_exp(cid, ml, n // 2, fe))
        res[sat] = fe.raw
        cases = g["pairing"]
        q1 = b"".join(_h(c["g1"]) for c in cases)
        q2 = b"".join(_h(c["g2"]) for c in cases)
        out = ctypes.create_string_buffer(gtb * len(cases))
        mlhip.check(lib.mlhip_pairing_batch(cid, q1, q2, len(cases), out))
        assert out.raw == b"".join(_h(c["fexp"]) for c in cases)
        # Gt.Exp on raw Miller-loop values (any Fp12 element is a valid input), 40 exponents incl. 0, 1, r - 1
        m = 40
        sc = _rand_scalars(m, 77, 254)
        sc[0] = 0
        sc[1] = (1, 0, 0, 0)
        r_order = int(g["r"], 16) - 1
        sc[2] = [(r_order >> (64 * j)) & (2**64 - 1) for j in range(4)]
        ge = ctypes.create_string_buffer(gtb * m)
        mlhip.check(lib.mlhip_gt_exp(cid, ml.raw[: gtb * m], sc.tobytes(), 0, m, ge))
        res["gt_exp" + sat] = ge.raw
    assert all(res["gt_exp" + x] == res["gt_exp0"] for x in sats)
    assert all(res[x] == cref.final_exp(cid, cref.miller_loop(cid, p1, p2, 2, n // 2, 8), n // 2, 8) for x in sats)


@pytest.mark.parametrize("curve", ["BLS12-381", "BLS12-377", "BN254"])
def test_pairing_quad_lane_kernels_agree(lib, mlhip, curve, monkeypatch):
    """MLHIP_PAIRING_QUAD=1: one pairing per quad of lanes (pairing_quad.h, k_pairing_q28; BLS12-381 and, since round 4,
    BLS12-377 with its D-twist line product and BN254 with its Frobenius lines and hard part) -- Miller loop (compared after the final exponentiation, the raw value is
    not canonical), final exponentiation on the lane-pair kernels' raw Miller values, and the fused pairing, against the
    oracle: goldens and a ragged batch with infinities; Gt.Exp on quads against the lane pairs."""
    from oracle import cref

    g = load_golden(curve)
    cid = g["curve_id"]
    _, g1b, g2b, gtb = mlhip.sizes(cid)
    n = 83  # ragged: the last wave holds 3 quads
    p1 = bytearray(cref.gen_points(cid, 1, 818, 27, n))
    p2 = bytearray(cref.gen_points(cid, 2, 919, 29, n))
    p1[5 * g1b : 6 * g1b] = bytes(g1b)
    p2[9 * g2b : 10 * g2b] = bytes(g2b)
    p1, p2 = bytes(p1), bytes(p2)
    want = cref.pairing_batch(cid, p1, p2, n, 8)
    monkeypatch.setenv("MLHIP_PAIRING_QUAD", "0")
    raw_pairs = ctypes.create_string_buffer(gtb * n)  # raw Miller values from the lane-pair kernels
    mlhip.check(lib.mlhip_miller_loop(cid, p1, p2, 1, n, raw_pairs))
    # batches up to 2^14 run on quads by default; "1" forces them, "0" keeps this small batch on the lane-pair kernels
    gt_exp_results = []
    for quad in ("1", "0", None):
        if quad is None:
            monkeypatch.delenv("MLHIP_PAIRING_QUAD")
        else:
            monkeypatch.setenv("MLHIP_PAIRING_QUAD", quad)
        out = ctypes.create_string_buffer(gtb * n)
        mlhip.check(lib.mlhip_pairing_batch(cid, p1, p2, n, out))
        assert out.raw == want, quad
        fe = ctypes.create_string_buffer(gtb * n)
        mlhip.check(lib.mlhip_final_exp(cid, raw_pairs, n, fe))
        assert fe.raw == want, quad
        ml = ctypes.create_string_buffer(gtb * n)
        mlhip.check(lib.mlhip_miller_loop(cid, p1, p2, 1, n, ml))
        assert cref.final_exp(cid, ml.raw, n, 8) == want, quad
        cases = g["pairing"]
        q1 = b"".join(_h(c["g1"]) for c in cases)
        q2 = b"".join(_h(c["g2"]) for c in cases)
        out = ctypes.create_string_buffer(gtb * len(cases))
        mlhip.check(lib.mlhip_pairing_batch(cid, q1, q2, len(cases), out))
        assert out.raw == b"".join(_h(c["fexp"]) for c in cases), quad
        # Pairing2 (two pairs per product sharing the squarings) runs on quads too (products of up to 4 pairs)
        ml2 = ctypes.create_string_buffer(gtb * (n // 2))
        mlhip.check(lib.mlhip_miller_loop(cid, p1, p2, 2, n // 2, ml2))
        assert cref.final_exp(cid, ml2.raw, n // 2, 8) == cref.final_exp(cid, cref.miller_loop(cid, p1, p2, 2, n // 2, 8), n // 2, 8)
        # Gt.Exp of the raw Miller values (any Fp12 value is a valid input): the same bytes on quads and on lane pairs
        m = 40
        sc = _rand_scalars(m, 4711, 253)
        sc[0] = 0
        sc[1] = (1, 0, 0, 0)
        ge = ctypes.create_string_buffer(gtb * m)
        mlhip.check(lib.mlhip_gt_exp(cid, raw_pairs.raw[: gtb * m], sc.tobytes(), 0, m, ge))
        gt_exp_results.append(ge.raw)
    assert gt_exp_results[0] == gt_exp_results[1] == gt_exp_results[2]
    one = ctypes.create_string_buffer(gtb)  # x^0 = 1 and x^1 = x
    mlhip.check(lib.mlhip_final_exp(cid, bytes(gtb), 0, one))
    assert gt_exp_results[0][gtb : 2 * gtb] == raw_pairs.raw[gtb : 2 * gtb]


@pytest.mark.parametrize("curve", CURVES)
def test_msm_differential_sweep(lib, mlhip, curve):
    """Seeded sweep over (n, window, scalar width, duplicates / negated duplicates / infinities) against the C
    oracle: the shapes the fixed cases do not hit -- ragged n around the sort tile and chunk sizes, buckets with a
    point and its negative (accumulator returns to infinity mid-bucket), runs of the same point (doubling path
    inside the carry-free accumulation), scalars at r-1 / 2^k boundaries."""
    import numpy as np
    import random
    from oracle import cref

    g = load_golden(curve)
    cid = g["curve_id"]
    _, g1b, g2b, _ = mlhip.sizes(cid)
    rnd = random.Random(1000 + cid)
    r_order = int(g["r"], 16)

    def scalars(count, bits):
        vals = [rnd.getrandbits(bits) % r_order for _ in range(count)]
        vals[0] = r_order - 1
        if count > 2:
            vals[1] = 1 << (bits - 1)
            vals[2] = 0
        return np.frombuffer(b"".join(v.to_bytes(32, "little") for v in vals), dtype=np.uint64).reshape(count, 4).copy()

    for trial in range(14):
        group = 2 if trial % 5 == 4 else 1
        sz = g1b if group == 1 else g2b
        n = rnd.choice([1, 2, 3, 7, 63, 64, 65, 255, 257, 1023, 1025, 2049, 4097]) if trial < 9 else rnd.randrange(1, 3000)
        c = rnd.choice([4, 5, 8, 11, 13, 16]) if trial % 3 else 0
        pts = bytearray(cref.gen_points(cid, group, 5000 + trial, 77 + trial, n))
        sc = scalars(n, rnd.choice([16, 64, 200, 252]))
        # structure: copies of earlier points with equal scalars (same bucket -> doubling), and copies whose
        # scalar is the negative (same bucket, opposite sign -> cancellation), a few infinities
        for k in range(n // 5):
            i, j = rnd.randrange(n), rnd.randrange(n)
            pts[j * sz : (j + 1) * sz] = pts[i * sz : (i + 1) * sz]
            if k % 2 == 0:
                sc[j] = sc[i]
            else:
                v = (r_order - int.from_bytes(sc[i].tobytes(), "little")) % r_order
                sc[j] = np.frombuffer(v.to_bytes(32, "little"), dtype=np.uint64)
        for k in range(n // 40):
            j = rnd.randrange(n)
            pts[j * sz : (j + 1) * sz] = bytes(sz)
        pts = bytes(pts)
        exp = cref.msm(cid, group, pts, sc, n, False, 0, 8)
        out = ctypes.create_string_buffer(sz)
        fn = lib.mlhip_msm_g1 if group == 1 else lib.mlhip_msm_g2
        mlhip.check(fn(cid, pts, sc.tobytes(), 0, n, c, out))
        assert out.raw == exp, (curve, trial, group, n, c)


def test_plan_pool_reuse_and_release(lib, mlhip):
    """the host-buffer entry points reuse pooled plans: same answers when sizes / curves / groups interleave, after
    mlhip_release_cache(), and when a pooled plan (capacity 1000) serves a smaller n"""
    order = ["BLS12-381", "BN254", "BLS12-381", "BLS12-377", "BN254", "BLS12-381"]
    for rep, curve in enumerate(order):
        g = load_golden(curve)
        cid = g["curve_id"]
        fpb, g1b, _, _ = mlhip.sizes(cid)
        pts, scs, exp = load_msm1000(curve, fpb)
        out = ctypes.create_string_buffer(g1b)
        mlhip.check(lib.mlhip_msm_g1(cid, pts, scs, 0, 1000, 12, out))
        assert out.raw == exp
        # a 300-point prefix of the same vectors through the pooled 1000-point plan (capacity <= 4 n), checked by
        # linearity: MSM(first 300) + MSM(rest) == MSM(all)
        a = ctypes.create_string_buffer(g1b)
        b = ctypes.create_string_buffer(g1b)
        mlhip.check(lib.mlhip_msm_g1(cid, pts[: 300 * g1b], scs[: 300 * 32], 0, 300, 12, a))
        mlhip.check(lib.mlhip_msm_g1(cid, pts[300 * g1b :], scs[300 * 32 :], 0, 700, 12, b))
        tot = ctypes.create_string_buffer(g1b)
        mlhip.check(lib.mlhip_g1_sum(cid, a.raw + b.raw, 2, tot))
        assert tot.raw == exp
        if rep == 2:
            assert lib.mlhip_release_cache() == 0


def test_plan_pool_eviction(lib, mlhip):
    """More distinct (curve, window) shapes than the pool holds (16): the least recently used entries are evicted and
    rebuilt on demand; every call still returns the golden result."""
    shapes = [(curve, c) for c in (5, 6, 7, 8, 9, 10, 11) for curve in CURVES]  # 21 shapes
    for rnd in range(2):
        for curve, c in shapes:
            g = load_golden(curve)
            cid = g["curve_id"]
            fpb, g1b, _, _ = mlhip.sizes(cid)
            pts, scs, exp = load_msm1000(curve, fpb)
            out = ctypes.create_string_buffer(g1b)
            mlhip.check(lib.mlhip_msm_g1(cid, pts, scs, 0, 1000, c, out))
            assert out.raw == exp, (curve, c, rnd)
    assert lib.mlhip_release_cache() == 0


def test_concurrent_callers(lib, mlhip):
    """the entry points are called from many OS threads at once (cgo).  Eight threads run
    MSMs and pairing batches on different curves concurrently; every result must match its golden."""
    import threading

    errors = []

    def worker(tid):
        try:
            curve = CURVES[tid % 3]
            g = load_golden(curve)
            cid = g["curve_id"]
            fpb, g1b, _, gtb = mlhip.sizes(cid)
            pts, scs, exp = load_msm1000(curve, fpb)
            cases = g["pairing"]
            p1 = b"".join(_h(c["g1"]) for c in cases)
            p2 = b"".join(_h(c["g2"]) for c in cases)
            pexp = b"".join(_h(c["fexp"]) for c in cases)
            for rep in range(4):
                out = ctypes.create_string_buffer(g1b)
                rc = lib.mlhip_msm_g1(cid, pts, scs, 0, 1000, (0, 8, 12, 16)[(tid + rep) % 4], out)
                if rc != 0 or out.raw != exp:
                    errors.append(("msm", tid, rep, rc))
                gt = ctypes.create_string_buffer(gtb * len(cases))
                rc = lib.mlhip_pairing_batch(cid, p1, p2, len(cases), gt)
                if rc != 0 or gt.raw != pexp:
                    errors.append(("pairing", tid, rep, rc))
        except Exception as e:  # noqa: BLE001
            errors.append(("exception", tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]


@pytest.mark.gpu
def test_bases_shared_by_threads(lib, mlhip):
    """One resident-bases handle used from six threads at once (a Go prover shares its SRS between goroutines):
    the library serializes the calls; every result equals the golden."""
    import threading

    curve = "BLS12-381"
    g = load_golden(curve)
    cid = g["curve_id"]
    fpb, g1b, _, _ = mlhip.sizes(cid)
    pts, scs, exp = load_msm1000(curve, fpb)
    handle = ctypes.c_void_p()
    assert lib.mlhip_bases_create(cid, 1, pts, 1000, 0, ctypes.byref(handle)) == 0
    errors = []

    def worker(tid):
        for rep in range(6):
            out = ctypes.create_string_buffer(g1b)
            rc = lib.mlhip_bases_msm(handle, scs, 0, 1000, out)
            if rc != 0 or out.raw != exp:
                errors.append((tid, rep, rc))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert lib.mlhip_bases_destroy(handle) == 0
    assert not errors, errors[:5]


@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("group", [1, 2])
def test_msm_resident_tiles(lib, mlhip, curve, group, monkeypatch):
    """Device-resident MSMs above 2^23 points are accumulated tile by tile (plan_stream with no uploads: the bucket
    accumulators travel between tiles).  MLHIP_TILE_LOG2 forces small tiles: ragged last tile, uniform and skewed
    scalars (long buckets spanning tiles), infinities, the plan reused with another n, and one tile (= untiled)."""
    import torch
    from oracle import cref

    g = load_golden(curve)
    cid = g["curve_id"]
    _, g1b, g2b, _ = mlhip.sizes(cid)
    sz = g1b if group == 1 else g2b
    n = 5000
    pts = bytearray(cref.gen_points(cid, group, 4242 + group, 99, n))
    pts[7 * sz : 8 * sz] = bytes(sz)
    pts[(n - 1) * sz : n * sz] = bytes(sz)
    pts = bytes(pts)
    uniform = _rand_scalars(n, 1234 + cid, 252)
    import numpy as np

    skew = np.zeros((n, 4), dtype=np.uint64)
    skew[:, 0] = np.random.default_rng(77 + cid).integers(0, 1 << 20, size=n, dtype=np.uint64)
    skew[::3] = skew[0]
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    dp = torch.frombuffer(bytearray(pts), dtype=torch.uint8).to(dev)
    for name, sc in (("uniform", uniform), ("skewed", skew)):
        ds = torch.frombuffer(bytearray(sc.tobytes()), dtype=torch.uint8).to(dev)
        want = cref.msm(cid, group, pts, sc, n, False, 0, 8)
        want_head = cref.msm(cid, group, pts, sc, 3001, False, 0, 8)
        for c in (8, 13):
            plan = mlhip.MsmPlan(cid, group, n, c)
            plan.set_profiling(True)
            for tile in ("10", "9", "12", "13", "0"):  # 5 tiles (ragged), 10, 2, 1 (= untiled), off
                monkeypatch.setenv("MLHIP_TILE_LOG2", tile)
                assert plan.run(dp.data_ptr(), ds.data_ptr(), n, False, st) == want, (curve, group, name, c, tile)
                t = plan.timings()
                assert t["accumulate"] > 0 and t["device_total"] >= t["accumulate"], (tile, t)
                assert plan.run(dp.data_ptr(), ds.data_ptr(), 3001, False, st) == want_head, (curve, group, name, c, tile)
            plan.close()


def test_sort_ahead_helper_allocation_failure(lib, mlhip, monkeypatch):
    """the sort-ahead helper records of a tiled MSM are an optimisation; when their buffers cannot be
    allocated (MLHIP_FAULT_INJECT=sort_helper_alloc stands in for the out-of-memory) the MSM runs with the sort in
    line, gives the same bytes, leaves no half-built helper behind, and the same plan then runs again with and without
    the helpers (it used to keep the half-allocated record and launch the next sort on null list pointers)."""
    import torch
    from oracle import cref

    g = load_golden("BLS12-381")
    cid = g["curve_id"]
    _, g1b, _, _ = mlhip.sizes(cid)
    n = 5000
    pts = cref.gen_points(cid, 1, 515, 7, n)
    sc = _rand_scalars(n, 9090, 252)
    want = cref.msm(cid, 1, pts, sc, n, False, 0, 8)
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    dp = torch.frombuffer(bytearray(pts), dtype=torch.uint8).to(dev)
    ds = torch.frombuffer(bytearray(sc.tobytes()), dtype=torch.uint8).to(dev)
    monkeypatch.setenv("MLHIP_TILE_LOG2", "10")  # five tiles: the sort of tile s + 1 runs ahead of tile s
    plan = mlhip.MsmPlan(cid, 1, n, 12)
    try:
        monkeypatch.setenv("MLHIP_FAULT_INJECT", "sort_helper_alloc")
        assert plan.run(dp.data_ptr(), ds.data_ptr(), n, False, st) == want  # first launch: helpers "fail", in line
        assert plan.run(dp.data_ptr(), ds.data_ptr(), n, False, st) == want  # relaunch on the same plan
        monkeypatch.delenv("MLHIP_FAULT_INJECT")
        assert plan.run(dp.data_ptr(), ds.data_ptr(), n, False, st) == want  # helpers allocate now: sort ahead
        monkeypatch.setenv("MLHIP_FAULT_INJECT", "sort_helper_alloc")
        assert plan.run(dp.data_ptr(), ds.data_ptr(), n, False, st) == want  # existing helpers are kept and used
    finally:
        plan.close()


def test_msm_edwards_trusted_plan(lib, mlhip, monkeypatch):
    """mlhip_msm_plan_assume_srs: a BLS12-377 G1 plan whose caller vouches for the prime-order subgroup sums its
    buckets in twisted Edwards coordinates (ed28.h / msm_ed.h).  Same bytes as the oracle and as the plan without the
    promise: uniform scalars, skewed ones (long buckets: the Weierstrass slice sums folded into the Edwards state, in the
    last tile and in earlier ones), points at infinity, one pass and tiles, a prefix of the points, the promise taken
    back.  Small bucket sets (one bucket per quad of lanes) and the other curves ignore the promise."""
    import numpy as np
    import torch
    from oracle import cref

    g = load_golden("BLS12-377")
    cid = g["curve_id"]
    _, g1b, _, _ = mlhip.sizes(cid)
    n = 6000
    pts = bytearray(cref.gen_points(cid, 1, 777, 31, n))
    pts[5 * g1b : 6 * g1b] = bytes(g1b)
    pts[(n - 1) * g1b : n * g1b] = bytes(g1b)
    pts = bytes(pts)
    uniform = _rand_scalars(n, 4077, 252)
    skew = np.zeros((n, 4), dtype=np.uint64)
    skew[:, 0] = np.random.default_rng(377).integers(0, 1 << 22, size=n, dtype=np.uint64)
    skew[::3] = skew[0]
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    dp = torch.frombuffer(bytearray(pts), dtype=torch.uint8).to(dev)
    for name, sc in (("uniform", uniform), ("skewed", skew)):
        ds = torch.frombuffer(bytearray(sc.tobytes()), dtype=torch.uint8).to(dev)
        want = cref.msm(cid, 1, pts, sc, n, False, 0, 8)
        want_head = cref.msm(cid, 1, pts, sc, 2999, False, 0, 8)
        for c in (13, 16):
            plan = mlhip.MsmPlan(cid, 1, n, c)
            plan.set_profiling(True)
            plan.assume_srs(True)
            for tile in ("0", "11", "10", "12"):  # one pass; 3 tiles (ragged); 6 tiles; 2 tiles
                monkeypatch.setenv("MLHIP_TILE_LOG2", tile)
                assert plan.run(dp.data_ptr(), ds.data_ptr(), n, False, st) == want, (name, c, tile)
                assert plan.timings()["edwards"] == 1.0
                assert plan.run(dp.data_ptr(), ds.data_ptr(), 2999, False, st) == want_head, (name, c, tile)
            monkeypatch.setenv("MLHIP_EDWARDS", "0")  # the switch: the Weierstrass kernels, whatever the promise
            assert plan.run(dp.data_ptr(), ds.data_ptr(), n, False, st) == want
            assert plan.timings()["edwards"] == 0.0
            monkeypatch.delenv("MLHIP_EDWARDS")
            plan.assume_srs(False)
            assert plan.run(dp.data_ptr(), ds.data_ptr(), n, False, st) == want
            assert plan.timings()["edwards"] == 0.0
            plan.close()
    # the G1 half of a shared-scalar launch (mlhip_msm_launch_shared) carries the promise: Edwards bucket sums for G1, the G2
    # plan reads the same entry lists -- one pass and tiles
    _, _, g2b, _ = mlhip.sizes(cid)
    p2 = cref.gen_points(cid, 2, 779, 33, n)
    d2 = torch.frombuffer(bytearray(p2), dtype=torch.uint8).to(dev)
    ds = torch.frombuffer(bytearray(skew.tobytes()), dtype=torch.uint8).to(dev)
    want1, want2 = cref.msm(cid, 1, pts, skew, n, False, 0, 8), cref.msm(cid, 2, p2, skew, n, False, 0, 8)
    a, b = mlhip.MsmPlan(cid, 1, n, 13), mlhip.MsmPlan(cid, 2, n, 13)
    a.set_profiling(True)
    a.assume_srs(True)
    for tile in ("0", "11"):
        monkeypatch.setenv("MLHIP_TILE_LOG2", tile)
        a.launch_shared(b, dp.data_ptr(), d2.data_ptr(), ds.data_ptr(), n, False, st)
        assert a.finish() == want1 and b.finish() == want2, tile
        assert a.timings()["edwards"] == 1.0
    a.close()
    b.close()
    monkeypatch.setenv("MLHIP_TILE_LOG2", "0")
    ds = torch.frombuffer(bytearray(uniform.tobytes()), dtype=torch.uint8).to(dev)
    small = mlhip.MsmPlan(cid, 1, n, 8)  # 32 x 128 buckets: one bucket per quad of lanes, on XYZZ
    small.assume_srs(True)
    assert small.run(dp.data_ptr(), ds.data_ptr(), n, False, st) == cref.msm(cid, 1, pts, uniform, n, False, 0, 8)
    assert small.timings()["edwards"] == 0.0
    small.close()
    g381 = load_golden("BLS12-381")  # no Edwards model: the promise changes nothing
    c381 = g381["curve_id"]
    p381 = cref.gen_points(c381, 1, 778, 32, 2000)
    s381 = _rand_scalars(2000, 4078, 252)
    plan = mlhip.MsmPlan(c381, 1, 2000, 13)
    plan.assume_srs(True)
    d1 = torch.frombuffer(bytearray(p381), dtype=torch.uint8).to(dev)
    d2 = torch.frombuffer(bytearray(s381.tobytes()), dtype=torch.uint8).to(dev)
    assert plan.run(d1.data_ptr(), d2.data_ptr(), 2000, False, st) == cref.msm(c381, 1, p381, s381, 2000, False, 0, 8)
    assert plan.timings()["edwards"] == 0.0
    plan.close()


def test_bases_edwards_only_for_subgroup_tables(lib, mlhip, monkeypatch):
    """mlhip_bases_create checks a BLS12-377 G1 table on the device: all points in the prime-order subgroup -> the table's
    MSMs sum their buckets in twisted Edwards coordinates; one curve point outside G1 (or one point off the curve) -> the
    Weierstrass kernels, and the reference's answer for that input.  Either way the bytes are the oracle's."""
    from oracle import cref
    from oracle import pyref as R

    cp = R.CURVES["BLS12-377"]
    cid = cp.curve_id
    _, g1b, _, _ = mlhip.sizes(cid)
    n = 5000
    good = cref.gen_points(cid, 1, 91, 17, n)
    sc = _rand_scalars(n, 8377, 252)
    x = 5
    while True:  # a curve point outside G1: almost every curve point is (the cofactor has 94 bits)
        y = R.fp_sqrt((x * x * x + cp.b) % cp.p, cp.p)
        if y is not None and R.g1_mul_unreduced(cp, (x, y), cp.r) is not None:
            break
        x += 1
    outside = bytearray(good)
    outside[11 * g1b : 12 * g1b] = R.g1_to_mont_bytes(cp, (x, y))
    off_curve = bytearray(good)
    off_curve[3 * g1b : 3 * g1b + 4] = bytes(4)  # some x limb zeroed: not on the curve
    for c in (13, 16):
        for name, pts, checked in (("subgroup", good, 1), ("outside", bytes(outside), 0), ("off-curve", bytes(off_curve), 0)):
            h = ctypes.c_void_p()
            mlhip.check(lib.mlhip_bases_create(cid, 1, pts, n, c, ctypes.byref(h)))
            assert lib.mlhip_bases_checked_subgroup(h) == checked, name
            if name != "off-curve":  # (the oracle's formulas are for curve points)
                want = cref.msm(cid, 1, pts, sc, n, False, 0, 8)
                for k in (n, 1234, n):
                    out = ctypes.create_string_buffer(g1b)
                    mlhip.check(lib.mlhip_bases_msm(h, sc.tobytes(), 0, k, out))
                    assert out.raw == (want if k == n else cref.msm(cid, 1, pts, sc, k, False, 0, 8)), (name, c, k)
            mlhip.check(lib.mlhip_bases_destroy(h))
    monkeypatch.setenv("MLHIP_EDWARDS", "0")
    h = ctypes.c_void_p()
    mlhip.check(lib.mlhip_bases_create(cid, 1, good, n, 13, ctypes.byref(h)))
    assert lib.mlhip_bases_checked_subgroup(h) == 0
    mlhip.check(lib.mlhip_bases_destroy(h))
    g381 = load_golden("BLS12-381")
    h = ctypes.c_void_p()
    mlhip.check(lib.mlhip_bases_create(g381["curve_id"], 1, cref.gen_points(g381["curve_id"], 1, 1, 2, 100), 100, 0, ctypes.byref(h)))
    assert lib.mlhip_bases_checked_subgroup(h) == 0  # no Edwards model: nothing to check
    mlhip.check(lib.mlhip_bases_destroy(h))


@pytest.mark.parametrize("curve", CURVES)
def test_msm_shared_scalars(lib, mlhip, curve, monkeypatch):
    """mlhip_msm_launch_shared: the G1 and the G2 MSM of one scalar vector, sorted once in the G1 plan and accumulated
    by both -- one pass and tile by tile, uniform and skewed scalars (long buckets), infinities; plans of different
    widths and plans on the boundary-form accumulation cannot share and must fall back to two launches."""
    import numpy as np
    import torch
    from oracle import cref

    g = load_golden(curve)
    cid = g["curve_id"]
    _, g1b, g2b, _ = mlhip.sizes(cid)
    n = 4000
    p1 = bytearray(cref.gen_points(cid, 1, 31, 5, n))
    p2 = bytearray(cref.gen_points(cid, 2, 32, 6, n))
    p1[3 * g1b : 4 * g1b] = bytes(g1b)
    p2[(n - 2) * g2b : (n - 1) * g2b] = bytes(g2b)
    p1, p2 = bytes(p1), bytes(p2)
    uniform = _rand_scalars(n, 4321 + cid, 252)
    skew = np.zeros((n, 4), dtype=np.uint64)
    skew[:, 0] = np.random.default_rng(5 + cid).integers(0, 1 << 18, size=n, dtype=np.uint64)
    skew[::2] = skew[1]
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    d1 = torch.frombuffer(bytearray(p1), dtype=torch.uint8).to(dev)
    d2 = torch.frombuffer(bytearray(p2), dtype=torch.uint8).to(dev)
    for name, sc in (("uniform", uniform), ("skewed", skew)):
        ds = torch.frombuffer(bytearray(sc.tobytes()), dtype=torch.uint8).to(dev)
        want1 = cref.msm(cid, 1, p1, sc, n, False, 0, 8)
        want2 = cref.msm(cid, 2, p2, sc, n, False, 0, 8)
        from conftest import alt_build

        for env, c1, c2 in (({}, 12, 12), ({"MLHIP_TILE_LOG2": "10"}, 12, 12), ({"MLHIP_TILE_LOG2": "9"}, 8, 8), ({}, 12, 9)) + (
                (({"MLHIP_ACC32": "1"}, 12, 12),) if alt_build() else ()):  # plans that cannot share: test build only
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            a = mlhip.MsmPlan(cid, 1, n, c1)
            b = mlhip.MsmPlan(cid, 2, n, c2)
            for m in (n, 1777, 0, n):  # the plans are reused, also with fewer pairs and with none
                a.launch_shared(b, d1.data_ptr(), d2.data_ptr(), ds.data_ptr(), m, False, st)
                r1, r2 = a.finish(), b.finish()
                if m == n:
                    assert (r1, r2) == (want1, want2), (curve, name, env, c1, c2)
                elif m == 0:
                    assert r1 == bytes(g1b) and r2 == bytes(g2b)
                else:
                    assert r1 == cref.msm(cid, 1, p1, sc, m, False, 0, 8) and r2 == cref.msm(cid, 2, p2, sc, m, False, 0, 8)
            a.close()
            b.close()
            for k in env:
                monkeypatch.delenv(k)
    # argument checks: the plans in the wrong order, a plan with a launch still pending, more pairs than a plan holds
    a, b = mlhip.MsmPlan(cid, 1, n, 12), mlhip.MsmPlan(cid, 2, 100, 12)
    args = (ctypes.c_void_p(d1.data_ptr()), ctypes.c_void_p(d2.data_ptr()), ctypes.c_void_p(ds.data_ptr()), 0)
    assert lib.mlhip_msm_launch_shared(b._h, a._h, *args, 10, ctypes.c_void_p(st)) == mlhip.EINVAL
    assert lib.mlhip_msm_launch_shared(a._h, b._h, *args, 101, ctypes.c_void_p(st)) == mlhip.EINVAL
    a.launch(d1.data_ptr(), ds.data_ptr(), 50, False, st)
    assert lib.mlhip_msm_launch_shared(a._h, b._h, *args, 50, ctypes.c_void_p(st)) == mlhip.EINVAL
    assert a.finish() == cref.msm(cid, 1, p1, sc, 50, False, 0, 8)  # the pending launch is untouched
    a.launch_shared(b, d1.data_ptr(), d2.data_ptr(), ds.data_ptr(), 100, False, st)
    assert a.finish() == cref.msm(cid, 1, p1, sc, 100, False, 0, 8) and b.finish() == cref.msm(cid, 2, p2, sc, 100, False, 0, 8)
    a.close()
    b.close()


@pytest.mark.parametrize("curve", CURVES)
def test_msm_g1g2_host_buffers(lib, mlhip, curve, monkeypatch):
    """mlhip_msm_g1g2 (host slices, one scalar vector for the G1 and the G2 MSM) equals the oracle's two MSMs: one pass,
    forced segments (ragged), and a size the library streams by itself; n = 0 gives both identities."""
    from oracle import cref

    g = load_golden(curve)
    cid = g["curve_id"]
    _, g1b, g2b, _ = mlhip.sizes(cid)

    def run(p1, p2, sc, n, c):
        o1, o2 = ctypes.create_string_buffer(g1b), ctypes.create_string_buffer(g2b)
        mlhip.check(lib.mlhip_msm_g1g2(cid, p1, p2, sc.tobytes() if n else None, 0, n, c, o1, o2))
        return o1.raw, o2.raw

    n = 4000
    p1 = cref.gen_points(cid, 1, 71, 3, n)
    p2 = cref.gen_points(cid, 2, 72, 4, n)
    sc = _rand_scalars(n, 99 + cid, 252)
    want = (cref.msm(cid, 1, p1, sc, n, False, 0, 8), cref.msm(cid, 2, p2, sc, n, False, 0, 8))
    assert run(p1, p2, sc, n, 0) == want
    for seg in ("3", "7"):
        monkeypatch.setenv("MLHIP_STREAM_SEGMENTS", seg)
        assert run(p1, p2, sc, n, 11) == want, (curve, seg)
    monkeypatch.delenv("MLHIP_STREAM_SEGMENTS")
    assert run(None, None, sc, 0, 0) == (bytes(g1b), bytes(g2b))
    if curve == "BLS12-381":  # 2 segments of 2^17 pairs chosen by the library
        n = (1 << 18) + 5
        p1 = cref.gen_points(cid, 1, 73, 3, n)
        p2 = cref.gen_points(cid, 2, 74, 4, n)
        sc = _rand_scalars(n, 100 + cid, 252)
        threads = max(1, min(64, len(os.sched_getaffinity(0))))
        assert run(p1, p2, sc, n, 16) == (cref.msm(cid, 1, p1, sc, n, False, 16, threads), cref.msm(cid, 2, p2, sc, n, False, 16, threads))


@pytest.mark.parametrize("group", [1, 2])
@pytest.mark.parametrize("curve", CURVES)
def test_bases_shifted_tables(lib, mlhip, curve, group, monkeypatch):
    """Shifted-base tables of a resident-bases handle (include/mlhip.h: mlhip_bases_create; msm_fold.h): rows 2^(c j) P_i,
    one bucket set for all digits, bucket groups in the reduction, the group-weight host tail.  Against the C oracle: several
    digit widths (one bucket group and many), tables of one tile and of several (ragged last tile), prefixes of the bases,
    host and device scalars, segment schedules that must be cut at the tile boundaries, bases at infinity, scalars 0 / 1 /
    r - 1 / all equal (long buckets: every digit of every scalar in one bucket), and the plain plan of the same bases."""
    import numpy as np
    import torch
    from oracle import cref
    from oracle import pyref as R

    g = load_golden(curve)
    cid = g["curve_id"]
    r = R.CURVES[curve].r
    g1b = mlhip.sizes(cid)[group]  # (bytes of a point of this group)
    n = 6000 if group == 1 else 2500
    pts = bytearray(cref.gen_points(cid, group, 55, 7, n))
    pts[17 * g1b : 18 * g1b] = bytes(g1b)  # bases at infinity
    pts[(n - 1) * g1b : n * g1b] = bytes(g1b)
    pts = bytes(pts)
    sc = _rand_scalars(n, 4100 + cid, 252)
    for i, v in ((0, 0), (1, 1), (2, r - 1), (3, r - 1), (4, (1 << 200) - 1)):
        sc[i] = [(v >> (64 * k)) & ((1 << 64) - 1) for k in range(4)]
    same = np.tile(np.array([[(0x123456789ABCDEF0F0E1D2C3B4A59687 >> (64 * k)) & ((1 << 64) - 1) for k in range(2)] + [77, 5]], dtype=np.uint64), (n, 1))
    want = {k: cref.msm(cid, group, pts, sc, k, False, 0, 8) for k in (n, 1, 2199)}
    want_same = cref.msm(cid, group, pts, same, n, False, 0, 8)
    dev = torch.device("cuda", 0)
    d_sc = torch.frombuffer(bytearray(sc.tobytes()), dtype=torch.uint8).to(dev)
    monkeypatch.setenv("MLHIP_BASES_TABLES", "1")
    for c, tile_lg in ((13, 20), (17, 20), (19, 11), (20, 9), (8, 12), (5, 10)) if group == 1 else ((17, 20), (20, 10), (9, 8)):
        monkeypatch.setenv("MLHIP_FOLD_WINDOW", str(c))
        monkeypatch.setenv("MLHIP_FOLD_TILE_LOG2", str(tile_lg))
        h = ctypes.c_void_p()
        mlhip.check(lib.mlhip_bases_create(cid, group, pts, n, 0, ctypes.byref(h)))
        plan = lib.mlhip_bases_plan(h)
        assert plan
        out = ctypes.create_string_buffer(g1b)
        for k in (n, 1, 2199, n):
            mlhip.check(lib.mlhip_bases_msm(h, sc.tobytes(), 0, k, out))
            assert out.raw == want[k], (curve, c, tile_lg, k)
        t = mlhip.plan_timings(lib, plan)
        assert t["tables"] == 1.0 and t["window_c"] == c, t
        mlhip.check(lib.mlhip_bases_msm_device(h, d_sc.data_ptr(), 0, n, torch.cuda.current_stream().cuda_stream, out))
        assert out.raw == want[n], (curve, c, tile_lg, "device scalars")
        mlhip.check(lib.mlhip_bases_msm(h, same.tobytes(), 0, n, out))
        assert out.raw == want_same, (curve, c, tile_lg, "equal scalars")
        if c == 17:  # the same handle made from points that are already on the device
            d_pts = torch.frombuffer(bytearray(pts), dtype=torch.uint8).to(dev)
            h2 = ctypes.c_void_p()
            mlhip.check(lib.mlhip_bases_create_device(cid, group, d_pts.data_ptr(), n, 0, ctypes.byref(h2)))
            d_pts.zero_()  # (the handle holds its own copy)
            torch.cuda.synchronize()
            mlhip.check(lib.mlhip_bases_msm(h2, sc.tobytes(), 0, n, out))
            assert out.raw == want[n], (curve, c, "device points")
            mlhip.check(lib.mlhip_bases_destroy(h2))
        for seg in ("5", "2"):  # streamed scalars, equal segments: cut again at the tile boundaries
            monkeypatch.setenv("MLHIP_STREAM_SEGMENTS", seg)
            mlhip.check(lib.mlhip_bases_msm(h, sc.tobytes(), 0, n, out))
            assert out.raw == want[n], (curve, c, tile_lg, "segments", seg)
        monkeypatch.delenv("MLHIP_STREAM_SEGMENTS")
        monkeypatch.setenv("MLHIP_STREAM_SCHEDULE", "1,2,5")
        mlhip.check(lib.mlhip_bases_msm(h, sc.tobytes(), 0, n, out))
        assert out.raw == want[n], (curve, c, tile_lg, "schedule")
        monkeypatch.delenv("MLHIP_STREAM_SCHEDULE")
        mlhip.check(lib.mlhip_bases_destroy(h))
    # digits below 5 bits are too many for the sort's 16-bit block counts: the width falls back to the default
    monkeypatch.setenv("MLHIP_FOLD_WINDOW", "4")
    h = ctypes.c_void_p()
    mlhip.check(lib.mlhip_bases_create(cid, group, pts, n, 4, ctypes.byref(h)))
    t = mlhip.plan_timings(lib, lib.mlhip_bases_plan(h))
    assert t["tables"] == 1.0 and t["window_c"] == (13 if n < 4096 else 14 if n < 8192 else 16 if n < 65536 else 20), t
    out = ctypes.create_string_buffer(g1b)
    mlhip.check(lib.mlhip_bases_msm(h, sc.tobytes(), 0, n, out))
    assert out.raw == want[n]
    mlhip.check(lib.mlhip_bases_destroy(h))
    # no tables: an explicit window width, or the switch off
    monkeypatch.delenv("MLHIP_FOLD_WINDOW")
    monkeypatch.delenv("MLHIP_FOLD_TILE_LOG2")
    monkeypatch.setenv("MLHIP_BASES_TABLES", "0")
    h = ctypes.c_void_p()
    mlhip.check(lib.mlhip_bases_create(cid, group, pts, n, 0, ctypes.byref(h)))
    assert mlhip.plan_timings(lib, lib.mlhip_bases_plan(h))["tables"] == 0.0
    out = ctypes.create_string_buffer(g1b)
    mlhip.check(lib.mlhip_bases_msm(h, sc.tobytes(), 0, n, out))
    assert out.raw == want[n]
    mlhip.check(lib.mlhip_bases_destroy(h))
