"""The C restatement (oracle/cref) against the committed golden vectors and the Python oracle."""
import pytest

from conftest import load_golden, load_msm1000
from oracle import cref
from oracle import pyref as R

h = bytes.fromhex
CURVES = ["BN254", "BLS12-381", "BLS12-377"]


@pytest.mark.parametrize("name", CURVES)
def test_cref_field_and_msm_golden(name):
    cp = R.CURVES[name]
    g = load_golden(name)
    cid = g["curve_id"]
    for c in g["fp_mul"]:
        assert cref.fp_mul(cid, h(c["a"]), h(c["b"])) == h(c["ab"])
    for case in g["msm_g1"]:
        pts = b"".join(h(p) for p in case["points"])
        sc = b"".join(h(s) for s in case["scalars"])
        for cc, th in ((0, 1), (3, 2), (8, 3), (16, 1)):
            assert cref.msm(cid, 1, pts, sc, len(case["points"]), False, cc, th) == h(case["expected"]), (name, case["name"], cc)
    for case in g["msm_g2"]:
        pts = b"".join(h(p) for p in case["points"])
        sc = b"".join(h(s) for s in case["scalars"])
        assert cref.msm(cid, 2, pts, sc, len(case["points"]), False, 0, 2) == h(case["expected"]), (name, case["name"])
    pts, sc, exp = load_msm1000(name, cp.fp_bytes)
    assert cref.msm(cid, 1, pts, sc, 1000, False, 0, 4) == exp
    case = next(c for c in g["msm_g1"] if c["name"] == "n10_random")
    scm = b"".join((int(s) % cp.r * (1 << 256) % cp.r).to_bytes(32, "little") for s in case["scalars_int"])
    assert cref.msm(cid, 1, b"".join(h(p) for p in case["points"]), scm, 10, True) == h(case["expected"])
    assert cref.msm(cid, 1, b"", b"", 0) == bytes(2 * cp.fp_bytes)


@pytest.mark.parametrize("name", CURVES)
def test_cref_pairing_golden(name):
    cp = R.CURVES[name]
    g = load_golden(name)
    cid = g["curve_id"]
    gt = cp.fp_bytes * 12
    n = len(g["pairing"])
    out = cref.pairing_batch(cid, b"".join(h(c["g1"]) for c in g["pairing"]), b"".join(h(c["g2"]) for c in g["pairing"]), n, 2)
    for i, c in enumerate(g["pairing"]):
        assert out[i * gt : (i + 1) * gt] == h(c["fexp"]), (name, i)
    p2 = g["pairing2"]
    ml = cref.miller_loop(cid, b"".join(h(x) for x in p2["g1"]), b"".join(h(x) for x in p2["g2"]), 2, 1)
    assert cref.final_exp(cid, ml, 1) == h(p2["fexp"])
    assert cref.final_exp(cid, h(g["fexp_io"]["input"]), 1) == h(g["fexp_io"]["output"])
    assert cref.gt_mul(cid, h(g["pairing"][1]["fexp"]), h(g["pairing"][2]["fexp"]), 1) == h(p2["fexp"])
    bl = g["bilinear"]
    assert cref.pairing_batch(cid, h(bl["g1"]), h(bl["g2"]), 1) == h(bl["fexp"])


@pytest.mark.parametrize("name", CURVES)
def test_cref_input_generator(name):
    cp = R.CURVES[name]
    cid = cp.curve_id
    k0, k1 = 12345678901234567890123, 987654321987654321
    ps, qs = cp.fp_bytes * 2, cp.fp_bytes * 4
    gp = cref.gen_points(cid, 1, k0, k1, 70)
    for i in (0, 1, 69):
        assert R.g1_from_mont_bytes(cp, gp[i * ps : (i + 1) * ps]) == R.g1_mul(cp, cp.g1, k0 + i * k1)
    gq = cref.gen_points(cid, 2, k0, k1, 5)
    assert R.g2_from_mont_bytes(cp, gq[3 * qs : 4 * qs]) == R.g2_mul(cp, R.g2_generator(cp), k0 + 3 * k1)
    assert cref.point_mul(cid, 1, R.g1_to_mont_bytes(cp, cp.g1), k0) == gp[:ps]
