"""Pins oracle/pyref.py to every known-answer value the reference holds for the hot path
(SURVEY.md 8c).  The reference has NO golden vector for an MSM / Miller-loop / FExp output
(math_test.go uses crypto/rand inputs and algebraic checks), so those outputs are 'parity unpinned'
by the reference; what it does pin -- constants, generators, group orders, the Montgomery layout --
is checked here, plus the algebraic identities its own tests assert (math_test.go:323-470)."""
import pytest

from conftest import load_golden
from oracle import pyref as R

# /root/reference/driver/kilic/custom.go:26
KILIC_MODULUS = [0xB9FEFFFFFFFFAAAB, 0x1EABFFFEB153FFFF, 0x6730D2A0F6B0F624, 0x64774B84F38512BF, 0x4B1BA7B6434BACD7, 0x1A0111EA397FE69A]
# /root/reference/driver/kilic/custom.go:29  (r1 = R mod p, R = 2^384)
KILIC_R1 = [0x760900000002FFFD, 0xEBF4000BC40C0002, 0x5F48985753C758BA, 0x77CE585370525745, 0x5C071A97A256EC6D, 0x15F65EC3FA80E493]
# /root/reference/driver/kilic/custom.go:329-336  (F = 2^256 * R mod p)
KILIC_F = [0x75B3CD7C5CE820F, 0x3EC6BA621C3EDB0B, 0x168A13D82BFF6BCE, 0x87663C4BF8C449D2, 0x15F34C83DDC8D830, 0xF9628B49CAA2E85]
# /root/reference/driver/kilic/custom_generic.go:64 (-p^-1 mod 2^64) and :65-75 (p as decimal limbs)
KILIC_INV = 9940570264628428797
KILIC_P_DEC = [13402431016077863595, 2210141511517208575, 7435674573564081700, 7239337960414712511, 5412103778470702295, 1873798617647539866]
# /root/reference/math_test.go:261-270 (expectedModuli, by CurveID)
MODULI = {
    "BN254": "30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001",
    "BLS12-381": "73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001",
    "BLS12-377": "12ab655e9a2ca55660b44d1e5c37b00159aa76fed00000010a11800000000001",
}
# /root/reference/math_test.go:250-259 (expectedG1Gens)
G1_GENS = {
    "BN254": (1, 2),
    "BLS12-381": (
        3685416753713387016781088315183077757961620795782546409894578378688607592378376318836054947676345821548104185464507,
        1339506544944476473020471379941921221584933875938349620426543736416511423956333506472724655353366534992391756441569,
    ),
    "BLS12-377": (
        81937999373150964239938255573465948239988671502647976594219695644855304257327692006745978603320413799295628339695,
        241266749859715473739788878240585681733927191168601896383759122102112907357779751001206799952863815012735208165030,
    ),
}


# /root/reference/driver/kilic/custom.go:31-43 (swuParamsForG1, every value in Montgomery form): the only
# reference-held vectors that exercise a field MULTIPLICATION.  z = 11 (the SWU non-square), zInv holds -1/z,
# minusBOverA holds -b/a, so with mont_mul(x, y) = x y R^-1 mod p:
#     mont_mul(z, zInv)        == p - r1   (Montgomery form of -1)
#     mont_mul(minusBOverA, a) == p - b    (Montgomery form of -b)
KILIC_SWU_A = [0x2F65AA0E9AF5AA51, 0x86464C2D1E8416C3, 0xB85CE591B7BD31E2, 0x27E11C91B5F24E7C, 0x28376EDA6BFC1835, 0x155455C3E5071D85]
KILIC_SWU_B = [0xFB996971FE22A1E0, 0x9AA93EB35B742D6F, 0x8C476013DE99C5C4, 0x873E27C3A221E571, 0xCA72B5E45A52D888, 0x06824061418A386B]
KILIC_SWU_Z = [0x886C00000023FFDC, 0x0F70008D3090001D, 0x77672417ED5828C3, 0x9DAC23E943DC1740, 0x50553F1B9C131521, 0x078C712FBE0AB6E8]
KILIC_SWU_ZINV = [0x0E8A2E8BA2E83E10, 0x5B28BA2CA4D745D1, 0x678CD5473847377A, 0x4C506DD8A8076116, 0x9BCB227D79284139, 0x0E8D3154B0BA099A]
KILIC_SWU_MINUS_B_OVER_A = [0x052583C93555A7FE, 0x3B40D72430F93C82, 0x1B75FAA0105EC983, 0x2527E7DC63851767, 0x99FFFD1F34FC181D, 0x097CAB54770CA0D3]


def _le(limbs):
    return b"".join(x.to_bytes(8, "little") for x in limbs)


def _int(limbs):
    return sum(x << (64 * i) for i, x in enumerate(limbs))


def reference_held_products():
    """(a, b, a*b) triples in the ABI's byte form whose three members are all fixed by the reference's constants"""
    p = _int(KILIC_MODULUS)
    minus_one = (p - _int(KILIC_R1)).to_bytes(48, "little")
    minus_b = (p - _int(KILIC_SWU_B)).to_bytes(48, "little")
    return [
        (_le(KILIC_SWU_Z), _le(KILIC_SWU_ZINV), minus_one),
        (_le(KILIC_SWU_ZINV), _le(KILIC_SWU_Z), minus_one),
        (_le(KILIC_SWU_MINUS_B_OVER_A), _le(KILIC_SWU_A), minus_b),
        (_le(KILIC_SWU_A), _le(KILIC_SWU_MINUS_B_OVER_A), minus_b),
        (_le(KILIC_R1), _le(KILIC_R1), _le(KILIC_R1)),  # 1 * 1 = 1
        (_le(KILIC_F), _le(KILIC_R1), _le(KILIC_F)),  # x * 1 = x on the 2^256 R constant (custom.go:329-336)
    ]


def _limbs(v, n):
    return [(v >> (64 * i)) & (2**64 - 1) for i in range(n)]


def test_bls12_381_field_constants_match_reference():
    cp = R.BLS12_381
    assert _limbs(cp.p, 6) == KILIC_MODULUS == KILIC_P_DEC
    assert (-pow(cp.p, -1, 1 << 64)) % (1 << 64) == KILIC_INV
    assert _limbs(cp.R % cp.p, 6) == KILIC_R1
    assert _limbs((1 << 256) * cp.R % cp.p, 6) == KILIC_F
    # the in-memory form of 1 is r1 (Montgomery, little-endian limbs)
    assert R.fp_to_mont_bytes(cp, 1) == b"".join(x.to_bytes(8, "little") for x in KILIC_R1)


def test_reference_held_montgomery_products():
    """driver/kilic/custom.go:38-42: the oracle's field multiplication (Python integers and the C restatement's
    CIOS) reproduces the products the reference's constants fix; z really is 11."""
    from oracle import cref

    cp = R.BLS12_381
    assert R.fp_from_mont_bytes(cp, _le(KILIC_SWU_Z)) == 11
    for a, b, ab in reference_held_products():
        x, y = R.fp_from_mont_bytes(cp, a), R.fp_from_mont_bytes(cp, b)
        assert R.fp_to_mont_bytes(cp, x * y % cp.p) == ab
        assert cref.fp_mul(cp.curve_id, a, b) == ab


def test_reference_held_products_through_the_kernel_headers(hostmath):
    """the product's own fp_mul (fp.h, built for the host by tests/hostmath) on the same reference-held vectors;
    the GPU run of the same table is tests/test_gpu_parity.py::test_fp_mul_reference_held_products"""
    import ctypes

    for a, b, ab in reference_held_products():
        out = ctypes.create_string_buffer(48)
        assert hostmath.hm_fp_op(1, 0, a, b, out) == 0
        assert out.raw == ab


@pytest.mark.parametrize("name", ["BN254", "BLS12-381", "BLS12-377"])
def test_group_orders_and_generators_match_reference(name):
    cp = R.CURVES[name]
    assert "%x" % cp.r == MODULI[name]
    assert cp.g1 == G1_GENS[name]
    assert R.g1_is_on_curve(cp, cp.g1)
    assert R.g1_mul_unreduced(cp, cp.g1, cp.r) is None
    q = R.g2_generator(cp)
    assert R.g2_is_on_curve(cp, q) and R.g2_mul_unreduced(cp, q, cp.r) is None
    # byte sizes the reference asserts through c.G1ByteSize etc. (math_test.go:309, 378-379)
    assert cp.fp_bytes == (32 if name == "BN254" else 48)


@pytest.mark.parametrize("name", ["BN254", "BLS12-381", "BLS12-377"])
def test_final_exponent_is_the_cofactor_times_p12m1_over_r(name):
    """hard-part identities recomputed"""
    cp = R.CURVES[name]
    p, r, x = cp.p, cp.r, cp.x
    assert (p**4 - p * p + 1) % r == 0
    if cp.family == "BLS12":
        assert (x - 1) ** 2 * (x + p) * (x * x + p * p - 1) + 3 == 3 * ((p**4 - p * p + 1) // r)
        assert cp.fexp_cofactor == 3
    else:
        l0 = 1 + 6 * x + 12 * x * x + 12 * x**3
        l1 = 4 * x + 6 * x * x + 12 * x**3
        l2 = 6 * x + 6 * x * x + 12 * x**3
        l3 = -1 + 4 * x + 6 * x * x + 12 * x**3
        assert l0 + l1 * p + l2 * p * p + l3 * p**3 == cp.fexp_cofactor * ((p**4 - p * p + 1) // r)


@pytest.mark.parametrize("name", ["BN254", "BLS12-381", "BLS12-377"])
def test_pairing_identities_of_the_reference_tests(name):
    """runPairingTest / runGtTest (math_test.go:423-470) restated on the oracle"""
    cp = R.CURVES[name]
    T = R.tower(cp)
    d = R.Drbg("pinned/" + name)
    q = R.g2_generator(cp)
    e = R.pairing(cp, cp.g1, q)
    assert not T.f12_is_one(e) and T.f12_is_one(T.f12_pow(e, cp.r))
    a, b = d.below(cp.r), d.below(cp.r)
    assert R.pairing(cp, R.g1_mul(cp, cp.g1, a), R.g2_mul(cp, q, b)) == T.f12_pow(e, a * b % cp.r)
    f = R.miller_loop(cp, [(cp.g1, q)])
    assert R.final_exp(cp, f) == R.final_exp_naive(cp, f)


@pytest.mark.parametrize("name", ["BN254", "BLS12-381", "BLS12-377"])
def test_committed_golden_vectors_are_what_the_oracle_produces(name):
    cp = R.CURVES[name]
    g = load_golden(name)
    assert int(g["p"], 16) == cp.p and int(g["r"], 16) == cp.r
    for case in g["msm_g1"][:3] + [c for c in g["msm_g1"] if c["name"] in ("unreduced_scalars", "p_and_minus_p")]:
        pts = [R.g1_from_mont_bytes(cp, bytes.fromhex(p)) for p in case["points"]]
        scs = [int(s) for s in case["scalars_int"]]
        assert R.g1_to_mont_bytes(cp, R.g1_msm(cp, pts, scs)).hex() == case["expected"]
        # MultiScalarMul == sum of Mul (math_test.go:336-345)
        acc = None
        for p_, s_ in zip(pts, scs):
            acc = R.g1_add(cp, acc, R.g1_mul(cp, p_, s_))
        assert R.g1_wire_compressed(cp, acc).hex() == case["expected_wire"]
    c = g["pairing"][0]
    P = R.g1_from_mont_bytes(cp, bytes.fromhex(c["g1"]))
    Q = R.g2_from_mont_bytes(cp, bytes.fromhex(c["g2"]))
    assert R.gt_to_mont_bytes(cp, R.pairing(cp, P, Q)).hex() == c["fexp"]
    assert R.gt_wire_bytes(cp, R.pairing(cp, P, Q)).hex() == g["gen_gt_wire"]
