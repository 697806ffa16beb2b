"""oracle/pyref.py -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

Pure-Python big-integer restatement of the hot path of IBM/mathlib: G1/G2 multi-scalar
multiplication and the optimal-ate pairing (Miller loop + final exponentiation) for
BN254, BLS12-381 and BLS12-377.  It is the ground truth the C restatement (oracle/cref)
and the HIP kernels are compared with on small inputs, and it generates tests/golden/.

Reference call sites this follows (all relative to /root/reference):
  * MultiScalarMul  driver/gurvy/bls12381/bls12-381.go:766-783, driver/gurvy/bn254.go:232-245,
                    driver/gurvy/bls12-377.go:229-242, driver/kilic/bls12-381.go:247-254
  * Pairing/Pairing2 (Miller loop only in gurvy) bls12-381.go:448-464, bn254.go:247-263,
                    bls12-377.go:244-260;  FExp bls12-381.go:466-468, bn254.go:265-267, bls12-377.go:262-264
  * scalar normalisation (BaseZr may be negative / >= r): driver/common/big.go:101-113, bn254.go:239
  * Fp Montgomery layout ([k]uint64 little-endian limbs, R = 2^(64k)): driver/kilic/custom.go:24-29,
                    driver/kilic/custom_generic.go:57-175, driver/gurvy/custom.go:24-40

The arithmetic itself lives in third-party Go modules that are NOT in the reference tree
(gnark-crypto v0.20.1, kilic/bls12-381 v0.1.0; go.mod:6,15) and no Go toolchain exists in
this image, so the algorithms are restated from the public curve specifications.

PARITY STATUS: the constants of this file are pinned against every known-answer value the
reference holds for this path (generators math_test.go:250-259, group orders :261-270,
BLS12-381 modulus / -p^-1 mod 2^64 / R mod p / 2^256*R mod p in driver/kilic/custom.go:26,29,329-336
and custom_generic.go:64) -- see tests/test_oracle_pinned.py.  The reference holds NO golden
vector for an MSM, Miller-loop or final-exponentiation *output* (its tests are algebraic
property checks on crypto/rand inputs), so for those outputs: **parity unpinned** by the
reference; they are fixed here by uniqueness (an MSM result is a unique group element with a
canonical affine form; a Gt value after FExp is f^(k(p^12-1)/r), k = 3 for BLS12 and
k = 2x(6x^2+3x+1) for BN254, whatever Miller-loop variant produced f).

Deliberately simple: affine formulas, schoolbook tower multiplication, Miller loop done
generically in Fp12 on the untwisted point, final exponentiation available as a plain pow().
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

# --------------------------------------------------------------------------------------
# Curve parameters
# --------------------------------------------------------------------------------------


def _bls12_p(x: int) -> int:
    return ((x - 1) ** 2 * (x**4 - x**2 + 1)) // 3 + x


def _bls12_r(x: int) -> int:
    return x**4 - x**2 + 1


@dataclass(frozen=True)
class CurveParams:
    name: str
    curve_id: int  # id used by the C ABI (include/mlhip.h)
    x: int  # curve seed (BLS: x, BN: t)
    p: int
    r: int
    b: int  # E: y^2 = x^3 + b
    beta: int  # Fp2 = Fp[u]/(u^2 - beta)
    xi: Tuple[int, int]  # Fp6 = Fp2[v]/(v^3 - xi), Fp12 = Fp6[w]/(w^2 - v)  => w^6 = xi
    twist: str  # 'M': E': y^2 = x^3 + b*xi ; 'D': E': y^2 = x^3 + b/xi
    family: str  # 'BLS12' or 'BN'
    g1: Tuple[int, int]
    g2: Optional[Tuple[Tuple[int, int], Tuple[int, int]]]
    limbs64: int  # number of 64-bit limbs of an Fp element in memory

    @property
    def fp_bytes(self) -> int:
        return self.limbs64 * 8

    @property
    def R(self) -> int:
        """Montgomery radix of the in-memory representation."""
        return 1 << (64 * self.limbs64)

    @property
    def fexp_cofactor(self) -> int:
        """k such that FExp(f) = f^(k*(p^12-1)/r) (gnark FinalExponentiation)."""
        if self.family == "BLS12":
            return 3
        x = self.x
        return 2 * x * (6 * x * x + 3 * x + 1)

    @property
    def ate_loop(self) -> int:
        return abs(self.x) if self.family == "BLS12" else 6 * self.x + 2


_X381 = -0xD201000000010000
_X377 = 0x8508C00000000001
_T254 = 4965661367192848881

BLS12_381 = CurveParams(
    name="BLS12-381",
    curve_id=1,
    x=_X381,
    p=_bls12_p(_X381),
    r=_bls12_r(_X381),
    b=4,
    beta=-1,
    xi=(1, 1),
    twist="M",
    family="BLS12",
    # math_test.go:253 (expectedG1Gens[BLS12_381])
    g1=(
        3685416753713387016781088315183077757961620795782546409894578378688607592378376318836054947676345821548104185464507,
        1339506544944476473020471379941921221584933875938349620426543736416511423956333506472724655353366534992391756441569,
    ),
    # standard BLS12-381 G2 generator (public curve specification; not in the reference tree)
    g2=(
        (
            0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
            0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E,
        ),
        (
            0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
            0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE,
        ),
    ),
    limbs64=6,
)

BN254 = CurveParams(
    name="BN254",
    curve_id=0,
    x=_T254,
    p=36 * _T254**4 + 36 * _T254**3 + 24 * _T254**2 + 6 * _T254 + 1,
    r=36 * _T254**4 + 36 * _T254**3 + 18 * _T254**2 + 6 * _T254 + 1,
    b=3,
    beta=-1,
    xi=(9, 1),
    twist="D",
    family="BN",
    g1=(1, 2),  # math_test.go:251
    # standard alt_bn128 G2 generator (public specification, EIP-197)
    g2=(
        (
            10857046999023057135944570762232829481370756359578518086990519993285655852781,
            11559732032986387107991004021392285783925812861821192530917403151452391805634,
        ),
        (
            8495653923123431417604973247489272438418190587263600148770280649306958101930,
            4082367875863433681332203403145435568316851327593401208105741076214120093531,
        ),
    ),
    limbs64=4,
)

BLS12_377 = CurveParams(
    name="BLS12-377",
    curve_id=2,
    x=_X377,
    p=_bls12_p(_X377),
    r=_bls12_r(_X377),
    b=1,
    beta=-5,
    xi=(0, 1),
    twist="D",
    family="BLS12",
    # math_test.go:254 (expectedG1Gens[BLS12_377_GURVY])
    g1=(
        81937999373150964239938255573465948239988671502647976594219695644855304257327692006745978603320413799295628339695,
        241266749859715473739788878240585681733927191168601896383759122102112907357779751001206799952863815012735208165030,
    ),
    g2=None,  # derived below (own generator of the r-torsion of E'(Fp2); gnark's is not in the tree)
    limbs64=6,
)

CURVES = {c.name: c for c in (BN254, BLS12_381, BLS12_377)}
CURVES_BY_ID = {c.curve_id: c for c in CURVES.values()}

# --------------------------------------------------------------------------------------
# Fp helpers
# --------------------------------------------------------------------------------------


def fp_inv(a: int, p: int) -> int:
    return pow(a, -1, p)


def fp_sqrt(a: int, p: int) -> Optional[int]:
    """Tonelli-Shanks; returns None when a is a non-residue."""
    a %= p
    if a == 0:
        return 0
    if pow(a, (p - 1) // 2, p) != 1:
        return None
    if p % 4 == 3:
        return pow(a, (p + 1) // 4, p)
    q, s = p - 1, 0
    while q % 2 == 0:
        q //= 2
        s += 1
    z = 2
    while pow(z, (p - 1) // 2, p) != p - 1:
        z += 1
    m, c, t, rr = s, pow(z, q, p), pow(a, q, p), pow(a, (q + 1) // 2, p)
    while t != 1:
        i, t2 = 0, t
        while t2 != 1:
            t2 = t2 * t2 % p
            i += 1
        bb = pow(c, 1 << (m - i - 1), p)
        m, c = i, bb * bb % p
        t, rr = t * c % p, rr * bb % p
    return rr


# --------------------------------------------------------------------------------------
# Tower fields.  An Fp2 element is a tuple (a0, a1) = a0 + a1*u.  An Fp12 element is a
# tuple of 6 Fp2 coefficients over the basis 1, w, ..., w^5 with w^6 = xi.
# (tower view: c0 = g0 + g2 v + g4 v^2, c1 = g1 + g3 v + g5 v^2, f = c0 + c1 w, v = w^2)
# --------------------------------------------------------------------------------------


class Tower:
    def __init__(self, cp: CurveParams):
        self.cp = cp
        self.p = cp.p
        self.beta = cp.beta % cp.p
        self.xi = (cp.xi[0] % cp.p, cp.xi[1] % cp.p)
        self.f2_zero = (0, 0)
        self.f2_one = (1, 0)
        self.f12_one = (self.f2_one,) + (self.f2_zero,) * 5
        # Frobenius constants gamma[k][i] = xi^(i*(p^k-1)/6), k = 1..3 (computed, not tabulated)
        self.gamma = {}
        for k in (1, 2, 3):
            e = (self.p**k - 1) // 6
            g1 = self.f2_pow(self.xi, e)
            gs = [self.f2_one]
            for _ in range(5):
                gs.append(self.f2_mul(gs[-1], g1))
            self.gamma[k] = gs

    # ---- Fp2 ----
    def f2(self, a0: int, a1: int = 0):
        return (a0 % self.p, a1 % self.p)

    def f2_add(self, a, b):
        return ((a[0] + b[0]) % self.p, (a[1] + b[1]) % self.p)

    def f2_sub(self, a, b):
        return ((a[0] - b[0]) % self.p, (a[1] - b[1]) % self.p)

    def f2_neg(self, a):
        return ((-a[0]) % self.p, (-a[1]) % self.p)

    def f2_mul(self, a, b):
        p = self.p
        return ((a[0] * b[0] + self.beta * a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)

    def f2_sqr(self, a):
        return self.f2_mul(a, a)

    def f2_muls(self, a, s: int):
        return (a[0] * s % self.p, a[1] * s % self.p)

    def f2_conj(self, a):
        return (a[0], (-a[1]) % self.p)

    def f2_norm(self, a):
        return (a[0] * a[0] - self.beta * a[1] * a[1]) % self.p

    def f2_inv(self, a):
        n = fp_inv(self.f2_norm(a), self.p)
        return (a[0] * n % self.p, (-a[1]) * n % self.p)

    def f2_pow(self, a, e: int):
        r = self.f2_one
        for bit in bin(e)[2:]:
            r = self.f2_mul(r, r)
            if bit == "1":
                r = self.f2_mul(r, a)
        return r

    def f2_is_zero(self, a):
        return a[0] == 0 and a[1] == 0

    def f2_sqrt(self, a):
        """Square root in Fp2 via the norm (None if a is a non-residue)."""
        p = self.p
        if self.f2_is_zero(a):
            return self.f2_zero
        if a[1] == 0:
            s = fp_sqrt(a[0], p)
            if s is not None:
                return (s, 0)
            # a0 is a non-residue in Fp: sqrt is purely imaginary: (y u)^2 = y^2 beta = a0
            s = fp_sqrt(a[0] * fp_inv(self.beta, p) % p, p)
            return None if s is None else (0, s)
        n = fp_sqrt(self.f2_norm(a), p)
        if n is None:
            return None
        inv2 = fp_inv(2, p)
        for nn in (n, (-n) % p):
            x0sq = (a[0] + nn) * inv2 % p
            x0 = fp_sqrt(x0sq, p)
            if x0 is None or x0 == 0:
                continue
            x1 = a[1] * fp_inv(2 * x0 % p, p) % p
            cand = (x0, x1)
            if self.f2_mul(cand, cand) == (a[0] % p, a[1] % p):
                return cand
        return None

    # ---- Fp12 (w-basis) ----
    def f12_from_f2(self, g, i: int = 0):
        out = [self.f2_zero] * 6
        out[i] = g
        return tuple(out)

    def f12_add(self, a, b):
        return tuple(self.f2_add(x, y) for x, y in zip(a, b))

    def f12_sub(self, a, b):
        return tuple(self.f2_sub(x, y) for x, y in zip(a, b))

    def f12_neg(self, a):
        return tuple(self.f2_neg(x) for x in a)

    def f12_mul(self, a, b):
        acc = [self.f2_zero] * 11
        for i in range(6):
            if self.f2_is_zero(a[i]):
                continue
            for j in range(6):
                acc[i + j] = self.f2_add(acc[i + j], self.f2_mul(a[i], b[j]))
        out = list(acc[:6])
        for k in range(6, 11):
            out[k - 6] = self.f2_add(out[k - 6], self.f2_mul(acc[k], self.xi))
        return tuple(out)

    def f12_sqr(self, a):
        return self.f12_mul(a, a)

    def f12_conj(self, a):
        """f^(p^6): negates the odd powers of w."""
        return tuple(self.f2_neg(x) if i % 2 else x for i, x in enumerate(a))

    def f12_frob(self, a, k: int = 1):
        """f^(p^k), k in 1..3 (coefficient-wise conjugation and gamma twist)."""
        g = self.gamma[k]
        out = []
        for i, x in enumerate(a):
            xc = self.f2_conj(x) if k % 2 else x
            out.append(self.f2_mul(xc, g[i]))
        return tuple(out)

    # Fp6 helpers on (b0,b1,b2), v^3 = xi, used only for inversion
    def _f6_mul(self, a, b):
        m = self.f2_mul
        ad = self.f2_add
        xi = self.xi
        t0 = ad(m(a[0], b[0]), m(xi, ad(m(a[1], b[2]), m(a[2], b[1]))))
        t1 = ad(ad(m(a[0], b[1]), m(a[1], b[0])), m(xi, m(a[2], b[2])))
        t2 = ad(ad(m(a[0], b[2]), m(a[1], b[1])), m(a[2], b[0]))
        return (t0, t1, t2)

    def _f6_inv(self, a):
        m, sub, xi = self.f2_mul, self.f2_sub, self.xi
        c0 = sub(m(a[0], a[0]), m(xi, m(a[1], a[2])))
        c1 = sub(m(xi, m(a[2], a[2])), m(a[0], a[1]))
        c2 = sub(m(a[1], a[1]), m(a[0], a[2]))
        t = self.f2_add(m(a[0], c0), m(xi, self.f2_add(m(a[2], c1), m(a[1], c2))))
        ti = self.f2_inv(t)
        return (m(c0, ti), m(c1, ti), m(c2, ti))

    def f12_inv(self, a):
        A = (a[0], a[2], a[4])
        B = (a[1], a[3], a[5])
        # (A + Bw)^-1 = (A - Bw) / (A^2 - v B^2)
        A2 = self._f6_mul(A, A)
        B2 = self._f6_mul(B, B)
        vB2 = (self.f2_mul(self.xi, B2[2]), B2[0], B2[1])
        D = tuple(self.f2_sub(x, y) for x, y in zip(A2, vB2))
        Di = self._f6_inv(D)
        RA = self._f6_mul(A, Di)
        RB = self._f6_mul(B, Di)
        RB = tuple(self.f2_neg(x) for x in RB)
        return (RA[0], RB[0], RA[1], RB[1], RA[2], RB[2])

    def f12_pow(self, a, e: int):
        if e < 0:
            return self.f12_pow(self.f12_inv(a), -e)
        r = self.f12_one
        for bit in bin(e)[2:]:
            r = self.f12_sqr(r)
            if bit == "1":
                r = self.f12_mul(r, a)
        return r

    def f12_is_one(self, a):
        return a == self.f12_one

    # conversion to/from the gnark struct order E12{C0,C1 E6{B0,B1,B2 E2{A0,A1}}}
    def f12_to_tower(self, a):
        """-> list of 12 ints in memory order C0.B0.A0, C0.B0.A1, C0.B1.A0, ... C1.B2.A1."""
        order = [0, 2, 4, 1, 3, 5]
        out = []
        for i in order:
            out += [a[i][0], a[i][1]]
        return out

    def f12_from_tower(self, c: Sequence[int]):
        order = [0, 2, 4, 1, 3, 5]
        out = [None] * 6
        for k, i in enumerate(order):
            out[i] = (c[2 * k] % self.p, c[2 * k + 1] % self.p)
        return tuple(out)


_TOWERS = {}


def tower(cp: CurveParams) -> Tower:
    if cp.name not in _TOWERS:
        _TOWERS[cp.name] = Tower(cp)
    return _TOWERS[cp.name]


# --------------------------------------------------------------------------------------
# Elliptic curves.  Points are None (infinity) or affine tuples.
# --------------------------------------------------------------------------------------

# ---- G1: E(Fp), coordinates are ints ----


def g1_is_on_curve(cp: CurveParams, P) -> bool:
    if P is None:
        return True
    x, y = P
    return (y * y - x * x * x - cp.b) % cp.p == 0


def g1_neg(cp, P):
    return None if P is None else (P[0], (-P[1]) % cp.p)


def g1_add(cp: CurveParams, P, Q):
    p = cp.p
    if P is None:
        return Q
    if Q is None:
        return P
    if P[0] == Q[0]:
        if (P[1] + Q[1]) % p == 0:
            return None
        lam = 3 * P[0] * P[0] * fp_inv(2 * P[1] % p, p) % p
    else:
        lam = (Q[1] - P[1]) * fp_inv((Q[0] - P[0]) % p, p) % p
    x3 = (lam * lam - P[0] - Q[0]) % p
    return (x3, (lam * (P[0] - x3) - P[1]) % p)


def _jac_dbl(p, X, Y, Z):
    if Z == 0 or Y == 0:
        return (1, 1, 0)
    A = X * X % p
    B = Y * Y % p
    C = B * B % p
    D = 2 * ((X + B) ** 2 - A - C) % p
    E = 3 * A % p
    X3 = (E * E - 2 * D) % p
    Y3 = (E * (D - X3) - 8 * C) % p
    Z3 = 2 * Y * Z % p
    return (X3, Y3, Z3)


def _jac_add_affine(p, X1, Y1, Z1, x2, y2):
    if Z1 == 0:
        return (x2, y2, 1)
    Z1Z1 = Z1 * Z1 % p
    U2 = x2 * Z1Z1 % p
    S2 = y2 * Z1 * Z1Z1 % p
    if U2 == X1:
        if S2 == Y1:
            return _jac_dbl(p, X1, Y1, Z1)
        return (1, 1, 0)
    H = (U2 - X1) % p
    Rr = (S2 - Y1) % p
    HH = H * H % p
    HHH = H * HH % p
    V = X1 * HH % p
    X3 = (Rr * Rr - HHH - 2 * V) % p
    Y3 = (Rr * (V - X3) - Y1 * HHH) % p
    Z3 = Z1 * H % p
    return (X3, Y3, Z3)


# ---- twisted Edwards model of a j = 0 curve with a point of order two (BLS12-377's G1: y^2 = x^3 + 1) -------------------
# Not in the reference: gnark's MultiExp stays on the Weierstrass curve.  The kernels may run the bucket accumulation of a
# BLS12-377 G1 MSM in extended twisted Edwards coordinates with a = -1 when the caller vouches for the prime-order
# subgroup (the addition law is complete there; d is a square, so it is not complete on the whole curve).  These
# functions are the test-side statement of the birational map:
#   Weierstrass (x, y), alpha = -1 the root of x^3 + 1, s = 1/sqrt(3):  Montgomery u = s (x + 1), v = s y,
#   twisted Edwards x_E = u / v, y_E = (u - 1)/(u + 1) on a_E x^2 + y^2 = 1 + d_E x^2 y^2, a_E = (A + 2)/B,
#   d_E = (A - 2)/B with A = -3 s, B = s; scaling x' = f x_E with f = sqrt(-a_E) gives -x'^2 + y^2 = 1 + d x'^2 y^2,
#   d = -d_E / a_E.
_ED_CACHE: dict = {}


def edwards_params(cp: CurveParams):
    """(s, f, d) of the a = -1 twisted Edwards model; the smaller square roots are taken so that the constants are fixed"""
    if cp.name in _ED_CACHE:
        return _ED_CACHE[cp.name]
    p = cp.p
    assert cp.b == 1 and cp.name == "BLS12-377"
    r3 = fp_sqrt(3, p)
    assert r3 is not None
    r3 = min(r3, p - r3)
    s = pow(r3, -1, p)
    A, B = (-3 * s) % p, s
    a_e = (A + 2) * pow(B, -1, p) % p
    d_e = (A - 2) * pow(B, -1, p) % p
    f = fp_sqrt((-a_e) % p, p)
    assert f is not None
    f = min(f, p - f)
    d = (-d_e) * pow(a_e, -1, p) % p
    _ED_CACHE[cp.name] = (s, f, d)
    return s, f, d


def g1_to_edwards(cp: CurveParams, P):
    """affine Weierstrass point (None = infinity) of ODD order -> affine (x', y) on -x^2 + y^2 = 1 + d x^2 y^2"""
    s, f, d = edwards_params(cp)
    p = cp.p
    if P is None:
        return (0, 1)
    x, y = P
    u = s * (x + 1) % p
    xe = f * (x + 1) % p * pow(y, -1, p) % p
    ye = (u - 1) * pow(u + 1, -1, p) % p
    assert (-xe * xe + ye * ye - 1 - d * xe * xe % p * ye * ye) % p == 0
    return (xe, ye)


def edwards_to_g1(cp: CurveParams, E):
    s, f, d = edwards_params(cp)
    p = cp.p
    xe, ye = E
    if xe == 0 and ye == 1:
        return None
    u = (1 + ye) * pow(1 - ye, -1, p) % p
    v = f * u % p * pow(xe, -1, p) % p
    x = (u * pow(s, -1, p) - 1) % p
    y = v * pow(s, -1, p) % p
    assert (y * y - x * x * x - cp.b) % p == 0
    return (x, y)


def edwards_add(cp: CurveParams, E1, E2):
    """the unified addition law of -x^2 + y^2 = 1 + d x^2 y^2 (affine)"""
    _, _, d = edwards_params(cp)
    p = cp.p
    x1, y1 = E1
    x2, y2 = E2
    t = d * x1 % p * x2 % p * y1 % p * y2 % p
    x3 = (x1 * y2 + y1 * x2) * pow(1 + t, -1, p) % p
    y3 = (y1 * y2 + x1 * x2) * pow(1 - t, -1, p) % p
    return (x3, y3)


def g1_mul(cp: CurveParams, P, k: int):
    """[k]P with k reduced mod r first (scalars may be negative or >= r: driver/common/big.go:101-113)."""
    p = cp.p
    k %= cp.r
    if P is None or k == 0:
        return None
    X, Y, Z = 1, 1, 0
    for bit in bin(k)[2:]:
        X, Y, Z = _jac_dbl(p, X, Y, Z)
        if bit == "1":
            X, Y, Z = _jac_add_affine(p, X, Y, Z, P[0], P[1])
    if Z == 0:
        return None
    zi = fp_inv(Z, p)
    zi2 = zi * zi % p
    return (X * zi2 % p, Y * zi2 * zi % p)


def g1_mul_unreduced(cp: CurveParams, P, k: int):
    """[k]P for a non-negative integer k WITHOUT reducing mod r (cofactor clearing, order checks)."""
    p = cp.p
    if P is None or k == 0:
        return None
    X, Y, Z = 1, 1, 0
    for bit in bin(k)[2:]:
        X, Y, Z = _jac_dbl(p, X, Y, Z)
        if bit == "1":
            X, Y, Z = _jac_add_affine(p, X, Y, Z, P[0], P[1])
    if Z == 0:
        return None
    zi = fp_inv(Z, p)
    zi2 = zi * zi % p
    return (X * zi2 % p, Y * zi2 * zi % p)


def g1_msm(cp: CurveParams, points: Sequence, scalars: Sequence[int]):
    """Sum_i [s_i]P_i, the naive definition (kilic's loop: driver/kilic/bls12-381.go:247-254).
    gnark's MultiExp returns an error (dropped by the driver -> identity) on a length mismatch
    (bls12-381.go:777); that rule is applied by the callers, not here."""
    acc = None
    for P, s in zip(points, scalars):
        acc = g1_add(cp, acc, g1_mul(cp, P, s))
    return acc


# ---- G2: E'(Fp2), coordinates are Fp2 tuples ----


def twist_b(cp: CurveParams):
    T = tower(cp)
    b = T.f2(cp.b, 0)
    if cp.twist == "M":
        return T.f2_mul(b, T.xi)
    return T.f2_mul(b, T.f2_inv(T.xi))


def g2_is_on_curve(cp: CurveParams, Q) -> bool:
    if Q is None:
        return True
    T = tower(cp)
    x, y = Q
    lhs = T.f2_sqr(y)
    rhs = T.f2_add(T.f2_mul(T.f2_sqr(x), x), twist_b(cp))
    return lhs == rhs


def g2_neg(cp, Q):
    return None if Q is None else (Q[0], tower(cp).f2_neg(Q[1]))


def g2_add(cp: CurveParams, P, Q):
    T = tower(cp)
    if P is None:
        return Q
    if Q is None:
        return P
    if P[0] == Q[0]:
        if T.f2_is_zero(T.f2_add(P[1], Q[1])):
            return None
        lam = T.f2_mul(T.f2_muls(T.f2_sqr(P[0]), 3), T.f2_inv(T.f2_muls(P[1], 2)))
    else:
        lam = T.f2_mul(T.f2_sub(Q[1], P[1]), T.f2_inv(T.f2_sub(Q[0], P[0])))
    x3 = T.f2_sub(T.f2_sub(T.f2_sqr(lam), P[0]), Q[0])
    y3 = T.f2_sub(T.f2_mul(lam, T.f2_sub(P[0], x3)), P[1])
    return (x3, y3)


def g2_mul_unreduced(cp: CurveParams, Q, k: int):
    if Q is None or k == 0:
        return None
    acc = None
    for bit in bin(k)[2:]:
        acc = g2_add(cp, acc, acc)
        if bit == "1":
            acc = g2_add(cp, acc, Q)
    return acc


def g2_mul(cp: CurveParams, Q, k: int):
    return g2_mul_unreduced(cp, Q, k % cp.r)


def g2_msm(cp: CurveParams, points: Sequence, scalars: Sequence[int]):
    acc = None
    for P, s in zip(points, scalars):
        acc = g2_add(cp, acc, g2_mul(cp, P, s))
    return acc


def g2_order(cp: CurveParams) -> int:
    """#E'(Fp2) of the sextic twist that carries the r-torsion (computed from the CM equation)."""
    from math import isqrt

    p = cp.p
    t = cp.x + 1 if cp.family == "BLS12" else 6 * cp.x * cp.x + 1
    t2 = t * t - 2 * p
    f2sq, rem = divmod(4 * p * p - t2 * t2, 3)
    assert rem == 0
    f2 = isqrt(f2sq)
    assert f2 * f2 == f2sq
    for n in (p * p + 1 - (t2 + 3 * f2) // 2, p * p + 1 - (t2 - 3 * f2) // 2):
        if n % cp.r == 0:
            # both candidate orders may be divisible by r only for one of them in practice
            Q = _g2_some_point(cp)
            if g2_mul_unreduced(cp, Q, n) is None:
                return n
    raise AssertionError("no sextic twist order found")


def _g2_some_point(cp: CurveParams, start: int = 1):
    T = tower(cp)
    bt = twist_b(cp)
    k = start
    while True:
        x = T.f2(k, 1)
        y = T.f2_sqrt(T.f2_add(T.f2_mul(T.f2_sqr(x), x), bt))
        if y is not None:
            return (x, y)
        k += 1


_G2_GEN_CACHE = {}


def g2_generator(cp: CurveParams):
    """The curve's G2 generator.  For BLS12-381 / BN254 the public standard generator; for
    BLS12-377 a generator derived here by cofactor clearing (any generator of the r-torsion
    serves the parity tests: the C ABI takes points, it has no notion of 'the' generator)."""
    if cp.g2 is not None:
        T = tower(cp)
        return (T.f2(*cp.g2[0]), T.f2(*cp.g2[1]))
    if cp.name not in _G2_GEN_CACHE:
        n = g2_order(cp)
        h = n // cp.r
        k = 1
        while True:
            Q = g2_mul_unreduced(cp, _g2_some_point(cp, k), h)
            if Q is not None:
                break
            k += 1
        assert g2_mul_unreduced(cp, Q, cp.r) is None
        _G2_GEN_CACHE[cp.name] = Q
    return _G2_GEN_CACHE[cp.name]


# --------------------------------------------------------------------------------------
# Pairing: optimal ate, done generically in Fp12 on the untwisted G2 point.
# --------------------------------------------------------------------------------------


def untwist(cp: CurveParams, Q):
    """E'(Fp2) -> E(Fp12).  D-twist: (x w^2, y w^3).  M-twist: (x / w^2, y / w^3)."""
    T = tower(cp)
    x, y = Q
    if cp.twist == "D":
        return (T.f12_from_f2(x, 2), T.f12_from_f2(y, 3))
    xii = T.f2_inv(T.xi)
    # w^-2 = w^4 / xi ; w^-3 = w^3 / xi
    return (T.f12_from_f2(T.f2_mul(x, xii), 4), T.f12_from_f2(T.f2_mul(y, xii), 3))


def _e12_add(T: Tower, A, B):
    """affine addition on E(Fp12) (a = 0); returns (sum, slope)."""
    if A[0] == B[0]:
        if A[1] == B[1]:
            three = T.f12_from_f2(T.f2(3))
            two = T.f12_from_f2(T.f2(2))
            lam = T.f12_mul(T.f12_mul(three, T.f12_sqr(A[0])), T.f12_inv(T.f12_mul(two, A[1])))
        else:
            return None, None
    else:
        lam = T.f12_mul(T.f12_sub(B[1], A[1]), T.f12_inv(T.f12_sub(B[0], A[0])))
    x3 = T.f12_sub(T.f12_sub(T.f12_sqr(lam), A[0]), B[0])
    y3 = T.f12_sub(T.f12_mul(lam, T.f12_sub(A[0], x3)), A[1])
    return (x3, y3), lam


def _line(T: Tower, A, lam, Pe):
    """l(P) = (yP - yA) - lam (xP - xA)."""
    return T.f12_sub(T.f12_sub(Pe[1], A[1]), T.f12_mul(lam, T.f12_sub(Pe[0], A[0])))


def _e12_frob(T: Tower, A, k: int):
    return (T.f12_frob(A[0], k), T.f12_frob(A[1], k))


def miller_loop(cp: CurveParams, pairs: Sequence[Tuple]) -> tuple:
    """prod_i f_{loop,Q_i}(P_i): pairs = [(P in G1, Q in G2), ...]; pairs holding an infinity are
    skipped (gnark MillerLoop).  The value is only defined up to factors killed by FExp; compare
    Miller-loop outputs across implementations only after final_exp()."""
    T = tower(cp)
    f = T.f12_one
    for P, Q in pairs:
        if P is None or Q is None:
            continue
        Pe = (T.f12_from_f2(T.f2(P[0])), T.f12_from_f2(T.f2(P[1])))
        Qe = untwist(cp, Q)
        loop = cp.ate_loop
        acc = Qe
        g = T.f12_one
        for bit in bin(loop)[3:]:
            nxt, lam = _e12_add(T, acc, acc)
            g = T.f12_mul(T.f12_sqr(g), _line(T, acc, lam, Pe))
            acc = nxt
            if bit == "1":
                nxt, lam = _e12_add(T, acc, Qe)
                g = T.f12_mul(g, _line(T, acc, lam, Pe))
                acc = nxt
        if cp.family == "BLS12":
            if cp.x < 0:
                g = T.f12_conj(g)
        else:
            # BN: two extra lines through pi(Q) and -pi^2(Q)
            Q1 = _e12_frob(T, Qe, 1)
            Q2 = _e12_frob(T, Qe, 2)
            Q2 = (Q2[0], T.f12_neg(Q2[1]))
            nxt, lam = _e12_add(T, acc, Q1)
            g = T.f12_mul(g, _line(T, acc, lam, Pe))
            acc = nxt
            nxt, lam = _e12_add(T, acc, Q2)
            if lam is not None:
                g = T.f12_mul(g, _line(T, acc, lam, Pe))
        f = T.f12_mul(f, g)
    return f


def final_exp_naive(cp: CurveParams, f):
    """f^(k (p^12 - 1)/r) by one plain exponentiation (slow; the definition)."""
    T = tower(cp)
    e = cp.fexp_cofactor * ((cp.p**12 - 1) // cp.r)
    return T.f12_pow(f, e)


def final_exp(cp: CurveParams, f):
    """Same value as final_exp_naive: easy part by conjugation/Frobenius, hard part by pow()."""
    T = tower(cp)
    p = cp.p
    t = T.f12_mul(T.f12_conj(f), T.f12_inv(f))  # f^(p^6-1)
    t = T.f12_mul(T.f12_frob(t, 2), t)  # ^(p^2+1)
    hard = cp.fexp_cofactor * ((p**4 - p * p + 1) // cp.r)
    return T.f12_pow(t, hard)


def pairing(cp: CurveParams, P, Q):
    """FExp(Pairing(Q, P)) in the reference's terms (math_test.go:430-434)."""
    return final_exp(cp, miller_loop(cp, [(P, Q)]))


# --------------------------------------------------------------------------------------
# In-memory (Montgomery, little-endian limbs) layout used at the C ABI, and gnark wire bytes
# --------------------------------------------------------------------------------------


def fp_to_mont_bytes(cp: CurveParams, a: int) -> bytes:
    return ((a % cp.p) * cp.R % cp.p).to_bytes(cp.fp_bytes, "little")


def fp_from_mont_bytes(cp: CurveParams, b: bytes) -> int:
    v = int.from_bytes(b, "little")
    return v * fp_inv(cp.R, cp.p) % cp.p


def g1_to_mont_bytes(cp: CurveParams, P) -> bytes:
    """gnark G1Affine{X,Y fp.Element}; infinity is (0,0)."""
    if P is None:
        return bytes(2 * cp.fp_bytes)
    return fp_to_mont_bytes(cp, P[0]) + fp_to_mont_bytes(cp, P[1])


def g1_from_mont_bytes(cp: CurveParams, b: bytes):
    n = cp.fp_bytes
    x = fp_from_mont_bytes(cp, b[:n])
    y = fp_from_mont_bytes(cp, b[n : 2 * n])
    return None if (x == 0 and y == 0) else (x, y)


def g2_to_mont_bytes(cp: CurveParams, Q) -> bytes:
    """gnark G2Affine{X,Y E2{A0,A1}}; infinity is all-zero."""
    if Q is None:
        return bytes(4 * cp.fp_bytes)
    return b"".join(fp_to_mont_bytes(cp, c) for c in (Q[0][0], Q[0][1], Q[1][0], Q[1][1]))


def g2_from_mont_bytes(cp: CurveParams, b: bytes):
    n = cp.fp_bytes
    c = [fp_from_mont_bytes(cp, b[i * n : (i + 1) * n]) for i in range(4)]
    if all(v == 0 for v in c):
        return None
    return ((c[0], c[1]), (c[2], c[3]))


def gt_to_mont_bytes(cp: CurveParams, f) -> bytes:
    return b"".join(fp_to_mont_bytes(cp, c) for c in tower(cp).f12_to_tower(f))


def gt_from_mont_bytes(cp: CurveParams, b: bytes):
    n = cp.fp_bytes
    return tower(cp).f12_from_tower([fp_from_mont_bytes(cp, b[i * n : (i + 1) * n]) for i in range(12)])


def scalar_to_bytes(s: int, cp: CurveParams, mont: bool = False) -> bytes:
    """32-byte little-endian scalar (fr.Element memory layout when mont=True, R = 2^256)."""
    s %= cp.r
    if mont:
        s = s * (1 << 256) % cp.r
    return s.to_bytes(32, "little")


def gt_wire_bytes(cp: CurveParams, f) -> bytes:
    """gnark GT.Bytes(): 12 big-endian Fp, order C1.B2.A1, C1.B2.A0, ..., C0.B0.A0."""
    c = tower(cp).f12_to_tower(f)
    return b"".join(v.to_bytes(cp.fp_bytes, "big") for v in reversed(c))


def g1_wire_uncompressed(cp: CurveParams, P) -> bytes:
    """gnark G1Affine.RawBytes() (zcash flags for the BLS12 curves, 2-bit header for BN254)."""
    n = cp.fp_bytes
    if P is None:
        out = bytearray(2 * n)
        out[0] |= 0x40
        return bytes(out)
    return P[0].to_bytes(n, "big") + P[1].to_bytes(n, "big")


def g1_wire_compressed(cp: CurveParams, P) -> bytes:
    """gnark G1Affine.Bytes(): compressed form; the sign bit is 'y lexicographically largest'."""
    n = cp.fp_bytes
    if cp.family == "BLS12":
        if P is None:
            out = bytearray(n)
            out[0] = 0xC0
            return bytes(out)
        out = bytearray(P[0].to_bytes(n, "big"))
        out[0] |= 0x80
        if P[1] > (cp.p - 1) // 2:
            out[0] |= 0x20
        return bytes(out)
    # BN254: mCompressedSmallest = 0b10 << 6, mCompressedLargest = 0b11 << 6, infinity = 0b01 << 6
    if P is None:
        out = bytearray(n)
        out[0] = 0x40
        return bytes(out)
    out = bytearray(P[0].to_bytes(n, "big"))
    out[0] |= 0xC0 if P[1] > (cp.p - 1) // 2 else 0x80
    return bytes(out)


def g1_from_wire(cp: CurveParams, b: bytes, subgroup_check: bool = True):
    """gnark G1Affine.SetBytes semantics (the reference's NewG1FromBytes / NewG1FromCompressed,
    driver/gurvy/bls12381/bls12-381.go:531-569): returns (point, status) with status 0 ok, 1 malformed
    encoding, 2 not on the curve (or x^3 + b is a non-residue), 3 not in the r-torsion subgroup.
    The format (compressed / uncompressed) follows from the flag bits; len(b) must match it."""
    n = cp.fp_bytes
    if len(b) not in (n, 2 * n):
        return None, 1
    flags = b[0]
    if cp.family == "BLS12":
        hdr, mask = flags & 0xE0, 0x1F
        compressed = bool(hdr & 0x80)
        infinity = bool(hdr & 0x40)
        largest = bool(hdr & 0x20)
        if not compressed and largest:
            return None, 1
    else:
        hdr, mask = flags & 0xC0, 0x3F
        compressed = hdr in (0x80, 0xC0) or (hdr == 0x40 and len(b) == n)
        infinity = hdr == 0x40
        largest = hdr == 0xC0
    if len(b) != (n if compressed else 2 * n):
        return None, 1
    body = bytes([b[0] & mask]) + b[1:]
    if infinity:
        if cp.family == "BLS12" and largest:
            return None, 1
        return (None, 0) if not any(body) else (None, 1)
    x = int.from_bytes(body[:n], "big")
    if x >= cp.p:
        return None, 1
    if compressed:
        y = fp_sqrt((x * x * x + cp.b) % cp.p, cp.p)
        if y is None:
            return None, 2
        if (y > (cp.p - 1) // 2) != largest:
            y = (-y) % cp.p
    else:
        y = int.from_bytes(body[n:], "big")
        if y >= cp.p:
            return None, 1
        if (y * y - x * x * x - cp.b) % cp.p:
            return None, 2
    P = (x, y)
    if subgroup_check and g1_mul_unreduced(cp, P, cp.r) is not None:
        return None, 3
    return P, 0


def _wire_header(cp: CurveParams, flags: int, nbytes: int, short: int):
    """(compressed, infinity, largest, mask) from byte 0, or None when the combination is malformed."""
    if cp.family == "BLS12":
        hdr, mask = flags & 0xE0, 0x1F
        compressed, infinity, largest = bool(hdr & 0x80), bool(hdr & 0x40), bool(hdr & 0x20)
        if (not compressed and largest) or (infinity and largest):
            return None
    else:
        hdr, mask = flags & 0xC0, 0x3F
        compressed = hdr in (0x80, 0xC0) or (hdr == 0x40 and nbytes == short)
        infinity, largest = hdr == 0x40, hdr == 0xC0
    return compressed, infinity, largest, mask


def f2_is_largest(cp: CurveParams, a) -> bool:
    """gnark E2.LexicographicallyLargest: decided by A1 unless it is zero, then by A0."""
    half = (cp.p - 1) // 2
    return a[1] > half if a[1] else a[0] > half


def g2_wire_uncompressed(cp: CurveParams, Q) -> bytes:
    """gnark G2Affine.RawBytes(): X.A1 | X.A0 | Y.A1 | Y.A0 big-endian."""
    n = cp.fp_bytes
    if Q is None:
        out = bytearray(4 * n)
        out[0] |= 0x40
        return bytes(out)
    (x0, x1), (y0, y1) = Q
    return b"".join(v.to_bytes(n, "big") for v in (x1, x0, y1, y0))


def g2_wire_compressed(cp: CurveParams, Q) -> bytes:
    """gnark G2Affine.Bytes(): X.A1 | X.A0 with the header in byte 0."""
    n = cp.fp_bytes
    if Q is None:
        out = bytearray(2 * n)
        out[0] = 0xC0 if cp.family == "BLS12" else 0x40
        return bytes(out)
    (x0, x1), y = Q
    out = bytearray(x1.to_bytes(n, "big") + x0.to_bytes(n, "big"))
    big = f2_is_largest(cp, y)
    if cp.family == "BLS12":
        out[0] |= 0x80 | (0x20 if big else 0)
    else:
        out[0] |= 0xC0 if big else 0x80
    return bytes(out)


def g2_from_wire(cp: CurveParams, b: bytes, subgroup_check: bool = True):
    """gnark G2Affine.SetBytes semantics (NewG2FromBytes / NewG2FromCompressed,
    driver/gurvy/bls12381/bls12-381.go:541-569); status codes as g1_from_wire."""
    n = cp.fp_bytes
    T = tower(cp)
    if len(b) not in (2 * n, 4 * n):
        return None, 1
    h = _wire_header(cp, b[0], len(b), 2 * n)
    if h is None:
        return None, 1
    compressed, infinity, largest, mask = h
    if len(b) != (2 * n if compressed else 4 * n):
        return None, 1
    body = bytes([b[0] & mask]) + b[1:]
    if infinity:
        return (None, 0) if not any(body) else (None, 1)
    vals = [int.from_bytes(body[i : i + n], "big") for i in range(0, len(body), n)]
    if any(v >= cp.p for v in vals):
        return None, 1
    x = (vals[1], vals[0])
    rhs = T.f2_add(T.f2_mul(T.f2_sqr(x), x), twist_b(cp))
    if compressed:
        y = T.f2_sqrt(rhs)
        if y is None:
            return None, 2
        if f2_is_largest(cp, y) != largest:
            y = T.f2_neg(y)
    else:
        y = (vals[3], vals[2])
        if T.f2_sqr(y) != rhs:
            return None, 2
    Q = (x, y)
    if subgroup_check and g2_mul_unreduced(cp, Q, cp.r) is not None:
        return None, 3
    return Q, 0


# --------------------------------------------------------------------------------------
# Deterministic inputs (SHA-256 counter DRBG, seed "mlhip-vec-1")
# --------------------------------------------------------------------------------------


class Drbg:
    def __init__(self, stream: str, seed: str = "mlhip-vec-1"):
        self.key = (seed + "/" + stream).encode()
        self.ctr = 0

    def block(self) -> bytes:
        h = hashlib.sha256(self.key + self.ctr.to_bytes(8, "big")).digest()
        self.ctr += 1
        return h

    def below(self, n: int) -> int:
        """uniform-ish integer in [0, n): 64 spare bits make the bias negligible."""
        nb = (n.bit_length() + 64 + 7) // 8
        buf = b""
        while len(buf) < nb:
            buf += self.block()
        return int.from_bytes(buf[:nb], "big") % n


def random_g1(cp: CurveParams, d: Drbg):
    return g1_mul(cp, cp.g1, 1 + d.below(cp.r - 1))


def random_g2(cp: CurveParams, d: Drbg):
    return g2_mul(cp, g2_generator(cp), 1 + d.below(cp.r - 1))
