"""Host-side mirror of the reference's plugin boundary for the hot path (driver/math.go:49-360).

The reference is Go; this image has no Go toolchain, so the binding a maintainer would add
(`driver/hip`, shown in INTEGRATION.md and shipped as source under go/driver/hip) cannot run
here.  This module restates the same interface in Python over the same C ABI -- same method
names, argument order and error behaviour -- so the parity tests read like math_test.go:

    Curve.MultiScalarMul(a []G1, b []Zr) G1      driver/math.go:169-170   -> mlhip_msm_g1
    Curve.Pairing(G2, G1) Gt                     driver/math.go:50-52     -> mlhip_miller_loop (ppp=1)
    Curve.Pairing2(p2a, p2b G2, p1a, p1b G1) Gt  driver/math.go:54-55     -> mlhip_miller_loop (ppp=2)
    Curve.FExp(Gt) Gt                            driver/math.go:56-57     -> mlhip_final_exp
    G1.Mul / G2.Mul / Gt.Mul / G1.Add ...        driver/math.go:249-360   -> n=1 MSM, group helpers
    Gt.Exp(Zr) Gt                                driver/math.go:358-359   -> mlhip_gt_exp
  additive: MultiScalarMulG2, PairingBatch, PairingProduct.

Only plumbing happens here (byte packing, Montgomery <-> integer conversion for printing and wire
bytes).  All group / field arithmetic is done by libmlhip.so; nothing under oracle/ is imported.
Like the reference's drivers, failures raise (the Go drivers panic: driver/gurvy/bn254.go:249-251).
"""
from __future__ import annotations

import ctypes
import hashlib
import secrets
from typing import List, Sequence

from . import _lib
from ._lib import CURVE_BLS12_377, CURVE_BLS12_381, CURVE_BN254, GROUP_G1, GROUP_G2, check, load
from ._lib import concrete as _concrete

_X381 = -0xD201000000010000
_X377 = 0x8508C00000000001
_T254 = 4965661367192848881


def _bls_p(x):
    return ((x - 1) ** 2 * (x**4 - x**2 + 1)) // 3 + x


_PARAMS = {
    CURVE_BN254: dict(
        name="BN254",
        p=36 * _T254**4 + 36 * _T254**3 + 24 * _T254**2 + 6 * _T254 + 1,
        r=36 * _T254**4 + 36 * _T254**3 + 18 * _T254**2 + 6 * _T254 + 1,
        g1=(1, 2),
        g2=(
            (10857046999023057135944570762232829481370756359578518086990519993285655852781,
             11559732032986387107991004021392285783925812861821192530917403151452391805634),
            (8495653923123431417604973247489272438418190587263600148770280649306958101930,
             4082367875863433681332203403145435568316851327593401208105741076214120093531),
        ),
        zcash_flags=False,
    ),
    CURVE_BLS12_381: dict(
        name="BLS12_381",
        p=_bls_p(_X381),
        r=_X381**4 - _X381**2 + 1,
        g1=(
            3685416753713387016781088315183077757961620795782546409894578378688607592378376318836054947676345821548104185464507,
            1339506544944476473020471379941921221584933875938349620426543736416511423956333506472724655353366534992391756441569,
        ),
        g2=(
            (0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
             0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
            (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
             0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE),
        ),
        zcash_flags=True,
    ),
    CURVE_BLS12_377: dict(
        name="BLS12_377",
        p=_bls_p(_X377),
        r=_X377**4 - _X377**2 + 1,
        g1=(
            81937999373150964239938255573465948239988671502647976594219695644855304257327692006745978603320413799295628339695,
            241266749859715473739788878240585681733927191168601896383759122102112907357779751001206799952863815012735208165030,
        ),
        g2=None,  # supplied by the caller (NewG2FromCoords); gnark's generator is not in the reference tree
        zcash_flags=True,
    ),
}


class Zr:
    """Scalar.  Like the reference's BaseZr (driver/common/big.go:41-137) the value may be negative or
    >= r between operations; it is reduced mod r where the reference reduces (Bytes, and on entry to
    MultiScalarMul via fr.Element.SetBigInt: driver/gurvy/bn254.go:239)."""

    __slots__ = ("v", "curve")

    def __init__(self, v: int, curve: "Curve"):
        self.v = int(v)
        self.curve = curve

    def Plus(self, o: "Zr") -> "Zr":
        return Zr(self.v + o.v, self.curve)

    def Minus(self, o: "Zr") -> "Zr":
        return Zr(self.v - o.v, self.curve)

    def Mul(self, o: "Zr") -> "Zr":
        return Zr(self.v * o.v % self.curve.r, self.curve)

    def Neg(self) -> "Zr":
        return Zr(-self.v, self.curve)

    def Equals(self, o: "Zr") -> bool:
        return (self.v - o.v) % self.curve.r == 0

    def Copy(self) -> "Zr":
        return Zr(self.v, self.curve)

    def Bytes(self) -> bytes:
        return (self.v % self.curve.r).to_bytes(32, "big")

    def le_bytes(self, mont: bool = False) -> bytes:
        """32-byte little-endian limbs as handed to the C ABI (fr.Element layout when mont)."""
        s = self.v % self.curve.r
        if mont:
            s = s * (1 << 256) % self.curve.r
        return s.to_bytes(32, "little")


class _Element:
    __slots__ = ("raw", "curve")

    def __init__(self, raw: bytes, curve: "Curve"):
        self.raw = bytes(raw)
        self.curve = curve

    def Equals(self, o) -> bool:
        return self.raw == o.raw  # affine Montgomery coordinates are canonical

    def Copy(self):
        return type(self)(self.raw, self.curve)


class G1(_Element):
    def IsInfinity(self) -> bool:
        return not any(self.raw)

    def Mul(self, s: Zr) -> "G1":
        return self.curve.MultiScalarMul([self], [s])

    def Mul2(self, e: Zr, Q: "G1", f: Zr) -> "G1":
        return self.curve.MultiScalarMul([self, Q], [e, f])

    def Add(self, o: "G1") -> None:
        out = ctypes.create_string_buffer(self.curve.g1_bytes)
        check(load().mlhip_g1_sum(self.curve.id, self.raw + o.raw, 2, out))
        self.raw = out.raw

    def Neg(self) -> None:
        if self.IsInfinity():
            return
        n = self.curve.fp_bytes
        y = int.from_bytes(self.raw[n:], "little")
        self.raw = self.raw[:n] + ((self.curve.p - y) % self.curve.p).to_bytes(n, "little")

    def Sub(self, o: "G1") -> None:
        t = o.Copy()
        t.Neg()
        self.Add(t)

    def coords(self):
        if self.IsInfinity():
            return None
        c = self.curve
        return tuple(c._from_mont(self.raw[i * c.fp_bytes : (i + 1) * c.fp_bytes]) for i in range(2))

    def Bytes(self) -> bytes:
        """uncompressed wire form (gnark RawBytes; reference bls12-381.go:286-290)"""
        c = self.curve
        n = c.fp_bytes
        xy = self.coords()
        if xy is None:
            out = bytearray(2 * n)
            out[0] |= 0x40
            return bytes(out)
        return xy[0].to_bytes(n, "big") + xy[1].to_bytes(n, "big")

    def Compressed(self) -> bytes:
        c = self.curve
        n = c.fp_bytes
        xy = self.coords()
        if c.zcash_flags:
            if xy is None:
                return bytes([0xC0]) + bytes(n - 1)
            out = bytearray(xy[0].to_bytes(n, "big"))
            out[0] |= 0x80 | (0x20 if xy[1] > (c.p - 1) // 2 else 0)
            return bytes(out)
        if xy is None:
            return bytes([0x40]) + bytes(n - 1)
        out = bytearray(xy[0].to_bytes(n, "big"))
        out[0] |= 0xC0 if xy[1] > (c.p - 1) // 2 else 0x80
        return bytes(out)

    def String(self) -> str:
        xy = self.coords()
        return "(0,0)" if xy is None else "(%d,%d)" % xy


class G2(_Element):
    def IsInfinity(self) -> bool:
        return not any(self.raw)

    def Mul(self, s: Zr) -> "G2":
        return self.curve.MultiScalarMulG2([self], [s])

    def Add(self, o: "G2") -> None:
        out = ctypes.create_string_buffer(self.curve.g2_bytes)
        check(load().mlhip_g2_sum(self.curve.id, self.raw + o.raw, 2, out))
        self.raw = out.raw

    def coords(self):
        if self.IsInfinity():
            return None
        c = self.curve
        v = [c._from_mont(self.raw[i * c.fp_bytes : (i + 1) * c.fp_bytes]) for i in range(4)]
        return ((v[0], v[1]), (v[2], v[3]))

    def Bytes(self) -> bytes:
        """uncompressed wire form: X.A1 | X.A0 | Y.A1 | Y.A0, big-endian"""
        c = self.curve
        n = c.fp_bytes
        q = self.coords()
        if q is None:
            out = bytearray(4 * n)
            out[0] |= 0x40
            return bytes(out)
        return b"".join(v.to_bytes(n, "big") for v in (q[0][1], q[0][0], q[1][1], q[1][0]))

    def Compressed(self) -> bytes:
        """compressed wire form X.A1 | X.A0; the sign bit follows gnark's E2 rule (A1 decides unless zero)"""
        c = self.curve
        n = c.fp_bytes
        q = self.coords()
        zcash = c.id != 0
        if q is None:
            out = bytearray(2 * n)
            out[0] = 0xC0 if zcash else 0x40
            return bytes(out)
        out = bytearray(q[0][1].to_bytes(n, "big") + q[0][0].to_bytes(n, "big"))
        half = (c.p - 1) // 2
        big = q[1][1] > half if q[1][1] else q[1][0] > half
        out[0] |= (0x80 | (0x20 if big else 0)) if zcash else (0xC0 if big else 0x80)
        return bytes(out)


class Gt(_Element):
    def Mul(self, o: "Gt") -> None:
        out = ctypes.create_string_buffer(self.curve.gt_bytes)
        check(load().mlhip_gt_mul(self.curve.id, self.raw, o.raw, 1, out))
        self.raw = out.raw

    def Exp(self, x: "Zr") -> "Gt":
        """driver/math.go:358-359 / bls12-381.go:399-407"""
        c = self.curve
        out = ctypes.create_string_buffer(c.gt_bytes)
        check(load().mlhip_gt_exp(c.id, self.raw, x.le_bytes(c.scalars_mont), 1 if c.scalars_mont else 0, 1, out))
        return Gt(out.raw, c)

    def IsUnity(self) -> bool:
        return self.raw == self.curve._gt_one

    def Bytes(self) -> bytes:
        """gnark GT.Bytes(): 12 big-endian Fp, C1.B2.A1 first ... C0.B0.A0 last"""
        c = self.curve
        n = c.fp_bytes
        v = [c._from_mont(self.raw[i * n : (i + 1) * n]) for i in range(12)]
        return b"".join(x.to_bytes(n, "big") for x in reversed(v))


class Curve:
    """One curve of the HIP backend (the reference registers one driver.Curve per entry of
    math.Curves, math.go:142-255).  Nothing touches the GPU until a hot method is called."""

    def __init__(self, curve_id: int, window_c: int = 0, scalars_mont: bool = True):
        prm = _PARAMS[curve_id]
        self.id = curve_id
        self.name = prm["name"]
        self.p, self.r = prm["p"], prm["r"]
        self.zcash_flags = prm["zcash_flags"]
        self.window_c = window_c
        # the gurvy BLS12-381 driver hands fr.Element (Montgomery) scalars to MultiExp (bls12-381.go:772)
        self.scalars_mont = scalars_mont
        self.fp_bytes, self.g1_bytes, self.g2_bytes, self.gt_bytes = _lib.sizes(curve_id)
        self.Rm = 1 << (8 * self.fp_bytes)
        self._Rinv = pow(self.Rm, -1, self.p)
        self._gt_one = self._to_mont(1) + bytes(11 * self.fp_bytes)
        self.GroupOrder = Zr(self.r, self)
        self._g1 = prm["g1"]
        self._g2 = prm["g2"]

    # ---- plumbing
    def _to_mont(self, a: int) -> bytes:
        return (a % self.p * self.Rm % self.p).to_bytes(self.fp_bytes, "little")

    def _from_mont(self, b: bytes) -> int:
        return int.from_bytes(b, "little") * self._Rinv % self.p

    def NewG1FromCoords(self, x: int, y: int) -> G1:
        return G1(self._to_mont(x) + self._to_mont(y), self)

    def NewG2FromCoords(self, x, y) -> G2:
        return G2(b"".join(self._to_mont(v) for v in (x[0], x[1], y[0], y[1])), self)

    _CODEC_ERRORS = {1: "invalid point encoding", 2: "invalid point: not on the curve", 3: "invalid point: subgroup check failed"}

    def _from_wire(self, group: int, b: bytes, compressed: bool):
        size = self.g1_bytes if group == 1 else self.g2_bytes
        out = ctypes.create_string_buffer(size)
        st = ctypes.create_string_buffer(1)
        if len(b) != (size // 2 if compressed else size):
            raise ValueError("set bytes failed [invalid length]")
        fn = load().mlhip_g1_from_bytes if group == 1 else load().mlhip_g2_from_bytes
        check(fn(self.id, bytes(b), 1, 1 if compressed else 0, 1, out, st))
        if st.raw[0]:
            # the reference panics ("set bytes failed [...]"), the facade turns it into an error (math.go:761-832)
            raise ValueError("set bytes failed [%s]" % self._CODEC_ERRORS[st.raw[0]])
        return G1(out.raw, self) if group == 1 else G2(out.raw, self)

    def _g1_from_wire(self, b: bytes, compressed: bool) -> G1:
        return self._from_wire(1, b, compressed)

    def NewG2FromBytes(self, b: bytes) -> G2:
        """bls12-381.go:541-549"""
        return self._from_wire(2, b, False)

    def NewG2FromCompressed(self, b: bytes) -> G2:
        """bls12-381.go:561-569"""
        return self._from_wire(2, b, True)

    def NewG1FromBytes(self, b: bytes) -> G1:
        """uncompressed wire form, subgroup-checked (driver/gurvy/bls12381/bls12-381.go:531-539)"""
        return self._g1_from_wire(b, False)

    def NewG1FromCompressed(self, b: bytes) -> G1:
        """compressed wire form (bls12-381.go:551-559)"""
        return self._g1_from_wire(b, True)

    def NewG1(self) -> G1:
        return G1(bytes(self.g1_bytes), self)

    def NewG2(self) -> G2:
        return G2(bytes(self.g2_bytes), self)

    def NewZrFromInt(self, i: int) -> Zr:
        return Zr(i, self)

    def NewRandomZr(self, rng=None) -> Zr:
        """driver/common/curve.go:77-84 (crypto/rand there; a caller-seeded stream here when given)"""
        if rng is None:
            return Zr(secrets.randbelow(self.r), self)
        return Zr(rng(self.r), self)

    def HashToZr(self, data: bytes) -> Zr:
        """SHA-256 mod r, driver/common/curve.go:86-92"""
        return Zr(int.from_bytes(hashlib.sha256(data).digest(), "big") % self.r, self)

    def GenG1(self) -> G1:
        return self.NewG1FromCoords(*self._g1)

    def GenG2(self) -> G2:
        if self._g2 is None:
            raise ValueError("no built-in G2 generator for %s; use NewG2FromCoords" % self.name)
        return self.NewG2FromCoords(*self._g2)

    def GenGt(self) -> Gt:
        """FExp(Pairing(GenG2, GenG1)), reference bls12-381.go:490-497"""
        return self.FExp(self.Pairing(self.GenG2(), self.GenG1()))

    # ---- hot path
    def _scalars(self, b: Sequence[Zr]) -> bytes:
        return b"".join(z.le_bytes(self.scalars_mont) for z in b)

    def MultiScalarMul(self, a: Sequence[G1], b: Sequence[Zr]) -> G1:
        """math.go:960-969 raises (index out of range) when len(b) < len(a); mirrored."""
        if len(b) < len(a):
            raise IndexError("MultiScalarMul: fewer scalars than points")
        out = ctypes.create_string_buffer(self.g1_bytes)
        if len(b) != len(a):
            # gnark's MultiExp errors on a length mismatch and the driver drops the error: identity
            return self.NewG1()
        pts = b"".join(p.raw for p in a)
        check(load().mlhip_msm_g1(self.id, pts, self._scalars(b), 1 if self.scalars_mont else 0, len(a), self.window_c, out))
        return G1(out.raw, self)

    def MultiScalarMulG2(self, a: Sequence[G2], b: Sequence[Zr]) -> G2:
        if len(b) < len(a):
            raise IndexError("MultiScalarMulG2: fewer scalars than points")
        out = ctypes.create_string_buffer(self.g2_bytes)
        if len(b) != len(a):
            return self.NewG2()
        pts = b"".join(p.raw for p in a)
        check(load().mlhip_msm_g2(self.id, pts, self._scalars(b), 1 if self.scalars_mont else 0, len(a), self.window_c, out))
        return G2(out.raw, self)

    def MultiScalarMulG1G2(self, a1: Sequence[G1], a2: Sequence[G2], b: Sequence[Zr]):
        """(MultiScalarMul(a1, b), MultiScalarMulG2(a2, b)) for ONE scalar vector: sorted once on the device, both groups
        accumulate from the same lists (additive; BASELINE configs[3]).  Same length rules as MultiScalarMul."""
        if len(b) < len(a1) or len(b) < len(a2):
            raise IndexError("MultiScalarMulG1G2: fewer scalars than points")
        if len(a1) != len(a2) or len(b) != len(a1):
            return self.NewG1(), self.NewG2()
        o1 = ctypes.create_string_buffer(self.g1_bytes)
        o2 = ctypes.create_string_buffer(self.g2_bytes)
        check(load().mlhip_msm_g1g2(self.id, b"".join(p.raw for p in a1), b"".join(p.raw for p in a2), self._scalars(b),
                                    1 if self.scalars_mont else 0, len(a1), self.window_c, o1, o2))
        return G1(o1.raw, self), G2(o2.raw, self)

    def NewBases(self, points: Sequence[G1]) -> "Bases":
        """Upload a G1 point table once; Bases.MultiScalarMul(scalars) then moves only the scalars."""
        return Bases(self, points)

    def Pairing(self, p2: G2, p1: G1) -> Gt:
        """Miller loop only, like the gurvy drivers (bls12-381.go:448-455); compare after FExp."""
        out = ctypes.create_string_buffer(self.gt_bytes)
        check(load().mlhip_miller_loop(self.id, p1.raw, p2.raw, 1, 1, out))
        return Gt(out.raw, self)

    def Pairing2(self, p2a: G2, p2b: G2, p1a: G1, p1b: G1) -> Gt:
        out = ctypes.create_string_buffer(self.gt_bytes)
        check(load().mlhip_miller_loop(self.id, p1a.raw + p1b.raw, p2a.raw + p2b.raw, 2, 1, out))
        return Gt(out.raw, self)

    def FExp(self, a: Gt) -> Gt:
        out = ctypes.create_string_buffer(self.gt_bytes)
        check(load().mlhip_final_exp(self.id, a.raw, 1, out))
        return Gt(out.raw, self)

    def PairingBatch(self, g2s: Sequence[G2], g1s: Sequence[G1]) -> List[Gt]:
        """out[i] = FExp(Pairing(g2s[i], g1s[i])) -- additive batch entry point."""
        if len(g2s) != len(g1s):
            raise ValueError("PairingBatch: length mismatch")
        n = len(g1s)
        out = ctypes.create_string_buffer(self.gt_bytes * max(n, 1))
        check(load().mlhip_pairing_batch(self.id, b"".join(p.raw for p in g1s), b"".join(q.raw for q in g2s), n, out))
        return [Gt(out.raw[i * self.gt_bytes : (i + 1) * self.gt_bytes], self) for i in range(n)]


    def MulBatch(self, points: Sequence, scalars: Sequence[Zr]) -> list:
        """out[i] = points[i].Mul(scalars[i]) for G1 or G2 points, one launch (additive; SURVEY 8f row 3: the batched form of
        G1.Mul / G2.Mul, bls12-381.go:238-247, :342-351)."""
        if len(points) != len(scalars):
            raise ValueError("MulBatch: length mismatch")
        n = len(points)
        if n == 0:
            return []
        cls = type(points[0])
        group, size = (GROUP_G1, self.g1_bytes) if cls is G1 else (GROUP_G2, self.g2_bytes)
        out = ctypes.create_string_buffer(size * n)
        check(load().mlhip_scalar_mul(self.id, group, b"".join(p.raw for p in points), 1, self._scalars(scalars),
                                      1 if self.scalars_mont else 0, n, out))
        return [cls(out.raw[i * size : (i + 1) * size], self) for i in range(n)]

    def BaseMulBatch(self, base, scalars: Sequence[Zr]) -> list:
        """out[i] = base.Mul(scalars[i]): one base (a generator, a Pedersen base), many scalars.  From 2^12 scalars on the
        library multiplies through a table of the base's multiples that it keeps on the device for later calls with the
        same base (mlhip_scalar_mul, point_stride 0)."""
        n = len(scalars)
        if n == 0:
            return []
        cls = type(base)
        group, size = (GROUP_G1, self.g1_bytes) if cls is G1 else (GROUP_G2, self.g2_bytes)
        out = ctypes.create_string_buffer(size * n)
        check(load().mlhip_scalar_mul(self.id, group, base.raw, 0, self._scalars(scalars), 1 if self.scalars_mont else 0, n, out))
        return [cls(out.raw[i * size : (i + 1) * size], self) for i in range(n)]

    def ExpBatch(self, gts: Sequence[Gt], scalars: Sequence[Zr]) -> List[Gt]:
        """out[i] = gts[i].Exp(scalars[i]), one launch (additive; SURVEY 8f row 2: Gt.Exp, bls12-381.go:399-407)."""
        if len(gts) != len(scalars):
            raise ValueError("ExpBatch: length mismatch")
        n = len(gts)
        if n == 0:
            return []
        out = ctypes.create_string_buffer(self.gt_bytes * n)
        check(load().mlhip_gt_exp(self.id, b"".join(g.raw for g in gts), self._scalars(scalars), 1 if self.scalars_mont else 0, n, out))
        return [Gt(out.raw[i * self.gt_bytes : (i + 1) * self.gt_bytes], self) for i in range(n)]

    def PairingProduct(self, g2s: Sequence[G2], g1s: Sequence[G1]) -> Gt:
        """FExp(prod_i Pairing(g2s[i], g1s[i])) with one shared final exponentiation (additive API)."""
        if len(g2s) != len(g1s):
            raise ValueError("PairingProduct: length mismatch")
        out = ctypes.create_string_buffer(self.gt_bytes)
        check(load().mlhip_pairing_product(self.id, b"".join(p.raw for p in g1s), b"".join(q.raw for q in g2s), len(g1s), out))
        return Gt(out.raw, self)


def SetDevices(*devices: int) -> None:
    """the GPUs of this process (hip.SetDevices in the Go shim -> mlhip_init): with two or more listed, large
    MultiScalarMul / MultiScalarMulG2 / NewBases / PairingBatch calls are sharded over them inside the library"""
    _lib.init_devices(list(devices))


def NewCurve(name: str, **kw) -> Curve:
    ids = {"BN254": CURVE_BN254, "BLS12_381": CURVE_BLS12_381, "BLS12_377": CURVE_BLS12_377}
    return Curve(ids[name.upper().replace("-", "_")], **kw)


class Bases:
    """Resident G1 bases (mlhip_bases_*): the additive API a prover with a fixed SRS would use."""

    def __init__(self, curve: "Curve", points: Sequence[G1]):
        self.curve = curve
        self.n = len(points)
        self._h = ctypes.c_void_p()
        blob = b"".join(p.raw for p in points)
        self._lib = _concrete()  # the handle stays with the library that made it
        check(self._lib.mlhip_bases_create(curve.id, GROUP_G1, blob, self.n, curve.window_c, ctypes.byref(self._h)))

    def MultiScalarMul(self, scalars: Sequence[Zr]) -> G1:
        if len(scalars) > self.n:
            raise IndexError("MultiScalarMul: more scalars than resident bases")
        out = ctypes.create_string_buffer(self.curve.g1_bytes)
        c = self.curve
        check(self._lib.mlhip_bases_msm(self._h, c._scalars(scalars), 1 if c.scalars_mont else 0, len(scalars), out))
        return G1(out.raw, c)

    def CheckedSubgroup(self) -> bool:
        """every point of the table was verified on the device to lie in G1 (BLS12-377: Edwards bucket sums)"""
        return self._lib.mlhip_bases_checked_subgroup(self._h) == 1

    def ShiftedTables(self) -> bool:
        """the handle keeps shifted-base tables (include/mlhip.h: mlhip_bases_create): one bucket set for all digits"""
        from ._lib import plan_timings

        plan = self._lib.mlhip_bases_plan(self._h)
        return bool(plan) and plan_timings(self._lib, plan).get("tables") == 1.0

    def Close(self) -> None:
        if self._h:
            self._lib.mlhip_bases_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.Close()
        except Exception:
            pass
