// mlhip_driver.hpp -- header-only C++17 mirror of the reference's plugin boundary for the hot path
// (github.com/IBM/mathlib driver/math.go:49-360: interfaces Curve, Zr, G1, G2, Gt) over the C ABI of
// libmlhip.so (include/mlhip.h).
//
// The reference is Go and the build image has no Go toolchain, so the cgo package a maintainer would add is
// shipped as source (go/driver/hip); this header is the same host-side layer in a compiled
// language that DOES build here: same method names, argument order and error behaviour (the Go drivers
// panic on failure, driver/gurvy/bn254.go:249-251 -> these methods throw std::runtime_error), so
// tests/cpp/driver_test.cpp reads like math_test.go.  Only plumbing lives here (byte packing, Montgomery
// <-> integer conversion for wire bytes, scalar arithmetic mod r); every group / field operation on points
// and Gt values is done by libmlhip.so.  There is no CPU fallback: without a GPU the hot methods throw.
//
//   Curve::MultiScalarMul(a, b)          driver/math.go:169-170  (math.go:960-969)   -> mlhip_msm_g1
//   Curve::Pairing(g2, g1)               driver/math.go:50-52    Miller loop only     -> mlhip_miller_loop
//   Curve::Pairing2(p2a, p2b, p1a, p1b)  driver/math.go:54-55                         -> mlhip_miller_loop (2 pairs)
//   Curve::FExp(gt)                      driver/math.go:56-57                         -> mlhip_final_exp
//   G1::Mul / Mul2 / Add / Sub / Neg     driver/math.go:249-288
//   G2::Mul / Add, Gt::Mul / Exp / IsUnity, Zr::Plus / Minus / Mul / ...             driver/math.go:191-360
//   additive: MultiScalarMulG2, MultiScalarMulG1G2, PairingBatch, PairingProduct, MulBatch, BaseMulBatch, ExpBatch
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "mlhip.h"

namespace mlhip_driver {

typedef std::vector<uint8_t> Bytes;

// ---------------------------------------------------------------------------------------------------
// small fixed-size modular arithmetic (host plumbing only: scalars mod r, coordinate (de)Montgomery)
// ---------------------------------------------------------------------------------------------------
struct Mod {
  int n = 0;  // 64-bit limbs
  uint64_t p[6] = {0}, r2[6] = {0}, inv = 0;

  static int hexval(char c) { return c <= '9' ? c - '0' : (c | 32) - 'a' + 10; }
  void init(const char* hex, int limbs) {
    n = limbs;
    std::string h(hex);
    for (int i = 0; i < 6; i++) p[i] = 0;
    int bit = 0;
    for (int i = (int)h.size() - 1; i >= 0; i--, bit += 4) p[bit >> 6] |= (uint64_t)hexval(h[i]) << (bit & 63);
    uint64_t x = p[0];  // Newton: x <- x (2 - p0 x)
    for (int i = 0; i < 6; i++) x *= 2 - p[0] * x;
    inv = (uint64_t)0 - x;
    // R^2 mod p by 2*64*n doublings of 1
    uint64_t t[6] = {1, 0, 0, 0, 0, 0};
    for (int i = 0; i < 128 * n; i++) add(t, t, t);
    memcpy(r2, t, sizeof(t));
  }
  bool geq_p(const uint64_t* a, uint64_t carry) const {
    if (carry) return true;
    for (int i = n - 1; i >= 0; i--)
      if (a[i] != p[i]) return a[i] > p[i];
    return true;
  }
  void add(uint64_t* r, const uint64_t* a, const uint64_t* b) const {
    unsigned __int128 c = 0;
    uint64_t t[6];
    for (int i = 0; i < n; i++) {
      c += (unsigned __int128)a[i] + b[i];
      t[i] = (uint64_t)c;
      c >>= 64;
    }
    if (geq_p(t, (uint64_t)c)) {
      unsigned __int128 br = 0;
      for (int i = 0; i < n; i++) {
        unsigned __int128 s = (unsigned __int128)t[i] - p[i] - br;
        t[i] = (uint64_t)s;
        br = (s >> 64) & 1;
      }
    }
    memcpy(r, t, 8 * n);
  }
  void sub(uint64_t* r, const uint64_t* a, const uint64_t* b) const {
    unsigned __int128 br = 0;
    uint64_t t[6];
    for (int i = 0; i < n; i++) {
      unsigned __int128 s = (unsigned __int128)a[i] - b[i] - br;
      t[i] = (uint64_t)s;
      br = (s >> 64) & 1;
    }
    if (br) {
      unsigned __int128 c = 0;
      for (int i = 0; i < n; i++) {
        c += (unsigned __int128)t[i] + p[i];
        t[i] = (uint64_t)c;
        c >>= 64;
      }
    }
    memcpy(r, t, 8 * n);
  }
  void mont_mul(uint64_t* r, const uint64_t* a, const uint64_t* b) const {
    uint64_t t[8] = {0};
    for (int i = 0; i < n; i++) {
      unsigned __int128 c = 0;
      for (int j = 0; j < n; j++) {
        unsigned __int128 acc = (unsigned __int128)a[j] * b[i] + t[j] + c;
        t[j] = (uint64_t)acc;
        c = acc >> 64;
      }
      unsigned __int128 acc = (unsigned __int128)t[n] + c;
      t[n] = (uint64_t)acc;
      t[n + 1] = (uint64_t)(acc >> 64);
      uint64_t m = t[0] * inv;
      acc = (unsigned __int128)m * p[0] + t[0];
      c = acc >> 64;
      for (int j = 1; j < n; j++) {
        acc = (unsigned __int128)m * p[j] + t[j] + c;
        t[j - 1] = (uint64_t)acc;
        c = acc >> 64;
      }
      acc = (unsigned __int128)t[n] + c;
      t[n - 1] = (uint64_t)acc;
      t[n] = t[n + 1] + (uint64_t)(acc >> 64);
    }
    if (geq_p(t, t[n])) {
      unsigned __int128 br = 0;
      for (int i = 0; i < n; i++) {
        unsigned __int128 s = (unsigned __int128)t[i] - p[i] - br;
        t[i] = (uint64_t)s;
        br = (s >> 64) & 1;
      }
    }
    memcpy(r, t, 8 * n);
  }
  void to_mont(uint64_t* r, const uint64_t* a) const { mont_mul(r, a, r2); }
  void from_mont(uint64_t* r, const uint64_t* a) const {
    uint64_t one[6] = {1, 0, 0, 0, 0, 0};
    mont_mul(r, a, one);
  }
  void mul(uint64_t* r, const uint64_t* a, const uint64_t* b) const {  // plain a*b mod p
    uint64_t am[6];
    to_mont(am, a);
    mont_mul(r, am, b);
  }
  bool is_zero(const uint64_t* a) const {
    uint64_t o = 0;
    for (int i = 0; i < n; i++) o |= a[i];
    return o == 0;
  }
  // a > (p-1)/2 ?  (the "lexicographically largest" flag of the compressed encodings)
  bool is_upper_half(const uint64_t* a) const {
    uint64_t t[6];
    unsigned __int128 c = 0;
    for (int i = 0; i < n; i++) {  // 2a
      c += (unsigned __int128)a[i] + a[i];
      t[i] = (uint64_t)c;
      c >>= 64;
    }
    if (c) return true;
    for (int i = n - 1; i >= 0; i--)
      if (t[i] != p[i]) return t[i] > p[i];
    return false;
  }
};

inline Bytes be_bytes(const uint64_t* a, int n) {
  Bytes out(8 * n);
  for (int i = 0; i < n; i++)
    for (int k = 0; k < 8; k++) out[8 * (n - 1 - i) + (7 - k)] = (uint8_t)(a[i] >> (8 * k));
  return out;
}

class Curve;

inline void check(int rc) {
  if (rc != 0) throw std::runtime_error(std::string("libmlhip error ") + std::to_string(rc) + ": " + mlhip_last_error());
}

// ---------------------------------------------------------------------------------------------------
class Zr {
 public:
  std::array<uint64_t, 4> v{};  // canonical, < r
  const Curve* curve = nullptr;
  Zr Plus(const Zr& o) const;
  Zr Minus(const Zr& o) const;
  Zr Mul(const Zr& o) const;
  Zr Neg() const;
  bool Equals(const Zr& o) const { return v == o.v; }
  Zr Copy() const { return *this; }
  Bytes ToBytes() const { return be_bytes(v.data(), 4); }  // driver.Zr.Bytes(): 32 bytes big-endian
  std::array<uint64_t, 4> abi_limbs() const;                // what crosses the C ABI (fr.Element when the curve says so)
};

class G1 {
 public:
  Bytes raw;  // G1Affine{X,Y}: Montgomery, little-endian limbs; infinity = all zero
  const Curve* curve = nullptr;
  bool IsInfinity() const {
    for (uint8_t b : raw)
      if (b) return false;
    return true;
  }
  bool Equals(const G1& o) const { return raw == o.raw; }
  G1 Copy() const { return *this; }
  G1 Mul(const Zr& s) const;
  G1 Mul2(const Zr& e, const G1& Q, const Zr& f) const;
  void Add(const G1& o);
  void Sub(const G1& o);
  void Neg();
  Bytes ToBytes() const;     // uncompressed wire form (gnark RawBytes)
  Bytes Compressed() const;  // compressed wire form (gnark Bytes)
};

class G2 {
 public:
  Bytes raw;
  const Curve* curve = nullptr;
  bool IsInfinity() const {
    for (uint8_t b : raw)
      if (b) return false;
    return true;
  }
  bool Equals(const G2& o) const { return raw == o.raw; }
  G2 Copy() const { return *this; }
  G2 Mul(const Zr& s) const;
  void Add(const G2& o);
  Bytes ToBytes() const;     // X.A1 | X.A0 | Y.A1 | Y.A0 (gnark RawBytes), encoded on the device
  Bytes Compressed() const;  // X.A1 | X.A0 with the header bits
};

class Gt {
 public:
  Bytes raw;
  const Curve* curve = nullptr;
  bool Equals(const Gt& o) const { return raw == o.raw; }
  void Mul(const Gt& o);
  Gt Exp(const Zr& x) const;
  bool IsUnity() const;
  Bytes ToBytes() const;  // gnark GT.Bytes(): 12 big-endian Fp, C1.B2.A1 first
};

// ---------------------------------------------------------------------------------------------------
// hip.SetDevices of the Go shim: the GPUs of this process (no argument: every visible device).  With two or more, large
// MultiScalarMul / MultiScalarMulG2 / NewBases / PairingBatch calls are sharded over them inside the library.
inline void SetDevices(const std::vector<int>& devices = {}) {
  check(mlhip_init(devices.empty() ? nullptr : devices.data(), (int)devices.size()));
}

class Curve {
 public:
  int id;
  int window_c = 0;
  bool scalars_mont = true;  // the gurvy BLS12-381 driver hands fr.Element (Montgomery) to MultiExp (bls12-381.go:772)
  bool zcash_flags;
  size_t fp_bytes, g1_bytes, g2_bytes, gt_bytes;
  Mod fp, fr;
  Zr GroupOrder;  // r itself: 0 mod r when it reaches the MSM

  explicit Curve(int curve_id) : id(curve_id) {
    check(mlhip_sizes(id, &fp_bytes, &g1_bytes, &g2_bytes, &gt_bytes));
    switch (id) {
      case MLHIP_CURVE_BN254:
        fp.init("30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47", 4);
        fr.init("30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001", 4);  // math_test.go:263
        zcash_flags = false;
        break;
      case MLHIP_CURVE_BLS12_381:
        fp.init("1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab", 6);
        fr.init("73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001", 4);  // math_test.go:265
        zcash_flags = true;
        break;
      case MLHIP_CURVE_BLS12_377:
        fp.init("1ae3a4617c510eac63b05c06ca1493b1a22d9f300f5138f1ef3622fba094800170b5d44300000008508c00000000001", 6);
        fr.init("12ab655e9a2ca55660b44d1e5c37b00159aa76fed00000010a11800000000001", 4);  // math_test.go:266
        zcash_flags = true;
        break;
      default:
        throw std::runtime_error("unknown curve id");
    }
    GroupOrder.curve = this;  // value 0: r mod r
  }

  // ---- element constructors
  Zr NewZrFromInt(int64_t i) const {
    Zr z;
    z.curve = this;
    uint64_t a[6] = {(uint64_t)(i < 0 ? -i : i), 0, 0, 0, 0, 0}, zero[6] = {0};
    if (i < 0)
      fr.sub(a, zero, a);
    for (int k = 0; k < 4; k++) z.v[k] = a[k];
    return z;
  }
  Zr NewZrFromLimbs(const uint64_t limbs[4]) const {  // reduced mod r
    Zr z;
    z.curve = this;
    uint64_t a[6] = {limbs[0], limbs[1], limbs[2], limbs[3], 0, 0}, one[6] = {1, 0, 0, 0, 0, 0}, am[6];
    // a mod r via one Montgomery round trip: (a * R) * 1 / R handles a < 2^256 only if a < r*... use subtraction loop
    while (fr.geq_p(a, 0)) {
      unsigned __int128 br = 0;
      for (int k = 0; k < 4; k++) {
        unsigned __int128 s = (unsigned __int128)a[k] - fr.p[k] - br;
        a[k] = (uint64_t)s;
        br = (s >> 64) & 1;
      }
    }
    (void)one;
    (void)am;
    for (int k = 0; k < 4; k++) z.v[k] = a[k];
    return z;
  }
  // deterministic pseudo-random scalar (the reference uses crypto/rand: driver/common/curve.go:77-84)
  Zr NewRandomZr(uint64_t& state) const {
    uint64_t l[4];
    for (int k = 0; k < 4; k++) {
      state += 0x9E3779B97F4A7C15ull;
      uint64_t z = state;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      l[k] = z ^ (z >> 31);
    }
    l[3] >>= 2;
    return NewZrFromLimbs(l);
  }
  G1 NewG1() const {
    G1 g;
    g.curve = this;
    g.raw.assign(g1_bytes, 0);
    return g;
  }
  // NewG1FromBytes / NewG1FromCompressed (driver/gurvy/bls12381/bls12-381.go:531-559): decoded, curve- and
  // subgroup-checked on the device; the reference panics on bad input ("set bytes failed [...]") -> throws here.
  G1 g1_from_wire(const Bytes& b, bool compressed) const {
    if (b.size() != (size_t)(compressed ? 1 : 2) * fp_bytes) throw std::invalid_argument("set bytes failed [invalid length]");
    G1 g = NewG1();
    unsigned char st = 0;
    check(mlhip_g1_from_bytes(id, b.data(), 1, compressed ? 1 : 0, 1, g.raw.data(), &st));
    if (st) throw std::invalid_argument(std::string("set bytes failed [status ") + std::to_string((int)st) + "]");
    return g;
  }
  G2 g2_from_wire(const Bytes& b, bool compressed) const {
    if (b.size() != (size_t)(compressed ? 2 : 4) * fp_bytes) throw std::invalid_argument("set bytes failed [invalid length]");
    G2 g = NewG2();
    unsigned char st = 0;
    check(mlhip_g2_from_bytes(id, b.data(), 1, compressed ? 1 : 0, 1, g.raw.data(), &st));
    if (st) throw std::invalid_argument(std::string("set bytes failed [status ") + std::to_string((int)st) + "]");
    return g;
  }
  G2 NewG2FromBytes(const Bytes& b) const { return g2_from_wire(b, false); }       // bls12-381.go:541-549
  G2 NewG2FromCompressed(const Bytes& b) const { return g2_from_wire(b, true); }  // bls12-381.go:561-569
  G1 NewG1FromBytes(const Bytes& b) const { return g1_from_wire(b, false); }
  G1 NewG1FromCompressed(const Bytes& b) const { return g1_from_wire(b, true); }
  G2 NewG2() const {
    G2 g;
    g.curve = this;
    g.raw.assign(g2_bytes, 0);
    return g;
  }
  Bytes coord_mont(const char* dec_or_hex, bool hex) const {
    uint64_t a[6] = {0};
    std::string s(dec_or_hex);
    for (char ch : s) {  // a = a*base + digit (mod p), plain representation
      uint64_t base = hex ? 16 : 10, carry = (uint64_t)Mod::hexval(ch);
      for (int i = 0; i < fp.n; i++) {
        unsigned __int128 t = (unsigned __int128)a[i] * base + carry;
        a[i] = (uint64_t)t;
        carry = (uint64_t)(t >> 64);
      }
    }
    uint64_t m[6];
    fp.to_mont(m, a);
    Bytes out(fp_bytes);
    memcpy(out.data(), m, fp_bytes);
    return out;
  }
  G1 NewG1FromCoords(const char* x_dec, const char* y_dec) const {
    G1 g;
    g.curve = this;
    g.raw = coord_mont(x_dec, false);
    Bytes y = coord_mont(y_dec, false);
    g.raw.insert(g.raw.end(), y.begin(), y.end());
    return g;
  }
  G2 NewG2FromCoords(const char* x0, const char* x1, const char* y0, const char* y1) const {
    G2 g;
    g.curve = this;
    for (const char* c : {x0, x1, y0, y1}) {
      Bytes b = coord_mont(c, false);
      g.raw.insert(g.raw.end(), b.begin(), b.end());
    }
    return g;
  }
  G1 GenG1() const {  // math_test.go:250-259 (expectedG1Gens)
    switch (id) {
      case MLHIP_CURVE_BN254: return NewG1FromCoords("1", "2");
      case MLHIP_CURVE_BLS12_381:
        return NewG1FromCoords(
            "3685416753713387016781088315183077757961620795782546409894578378688607592378376318836054947676345821548104185464507",
            "1339506544944476473020471379941921221584933875938349620426543736416511423956333506472724655353366534992391756441569");
      default:
        return NewG1FromCoords(
            "81937999373150964239938255573465948239988671502647976594219695644855304257327692006745978603320413799295628339695",
            "241266749859715473739788878240585681733927191168601896383759122102112907357779751001206799952863815012735208165030");
    }
  }
  G2 GenG2() const {
    switch (id) {
      case MLHIP_CURVE_BN254:
        return NewG2FromCoords("10857046999023057135944570762232829481370756359578518086990519993285655852781",
                               "11559732032986387107991004021392285783925812861821192530917403151452391805634",
                               "8495653923123431417604973247489272438418190587263600148770280649306958101930",
                               "4082367875863433681332203403145435568316851327593401208105741076214120093531");
      case MLHIP_CURVE_BLS12_381:
        return NewG2FromCoords(
            "352701069587466618187139116011060144890029952792775240219908644239793785735715026873347600343865175952761926303160",
            "3059144344244213709971259814753781636986470325476647558659373206291635324768958432433509563104347017837885763365758",
            "1985150602287291935568054521177171638300868978215655730859378665066344726373823718423869104263333984641494340347905",
            "927553665492332455747201965776037880757740193453592970025027978793976877002675564980949289727957565575433344219582");
      default:
        throw std::runtime_error("no built-in G2 generator for BLS12-377; use NewG2FromCoords");
    }
  }

  // ---- hot path ---------------------------------------------------------------------------------
  G1 MultiScalarMul(const std::vector<G1>& a, const std::vector<Zr>& b) const {
    if (b.size() < a.size()) throw std::out_of_range("MultiScalarMul: fewer scalars than points");  // math.go:963-966
    G1 out = NewG1();
    if (b.size() != a.size() || a.empty()) return out;  // MultiExp's dropped error -> identity (bls12-381.go:777)
    Bytes pts, sc;
    pack(a, b, pts, sc);
    check(mlhip_msm_g1(id, pts.data(), sc.data(), scalars_mont ? 1 : 0, a.size(), window_c, out.raw.data()));
    return out;
  }
  G2 MultiScalarMulG2(const std::vector<G2>& a, const std::vector<Zr>& b) const {
    if (b.size() < a.size()) throw std::out_of_range("MultiScalarMulG2: fewer scalars than points");
    G2 out = NewG2();
    if (b.size() != a.size() || a.empty()) return out;
    Bytes pts, sc;
    pack(a, b, pts, sc);
    check(mlhip_msm_g2(id, pts.data(), sc.data(), scalars_mont ? 1 : 0, a.size(), window_c, out.raw.data()));
    return out;
  }
  // (MultiScalarMul(a1, b), MultiScalarMulG2(a2, b)) over ONE scalar vector: one sort on the device for both groups
  std::pair<G1, G2> MultiScalarMulG1G2(const std::vector<G1>& a1, const std::vector<G2>& a2, const std::vector<Zr>& b) const {
    if (b.size() < a1.size() || b.size() < a2.size()) throw std::out_of_range("MultiScalarMulG1G2: fewer scalars than points");
    std::pair<G1, G2> out(NewG1(), NewG2());
    if (a1.size() != a2.size() || b.size() != a1.size() || a1.empty()) return out;
    Bytes p1, p2, sc, sc2;
    pack(a1, b, p1, sc);
    pack(a2, b, p2, sc2);
    check(mlhip_msm_g1g2(id, p1.data(), p2.data(), sc.data(), scalars_mont ? 1 : 0, a1.size(), window_c, out.first.raw.data(),
                         out.second.raw.data()));
    return out;
  }
  Gt Pairing(const G2& p2, const G1& p1) const {  // Miller loop only, like gurvy (bls12-381.go:448-455)
    Gt out = new_gt();
    check(mlhip_miller_loop(id, p1.raw.data(), p2.raw.data(), 1, 1, out.raw.data()));
    return out;
  }
  Gt Pairing2(const G2& p2a, const G2& p2b, const G1& p1a, const G1& p1b) const {
    Bytes g1 = p1a.raw, g2 = p2a.raw;
    g1.insert(g1.end(), p1b.raw.begin(), p1b.raw.end());
    g2.insert(g2.end(), p2b.raw.begin(), p2b.raw.end());
    Gt out = new_gt();
    check(mlhip_miller_loop(id, g1.data(), g2.data(), 2, 1, out.raw.data()));
    return out;
  }
  Gt FExp(const Gt& a) const {
    Gt out = new_gt();
    check(mlhip_final_exp(id, a.raw.data(), 1, out.raw.data()));
    return out;
  }
  std::vector<Gt> PairingBatch(const std::vector<G2>& g2s, const std::vector<G1>& g1s) const {
    if (g2s.size() != g1s.size()) throw std::invalid_argument("PairingBatch: length mismatch");
    Bytes p, q, o(gt_bytes * g1s.size());
    for (auto& x : g1s) p.insert(p.end(), x.raw.begin(), x.raw.end());
    for (auto& x : g2s) q.insert(q.end(), x.raw.begin(), x.raw.end());
    check(mlhip_pairing_batch(id, p.data(), q.data(), g1s.size(), o.data()));
    std::vector<Gt> out;
    for (size_t i = 0; i < g1s.size(); i++) {
      Gt g = new_gt();
      memcpy(g.raw.data(), o.data() + i * gt_bytes, gt_bytes);
      out.push_back(g);
    }
    return out;
  }
  // out[i] = points[i].Mul(scalars[i]), one launch (additive; SURVEY 8f row 3: G1.Mul / G2.Mul, bls12-381.go:238-247, :342-351)
  std::vector<G1> MulBatch(const std::vector<G1>& points, const std::vector<Zr>& scalars) const {
    return mul_batch<G1>(MLHIP_GROUP_G1, g1_bytes, points, 1, scalars);
  }
  std::vector<G2> MulBatch(const std::vector<G2>& points, const std::vector<Zr>& scalars) const {
    return mul_batch<G2>(MLHIP_GROUP_G2, g2_bytes, points, 1, scalars);
  }
  // out[i] = base.Mul(scalars[i]): one base, many scalars -- from 2^12 scalars on through a table of the base's multiples that
  // the library keeps on the device for later calls with the same base
  std::vector<G1> BaseMulBatch(const G1& base, const std::vector<Zr>& scalars) const {
    return mul_batch<G1>(MLHIP_GROUP_G1, g1_bytes, std::vector<G1>(1, base), 0, scalars);
  }
  std::vector<G2> BaseMulBatch(const G2& base, const std::vector<Zr>& scalars) const {
    return mul_batch<G2>(MLHIP_GROUP_G2, g2_bytes, std::vector<G2>(1, base), 0, scalars);
  }
  // out[i] = gts[i].Exp(scalars[i]), one launch (additive; SURVEY 8f row 2: Gt.Exp, bls12-381.go:399-407)
  std::vector<Gt> ExpBatch(const std::vector<Gt>& gts, const std::vector<Zr>& scalars) const {
    if (gts.size() != scalars.size()) throw std::invalid_argument("ExpBatch: length mismatch");
    std::vector<Gt> out;
    if (gts.empty()) return out;
    Bytes in, sc, o(gt_bytes * gts.size());
    pack(gts, scalars, in, sc);
    check(mlhip_gt_exp(id, in.data(), sc.data(), scalars_mont ? 1 : 0, gts.size(), o.data()));
    for (size_t i = 0; i < gts.size(); i++) {
      Gt g = new_gt();
      memcpy(g.raw.data(), o.data() + i * gt_bytes, gt_bytes);
      out.push_back(g);
    }
    return out;
  }
  Gt PairingProduct(const std::vector<G2>& g2s, const std::vector<G1>& g1s) const {
    if (g2s.size() != g1s.size()) throw std::invalid_argument("PairingProduct: length mismatch");
    Bytes p, q;
    for (auto& x : g1s) p.insert(p.end(), x.raw.begin(), x.raw.end());
    for (auto& x : g2s) q.insert(q.end(), x.raw.begin(), x.raw.end());
    Gt out = new_gt();
    check(mlhip_pairing_product(id, p.data(), q.data(), g1s.size(), out.raw.data()));
    return out;
  }
  Gt new_gt() const {
    Gt g;
    g.curve = this;
    g.raw.assign(gt_bytes, 0);
    return g;
  }

 private:
  template <class P>
  std::vector<P> mul_batch(int group, size_t size, const std::vector<P>& points, size_t stride, const std::vector<Zr>& scalars) const {
    if (stride != 0 && points.size() != scalars.size()) throw std::invalid_argument("MulBatch: length mismatch");
    std::vector<P> out;
    const size_t n = scalars.size();
    if (n == 0) return out;
    Bytes pts, sc, o(size * n);
    pack(points, scalars, pts, sc);
    check(mlhip_scalar_mul(id, group, pts.data(), stride, sc.data(), scalars_mont ? 1 : 0, n, o.data()));
    for (size_t i = 0; i < n; i++) {
      P g;
      g.curve = this;
      g.raw.assign(o.begin() + i * size, o.begin() + (i + 1) * size);
      out.push_back(g);
    }
    return out;
  }
  template <class P>
  void pack(const std::vector<P>& a, const std::vector<Zr>& b, Bytes& pts, Bytes& sc) const {
    for (auto& x : a) pts.insert(pts.end(), x.raw.begin(), x.raw.end());
    sc.resize(32 * b.size());
    for (size_t i = 0; i < b.size(); i++) {
      auto l = b[i].abi_limbs();
      memcpy(sc.data() + 32 * i, l.data(), 32);
    }
  }
  friend class Bases;
};

// A G1 point table kept on the device (mlhip_bases_*): uploaded once, then only the scalars move per call.
class Bases {
 public:
  Bases(const Curve& c, const std::vector<G1>& points) : curve_(&c), n_(points.size()) {
    Bytes pts;
    for (auto& x : points) pts.insert(pts.end(), x.raw.begin(), x.raw.end());
    check(mlhip_bases_create(c.id, MLHIP_GROUP_G1, pts.data(), n_, c.window_c, &h_));
  }
  Bases(const Bases&) = delete;
  Bases& operator=(const Bases&) = delete;
  ~Bases() { mlhip_bases_destroy(h_); }
  G1 MultiScalarMul(const std::vector<Zr>& b) const {
    if (b.size() > n_) throw std::out_of_range("MultiScalarMul: more scalars than resident bases");
    G1 out = curve_->NewG1();
    if (b.empty()) return out;
    Bytes sc(32 * b.size());
    for (size_t i = 0; i < b.size(); i++) {
      auto l = b[i].abi_limbs();
      memcpy(sc.data() + 32 * i, l.data(), 32);
    }
    check(mlhip_bases_msm(h_, sc.data(), curve_->scalars_mont ? 1 : 0, b.size(), out.raw.data()));
    return out;
  }
  // every point of the table was verified on the device to lie in G1 (BLS12-377: its MSMs then sum their buckets in
  // twisted Edwards coordinates; a table with a point outside G1 keeps the Weierstrass kernels and the reference's result)
  bool CheckedSubgroup() const { return mlhip_bases_checked_subgroup(h_) == 1; }
  // the handle keeps shifted-base tables (include/mlhip.h: mlhip_bases_create): one bucket set for all digits of a scalar
  bool ShiftedTables() const {
    float t[11] = {0};
    mlhip_msm_plan* p = mlhip_bases_plan(h_);
    return p && mlhip_msm_plan_timings(p, t, 11) >= 11 && t[10] == 1.0f;
  }

 private:
  const Curve* curve_;
  size_t n_;
  mlhip_bases* h_ = nullptr;
};

// ---------------------------------------------------------------------------------------------------
inline Zr Zr::Plus(const Zr& o) const {
  Zr z = *this;
  uint64_t a[6] = {v[0], v[1], v[2], v[3], 0, 0}, b[6] = {o.v[0], o.v[1], o.v[2], o.v[3], 0, 0};
  curve->fr.add(a, a, b);
  for (int k = 0; k < 4; k++) z.v[k] = a[k];
  return z;
}
inline Zr Zr::Minus(const Zr& o) const {
  Zr z = *this;
  uint64_t a[6] = {v[0], v[1], v[2], v[3], 0, 0}, b[6] = {o.v[0], o.v[1], o.v[2], o.v[3], 0, 0};
  curve->fr.sub(a, a, b);
  for (int k = 0; k < 4; k++) z.v[k] = a[k];
  return z;
}
inline Zr Zr::Mul(const Zr& o) const {
  Zr z = *this;
  uint64_t a[6] = {v[0], v[1], v[2], v[3], 0, 0}, b[6] = {o.v[0], o.v[1], o.v[2], o.v[3], 0, 0}, r[6];
  curve->fr.mul(r, a, b);
  for (int k = 0; k < 4; k++) z.v[k] = r[k];
  return z;
}
inline Zr Zr::Neg() const {
  Zr zero;
  zero.curve = curve;
  return zero.Minus(*this);
}
inline std::array<uint64_t, 4> Zr::abi_limbs() const {
  if (!curve->scalars_mont) return v;
  uint64_t a[6] = {v[0], v[1], v[2], v[3], 0, 0}, m[6];
  curve->fr.to_mont(m, a);
  return {m[0], m[1], m[2], m[3]};
}

inline G1 G1::Mul(const Zr& s) const { return curve->MultiScalarMul({*this}, {s}); }
inline G1 G1::Mul2(const Zr& e, const G1& Q, const Zr& f) const { return curve->MultiScalarMul({*this, Q}, {e, f}); }
inline void G1::Add(const G1& o) {
  Bytes both = raw;
  both.insert(both.end(), o.raw.begin(), o.raw.end());
  check(mlhip_g1_sum(curve->id, both.data(), 2, raw.data()));
}
inline void G1::Neg() {
  if (IsInfinity()) return;
  const Mod& fp = curve->fp;
  uint64_t y[6] = {0}, zero[6] = {0};
  memcpy(y, raw.data() + curve->fp_bytes, curve->fp_bytes);
  fp.sub(y, zero, y);
  memcpy(raw.data() + curve->fp_bytes, y, curve->fp_bytes);
}
inline void G1::Sub(const G1& o) {
  G1 t = o;
  t.Neg();
  Add(t);
}
inline Bytes G1::ToBytes() const {
  const Mod& fp = curve->fp;
  size_t n = curve->fp_bytes;
  Bytes out(2 * n, 0);
  if (IsInfinity()) {
    out[0] |= 0x40;
    return out;
  }
  for (int c = 0; c < 2; c++) {
    uint64_t m[6] = {0}, a[6];
    memcpy(m, raw.data() + c * n, n);
    fp.from_mont(a, m);
    Bytes b = be_bytes(a, fp.n);
    memcpy(out.data() + c * n, b.data(), n);
  }
  return out;
}
inline Bytes G2::ToBytes() const {
  Bytes out(4 * curve->fp_bytes);
  check(mlhip_g2_to_bytes(curve->id, raw.data(), 1, 0, out.data()));
  return out;
}
inline Bytes G2::Compressed() const {
  Bytes out(2 * curve->fp_bytes);
  check(mlhip_g2_to_bytes(curve->id, raw.data(), 1, 1, out.data()));
  return out;
}

inline Bytes G1::Compressed() const {
  const Mod& fp = curve->fp;
  size_t n = curve->fp_bytes;
  Bytes out(n, 0);
  if (IsInfinity()) {
    out[0] = curve->zcash_flags ? 0xC0 : 0x40;
    return out;
  }
  uint64_t m[6] = {0}, x[6], y[6];
  memcpy(m, raw.data(), n);
  fp.from_mont(x, m);
  memset(m, 0, sizeof(m));
  memcpy(m, raw.data() + n, n);
  fp.from_mont(y, m);
  out = be_bytes(x, fp.n);
  bool upper = fp.is_upper_half(y);
  if (curve->zcash_flags)
    out[0] |= 0x80 | (upper ? 0x20 : 0);
  else
    out[0] |= upper ? 0xC0 : 0x80;
  return out;
}

inline G2 G2::Mul(const Zr& s) const { return curve->MultiScalarMulG2({*this}, {s}); }
inline void G2::Add(const G2& o) {
  Bytes both = raw;
  both.insert(both.end(), o.raw.begin(), o.raw.end());
  check(mlhip_g2_sum(curve->id, both.data(), 2, raw.data()));
}

inline void Gt::Mul(const Gt& o) {
  Bytes out(curve->gt_bytes);
  check(mlhip_gt_mul(curve->id, raw.data(), o.raw.data(), 1, out.data()));
  raw = out;
}
inline Gt Gt::Exp(const Zr& x) const {
  Gt out = curve->new_gt();
  auto l = x.abi_limbs();
  check(mlhip_gt_exp(curve->id, raw.data(), l.data(), curve->scalars_mont ? 1 : 0, 1, out.raw.data()));
  return out;
}
inline bool Gt::IsUnity() const {
  uint64_t one[6] = {1, 0, 0, 0, 0, 0}, m[6];
  curve->fp.to_mont(m, one);
  if (memcmp(raw.data(), m, curve->fp_bytes) != 0) return false;
  for (size_t i = curve->fp_bytes; i < raw.size(); i++)
    if (raw[i]) return false;
  return true;
}
inline Bytes Gt::ToBytes() const {
  const Mod& fp = curve->fp;
  size_t n = curve->fp_bytes;
  Bytes out;
  for (int i = 11; i >= 0; i--) {
    uint64_t m[6] = {0}, a[6];
    memcpy(m, raw.data() + i * n, n);
    fp.from_mont(a, m);
    Bytes b = be_bytes(a, fp.n);
    out.insert(out.end(), b.begin(), b.end());
  }
  return out;
}

}  // namespace mlhip_driver
