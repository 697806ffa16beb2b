// modinv.h -- constant-time modular inversion by Bernstein-Yang divsteps ("safegcd"), signed 30-bit limbs.
//
// Fermat inversion costs ~570 Montgomery products (381 squarings + ~190 multiplications, each through an
// out-of-line fp_mul): 5 % of a pairing's instructions, 13 % of its final exponentiation.  The divsteps form does
// DIVSTEP_BATCHES batches of 30 steps: each batch runs 30 steps on the low words only (about 14 full-rate
// instructions per step) while collecting a 2x2 transition matrix of 31-bit entries, then applies the matrix to
// the full-width (f, g) and, modulo p, to (d, e) -- N30 limbs x 10 v_mad_i64_i32.  About 27 k instructions instead
// of ~390 k, no data-dependent branch or index (every lane of a wave follows the same path), no table.
//
// Algorithm: D. J. Bernstein, B.-Y. Yang, "Fast constant-time gcd computation and modular inversion" (2019), the
// delta = 1 divstep with the proven bound floor((49 d + 57)/17) steps for d-bit inputs; batching and the modular
// update of (d, e) as in the public description of that method (transition matrices scaled by 2^30, the division
// by 2^30 done exactly by adding the multiple of p that clears the low 30 bits).  The reference reaches inversion
// through gnark-crypto's fp.Element.Inverse (driver/gurvy/bls12381/bls12-381.go G1/G2 affine conversions, GT
// inverse in FExp); the result is a canonical field element either way.
// Invariants: d x = f (mod p), e x = g (mod p); f, g in (-2^(30 N30 - 1), ...) two's complement over N30 limbs of
// 30 bits (top limb signed); d, e in (-2p, p).
#pragma once
#include "fp.h"

namespace mlhip {

template <class C>
struct S30 {
  int32_t v[C::N30];
};

struct Trans30 {
  int32_t u, v, q, r;
};

// 30 divsteps on the low words; returns the new delta.  After the batch 2^30 (f', g') = t (f, g).
MLHIP_HD int32_t divsteps_30(int32_t delta, uint32_t f0, uint32_t g0, Trans30& t) {
  uint32_t u = 1, v = 0, q = 0, r = 1;
  uint32_t f = f0, g = g0;
#pragma unroll 5
  for (int i = 0; i < 30; i++) {
    const uint32_t c2 = (uint32_t)0 - (g & 1u);                             // g odd
    uint32_t c1 = (uint32_t)((int32_t)(0 - delta) >> 31) & c2;              // delta > 0 and g odd: swap
    const uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;  // +-f, +-u, +-v
    g += x & c2;
    q += y & c2;
    r += z & c2;
    delta = (int32_t)(((uint32_t)delta ^ c1) - c1) + 1;                      // swap: 1 - delta, else 1 + delta
    f += g & c1;
    u += q & c1;
    v += r & c1;
    g >>= 1;
    u <<= 1;
    v <<= 1;
  }
  t.u = (int32_t)u;
  t.v = (int32_t)v;
  t.q = (int32_t)q;
  t.r = (int32_t)r;
  return delta;
}

// (f, g) <- t (f, g) / 2^30 (exact)
template <class C>
MLHIP_HD void update_fg_30(S30<C>& f, S30<C>& g, const Trans30& t) {
  constexpr int L = C::N30;
  constexpr int32_t M30 = 0x3FFFFFFF;
  const int64_t u = t.u, v = t.v, q = t.q, r = t.r;
  int64_t cf = u * f.v[0] + v * g.v[0];
  int64_t cg = q * f.v[0] + r * g.v[0];
  cf >>= 30;
  cg >>= 30;
#pragma unroll
  for (int i = 1; i < L; i++) {
    cf += u * f.v[i] + v * g.v[i];
    cg += q * f.v[i] + r * g.v[i];
    f.v[i - 1] = (int32_t)cf & M30;
    g.v[i - 1] = (int32_t)cg & M30;
    cf >>= 30;
    cg >>= 30;
  }
  f.v[L - 1] = (int32_t)cf;
  g.v[L - 1] = (int32_t)cg;
}

// (d, e) <- t (d, e) / 2^30 mod p, kept in (-2p, p)
template <class C>
MLHIP_HD void update_de_30(S30<C>& d, S30<C>& e, const Trans30& t) {
  constexpr int L = C::N30;
  constexpr int32_t M30 = 0x3FFFFFFF;
  const int64_t u = t.u, v = t.v, q = t.q, r = t.r;
  const int32_t sd = d.v[L - 1] >> 31, se = e.v[L - 1] >> 31;  // sign masks
  int32_t md = (t.u & sd) + (t.v & se);
  int32_t me = (t.q & sd) + (t.r & se);
  int64_t cd = u * d.v[0] + v * e.v[0];
  int64_t ce = q * d.v[0] + r * e.v[0];
  // the multiples of p that clear the low 30 bits
  md -= (int32_t)((C::PINV30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
  me -= (int32_t)((C::PINV30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
  cd += (int64_t)C::P30[0] * md;
  ce += (int64_t)C::P30[0] * me;
  cd >>= 30;
  ce >>= 30;
#pragma unroll
  for (int i = 1; i < L; i++) {
    cd += u * d.v[i] + v * e.v[i];
    ce += q * d.v[i] + r * e.v[i];
    cd += (int64_t)C::P30[i] * md;
    ce += (int64_t)C::P30[i] * me;
    d.v[i - 1] = (int32_t)cd & M30;
    e.v[i - 1] = (int32_t)ce & M30;
    cd >>= 30;
    ce >>= 30;
  }
  d.v[L - 1] = (int32_t)cd;
  e.v[L - 1] = (int32_t)ce;
}

// r in (-2p, p) -> [0, p), negated first when sign < 0
template <class C>
MLHIP_HD void normalize_30(S30<C>& r, int32_t sign) {
  constexpr int L = C::N30;
  constexpr int32_t M30 = 0x3FFFFFFF;
  // add p when negative
  int32_t cond_add = r.v[L - 1] >> 31;
  const int32_t cond_negate = sign >> 31;
#pragma unroll
  for (int i = 0; i < L; i++) r.v[i] += C::P30[i] & cond_add;
  // conditionally negate
#pragma unroll
  for (int i = 0; i < L; i++) r.v[i] = (r.v[i] ^ cond_negate) - cond_negate;
  // carry propagation
#pragma unroll
  for (int i = 0; i < L - 1; i++) {
    r.v[i + 1] += r.v[i] >> 30;
    r.v[i] &= M30;
  }
  // now in (-p, p): add p once more when negative
  cond_add = r.v[L - 1] >> 31;
#pragma unroll
  for (int i = 0; i < L; i++) r.v[i] += C::P30[i] & cond_add;
#pragma unroll
  for (int i = 0; i < L - 1; i++) {
    r.v[i + 1] += r.v[i] >> 30;
    r.v[i] &= M30;
  }
}

// little-endian 32-bit limb string (canonical, < p) <-> 30-bit limbs
template <class C>
MLHIP_HD void s30_from_fp(S30<C>& r, const Fp<C>& a) {
#pragma unroll
  for (int j = 0; j < C::N30; j++) {
    const int bit = 30 * j, wi = bit >> 5, off = bit & 31;
    uint32_t x = 0;
    if (wi < C::N) x = a.l[wi] >> off;
    if (off > 2 && wi + 1 < C::N) x |= a.l[wi + 1] << (32 - off);
    r.v[j] = (int32_t)(x & 0x3FFFFFFFu);
  }
}
template <class C>
MLHIP_HD void s30_to_fp(Fp<C>& r, const S30<C>& a) {
#pragma unroll
  for (int i = 0; i < C::N; i++) {
    const int bit = 32 * i, j = bit / 30, off = bit - 30 * j;
    uint32_t x = 0;
    if (j < C::N30) x = (uint32_t)a.v[j] >> off;
    if (j + 1 < C::N30) x |= (uint32_t)a.v[j + 1] << (30 - off);
    if (off > 28 && j + 2 < C::N30) x |= (uint32_t)a.v[j + 2] << (60 - off);
    r.l[i] = x;
  }
}

// r = a^-1 in the Montgomery form of fp.h (a R -> a^-1 R); 0 -> 0
template <class C>
MLHIP_HD void fp_inv_divsteps(Fp<C>& r, const Fp<C>& a) {
  constexpr int L = C::N30;
  S30<C> d, e, f, g;
#pragma unroll
  for (int i = 0; i < L; i++) {
    d.v[i] = 0;
    e.v[i] = i == 0 ? 1 : 0;
    f.v[i] = C::P30[i];
  }
  s30_from_fp<C>(g, a);
  int32_t delta = 1;
#pragma unroll 1
  for (int it = 0; it < C::DIVSTEP_BATCHES; it++) {
    Trans30 t;
    delta = divsteps_30(delta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
    update_de_30<C>(d, e, t);
    update_fg_30<C>(f, g, t);
  }
  // g = 0, f = +-1 (or +-p when a = 0): d f is the inverse of the integer a R
  normalize_30<C>(d, f.v[L - 1]);
  Fp<C> y, k;
  s30_to_fp<C>(y, d);
  fp_from_const<C>(k, C::R3);
  fp_mul_i<C>(r, y, k);  // (a R)^-1 R^3 / R = a^-1 R
}

// the inversion every caller uses (towers, affine conversions, the host tail): one out-of-line copy per kernel
template <class C>
MLHIP_HD_NOINLINE void fp_inv(Fp<C>& r, const Fp<C>& a) {
  fp_inv_divsteps<C>(r, a);
}

}  // namespace mlhip
