// fp28.h -- carry-free Fp arithmetic for the bucket accumulation: N28 signed 28-bit limbs, Montgomery with
// R28 = 2^(28*N28).
//
// Why a second representation.  The boundary form (fp.h: N saturated 32-bit limbs, byte-identical to gnark's
// fp.Element) makes every limb product a v_mad_u64_u32 PLUS a v_addc on a third accumulator word, and every
// field addition a carry chain with a conditional subtraction.  With 28-bit limbs a whole column of the
// product-scanning multiplication (<= 2*N28 products of < 2^59) fits one 64-bit accumulator, so a limb product
// is a single v_mad_i64_i32 and nothing else; R28 is 2^11 (BLS12) / 2^26 (BN254) times larger than p, so a
// product of operands as large as 16p is already < 1.1p in magnitude and no conditional subtraction is ever
// needed; additions and subtractions are N28 independent 32-bit adds on signed limbs.  Nothing here needs
// inline asm -- hipcc selects v_mad_i64_i32 for  acc += (int64)a * b  -- so the same code runs on the host.
//
// Invariants (checked by tests/test_host_math.py against Python integers):
//   * a "normalized" value has limbs 0..N28-2 in [0, 2^28) and a small signed top limb; its value is in
//     (-0.2p, 1.2p).  Every fp28_mul / fp28_sqr / fp28_mul2 result and every fp28_normalize result is normalized.
//   * a value of weight w is a sum/difference of w normalized values: |limb| < w 2^28.
//   * fp28_mul(a, b) needs w_a w_b <= 8, fp28_mul2(a, b, c, d) needs w_a w_b + w_c w_d <= 8, fp28_sqr(a)
//     needs w_a <= 2  (column sums stay below 2^63).
// Not part of the C ABI: values are converted from / to the boundary form by fp28_from_fp / fp28_to_fp.
#pragma once
#include "fp.h"

namespace mlhip {

constexpr uint32_t MASK28 = 0x0FFFFFFFu;

template <class C>
struct Fp28 {
  int32_t l[C::N28];
};

template <class C>
MLHIP_HD void fp28_zero(Fp28<C>& r) {
#pragma unroll
  for (int i = 0; i < C::N28; i++) r.l[i] = 0;
}
template <class C>
MLHIP_HD void fp28_from_const(Fp28<C>& r, const int32_t (&k)[C::N28]) {
#pragma unroll
  for (int i = 0; i < C::N28; i++) r.l[i] = k[i];
}
template <class C>
MLHIP_HD void fp28_add(Fp28<C>& r, const Fp28<C>& a, const Fp28<C>& b) {
#pragma unroll
  for (int i = 0; i < C::N28; i++) r.l[i] = a.l[i] + b.l[i];
}
template <class C>
MLHIP_HD void fp28_sub(Fp28<C>& r, const Fp28<C>& a, const Fp28<C>& b) {
#pragma unroll
  for (int i = 0; i < C::N28; i++) r.l[i] = a.l[i] - b.l[i];
}
template <class C>
MLHIP_HD void fp28_neg(Fp28<C>& r, const Fp28<C>& a) {
#pragma unroll
  for (int i = 0; i < C::N28; i++) r.l[i] = -a.l[i];
}
template <class C>
MLHIP_HD void fp28_select(Fp28<C>& r, bool c, const Fp28<C>& a, const Fp28<C>& b) {
#pragma unroll
  for (int i = 0; i < C::N28; i++) r.l[i] = c ? a.l[i] : b.l[i];
}
// every limb zero (only meaningful for canonical values such as converted inputs)
template <class C>
MLHIP_HD bool fp28_all_zero(const Fp28<C>& a) {
  int32_t o = 0;
#pragma unroll
  for (int i = 0; i < C::N28; i++) o |= a.l[i];
  return o == 0;
}

// carry propagation: limbs 0..N28-2 into [0, 2^28), the (signed) rest into the top limb.  Value unchanged.
template <class C>
MLHIP_HD void fp28_normalize(Fp28<C>& r, const Fp28<C>& a) {
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < C::N28 - 1; i++) {
    int32_t v = a.l[i] + c;
    r.l[i] = (int32_t)((uint32_t)v & MASK28);
    c = v >> 28;  // arithmetic: floor
  }
  r.l[C::N28 - 1] = a.l[C::N28 - 1] + c;
}

// k x for the small positive k = -BETA of u^2 = BETA (5 for BLS12-377): shifts and adds, the weight grows k-fold
template <class C>
MLHIP_HD void fp28_times_k(Fp28<C>& r, const Fp28<C>& a) {
  constexpr int K = -C::BETA;
  static_assert(K == 1 || K == 5, "u^2 = -1 or -5");
#pragma unroll
  for (int i = 0; i < C::N28; i++) r.l[i] = K == 5 ? (int32_t)(((uint32_t)a.l[i] << 2) + (uint32_t)a.l[i]) : a.l[i];
}

// value -> value - round(value / p) p, carry-propagated: limbs normalized, |result| < 0.6 p.  The quotient comes from
// the top limb (value / 2^364 up to the weight) in single precision: exact to well within +-0.01.
template <class C>
MLHIP_HD void fp28_reduce(Fp28<C>& r, const Fp28<C>& a) {
  constexpr int L = C::N28;
  constexpr float inv_ptop = 1.0f / ((float)C::P28[L - 1] + (float)C::P28[L - 2] * (1.0f / 268435456.0f));
  const float qf = (float)a.l[L - 1] * inv_ptop;
  const int32_t q = (int32_t)(qf + (qf >= 0.0f ? 0.5f : -0.5f));
  int64_t c = 0;
#pragma unroll
  for (int i = 0; i < L - 1; i++) {
    const int64_t v = (int64_t)a.l[i] - (int64_t)q * C::P28[i] + c;
    r.l[i] = (int32_t)((uint32_t)v & MASK28);
    c = v >> 28;
  }
  r.l[L - 1] = (int32_t)((int64_t)a.l[L - 1] - (int64_t)q * C::P28[L - 1] + c);
}

// r = (a b + c d) / R28 mod p in one reduction; DUAL = false drops the second product.  SQR: b is ignored and
// the product a a is formed from the N28 (N28+1)/2 distinct limb products.
template <class C, bool DUAL, bool SQR>
MLHIP_HD void fp28_mont(Fp28<C>& r, const Fp28<C>& a, const Fp28<C>& b, const Fp28<C>& c, const Fp28<C>& d) {
  constexpr int L = C::N28;
  int64_t acc = 0;
  int32_t m[L];
  int32_t t[L];
  int32_t a2[L];
  if (SQR) {
#pragma unroll
    for (int i = 0; i < L; i++) a2[i] = a.l[i] + a.l[i];
  }
#pragma unroll
  for (int k = 0; k < 2 * L - 1; k++) {
    const int lo = k < L ? 0 : k - L + 1, hi = k < L ? k : L - 1;
    if (SQR) {
#pragma unroll
      for (int i = lo; i <= hi; i++) {
        const int j = k - i;
        if (i < j) acc += (int64_t)a.l[i] * a2[j];
        if (i == j) acc += (int64_t)a.l[i] * a.l[i];
      }
    } else {
#pragma unroll
      for (int i = lo; i <= hi; i++) acc += (int64_t)a.l[i] * b.l[k - i];
      if (DUAL) {
#pragma unroll
        for (int i = lo; i <= hi; i++) acc += (int64_t)c.l[i] * d.l[k - i];
      }
    }
#pragma unroll
    for (int i = lo; i <= hi; i++) {
      if (k < L && i == k) continue;  // m[k] is not known yet
      acc += (int64_t)m[i] * C::P28[k - i];
    }
    if (k < L) {
      m[k] = (int32_t)(((uint32_t)acc * C::PINV28) & MASK28);
      acc += (int64_t)m[k] * C::P28[0];  // low 28 bits are now zero
    } else {
      t[k - L] = (int32_t)((uint32_t)acc & MASK28);
    }
    acc >>= 28;  // arithmetic shift = floor division, also for negative columns
  }
  t[L - 1] = (int32_t)acc;
#pragma unroll
  for (int i = 0; i < L; i++) r.l[i] = t[i];
}

#if defined(__HIP_DEVICE_COMPILE__) && !defined(MLHIP_FP28_PORTABLE)
#include "fp28_comba.inc"  // same arithmetic, one accumulator chain per column (tools/gen_fp28_comba.py)
#define MLHIP_FP28_DEV(fn, ...) \
  if constexpr (C::N28 == 14) fn##14<C>(__VA_ARGS__); else fn##10<C>(__VA_ARGS__)
#endif

template <class C>
MLHIP_HD void fp28_mul(Fp28<C>& r, const Fp28<C>& a, const Fp28<C>& b) {
#ifdef MLHIP_FP28_DEV
  MLHIP_FP28_DEV(fp28_mul_dev, r, a, b);
#else
  fp28_mont<C, false, false>(r, a, b, a, b);
#endif
}
template <class C>
MLHIP_HD void fp28_sqr(Fp28<C>& r, const Fp28<C>& a) {
#ifdef MLHIP_FP28_DEV
  MLHIP_FP28_DEV(fp28_sqr_dev, r, a);
#else
  fp28_mont<C, false, true>(r, a, a, a, a);
#endif
}
template <class C>
MLHIP_HD void fp28_mul2(Fp28<C>& r, const Fp28<C>& a, const Fp28<C>& b, const Fp28<C>& c, const Fp28<C>& d) {
#ifdef MLHIP_FP28_DEV
  MLHIP_FP28_DEV(fp28_mul2_dev, r, a, b, c, d);
#else
  fp28_mont<C, true, false>(r, a, b, c, d);
#endif
}

// Fp2 product (u^2 = -1) with both components on ONE lane: Karatsuba over the components, the three limb products
// interleaved column by column so that the unreduced products never exist as values:
//     c0 = a0 b0 - a1 b1        c1 = (a0 + a1)(b0 + b1) - a0 b0 - a1 b1
// 3 limb products + 2 reductions instead of the 4 + 2 of two dual products.  All four inputs must be normalized (weight 1):
// a column of c1 is then below (4 + 1 + 1 + 1) N28 2^56 < 2^63.  The integers that are reduced are exactly those of
// fp28_mul2(a0, b0, -a1, b1) and fp28_mul2(a0, b1, a1, b0), so the results are bit-identical to the lane-pair product.
template <class C>
MLHIP_HD void fp28_k2mul_portable(Fp28<C>& r0, Fp28<C>& r1, const Fp28<C>& a0, const Fp28<C>& a1, const Fp28<C>& b0,
                                  const Fp28<C>& b1) {
  constexpr int L = C::N28;
  int32_t s[L], t[L], m0[L], m1[L], t0[L], t1[L];
#pragma unroll
  for (int i = 0; i < L; i++) {
    s[i] = a0.l[i] + a1.l[i];
    t[i] = b0.l[i] + b1.l[i];
  }
  int64_t c0 = 0, c1 = 0;
#pragma unroll
  for (int k = 0; k < 2 * L - 1; k++) {
    const int lo = k < L ? 0 : k - L + 1, hi = k < L ? k : L - 1;
    int64_t p0 = 0, p1 = 0;
#pragma unroll
    for (int i = lo; i <= hi; i++) {
      p0 += (int64_t)a0.l[i] * b0.l[k - i];
      p1 += (int64_t)a1.l[i] * b1.l[k - i];
      c1 += (int64_t)s[i] * t[k - i];
    }
#pragma unroll
    for (int i = lo; i <= hi; i++) {
      if (k < L && i == k) continue;
      c0 += (int64_t)m0[i] * C::P28[k - i];
      c1 += (int64_t)m1[i] * C::P28[k - i];
    }
    c0 += p0 - p1;
    c1 -= p0 + p1;
    if (k < L) {
      m0[k] = (int32_t)(((uint32_t)c0 * C::PINV28) & MASK28);
      c0 += (int64_t)m0[k] * C::P28[0];
      m1[k] = (int32_t)(((uint32_t)c1 * C::PINV28) & MASK28);
      c1 += (int64_t)m1[k] * C::P28[0];
    } else {
      t0[k - L] = (int32_t)((uint32_t)c0 & MASK28);
      t1[k - L] = (int32_t)((uint32_t)c1 & MASK28);
    }
    c0 >>= 28;
    c1 >>= 28;
  }
  t0[L - 1] = (int32_t)c0;
  t1[L - 1] = (int32_t)c1;
#pragma unroll
  for (int i = 0; i < L; i++) {
    r0.l[i] = t0[i];
    r1.l[i] = t1[i];
  }
}
template <class C>
MLHIP_HD void fp28_k2mul(Fp28<C>& r0, Fp28<C>& r1, const Fp28<C>& a0, const Fp28<C>& a1, const Fp28<C>& b0, const Fp28<C>& b1) {
#ifdef MLHIP_FP28_DEV
  MLHIP_FP28_DEV(fp28_k2mul_dev, r0, r1, a0, a1, b0, b1);
#else
  fp28_k2mul_portable<C>(r0, r1, a0, a1, b0, b1);
#endif
}

// ---- conversions to / from the boundary form (canonical, Montgomery R = 2^(32 N)) ---------------------------
// bits [28 j, 28 j + 28) of a little-endian 32-bit limb string
template <class C>
MLHIP_HD void fp28_repack_from32(Fp28<C>& r, const Fp<C>& a) {
#pragma unroll
  for (int j = 0; j < C::N28; j++) {
    const int bit = 28 * j, wi = bit >> 5, off = bit & 31;
    uint32_t v = 0;
    if (wi < C::N) v = a.l[wi] >> off;
    if (off > 4 && wi + 1 < C::N) v |= a.l[wi + 1] << (32 - off);
    r.l[j] = (int32_t)(v & MASK28);
  }
}
// inverse; `a` must have all limbs in [0, 2^28) and a value < 2^(32 N)
template <class C>
MLHIP_HD void fp28_repack_to32(Fp<C>& r, const Fp28<C>& a) {
#pragma unroll
  for (int i = 0; i < C::N; i++) {
    const int bit = 32 * i, j = bit / 28, off = bit - 28 * j;
    uint32_t v = 0;
    if (j < C::N28) v = (uint32_t)a.l[j] >> off;
    if (j + 1 < C::N28) v |= (uint32_t)a.l[j + 1] << (28 - off);
    if (off > 24 && j + 2 < C::N28) v |= (uint32_t)a.l[j + 2] << (56 - off);
    r.l[i] = v;
  }
}

// x R (canonical, 32-bit limbs)  ->  x R28 (normalized)
template <class C>
MLHIP_HD void fp28_from_fp(Fp28<C>& r, const Fp<C>& a) {
  Fp28<C> t, k;
  fp28_repack_from32<C>(t, a);
  fp28_from_const<C>(k, C::TO28);
  fp28_mul<C>(r, t, k);
}

// any value of weight <= 8 representing x R28  ->  x R canonical in 32-bit limbs
template <class C>
MLHIP_HD void fp28_to_fp(Fp<C>& r, const Fp28<C>& a) {
  constexpr int L = C::N28;
  Fp28<C> t, k;
  fp28_from_const<C>(k, C::FROM28);
  fp28_mul<C>(t, a, k);  // in (-0.2p, 1.2p), limbs 0..L-2 in [0, 2^28)
  // canonical representative: add p when negative, subtract p when >= p
  if (t.l[L - 1] < 0) {
    Fp28<C> pp;
    fp28_from_const<C>(pp, C::P28);
    fp28_add<C>(t, t, pp);
    fp28_normalize<C>(t, t);
  }
  Fp28<C> u;
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < L; i++) {
    int32_t v = t.l[i] - C::P28[i] + c;
    u.l[i] = i < L - 1 ? (int32_t)((uint32_t)v & MASK28) : v;
    c = i < L - 1 ? (v >> 28) : 0;
  }
  const bool ge = u.l[L - 1] >= 0;  // t - p >= 0
  fp28_select<C>(t, ge, u, t);
  fp28_repack_to32<C>(r, t);
}

// is the value (weight <= 8, |value| < 8p) congruent to 0 mod p?  Exact.  The fast path rejects on the low limb:
// v = k p  =>  k = v[0] p^-1 (mod 2^28) must be a small signed integer.
template <class C>
MLHIP_HD bool fp28_maybe_zero(const Fp28<C>& v) {
  uint32_t k = ((uint32_t)v.l[0] * C::PINVPOS28) & MASK28;
  return ((k + 8u) & MASK28) <= 16u;
}
template <class C>
MLHIP_HD bool fp28_is_zero_exact(const Fp28<C>& v) {
  uint32_t k = ((uint32_t)v.l[0] * C::PINVPOS28) & MASK28;
  if (((k + 8u) & MASK28) > 16u) return false;
  const int32_t ks = (int32_t)(k << 4) >> 4;  // sign-extend 28 bits
  int64_t c = 0;
  int64_t o = 0;
#pragma unroll
  for (int i = 0; i < C::N28; i++) {
    int64_t x = (int64_t)v.l[i] - (int64_t)ks * C::P28[i] + c;
    if (i < C::N28 - 1) {
      o |= x & MASK28;
      c = x >> 28;
    } else {
      o |= x;
    }
  }
  return o == 0;
}

}  // namespace mlhip
