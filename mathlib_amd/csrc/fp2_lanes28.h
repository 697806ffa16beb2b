// fp2_lanes28.h -- Fp2 elements over lane pairs in the CARRY-FREE form (fp28.h): the element type of the batched
// pairing kernels (pairing_kernels.h: k_pairing_lp28; BLS12-381 first, BLS12-377 and BN254 since round 3).
//
// Same layout as fp2_lanes.h -- lane 2i owns c0 and lane 2i+1 owns c1 of every Fp2 value, one pairing per lane pair --
// but each component is an Fp28 (14 signed 28-bit limbs, Montgomery R28 = 2^392) instead of 12 saturated 32-bit limbs:
//   * a limb product is ONE v_mad_i64_i32 (no v_addc on a third accumulator word): an Fp2 product (one fused dual
//     product per lane) costs 602 multiplier instructions and ~70 others instead of 432 v_mad + 432 v_addc + ~200;
//   * add / sub / neg / conj / small multiples are 14 independent 32-bit operations with no carry chain and no
//     conditional subtraction (a saturated fp_add is 12 v_addc + 12 v_subb + 12 selects);
//   * the price is bookkeeping: a value carries a WEIGHT w (it is a sum / difference of w normalized values,
//     |limb| < w 2^28), storage needs w <= 8, a dual product needs 2 w_a w_b <= 8, a single product w_a w_b <= 8, the
//     one-product square w = 1 (fp28.h).  The tower / pairing formulas (tower.h, pairing.h) call fp2_norm() -- one
//     carry propagation, ~40 full-rate instructions -- where a weight would otherwise exceed its budget; for the
//     saturated element types fp2_norm is a no-op.
// A second budget is the VALUE: carry propagation does not reduce mod p, so a value also carries a bound vb (|value| <
// 1.2 vb p; a Montgomery product resets it to 1 whenever sum vb_x vb_y <= 280, because R28 = 2^11 p).  Sums of products
// stay small by themselves; only linear terms carried from one loop iteration to the next (the -2 a / +2 a terms of the
// cyclotomic squarings) would double per iteration -- there fp2_reduce() subtracts round(value / p) p, a float
// estimate from the top limb plus one multiply-and-propagate pass (~1.7 x the cost of fp2_norm).
// The weights do not depend on the data (the formulas are straight-line), so the budget is verified once and for all on
// the CPU: Fp2H28 below is the host model of one lane pair (both components + the weight), every operation checks its
// precondition, and tests/test_host_math.py runs the whole pairing through it against the oracle.
// u^2 = -1 with xi = m + u (BLS12-381: m = 1; BN254: m = 9, 10 limbs) and u^2 = -5 with xi = u (BLS12-377: every product operand
// carry-propagated first, see lp28_mul) are implemented.
#pragma once
#include "fp28.h"
#include "fp2_lanes.h"

#if !defined(__HIPCC__)
#include <cstdio>
#include <cstdlib>
#include <execinfo.h>
#endif

namespace mlhip {

// ---- device element: this lane's component --------------------------------------------------------------------
template <class C>
struct Fp2L28 {
  Fp28<C> v;
  static constexpr int LANES = 1;
  MLHIP_HD Fp28<C>& at(int) { return v; }
  MLHIP_HD const Fp28<C>& at(int) const { return v; }
  MLHIP_HD static bool hi(int) { return lane_is_hi(); }
  MLHIP_HD int w() const { return 1; }
  MLHIP_HD void set_w(int) {}
  MLHIP_HD int vb() const { return 1; }
  MLHIP_HD void set_vb(int) {}
  MLHIP_HD static void require(bool, const char*, int, int) {}
};
template <class C>
MLHIP_HD void lp28_partner(Fp2L28<C>& r, const Fp2L28<C>& a) {
#pragma unroll
  for (int i = 0; i < C::N28; i++) r.v.l[i] = (int32_t)pair_xchg_u32((uint32_t)a.v.l[i]);
}
// the even (real) lane's value on both lanes of the pair: quad_perm [0,0,2,2]
template <class C>
MLHIP_HD void lp28_real_on_both(Fp2L28<C>& r, const Fp2L28<C>& a) {
#pragma unroll
  for (int i = 0; i < C::N28; i++) {
#if defined(__HIP_DEVICE_COMPILE__)
    r.v.l[i] = __builtin_amdgcn_update_dpp(0, a.v.l[i], 0xA0, 0xF, 0xF, true);
#else
    r.v.l[i] = a.v.l[i];
#endif
  }
}
template <class C>
MLHIP_HD bool lp28_both(const Fp2L28<C>&, const bool (&b)[1]) {
  const uint32_t z = b[0] ? 1u : 0u;
  return (z & pair_xchg_u32(z)) != 0;
}

#if !defined(__HIPCC__)
// ---- host model of one lane pair (g++ test build only): both components and the weight ------------------------
template <class C>
struct Fp2H28 {
  Fp28<C> c[2];
  int wt = 1, vbound = 1;
  static constexpr int LANES = 2;
  Fp28<C>& at(int i) { return c[i]; }
  const Fp28<C>& at(int i) const { return c[i]; }
  static bool hi(int i) { return i == 1; }
  int w() const { return wt; }
  void set_w(int x) { wt = x; }
  int vb() const { return vbound; }
  void set_vb(int x) {
    require(x <= 512, "a stored value (value bound)", x, 0);
    vbound = x;
  }
  static void require(bool ok, const char* what, int wa, int wb) {
    if (!ok) {
      fprintf(stderr, "fp2_lanes28: weight budget exceeded in %s (w_a = %d, w_b = %d)\n", what, wa, wb);
      void* frames[24];
      backtrace_symbols_fd(frames, backtrace(frames, 24), 2);  // which formula: pipe through c++filt
      abort();
    }
  }
};
template <class C>
inline void lp28_partner(Fp2H28<C>& r, const Fp2H28<C>& a) {
  const Fp28<C> t0 = a.c[0], t1 = a.c[1];
  r.c[0] = t1;
  r.c[1] = t0;
  r.wt = a.wt;
  r.vbound = a.vbound;
}
template <class C>
inline void lp28_real_on_both(Fp2H28<C>& r, const Fp2H28<C>& a) {
  const Fp28<C> t0 = a.c[0];
  r.c[0] = t0;
  r.c[1] = t0;
  r.wt = a.wt;
  r.vbound = a.vbound;
}
template <class C>
inline bool lp28_both(const Fp2H28<C>&, const bool (&b)[2]) {
  return b[0] && b[1];
}
#endif

constexpr int LP28_MAXW = 8;    // |limb| < w 2^28 must fit an int32
constexpr int LP28_MAXU = 280;  // sum of vb_x vb_y over the products of one reduction: the result is then within 1.2 p

// ---- generic bodies over either element -------------------------------------------------------------------------
template <class C, class E>
MLHIP_HD void lp28_zero(E& r) {
  for (int i = 0; i < E::LANES; i++) fp28_zero<C>(r.at(i));
  r.set_w(1);
  r.set_vb(1);
}
template <class C, class E>
MLHIP_HD void lp28_one(E& r) {
  for (int i = 0; i < E::LANES; i++) {
    Fp28<C> one, zero;
    fp28_from_const<C>(one, C::ONE28);
    fp28_zero<C>(zero);
    fp28_select<C>(r.at(i), E::hi(i), zero, one);
  }
  r.set_w(1);
  r.set_vb(1);
}
template <class C, class E>
MLHIP_HD void lp28_add(E& r, const E& a, const E& b) {
  const int w = a.w() + b.w();
  E::require(w <= LP28_MAXW, "fp2_add / fp2_sub", a.w(), b.w());
  const int v = a.vb() + b.vb();
  for (int i = 0; i < E::LANES; i++) fp28_add<C>(r.at(i), a.at(i), b.at(i));
  r.set_w(w);
  r.set_vb(v);
}
template <class C, class E>
MLHIP_HD void lp28_sub(E& r, const E& a, const E& b) {
  const int w = a.w() + b.w();
  E::require(w <= LP28_MAXW, "fp2_add / fp2_sub", a.w(), b.w());
  const int v = a.vb() + b.vb();
  for (int i = 0; i < E::LANES; i++) fp28_sub<C>(r.at(i), a.at(i), b.at(i));
  r.set_w(w);
  r.set_vb(v);
}
template <class C, class E>
MLHIP_HD void lp28_neg(E& r, const E& a) {
  for (int i = 0; i < E::LANES; i++) fp28_neg<C>(r.at(i), a.at(i));
  r.set_w(a.w());
  r.set_vb(a.vb());
}
template <class C, class E>
MLHIP_HD void lp28_conj(E& r, const E& a) {
  for (int i = 0; i < E::LANES; i++) {
    Fp28<C> n;
    fp28_neg<C>(n, a.at(i));
    fp28_select<C>(r.at(i), E::hi(i), n, a.at(i));
  }
  r.set_w(a.w());
  r.set_vb(a.vb());
}
template <class C, class E>
MLHIP_HD void lp28_select(E& r, bool c, const E& a, const E& b) {
  const int w = a.w() > b.w() ? a.w() : b.w(), v = a.vb() > b.vb() ? a.vb() : b.vb();
  for (int i = 0; i < E::LANES; i++) fp28_select<C>(r.at(i), c, a.at(i), b.at(i));
  r.set_w(w);
  r.set_vb(v);
}
template <class C, class E>
MLHIP_HD void lp28_norm(E& r) {
  for (int i = 0; i < E::LANES; i++) fp28_normalize<C>(r.at(i), r.at(i));
  r.set_w(1);
}
template <class C, class E>
MLHIP_HD void lp28_reduce(E& r) {
  for (int i = 0; i < E::LANES; i++) fp28_reduce<C>(r.at(i), r.at(i));
  r.set_w(1);
  r.set_vb(1);
}
// k a for a small positive k; for k > 2 carry-propagated in the same pass (64-bit intermediates): normalized result
template <class C, class E>
MLHIP_HD void lp28_mul_small(E& r, const E& a, int k) {
  E::require(k >= 1 && k <= 64, "fp2_mul_small", a.w(), k);
  if (k <= 2) {  // doubling stays lazy (the branch depends on k only: the host model and the device take the same one)
    E::require(k * a.w() <= LP28_MAXW, "fp2_mul_small", a.w(), k);
    for (int i = 0; i < E::LANES; i++)
#pragma unroll
      for (int j = 0; j < C::N28; j++) r.at(i).l[j] = a.at(i).l[j] * k;
    r.set_w(k * a.w());
    r.set_vb(k * a.vb());
    return;
  }
  const int v = k * a.vb();
  for (int i = 0; i < E::LANES; i++) {
    int64_t c = 0;
#pragma unroll
    for (int j = 0; j < C::N28 - 1; j++) {
      const int64_t v = (int64_t)a.at(i).l[j] * k + c;
      r.at(i).l[j] = (int32_t)((uint32_t)v & MASK28);
      c = v >> 28;
    }
    r.at(i).l[C::N28 - 1] = (int32_t)((int64_t)a.at(i).l[C::N28 - 1] * k + c);
  }
  r.set_w(1);
  r.set_vb(v);
}
// (a + (a odd ? p : 0)) / 2, limb-wise: bit 0 of limb j+1 moves to bit 27 of limb j.  Needs limbs that are integers of
// the right parity only -- any weight; the weight does not grow.
template <class C, class E>
MLHIP_HD void lp28_halve(E& r, const E& a) {
  E::require(a.w() + 1 <= LP28_MAXW, "fp2_halve", a.w(), 0);
  for (int i = 0; i < E::LANES; i++) {
    const int32_t odd = a.at(i).l[0] & 1;
    int32_t t[C::N28];
#pragma unroll
    for (int j = 0; j < C::N28; j++) t[j] = a.at(i).l[j] + (odd ? C::P28[j] : 0);
#pragma unroll
    for (int j = 0; j < C::N28 - 1; j++) r.at(i).l[j] = (t[j] >> 1) + ((t[j + 1] & 1) << 27);
    r.at(i).l[C::N28 - 1] = t[C::N28 - 1] >> 1;
  }
  const int v = (a.vb() + 1) / 2 + 1;
  r.set_w((a.w() + 1) / 2 + 1);  // |limb| < ((w + 1) 2^28) / 2 + 2^27
  r.set_vb(v);
}

// (a0 + a1 u)(b0 + b1 u), u^2 = BETA = -k: one fused dual product per lane
//   lane c0: a0 b0 + a1 (-k b1)        lane c1: a1 b0 + a0 b1
// k = 1 (BN254, BLS12-381): operands of weight w_a w_b <= 4.  k = 5 (BLS12-377): the c0 lane's product has weight
// (1 + k) w_a w_b, so both operands are carry-propagated first, whatever they were -- the weights are not known on the
// device, and a propagation is ~40 full-rate instructions beside a 616-instruction product.
template <class C, class E>
MLHIP_HD void lp28_mul(E& r, const E& a_in, const E& b_in) {
  constexpr int K = -C::BETA;
  E a = a_in, b = b_in;
  if constexpr (K != 1) {
    E::require(a.w() <= LP28_MAXW && b.w() <= LP28_MAXW, "fp2_mul", a.w(), b.w());
    lp28_norm<C>(a);
    lp28_norm<C>(b);
  }
  E::require((1 + K) * a.w() * b.w() <= 8, "fp2_mul", a.w(), b.w());
  E::require((1 + K) * a.vb() * b.vb() <= LP28_MAXU, "fp2_mul (value bound)", a.vb(), b.vb());
  //   both lanes: own a * b0 + partner's a * (+-b1) -- b0 arrives by one broadcast (no select), the sign of b1 by one
  //   negation of the exchanged value and one select
  E ax, bx, b0, o;
  lp28_partner<C>(ax, a);
  lp28_partner<C>(bx, b);
  lp28_real_on_both<C>(b0, b);
  for (int i = 0; i < E::LANES; i++) {
    Fp28<C> kb, nb, y2;
    fp28_times_k<C>(kb, bx.at(i));
    fp28_neg<C>(nb, kb);
    fp28_select<C>(y2, E::hi(i), b.at(i), nb);
    fp28_mul2<C>(o.at(i), a.at(i), b0.at(i), ax.at(i), y2);
  }
  o.set_w(1);
  o.set_vb(1);
  r = o;
}
// u^2 = -1: c0 = (a0 + a1)(a0 - a1), c1 = (2 a1) a0 -- one single product per lane; the operands have twice the weight of a.
// u^2 = -k: c0 = (a0 + a1)(a0 - k a1) + (k - 1) a0 a1, c1 = 2 a0 a1 -- still one single product per lane (the c1 lane's
// a0 a1 crosses to the c0 lane afterwards); both factors and the result are carry-propagated.
template <class C, class E>
MLHIP_HD void lp28_sqr(E& r, const E& a_in) {
  constexpr int K = -C::BETA;
  if constexpr (K == 1) {
    const E& a = a_in;
    E::require(4 * a.w() * a.w() <= 8, "fp2_sqr", a.w(), a.w());
    E::require(4 * a.vb() * a.vb() <= LP28_MAXU, "fp2_sqr (value bound)", a.vb(), a.vb());
    E ax, o;
    lp28_partner<C>(ax, a);
    for (int i = 0; i < E::LANES; i++) {
      const bool hi = E::hi(i);
      Fp28<C> t, d, x, y;
      fp28_select<C>(t, hi, a.at(i), ax.at(i));
      fp28_add<C>(x, a.at(i), t);  // a0 + a1 | 2 a1
      fp28_sub<C>(d, a.at(i), ax.at(i));
      fp28_select<C>(y, hi, ax.at(i), d);  // a0 - a1 | a0
      fp28_mul<C>(o.at(i), x, y);
    }
    o.set_w(1);
    o.set_vb(1);
    r = o;
  } else {
    E a = a_in;
    E::require(a.w() <= LP28_MAXW, "fp2_sqr", a.w(), a.w());
    lp28_reduce<C>(a);  // (a0 + a1)(a0 - k a1): 2 (1 + k) vb^2 -- the operand is brought below p, not just carry-propagated
    E::require(2 * (1 + K) * a.vb() * a.vb() <= LP28_MAXU, "fp2_sqr (value bound)", a.vb(), a.vb());
    E ax, o, ox;
    lp28_partner<C>(ax, a);
    for (int i = 0; i < E::LANES; i++) {
      const bool hi = E::hi(i);
      Fp28<C> kp, s, d, x, y;
      fp28_times_k<C>(kp, ax.at(i));
      fp28_add<C>(s, a.at(i), ax.at(i));  // a0 + a1 (c0 lane)
      fp28_sub<C>(d, a.at(i), kp);        // a0 - k a1 (c0 lane)
      fp28_normalize<C>(s, s);
      fp28_normalize<C>(d, d);
      fp28_select<C>(x, hi, a.at(i), s);   // a0 + a1 | a1
      fp28_select<C>(y, hi, ax.at(i), d);  // a0 - k a1 | a0
      fp28_mul<C>(o.at(i), x, y);          // t | a0 a1
    }
    o.set_w(1);
    o.set_vb(1);
    lp28_partner<C>(ox, o);
    for (int i = 0; i < E::LANES; i++) {
      const bool hi = E::hi(i);
      Fp28<C> v4, c0, c1;
#pragma unroll
      for (int j = 0; j < C::N28; j++) v4.l[j] = ox.at(i).l[j] * (K - 1);  // (k - 1) a0 a1 on the c0 lane
      fp28_add<C>(c0, o.at(i), v4);
      fp28_add<C>(c1, o.at(i), o.at(i));
      fp28_select<C>(o.at(i), hi, c1, c0);
      fp28_reduce<C>(o.at(i), o.at(i));  // |c0| <= t + (k - 1) |a0 a1|, k products: back below p like every other product
    }
    o.set_w(1);
    o.set_vb(1);
    r = o;
  }
}
// a * k, k in Fp (normalized carry-free form)
template <class C, class E>
MLHIP_HD void lp28_mul_fp(E& r, const E& a, const Fp28<C>& k) {
  E::require(a.w() <= 8, "fp2_mul_fp", a.w(), 1);
  E::require(a.vb() <= LP28_MAXU, "fp2_mul_fp (value bound)", a.vb(), 1);
  for (int i = 0; i < E::LANES; i++) fp28_mul<C>(r.at(i), a.at(i), k);
  r.set_w(1);
  r.set_vb(1);
}
// xi a.  xi = 1 + u, u^2 = -1 (BLS12-381): c0 = a0 - a1, c1 = a0 + a1.  xi = u, u^2 = -k (BLS12-377): c0 = -k a1, c1 = a0
template <class C, class E>
MLHIP_HD void lp28_mul_xi(E& r, const E& a) {
  constexpr int K = -C::BETA;
  static_assert((C::XI0 >= 1 && C::XI1 == 1 && K == 1) || (C::XI0 == 0 && C::XI1 == 1),
                "carry-free lane pairs: xi = m + u (u^2 = -1) or xi = u");
  E ax, o;
  if constexpr (C::XI0 > 1) {
    // xi = m + u (BN254: m = 9): c0 = m a0 - a1, c1 = m a1 + a0 -- multiplied and carry-propagated in one pass (64-bit
    // intermediates), then reduced mod p like the other xi-multiples
    E an = a;
    E::require(an.w() <= LP28_MAXW, "fp2_mul_xi", an.w(), 0);
    lp28_norm<C>(an);
    lp28_partner<C>(ax, an);
    for (int i = 0; i < E::LANES; i++) {
      Fp28<C> n, t;
      fp28_neg<C>(n, ax.at(i));
      fp28_select<C>(t, E::hi(i), ax.at(i), n);
      int64_t c = 0;
#pragma unroll
      for (int j = 0; j < C::N28 - 1; j++) {
        const int64_t v = (int64_t)an.at(i).l[j] * C::XI0 + t.l[j] + c;
        o.at(i).l[j] = (int32_t)((uint32_t)v & MASK28);
        c = v >> 28;
      }
      o.at(i).l[C::N28 - 1] = (int32_t)((int64_t)an.at(i).l[C::N28 - 1] * C::XI0 + t.l[C::N28 - 1] + c);
    }
    o.set_w(1);
    o.set_vb((C::XI0 + 1) * an.vb());
    lp28_reduce<C>(o);
  } else if constexpr (C::XI0 == 1) {
    lp28_partner<C>(ax, a);
    E::require(2 * a.w() <= LP28_MAXW, "fp2_mul_xi", a.w(), 0);
    for (int i = 0; i < E::LANES; i++) {
      Fp28<C> n, t;
      fp28_neg<C>(n, ax.at(i));
      fp28_select<C>(t, E::hi(i), ax.at(i), n);
      fp28_add<C>(o.at(i), a.at(i), t);
    }
    o.set_w(2 * a.w());
    o.set_vb(2 * a.vb());
  } else {
    E an = a;  // the c0 lane's -k a1 multiplies the weight by k: carry-propagated first, whatever it was
    E::require(an.w() <= LP28_MAXW, "fp2_mul_xi", an.w(), 0);
    lp28_norm<C>(an);
    lp28_partner<C>(ax, an);
    for (int i = 0; i < E::LANES; i++) {
      Fp28<C> kp, n;
      fp28_times_k<C>(kp, ax.at(i));
      fp28_neg<C>(n, kp);
      fp28_select<C>(o.at(i), E::hi(i), ax.at(i), n);
    }
    o.set_w(K);
    o.set_vb(K * an.vb());
    // ... and reduced mod p afterwards: xi-multiples are added to products all over the towers, and a k-fold value bound
    // would run their sums out of the product budget (sum vb_x vb_y <= 280 with the factor 1 + k on the c0 lane)
    lp28_reduce<C>(o);
  }
  r = o;
}
// both components congruent to 0 mod p (exact, any weight <= 8)
template <class C, class E>
MLHIP_HD bool lp28_is_zero(const E& a) {
  E::require(a.vb() <= 6, "fp2_is_zero (value bound)", a.vb(), 0);  // fp28_is_zero_exact: |value| < 8 p
  bool z[E::LANES];
  for (int i = 0; i < E::LANES; i++) z[i] = fp28_is_zero_exact<C>(a.at(i));
  return lp28_both<C>(a, z);
}
// constants held in the boundary form (curve_constants.h): converted on the fly, one product
template <class C, class E>
MLHIP_HD void lp28_from_const(E& r, const uint32_t (&k)[2][C::N]) {
  for (int i = 0; i < E::LANES; i++) {
    Fp<C> sel;
#pragma unroll
    for (int j = 0; j < C::N; j++) sel.l[j] = E::hi(i) ? k[1][j] : k[0][j];
    fp28_from_fp<C>(r.at(i), sel);
  }
  r.set_w(1);
  r.set_vb(1);
}
template <class C, class E>
MLHIP_HD void lp28_mul_by_real_const(E& r, const E& a, const uint32_t (&k)[2][C::N]) {
  Fp<C> kr;
  fp_from_const<C>(kr, k[0]);
  Fp28<C> k28;
  fp28_from_fp<C>(k28, kr);
  lp28_mul_fp<C>(r, a, k28);
}
// 1 / (a0 + a1 u) = (a0 - a1 u) / (a0^2 + k a1^2), u^2 = -k: the norm goes through the boundary form for the divsteps inversion
template <class C, class E>
MLHIP_HD void lp28_inv(E& r, const E& a) {
  constexpr int K = -C::BETA;
  E::require(a.w() * a.w() <= 8, "fp2_inv", a.w(), a.w());
  E::require(a.vb() * a.vb() <= LP28_MAXU, "fp2_inv (value bound)", a.vb(), a.vb());
  E sq, sqx, o;
  for (int i = 0; i < E::LANES; i++) fp28_mul<C>(sq.at(i), a.at(i), a.at(i));
  sq.set_w(1);
  sq.set_vb(1);
  lp28_partner<C>(sqx, sq);
  for (int i = 0; i < E::LANES; i++) {
    Fp28<C> n, ni28, t, nt, k0, k1, im;
    fp28_times_k<C>(k0, sq.at(i));
    fp28_times_k<C>(k1, sqx.at(i));
    fp28_select<C>(im, E::hi(i), k0, k1);                         // k a1^2 on both lanes
    fp28_select<C>(n, E::hi(i), sqx.at(i), sq.at(i));             // a0^2 on both lanes
    fp28_add<C>(n, n, im);                                        // a0^2 + k a1^2, weight 1 + k
    Fp<C> n32, ni32;
    fp28_to_fp<C>(n32, n);
    fp_inv<C>(ni32, n32);
    fp28_from_fp<C>(ni28, ni32);
    fp28_mul<C>(t, a.at(i), ni28);
    fp28_neg<C>(nt, t);
    fp28_select<C>(o.at(i), E::hi(i), nt, t);
  }
  o.set_w(1);
  o.set_vb(1);
  r = o;
}

// ---- the fp2_* overload set the towers are written against (tower.h, pairing.h) ----------------------------------
#define MLHIP_LP28_OVERLOADS(E)                                                                                       \
  template <class C>                                                                                                  \
  MLHIP_HD void fp2_zero(E<C>& r) { lp28_zero<C>(r); }                                                                \
  template <class C>                                                                                                  \
  MLHIP_HD void fp2_one(E<C>& r) { lp28_one<C>(r); }                                                                  \
  template <class C>                                                                                                  \
  MLHIP_HD bool fp2_is_zero(const E<C>& a) { return lp28_is_zero<C>(a); }                                             \
  template <class C>                                                                                                  \
  MLHIP_HD bool fp2_eq(const E<C>& a, const E<C>& b) {                                                                \
    E<C> d;                                                                                                           \
    lp28_sub<C>(d, a, b);                                                                                             \
    return lp28_is_zero<C>(d);                                                                                        \
  }                                                                                                                   \
  template <class C>                                                                                                  \
  MLHIP_HD void fp2_add(E<C>& r, const E<C>& a, const E<C>& b) { lp28_add<C>(r, a, b); }                              \
  template <class C>                                                                                                  \
  MLHIP_HD void fp2_sub(E<C>& r, const E<C>& a, const E<C>& b) { lp28_sub<C>(r, a, b); }                              \
  template <class C>                                                                                                  \
  MLHIP_HD void fp2_dbl(E<C>& r, const E<C>& a) { lp28_add<C>(r, a, a); }                                             \
  template <class C>                                                                                                  \
  MLHIP_HD void fp2_neg(E<C>& r, const E<C>& a) { lp28_neg<C>(r, a); }                                                \
  template <class C>                                                                                                  \
  MLHIP_HD void fp2_conj(E<C>& r, const E<C>& a) { lp28_conj<C>(r, a); }                                              \
  template <class C>                                                                                                  \
  MLHIP_HD void fp2_select(E<C>& r, bool c, const E<C>& a, const E<C>& b) { lp28_select<C>(r, c, a, b); }             \
  template <class C>                                                                                                  \
  MLHIP_HD void fp2_halve(E<C>& r, const E<C>& a) { lp28_halve<C>(r, a); }                                            \
  template <class C>                                                                                                  \
  MLHIP_HD void fp2_mul(E<C>& r, const E<C>& a, const E<C>& b) { lp28_mul<C>(r, a, b); }                              \
  template <class C>                                                                                                  \
  MLHIP_HD void fp2_sqr(E<C>& r, const E<C>& a) { lp28_sqr<C>(r, a); }                                                \
  template <class C>                                                                                                  \
  MLHIP_HD void fp2_mul_fp(E<C>& r, const E<C>& a, const Fp28<C>& k) { lp28_mul_fp<C>(r, a, k); }                     \
  template <class C>                                                                                                  \
  MLHIP_HD void fp2_mul_small(E<C>& r, const E<C>& a, int k) { lp28_mul_small<C>(r, a, k); }                          \
  template <class C>                                                                                                  \
  MLHIP_HD void fp2_mul_xi(E<C>& r, const E<C>& a) { lp28_mul_xi<C>(r, a); }                                          \
  template <class C>                                                                                                  \
  MLHIP_HD void fp2_inv(E<C>& r, const E<C>& a) { lp28_inv<C>(r, a); }                                                \
  template <class C>                                                                                                  \
  MLHIP_HD void fp2_from_const(E<C>& r, const uint32_t (&k)[2][C::N]) { lp28_from_const<C>(r, k); }                   \
  template <class C>                                                                                                  \
  MLHIP_HD void fp2_mul_by_real_const(E<C>& r, const E<C>& a, const uint32_t (&k)[2][C::N]) {                         \
    lp28_mul_by_real_const<C>(r, a, k);                                                                               \
  }                                                                                                                   \
  template <class C>                                                                                                  \
  MLHIP_HD void fp2_norm(E<C>& r) { lp28_norm<C>(r); }                                                                \
  template <class C>                                                                                                  \
  MLHIP_HD void fp2_reduce(E<C>& r) { lp28_reduce<C>(r); }                                                            \
  template <class C>                                                                                                  \
  MLHIP_HD int fp2_weight(const E<C>& a) { return a.w(); }

MLHIP_LP28_OVERLOADS(Fp2L28)
#if !defined(__HIPCC__)
MLHIP_LP28_OVERLOADS(Fp2H28)
#endif

}  // namespace mlhip
