// ec28_lp.h -- G2 bucket accumulation in the carry-free form (fp28.h) over lane pairs.
//
// Same idea as ec28.h for G1, combined with the lane-pair layout of fp2_lanes.h: lane 2i holds the real part
// and lane 2i+1 the imaginary part of every Fp2 coordinate, each as an Fp28.  An Fp2 product is ONE fused dual
// product per lane,
//     c0 = a0 b0 + (-k a1) b1        c1 = a1 b0 + a0 b1        (u^2 = -k)
// and an Fp2 square is one single product per lane: c0 = (a0 + a1)(a0 - a1), c1 = (2 a1) a0 for k = 1.  For k = 5
// (BLS12-377, round 3) the c0 lane's product weighs (1 + 5) w_a w_b: every product operand is normalized, the differences
// that feed products are reduced mod p rather than carry-propagated (lp28_settle), and the square is the two-term form.
// Weight rules (fp28.h): a dual product needs w_a w_b <= 4, the single-product square needs w_a = 1 -- so P and
// R are carry-propagated right after the subtraction, X3 and Y3 before they are stored: four propagations and
// 8 dual + 2 single products per mixed addition.  (Fusing Y3 = R (Q - X3) - Y1 PPP into one four-product reduction
// saves 4 % of the multiplies but holds eight prepared operands live: 73 spilled registers instead of 49 and +1.8 %
// kernel time measured -- DESIGN.md section 7.)
// Written over a backend (device: one Fp28 per lane + DPP; host: a 2-entry array) like ec_quad.h.
#pragma once
#include "ec.h"
#include "fp28.h"
#include "tower.h"

namespace mlhip {

template <class C>
struct PairHost {
  struct V {
    Fp28<C> v[2];
  };
  static void xchg(V& r, const V& a) {
    V t = a;
    r.v[0] = t.v[1];
    r.v[1] = t.v[0];
  }
  static void sel_hi(V& r, const V& hi_val, const V& lo_val) {
    r.v[0] = lo_val.v[0];
    r.v[1] = hi_val.v[1];
  }
  static void real_on_both(V& r, const V& a) {
    V t = a;
    r.v[0] = t.v[0];
    r.v[1] = t.v[0];
  }
  template <class FN>
  static void each(FN fn) {
    for (int i = 0; i < 2; i++) fn(i);
  }
  static Fp28<C>& at(V& x, int i) { return x.v[i]; }
  static const Fp28<C>& at(const V& x, int i) { return x.v[i]; }
  static bool both(const bool (&b)[2]) { return b[0] && b[1]; }
  static void gather(Fp2<C>& out, const Fp<C> (&own)[2]) {
    out.c0 = own[0];
    out.c1 = own[1];
  }
  static void scatter(Fp<C> (&own)[2], const Fp2<C>& in) {
    own[0] = in.c0;
    own[1] = in.c1;
  }
  static constexpr int LANES = 2;
};

#if defined(__HIPCC__)
template <class C>
struct PairDevice {
  typedef Fp28<C> V;
  static __device__ __forceinline__ uint32_t x1(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]; bound_ctrl: no "old" value to materialize
#else
    return v;
#endif
  }
  static __device__ __forceinline__ bool hi() { return (threadIdx.x & 1u) != 0; }
  static __device__ __forceinline__ void xchg(V& r, const V& a) {
#pragma unroll
    for (int i = 0; i < C::N28; i++) r.l[i] = (int32_t)x1((uint32_t)a.l[i]);
  }
  static __device__ __forceinline__ void sel_hi(V& r, const V& hi_val, const V& lo_val) {
    fp28_select<C>(r, hi(), hi_val, lo_val);
  }
  // the even (real) lane's value on both lanes: quad_perm [0,0,2,2]
  static __device__ __forceinline__ void real_on_both(V& r, const V& a) {
#pragma unroll
    for (int i = 0; i < C::N28; i++) {
#if defined(__HIP_DEVICE_COMPILE__)
      r.l[i] = __builtin_amdgcn_update_dpp(0, a.l[i], 0xA0, 0xF, 0xF, true);
#else
      r.l[i] = a.l[i];
#endif
    }
  }
  template <class FN>
  static __device__ __forceinline__ void each(FN fn) {
    fn(0);
  }
  static __device__ __forceinline__ Fp28<C>& at(V& x, int) { return x; }
  static __device__ __forceinline__ const Fp28<C>& at(const V& x, int) { return x; }
  static __device__ __forceinline__ bool both(const bool (&b)[1]) {
    uint32_t z = b[0] ? 1u : 0u;
    return (z & x1(z)) != 0;
  }
  static __device__ __forceinline__ void gather(Fp2<C>& out, const Fp<C> (&own)[1]) {
    Fp<C> other;
#pragma unroll
    for (int i = 0; i < C::N; i++) other.l[i] = x1(own[0].l[i]);
    fp_select<C>(out.c0, hi(), other, own[0]);
    fp_select<C>(out.c1, hi(), own[0], other);
  }
  static __device__ __forceinline__ void scatter(Fp<C> (&own)[1], const Fp2<C>& in) {
    fp_select<C>(own[0], hi(), in.c1, in.c0);
  }
  static constexpr int LANES = 1;
};
#endif

// lane-wise helpers over a backend value
#define MLHIP_LP28_EACH(B, ...) B::each([&](int li_) { __VA_ARGS__; })

template <class C, class B>
MLHIP_HD void lp28_sub(typename B::V& r, const typename B::V& a, const typename B::V& b) {
  MLHIP_LP28_EACH(B, fp28_sub<C>(B::at(r, li_), B::at(a, li_), B::at(b, li_)));
}
template <class C, class B>
MLHIP_HD void lp28_neg(typename B::V& r, const typename B::V& a) {
  MLHIP_LP28_EACH(B, fp28_neg<C>(B::at(r, li_), B::at(a, li_)));
}
template <class C, class B>
MLHIP_HD void lp28_normalize(typename B::V& r, const typename B::V& a) {
  MLHIP_LP28_EACH(B, fp28_normalize<C>(B::at(r, li_), B::at(a, li_)));
}
template <class C, class B>
MLHIP_HD void lp28_select(typename B::V& r, bool c, const typename B::V& a, const typename B::V& b) {
  MLHIP_LP28_EACH(B, fp28_select<C>(B::at(r, li_), c, B::at(a, li_), B::at(b, li_)));
}
template <class C, class B>
MLHIP_HD bool lp28_all_zero(const typename B::V& a) {
  bool z[B::LANES];
  MLHIP_LP28_EACH(B, z[li_] = fp28_all_zero<C>(B::at(a, li_)));
  return B::both(z);
}
template <class C, class B>
MLHIP_HD bool lp28_is_zero_exact(const typename B::V& a) {
  bool z[B::LANES];
  MLHIP_LP28_EACH(B, z[li_] = fp28_maybe_zero<C>(B::at(a, li_)));
  if (!B::both(z)) return false;
  MLHIP_LP28_EACH(B, z[li_] = fp28_is_zero_exact<C>(B::at(a, li_)));
  return B::both(z);
}

template <class C, class B>
MLHIP_HD void lp28_reduce_p(typename B::V& r, const typename B::V& a) {
  MLHIP_LP28_EACH(B, fp28_reduce<C>(B::at(r, li_), B::at(a, li_)));
}
// the carry propagation of a difference that feeds products: u^2 = -1 keeps the value (fp28_normalize); u^2 = -5, whose c0
// lane weighs a product (1 + 5)-fold, also brings it back below p (fp28_reduce: ~1.7 x the instructions)
template <class C, class B>
MLHIP_HD void lp28_settle(typename B::V& r, const typename B::V& a) {
  if constexpr (C::BETA == -1)
    lp28_normalize<C, B>(r, a);
  else
    lp28_reduce_p<C, B>(r, a);
}

// r = a b in Fp2 (u^2 = -k): one dual product per lane, c0 = a0 b0 + a1 (-k b1), c1 = a1 b0 + a0 b1.
// k = 1 needs w_a w_b <= 4; k = 5 (BLS12-377) needs normalized operands with |values| such that 6 vb_a vb_b <= 280
template <class C, class B>
MLHIP_HD void lp28_mul(typename B::V& r, const typename B::V& a, const typename B::V& b) {
  // both lanes: own a * b0 + partner's a * (+-k b1): b0 by one broadcast, the sign of b1 by one negation and one select
  typename B::V ax, bx, b0, kbx, nbx, y2;
  B::xchg(ax, a);
  B::xchg(bx, b);
  B::real_on_both(b0, b);
  MLHIP_LP28_EACH(B, fp28_times_k<C>(B::at(kbx, li_), B::at(bx, li_)));
  lp28_neg<C, B>(nbx, kbx);
  B::sel_hi(y2, b, nbx);  // c0: -k b1 | c1: b1 (own)
  MLHIP_LP28_EACH(B, fp28_mul2<C>(B::at(r, li_), B::at(a, li_), B::at(b0, li_), B::at(ax, li_), B::at(y2, li_)));
}

// r = a^2, a normalized (weight 1): one single product per lane.  u^2 = -1: c0 = (a0 + a1)(a0 - a1), c1 = (2 a1) a0.
// u^2 = -k: c0 = (a0 + a1)(a0 - k a1) + (k - 1) a0 a1, c1 = 2 a0 a1 -- the c1 lane's a0 a1 crosses to the c0 lane afterwards;
// `a` must be reduced (|value| < p), the result is reduced again.
template <class C, class B>
MLHIP_HD void lp28_sqr(typename B::V& r, const typename B::V& a) {
  constexpr int K = -C::BETA;
  if constexpr (K == 1) {
    typename B::V ax, t, d, x, y;
    B::xchg(ax, a);
    B::sel_hi(t, a, ax);
    MLHIP_LP28_EACH(B, fp28_add<C>(B::at(x, li_), B::at(a, li_), B::at(t, li_)));    // a0 + a1 | 2 a1
    MLHIP_LP28_EACH(B, fp28_sub<C>(B::at(d, li_), B::at(a, li_), B::at(ax, li_)));   // c0 lane: a0 - a1
    B::sel_hi(y, ax, d);
    MLHIP_LP28_EACH(B, fp28_mul<C>(B::at(r, li_), B::at(x, li_), B::at(y, li_)));
  } else {
    typename B::V ax, kax, s, d, x, y, o, ox, v4, c0, c1;
    B::xchg(ax, a);
    MLHIP_LP28_EACH(B, fp28_times_k<C>(B::at(kax, li_), B::at(ax, li_)));
    MLHIP_LP28_EACH(B, fp28_add<C>(B::at(s, li_), B::at(a, li_), B::at(ax, li_)));   // c0 lane: a0 + a1
    MLHIP_LP28_EACH(B, fp28_sub<C>(B::at(d, li_), B::at(a, li_), B::at(kax, li_)));  // c0 lane: a0 - k a1
    lp28_normalize<C, B>(s, s);
    lp28_normalize<C, B>(d, d);
    B::sel_hi(x, a, s);   // a0 + a1 | a1
    B::sel_hi(y, ax, d);  // a0 - k a1 | a0
    MLHIP_LP28_EACH(B, fp28_mul<C>(B::at(o, li_), B::at(x, li_), B::at(y, li_)));  // t | a0 a1
    B::xchg(ox, o);
    MLHIP_LP28_EACH(B, {
      for (int j = 0; j < C::N28; j++) B::at(v4, li_).l[j] = B::at(ox, li_).l[j] * (K - 1);
    });
    MLHIP_LP28_EACH(B, fp28_add<C>(B::at(c0, li_), B::at(o, li_), B::at(v4, li_)));
    MLHIP_LP28_EACH(B, fp28_add<C>(B::at(c1, li_), B::at(o, li_), B::at(o, li_)));
    B::sel_hi(o, c1, c0);
    lp28_reduce_p<C, B>(r, o);
  }
}

template <class V>
struct XYZZ28L {
  V x, y, zz, zzz;  // normalized
};
template <class V>
struct Affine28L {
  V x, y;
};

// exceptional cases (q = +-acc): every lane rebuilds the full Fp2 points in the boundary form, runs the one-lane
// formulas and keeps its own component
template <class C, class B>
MLHIP_HD_NOINLINE void xyzz28_lp_madd_exact(XYZZ28L<typename B::V>& acc, bool& inf, const Affine28L<typename B::V>& q) {
  typedef Fp2Field<C> F2;
  XYZZ<F2> a;
  Affine<F2> p;
  auto to2 = [](Fp2<C>& out, const typename B::V& v) {
    Fp<C> own[B::LANES];
    MLHIP_LP28_EACH(B, fp28_to_fp<C>(own[li_], B::at(v, li_)));
    B::gather(out, own);
  };
  auto from2 = [](typename B::V& v, const Fp2<C>& in) {
    Fp<C> own[B::LANES];
    B::scatter(own, in);
    MLHIP_LP28_EACH(B, fp28_from_fp<C>(B::at(v, li_), own[li_]));
  };
  if (inf) {
    xyzz_set_inf<F2>(a);
  } else {
    to2(a.x, acc.x);
    to2(a.y, acc.y);
    to2(a.zz, acc.zz);
    to2(a.zzz, acc.zzz);
  }
  to2(p.x, q.x);
  to2(p.y, q.y);
  xyzz_madd<F2>(a, p, false);
  inf = xyzz_is_inf<F2>(a);
  if (!inf) {
    from2(acc.x, a.x);
    from2(acc.y, a.y);
    from2(acc.zz, a.zz);
    from2(acc.zzz, a.zzz);
  }
}

// acc += q (q negated first when `negate`); pair-uniform control flow
template <class C, class B>
MLHIP_HD void xyzz28_lp_madd(XYZZ28L<typename B::V>& acc, bool& inf, const Affine28L<typename B::V>& q_in, bool negate) {
  typedef typename B::V V;
  if (lp28_all_zero<C, B>(q_in.x) && lp28_all_zero<C, B>(q_in.y)) return;  // point at infinity
  Affine28L<V> q;
  q.x = q_in.x;
  V ny;
  lp28_neg<C, B>(ny, q_in.y);
  lp28_select<C, B>(q.y, negate, ny, q_in.y);
  if (inf) {
    acc.x = q.x;
    acc.y = q.y;
    V zero, one;
    MLHIP_LP28_EACH(B, fp28_zero<C>(B::at(zero, li_)));
    MLHIP_LP28_EACH(B, fp28_from_const<C>(B::at(one, li_), C::ONE28));
    B::sel_hi(acc.zz, zero, one);  // 1 + 0 u
    acc.zzz = acc.zz;
    inf = false;
    return;
  }
  V U2, S2, P, R, PP, PPP, Q, X3, t, e, Vv, T;
  lp28_mul<C, B>(U2, q.x, acc.zz);
  lp28_mul<C, B>(S2, q.y, acc.zzz);
  lp28_sub<C, B>(t, U2, acc.x);
  lp28_settle<C, B>(P, t);
  lp28_sub<C, B>(t, S2, acc.y);
  lp28_settle<C, B>(R, t);
  if (lp28_is_zero_exact<C, B>(P)) {
    XYZZ28L<V> ta = acc;  // cold-path copies: keep the caller's accumulator in registers
    Affine28L<V> tq = q;
    bool ti = inf;
    xyzz28_lp_madd_exact<C, B>(ta, ti, tq);
    acc = ta;
    inf = ti;
    return;
  }
  lp28_sqr<C, B>(PP, P);
  lp28_mul<C, B>(PPP, P, PP);
  lp28_mul<C, B>(Q, acc.x, PP);
  lp28_sqr<C, B>(t, R);
  lp28_sub<C, B>(t, t, PPP);
  lp28_sub<C, B>(t, t, Q);
  lp28_sub<C, B>(t, t, Q);
  lp28_settle<C, B>(X3, t);
  lp28_sub<C, B>(e, Q, X3);                      // weight 2
  if constexpr (C::BETA != -1) lp28_normalize<C, B>(e, e);  // (the c0 lane's product leaves room for weight 1 only)
  lp28_mul<C, B>(Vv, R, e);            // 1 x 2
  lp28_mul<C, B>(T, acc.y, PPP);       // 1 x 1
  lp28_sub<C, B>(t, Vv, T);
  lp28_settle<C, B>(acc.y, t);
  acc.x = X3;
  lp28_mul<C, B>(t, acc.zz, PP);
  acc.zz = t;
  lp28_mul<C, B>(t, acc.zzz, PPP);
  acc.zzz = t;
}

// ---- full addition XYZZ + XYZZ (add-2008-s) for the bucket reduction --------------------------------------------------
// the exceptional cases (same x: doubling or cancellation; decided exactly), every lane on the rebuilt Fp2 points
template <class C, class B>
MLHIP_HD_NOINLINE void xyzz28_lp_add_exact(XYZZ28L<typename B::V>& acc, bool& inf, const XYZZ28L<typename B::V>& q) {
  typedef Fp2Field<C> F2;
  XYZZ<F2> a, b;
  auto to2 = [](Fp2<C>& out, const typename B::V& v) {
    Fp<C> own[B::LANES];
    MLHIP_LP28_EACH(B, fp28_to_fp<C>(own[li_], B::at(v, li_)));
    B::gather(out, own);
  };
  auto from2 = [](typename B::V& v, const Fp2<C>& in) {
    Fp<C> own[B::LANES];
    B::scatter(own, in);
    MLHIP_LP28_EACH(B, fp28_from_fp<C>(B::at(v, li_), own[li_]));
  };
  to2(a.x, acc.x);
  to2(a.y, acc.y);
  to2(a.zz, acc.zz);
  to2(a.zzz, acc.zzz);
  to2(b.x, q.x);
  to2(b.y, q.y);
  to2(b.zz, q.zz);
  to2(b.zzz, q.zzz);
  xyzz_add<F2>(a, b);
  inf = xyzz_is_inf<F2>(a);
  if (!inf) {
    from2(acc.x, a.x);
    from2(acc.y, a.y);
    from2(acc.zz, a.zz);
    from2(acc.zzz, a.zzz);
  }
}

// acc += q, both finite-or-flagged XYZZ28L with normalized coordinates; pair-uniform control flow.  12 dual + 2 single
// products per lane and four carry propagations (P, R, X3, Y3: the dual product needs w_a w_b <= 4, the square w = 1).
template <class C, class B>
MLHIP_HD void xyzz28_lp_add(XYZZ28L<typename B::V>& acc, bool& inf, const XYZZ28L<typename B::V>& q, bool q_inf) {
  typedef typename B::V V;
  if (q_inf) return;
  if (inf) {
    acc = q;
    inf = false;
    return;
  }
  V U1, U2, S1, S2, P, R, PP, PPP, Q, X3, t, e, Vv, T;
  lp28_mul<C, B>(U1, acc.x, q.zz);
  lp28_mul<C, B>(U2, q.x, acc.zz);
  lp28_mul<C, B>(S1, acc.y, q.zzz);
  lp28_mul<C, B>(S2, q.y, acc.zzz);
  lp28_sub<C, B>(t, U2, U1);
  lp28_settle<C, B>(P, t);
  lp28_sub<C, B>(t, S2, S1);
  lp28_settle<C, B>(R, t);
  if (lp28_is_zero_exact<C, B>(P)) {
    XYZZ28L<V> ta = acc, tq = q;  // cold-path copies: keep the caller's values in registers
    bool ti = inf;
    xyzz28_lp_add_exact<C, B>(ta, ti, tq);
    acc = ta;
    inf = ti;
    return;
  }
  lp28_sqr<C, B>(PP, P);
  lp28_mul<C, B>(PPP, P, PP);
  lp28_mul<C, B>(Q, U1, PP);
  lp28_sqr<C, B>(t, R);
  lp28_sub<C, B>(t, t, PPP);
  lp28_sub<C, B>(t, t, Q);
  lp28_sub<C, B>(t, t, Q);
  lp28_settle<C, B>(X3, t);
  lp28_sub<C, B>(e, Q, X3);       // weight 2
  if constexpr (C::BETA != -1) lp28_normalize<C, B>(e, e);
  lp28_mul<C, B>(Vv, R, e);       // 1 x 2
  lp28_mul<C, B>(T, S1, PPP);     // 1 x 1
  lp28_sub<C, B>(t, Vv, T);
  lp28_settle<C, B>(acc.y, t);
  acc.x = X3;
  lp28_mul<C, B>(t, acc.zz, q.zz);
  lp28_mul<C, B>(acc.zz, t, PP);
  lp28_mul<C, B>(t, acc.zzz, q.zzz);
  lp28_mul<C, B>(acc.zzz, t, PPP);
}

}  // namespace mlhip
