// ec28.h -- XYZZ bucket accumulation over the carry-free field form (fp28.h).  Only the hot loop of the MSM
// (k_accumulate) uses it: points are converted once per MSM, bucket sums are converted back to the boundary form
// before the reduction kernels.  Same formulas as ec.h: xyzz_madd (madd-2008-s).
#pragma once
#include "ec.h"
#include "fp28.h"

namespace mlhip {

template <class C>
struct alignas(16) Affine28 {  // 112 B (BLS12) / 80 B (BN254): whole 16-byte vectors for the gathered loads
  Fp28<C> x, y;
};
template <class C>
struct XYZZ28 {
  Fp28<C> x, y, zz, zzz;  // all normalized (weight 1); "infinity" is carried in a separate flag
};

template <class C>
MLHIP_HD void affine28_from(Affine28<C>& r, const Affine<FpField<C>>& p) {
  fp28_from_fp<C>(r.x, p.x);
  fp28_from_fp<C>(r.y, p.y);
}

template <class C>
MLHIP_HD void xyzz28_to(XYZZ<FpField<C>>& r, const XYZZ28<C>& a, bool inf) {
  if (inf) {
    xyzz_set_inf<FpField<C>>(r);
    return;
  }
  fp28_to_fp<C>(r.x, a.x);
  fp28_to_fp<C>(r.y, a.y);
  fp28_to_fp<C>(r.zz, a.zz);
  fp28_to_fp<C>(r.zzz, a.zzz);
}

// the exceptional cases of the mixed addition (q = +-acc): done exactly in the boundary form.  Out of line and
// practically never executed (it needs U2 = X1 mod p), so its registers do not count against the hot loop.
template <class C>
MLHIP_HD_NOINLINE void xyzz28_madd_exact(XYZZ28<C>& acc, bool& inf, const Affine28<C>& q) {
  typedef FpField<C> F;
  XYZZ<F> a;
  Affine<F> p;
  xyzz28_to<C>(a, acc, inf);
  fp28_to_fp<C>(p.x, q.x);
  fp28_to_fp<C>(p.y, q.y);
  xyzz_madd<F>(a, p, false);
  inf = xyzz_is_inf<F>(a);
  if (!inf) {
    fp28_from_fp<C>(acc.x, a.x);
    fp28_from_fp<C>(acc.y, a.y);
    fp28_from_fp<C>(acc.zz, a.zz);
    fp28_from_fp<C>(acc.zzz, a.zzz);
  }
}

// acc += q (q negated first when `negate`).  Weights: see fp28.h; every stored coordinate is normalized.
template <class C>
MLHIP_HD void xyzz28_madd(XYZZ28<C>& acc, bool& inf, const Affine28<C>& q_in, bool negate) {
  if (fp28_all_zero<C>(q_in.x) & fp28_all_zero<C>(q_in.y)) return;  // point at infinity
  Affine28<C> q;
  q.x = q_in.x;
  Fp28<C> ny;
  fp28_neg<C>(ny, q_in.y);
  fp28_select<C>(q.y, negate, ny, q_in.y);
  if (inf) {
    acc.x = q.x;
    acc.y = q.y;
    fp28_from_const<C>(acc.zz, C::ONE28);
    fp28_from_const<C>(acc.zzz, C::ONE28);
    inf = false;
    return;
  }
  Fp28<C> U2, S2, P, R, PP, PPP, Q, X3, t;
  fp28_mul<C>(U2, q.x, acc.zz);   // 1 x 1
  fp28_mul<C>(S2, q.y, acc.zzz);  // 1 x 1
  fp28_sub<C>(P, U2, acc.x);      // weight 2
  fp28_sub<C>(R, S2, acc.y);      // weight 2
  if (fp28_maybe_zero<C>(P)) {
    if (fp28_is_zero_exact<C>(P)) {
      // copies: only these cold-path temporaries have their address taken, so the caller's accumulator stays in
      // registers (handing `acc` itself to the out-of-line function pins it in scratch for the whole loop)
      XYZZ28<C> ta = acc;
      Affine28<C> tq = q;
      bool ti = inf;
      xyzz28_madd_exact<C>(ta, ti, tq);
      acc = ta;
      inf = ti;
      return;
    }
  }
  fp28_sqr<C>(PP, P);           // weight 2 squared
  fp28_mul<C>(PPP, P, PP);      // 2 x 1
  fp28_mul<C>(Q, acc.x, PP);    // 1 x 1
  fp28_sqr<C>(t, R);            // weight 2 squared
  fp28_sub<C>(t, t, PPP);
  fp28_sub<C>(t, t, Q);
  fp28_sub<C>(t, t, Q);         // X3, weight 4
  fp28_normalize<C>(X3, t);     // weight 1
  fp28_sub<C>(Q, Q, X3);        // weight 2
  fp28_neg<C>(t, acc.y);        // weight 1
  fp28_mul2<C>(acc.y, R, Q, t, PPP);  // R (Q - X3) - Y1 PPP: 2 x 2 + 1 x 1
  acc.x = X3;
  fp28_mul<C>(acc.zz, acc.zz, PP);
  fp28_mul<C>(acc.zzz, acc.zzz, PPP);
}

// r = 2 p (dbl-2008-s-1, a = 0) in the carry-free form, p finite with normalized coordinates; the result is normalized,
// or all-zero limbs (infinity) when p has order two (y = 0 mod p, however it is represented: BLS12-377's curve has such
// a point).  Weights: U = 2Y (2), M = 3 X^2 (3, carry-propagated before it is squared), X3 = M^2 - 2S (3, propagated).
template <class C>
MLHIP_HD void xyzz28_dbl(XYZZ28<C>& r, const XYZZ28<C>& p) {
  Fp28<C> U, V, W, S, M, X3, t, nW;
  fp28_add<C>(U, p.y, p.y);
  if (fp28_maybe_zero<C>(U) && fp28_is_zero_exact<C>(U)) {
    fp28_zero<C>(r.x);
    fp28_zero<C>(r.y);
    fp28_zero<C>(r.zz);
    fp28_zero<C>(r.zzz);
    return;
  }
  fp28_sqr<C>(V, U);       // weight 2 squared
  fp28_mul<C>(W, U, V);    // 2 x 1
  fp28_mul<C>(S, p.x, V);  // 1 x 1
  fp28_sqr<C>(M, p.x);
  fp28_add<C>(t, M, M);
  fp28_add<C>(t, t, M);     // weight 3
  fp28_normalize<C>(M, t);  // weight 1, |value| < 3.6 p
  fp28_sqr<C>(t, M);
  fp28_sub<C>(t, t, S);
  fp28_sub<C>(t, t, S);      // weight 3
  fp28_normalize<C>(X3, t);  // weight 1
  fp28_sub<C>(t, S, X3);     // weight 2
  fp28_neg<C>(nW, W);
  fp28_mul2<C>(r.y, M, t, nW, p.y);  // M (S - X3) - W Y1: 1 x 2 + 1 x 1
  fp28_mul<C>(r.zz, V, p.zz);
  fp28_mul<C>(r.zzz, W, p.zzz);
  r.x = X3;
}

// acc += q, both XYZZ28 with normalized coordinates and their own infinity flags (add-2008-s), one lane: 12 products + 2
// squares, Y3 = R (Q - X3) - S1 PPP one fused dual product.  Same point: doubled (xyzz28_dbl); opposite points: infinity.
// For the chains of the subgroup test (codec.h); the bucket reduction runs the quad-lane schedule (ec_quad28.h).
template <class C>
MLHIP_HD void xyzz28_add(XYZZ28<C>& acc, bool& inf, const XYZZ28<C>& q, bool q_inf) {
  if (q_inf) return;
  if (inf) {
    acc = q;
    inf = false;
    return;
  }
  Fp28<C> U1, U2, S1, S2, P, R, PP, PPP, Q, X3, t, n;
  fp28_mul<C>(U1, acc.x, q.zz);
  fp28_mul<C>(U2, q.x, acc.zz);
  fp28_mul<C>(S1, acc.y, q.zzz);
  fp28_mul<C>(S2, q.y, acc.zzz);
  fp28_sub<C>(P, U2, U1);  // weight 2
  fp28_sub<C>(R, S2, S1);
  if (fp28_maybe_zero<C>(P) && fp28_is_zero_exact<C>(P)) {
    if (fp28_maybe_zero<C>(R) && fp28_is_zero_exact<C>(R)) {
      XYZZ28<C> d;
      xyzz28_dbl<C>(d, q);
      acc = d;
      inf = fp28_all_zero<C>(d.zz);  // a point of order two doubles to infinity (canonical zeros)
    } else {
      inf = true;
    }
    return;
  }
  fp28_sqr<C>(PP, P);
  fp28_mul<C>(PPP, P, PP);  // 2 x 1
  fp28_mul<C>(Q, U1, PP);
  fp28_sqr<C>(t, R);
  fp28_sub<C>(t, t, PPP);
  fp28_sub<C>(t, t, Q);
  fp28_sub<C>(t, t, Q);  // weight 4
  fp28_normalize<C>(X3, t);
  fp28_sub<C>(Q, Q, X3);  // weight 2
  fp28_neg<C>(n, S1);
  fp28_mul2<C>(acc.y, R, Q, n, PPP);  // 2 x 2 + 1 x 1
  acc.x = X3;
  fp28_mul<C>(t, acc.zz, q.zz);
  fp28_mul<C>(acc.zz, t, PP);
  fp28_mul<C>(t, acc.zzz, q.zzz);
  fp28_mul<C>(acc.zzz, t, PPP);
}

}  // namespace mlhip
