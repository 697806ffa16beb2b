// ec.h -- short-Weierstrass (a = 0) group arithmetic for G1 (over Fp) and G2 (over Fp2), gfx950.
//
// Affine points use gnark-crypto's in-memory form G1Affine{X,Y} / G2Affine{X,Y} with the point at
// infinity encoded as (0,0) (reference: driver/gurvy/bls12381/bls12-381.go:211-213, :323-325 and the
// g1Infinity note at :33,41-42).  Bucket / partial sums use extended Jacobian "XYZZ" coordinates
// (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; infinity <=> ZZ = 0), the representation gnark's MultiExp uses
// for its buckets (call site bls12-381.go:777) -- formulas restated from the EFD
// (madd-2008-s, add-2008-s, dbl-2008-s-1, mdbl-2008-s-1), complete via explicit branches for
// infinity / doubling / inverse inputs so adversarial inputs (duplicates, P and -P) stay exact.
#pragma once
#include "tower.h"

namespace mlhip {

// Field policies: one interface over Fp (G1) and Fp2 (G2)
template <class C>
struct FpField {
  using Curve = C;
  using T = Fp<C>;
  static constexpr int WORDS = C::N;
  MLHIP_HD static void zero(T& r) { fp_zero<C>(r); }
  MLHIP_HD static void one(T& r) { fp_one<C>(r); }
  MLHIP_HD static bool is_zero(const T& a) { return fp_is_zero<C>(a); }
  MLHIP_HD static bool eq(const T& a, const T& b) { return fp_eq<C>(a, b); }
  MLHIP_HD static void add(T& r, const T& a, const T& b) { fp_add<C>(r, a, b); }
  MLHIP_HD static void sub(T& r, const T& a, const T& b) { fp_sub<C>(r, a, b); }
  MLHIP_HD static void dbl(T& r, const T& a) { fp_dbl<C>(r, a); }
  MLHIP_HD static void neg(T& r, const T& a) { fp_neg<C>(r, a); }
  MLHIP_HD static void mul(T& r, const T& a, const T& b) { fp_mul_i<C>(r, a, b); }
  MLHIP_HD static void sqr(T& r, const T& a) { fp_sqr_i<C>(r, a); }
  MLHIP_HD static void inv(T& r, const T& a) { fp_inv<C>(r, a); }
  MLHIP_HD static void select(T& r, bool c, const T& a, const T& b) { fp_select<C>(r, c, a, b); }
};

template <class C>
struct Fp2Field {
  using Curve = C;
  using T = Fp2<C>;
  static constexpr int WORDS = 2 * C::N;
  MLHIP_HD static void zero(T& r) { fp2_zero<C>(r); }
  MLHIP_HD static void one(T& r) { fp2_one<C>(r); }
  MLHIP_HD static bool is_zero(const T& a) { return fp2_is_zero<C>(a); }
  MLHIP_HD static bool eq(const T& a, const T& b) { return fp2_eq<C>(a, b); }
  MLHIP_HD static void add(T& r, const T& a, const T& b) { fp2_add<C>(r, a, b); }
  MLHIP_HD static void sub(T& r, const T& a, const T& b) { fp2_sub<C>(r, a, b); }
  MLHIP_HD static void dbl(T& r, const T& a) { fp2_dbl<C>(r, a); }
  MLHIP_HD static void neg(T& r, const T& a) { fp2_neg<C>(r, a); }
  MLHIP_HD static void mul(T& r, const T& a, const T& b) { fp2_mul<C>(r, a, b); }
  MLHIP_HD static void sqr(T& r, const T& a) { fp2_sqr<C>(r, a); }
  MLHIP_HD static void inv(T& r, const T& a) { fp2_inv<C>(r, a); }
  MLHIP_HD static void select(T& r, bool c, const T& a, const T& b) { fp2_select<C>(r, c, a, b); }
};

template <class F>
struct Affine {
  typename F::T x, y;
};

template <class F>
struct XYZZ {
  typename F::T x, y, zz, zzz;
};

template <class F>
MLHIP_HD bool affine_is_inf(const Affine<F>& p) {
  return F::is_zero(p.x) & F::is_zero(p.y);
}

template <class F>
MLHIP_HD void xyzz_set_inf(XYZZ<F>& r) {
  F::one(r.x);
  F::one(r.y);
  F::zero(r.zz);
  F::zero(r.zzz);
}

template <class F>
MLHIP_HD bool xyzz_is_inf(const XYZZ<F>& p) {
  return F::is_zero(p.zz);
}

template <class F>
MLHIP_HD void xyzz_from_affine(XYZZ<F>& r, const Affine<F>& p) {
  if (affine_is_inf<F>(p)) {
    xyzz_set_inf<F>(r);
    return;
  }
  r.x = p.x;
  r.y = p.y;
  F::one(r.zz);
  F::one(r.zzz);
}

// r = 2 * (affine p)        (mdbl-2008-s-1)
template <class F>
MLHIP_HD void xyzz_mdbl(XYZZ<F>& r, const Affine<F>& p) {
  typename F::T U, V, W, S, M, t;
  F::dbl(U, p.y);
  F::sqr(V, U);
  F::mul(W, U, V);
  F::mul(S, p.x, V);
  F::sqr(M, p.x);
  F::dbl(t, M);
  F::add(M, M, t);  // 3 x^2
  F::sqr(r.x, M);
  F::sub(r.x, r.x, S);
  F::sub(r.x, r.x, S);
  F::sub(t, S, r.x);
  F::mul(t, M, t);
  F::mul(U, W, p.y);
  F::sub(r.y, t, U);
  r.zz = V;
  r.zzz = W;
}

// r = 2 * p                (dbl-2008-s-1)
template <class F>
MLHIP_HD void xyzz_dbl(XYZZ<F>& r, const XYZZ<F>& p) {
  if (xyzz_is_inf<F>(p)) {
    xyzz_set_inf<F>(r);
    return;
  }
  typename F::T U, V, W, S, M, t, X3, Y3;
  F::dbl(U, p.y);
  F::sqr(V, U);
  F::mul(W, U, V);
  F::mul(S, p.x, V);
  F::sqr(M, p.x);
  F::dbl(t, M);
  F::add(M, M, t);
  F::sqr(X3, M);
  F::sub(X3, X3, S);
  F::sub(X3, X3, S);
  F::sub(t, S, X3);
  F::mul(t, M, t);
  F::mul(U, W, p.y);
  F::sub(Y3, t, U);
  F::mul(r.zz, V, p.zz);
  F::mul(r.zzz, W, p.zzz);
  r.x = X3;
  r.y = Y3;
}

// acc += (affine q), q negated first when `negate` (signed Pippenger digits).  madd-2008-s.
template <class F>
MLHIP_HD void xyzz_madd(XYZZ<F>& acc, const Affine<F>& q_in, bool negate) {
  if (affine_is_inf<F>(q_in)) return;
  Affine<F> q;
  q.x = q_in.x;
  typename F::T ny;
  F::neg(ny, q_in.y);
  F::select(q.y, negate, ny, q_in.y);
  if (xyzz_is_inf<F>(acc)) {
    acc.x = q.x;
    acc.y = q.y;
    F::one(acc.zz);
    F::one(acc.zzz);
    return;
  }
  typename F::T U2, S2, P, R, PP, PPP, Q, t;
  F::mul(U2, q.x, acc.zz);
  F::mul(S2, q.y, acc.zzz);
  F::sub(P, U2, acc.x);
  F::sub(R, S2, acc.y);
  if (F::is_zero(P)) {
    if (F::is_zero(R)) {
      xyzz_mdbl<F>(acc, q);
    } else {
      xyzz_set_inf<F>(acc);
    }
    return;
  }
  F::sqr(PP, P);
  F::mul(PPP, P, PP);
  F::mul(Q, acc.x, PP);
  F::sqr(t, R);
  F::sub(t, t, PPP);
  F::sub(t, t, Q);
  F::sub(t, t, Q);  // X3
  F::sub(Q, Q, t);
  F::mul(Q, R, Q);
  F::mul(S2, acc.y, PPP);
  F::sub(acc.y, Q, S2);
  acc.x = t;
  F::mul(acc.zz, acc.zz, PP);
  F::mul(acc.zzz, acc.zzz, PPP);
}

// acc += q                 (add-2008-s)
template <class F>
MLHIP_HD void xyzz_add(XYZZ<F>& acc, const XYZZ<F>& q) {
  if (xyzz_is_inf<F>(q)) return;
  if (xyzz_is_inf<F>(acc)) {
    acc = q;
    return;
  }
  typename F::T U1, U2, S1, S2, P, R, PP, PPP, Q, t;
  F::mul(U1, acc.x, q.zz);
  F::mul(U2, q.x, acc.zz);
  F::mul(S1, acc.y, q.zzz);
  F::mul(S2, q.y, acc.zzz);
  F::sub(P, U2, U1);
  F::sub(R, S2, S1);
  if (F::is_zero(P)) {
    if (F::is_zero(R)) {
      XYZZ<F> d;
      xyzz_dbl<F>(d, q);
      acc = d;
    } else {
      xyzz_set_inf<F>(acc);
    }
    return;
  }
  F::sqr(PP, P);
  F::mul(PPP, P, PP);
  F::mul(Q, U1, PP);
  F::sqr(t, R);
  F::sub(t, t, PPP);
  F::sub(t, t, Q);
  F::sub(t, t, Q);  // X3
  F::sub(Q, Q, t);
  F::mul(Q, R, Q);
  F::mul(S1, S1, PPP);
  F::sub(acc.y, Q, S1);
  acc.x = t;
  F::mul(acc.zz, acc.zz, q.zz);
  F::mul(acc.zz, acc.zz, PP);
  F::mul(acc.zzz, acc.zzz, q.zzz);
  F::mul(acc.zzz, acc.zzz, PPP);
}

// affine = X/ZZ, Y/ZZZ  (one field inversion; infinity -> (0,0) as gnark's FromJacobian gives)
template <class F>
MLHIP_HD void xyzz_to_affine(Affine<F>& r, const XYZZ<F>& p) {
  if (xyzz_is_inf<F>(p)) {
    F::zero(r.x);
    F::zero(r.y);
    return;
  }
  typename F::T zi, zi2, zi3;
  // 1/ZZZ and 1/ZZ from a single inversion: 1/ZZ = ZZ^2 * ZZZ^-2 ... use i = 1/(ZZ*ZZZ)
  typename F::T m, i;
  F::mul(m, p.zz, p.zzz);
  F::inv(i, m);
  F::mul(zi2, i, p.zzz);  // 1/ZZ
  F::mul(zi3, i, p.zz);   // 1/ZZZ
  (void)zi;
  F::mul(r.x, p.x, zi2);
  F::mul(r.y, p.y, zi3);
}

}  // namespace mlhip
