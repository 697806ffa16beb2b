// tower.h -- Fp2 / Fp6 / Fp12 extension towers for BN254, BLS12-381, BLS12-377 (gfx950 kernels).
//
//   Fp2  = Fp[u]/(u^2 - BETA)           BETA = -1 (BN254, BLS12-381), -5 (BLS12-377)
//   Fp6  = Fp2[v]/(v^3 - XI)            XI = 9+u, 1+u, u
//   Fp12 = Fp6[w]/(w^2 - v)
//
// The struct nesting Fp12{c0,c1: Fp6{c0,c1,c2: Fp2{c0,c1}}} is byte-for-byte gnark-crypto's
// E12{C0,C1 E6{B0,B1,B2 E2{A0,A1}}} (the value inside the reference's Gt wrappers:
// driver/gurvy/bls12381/bls12-381.go:395-397, driver/gurvy/bn254.go, driver/gurvy/bls12-377.go),
// so a Gt crosses the C ABI without conversion.
//
// These functions replace, for the GPU path, the gnark-crypto tower arithmetic behind
// MillerLoop / FinalExponentiation (call sites bls12-381.go:449,458,467; bn254.go:248,257,266;
// bls12-377.go:245,254,263).  The algorithms (Karatsuba, Chung-Hasan squaring, Granger-Scott
// cyclotomic squaring, sparse line multiplication) are the textbook ones, restated from the
// public literature; results are checked against oracle/pyref.py.
#pragma once
#include "fp.h"

namespace mlhip {

template <class C>
struct Fp2 {
  Fp<C> c0, c1;
};
// E2 is the Fp2 element type: Fp2<C> (one element per lane) or Fp2L<C> (fp2_lanes.h: one component per
// lane, an element per lane pair).  Everything from Fp6 up is written against the fp2_* overload set.
template <class C, class E2 = Fp2<C>>
struct Fp6 {
  E2 c0, c1, c2;
};
template <class C, class E2 = Fp2<C>>
struct Fp12 {
  Fp6<C, E2> c0, c1;
};

// ------------------------------------------------------------------ Fp2
template <class C>
MLHIP_HD void fp2_zero(Fp2<C>& r) {
  fp_zero<C>(r.c0);
  fp_zero<C>(r.c1);
}
template <class C>
MLHIP_HD void fp2_one(Fp2<C>& r) {
  fp_one<C>(r.c0);
  fp_zero<C>(r.c1);
}
template <class C>
MLHIP_HD bool fp2_is_zero(const Fp2<C>& a) {
  return fp_is_zero<C>(a.c0) & fp_is_zero<C>(a.c1);
}
template <class C>
MLHIP_HD bool fp2_eq(const Fp2<C>& a, const Fp2<C>& b) {
  return fp_eq<C>(a.c0, b.c0) & fp_eq<C>(a.c1, b.c1);
}
template <class C>
MLHIP_HD void fp2_add(Fp2<C>& r, const Fp2<C>& a, const Fp2<C>& b) {
  fp_add<C>(r.c0, a.c0, b.c0);
  fp_add<C>(r.c1, a.c1, b.c1);
}
template <class C>
MLHIP_HD void fp2_sub(Fp2<C>& r, const Fp2<C>& a, const Fp2<C>& b) {
  fp_sub<C>(r.c0, a.c0, b.c0);
  fp_sub<C>(r.c1, a.c1, b.c1);
}
template <class C>
MLHIP_HD void fp2_dbl(Fp2<C>& r, const Fp2<C>& a) {
  fp_dbl<C>(r.c0, a.c0);
  fp_dbl<C>(r.c1, a.c1);
}
template <class C>
MLHIP_HD void fp2_neg(Fp2<C>& r, const Fp2<C>& a) {
  fp_neg<C>(r.c0, a.c0);
  fp_neg<C>(r.c1, a.c1);
}
template <class C>
MLHIP_HD void fp2_conj(Fp2<C>& r, const Fp2<C>& a) {
  r.c0 = a.c0;
  fp_neg<C>(r.c1, a.c1);
}
template <class C>
MLHIP_HD void fp2_select(Fp2<C>& r, bool c, const Fp2<C>& a, const Fp2<C>& b) {
  fp_select<C>(r.c0, c, a.c0, b.c0);
  fp_select<C>(r.c1, c, a.c1, b.c1);
}

// Weight hooks of the carry-free element type (fp2_lanes28.h): fp2_norm carry-propagates a value whose lazily
// accumulated additions would exceed a product's budget.  Saturated elements are always fully reduced: no-ops.
template <class C>
MLHIP_HD void fp2_norm(Fp2<C>&) {}
template <class C>
MLHIP_HD void fp2_reduce(Fp2<C>&) {}
template <class C>
MLHIP_HD int fp2_weight(const Fp2<C>&) { return 1; }

// r = BETA * a  (BETA = -1 or -5)
template <class C>
MLHIP_HD void fp_mul_beta(Fp<C>& r, const Fp<C>& a) {
  if (C::BETA == -1) {
    fp_neg<C>(r, a);
  } else {
    Fp<C> t;
    fp_mul_small<C>(t, a, -C::BETA);
    fp_neg<C>(r, t);
  }
}

// fp2_mul / fp2_sqr are the out-of-line units of the towers and of G2: three inlined Fp multiplies
// each, operands loaded once into registers.
template <class C>
MLHIP_HD_NOINLINE void fp2_mul(Fp2<C>& r, const Fp2<C>& a, const Fp2<C>& b) {
  Fp<C> t0, t1, t2, s0, s1;
  fp_mul_i<C>(t0, a.c0, b.c0);
  fp_mul_i<C>(t1, a.c1, b.c1);
  fp_add<C>(s0, a.c0, a.c1);
  fp_add<C>(s1, b.c0, b.c1);
  fp_mul_i<C>(t2, s0, s1);
  fp_sub<C>(t2, t2, t0);
  fp_sub<C>(r.c1, t2, t1);
  fp_mul_beta<C>(t1, t1);
  fp_add<C>(r.c0, t0, t1);
}

template <class C>
MLHIP_HD_NOINLINE void fp2_sqr(Fp2<C>& r, const Fp2<C>& a) {
  if (C::BETA == -1) {
    Fp<C> s, d, m;
    fp_add<C>(s, a.c0, a.c1);
    fp_sub<C>(d, a.c0, a.c1);
    fp_mul_i<C>(m, a.c0, a.c1);
    fp_mul_i<C>(r.c0, s, d);
    fp_dbl<C>(r.c1, m);
  } else {
    Fp<C> t0, t1, m;
    fp_sqr_i<C>(t0, a.c0);
    fp_sqr_i<C>(t1, a.c1);
    fp_mul_i<C>(m, a.c0, a.c1);
    fp_mul_beta<C>(t1, t1);
    fp_add<C>(r.c0, t0, t1);
    fp_dbl<C>(r.c1, m);
  }
}

template <class C>
MLHIP_HD void fp2_mul_fp(Fp2<C>& r, const Fp2<C>& a, const Fp<C>& k) {
  fp_mul<C>(r.c0, a.c0, k);
  fp_mul<C>(r.c1, a.c1, k);
}

template <class C>
MLHIP_HD void fp2_mul_small(Fp2<C>& r, const Fp2<C>& a, int k) {
  fp_mul_small<C>(r.c0, a.c0, k);
  fp_mul_small<C>(r.c1, a.c1, k);
}

// r = XI * a,  XI = XI0 + XI1 u (small integers)
template <class C>
MLHIP_HD void fp2_mul_xi(Fp2<C>& r, const Fp2<C>& a) {
  Fp<C> t0, t1, n0, n1;
  // (a0 + a1 u)(x0 + x1 u) = (a0 x0 + BETA a1 x1) + (a0 x1 + a1 x0) u
  if (C::XI0 == 0) {
    // XI = u (XI1 == 1)
    fp_mul_beta<C>(n0, a.c1);
    n1 = a.c0;
  } else {
    fp_mul_small<C>(t0, a.c0, C::XI0);
    fp_mul_small<C>(t1, a.c1, C::XI1);
    fp_mul_beta<C>(t1, t1);
    fp_add<C>(n0, t0, t1);
    fp_mul_small<C>(t0, a.c0, C::XI1);
    fp_mul_small<C>(t1, a.c1, C::XI0);
    fp_add<C>(n1, t0, t1);
  }
  r.c0 = n0;
  r.c1 = n1;
}

template <class C>
MLHIP_HD void fp2_inv(Fp2<C>& r, const Fp2<C>& a) {
  // 1/(a0 + a1 u) = (a0 - a1 u) / (a0^2 - BETA a1^2)
  Fp<C> t0, t1, n;
  fp_sqr<C>(t0, a.c0);
  fp_sqr<C>(t1, a.c1);
  fp_mul_beta<C>(t1, t1);
  fp_sub<C>(n, t0, t1);
  fp_inv<C>(n, n);
  fp_mul<C>(r.c0, a.c0, n);
  fp_mul<C>(t0, a.c1, n);
  fp_neg<C>(r.c1, t0);
}

template <class C>
MLHIP_HD void fp2_from_const(Fp2<C>& r, const uint32_t (&k)[2][C::N]) {
  fp_from_const<C>(r.c0, k[0]);
  fp_from_const<C>(r.c1, k[1]);
}

// r = a * Re(k) for a constant k whose imaginary part is zero (Frobenius^2 coefficients)
template <class C>
MLHIP_HD void fp2_mul_by_real_const(Fp2<C>& r, const Fp2<C>& a, const uint32_t (&k)[2][C::N]) {
  Fp<C> kr;
  fp_from_const<C>(kr, k[0]);
  fp2_mul_fp<C>(r, a, kr);
}

// ------------------------------------------------------------------ Fp6
template <class C, class E2>
MLHIP_HD void fp6_zero(Fp6<C, E2>& r) {
  fp2_zero<C>(r.c0);
  fp2_zero<C>(r.c1);
  fp2_zero<C>(r.c2);
}
template <class C, class E2>
MLHIP_HD void fp6_add(Fp6<C, E2>& r, const Fp6<C, E2>& a, const Fp6<C, E2>& b) {
  fp2_add<C>(r.c0, a.c0, b.c0);
  fp2_add<C>(r.c1, a.c1, b.c1);
  fp2_add<C>(r.c2, a.c2, b.c2);
}
template <class C, class E2>
MLHIP_HD void fp6_sub(Fp6<C, E2>& r, const Fp6<C, E2>& a, const Fp6<C, E2>& b) {
  fp2_sub<C>(r.c0, a.c0, b.c0);
  fp2_sub<C>(r.c1, a.c1, b.c1);
  fp2_sub<C>(r.c2, a.c2, b.c2);
}
template <class C, class E2>
MLHIP_HD void fp6_neg(Fp6<C, E2>& r, const Fp6<C, E2>& a) {
  fp2_neg<C>(r.c0, a.c0);
  fp2_neg<C>(r.c1, a.c1);
  fp2_neg<C>(r.c2, a.c2);
}
template <class C, class E2>
MLHIP_HD void fp6_dbl(Fp6<C, E2>& r, const Fp6<C, E2>& a) {
  fp2_dbl<C>(r.c0, a.c0);
  fp2_dbl<C>(r.c1, a.c1);
  fp2_dbl<C>(r.c2, a.c2);
}
template <class C, class E2>
MLHIP_HD void fp6_norm(Fp6<C, E2>& r) {
  fp2_norm<C>(r.c0);
  fp2_norm<C>(r.c1);
  fp2_norm<C>(r.c2);
}
template <class C, class E2>
MLHIP_HD void fp6_reduce(Fp6<C, E2>& r) {
  fp2_reduce<C>(r.c0);
  fp2_reduce<C>(r.c1);
  fp2_reduce<C>(r.c2);
}
// Weight discipline of the carry-free element (fp2_lanes28.h; the numbers in the comments below are weights): every
// function from here on takes operands whose coefficients are normalized (weight 1) and leaves normalized results,
// except the ones marked raw, whose results are combined by the caller before ONE carry propagation.  The Fp12-level
// functions also bring the VALUE of every result coefficient back under p (fp6_reduce instead of fp6_norm), so that
// each of them can be analysed by itself: operands of weight 1 and value bound 1 in, the same out.

// r = v * a
template <class C, class E2>
MLHIP_HD void fp6_mul_v(Fp6<C, E2>& r, const Fp6<C, E2>& a) {
  E2 t;
  fp2_mul_xi<C>(t, a.c2);
  r.c2 = a.c1;
  r.c1 = a.c0;
  r.c0 = t;
}

// inlined body (callers that keep the Fp6 operands in registers) and the shared out-of-line copy
template <class C, class E2>
MLHIP_HD void fp6_mul_i(Fp6<C, E2>& r, const Fp6<C, E2>& a, const Fp6<C, E2>& b) {
  E2 t0, t1, t2, s0, s1, x0, x1, x2;
  fp2_mul<C>(t0, a.c0, b.c0);
  fp2_mul<C>(t1, a.c1, b.c1);
  fp2_mul<C>(t2, a.c2, b.c2);
  // c0 = xi((a1+a2)(b1+b2) - t1 - t2) + t0
  fp2_add<C>(s0, a.c1, a.c2);
  fp2_add<C>(s1, b.c1, b.c2);
  fp2_mul<C>(x0, s0, s1);
  fp2_sub<C>(x0, x0, t1);
  fp2_sub<C>(x0, x0, t2);
  fp2_mul_xi<C>(x0, x0);
  fp2_add<C>(x0, x0, t0);
  // c1 = (a0+a1)(b0+b1) - t0 - t1 + xi t2
  fp2_add<C>(s0, a.c0, a.c1);
  fp2_add<C>(s1, b.c0, b.c1);
  fp2_mul<C>(x1, s0, s1);
  fp2_sub<C>(x1, x1, t0);
  fp2_sub<C>(x1, x1, t1);
  fp2_mul_xi<C>(s0, t2);
  fp2_add<C>(x1, x1, s0);
  // c2 = (a0+a2)(b0+b2) - t0 - t2 + t1
  fp2_add<C>(s0, a.c0, a.c2);
  fp2_add<C>(s1, b.c0, b.c2);
  fp2_mul<C>(x2, s0, s1);
  fp2_sub<C>(x2, x2, t0);
  fp2_sub<C>(x2, x2, t2);
  fp2_add<C>(x2, x2, t1);
  // weights 7, 5, 4 (the operand sums have weight 2: their products are at the dual product's limit of 2 x 2 x 2)
  fp2_norm<C>(x0);
  fp2_norm<C>(x1);
  fp2_norm<C>(x2);
  r.c0 = x0;
  r.c1 = x1;
  r.c2 = x2;
}
template <class C, class E2>
MLHIP_HD_NOINLINE void fp6_mul(Fp6<C, E2>& r, const Fp6<C, E2>& a, const Fp6<C, E2>& b) {
  fp6_mul_i<C>(r, a, b);
}

template <class C, class E2>
MLHIP_HD_NOINLINE void fp6_sqr(Fp6<C, E2>& r, const Fp6<C, E2>& a) {
  // Chung-Hasan SQR2
  E2 s0, s1, s2, s3, s4, t;
  fp2_sqr<C>(s0, a.c0);
  fp2_mul<C>(s1, a.c0, a.c1);
  fp2_dbl<C>(s1, s1);
  fp2_sub<C>(t, a.c0, a.c1);
  fp2_add<C>(t, t, a.c2);
  fp2_norm<C>(t);  // 3 -> 1: the one-product square takes a normalized operand
  fp2_sqr<C>(s2, t);
  fp2_mul<C>(s3, a.c1, a.c2);
  fp2_dbl<C>(s3, s3);
  fp2_sqr<C>(s4, a.c2);
  // c0 = s0 + xi s3 ; c1 = s1 + xi s4 ; c2 = s1 + s2 + s3 - s0 - s4
  fp2_mul_xi<C>(t, s3);
  fp2_add<C>(r.c0, s0, t);
  fp2_mul_xi<C>(t, s4);
  fp2_add<C>(r.c1, s1, t);
  fp2_add<C>(t, s1, s2);
  fp2_add<C>(t, t, s3);
  fp2_sub<C>(t, t, s0);
  fp2_sub<C>(r.c2, t, s4);
  fp6_norm<C>(r);  // 5, 4, 7
}

// r = a * (b0 + b1 v); raw: result weights 3, 3, 2
template <class C, class E2>
MLHIP_HD void fp6_mul_by_01(Fp6<C, E2>& r, const Fp6<C, E2>& a, const E2& b0, const E2& b1) {
  E2 t0, t1, t2, x0, x1, x2, s0, s1;
  fp2_mul<C>(t0, a.c0, b0);
  fp2_mul<C>(t1, a.c1, b1);
  // c0 = a0 b0 + xi a2 b1
  fp2_mul<C>(t2, a.c2, b1);
  fp2_mul_xi<C>(t2, t2);
  fp2_add<C>(x0, t0, t2);
  // c1 = (a0+a1)(b0+b1) - t0 - t1
  fp2_add<C>(s0, a.c0, a.c1);
  fp2_add<C>(s1, b0, b1);
  fp2_mul<C>(x1, s0, s1);
  fp2_sub<C>(x1, x1, t0);
  fp2_sub<C>(x1, x1, t1);
  // c2 = a1 b1 + a2 b0
  fp2_mul<C>(x2, a.c2, b0);
  fp2_add<C>(x2, x2, t1);
  r.c0 = x0;
  r.c1 = x1;
  r.c2 = x2;
}

// r = a * (b1 v); raw: result weights 2, 1, 1
template <class C, class E2>
MLHIP_HD void fp6_mul_by_1(Fp6<C, E2>& r, const Fp6<C, E2>& a, const E2& b1) {
  E2 x0, x1, x2;
  fp2_mul<C>(x0, a.c2, b1);
  fp2_mul_xi<C>(x0, x0);
  fp2_mul<C>(x1, a.c0, b1);
  fp2_mul<C>(x2, a.c1, b1);
  r.c0 = x0;
  r.c1 = x1;
  r.c2 = x2;
}

// r = a * b0  (b0 in Fp2)
template <class C, class E2>
MLHIP_HD void fp6_mul_by_0(Fp6<C, E2>& r, const Fp6<C, E2>& a, const E2& b0) {
  fp2_mul<C>(r.c0, a.c0, b0);
  fp2_mul<C>(r.c1, a.c1, b0);
  fp2_mul<C>(r.c2, a.c2, b0);
}

template <class C, class E2>
MLHIP_HD_NOINLINE void fp6_inv(Fp6<C, E2>& r, const Fp6<C, E2>& a) {
  E2 c0, c1, c2, t, u;
  // c0 = a0^2 - xi a1 a2 ; c1 = xi a2^2 - a0 a1 ; c2 = a1^2 - a0 a2
  fp2_sqr<C>(c0, a.c0);
  fp2_mul<C>(t, a.c1, a.c2);
  fp2_mul_xi<C>(t, t);
  fp2_sub<C>(c0, c0, t);
  fp2_sqr<C>(c1, a.c2);
  fp2_mul_xi<C>(c1, c1);
  fp2_mul<C>(t, a.c0, a.c1);
  fp2_sub<C>(c1, c1, t);
  fp2_sqr<C>(c2, a.c1);
  fp2_mul<C>(t, a.c0, a.c2);
  fp2_sub<C>(c2, c2, t);
  // t = a0 c0 + xi (a2 c1 + a1 c2)
  fp2_mul<C>(t, a.c2, c1);
  fp2_mul<C>(u, a.c1, c2);
  fp2_add<C>(t, t, u);
  fp2_mul_xi<C>(t, t);
  fp2_mul<C>(u, a.c0, c0);
  fp2_add<C>(t, t, u);
  fp2_norm<C>(t);  // 5 -> 1
  fp2_inv<C>(t, t);
  fp2_mul<C>(r.c0, c0, t);
  fp2_mul<C>(r.c1, c1, t);
  fp2_mul<C>(r.c2, c2, t);
}

// ------------------------------------------------------------------ Fp12
template <class C, class E2>
MLHIP_HD void fp12_one(Fp12<C, E2>& r) {
  fp6_zero<C>(r.c0);
  fp6_zero<C>(r.c1);
  fp2_one<C>(r.c0.c0);
}
template <class C, class E2>
MLHIP_HD bool fp12_eq(const Fp12<C, E2>& a, const Fp12<C, E2>& b) {
  return fp2_eq<C>(a.c0.c0, b.c0.c0) & fp2_eq<C>(a.c0.c1, b.c0.c1) & fp2_eq<C>(a.c0.c2, b.c0.c2) &
         fp2_eq<C>(a.c1.c0, b.c1.c0) & fp2_eq<C>(a.c1.c1, b.c1.c1) & fp2_eq<C>(a.c1.c2, b.c1.c2);
}
template <class C, class E2>
MLHIP_HD void fp12_conj(Fp12<C, E2>& r, const Fp12<C, E2>& a) {
  r.c0 = a.c0;
  fp6_neg<C>(r.c1, a.c1);
}

template <class C, class E2>
MLHIP_HD_NOINLINE void fp12_mul(Fp12<C, E2>& r, const Fp12<C, E2>& a, const Fp12<C, E2>& b) {
  Fp6<C, E2> t0, t1, s0, s1, x;
  fp6_mul_i<C>(t0, a.c0, b.c0);
  fp6_mul_i<C>(t1, a.c1, b.c1);
  fp6_add<C>(s0, a.c0, a.c1);
  fp6_add<C>(s1, b.c0, b.c1);
  fp6_norm<C>(s0);  // 2 -> 1
  fp6_norm<C>(s1);
  fp6_mul_i<C>(x, s0, s1);
  fp6_sub<C>(x, x, t0);
  fp6_sub<C>(r.c1, x, t1);
  fp6_reduce<C>(r.c1);  // 3
  fp6_mul_v<C>(t1, t1);
  fp6_add<C>(r.c0, t0, t1);
  fp6_reduce<C>(r.c0);  // 3, 2, 2
}

// (inlined bodies + the shared out-of-line copies; fp12_sqr_mul_by_014 below chains the two bodies in one function)
template <class C, class E2>
MLHIP_HD void fp12_sqr_i(Fp12<C, E2>& r, const Fp12<C, E2>& a) {
  // complex squaring: c0 = (a0+a1)(a0+v a1) - ab - v ab ; c1 = 2ab
  Fp6<C, E2> ab, s0, s1, t;
  fp6_mul_i<C>(ab, a.c0, a.c1);
  fp6_add<C>(s0, a.c0, a.c1);
  fp6_norm<C>(s0);  // 2 -> 1
  fp6_mul_v<C>(t, a.c1);
  fp6_add<C>(s1, a.c0, t);
  fp6_norm<C>(s1);  // 3, 2, 2
  fp6_mul_i<C>(t, s0, s1);
  fp6_sub<C>(t, t, ab);
  fp6_mul_v<C>(s0, ab);
  fp6_sub<C>(r.c0, t, s0);
  fp6_reduce<C>(r.c0);  // 4, 3, 3
  fp6_dbl<C>(r.c1, ab);
  fp6_reduce<C>(r.c1);  // 2
}
template <class C, class E2>
MLHIP_HD_NOINLINE void fp12_sqr(Fp12<C, E2>& r, const Fp12<C, E2>& a) {
  fp12_sqr_i<C>(r, a);
}

template <class C, class E2>
MLHIP_HD_NOINLINE void fp12_inv(Fp12<C, E2>& r, const Fp12<C, E2>& a) {
  Fp6<C, E2> t0, t1;
  fp6_sqr<C>(t0, a.c0);
  fp6_sqr<C>(t1, a.c1);
  fp6_mul_v<C>(t1, t1);
  fp6_sub<C>(t0, t0, t1);
  fp6_reduce<C>(t0);  // 3, 2, 2 in weight; the values are sums of un-reduced Fp6 squares
  fp6_inv<C>(t0, t0);
  fp6_mul<C>(r.c0, a.c0, t0);
  fp6_mul<C>(t1, a.c1, t0);
  fp6_neg<C>(r.c1, t1);
  if constexpr (C::BETA != -1) {  // u^2 = -5: the c0 lane's products weigh 1 + 5, so un-reduced Fp6 products may not travel on
    fp6_reduce<C>(r.c0);
    fp6_reduce<C>(r.c1);
  }
}

// Frobenius f -> f^(p^K), K = 1, 2, 3.  Coefficient of w^i gets multiplied by GAMMAK[i]
// (after conjugation for odd K).  w-basis positions: g0=c0.c0 g1=c1.c0 g2=c0.c1 g3=c1.c1 g4=c0.c2 g5=c1.c2
template <class C, int K, class E2>
MLHIP_HD_NOINLINE void fp12_frob(Fp12<C, E2>& r, const Fp12<C, E2>& a) {
  const E2* src[6] = {&a.c0.c0, &a.c1.c0, &a.c0.c1, &a.c1.c1, &a.c0.c2, &a.c1.c2};
  E2* dst[6] = {&r.c0.c0, &r.c1.c0, &r.c0.c1, &r.c1.c1, &r.c0.c2, &r.c1.c2};
#pragma unroll
  for (int i = 0; i < 6; i++) {
    E2 x, g;
    if (K & 1)
      fp2_conj<C>(x, *src[i]);
    else
      x = *src[i];
    if (i == 0) {
      *dst[i] = x;
    } else {
      if (K == 2) {
        // gamma2[i] is a 6th root of unity in Fp (imaginary part is zero)
        fp2_mul_by_real_const<C>(*dst[i], x, C::GAMMA2[i]);
      } else {
        if (K == 1) fp2_from_const<C>(g, C::GAMMA1[i]);
        if (K == 3) fp2_from_const<C>(g, C::GAMMA3[i]);
        fp2_mul<C>(*dst[i], x, g);
      }
    }
  }
}

// Granger-Scott squaring, valid for f in the cyclotomic subgroup (after the easy part of FExp).
template <class C, class E2>
MLHIP_HD_NOINLINE void fp12_cyclo_sqr(Fp12<C, E2>& r, const Fp12<C, E2>& a) {
  E2 t0, t1, t2, t3, t4, t5, t6, t7, t8, s;
  fp2_sqr<C>(t0, a.c1.c1);
  fp2_sqr<C>(t1, a.c0.c0);
  fp2_add<C>(s, a.c1.c1, a.c0.c0);
  fp2_norm<C>(s);
  fp2_sqr<C>(t6, s);
  fp2_sub<C>(t6, t6, t0);
  fp2_sub<C>(t6, t6, t1);  // 2 a.c1.c1 a.c0.c0
  fp2_sqr<C>(t2, a.c0.c2);
  fp2_sqr<C>(t3, a.c1.c0);
  fp2_add<C>(s, a.c0.c2, a.c1.c0);
  fp2_norm<C>(s);
  fp2_sqr<C>(t7, s);
  fp2_sub<C>(t7, t7, t2);
  fp2_sub<C>(t7, t7, t3);  // 2 a.c0.c2 a.c1.c0
  fp2_sqr<C>(t4, a.c1.c2);
  fp2_sqr<C>(t5, a.c0.c1);
  fp2_add<C>(s, a.c1.c2, a.c0.c1);
  fp2_norm<C>(s);
  fp2_sqr<C>(t8, s);
  fp2_sub<C>(t8, t8, t4);
  fp2_sub<C>(t8, t8, t5);
  fp2_mul_xi<C>(t8, t8);  // 2 xi a.c1.c2 a.c0.c1
  fp2_mul_xi<C>(t0, t0);
  fp2_add<C>(t0, t0, t1);  // xi a.c1.c1^2 + a.c0.c0^2
  fp2_mul_xi<C>(t2, t2);
  fp2_add<C>(t2, t2, t3);  // xi a.c0.c2^2 + a.c1.c0^2
  fp2_mul_xi<C>(t4, t4);
  fp2_add<C>(t4, t4, t5);  // xi a.c1.c2^2 + a.c0.c1^2
  // weights 3 (t0, t2, t4, t6, t7) and 6 (t8): one propagation each before the 3 t -/+ 2 a combinations (5)
  fp2_norm<C>(t0);
  fp2_norm<C>(t2);
  fp2_norm<C>(t4);
  fp2_norm<C>(t6);
  fp2_norm<C>(t7);
  fp2_norm<C>(t8);
  Fp12<C, E2> o;
  // z = 3 t - 2 a (c0 part), 3 t + 2 a (c1 part)
  fp2_sub<C>(s, t0, a.c0.c0);
  fp2_dbl<C>(s, s);
  fp2_add<C>(o.c0.c0, s, t0);
  fp2_sub<C>(s, t2, a.c0.c1);
  fp2_dbl<C>(s, s);
  fp2_add<C>(o.c0.c1, s, t2);
  fp2_sub<C>(s, t4, a.c0.c2);
  fp2_dbl<C>(s, s);
  fp2_add<C>(o.c0.c2, s, t4);
  fp2_add<C>(s, t8, a.c1.c0);
  fp2_dbl<C>(s, s);
  fp2_add<C>(o.c1.c0, s, t8);
  fp2_add<C>(s, t6, a.c1.c1);
  fp2_dbl<C>(s, s);
  fp2_add<C>(o.c1.c1, s, t6);
  fp2_add<C>(s, t7, a.c1.c2);
  fp2_dbl<C>(s, s);
  fp2_add<C>(o.c1.c2, s, t7);
  // the +-2 a terms are linear in the input: reduce mod p (not only carry-propagate) so that a chain of squarings
  // does not double the value each step
  fp2_reduce<C>(o.c0.c0);
  fp2_reduce<C>(o.c0.c1);
  fp2_reduce<C>(o.c0.c2);
  fp2_reduce<C>(o.c1.c0);
  fp2_reduce<C>(o.c1.c1);
  fp2_reduce<C>(o.c1.c2);
  r = o;
}

// f *= (c0 + c1 v + c4 v w)   -- line of an M-twist curve (BLS12-381)
template <class C, class E2>
MLHIP_HD void fp12_mul_by_014_i(Fp12<C, E2>& f, const E2& c0, const E2& c1, const E2& c4) {
  Fp6<C, E2> t0, t1, s, x;
  E2 d;
  fp6_mul_by_01<C>(t0, f.c0, c0, c1);  // raw: 3, 3, 2
  fp6_mul_by_1<C>(t1, f.c1, c4);       // raw: 2, 1, 1
  fp6_add<C>(s, f.c0, f.c1);
  fp6_norm<C>(s);  // 2 -> 1
  fp2_add<C>(d, c1, c4);
  fp2_norm<C>(d);
  fp6_mul_by_01<C>(x, s, c0, d);  // raw: 3, 3, 2
  fp6_sub<C>(x, x, t0);
  fp6_sub<C>(f.c1, x, t1);
  fp6_reduce<C>(f.c1);  // 8, 7, 5
  fp6_mul_v<C>(t1, t1);
  fp6_add<C>(f.c0, t0, t1);
  fp6_reduce<C>(f.c0);  // 5, 5, 3
}
template <class C, class E2>
MLHIP_HD_NOINLINE void fp12_mul_by_014(Fp12<C, E2>& f, const E2& c0, const E2& c1, const E2& c4) {
  fp12_mul_by_014_i<C>(f, c0, c1, c4);
}
// f = f^2 (c0 + c1 v + c4 v w): one Miller-loop iteration's squaring and line product in ONE out-of-line function -- f
// crosses memory once per iteration instead of twice (the square stays in registers / the function's own frame between
// the two bodies).  Same operations in the same order as fp12_sqr followed by fp12_mul_by_014.
template <class C, class E2>
MLHIP_HD_NOINLINE void fp12_sqr_mul_by_014(Fp12<C, E2>& f, const E2& c0, const E2& c1, const E2& c4) {
  Fp12<C, E2> g;
  fp12_sqr_i<C>(g, f);
  fp12_mul_by_014_i<C>(g, c0, c1, c4);
  f = g;
}

// f *= (c0 + c3 w + c4 v w)   -- line of a D-twist curve (BN254, BLS12-377)
template <class C, class E2>
MLHIP_HD void fp12_mul_by_034_i(Fp12<C, E2>& f, const E2& c0, const E2& c3, const E2& c4) {
  Fp6<C, E2> t0, t1, s, x;
  E2 d;
  fp6_mul_by_0<C>(t0, f.c0, c0);       // 1, 1, 1
  fp6_mul_by_01<C>(t1, f.c1, c3, c4);  // raw: 3, 3, 2
  fp6_add<C>(s, f.c0, f.c1);
  fp6_norm<C>(s);
  fp2_add<C>(d, c0, c3);
  fp2_norm<C>(d);
  fp6_mul_by_01<C>(x, s, d, c4);  // raw: 3, 3, 2
  fp6_sub<C>(x, x, t0);
  fp6_sub<C>(f.c1, x, t1);
  fp6_reduce<C>(f.c1);  // 7, 7, 5
  fp6_mul_v<C>(t1, t1);
  fp6_add<C>(f.c0, t0, t1);
  fp6_reduce<C>(f.c0);  // 5, 4, 4
}
template <class C, class E2>
MLHIP_HD_NOINLINE void fp12_mul_by_034(Fp12<C, E2>& f, const E2& c0, const E2& c3, const E2& c4) {
  fp12_mul_by_034_i<C>(f, c0, c3, c4);
}
// f = f^2 (c0 + c3 w + c4 v w): see fp12_sqr_mul_by_014
template <class C, class E2>
MLHIP_HD_NOINLINE void fp12_sqr_mul_by_034(Fp12<C, E2>& f, const E2& c0, const E2& c3, const E2& c4) {
  Fp12<C, E2> g;
  fp12_sqr_i<C>(g, f);
  fp12_mul_by_034_i<C>(g, c0, c3, c4);
  f = g;
}

}  // namespace mlhip
