// pairing.h -- optimal-ate Miller loop and final exponentiation (BN254, BLS12-381, BLS12-377), gfx950.
//
// Replaces, on the GPU path, the calls the reference's drivers make into gnark-crypto:
//   MillerLoop            driver/gurvy/bls12381/bls12-381.go:449,458 ; bn254.go:248,257 ; bls12-377.go:245,254
//   FinalExponentiation   driver/gurvy/bls12381/bls12-381.go:467     ; bn254.go:266     ; bls12-377.go:263
// and kilic's Engine.AddPair/Result (driver/kilic/bls12-381.go:260-267), which is the composition.
//
// Contract: the raw Miller-loop value is only defined up to factors the final
// exponentiation kills; what is bit-exact is final_exp(miller_loop(..)) = f^(k (p^12-1)/r) with
// k = 3 (BLS12, Hayashida-Hayasaka-Teruya chain) and k = 2x(6x^2+3x+1) (BN254, Fuentes-Castaneda),
// the cofactors gnark and kilic use.  Checked against oracle/pyref.py's plain pow().
//
// Formulas: homogeneous projective doubling/addition on the twist with line coefficients
// (Costello-Lange-Naehrig), lines multiplied in sparsely (mul_by_014 for the M-twist, mul_by_034
// for D-twists), cyclotomic squarings in the hard part.
#pragma once
#include "ec.h"

namespace mlhip {

template <class C, class E2 = Fp2<C>>
struct G2Proj {
  E2 x, y, z;
};

template <class C, class E2 = Fp2<C>>
struct Line {
  E2 r0, r1, r2;  // r0 pairs with yP, r1 with xP, r2 is the constant coefficient
};

template <class C>
MLHIP_HD void fp_halve(Fp<C>& r, const Fp<C>& a) {
  constexpr int N = C::N;
  uint32_t mask = (uint32_t)0 - (a.l[0] & 1u);
  uint32_t t[N];
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    uint64_t s = (uint64_t)a.l[i] + (C::P[i] & mask) + c;
    t[i] = (uint32_t)s;
    c = s >> 32;
  }
#pragma unroll
  for (int i = 0; i < N - 1; i++) r.l[i] = (t[i] >> 1) | (t[i + 1] << 31);
  r.l[N - 1] = t[N - 1] >> 1;
}

template <class C>
MLHIP_HD void fp2_halve(Fp2<C>& r, const Fp2<C>& a) {
  fp_halve<C>(r.c0, a.c0);
  fp_halve<C>(r.c1, a.c1);
}

// T <- 2T, line through T,T
template <class C, class E2>
MLHIP_HD void g2_double_step(G2Proj<C, E2>& T, Line<C, E2>& l) {
  E2 A, B, Cc, E, F, G, H, I, J, EE, t, b3;
  fp2_mul<C>(A, T.x, T.y);
  fp2_halve<C>(A, A);
  fp2_sqr<C>(B, T.y);
  fp2_sqr<C>(Cc, T.z);
  if constexpr (C::ID == 1) {
    // BLS12-381: b' = 4 (1 + u) = 4 xi, so 3 b' Z^2 = 12 xi Z^2 -- additions instead of an Fp2 product
    fp2_mul_xi<C>(E, Cc);
    fp2_mul_small<C>(E, E, 12);
    (void)b3;
  } else {
    fp2_from_const<C>(b3, C::B3_TW);
    fp2_mul<C>(E, Cc, b3);  // 3 b' Z^2
  }
  // (carry-free element: T is normalized on entry and on exit; weights in the comments)
  fp2_reduce<C>(E);     // E = 3 b' Z^2 = 12 xi Z^2 is squared below: bring the VALUE back under p as well
  fp2_dbl<C>(F, E);
  fp2_add<C>(F, F, E);  // 3E: 3
  fp2_add<C>(G, B, F);  // 4
  fp2_halve<C>(G, G);
  fp2_norm<C>(G);
  fp2_add<C>(H, T.y, T.z);
  fp2_norm<C>(H);
  fp2_sqr<C>(H, H);
  fp2_add<C>(t, B, Cc);
  fp2_sub<C>(H, H, t);  // 2YZ: 3
  fp2_sub<C>(I, E, B);  // 2
  fp2_sqr<C>(J, T.x);
  fp2_sqr<C>(EE, E);
  // X3 = A (B - F) ; Y3 = G^2 - 3 EE ; Z3 = B H
  fp2_sub<C>(t, B, F);  // 4
  fp2_norm<C>(t);
  fp2_mul<C>(T.x, A, t);  // A has weight 2 (halved)
  fp2_sqr<C>(G, G);
  fp2_dbl<C>(t, EE);
  fp2_add<C>(t, t, EE);
  fp2_sub<C>(T.y, G, t);  // 4
  fp2_norm<C>(T.y);
  fp2_mul<C>(T.z, B, H);  // 2 x 1 x 3
  fp2_neg<C>(l.r0, H);
  fp2_dbl<C>(l.r1, J);
  fp2_add<C>(l.r1, l.r1, J);
  l.r2 = I;
}

// T <- T + Q (Q affine), line through T,Q
template <class C, class E2>
MLHIP_HD_NOINLINE void g2_add_step(G2Proj<C, E2>& T, const E2& qx, const E2& qy, Line<C, E2>& l) {
  E2 O, L, Cc, D, E, F, G, H, t, t2;
  fp2_mul<C>(t, qy, T.z);
  fp2_sub<C>(O, T.y, t);
  fp2_norm<C>(O);
  fp2_mul<C>(t, qx, T.z);
  fp2_sub<C>(L, T.x, t);
  fp2_norm<C>(L);
  fp2_sqr<C>(Cc, O);
  fp2_sqr<C>(D, L);
  fp2_mul<C>(E, L, D);
  fp2_mul<C>(F, T.z, Cc);
  fp2_mul<C>(G, T.x, D);
  fp2_dbl<C>(t, G);
  fp2_add<C>(H, E, F);
  fp2_sub<C>(H, H, t);  // 4
  fp2_norm<C>(H);
  fp2_mul<C>(t2, T.y, E);
  fp2_mul<C>(T.x, L, H);
  fp2_sub<C>(t, G, H);
  fp2_mul<C>(t, O, t);
  fp2_sub<C>(T.y, t, t2);
  fp2_norm<C>(T.y);
  fp2_mul<C>(T.z, T.z, E);
  // line
  fp2_mul<C>(t, L, qy);
  fp2_mul<C>(t2, qx, O);
  fp2_sub<C>(l.r2, t2, t);
  l.r0 = L;
  fp2_neg<C>(l.r1, O);
}

// EP: the type of P's coordinates -- Fp<C>, or Fp28<C> when E2 is the carry-free element
template <class C, class E2, class EP>
MLHIP_HD void mul_by_line(Fp12<C, E2>& f, const Line<C, E2>& l, const EP& px, const EP& py) {
  E2 a, b, c = l.r2;
  fp2_mul_fp<C>(a, l.r0, py);  // single products: any storable weight
  fp2_mul_fp<C>(b, l.r1, px);
  fp2_norm<C>(c);
  if (C::MTWIST)
    fp12_mul_by_014<C>(f, c, b, a);
  else
    fp12_mul_by_034<C>(f, a, b, c);
}
// f = f^2 * line in one out-of-line call (tower.h: fp12_sqr_mul_by_014 / _034): f crosses memory once, not twice
template <class C, class E2, class EP>
MLHIP_HD void sqr_mul_by_line(Fp12<C, E2>& f, const Line<C, E2>& l, const EP& px, const EP& py) {
  E2 a, b, c = l.r2;
  fp2_mul_fp<C>(a, l.r0, py);
  fp2_mul_fp<C>(b, l.r1, px);
  fp2_norm<C>(c);
  if (C::MTWIST)
    fp12_sqr_mul_by_014<C>(f, c, b, a);
  else
    fp12_sqr_mul_by_034<C>(f, a, b, c);
}

// f = prod_k f_{loop,Q_k}(P_k) over n_pairs pairs (shared squaring chain: the reference's Pairing2,
// driver/gurvy/bls12381/bls12-381.go:457-464).  Pairs flagged not live (one side at infinity) are skipped, as
// gnark does.  MAXP bounds n_pairs (state is kept per pair).  Coordinates: px/py in Fp, qx/qy in E2.
template <class C, int MAXP, class E2, class EP>
MLHIP_HD void miller_loop_core(Fp12<C, E2>& f, const EP* px, const EP* py, const E2* qx, const E2* qy,
                               const bool* live, int n_pairs) {
  G2Proj<C, E2> T[MAXP];
  int any = 0;
  for (int k = 0; k < n_pairs && k < MAXP; k++) {
    T[k].x = qx[k];
    T[k].y = qy[k];
    fp2_one<C>(T[k].z);
    any |= live[k];
  }
  fp12_one<C>(f);
  if (!any) return;
  Line<C, E2> l;
  bool first = true;
  for (int i = C::ATE_BITS - 2; i >= 0; i--) {
    // the iteration's squaring rides with the first live pair's line (the doubling step needs T only, so the line is
    // ready before f is touched): f^2 * line is one call -- same operations, same results
    bool square = !first;
    first = false;
    bool bit = (i >= 64) ? ((C::ATE_HI >> (i - 64)) & 1) : ((C::ATE_LO >> i) & 1);
    for (int k = 0; k < n_pairs && k < MAXP; k++) {
      if (!live[k]) continue;
      g2_double_step<C>(T[k], l);
      if (square)
        sqr_mul_by_line<C>(f, l, px[k], py[k]);
      else
        mul_by_line<C>(f, l, px[k], py[k]);
      square = false;
      if (bit) {
        g2_add_step<C>(T[k], qx[k], qy[k], l);
        mul_by_line<C>(f, l, px[k], py[k]);
      }
    }
  }
  if (C::IS_BN) {
    // lines through pi(Q) and -pi^2(Q)
    for (int k = 0; k < n_pairs && k < MAXP; k++) {
      if (!live[k]) continue;
      E2 x1, y1, x2, y2, g;
      fp2_conj<C>(x1, qx[k]);
      fp2_from_const<C>(g, C::GAMMA1[2]);
      fp2_mul<C>(x1, x1, g);
      fp2_conj<C>(y1, qy[k]);
      fp2_from_const<C>(g, C::GAMMA1[3]);
      fp2_mul<C>(y1, y1, g);
      fp2_mul_by_real_const<C>(x2, qx[k], C::GAMMA2[2]);
      fp2_mul_by_real_const<C>(y2, qy[k], C::GAMMA2[3]);
      fp2_neg<C>(y2, y2);
      g2_add_step<C>(T[k], x1, y1, l);
      mul_by_line<C>(f, l, px[k], py[k]);
      g2_add_step<C>(T[k], x2, y2, l);
      mul_by_line<C>(f, l, px[k], py[k]);
    }
  }
  if (C::X_NEG) fp12_conj<C>(f, f);
}

// one-element-per-lane entry point over the C-ABI point types
template <class C, int MAXP>
MLHIP_HD void miller_loop(Fp12<C>& f, const Affine<FpField<C>>* P, const Affine<Fp2Field<C>>* Q, int n_pairs) {
  Fp<C> px[MAXP], py[MAXP];
  Fp2<C> qx[MAXP], qy[MAXP];
  bool live[MAXP];
  for (int k = 0; k < n_pairs && k < MAXP; k++) {
    live[k] = !(affine_is_inf<FpField<C>>(P[k]) | affine_is_inf<Fp2Field<C>>(Q[k]));
    px[k] = P[k].x;
    py[k] = P[k].y;
    qx[k] = Q[k].x;
    qy[k] = Q[k].y;
  }
  miller_loop_core<C, MAXP, Fp2<C>, Fp<C>>(f, px, py, qx, qy, live, n_pairs);
}

// ---- Karabina's compressed cyclotomic squaring ----------------------------------------------------------------
// Write a cyclotomic element over Fp4 = Fp2[s]/(s^2 - xi), s = w^3:  f = A + B w + C w^2 with
// A = (c0.c0, c1.c1), B = (c1.c0, c0.c2), C = (c0.c1, c1.c2).  In the Granger-Scott squaring (fp12_cyclo_sqr) the
// new B and C depend on B and C only, so a chain of squarings can carry just these four Fp2 values: 6 Fp2
// squarings per step instead of 9.  A is recovered once at the end from the subgroup relation
//     a1 = (xi d1^2 + 3 d0^2 - 2 b1) / (4 b0)      [b0 = 0:  a1 = 2 d0 d1 / b1]
//     a0 = (2 a1^2 + b0 d1 - 3 b1 d0) xi + 1
// (K. Karabina, "Squaring in cyclotomic subgroups", 2013; checked numerically in this basis against
// oracle/pyref.py).  z^|x| = prod over the set bits of z^(2^i): every z^(2^i) comes out of ONE chain of compressed
// squarings and all of them are decompressed with one shared inversion (Montgomery's trick).  Worth it when |x|
// has few set bits (BLS12-381: 6, BLS12-377: 7); BN254's seed (27 set bits) keeps the plain chain.
template <class C, class E2>
struct CycloComp {
  E2 b0, b1, d0, d1;
};

template <class C, class E2>
MLHIP_HD void cyclo_sqr_compressed(CycloComp<C, E2>& k) {
  E2 t2, t3, t4, t5, t7, t8, s;
  fp2_sqr<C>(t2, k.b1);
  fp2_sqr<C>(t3, k.b0);
  fp2_add<C>(s, k.b0, k.b1);
  fp2_norm<C>(s);
  fp2_sqr<C>(t7, s);
  fp2_sub<C>(t7, t7, t2);
  fp2_sub<C>(t7, t7, t3);  // 2 b0 b1: 3
  fp2_sqr<C>(t4, k.d1);
  fp2_sqr<C>(t5, k.d0);
  fp2_add<C>(s, k.d0, k.d1);
  fp2_norm<C>(s);
  fp2_sqr<C>(t8, s);
  fp2_sub<C>(t8, t8, t4);
  fp2_sub<C>(t8, t8, t5);
  fp2_mul_xi<C>(t8, t8);  // 2 xi d0 d1: 6
  fp2_mul_xi<C>(t2, t2);
  fp2_add<C>(t2, t2, t3);  // xi b1^2 + b0^2: 3
  fp2_mul_xi<C>(t4, t4);
  fp2_add<C>(t4, t4, t5);  // xi d1^2 + d0^2: 3
  fp2_norm<C>(t2);
  fp2_norm<C>(t4);
  fp2_norm<C>(t7);
  fp2_norm<C>(t8);
  // d0' = 3 t2 - 2 d0 ; d1' = 3 t7 + 2 d1 ; b1' = 3 t4 - 2 b1 ; b0' = 3 t8 + 2 b0   (5 each)
  fp2_sub<C>(s, t2, k.d0);
  fp2_dbl<C>(s, s);
  fp2_add<C>(k.d0, s, t2);
  fp2_add<C>(s, t7, k.d1);
  fp2_dbl<C>(s, s);
  fp2_add<C>(k.d1, s, t7);
  fp2_sub<C>(s, t4, k.b1);
  fp2_dbl<C>(s, s);
  fp2_add<C>(k.b1, s, t4);
  fp2_add<C>(s, t8, k.b0);
  fp2_dbl<C>(s, s);
  fp2_add<C>(k.b0, s, t8);
  // the -+2 x terms are linear in the input: reduce mod p (fp2_reduce), or the value would double every step
  fp2_reduce<C>(k.d0);
  fp2_reduce<C>(k.d1);
  fp2_reduce<C>(k.b1);
  fp2_reduce<C>(k.b0);
}

// numerator and denominator of a1 for one compressed value (both formulas, selected by b0 = 0 / b1 = 0)
template <class C, class E2>
MLHIP_HD void cyclo_a1_fraction(E2& num, E2& den, const CycloComp<C, E2>& k) {
  E2 n1, n2, t, one;
  fp2_sqr<C>(t, k.d1);
  fp2_mul_xi<C>(n1, t);
  fp2_sqr<C>(t, k.d0);
  fp2_add<C>(n1, n1, t);
  fp2_dbl<C>(t, t);
  fp2_add<C>(n1, n1, t);  // xi d1^2 + 3 d0^2
  fp2_dbl<C>(t, k.b1);
  fp2_sub<C>(n1, n1, t);
  fp2_mul<C>(n2, k.d0, k.d1);
  fp2_dbl<C>(n2, n2);
  const bool z0 = fp2_is_zero<C>(k.b0), z1 = fp2_is_zero<C>(k.b1);
  fp2_select<C>(num, z0, n2, n1);
  fp2_dbl<C>(t, k.b0);
  fp2_dbl<C>(t, t);  // 4 b0
  fp2_one<C>(one);
  fp2_select<C>(den, z0, k.b1, t);
  fp2_select<C>(den, z0 & z1, one, den);  // f = A + C w^2 with B = 0 (e.g. f = 1): a1 = 0 / 1
  fp2_norm<C>(num);  // 7
  fp2_norm<C>(den);  // 4
}

template <class C, class E2>
MLHIP_HD void cyclo_decompress(Fp12<C, E2>& r, const CycloComp<C, E2>& k, const E2& a1) {
  E2 t, u, one;
  fp2_sqr<C>(t, a1);
  fp2_dbl<C>(t, t);
  fp2_mul<C>(u, k.b0, k.d1);
  fp2_add<C>(t, t, u);
  fp2_mul<C>(u, k.b1, k.d0);
  fp2_sub<C>(t, t, u);
  fp2_dbl<C>(u, u);
  fp2_sub<C>(t, t, u);  // 2 a1^2 + b0 d1 - 3 b1 d0: 6
  fp2_norm<C>(t);
  fp2_mul_xi<C>(t, t);
  fp2_one<C>(one);
  fp2_add<C>(r.c0.c0, t, one);
  fp2_reduce<C>(r.c0.c0);  // value bound 13
  r.c1.c1 = a1;
  r.c1.c0 = k.b0;
  r.c0.c2 = k.b1;
  r.c0.c1 = k.d0;
  r.c1.c2 = k.d1;
}

constexpr int mlhip_popcount64(uint64_t v) {
  int n = 0;
  while (v) {
    n += (int)(v & 1);
    v >>= 1;
  }
  return n;
}

// z^|x| by cyclotomic squarings (z in the cyclotomic subgroup), conjugated when the seed is negative
template <class C, class E2>
MLHIP_HD_NOINLINE void fp12_expt(Fp12<C, E2>& r, const Fp12<C, E2>& z) {
  int top = 63;
  while (!((C::X_ABS >> top) & 1)) top--;
  constexpr int NSET = mlhip_popcount64(C::X_ABS);
  if constexpr (NSET <= 8) {
    // one chain of compressed squarings; the values at the set bits are kept, decompressed together, multiplied
    constexpr int NS = NSET - (int)(C::X_ABS & 1);  // saved compressed values (bit 0 is z itself)
    CycloComp<C, E2> k, saved[NS > 0 ? NS : 1];
    k.b0 = z.c1.c0;
    k.b1 = z.c0.c2;
    k.d0 = z.c0.c1;
    k.d1 = z.c1.c2;
    int ns = 0;
    for (int i = 1; i <= top; i++) {
      cyclo_sqr_compressed<C>(k);
      if ((C::X_ABS >> i) & 1) saved[ns++] = k;
    }
    // shared inversion of the NS denominators
    E2 num[NS > 0 ? NS : 1], den[NS > 0 ? NS : 1], pre[NS > 0 ? NS : 1], inv, t;
    for (int j = 0; j < NS; j++) {
      cyclo_a1_fraction<C>(num[j], den[j], saved[j]);
      if (j == 0)
        pre[0] = den[0];
      else
        fp2_mul<C>(pre[j], pre[j - 1], den[j]);
    }
    Fp12<C, E2> acc, v;
    bool have = false;
    if (C::X_ABS & 1) {
      acc = z;
      have = true;
    }
    if (NS > 0) {
      fp2_inv<C>(inv, pre[NS - 1]);
      for (int j = NS - 1; j >= 0; j--) {
        E2 dj_inv;
        if (j > 0) {
          fp2_mul<C>(dj_inv, inv, pre[j - 1]);
          fp2_mul<C>(inv, inv, den[j]);
        } else {
          dj_inv = inv;
        }
        fp2_mul<C>(t, num[j], dj_inv);  // a1
        cyclo_decompress<C>(v, saved[j], t);
        if (have) {
          fp12_mul<C>(acc, acc, v);
        } else {
          acc = v;
          have = true;
        }
      }
    }
    if (C::X_NEG) fp12_conj<C>(acc, acc);
    r = acc;
  } else {
    Fp12<C, E2> acc = z;
    for (int i = top - 1; i >= 0; i--) {
      fp12_cyclo_sqr<C>(acc, acc);
      if ((C::X_ABS >> i) & 1) fp12_mul<C>(acc, acc, z);
    }
    if (C::X_NEG) fp12_conj<C>(acc, acc);
    r = acc;
  }
}

// r = f^(k (p^12 - 1)/r_order): the reference's FExp (bls12-381.go:466-468 etc.)
template <class C, class E2>
MLHIP_HD void final_exp(Fp12<C, E2>& out, const Fp12<C, E2>& f) {
  Fp12<C, E2> r, t0, t1, t2;
  // easy part: f^((p^6-1)(p^2+1))
  fp12_conj<C>(t0, f);
  fp12_inv<C>(t1, f);
  fp12_mul<C>(t0, t0, t1);
  fp12_frob<C, 2>(t1, t0);
  fp12_mul<C>(r, t1, t0);
  if (!C::IS_BN) {
    // hard part, exponent (x-1)^2 (x+p) (x^2+p^2-1) + 3 = 3 (p^4-p^2+1)/r
    fp12_cyclo_sqr<C>(t0, r);
    fp12_expt<C>(t1, r);
    fp12_conj<C>(t2, r);
    fp12_mul<C>(t1, t1, t2);  // r^(x-1)
    fp12_expt<C>(t2, t1);     // r^((x-1)x)
    fp12_conj<C>(t1, t1);
    fp12_mul<C>(t1, t1, t2);  // r^((x-1)^2)
    fp12_expt<C>(t2, t1);     // ^x
    fp12_frob<C, 1>(t1, t1);  // ^p
    fp12_mul<C>(t1, t1, t2);  // r^((x-1)^2 (x+p))
    fp12_mul<C>(r, r, t0);    // r^3
    fp12_expt<C>(t0, t1);     // ^x
    fp12_expt<C>(t2, t0);     // ^x^2
    fp12_frob<C, 2>(t0, t1);  // ^p^2
    fp12_conj<C>(t1, t1);     // ^-1
    fp12_mul<C>(t1, t1, t2);
    fp12_mul<C>(t1, t1, t0);  // r^((x-1)^2 (x+p)(x^2+p^2-1))
    fp12_mul<C>(out, r, t1);
  } else {
    // hard part, exponent l0 + l1 p + l2 p^2 + l3 p^3 = 2x(6x^2+3x+1) (p^4-p^2+1)/r
    //   a = 12x^3+6x^2+6x ; b = a - 2x ; l0 = a + 6x^2 + 1 ; l1 = b ; l2 = a ; l3 = b - 1
    Fp12<C, E2> fx, f2x, f6x, f6x2, f12x3, a, b;
    fp12_expt<C>(fx, r);
    fp12_cyclo_sqr<C>(f2x, fx);
    fp12_cyclo_sqr<C>(t0, f2x);  // 4x
    fp12_mul<C>(f6x, t0, f2x);
    fp12_expt<C>(f6x2, f6x);
    fp12_cyclo_sqr<C>(t0, f6x2);  // 12x^2
    fp12_expt<C>(f12x3, t0);
    fp12_mul<C>(a, f12x3, f6x2);
    fp12_mul<C>(a, a, f6x);
    fp12_conj<C>(t0, f2x);
    fp12_mul<C>(b, a, t0);
    // result = (a f6x2 r) * frob(b) * frob2(a) * frob3(b conj(r))
    fp12_mul<C>(t0, a, f6x2);
    fp12_mul<C>(t0, t0, r);
    fp12_frob<C, 1>(t1, b);
    fp12_mul<C>(t0, t0, t1);
    fp12_frob<C, 2>(t1, a);
    fp12_mul<C>(t0, t0, t1);
    fp12_conj<C>(t1, r);
    fp12_mul<C>(t1, b, t1);
    fp12_frob<C, 3>(t2, t1);
    fp12_mul<C>(out, t0, t2);
  }
}

}  // namespace mlhip
