// pairing_quad.h -- the pairings (BLS12-381; since round 4 BLS12-377: D-twist line, u^2 = -5, and BN254: 10 limbs, xi = 9 + u,
// the two Frobenius lines, the Fuentes-Castaneda hard part on plain squarings) with ONE PAIRING PER QUAD OF LANES (carry-free element, fp2_lanes28.h).
//
// Layout.  Lane pairs stay what they are in fp2_lanes28.h: lane 2i holds the real and lane 2i+1 the imaginary part of
// an Fp2 value, and every fp2_* / fp6_* function of tower.h works on a lane pair without knowing its neighbours -- so in
// a quad the SAME call computes two independent Fp2 / Fp6 results, one in pair A (lanes 0, 1) and one in pair B (lanes
// 2, 3).  An Fp12 value f = c0 + c1 w is stored as ONE Fp6-shaped object: pair A holds c0, pair B holds c1 (42 words per
// lane instead of 84).  The Fp12 formulas then read
//     f^2   : one Fp6 product        (A: (a0 + a1)(a0 + v a1),  B: a0 a1)                 instead of two in a row
//     f g   : one Fp6 product + three rounds (Karatsuba: m0 | m1, then m2 split over the pairs)  instead of three products
//     f * l : fp6_mul_by_01 + fp6_mul_by_1 (8 Fp2 products deep)                          instead of 13
//     compressed cyclotomic squaring: (b0 | d0), (b1 | d1) -- three Fp2 squarings         instead of six
// with the halves exchanged by quad_perm [2,3,0,1] moves (14 per Fp2 value) and merged by per-pair selects.  The point
// T of the Miller loop and the line coefficients are replicated on both pairs; the doubling step spreads its nine Fp2
// products over the pairs (g2_double_step_q), the five addition steps run as they are (g2_add_step).  What this buys: the
// dependent chain of one pairing is 1.6 x shorter (measured: a batch that leaves the chip under-filled takes 5.2 ms
// instead of 8.5), and an Fp12 is 42 words per lane.  What it costs: 16 instead of 13 product slots per line, exchanges and
// per-pair selects around every round -- more instructions in total (65 536 pairings: 20.9 ms against the lane pairs'
// 18.6), so a full chip (issue bound) stays on lane pairs; and the Fp12 operations of the final exponentiation are still
// out-of-line functions whose operands cross scratch (DESIGN.md section 4 lists what is left to do).
//
// Every function is written over the element type E with four helpers (quad_swap, quad_sel_b, quad_on_a, quad_on_b):
// Fp2L28 on the device (DPP), Fp2Q28H on the host -- the host model of one quad, with the weight / value-bound checks of
// Fp2H28 -- so tests/test_host_math.py runs the whole pairing through the same code against the oracle.
#pragma once
#include "fp2_lanes28.h"
#include "pairing.h"

// the two Fp12-level functions of the Miller loop are inlined around f (measured: Miller loop of 65 536 pairs 11.2 ->
// 10.0 ms, 1 024 pairs 2.65 -> 2.56 ms; out of line their operands cross scratch on every call); MLHIP_Q28_OUTLINE keeps
// them out of line
#if defined(MLHIP_Q28_OUTLINE)
#define MLHIP_Q28_FN MLHIP_HD_NOINLINE
#else
#define MLHIP_Q28_FN MLHIP_HD
#endif

namespace mlhip {

// ---- quad helpers: device element --------------------------------------------------------------------------------------
template <class C>
MLHIP_HD bool quad_is_b(const Fp2L28<C>&) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (threadIdx.x & 2u) != 0;
#else
  return false;
#endif
}
template <int CTRL, class C>
MLHIP_HD void quad_perm28(Fp2L28<C>& r, const Fp2L28<C>& a) {
#pragma unroll
  for (int i = 0; i < C::N28; i++) {
#if defined(__HIP_DEVICE_COMPILE__)
    r.v.l[i] = __builtin_amdgcn_update_dpp(0, a.v.l[i], CTRL, 0xF, 0xF, true);  // every quad_perm source lane is valid
#else
    r.v.l[i] = a.v.l[i];
#endif
  }
}
// the other pair's value: quad_perm [2,3,0,1]
template <class C>
MLHIP_HD void quad_swap(Fp2L28<C>& r, const Fp2L28<C>& a) {
  quad_perm28<0x4E>(r, a);
}
// pair A's value on both pairs ([0,1,0,1]) / pair B's ([2,3,2,3])
template <class C>
MLHIP_HD void quad_on_a(Fp2L28<C>& r, const Fp2L28<C>& a) {
  quad_perm28<0x44>(r, a);
}
template <class C>
MLHIP_HD void quad_on_b(Fp2L28<C>& r, const Fp2L28<C>& a) {
  quad_perm28<0xEE>(r, a);
}
// r = (pair B ? b_val : a_val)
template <class C>
MLHIP_HD void quad_sel_b(Fp2L28<C>& r, const Fp2L28<C>& b_val, const Fp2L28<C>& a_val) {
  fp28_select<C>(r.v, quad_is_b(r), b_val.v, a_val.v);
}

#if !defined(__HIPCC__)
// ---- host model of one quad: pair A = c[0], c[1]; pair B = c[2], c[3] ------------------------------------------------
template <class C>
struct Fp2Q28H {
  Fp28<C> c[4];
  int wt = 1, vbound = 1;
  static constexpr int LANES = 4;
  Fp28<C>& at(int i) { return c[i]; }
  const Fp28<C>& at(int i) const { return c[i]; }
  static bool hi(int i) { return (i & 1) != 0; }
  int w() const { return wt; }
  void set_w(int x) { wt = x; }
  int vb() const { return vbound; }
  void set_vb(int x) {
    require(x <= 512, "a stored value (value bound)", x, 0);
    vbound = x;
  }
  static void require(bool ok, const char* what, int wa, int wb) { Fp2H28<C>::require(ok, what, wa, wb); }
};
template <class C>
inline void lp28_partner(Fp2Q28H<C>& r, const Fp2Q28H<C>& a) {
  const Fp2Q28H<C> t = a;
  r = t;
  r.c[0] = t.c[1];
  r.c[1] = t.c[0];
  r.c[2] = t.c[3];
  r.c[3] = t.c[2];
}
template <class C>
inline void lp28_real_on_both(Fp2Q28H<C>& r, const Fp2Q28H<C>& a) {
  const Fp2Q28H<C> t = a;
  r = t;
  r.c[1] = t.c[0];
  r.c[3] = t.c[2];
}
// a zero test is a branch: both pairs must agree (the formulas only test values that are replicated on the pairs)
template <class C>
inline bool lp28_both(const Fp2Q28H<C>&, const bool (&b)[4]) {
  const bool pa = b[0] && b[1], pb = b[2] && b[3];
  Fp2Q28H<C>::require(pa == pb, "a branch on a value that differs between the pairs of a quad", pa, pb);
  return pa;
}
template <class C>
inline void quad_swap(Fp2Q28H<C>& r, const Fp2Q28H<C>& a) {
  const Fp2Q28H<C> t = a;
  r = t;
  r.c[0] = t.c[2];
  r.c[1] = t.c[3];
  r.c[2] = t.c[0];
  r.c[3] = t.c[1];
}
template <class C>
inline void quad_on_a(Fp2Q28H<C>& r, const Fp2Q28H<C>& a) {
  const Fp2Q28H<C> t = a;
  r = t;
  r.c[2] = t.c[0];
  r.c[3] = t.c[1];
}
template <class C>
inline void quad_on_b(Fp2Q28H<C>& r, const Fp2Q28H<C>& a) {
  const Fp2Q28H<C> t = a;
  r = t;
  r.c[0] = t.c[2];
  r.c[1] = t.c[3];
}
template <class C>
inline void quad_sel_b(Fp2Q28H<C>& r, const Fp2Q28H<C>& b_val, const Fp2Q28H<C>& a_val) {
  Fp2Q28H<C> o = a_val;
  o.c[2] = b_val.c[2];
  o.c[3] = b_val.c[3];
  o.wt = a_val.wt > b_val.wt ? a_val.wt : b_val.wt;
  o.vbound = a_val.vbound > b_val.vbound ? a_val.vbound : b_val.vbound;
  r = o;
}
MLHIP_LP28_OVERLOADS(Fp2Q28H)
#endif

// ---- Fp6-shaped helpers -------------------------------------------------------------------------------------------------
template <class C, class E>
MLHIP_HD void fp6q_swap(Fp6<C, E>& r, const Fp6<C, E>& a) {
  quad_swap<C>(r.c0, a.c0);
  quad_swap<C>(r.c1, a.c1);
  quad_swap<C>(r.c2, a.c2);
}
template <class C, class E>
MLHIP_HD void fp6q_sel_b(Fp6<C, E>& r, const Fp6<C, E>& b_val, const Fp6<C, E>& a_val) {
  quad_sel_b<C>(r.c0, b_val.c0, a_val.c0);
  quad_sel_b<C>(r.c1, b_val.c1, a_val.c1);
  quad_sel_b<C>(r.c2, b_val.c2, a_val.c2);
}

// f = c0 + c1 w with c0 on pair A and c1 on pair B of every coefficient
template <class C, class E>
struct Fp12Q {
  Fp6<C, E> v;
};

// an Fp12 whose coefficients are replicated on both pairs <-> the quad form
template <class C, class E>
MLHIP_HD void fp12q_from_replicated(Fp12Q<C, E>& r, const Fp12<C, E>& a) {
  fp6q_sel_b<C>(r.v, a.c1, a.c0);
}
template <class C, class E>
MLHIP_HD void fp12q_to_replicated(Fp12<C, E>& r, const Fp12Q<C, E>& a) {
  quad_on_a<C>(r.c0.c0, a.v.c0);
  quad_on_a<C>(r.c0.c1, a.v.c1);
  quad_on_a<C>(r.c0.c2, a.v.c2);
  quad_on_b<C>(r.c1.c0, a.v.c0);
  quad_on_b<C>(r.c1.c1, a.v.c1);
  quad_on_b<C>(r.c1.c2, a.v.c2);
}

template <class C, class E>
MLHIP_HD void fp12q_one(Fp12Q<C, E>& r) {
  E one, zero;
  fp2_one<C>(one);
  fp2_zero<C>(zero);
  fp6_zero<C>(r.v);
  quad_sel_b<C>(r.v.c0, zero, one);
}
template <class C, class E>
MLHIP_HD void fp12q_conj(Fp12Q<C, E>& r, const Fp12Q<C, E>& a) {
  Fp6<C, E> n;
  fp6_neg<C>(n, a.v);
  fp6q_sel_b<C>(r.v, n, a.v);
}

// r = a b, Karatsuba over the pairs: m0 = a0 b0 | m1 = a1 b1 is ONE Fp6 product (six rounds), and the six Fp2 products of
// m2 = (a0 + a1)(b0 + b1) take three more rounds -- pair A computes s_j r_j, pair B the three products of sums:
//     c0 = m0 + v m1 (pair A)        c1 = m2 - m0 - m1 (pair B)
// nine rounds instead of the twelve of a * b and a * swap(b).
template <class C, class E>
MLHIP_HD_NOINLINE void fp12q_mul(Fp12Q<C, E>& r, const Fp12Q<C, E>& a, const Fp12Q<C, E>& b) {
  Fp6<C, E> t, ts, sa, sb, m2, u;
  fp6_mul_i<C>(t, a.v, b.v);  // A: m0 | B: m1
  fp6q_swap<C>(sa, a.v);
  fp6_add<C>(sa, a.v, sa);  // a0 + a1 on both pairs   (2)
  fp6_norm<C>(sa);
  fp6q_swap<C>(sb, b.v);
  fp6_add<C>(sb, b.v, sb);
  fp6_norm<C>(sb);
  {
    // the Karatsuba products of sa * sb: pair A takes x_j y_j, pair B (x_i + x_k)(y_i + y_k)
    E xs, ys, X, Y, p0, p1, p2, t0, t1, t2, u0, u1, u2, w;
    fp2_add<C>(xs, sa.c1, sa.c2);
    fp2_add<C>(ys, sb.c1, sb.c2);
    quad_sel_b<C>(X, xs, sa.c0);
    quad_sel_b<C>(Y, ys, sb.c0);
    fp2_mul<C>(p0, X, Y);  // x0 y0 | (x1 + x2)(y1 + y2)     (2 x 2 on pair B)
    fp2_add<C>(xs, sa.c0, sa.c1);
    fp2_add<C>(ys, sb.c0, sb.c1);
    quad_sel_b<C>(X, xs, sa.c1);
    quad_sel_b<C>(Y, ys, sb.c1);
    fp2_mul<C>(p1, X, Y);  // x1 y1 | (x0 + x1)(y0 + y1)
    fp2_add<C>(xs, sa.c0, sa.c2);
    fp2_add<C>(ys, sb.c0, sb.c2);
    quad_sel_b<C>(X, xs, sa.c2);
    quad_sel_b<C>(Y, ys, sb.c2);
    fp2_mul<C>(p2, X, Y);  // x2 y2 | (x0 + x2)(y0 + y2)
    quad_on_a<C>(t0, p0);
    quad_on_b<C>(u0, p0);
    quad_on_a<C>(t1, p1);
    quad_on_b<C>(u1, p1);
    quad_on_a<C>(t2, p2);
    quad_on_b<C>(u2, p2);
    // m2 (on both pairs), as fp6_mul_i assembles it
    fp2_sub<C>(w, u0, t1);
    fp2_sub<C>(w, w, t2);
    fp2_mul_xi<C>(w, w);
    fp2_add<C>(m2.c0, w, t0);  // 7
    fp2_sub<C>(w, u1, t0);
    fp2_sub<C>(w, w, t1);
    fp2_mul_xi<C>(xs, t2);
    fp2_add<C>(m2.c1, w, xs);  // 5
    fp2_sub<C>(w, u2, t0);
    fp2_sub<C>(w, w, t2);
    fp2_add<C>(m2.c2, w, t1);  // 4
    fp6_norm<C>(m2);
  }
  fp6q_swap<C>(ts, t);  // A: m1 | B: m0
  fp6_mul_v<C>(u, ts);
  fp6_add<C>(u, t, u);  // A: m0 + v m1   (3, 2, 2)
  fp6_sub<C>(m2, m2, t);
  fp6_sub<C>(m2, m2, ts);  // m2 - m0 - m1   (3)
  fp6q_sel_b<C>(r.v, m2, u);
  fp6_reduce<C>(r.v);
}

// r = a^2, complex squaring: c0 = (a0 + a1)(a0 + v a1) - ab - v ab ; c1 = 2 ab -- ONE Fp6 product
template <class C, class E>
MLHIP_Q28_FN void fp12q_sqr(Fp12Q<C, E>& r, const Fp12Q<C, E>& a) {
  Fp6<C, E> as, s, y, X, Y, t, ts, u;
  fp6q_swap<C>(as, a.v);
  fp6_add<C>(s, a.v, as);  // a0 + a1 on both pairs   (2)
  fp6_norm<C>(s);
  fp6_mul_v<C>(y, as);
  fp6_add<C>(y, a.v, y);  // A: a0 + v a1   (3, 2, 2)
  fp6_norm<C>(y);
  fp6q_sel_b<C>(X, as, s);   // A: a0 + a1   | B: a0
  fp6q_sel_b<C>(Y, a.v, y);  // A: a0 + v a1 | B: a1
  fp6_mul_i<C>(t, X, Y);     // A: (a0 + a1)(a0 + v a1) | B: ab
  fp6q_swap<C>(ts, t);       // A: ab
  fp6_mul_v<C>(u, ts);
  fp6_sub<C>(u, t, u);
  fp6_sub<C>(u, u, ts);  // A: c0   (4, 3, 3)
  fp6_dbl<C>(t, t);      // B: c1   (2)
  fp6q_sel_b<C>(r.v, t, u);
  fp6_reduce<C>(r.v);
}

// r = 1 / a = conj(a) / (a0^2 - v a1^2)
template <class C, class E>
MLHIP_HD_NOINLINE void fp12q_inv(Fp12Q<C, E>& r, const Fp12Q<C, E>& a) {
  Fp6<C, E> s, ss, t, ts, m, n;
  fp6_sqr<C>(s, a.v);  // A: a0^2 | B: a1^2
  fp6q_swap<C>(ss, s);
  fp6_mul_v<C>(ss, ss);
  fp6_sub<C>(t, s, ss);  // A: a0^2 - v a1^2
  fp6q_swap<C>(ts, t);
  fp6q_sel_b<C>(t, ts, t);  // ... on both pairs
  fp6_reduce<C>(t);
  fp6_inv<C>(t, t);
  fp6_mul<C>(m, a.v, t);  // A: a0 / n | B: a1 / n
  fp6_neg<C>(n, m);
  fp6q_sel_b<C>(r.v, n, m);
  if constexpr (C::BETA != -1) fp6_reduce<C>(r.v);  // u^2 = -5: un-reduced Fp6 products may not travel on (as fp12_inv, tower.h)
}

// Frobenius f -> f^(p^K): w-basis positions g0 = c0.c0, g1 = c1.c0, g2 = c0.c1, g3 = c1.c1, g4 = c0.c2, g5 = c1.c2 --
// coefficient j of the quad form holds g(2j) on pair A and g(2j+1) on pair B
template <class C, int K, class E>
MLHIP_HD_NOINLINE void fp12q_frob(Fp12Q<C, E>& r, const Fp12Q<C, E>& a) {
  const E* src[3] = {&a.v.c0, &a.v.c1, &a.v.c2};
  E* dst[3] = {&r.v.c0, &r.v.c1, &r.v.c2};
#pragma unroll
  for (int j = 0; j < 3; j++) {
    E x, ga, gb, g, ra, rb;
    if (K & 1)
      fp2_conj<C>(x, *src[j]);
    else
      x = *src[j];
    if (K == 2) {
      // gamma2[i] are 6th roots of unity in Fp: two single products, one kept per pair
      if (j == 0)
        ra = x;
      else
        fp2_mul_by_real_const<C>(ra, x, C::GAMMA2[2 * j]);
      fp2_mul_by_real_const<C>(rb, x, C::GAMMA2[2 * j + 1]);
      quad_sel_b<C>(*dst[j], rb, ra);
    } else {
      if (j == 0)
        fp2_one<C>(ga);
      else if (K == 1)
        fp2_from_const<C>(ga, C::GAMMA1[2 * j]);
      else
        fp2_from_const<C>(ga, C::GAMMA3[2 * j]);
      if (K == 1)
        fp2_from_const<C>(gb, C::GAMMA1[2 * j + 1]);
      else
        fp2_from_const<C>(gb, C::GAMMA3[2 * j + 1]);
      quad_sel_b<C>(g, gb, ga);
      fp2_mul<C>(*dst[j], x, g);
    }
  }
}

// f *= (c0 + c1 v + c4 v w), the line of an M-twist curve; c0, c1, c4 replicated on both pairs
template <class C, class E>
MLHIP_Q28_FN void fp12q_mul_by_014(Fp12Q<C, E>& f, const E& c0, const E& c1, const E& c4) {
  Fp6<C, E> t0, t1, fs, u;
  fp6_mul_by_01<C>(t0, f.v, c0, c1);  // A: f0 (c0 + c1 v) | B: f1 (c0 + c1 v)    raw: 3, 3, 2
  fp6q_swap<C>(fs, f.v);
  fp6_mul_by_1<C>(t1, fs, c4);  // A: f1 c4 v | B: f0 c4 v                          raw: 2, 1, 1
  fp6_mul_v<C>(u, t1);
  fp6_add<C>(u, t0, u);    // A: f0 l0 + v (f1 c4 v)                                 5, 5, 3
  fp6_add<C>(t0, t0, t1);  // B: f1 l0 + f0 c4 v                                     5, 4, 3
  fp6q_sel_b<C>(f.v, t0, u);
  fp6_reduce<C>(f.v);
}

// f *= (c0 + c3 w + c4 v w), the line of a D-twist curve (BLS12-377; round 4); c0, c3, c4 replicated on both pairs:
//     pair A: f0 c0 + v f1 (c3 + c4 v)        pair B: f1 c0 + f0 (c3 + c4 v)
// fp6_mul_by_0 on (f0 | f1) and fp6_mul_by_01 on the swapped halves (f1 | f0): 3 + 5 Fp2 products deep, as the M-twist form
template <class C, class E>
MLHIP_Q28_FN void fp12q_mul_by_034(Fp12Q<C, E>& f, const E& c0, const E& c3, const E& c4) {
  Fp6<C, E> t0, t1, fs, u;
  fp6_mul_by_0<C>(t0, f.v, c0);  // A: f0 c0 | B: f1 c0                                  1, 1, 1
  fp6q_swap<C>(fs, f.v);
  fp6_mul_by_01<C>(t1, fs, c3, c4);  // A: f1 (c3 + c4 v) | B: f0 (c3 + c4 v)            raw: 3, 3, 2
  fp6_mul_v<C>(u, t1);
  fp6_add<C>(u, t0, u);    // A: f0 c0 + v f1 (c3 + c4 v)
  fp6_add<C>(t0, t0, t1);  // B: f1 c0 + f0 (c3 + c4 v)
  fp6q_sel_b<C>(f.v, t0, u);
  fp6_reduce<C>(f.v);
}

template <class C, class E, class EP>
MLHIP_HD void mul_by_line_q(Fp12Q<C, E>& f, const Line<C, E>& l, const EP& px, const EP& py) {
  E a, b, c = l.r2;
  fp2_mul_fp<C>(a, l.r0, py);
  fp2_mul_fp<C>(b, l.r1, px);
  fp2_norm<C>(c);
  if constexpr (C::MTWIST)
    fp12q_mul_by_014<C>(f, c, b, a);
  else
    fp12q_mul_by_034<C>(f, a, b, c);
}

// T <- 2T with the line through T, T: the formulas of g2_double_step (pairing.h) with the nine Fp2 products spread over the
// two pairs of the quad -- three squaring rounds and two product rounds instead of six and three in a row:
//     (Y^2 | Z^2)    ((Y + Z)^2 | X^2)    X Y on both    (E^2 | G^2)    (A (B - F) | B H)
// T, the line and every linear combination are replicated on both pairs (quad_on_a / quad_on_b pick a round's results).
template <class C, class E2>
MLHIP_HD void g2_double_step_q(G2Proj<C, E2>& T, Line<C, E2>& l) {
  E2 A, B, Cc, E, F, G, H, I, J, EE, GG, s, t, u, r1, r2;
  fp2_add<C>(s, T.y, T.z);
  fp2_norm<C>(s);
  quad_sel_b<C>(t, T.z, T.y);
  fp2_sqr<C>(r1, t);  // Y^2 | Z^2
  quad_sel_b<C>(t, T.x, s);
  fp2_sqr<C>(r2, t);  // (Y + Z)^2 | X^2
  fp2_mul<C>(A, T.x, T.y);
  fp2_halve<C>(A, A);
  quad_on_a<C>(B, r1);
  quad_on_b<C>(Cc, r1);
  quad_on_a<C>(H, r2);
  quad_on_b<C>(J, r2);
  if constexpr (C::ID == 1) {
    // BLS12-381: b' = 4 (1 + u) = 4 xi, so 3 b' Z^2 = 12 xi Z^2 -- additions instead of an Fp2 product
    fp2_mul_xi<C>(E, Cc);
    fp2_mul_small<C>(E, E, 12);
  } else {
    E2 b3;  // (Cc is replicated on both pairs: so is this product)
    fp2_from_const<C>(b3, C::B3_TW);
    fp2_mul<C>(E, Cc, b3);
  }
  fp2_reduce<C>(E);  // 3 b' Z^2: squared below
  fp2_dbl<C>(F, E);
  fp2_add<C>(F, F, E);  // 3E
  fp2_add<C>(G, B, F);
  fp2_halve<C>(G, G);
  fp2_norm<C>(G);
  fp2_add<C>(t, B, Cc);
  fp2_sub<C>(H, H, t);  // 2YZ   (3)
  fp2_sub<C>(I, E, B);  // 2
  quad_sel_b<C>(t, G, E);
  fp2_sqr<C>(r1, t);  // E^2 | G^2
  fp2_sub<C>(t, B, F);
  fp2_norm<C>(t);
  quad_sel_b<C>(u, B, A);  // A | B
  fp2_norm<C>(H);
  quad_sel_b<C>(t, H, t);  // B - F | H
  fp2_mul<C>(r2, u, t);    // X3 = A (B - F) | Z3 = B H
  quad_on_a<C>(EE, r1);
  quad_on_b<C>(GG, r1);
  quad_on_a<C>(T.x, r2);
  quad_on_b<C>(T.z, r2);
  fp2_dbl<C>(t, EE);
  fp2_add<C>(t, t, EE);
  fp2_sub<C>(T.y, GG, t);  // G^2 - 3 E^2   (4)
  fp2_norm<C>(T.y);
  fp2_neg<C>(l.r0, H);
  fp2_dbl<C>(l.r1, J);
  fp2_add<C>(l.r1, l.r1, J);
  l.r2 = I;
}

// f = prod_k f_{loop,Q_k}(P_k) over n_pairs pairs sharing the squarings (the reference's Pairing / Pairing2,
// driver/gurvy/bls12381/bls12-381.go:448-464); qx, qy, the points T_k and the lines are replicated on both pairs of the
// quad.  Pairs flagged not live (one side at infinity) are skipped, as gnark does.
template <class C, int MAXP, class E, class EP>
MLHIP_HD void miller_loop_q(Fp12Q<C, E>& f, const EP* px, const EP* py, const E* qx, const E* qy, const bool* live,
                            int n_pairs) {
  G2Proj<C, E> T[MAXP];
  int any = 0;
  for (int k = 0; k < n_pairs && k < MAXP; k++) {
    T[k].x = qx[k];
    T[k].y = qy[k];
    fp2_one<C>(T[k].z);
    any |= live[k];
  }
  fp12q_one<C>(f);
  if (!any) return;
  Line<C, E> l;
  bool first = true;
  for (int i = C::ATE_BITS - 2; i >= 0; i--) {
    if (!first) fp12q_sqr<C>(f, f);
    first = false;
    const bool bit = (i >= 64) ? ((C::ATE_HI >> (i - 64)) & 1) : ((C::ATE_LO >> i) & 1);
    for (int k = 0; k < n_pairs && k < MAXP; k++) {
      if (!live[k]) continue;
      g2_double_step_q<C>(T[k], l);
      mul_by_line_q<C>(f, l, px[k], py[k]);
      if (bit) {
        g2_add_step<C>(T[k], qx[k], qy[k], l);
        mul_by_line_q<C>(f, l, px[k], py[k]);
      }
    }
  }
  if constexpr (C::IS_BN) {
    // BN254 (round 4): the lines through pi(Q) and -pi^2(Q), as miller_loop_core (pairing.h); everything replicated on the pairs
    for (int k = 0; k < n_pairs && k < MAXP; k++) {
      if (!live[k]) continue;
      E x1, y1, x2, y2, g;
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
      mul_by_line_q<C>(f, l, px[k], py[k]);
      g2_add_step<C>(T[k], x2, y2, l);
      mul_by_line_q<C>(f, l, px[k], py[k]);
    }
  }
  if (C::X_NEG) fp12q_conj<C>(f, f);
}

// ---- Karabina's compressed squarings on the quad: P = (b0 | d0), Q = (b1 | d1) --------------------------------------------
template <class C, class E>
struct CycloCompQ {
  E p, q;
};
template <class C, class E>
MLHIP_HD void cyclo_compress_q(CycloCompQ<C, E>& k, const Fp12Q<C, E>& z) {
  // b0 = c1.c0 (pair B of coefficient 0), d0 = c0.c1 (pair A of coefficient 1); b1 = c0.c2, d1 = c1.c2: coefficient 2 as it is
  E t;
  quad_sel_b<C>(t, z.v.c0, z.v.c1);  // (d0 | b0)
  quad_swap<C>(k.p, t);
  k.q = z.v.c2;
}
template <class C, class E>
MLHIP_HD void cyclo_sqr_compressed_q(CycloCompQ<C, E>& k) {
  E s0, s1, s, cr, x, t, m, n, xs, cs, np;
  fp2_sqr<C>(s0, k.p);  // b0^2 | d0^2
  fp2_sqr<C>(s1, k.q);  // b1^2 | d1^2
  fp2_add<C>(s, k.p, k.q);
  fp2_norm<C>(s);
  fp2_sqr<C>(s, s);
  fp2_sub<C>(cr, s, s0);
  fp2_sub<C>(cr, cr, s1);  // 2 b0 b1 | 2 d0 d1   (3)
  fp2_mul_xi<C>(t, cr);
  quad_sel_b<C>(cr, t, cr);  // t7 = 2 b0 b1 | t8 = 2 xi d0 d1   (<= 6)
  fp2_mul_xi<C>(x, s1);
  fp2_add<C>(x, x, s0);  // t2 = xi b1^2 + b0^2 | t4 = xi d1^2 + d0^2   (3)
  fp2_norm<C>(cr);
  fp2_norm<C>(x);
  quad_swap<C>(xs, x);   // t4 | t2
  quad_swap<C>(cs, cr);  // t8 | t7
  // b0' = 3 t8 + 2 b0 | d0' = 3 t2 - 2 d0 ;  b1' = 3 t4 - 2 b1 | d1' = 3 t7 + 2 d1
  quad_sel_b<C>(m, xs, cs);  // t8 | t2
  quad_sel_b<C>(n, cs, xs);  // t4 | t7
  fp2_neg<C>(np, k.p);
  quad_sel_b<C>(t, np, k.p);  // b0 | -d0
  fp2_add<C>(t, t, m);
  fp2_dbl<C>(t, t);
  fp2_add<C>(k.p, t, m);  // 2 (m +- p) + m   (5)
  fp2_neg<C>(np, k.q);
  quad_sel_b<C>(t, k.q, np);  // -b1 | d1
  fp2_add<C>(t, t, n);
  fp2_dbl<C>(t, t);
  fp2_add<C>(k.q, t, n);
  fp2_reduce<C>(k.p);
  fp2_reduce<C>(k.q);
}
// the four compressed coefficients, each replicated on both pairs, for the (replicated) decompression of pairing.h
template <class C, class E>
MLHIP_HD void cyclo_replicate_q(CycloComp<C, E>& r, const CycloCompQ<C, E>& k) {
  quad_on_a<C>(r.b0, k.p);
  quad_on_b<C>(r.d0, k.p);
  quad_on_a<C>(r.b1, k.q);
  quad_on_b<C>(r.d1, k.q);
}

// z^|x| for z in the cyclotomic subgroup (conjugated when the seed is negative): the chain of pairing.h's fp12_expt with
// the squarings on the quad; the saved values are decompressed replicated (one shared inversion) and multiplied in
template <class C, class E>
MLHIP_HD_NOINLINE void fp12q_expt(Fp12Q<C, E>& r, const Fp12Q<C, E>& z) {
  int top = 63;
  while (!((C::X_ABS >> top) & 1)) top--;
  constexpr int NSET = mlhip_popcount64(C::X_ABS);
  if constexpr (NSET > 8) {
    // BN254's seed: plain square-and-multiply; on a quad the generic square is ONE Fp6 product, which is what a Granger-Scott
    // squaring spread over the pairs would cost as well
    Fp12Q<C, E> acc = z;
    for (int i = top - 1; i >= 0; i--) {
      fp12q_sqr<C>(acc, acc);
      if ((C::X_ABS >> i) & 1) fp12q_mul<C>(acc, acc, z);
    }
    if (C::X_NEG) fp12q_conj<C>(acc, acc);
    r = acc;
    return;
  }
  constexpr int NS = (NSET <= 8 ? NSET : 1) - (int)(C::X_ABS & 1);
  CycloCompQ<C, E> k, saved[NS > 0 ? NS : 1];
  cyclo_compress_q<C>(k, z);
  int ns = 0;
  for (int i = 1; i <= top; i++) {
    cyclo_sqr_compressed_q<C>(k);
    if ((C::X_ABS >> i) & 1) saved[ns++] = k;
  }
  E num[NS > 0 ? NS : 1], den[NS > 0 ? NS : 1], pre[NS > 0 ? NS : 1], inv, t;
  for (int j = 0; j < NS; j++) {
    CycloComp<C, E> kc;
    cyclo_replicate_q<C>(kc, saved[j]);
    cyclo_a1_fraction<C>(num[j], den[j], kc);
    if (j == 0)
      pre[0] = den[0];
    else
      fp2_mul<C>(pre[j], pre[j - 1], den[j]);
  }
  Fp12Q<C, E> acc, vq;
  bool have = false;
  if (C::X_ABS & 1) {
    acc = z;
    have = true;
  }
  if (NS > 0) {
    fp2_inv<C>(inv, pre[NS - 1]);
    for (int j = NS - 1; j >= 0; j--) {
      E dj_inv;
      if (j > 0) {
        fp2_mul<C>(dj_inv, inv, pre[j - 1]);
        fp2_mul<C>(inv, inv, den[j]);
      } else {
        dj_inv = inv;
      }
      fp2_mul<C>(t, num[j], dj_inv);  // a1
      CycloComp<C, E> kc;
      cyclo_replicate_q<C>(kc, saved[j]);
      Fp12<C, E> v;
      cyclo_decompress<C>(v, kc, t);
      fp12q_from_replicated<C>(vq, v);
      if (have) {
        fp12q_mul<C>(acc, acc, vq);
      } else {
        acc = vq;
        have = true;
      }
    }
  }
  if (C::X_NEG) fp12q_conj<C>(acc, acc);
  r = acc;
}

// r = f^(3 (p^12 - 1)/r_order): pairing.h's final_exp (BLS12 branch) on the quad form
template <class C, class E>
MLHIP_HD void final_exp_q(Fp12Q<C, E>& out, const Fp12Q<C, E>& f) {
  Fp12Q<C, E> r, t0, t1, t2;
  // easy part: f^((p^6-1)(p^2+1))
  fp12q_conj<C>(t0, f);
  fp12q_inv<C>(t1, f);
  fp12q_mul<C>(t0, t0, t1);
  fp12q_frob<C, 2>(t1, t0);
  fp12q_mul<C>(r, t1, t0);
  if constexpr (C::IS_BN) {
    // hard part of pairing.h's BN branch (Fuentes-Castaneda): exponent l0 + l1 p + l2 p^2 + l3 p^3 = 2x(6x^2+3x+1)(p^4-p^2+1)/r
    Fp12Q<C, E> fx, f2x, f6x, f6x2, f12x3, a, b;
    fp12q_expt<C>(fx, r);
    fp12q_sqr<C>(f2x, fx);
    fp12q_sqr<C>(t0, f2x);  // 4x
    fp12q_mul<C>(f6x, t0, f2x);
    fp12q_expt<C>(f6x2, f6x);
    fp12q_sqr<C>(t0, f6x2);  // 12x^2
    fp12q_expt<C>(f12x3, t0);
    fp12q_mul<C>(a, f12x3, f6x2);
    fp12q_mul<C>(a, a, f6x);
    fp12q_conj<C>(t0, f2x);
    fp12q_mul<C>(b, a, t0);
    fp12q_mul<C>(t0, a, f6x2);
    fp12q_mul<C>(t0, t0, r);
    fp12q_frob<C, 1>(t1, b);
    fp12q_mul<C>(t0, t0, t1);
    fp12q_frob<C, 2>(t1, a);
    fp12q_mul<C>(t0, t0, t1);
    fp12q_conj<C>(t1, r);
    fp12q_mul<C>(t1, b, t1);
    fp12q_frob<C, 3>(t2, t1);
    fp12q_mul<C>(out, t0, t2);
    return;
  }
  // hard part, exponent (x-1)^2 (x+p) (x^2+p^2-1) + 3
  fp12q_sqr<C>(t0, r);
  fp12q_expt<C>(t1, r);
  fp12q_conj<C>(t2, r);
  fp12q_mul<C>(t1, t1, t2);  // r^(x-1)
  fp12q_expt<C>(t2, t1);
  fp12q_conj<C>(t1, t1);
  fp12q_mul<C>(t1, t1, t2);  // r^((x-1)^2)
  fp12q_expt<C>(t2, t1);
  fp12q_frob<C, 1>(t1, t1);
  fp12q_mul<C>(t1, t1, t2);  // r^((x-1)^2 (x+p))
  fp12q_mul<C>(r, r, t0);    // r^3
  fp12q_expt<C>(t0, t1);
  fp12q_expt<C>(t2, t0);
  fp12q_frob<C, 2>(t0, t1);
  fp12q_conj<C>(t1, t1);
  fp12q_mul<C>(t1, t1, t2);
  fp12q_mul<C>(t1, t1, t0);
  fp12q_mul<C>(out, r, t1);
}

}  // namespace mlhip
