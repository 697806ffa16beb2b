// ed28.h -- bucket accumulation of a G1 MSM in extended twisted Edwards coordinates (a = -1) over the carry-free field
// form (fp28.h).  BLS12-377 only (C::HAS_EDWARDS): its G1 curve y^2 = x^3 + 1 has a point of order two and therefore the
// model  -x^2 + y^2 = 1 + d x^2 y^2  (oracle/pyref.py: edwards_params states the birational map; the reference itself --
// gnark's MultiExp, driver/gurvy/bls12-377.go:229-242 -- stays on the Weierstrass curve, so this is an internal form like
// XYZZ, never seen at the ABI).
//
// Why: the mixed addition of a bucket loop is 7 products and 6 additions, with no squaring, no carry propagation and no
// exceptional case, against 8 products + 2 squares + the P = +-Q test of XYZZ (ec28.h).  The addition law is COMPLETE on
// the prime-order subgroup only (d is a square): the caller must vouch that every input lies in G1 (or is the point at
// infinity) -- mlhip_msm_plan_assume_srs; without that promise the Weierstrass kernels run.
//
// Coordinates.  A bucket is (X : Y : Z : T), x = X/Z, y = Y/Z, T = X Y / Z -- the footprint of an XYZZ28, so the bucket
// state of a plan serves both.  A point of the input is the HALVED "Niels" triple ((y + x)/2, (y - x)/2, d x y): with
//     A = (Y1 - X1) ymx   B = (Y1 + X1) ypx   C = T1 td   E = B - A   H = B + A   F = Z1 - C   G = Z1 + C
//     X3 = E F   Y3 = G H   T3 = E H   Z3 = F G
// every quantity is one half of the textbook one (add-2008-hwcd-3 with D = 2 Z1), the result one quarter: the same point.
// The halves keep every product at weight 2 x 2 (fp28.h: w_a w_b <= 8) with all stored coordinates normalized.
// Negation is (x, y) -> (-x, y): swap ypx / ymx, negate td.
#pragma once
#include "ec28.h"

namespace mlhip {

template <class C>
struct alignas(8) EdNiels28 {  // 168 B
  Fp28<C> ypx, ymx, td;
};
template <class C>
struct EdExt28 {
  Fp28<C> x, y, z, t;  // normalized; the identity is (0 : 1 : 1 : 0)
};

template <class C>
MLHIP_HD void ed28_set_identity(EdExt28<C>& r) {
  fp28_zero<C>(r.x);
  fp28_from_const<C>(r.y, C::ONE28);
  fp28_from_const<C>(r.z, C::ONE28);
  fp28_zero<C>(r.t);
}

// acc += q (q negated first when `negate`); any acc, q of odd order (or the identity)
template <class C>
MLHIP_HD void ed28_madd(EdExt28<C>& acc, const EdNiels28<C>& q, bool negate) {
  Fp28<C> qp, qm, qt, nt, a, b, A, B, Cc, E, H, F, G;
  fp28_select<C>(qp, negate, q.ymx, q.ypx);
  fp28_select<C>(qm, negate, q.ypx, q.ymx);
  fp28_neg<C>(nt, q.td);
  fp28_select<C>(qt, negate, nt, q.td);
  fp28_sub<C>(a, acc.y, acc.x);  // weight 2
  fp28_add<C>(b, acc.y, acc.x);
  fp28_mul<C>(A, a, qm);
  fp28_mul<C>(B, b, qp);
  fp28_mul<C>(Cc, acc.t, qt);
  fp28_sub<C>(E, B, A);  // weight 2 each
  fp28_add<C>(H, B, A);
  fp28_sub<C>(F, acc.z, Cc);
  fp28_add<C>(G, acc.z, Cc);
  fp28_mul<C>(acc.x, E, F);
  fp28_mul<C>(acc.y, G, H);
  fp28_mul<C>(acc.t, E, H);
  fp28_mul<C>(acc.z, F, G);
}

// acc += q, both extended (add-2008-hwcd-3, a = -1): 10 products.  For the folds of the cold paths.
template <class C>
MLHIP_HD void ed28_add(EdExt28<C>& acc, const EdExt28<C>& q) {
  Fp28<C> a1, b1, a2, b2, A, B, tt, Cc, D, z2, k, E, H, F, G;
  fp28_sub<C>(a1, acc.y, acc.x);
  fp28_add<C>(b1, acc.y, acc.x);
  fp28_sub<C>(a2, q.y, q.x);
  fp28_add<C>(b2, q.y, q.x);
  fp28_mul<C>(A, a1, a2);  // 2 x 2
  fp28_mul<C>(B, b1, b2);
  fp28_mul<C>(tt, acc.t, q.t);
  fp28_from_const<C>(k, C::ED_2D28);
  fp28_mul<C>(Cc, tt, k);
  fp28_add<C>(z2, acc.z, acc.z);
  fp28_mul<C>(D, z2, q.z);  // 2 x 1
  fp28_sub<C>(E, B, A);
  fp28_add<C>(H, B, A);
  fp28_sub<C>(F, D, Cc);
  fp28_add<C>(G, D, Cc);
  fp28_mul<C>(acc.x, E, F);
  fp28_mul<C>(acc.y, G, H);
  fp28_mul<C>(acc.t, E, H);
  fp28_mul<C>(acc.z, F, G);
}

// (X : Y : Z : T) -> the Weierstrass point in XYZZ28 (what the bucket reduction reads): with u = (Z + Y)/(Z - Y),
//   x_W = u/s - 1 = ((Z + Y) - s (Z - Y)) X / den,   y_W = f u Z / (s (Z - Y) X) = f (Z + Y) Z / den,   den = s (Z - Y) X,
// as X_W = Nx den, Y_W = Ny den^2, ZZ = den^2, ZZZ = den^3.  10 products, no inversion.  The identity (X = 0 mod p, in
// whatever representation) gives `inf`.
template <class C>
MLHIP_HD void ed28_to_xyzz28(XYZZ28<C>& r, bool& inf, const EdExt28<C>& e) {
  inf = fp28_maybe_zero<C>(e.x) && fp28_is_zero_exact<C>(e.x);
  if (inf) return;
  Fp28<C> zmy, zpy, k, szy, den, t1, nx, fz, ny;
  fp28_sub<C>(zmy, e.z, e.y);  // weight 2
  fp28_add<C>(zpy, e.z, e.y);
  fp28_from_const<C>(k, C::ED_S28);
  fp28_mul<C>(szy, zmy, k);
  fp28_mul<C>(den, szy, e.x);
  fp28_sub<C>(t1, zpy, szy);  // weight 3
  fp28_mul<C>(nx, t1, e.x);
  fp28_mul<C>(r.x, nx, den);
  fp28_sqr<C>(r.zz, den);
  fp28_mul<C>(r.zzz, r.zz, den);
  fp28_from_const<C>(k, C::ED_F28);
  fp28_mul<C>(fz, e.z, k);
  fp28_mul<C>(ny, zpy, fz);  // 2 x 1
  fp28_mul<C>(r.y, ny, r.zz);
}

// ---- Weierstrass affine (boundary form) -> Edwards, K points sharing one inversion -----------------------------------------
// x' = f (x + 1)/y, y' = (u - 1)/(u + 1), u = s (x + 1): the denominators y (u + 1) of the K points are inverted together
// (Montgomery's trick); (0, 0) -- the point at infinity -- maps to the identity.  Inputs of odd order have y != 0 and
// u != -1 (those are the points of order two and four).  Out: the halved affine pair (x'/2, y'/2) in the boundary form.
template <class C, int K>
MLHIP_HD void ed_affine_halves_batch(Fp<C> (&xh)[K], Fp<C> (&yh)[K], const Affine<FpField<C>> (&in)[K]) {
  Fp<C> one, s, fh, half, xp1[K], u[K], up1[K], den[K], pre[K], inv, t;
  bool isinf[K];
  fp_one<C>(one);
  fp_from_const<C>(s, C::ED_S);
  fp_from_const<C>(fh, C::ED_FH);
  fp_from_const<C>(half, C::ED_HALF);
#pragma unroll
  for (int i = 0; i < K; i++) {
    isinf[i] = fp_is_zero<C>(in[i].x) && fp_is_zero<C>(in[i].y);
    fp_add<C>(xp1[i], in[i].x, one);
    fp_mul<C>(u[i], s, xp1[i]);
    fp_add<C>(up1[i], u[i], one);
    fp_mul<C>(den[i], in[i].y, up1[i]);
    fp_select<C>(den[i], isinf[i], one, den[i]);
    if (i == 0)
      pre[0] = den[0];
    else
      fp_mul<C>(pre[i], pre[i - 1], den[i]);
  }
  fp_inv<C>(inv, pre[K - 1]);
#pragma unroll
  for (int i = K - 1; i >= 0; i--) {
    Fp<C> di;  // 1 / den[i]
    if (i == 0) {
      di = inv;
    } else {
      fp_mul<C>(di, inv, pre[i - 1]);
      fp_mul<C>(inv, inv, den[i]);
    }
    Fp<C> iy, iu, um1;
    fp_mul<C>(iy, di, up1[i]);   // 1 / y
    fp_mul<C>(iu, di, in[i].y);  // 1 / (u + 1)
    fp_mul<C>(t, fh, xp1[i]);
    fp_mul<C>(xh[i], t, iy);
    fp_sub<C>(um1, u[i], one);
    fp_mul<C>(t, um1, iu);
    fp_mul<C>(yh[i], t, half);
    Fp<C> zero;
    fp_zero<C>(zero);
    fp_select<C>(xh[i], isinf[i], zero, xh[i]);
    fp_select<C>(yh[i], isinf[i], half, yh[i]);
  }
}
template <class C>
MLHIP_HD void ed_niels_from_halves(EdNiels28<C>& r, const Fp<C>& xh, const Fp<C>& yh) {
  Fp<C> k, t, td, p, m;
  fp_from_const<C>(k, C::ED_4D);
  fp_mul<C>(t, xh, yh);
  fp_mul<C>(td, t, k);  // d x y
  fp_add<C>(p, yh, xh);
  fp_sub<C>(m, yh, xh);
  fp28_from_fp<C>(r.ypx, p);
  fp28_from_fp<C>(r.ymx, m);
  fp28_from_fp<C>(r.td, td);
}
// one affine Weierstrass point -> extended coordinates (Z = 1); cold paths only (one inversion)
template <class C>
MLHIP_HD void ed28_from_affine(EdExt28<C>& r, const Affine<FpField<C>>& p) {
  Fp<C> xh[1], yh[1], x, y, t;
  Affine<FpField<C>> in[1] = {p};
  ed_affine_halves_batch<C, 1>(xh, yh, in);
  fp_add<C>(x, xh[0], xh[0]);
  fp_add<C>(y, yh[0], yh[0]);
  fp_mul<C>(t, x, y);
  fp28_from_fp<C>(r.x, x);
  fp28_from_fp<C>(r.y, y);
  fp28_from_fp<C>(r.t, t);
  fp28_from_const<C>(r.z, C::ONE28);
}

// ---- the unified addition over the four lanes of a quad (the bucket reduction; backends: ec_quad28.h) -------------------
// Lane q of a quad holds coordinate q of (X : Y : Z : T), as the XYZZ reduction holds X, Y, ZZ, ZZZ.  add-2008-hwcd-3 is then
// THREE rounds of one product per lane -- A | B | D | T1 T2, then C = 2 d T1 T2 on the fourth lane, then X3 | Y3 | Z3 | T3 --
// against four for XYZZ, with no carry propagation, no zero test and no branch: the identity (0 : 1 : 1 : 0) and equal
// operands go through the same instructions.
template <class C, class B>
MLHIP_HD void ed_quad28_add(typename B::V& a, const typename B::V& b) {
  typedef typename B::V V;
  V t, s, d, x, y, k, m1, m2, v, p1, sum, diff, ea, fa, ha, ga;
  B::template perm<0xE1>(t, a);  // Y1 | X1 | Z1 | T1   (quad_perm [1,0,2,3])
  B::add(s, a, t);               // Y1 + X1 | . | 2 Z1 | 2 T1
  B::sub(d, t, a);               // Y1 - X1 | . | 0 | 0
  B::sel(x, 0x1u, d, s);
  B::sel(x, 0x8u, a, x);  // Y1 - X1 | Y1 + X1 | 2 Z1 | T1                              weight 2
  B::template perm<0xE1>(t, b);
  B::add(s, b, t);
  B::sub(d, t, b);
  B::sel(y, 0x1u, d, b);
  B::sel(y, 0x2u, s, y);  // Y2 - X2 | Y2 + X2 | Z2 | T2
  B::mul(m1, x, y);       // A | B | D | T1 T2                                           <= 2 x 2
  B::konst(k, C::ED_2D28);
  B::mul(m2, m1, k);      // . | . | . | C
  B::sel(v, 0x8u, m2, m1);             // A | B | D | C
  B::template perm<0xB1>(p1, v);       // B | A | C | D   (quad_perm [1,0,3,2])
  B::add(sum, v, p1);                  // H | H | G | G                                  weight 2
  B::sub(diff, p1, v);                 // E | -E | -F | F
  B::template perm<0x00>(ea, diff);    // E on every lane
  B::template perm<0xFF>(fa, diff);    // F
  B::template perm<0x00>(ha, sum);     // H
  B::template perm<0xAA>(ga, sum);     // G
  B::sel(x, 0x2u, ga, ea);
  B::sel(x, 0x4u, fa, x);  // E | G | F | E
  B::sel(y, 0x1u, fa, ha);
  B::sel(y, 0x4u, ga, y);  // F | H | G | H
  B::mul(a, x, y);         // X3 | Y3 | Z3 | T3                                          2 x 2
}

}  // namespace mlhip
