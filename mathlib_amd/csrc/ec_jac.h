// ec_jac.h -- Jacobian coordinates (x = X/Z^2, y = Y/Z^3; infinity <=> Z = 0) on the a = 0 curves, for the one place where a
// long chain of DOUBLINGS decides the time: the Horner pass of an MSM's host tail (msm_plan.h: host_tail -- one doubling
// per scalar bit, 256 of them in a row, 16 additions).  dbl-2009-l is 2M + 5S against the 6M + 3S of the XYZZ doubling the
// bucket code uses (ec.h); the general addition add-2007-bl (11M + 5S) is dearer than XYZZ's 12M + 2S but runs 16 times.
// Formulas restated from the EFD; complete through explicit branches like the XYZZ forms (infinity on either side, P = +-Q).
// Reference semantics of what this computes: the window combination inside gnark's MultiExp behind
// driver/gurvy/bls12381/bls12-381.go:777 (the result, a unique group element, is converted back to XYZZ / affine).
#pragma once
#include "ec.h"

namespace mlhip {

template <class F>
struct Jac {
  typename F::T x, y, z;
};

template <class F>
MLHIP_HD void jac_set_inf(Jac<F>& r) {
  F::one(r.x);
  F::one(r.y);
  F::zero(r.z);
}
template <class F>
MLHIP_HD bool jac_is_inf(const Jac<F>& p) {
  return F::is_zero(p.z);
}

// (X, Y, ZZ, ZZZ) with ZZ^3 = ZZZ^2 is the Jacobian point (X ZZ, Y ZZZ, ZZ): Z = ZZ gives Z^2 = ZZ^2 and Z^3 = ZZZ^2
template <class F>
MLHIP_HD void jac_from_xyzz(Jac<F>& r, const XYZZ<F>& p) {
  if (xyzz_is_inf<F>(p)) {
    jac_set_inf<F>(r);
    return;
  }
  F::mul(r.x, p.x, p.zz);
  F::mul(r.y, p.y, p.zzz);
  r.z = p.zz;
}
// ... and back: (X, Y, Z^2, Z^3)
template <class F>
MLHIP_HD void jac_to_xyzz(XYZZ<F>& r, const Jac<F>& p) {
  if (jac_is_inf<F>(p)) {
    xyzz_set_inf<F>(r);
    return;
  }
  r.x = p.x;
  r.y = p.y;
  F::sqr(r.zz, p.z);
  F::mul(r.zzz, r.zz, p.z);
}

// r = 2 p   (dbl-2009-l; Y = 0, a point of order two, gives Z3 = 0 = infinity by itself); r may alias p
template <class F>
MLHIP_HD void jac_dbl(Jac<F>& r, const Jac<F>& p) {
  typename F::T A, B, Cc, D, E, Fq, t;
  F::sqr(A, p.x);
  F::sqr(B, p.y);
  F::sqr(Cc, B);
  F::add(t, p.x, B);
  F::sqr(t, t);
  F::sub(t, t, A);
  F::sub(t, t, Cc);
  F::dbl(D, t);  // D = 2 ((X + B)^2 - A - C)
  F::dbl(E, A);
  F::add(E, E, A);  // E = 3 A
  F::sqr(Fq, E);
  F::mul(t, p.y, p.z);  // before r.y is written
  F::sub(r.x, Fq, D);
  F::sub(r.x, r.x, D);  // X3 = F - 2 D
  F::dbl(r.z, t);       // Z3 = 2 Y Z
  F::sub(t, D, r.x);
  F::mul(t, E, t);
  F::dbl(Cc, Cc);
  F::dbl(Cc, Cc);
  F::dbl(Cc, Cc);
  F::sub(r.y, t, Cc);  // Y3 = E (D - X3) - 8 C
}

// acc += q   (add-2007-bl)
template <class F>
MLHIP_HD void jac_add(Jac<F>& acc, const Jac<F>& q) {
  if (jac_is_inf<F>(q)) return;
  if (jac_is_inf<F>(acc)) {
    acc = q;
    return;
  }
  typename F::T Z1Z1, Z2Z2, U1, U2, S1, S2, H, I, J, rr, V, t;
  F::sqr(Z1Z1, acc.z);
  F::sqr(Z2Z2, q.z);
  F::mul(U1, acc.x, Z2Z2);
  F::mul(U2, q.x, Z1Z1);
  F::mul(S1, acc.y, q.z);
  F::mul(S1, S1, Z2Z2);
  F::mul(S2, q.y, acc.z);
  F::mul(S2, S2, Z1Z1);
  F::sub(H, U2, U1);
  F::sub(rr, S2, S1);
  if (F::is_zero(H)) {
    if (F::is_zero(rr)) {
      Jac<F> d;
      jac_dbl<F>(d, q);
      acc = d;
    } else {
      jac_set_inf<F>(acc);
    }
    return;
  }
  F::dbl(rr, rr);  // r = 2 (S2 - S1)
  F::dbl(I, H);
  F::sqr(I, I);  // I = (2 H)^2
  F::mul(J, H, I);
  F::mul(V, U1, I);
  F::add(t, acc.z, q.z);
  F::sqr(t, t);
  F::sub(t, t, Z1Z1);
  F::sub(t, t, Z2Z2);
  F::mul(acc.z, t, H);  // Z3 = ((Z1 + Z2)^2 - Z1Z1 - Z2Z2) H
  F::sqr(t, rr);
  F::sub(t, t, J);
  F::sub(t, t, V);
  F::sub(t, t, V);  // X3 = r^2 - J - 2 V
  F::sub(V, V, t);
  F::mul(V, rr, V);
  F::mul(J, S1, J);
  F::dbl(J, J);
  F::sub(acc.y, V, J);  // Y3 = r (V - X3) - 2 S1 J
  acc.x = t;
}

// total = sum_w 2^off[w] V[w]: Horner from the top window down, `down[w]` = off[w] - off[w - 1] doublings after adding V[w]
// (down[0] = 0).  The XYZZ form of the same pass is kept as horner_xyzz for the tests.
template <class F>
MLHIP_HD void horner_jac(XYZZ<F>& total, const XYZZ<F>* V, int W, const int* down) {
  Jac<F> acc, v;
  jac_set_inf<F>(acc);
  for (int w = W - 1; w >= 0; w--) {
    jac_from_xyzz<F>(v, V[w]);
    jac_add<F>(acc, v);
    for (int k = 0; k < down[w]; k++) jac_dbl<F>(acc, acc);
  }
  jac_to_xyzz<F>(total, acc);
}
template <class F>
MLHIP_HD void horner_xyzz(XYZZ<F>& total, const XYZZ<F>* V, int W, const int* down) {
  xyzz_set_inf<F>(total);
  for (int w = W - 1; w >= 0; w--) {
    xyzz_add<F>(total, V[w]);
    for (int k = 0; k < down[w]; k++) {
      XYZZ<F> d;
      xyzz_dbl<F>(d, total);
      total = d;
    }
  }
}

}  // namespace mlhip
