// ec_jac.h -- NOT product code (nothing under mathlib_amd/csrc includes it; tools/jac_repro.hip does).
// Jacobian coordinates (x = X/Z^2, y = Y/Z^3; infinity <=> Z = 0) for the batched double-and-add of
// k_scalar_mul (msm_scalar_mul.h): a = 0 curves, formulas restated from the EFD -- dbl-2009-l (2M + 5S against XYZZ's
// 6M + 3S) and madd-2007-bl (7M + 4S against 8M + 2S) -- complete through explicit branches (infinity on either side,
// acc = +-q), so that points outside the prime-order subgroup and points of small order stay exact, as the XYZZ forms do.
// Reference semantics: G1.Mul, driver/gurvy/bls12381/bls12-381.go:238-247 (double-and-add shape of :920-932).
//
// History: this is the round-3 experiment whose first device form ended in HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION.
// The cause (profiles/r04_aperture_fault_isa.txt) was not in these formulas: the out-of-line table
// helper walked the caller's table BACKWARDS through a generic (flat) reference, the compiler strength-reduced the walk to a
// decremented 64-bit base plus a positive immediate offset, and a FLAT access picks its aperture from the base alone.
// jac_small_multiples below therefore reaches the caller's table through a PRIVATE-address-space pointer (scratch_
// instructions: 32-bit offsets, no aperture decision), and tools/check_codeobj.py rejects any code object in which a
// decremented register pair addresses a flat access.  Measured once fixed (profiles/r04_jac_repro.txt): bit-identical to
// the XYZZ kernel at 64 / 4 099 / 2^20 products and SLOWER, 76.1 against 64.5 ms per 2^20 -- the XYZZ kernel stays.
#pragma once
#include "../mathlib_amd/csrc/ec_jac.h"

namespace mlhip {

// (Jac, jac_set_inf, jac_is_inf, jac_dbl: the product's mathlib_amd/csrc/ec_jac.h, where the host tail's Horner pass uses them)
template <class F>
MLHIP_HD void jac_from_affine(Jac<F>& r, const Affine<F>& p) {
  if (affine_is_inf<F>(p)) {
    jac_set_inf<F>(r);
    return;
  }
  r.x = p.x;
  r.y = p.y;
  F::one(r.z);
}

// acc += (affine q), q negated first when `negate`   (madd-2007-bl)
template <class F>
MLHIP_HD void jac_madd(Jac<F>& acc, const Affine<F>& q_in, bool negate) {
  if (affine_is_inf<F>(q_in)) return;
  typename F::T qy, ny;
  F::neg(ny, q_in.y);
  F::select(qy, negate, ny, q_in.y);
  if (jac_is_inf<F>(acc)) {
    acc.x = q_in.x;
    acc.y = qy;
    F::one(acc.z);
    return;
  }
  typename F::T Z1Z1, U2, S2, H, HH, I, J, rr, V, t;
  F::sqr(Z1Z1, acc.z);
  F::mul(U2, q_in.x, Z1Z1);
  F::mul(S2, qy, acc.z);
  F::mul(S2, S2, Z1Z1);
  F::sub(H, U2, acc.x);
  F::sub(rr, S2, acc.y);
  if (F::is_zero(H)) {
    if (F::is_zero(rr)) {  // acc = q: double the affine point
      Jac<F> d;
      d.x = q_in.x;
      d.y = qy;
      F::one(d.z);
      jac_dbl<F>(acc, d);
    } else {
      jac_set_inf<F>(acc);
    }
    return;
  }
  F::dbl(rr, rr);  // r = 2 (S2 - Y1)
  F::sqr(HH, H);
  F::dbl(I, HH);
  F::dbl(I, I);  // I = 4 HH
  F::mul(J, H, I);
  F::mul(V, acc.x, I);
  F::add(t, acc.z, H);
  F::sqr(t, t);
  F::sub(t, t, Z1Z1);
  F::sub(acc.z, t, HH);  // Z3 = (Z1 + H)^2 - Z1Z1 - HH
  F::sqr(t, rr);
  F::sub(t, t, J);
  F::sub(t, t, V);
  F::sub(t, t, V);  // X3 = r^2 - J - 2 V
  F::sub(V, V, t);
  F::mul(V, rr, V);
  F::mul(J, acc.y, J);
  F::dbl(J, J);
  F::sub(acc.y, V, J);  // Y3 = r (V - X3) - 2 Y1 J
  acc.x = t;
}

// affine = X/Z^2, Y/Z^3 (infinity -> (0,0))
template <class F>
MLHIP_HD void jac_to_affine(Affine<F>& r, const Jac<F>& p) {
  if (jac_is_inf<F>(p)) {
    F::zero(r.x);
    F::zero(r.y);
    return;
  }
  typename F::T zi, zi2;
  F::inv(zi, p.z);
  F::sqr(zi2, zi);
  F::mul(r.x, p.x, zi2);
  F::mul(zi2, zi2, zi);
  F::mul(r.y, p.y, zi2);
}

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
// The caller's table as the callee sees it: a pointer into the PRIVATE address space.  Loads and stores through it are
// scratch_ instructions with 32-bit offsets; a generic `Affine<F>(&)[8]` would make them FLAT ones, and a flat access whose
// 64-bit base the loop optimiser has moved below the start of the private aperture is routed to the wrong aperture (the
// round-3 fault; see the header comment).
template <class T>
using PrivatePtr = __attribute__((address_space(5))) T*;
template <class T>
__device__ __forceinline__ PrivatePtr<T> to_private(T* p) {
  return (PrivatePtr<T>)p;
}
// tab[k] = a, word by word (a class type has no assignment operator across address spaces)
template <class F>
__device__ __forceinline__ void jac_tab_store(PrivatePtr<Affine<F>> tab, int k, const Affine<F>& a) {
  PrivatePtr<uint32_t> w = (PrivatePtr<uint32_t>)(tab + k);
  const uint32_t* s = reinterpret_cast<const uint32_t*>(&a);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(Affine<F>) / 4); i++) w[i] = s[i];
}
#endif
template <class F>
MLHIP_HD void jac_tab_store(Affine<F>* tab, int k, const Affine<F>& a) {
  tab[k] = a;
}

// tab[k] = (k + 1) P for k = 0 .. 7 as AFFINE points: seven Jacobian additions, then ONE inversion for all of them
// (Montgomery's trick: prefix products of the Z_k forwards, the inverses peeled off backwards).  Entries that are the point
// at infinity (P itself, or k P for a P of order <= 8) are left out of the products and stored as (0, 0).
// TabRef is `Affine<F>*` on the host and a PrivatePtr on the device (jac_small_multiples_ool).
template <class F, class TabRef>
MLHIP_HD void jac_small_multiples_body(TabRef tab, const Affine<F>& P) {
  Jac<F> m[8];
  typename F::T pre[8];  // pre[k] = product of the finite Z_j, j <= k
  jac_from_affine<F>(m[0], P);
  for (int k = 1; k < 8; k++) {
    m[k] = m[k - 1];
    jac_madd<F>(m[k], P, false);
  }
  typename F::T run;
  F::one(run);
  for (int k = 0; k < 8; k++) {
    if (!jac_is_inf<F>(m[k])) F::mul(run, run, m[k].z);
    pre[k] = run;
  }
  typename F::T inv;
  F::inv(inv, run);  // run != 0: a product of non-zero Z_k (or one)
  for (int k = 7; k >= 0; k--) {
    Affine<F> a;
    if (jac_is_inf<F>(m[k])) {
      F::zero(a.x);
      F::zero(a.y);
    } else {
      typename F::T zi, zi2;
      if (k > 0)
        F::mul(zi, inv, pre[k - 1]);  // 1/Z_k
      else
        zi = inv;
      F::mul(inv, inv, m[k].z);  // drop Z_k from the running inverse
      F::sqr(zi2, zi);
      F::mul(a.x, m[k].x, zi2);
      F::mul(zi2, zi2, zi);
      F::mul(a.y, m[k].y, zi2);
    }
    jac_tab_store<F>(tab, k, a);
  }
}

}  // namespace mlhip
