// fp2_lanes.h -- Fp2 elements spread over lane pairs (gfx950, device only).
//
// Fp2L<C> holds ONE Fp component per lane: lane 2i owns c0 and lane 2i+1 owns c1 of the same Fp2 element,
// so one pairing is computed by two adjacent lanes.  Why: at the configured batch of 65 536 pairings a
// one-pairing-per-lane grid is exactly one wave per SIMD (a lone wave issues at most ~50 % of the
// v_mad_u64_u32 peak and every scratch access is exposed); two lanes per pairing double the resident waves
// and halve the per-lane state (an Fp12 is 72 words per lane instead of 144).
//
//   * add / sub / neg / halve / scalar multiples: each lane works on its own component;
//   * multiplication: the partner's components arrive through a quad-permute DPP move, and each lane
//     forms its result component as ONE fused dual product  (a*y1 + a'*y2) R^-1  with a single Montgomery
//     reduction (fp_mul2_device):  c0 = a0 b0 + (BETA a1) b1 ,  c1 = a1 b0 + a0 b1;
//   * everything from Fp6 up (tower.h, pairing.h) is shared with the one-lane-per-element code through the
//     fp2_* overload set.
// Control flow must be uniform across a lane pair (both lanes execute every exchange); the callers only
// branch on pair-uniform conditions.
#pragma once
#include "tower.h"

namespace mlhip {

template <class C>
struct Fp2L {
  Fp<C> v;
};

// The towers above are __host__ __device__ templates, so these overloads must exist in the host pass too;
// there they are never called (the lane-pair kernels are device code) and the primitives are stubs.
MLHIP_HD bool lane_is_hi() {
#if defined(__HIP_DEVICE_COMPILE__)
  return (threadIdx.x & 1u) != 0;
#else
  return false;
#endif
}

// value held by the other lane of the pair (quad_perm [1,0,3,2])
MLHIP_HD uint32_t pair_xchg_u32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true);  // bound_ctrl: with it hipcc emits the move alone; without, a v_mov of the "old" value 0 in front of every one
#else
  return x;
#endif
}
// single / fused dual Montgomery products (device asm; portable loop in the never-executed host pass)
template <class C>
MLHIP_HD void lp_mul(Fp<C>& r, const Fp<C>& a, const Fp<C>& b) {
#if defined(__HIP_DEVICE_COMPILE__)
  fp_mul_device<C>(r, a, b);
#else
  fp_mul_inline<C>(r, a, b);
#endif
}
template <class C>
MLHIP_HD void lp_mul2(Fp<C>& r, const Fp<C>& a, const Fp<C>& b, const Fp<C>& c, const Fp<C>& d) {
#if defined(__HIP_DEVICE_COMPILE__)
  fp_mul2_device<C>(r, a, b, c, d);
#else
  Fp<C> t0, t1;
  fp_mul_inline<C>(t0, a, b);
  fp_mul_inline<C>(t1, c, d);
  fp_add<C>(r, t0, t1);
#endif
}
template <class C>
MLHIP_HD void fp_pair_xchg(Fp<C>& r, const Fp<C>& a) {
#pragma unroll
  for (int i = 0; i < C::N; i++) r.l[i] = pair_xchg_u32(a.l[i]);
}

template <class C>
MLHIP_HD void fp2_zero(Fp2L<C>& r) { fp_zero<C>(r.v); }
template <class C>
MLHIP_HD void fp2_one(Fp2L<C>& r) {
  Fp<C> one, zero;
  fp_one<C>(one);
  fp_zero<C>(zero);
  fp_select<C>(r.v, lane_is_hi(), zero, one);
}
template <class C>
MLHIP_HD bool fp2_is_zero(const Fp2L<C>& a) {
  uint32_t z = fp_is_zero<C>(a.v) ? 1u : 0u;
  return (z & pair_xchg_u32(z)) != 0;
}
template <class C>
MLHIP_HD bool fp2_eq(const Fp2L<C>& a, const Fp2L<C>& b) {
  uint32_t e = fp_eq<C>(a.v, b.v) ? 1u : 0u;
  return (e & pair_xchg_u32(e)) != 0;
}
template <class C>
MLHIP_HD void fp2_add(Fp2L<C>& r, const Fp2L<C>& a, const Fp2L<C>& b) { fp_add<C>(r.v, a.v, b.v); }
template <class C>
MLHIP_HD void fp2_sub(Fp2L<C>& r, const Fp2L<C>& a, const Fp2L<C>& b) { fp_sub<C>(r.v, a.v, b.v); }
template <class C>
MLHIP_HD void fp2_dbl(Fp2L<C>& r, const Fp2L<C>& a) { fp_dbl<C>(r.v, a.v); }
template <class C>
MLHIP_HD void fp2_neg(Fp2L<C>& r, const Fp2L<C>& a) { fp_neg<C>(r.v, a.v); }
template <class C>
MLHIP_HD void fp2_conj(Fp2L<C>& r, const Fp2L<C>& a) {
  Fp<C> n;
  fp_neg<C>(n, a.v);
  fp_select<C>(r.v, lane_is_hi(), n, a.v);
}
template <class C>
MLHIP_HD void fp2_select(Fp2L<C>& r, bool c, const Fp2L<C>& a, const Fp2L<C>& b) {
  fp_select<C>(r.v, c, a.v, b.v);
}
template <class C>
MLHIP_HD void fp2_norm(Fp2L<C>&) {}
template <class C>
MLHIP_HD void fp2_reduce(Fp2L<C>&) {}
template <class C>
MLHIP_HD int fp2_weight(const Fp2L<C>&) { return 1; }
template <class C>
MLHIP_HD void fp2_halve(Fp2L<C>& r, const Fp2L<C>& a);  // defined after pairing.h's fp_halve

// (a0 + a1 u)(b0 + b1 u) = (a0 b0 + BETA a1 b1) + (a1 b0 + a0 b1) u
//   lane c0: own a=a0, b=b0, partner a'=a1, b'=b1  ->  a*b  + a'*(BETA b')
//   lane c1: own a=a1, b=b1, partner a'=a0, b'=b0  ->  a*b' + a'*b
template <class C>
MLHIP_HD void fp2_mul(Fp2L<C>& r, const Fp2L<C>& a, const Fp2L<C>& b) {
  const bool hi = lane_is_hi();
  Fp<C> ax, bx, nb, y1, y2;
  fp_pair_xchg<C>(ax, a.v);
  fp_pair_xchg<C>(bx, b.v);
  fp_mul_beta<C>(nb, bx);
  fp_select<C>(y1, hi, bx, b.v);
  fp_select<C>(y2, hi, b.v, nb);
  lp_mul2<C>(r.v, a.v, y1, ax, y2);
}

template <class C>
MLHIP_HD void fp2_sqr(Fp2L<C>& r, const Fp2L<C>& a) {
  const bool hi = lane_is_hi();
  Fp<C> ax;
  fp_pair_xchg<C>(ax, a.v);
  if (C::BETA == -1) {
    // c0 = (a0 + a1)(a0 - a1) ; c1 = (2 a1) a0   -- one single product per lane
    Fp<C> s, d, dd, x, y;
    fp_add<C>(s, a.v, ax);
    fp_sub<C>(d, a.v, ax);
    fp_dbl<C>(dd, a.v);
    fp_select<C>(x, hi, dd, s);
    fp_select<C>(y, hi, ax, d);
    lp_mul<C>(r.v, x, y);
  } else {
    // c0 = a0 a0 + a1 (BETA a1) ; c1 = a1 a0 + a0 a1
    Fp<C> nb, y1, y2;
    fp_mul_beta<C>(nb, ax);
    fp_select<C>(y1, hi, ax, a.v);
    fp_select<C>(y2, hi, a.v, nb);
    lp_mul2<C>(r.v, a.v, y1, ax, y2);
  }
}

template <class C>
MLHIP_HD void fp2_mul_fp(Fp2L<C>& r, const Fp2L<C>& a, const Fp<C>& k) {
  lp_mul<C>(r.v, a.v, k);
}
template <class C>
MLHIP_HD void fp2_mul_small(Fp2L<C>& r, const Fp2L<C>& a, int k) { fp_mul_small<C>(r.v, a.v, k); }

template <class C>
MLHIP_HD void fp2_mul_by_real_const(Fp2L<C>& r, const Fp2L<C>& a, const uint32_t (&k)[2][C::N]) {
  Fp<C> kr;
  fp_from_const<C>(kr, k[0]);
  lp_mul<C>(r.v, a.v, kr);
}

// XI * a, XI = XI0 + XI1 u:  c0 = XI0 a0 + BETA XI1 a1 ; c1 = XI0 a1 + XI1 a0
template <class C>
MLHIP_HD void fp2_mul_xi(Fp2L<C>& r, const Fp2L<C>& a) {
  const bool hi = lane_is_hi();
  Fp<C> ax, t0, t1, tb;
  fp_pair_xchg<C>(ax, a.v);
  if (C::XI0 == 0) {
    // XI = u: c0 = BETA a1, c1 = a0
    fp_mul_beta<C>(tb, ax);
    fp_select<C>(r.v, hi, ax, tb);
  } else {
    fp_mul_small<C>(t0, a.v, C::XI0);
    fp_mul_small<C>(t1, ax, C::XI1);
    fp_mul_beta<C>(tb, t1);
    Fp<C> other;
    fp_select<C>(other, hi, t1, tb);
    fp_add<C>(r.v, t0, other);
  }
}

template <class C>
MLHIP_HD void fp2_inv(Fp2L<C>& r, const Fp2L<C>& a) {
  // 1/(a0 + a1 u) = (a0 - a1 u) / (a0^2 - BETA a1^2); both lanes compute the same norm and inverse
  const bool hi = lane_is_hi();
  Fp<C> sq, sqx, mine, theirs, bsq, n, ni, t, nt;
  lp_mul<C>(sq, a.v, a.v);
  fp_pair_xchg<C>(sqx, sq);
  fp_select<C>(mine, hi, sqx, sq);    // a0^2 on both lanes
  fp_select<C>(theirs, hi, sq, sqx);  // a1^2 on both lanes
  fp_mul_beta<C>(bsq, theirs);
  fp_sub<C>(n, mine, bsq);
  fp_inv<C>(ni, n);
  lp_mul<C>(t, a.v, ni);
  fp_neg<C>(nt, t);
  fp_select<C>(r.v, hi, nt, t);
}

template <class C>
MLHIP_HD void fp2_from_const(Fp2L<C>& r, const uint32_t (&k)[2][C::N]) {
  const bool hi = lane_is_hi();
#pragma unroll
  for (int i = 0; i < C::N; i++) r.v.l[i] = hi ? k[1][i] : k[0][i];
}

}  // namespace mlhip
