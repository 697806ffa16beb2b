// ec_quad.h -- XYZZ + XYZZ addition spread over the four lanes of a quad (G1, boundary-form Fp).
//
// The bucket reduction (k_chunks, k_masked_sums) is a chain of dependent group additions: its run time is the
// DEPTH of that chain, not the amount of work.  One lane doing add-2008-s executes 14 Fp multiplications one
// after another; here lane q of a quad owns coordinate q of every point (0: X, 1: Y, 2: ZZ, 3: ZZZ) and the same
// formula becomes FOUR rounds of one multiplication per lane (16 lane-multiplications instead of 14, 3.5x
// shallower), with the operands moved by DPP quad permutes:
//
//   round 1   a * perm[2,3,0,1](b)                      -> U1 = X1 ZZ2 | S1 = Y1 ZZZ2 | U2 = X2 ZZ1 | S2 = Y2 ZZZ1
//             d = U2 - U1 | S2 - S1 | U2 - U1 | S2 - S1  -> P | R | P | R
//   round 2   d d | d d | a b | a b                      -> PP | RR | ZZ1 ZZ2 | ZZZ1 ZZZ2
//   round 3   U1 PP | P PP | (ZZ1 ZZ2) PP | P PP         -> Q | PPP | ZZ3 | PPP
//             X3 = RR - PPP - 2Q  (every lane)
//   round 4   R (Q - X3) | S1 PPP | - | (ZZZ1 ZZZ2) PPP  -> V | T | - | ZZZ3
//             Y3 = V - T
//
// The algorithm is written once over a backend: QuadDevice (one Fp per lane, DPP) for the kernels and QuadHost
// (four values in an array) so that tests/test_host_math.py can check it against the oracle without a GPU.
// Control flow is quad-uniform: every branch is taken on masks that all four lanes hold.
#pragma once
#include "ec.h"

namespace mlhip {

// ---- host emulation backend --------------------------------------------------------------------------------
template <class C>
struct QuadHost {
  struct V {
    Fp<C> v[4];
  };
  static void mul(V& r, const V& x, const V& y) {
    for (int i = 0; i < 4; i++) fp_mul_inline<C>(r.v[i], x.v[i], y.v[i]);
  }
  static void sub(V& r, const V& x, const V& y) {
    for (int i = 0; i < 4; i++) fp_sub<C>(r.v[i], x.v[i], y.v[i]);
  }
  template <int CTRL>
  static void perm(V& r, const V& x) {
    V t = x;
    for (int i = 0; i < 4; i++) r.v[i] = t.v[(CTRL >> (2 * i)) & 3];
  }
  static void sel(V& r, unsigned lanes, const V& x, const V& y) {
    for (int i = 0; i < 4; i++) r.v[i] = ((lanes >> i) & 1u) ? x.v[i] : y.v[i];
  }
  static unsigned zero_mask(const V& x) {
    unsigned m = 0;
    for (int i = 0; i < 4; i++) m |= (fp_is_zero<C>(x.v[i]) ? 1u : 0u) << i;
    return m;
  }
  static void gather(XYZZ<FpField<C>>& p, const V& x) {
    p.x = x.v[0];
    p.y = x.v[1];
    p.zz = x.v[2];
    p.zzz = x.v[3];
  }
  static void scatter(V& r, const XYZZ<FpField<C>>& p) {
    r.v[0] = p.x;
    r.v[1] = p.y;
    r.v[2] = p.zz;
    r.v[3] = p.zzz;
  }
};

#if defined(__HIPCC__)
// ---- device backend: lane (threadIdx.x & 3) of every aligned group of four lanes ------------------------------
template <class C>
struct QuadDevice {
  typedef Fp<C> V;
  // (the bodies are empty in hipcc's host pass, which still parses the kernels that name this type)
  static __device__ __forceinline__ unsigned lane() { return threadIdx.x & 3u; }
  static __device__ __forceinline__ void mul(V& r, const V& x, const V& y) {
#if defined(__HIP_DEVICE_COMPILE__)
    fp_mul_device<C>(r, x, y);
#endif
  }
  static __device__ __forceinline__ void sub(V& r, const V& x, const V& y) { fp_sub<C>(r, x, y); }
  template <int CTRL>
  static __device__ __forceinline__ void perm(V& r, const V& x) {
#pragma unroll
    for (int i = 0; i < C::N; i++) r.l[i] = xlane<CTRL>(x.l[i]);
  }
  template <int CTRL>
  static __device__ __forceinline__ uint32_t xlane(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);  // bound_ctrl: every quad_perm source lane is valid, and hipcc then needs no v_mov of the "old" value per move
#else
    return v;
#endif
  }
  static __device__ __forceinline__ void sel(V& r, unsigned lanes, const V& x, const V& y) {
    fp_select<C>(r, ((lanes >> lane()) & 1u) != 0, x, y);
  }
  static __device__ __forceinline__ unsigned zero_mask(const V& x) {
    unsigned m = (fp_is_zero<C>(x) ? 1u : 0u) << lane();
    m |= xlane<0xB1>(m);  // [1,0,3,2]
    m |= xlane<0x4E>(m);  // [2,3,0,1]
    return m;
  }
  static __device__ __forceinline__ void gather(XYZZ<FpField<C>>& p, const V& x) {
    perm<0x00>(p.x, x);
    perm<0x55>(p.y, x);
    perm<0xAA>(p.zz, x);
    perm<0xFF>(p.zzz, x);
  }
  static __device__ __forceinline__ void scatter(V& r, const XYZZ<FpField<C>>& p) {
    const unsigned q = lane();
    V t;
    fp_select<C>(t, q == 0, p.x, p.y);
    fp_select<C>(r, q >= 2, p.zz, t);
    fp_select<C>(r, q == 3, p.zzz, r);
  }
};
#endif

// the doubling that replaces the addition when both operands are the same point: rare, done by every lane on the
// gathered point with the one-lane formulas
template <class C>
MLHIP_HD_NOINLINE void quad_dbl_slow(XYZZ<FpField<C>>& r, const XYZZ<FpField<C>>& p) {
  xyzz_dbl<FpField<C>>(r, p);
}

// a += b; both hold coordinate `lane` of an XYZZ point (infinity: ZZ = 0)
template <class C, class B>
MLHIP_HD void quad_xyzz_add(typename B::V& a, const typename B::V& b) {
  typedef typename B::V V;
  if (B::zero_mask(b) & 4u) return;  // b = infinity
  if (B::zero_mask(a) & 4u) {        // a = infinity
    a = b;
    return;
  }
  V t, m1, o, x, y, d, m2, pp, pP, m3, rr, ppp, qq, X3, rR, e, m4, vv;
  B::template perm<0x4E>(t, b);  // ZZ2 | ZZZ2 | X2 | Y2
  B::mul(m1, a, t);              // U1 | S1 | U2 | S2
  B::template perm<0x4E>(o, m1);
  B::sel(x, 0x3u, o, m1);
  B::sel(y, 0x3u, m1, o);
  B::sub(d, x, y);  // P | R | P | R
  const unsigned zd = B::zero_mask(d);
  if (zd & 1u) {  // same x: the same point (double it) or opposite points (infinity)
    XYZZ<FpField<C>> p, r;
    if (zd & 2u) {
      B::gather(p, b);
      quad_dbl_slow<C>(r, p);
    } else {
      xyzz_set_inf<FpField<C>>(r);
    }
    B::scatter(a, r);
    return;
  }
  B::sel(x, 0x3u, d, a);
  B::sel(y, 0x3u, d, b);
  B::mul(m2, x, y);  // PP | RR | ZZ1 ZZ2 | ZZZ1 ZZZ2
  B::template perm<0x00>(pp, m2);
  B::template perm<0x00>(pP, d);
  B::sel(x, 0x1u, m1, pP);
  B::sel(x, 0x4u, m2, x);
  B::mul(m3, x, pp);  // Q | PPP | ZZ3 | PPP
  B::template perm<0x55>(rr, m2);
  B::template perm<0x55>(ppp, m3);
  B::template perm<0x00>(qq, m3);
  B::sub(X3, rr, ppp);
  B::sub(X3, X3, qq);
  B::sub(X3, X3, qq);
  B::template perm<0x55>(rR, d);
  B::sub(e, qq, X3);
  B::sel(x, 0x1u, rR, m1);
  B::sel(x, 0x8u, m2, x);
  B::sel(y, 0x1u, e, m3);
  B::mul(m4, x, y);  // V | T | - | ZZZ3
  B::template perm<0x00>(vv, m4);
  B::sub(y, vv, m4);  // lane 1: Y3
  B::sel(x, 0x1u, X3, y);
  B::sel(x, 0x4u, m3, x);
  B::sel(a, 0x8u, m4, x);
}

}  // namespace mlhip
