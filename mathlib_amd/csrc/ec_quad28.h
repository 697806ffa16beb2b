// ec_quad28.h -- XYZZ + XYZZ over the four lanes of a quad in the carry-free field form (fp28.h): the schedule of
// ec_quad.h (four rounds of one product per lane, operands moved by DPP quad permutes) on Fp28 values, for the G1
// bucket reduction behind the carry-free accumulation kernels.
//
// What changes against the boundary form: a product is 196 + 210 v_mad_i64_i32 instead of 288 v_mad_u64_u32 + 288
// v_addc, a subtraction is 14 independent v_sub instead of a borrow chain with a conditional correction, and the
// buckets arrive exactly as the accumulation kernel keeps them (XYZZ28, no conversion on either side).  What it costs:
// weights.  Every stored coordinate is normalized (weight 1); the differences P | R have weight 2 and enter products
// as 2 x 2 = 4 <= 8; X3 = RR - PPP - 2Q (weight 4) and Y3 = V - T (weight 2) are carry-propagated once each.
//
// Infinity is "ZZ has all limbs zero" -- the encoding of the accumulation kernels' bucket state (msm_accumulate.h).  A
// regular addition never produces it (ZZ3 = ZZ1 ZZ2 PP with P != 0 mod p); the exceptional cases (same x, found by the
// exact test fp28_is_zero_exact behind the one-multiply filter fp28_maybe_zero) write it as canonical zeros: opposite
// points directly, the same point through xyzz28_dbl, which tests 2Y = 0 exactly (a point of order two).
#pragma once
#include "ec28.h"
#include "ec_quad.h"

namespace mlhip {

// ---- host emulation backend ----------------------------------------------------------------------------------
template <class C>
struct QuadHost28 {
  struct V {
    Fp28<C> v[4];
  };
  static void mul(V& r, const V& x, const V& y) {
    for (int i = 0; i < 4; i++) fp28_mul<C>(r.v[i], x.v[i], y.v[i]);
  }
  static void sub(V& r, const V& x, const V& y) {
    for (int i = 0; i < 4; i++) fp28_sub<C>(r.v[i], x.v[i], y.v[i]);
  }
  static void add(V& r, const V& x, const V& y) {
    for (int i = 0; i < 4; i++) fp28_add<C>(r.v[i], x.v[i], y.v[i]);
  }
  static void konst(V& r, const int32_t (&k)[C::N28]) {
    for (int i = 0; i < 4; i++) fp28_from_const<C>(r.v[i], k);
  }
  static void norm(V& r, const V& x) {
    for (int i = 0; i < 4; i++) fp28_normalize<C>(r.v[i], x.v[i]);
  }
  template <int CTRL>
  static void perm(V& r, const V& x) {
    V t = x;
    for (int i = 0; i < 4; i++) r.v[i] = t.v[(CTRL >> (2 * i)) & 3];
  }
  static void sel(V& r, unsigned lanes, const V& x, const V& y) {
    for (int i = 0; i < 4; i++) r.v[i] = ((lanes >> i) & 1u) ? x.v[i] : y.v[i];
  }
  static unsigned allzero_mask(const V& x) {
    unsigned m = 0;
    for (int i = 0; i < 4; i++) m |= (fp28_all_zero<C>(x.v[i]) ? 1u : 0u) << i;
    return m;
  }
  static unsigned zero_mask(const V& x) {
    unsigned m = 0;
    for (int i = 0; i < 4; i++) m |= ((fp28_maybe_zero<C>(x.v[i]) && fp28_is_zero_exact<C>(x.v[i])) ? 1u : 0u) << i;
    return m;
  }
  static void gather(XYZZ28<C>& p, const V& x) {
    p.x = x.v[0];
    p.y = x.v[1];
    p.zz = x.v[2];
    p.zzz = x.v[3];
  }
  static void scatter(V& r, const XYZZ28<C>& p) {
    r.v[0] = p.x;
    r.v[1] = p.y;
    r.v[2] = p.zz;
    r.v[3] = p.zzz;
  }
};

#if defined(__HIPCC__)
// ---- device backend: lane (threadIdx.x & 3) of every aligned group of four lanes ---------------------------------
template <class C>
struct QuadDevice28 {
  typedef Fp28<C> V;
  static __device__ __forceinline__ unsigned lane() { return threadIdx.x & 3u; }
  static __device__ __forceinline__ void mul(V& r, const V& x, const V& y) { fp28_mul<C>(r, x, y); }
  static __device__ __forceinline__ void sub(V& r, const V& x, const V& y) { fp28_sub<C>(r, x, y); }
  static __device__ __forceinline__ void add(V& r, const V& x, const V& y) { fp28_add<C>(r, x, y); }
  static __device__ __forceinline__ void konst(V& r, const int32_t (&k)[C::N28]) { fp28_from_const<C>(r, k); }
  static __device__ __forceinline__ void norm(V& r, const V& x) { fp28_normalize<C>(r, x); }
  template <int CTRL>
  static __device__ __forceinline__ int32_t xlane(int32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);  // every quad_perm source lane is valid
#else
    return v;
#endif
  }
  template <int CTRL>
  static __device__ __forceinline__ void perm(V& r, const V& x) {
#pragma unroll
    for (int i = 0; i < C::N28; i++) r.l[i] = xlane<CTRL>(x.l[i]);
  }
  static __device__ __forceinline__ void sel(V& r, unsigned lanes, const V& x, const V& y) {
    fp28_select<C>(r, ((lanes >> lane()) & 1u) != 0, x, y);
  }
  static __device__ __forceinline__ unsigned quad_or(unsigned m) {
    m |= (unsigned)xlane<0xB1>((int32_t)m);  // [1,0,3,2]
    m |= (unsigned)xlane<0x4E>((int32_t)m);  // [2,3,0,1]
    return m;
  }
  static __device__ __forceinline__ unsigned allzero_mask(const V& x) {
    return quad_or((fp28_all_zero<C>(x) ? 1u : 0u) << lane());
  }
  static __device__ __forceinline__ unsigned zero_mask(const V& x) {
    bool z = fp28_maybe_zero<C>(x);
    if (z) z = fp28_is_zero_exact<C>(x);  // practically never reached
    return quad_or((z ? 1u : 0u) << lane());
  }
  static __device__ __forceinline__ void gather(XYZZ28<C>& p, const V& x) {
    perm<0x00>(p.x, x);
    perm<0x55>(p.y, x);
    perm<0xAA>(p.zz, x);
    perm<0xFF>(p.zzz, x);
  }
  static __device__ __forceinline__ void scatter(V& r, const XYZZ28<C>& p) {
    const unsigned q = lane();
    V t;
    fp28_select<C>(t, q == 0, p.x, p.y);
    fp28_select<C>(r, q >= 2, p.zz, t);
    fp28_select<C>(r, q == 3, p.zzz, r);
  }
};
#endif

// the doubling that replaces the addition when both operands are the same point -- reached after empty buckets
// (w0 += acc right after w0 = acc) and on degenerate inputs: every lane doubles the gathered point with the one-lane
// carry-free formulas.  Out of line: its registers do not count against the addition.
template <class C>
MLHIP_HD_NOINLINE void quad28_dbl_slow(XYZZ28<C>& r, const XYZZ28<C>& p) {
  xyzz28_dbl<C>(r, p);
}

// a += b; both hold coordinate `lane` of an XYZZ28 point, normalized; infinity: ZZ all limbs zero
template <class C, class B>
MLHIP_HD void quad28_xyzz_add(typename B::V& a, const typename B::V& b) {
  typedef typename B::V V;
  if (B::allzero_mask(b) & 4u) return;  // b = infinity
  if (B::allzero_mask(a) & 4u) {        // a = infinity
    a = b;
    return;
  }
  V t, m1, o, x, y, d, m2, pp, pP, m3, rr, ppp, qq, X3, rR, e, m4, vv;
  B::template perm<0x4E>(t, b);  // ZZ2 | ZZZ2 | X2 | Y2
  B::mul(m1, a, t);              // U1 | S1 | U2 | S2                               1 x 1
  B::template perm<0x4E>(o, m1);
  B::sel(x, 0x3u, o, m1);
  B::sel(y, 0x3u, m1, o);
  B::sub(d, x, y);  // P | R | P | R                                                  weight 2
  const unsigned zd = B::zero_mask(d);
  if (zd & 1u) {  // same x: the same point (double it) or opposite points (infinity)
    XYZZ28<C> pb, r;
    if (zd & 2u) {
      B::gather(pb, b);
      quad28_dbl_slow<C>(r, pb);
    } else {
      fp28_zero<C>(r.x);
      fp28_zero<C>(r.y);
      fp28_zero<C>(r.zz);
      fp28_zero<C>(r.zzz);
    }
    B::scatter(a, r);
    return;
  }
  B::sel(x, 0x3u, d, a);
  B::sel(y, 0x3u, d, b);
  B::mul(m2, x, y);  // PP | RR | ZZ1 ZZ2 | ZZZ1 ZZZ2                                 2 x 2, 1 x 1
  B::template perm<0x00>(pp, m2);
  B::template perm<0x00>(pP, d);
  B::sel(x, 0x1u, m1, pP);
  B::sel(x, 0x4u, m2, x);
  B::mul(m3, x, pp);  // Q | PPP | ZZ3 | PPP                                          <= 2 x 1
  B::template perm<0x55>(rr, m2);
  B::template perm<0x55>(ppp, m3);
  B::template perm<0x00>(qq, m3);
  B::sub(X3, rr, ppp);
  B::sub(X3, X3, qq);
  B::sub(X3, X3, qq);  // weight 4
  B::norm(X3, X3);     // weight 1 (|value| < 4 p)
  B::template perm<0x55>(rR, d);
  B::sub(e, qq, X3);  // weight 2
  B::sel(x, 0x1u, rR, m1);
  B::sel(x, 0x8u, m2, x);
  B::sel(y, 0x1u, e, m3);
  B::mul(m4, x, y);  // V | T | - | ZZZ3                                              2 x 2, 1 x 1
  B::template perm<0x00>(vv, m4);
  B::sub(y, vv, m4);  // lane 1: Y3, weight 2
  B::norm(y, y);
  B::sel(x, 0x1u, X3, y);
  B::sel(x, 0x4u, m3, x);
  B::sel(a, 0x8u, m4, x);
}

}  // namespace mlhip
