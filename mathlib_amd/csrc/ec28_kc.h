// ec28_kc.h -- G2 bucket accumulation in the carry-free form, lane pairs split BY COORDINATE.
//
// ec28_lp.h splits every Fp2 value of a G2 bucket over its two lanes by component, which makes an Fp2 product one fused
// dual product per lane: 4 limb products + 2 reductions per Fp2 product.  Here the pair splits the POINT instead: lane A
// (even) owns X and ZZ, lane B (odd) owns Y and ZZZ, each as a whole Fp2 value (two Fp28), and an Fp2 product runs on one
// lane as Karatsuba over the components (fp28_k2mul: 3 limb products + 2 reductions, interleaved column by column).  The
// XYZZ mixed addition (add-2008-s mmadd) pairs its eight products so that both lanes always have one to do:
//
//        lane A                         lane B
//   R1   U2 = X2 ZZ1                    S2 = Y2 ZZZ1
//        P  = U2 - X1                   R  = S2 - Y1
//   R2   PP = P^2                       RR = R^2
//   R3   Q  = X1 PP                     PPP = P PP              (P, PP from A)
//   R4   ZZ3 = ZZ1 PP                   ZZZ3 = ZZZ1 PPP
//        X3 = RR - PPP - 2 Q            (RR, PPP from B)
//   R5   V  = R (Q - X3)                T = Y1 PPP              (R from B)
//                                       Y3 = V - T              (V from A)
//
// Per lane: 4 Karatsuba products + 1 square (two single products) = 12 + 2 limb products and 8 + 2 reductions, against the
// 8 dual + 2 single products (18 limb products, 10 reductions) of the component split; five Fp2 values cross the pair per
// addition (28 DPP moves each).  Every product operand is normalized (fp28_k2mul needs weight 1): P, R, X3, Q - X3 and Y3
// are carry-propagated where they are formed.  The reduced integers are those of the component split, so the bucket state
// is bit-identical to ec28_lp.h's and the two kernels share the state buffers and the reduction.
// Written over a backend like ec28_lp.h (device: one Fp2 per lane + DPP; host: a 2-entry array).
#pragma once
#include "ec28_lp.h"

namespace mlhip {

template <class C>
struct Fp2x28 {  // one Fp2 value, both components on this lane
  Fp28<C> c0, c1;
};

template <class C>
struct KcHost {
  struct V {
    Fp2x28<C> v[2];  // [0] = lane A, [1] = lane B
  };
  static constexpr int LANES = 2;
  template <class FN>
  static void each(FN fn) {
    for (int i = 0; i < 2; i++) fn(i);
  }
  static Fp2x28<C>& at(V& x, int i) { return x.v[i]; }
  static const Fp2x28<C>& at(const V& x, int i) { return x.v[i]; }
  static bool is_b(int i) { return i == 1; }
  static void xchg(V& r, const V& a) {
    V t = a;
    r.v[0] = t.v[1];
    r.v[1] = t.v[0];
  }
  static void bcast_a(V& r, const V& a) {
    V t = a;
    r.v[0] = t.v[0];
    r.v[1] = t.v[0];
  }
  static bool both(const bool (&b)[2]) { return b[0] && b[1]; }
  static bool of_a(const bool (&b)[2]) { return b[0]; }
};

#if defined(__HIPCC__)
template <class C>
struct KcDevice {
  typedef Fp2x28<C> V;
  static constexpr int LANES = 1;
  template <class FN>
  static __device__ __forceinline__ void each(FN fn) {
    fn(0);
  }
  static __device__ __forceinline__ V& at(V& x, int) { return x; }
  static __device__ __forceinline__ const V& at(const V& x, int) { return x; }
  static __device__ __forceinline__ bool is_b(int) { return (threadIdx.x & 1u) != 0; }
  static __device__ __forceinline__ int32_t x1(int32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
#else
    return v;
#endif
  }
  static __device__ __forceinline__ int32_t a1(int32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_update_dpp(0, v, 0xA0, 0xF, 0xF, true);  // quad_perm [0,0,2,2]
#else
    return v;
#endif
  }
  static __device__ __forceinline__ void xchg(V& r, const V& a) {
#pragma unroll
    for (int i = 0; i < C::N28; i++) {
      r.c0.l[i] = x1(a.c0.l[i]);
      r.c1.l[i] = x1(a.c1.l[i]);
    }
  }
  static __device__ __forceinline__ void bcast_a(V& r, const V& a) {
#pragma unroll
    for (int i = 0; i < C::N28; i++) {
      r.c0.l[i] = a1(a.c0.l[i]);
      r.c1.l[i] = a1(a.c1.l[i]);
    }
  }
  static __device__ __forceinline__ bool both(const bool (&b)[1]) {
    const int32_t z = b[0] ? 1 : 0;
    return (z & x1(z)) != 0;
  }
  static __device__ __forceinline__ bool of_a(const bool (&b)[1]) { return a1(b[0] ? 1 : 0) != 0; }
};
#endif

#define MLHIP_KC_EACH(B, ...) B::each([&](int li_) { __VA_ARGS__; })

template <class C, class B>
MLHIP_HD void kc_mul(typename B::V& r, const typename B::V& a, const typename B::V& b) {
  static_assert(C::BETA == -1, "the one-lane Karatsuba product assumes u^2 = -1");
  MLHIP_KC_EACH(B, fp28_k2mul<C>(B::at(r, li_).c0, B::at(r, li_).c1, B::at(a, li_).c0, B::at(a, li_).c1, B::at(b, li_).c0,
                                 B::at(b, li_).c1));
}
// r = a^2, a normalized: c0 = (a0 + a1)(a0 - a1), c1 = (2 a0) a1 -- two single products
template <class C, class B>
MLHIP_HD void kc_sqr(typename B::V& r, const typename B::V& a) {
  MLHIP_KC_EACH(B, {
    const Fp2x28<C>& x = B::at(a, li_);
    Fp28<C> s, d, t;
    fp28_add<C>(s, x.c0, x.c1);
    fp28_sub<C>(d, x.c0, x.c1);
    fp28_add<C>(t, x.c0, x.c0);
    Fp28<C> r0, r1;
    fp28_mul<C>(r0, s, d);
    fp28_mul<C>(r1, t, x.c1);
    B::at(r, li_).c0 = r0;
    B::at(r, li_).c1 = r1;
  });
}
template <class C, class B>
MLHIP_HD void kc_sub(typename B::V& r, const typename B::V& a, const typename B::V& b) {
  MLHIP_KC_EACH(B, {
    fp28_sub<C>(B::at(r, li_).c0, B::at(a, li_).c0, B::at(b, li_).c0);
    fp28_sub<C>(B::at(r, li_).c1, B::at(a, li_).c1, B::at(b, li_).c1);
  });
}
template <class C, class B>
MLHIP_HD void kc_normalize(typename B::V& r, const typename B::V& a) {
  MLHIP_KC_EACH(B, {
    fp28_normalize<C>(B::at(r, li_).c0, B::at(a, li_).c0);
    fp28_normalize<C>(B::at(r, li_).c1, B::at(a, li_).c1);
  });
}
// r = lane B ? b_val : a_val
template <class C, class B>
MLHIP_HD void kc_sel_b(typename B::V& r, const typename B::V& b_val, const typename B::V& a_val) {
  MLHIP_KC_EACH(B, {
    const bool b = B::is_b(li_);
    Fp2x28<C> t;
    fp28_select<C>(t.c0, b, B::at(b_val, li_).c0, B::at(a_val, li_).c0);
    fp28_select<C>(t.c1, b, B::at(b_val, li_).c1, B::at(a_val, li_).c1);
    B::at(r, li_) = t;
  });
}

// exceptional cases (q = +-acc): every lane rebuilds the full Fp2 points in the boundary form, runs the one-lane formulas
// and keeps its own two coordinates
template <class C, class B>
MLHIP_HD_NOINLINE void xyzz28_kc_madd_exact(typename B::V& u, typename B::V& z, bool& inf, const typename B::V& q) {
  typedef Fp2Field<C> F2;
  typename B::V xu, xz, xq;
  B::xchg(xu, u);
  B::xchg(xz, z);
  B::xchg(xq, q);
  bool res_inf[B::LANES];
  const bool was_inf = inf;
  MLHIP_KC_EACH(B, {
    const bool b = B::is_b(li_);
    auto to2 = [](Fp2<C>& o, const Fp2x28<C>& v) {
      fp28_to_fp<C>(o.c0, v.c0);
      fp28_to_fp<C>(o.c1, v.c1);
    };
    auto from2 = [](Fp2x28<C>& v, const Fp2<C>& in) {
      fp28_from_fp<C>(v.c0, in.c0);
      fp28_from_fp<C>(v.c1, in.c1);
    };
    auto pick = [](bool c, const Fp2x28<C>& t, const Fp2x28<C>& f) {
      Fp2x28<C> r;
      fp28_select<C>(r.c0, c, t.c0, f.c0);
      fp28_select<C>(r.c1, c, t.c1, f.c1);
      return r;
    };
    XYZZ<F2> a;
    Affine<F2> p;
    if (was_inf) {
      xyzz_set_inf<F2>(a);
    } else {
      to2(a.x, pick(b, B::at(xu, li_), B::at(u, li_)));
      to2(a.y, pick(b, B::at(u, li_), B::at(xu, li_)));
      to2(a.zz, pick(b, B::at(xz, li_), B::at(z, li_)));
      to2(a.zzz, pick(b, B::at(z, li_), B::at(xz, li_)));
    }
    to2(p.x, pick(b, B::at(xq, li_), B::at(q, li_)));
    to2(p.y, pick(b, B::at(q, li_), B::at(xq, li_)));
    xyzz_madd<F2>(a, p, false);
    res_inf[li_] = xyzz_is_inf<F2>(a);
    if (!res_inf[li_]) {
      Fp2<C> mu, mz;
      F2::select(mu, b, a.y, a.x);
      F2::select(mz, b, a.zzz, a.zz);
      from2(B::at(u, li_), mu);
      from2(B::at(z, li_), mz);
    }
  });
  inf = res_inf[0];
}

// bucket (u, z) += q; lane A: u = X, z = ZZ, q = x2; lane B: u = Y, z = ZZZ, q = y2 (negated here when `negate`).
// Pair-uniform control flow.
template <class C, class B>
MLHIP_HD void xyzz28_kc_madd(typename B::V& u, typename B::V& z, bool& inf, const typename B::V& q_in, bool negate) {
  typedef typename B::V V;
  {
    bool zf[B::LANES];
    MLHIP_KC_EACH(B, zf[li_] = fp28_all_zero<C>(B::at(q_in, li_).c0) && fp28_all_zero<C>(B::at(q_in, li_).c1));
    if (B::both(zf)) return;  // point at infinity: x = 0 on A and y = 0 on B
  }
  V q;
  MLHIP_KC_EACH(B, {
    const bool ng = negate && B::is_b(li_);
    Fp28<C> n0, n1;
    fp28_neg<C>(n0, B::at(q_in, li_).c0);
    fp28_neg<C>(n1, B::at(q_in, li_).c1);
    fp28_select<C>(B::at(q, li_).c0, ng, n0, B::at(q_in, li_).c0);
    fp28_select<C>(B::at(q, li_).c1, ng, n1, B::at(q_in, li_).c1);
  });
  if (inf) {
    u = q;
    MLHIP_KC_EACH(B, {
      fp28_from_const<C>(B::at(z, li_).c0, C::ONE28);
      fp28_zero<C>(B::at(z, li_).c1);
    });
    inf = false;
    return;
  }
  V w, d, t;
  kc_mul<C, B>(w, q, z);  // U2 | S2
  kc_sub<C, B>(t, w, u);
  kc_normalize<C, B>(d, t);  // P | R
  {
    bool f[B::LANES];
    MLHIP_KC_EACH(B, f[li_] = fp28_maybe_zero<C>(B::at(d, li_).c0) && fp28_maybe_zero<C>(B::at(d, li_).c1));
    if (B::of_a(f)) {
      MLHIP_KC_EACH(B, f[li_] = fp28_is_zero_exact<C>(B::at(d, li_).c0) && fp28_is_zero_exact<C>(B::at(d, li_).c1));
      if (B::of_a(f)) {  // P = 0: doubling or cancellation
        V tu = u, tz = z, tq = q;  // cold-path copies: keep the caller's accumulator in registers
        bool ti = inf;
        xyzz28_kc_madd_exact<C, B>(tu, tz, ti, tq);
        u = tu;
        z = tz;
        inf = ti;
        return;
      }
    }
  }
  V sq, dx, sqx, sqa, opa, opb, r3, r3x, x3, e, r5, r5x, y3;
  kc_sqr<C, B>(sq, d);        // PP | RR
  B::xchg(dx, d);             // R | P
  B::bcast_a(sqa, sq);        // PP | PP
  kc_sel_b<C, B>(opa, dx, u);  // X1 | P
  kc_mul<C, B>(r3, opa, sqa);  // Q | PPP
  kc_sel_b<C, B>(opb, r3, sq);  // PP | PPP
  kc_mul<C, B>(t, z, opb);      // ZZ3 | ZZZ3
  z = t;
  B::xchg(sqx, sq);   // RR | PP
  B::xchg(r3x, r3);   // PPP | Q
  kc_sub<C, B>(t, sqx, r3x);
  kc_sub<C, B>(t, t, r3);
  kc_sub<C, B>(t, t, r3);
  kc_normalize<C, B>(x3, t);  // X3 | (unused)
  kc_sub<C, B>(t, r3, x3);
  kc_normalize<C, B>(e, t);   // Q - X3 | (unused)
  kc_sel_b<C, B>(opa, u, dx);  // R | Y1
  kc_sel_b<C, B>(opb, r3, e);  // Q - X3 | PPP
  kc_mul<C, B>(r5, opa, opb);  // V | T
  B::xchg(r5x, r5);            // T | V
  kc_sub<C, B>(t, r5x, r5);    // (unused) | V - T
  kc_normalize<C, B>(y3, t);
  kc_sel_b<C, B>(u, y3, x3);   // X3 | Y3
}

}  // namespace mlhip
