// TEST ARTIFACT -- host (g++) build of the kernels' __host__ __device__ math headers, loaded by
// tests/test_host_math.py through ctypes and compared with oracle/pyref.py.  It lets the field /
// curve / pairing formulas be verified in the GPU-less container; it is NOT part of libmlhip.so
// and nothing in the product path links or loads it.
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "../../mathlib_amd/csrc/pairing.h"
#include "../../mathlib_amd/csrc/msm_body.h"
#include "../../mathlib_amd/csrc/msm_fold_body.h"
#include "../../mathlib_amd/csrc/codec.h"
#include "../../mathlib_amd/csrc/ec28.h"
#include "../../mathlib_amd/csrc/ec_jac.h"
#include "../../mathlib_amd/csrc/ec_quad.h"
#include "../../mathlib_amd/csrc/ec_quad28.h"
#include "../../mathlib_amd/csrc/ec28_lp.h"
#include "../../mathlib_amd/csrc/ec28_kc.h"
#include "../../mathlib_amd/csrc/ed28.h"
#include "../../mathlib_amd/csrc/modinv.h"
#include "../../mathlib_amd/csrc/fp2_lanes28.h"
#include "../../mathlib_amd/csrc/pairing_quad.h"

using namespace mlhip;

template <class C>
struct Ops {
  typedef Fp<C> F;
  typedef Fp2<C> F2;
  typedef Fp12<C> F12;
  typedef Affine<FpField<C>> A1;
  typedef Affine<Fp2Field<C>> A2;
  typedef XYZZ<FpField<C>> X1;
  typedef XYZZ<Fp2Field<C>> X2;

  static int fp_op(int op, const void* a, const void* b, void* out) {
    F x, y, r;
    memcpy(&x, a, sizeof(F));
    if (b) memcpy(&y, b, sizeof(F));
    switch (op) {
      case 0: fp_mul<C>(r, x, y); break;
      case 1: fp_add<C>(r, x, y); break;
      case 2: fp_sub<C>(r, x, y); break;
      case 3: fp_neg<C>(r, x); break;
      case 4: fp_inv<C>(r, x); break;
      case 5: fp_sqr<C>(r, x); break;
      case 6: fp_halve<C>(r, x); break;
      case 7: fp_to_mont<C>(r, x); break;
      case 8: fp_from_mont<C>(r, x); break;
      case 9: fp_inv_divsteps<C>(r, x); break;
      case 10: fp_inv_fermat<C>(r, x); break;
      default: return -1;
    }
    memcpy(out, &r, sizeof(F));
    return 0;
  }
  static int fp2_op(int op, const void* a, const void* b, void* out) {
    F2 x, y, r;
    memcpy(&x, a, sizeof(F2));
    if (b) memcpy(&y, b, sizeof(F2));
    switch (op) {
      case 0: fp2_mul<C>(r, x, y); break;
      case 1: fp2_sqr<C>(r, x); break;
      case 2: fp2_inv<C>(r, x); break;
      case 3: fp2_mul_xi<C>(r, x); break;
      case 4:  // square root; returns 1 when x is not a square
        if (!fp2_sqrt<C>(r, x)) return 1;
        break;
      default: return -1;
    }
    memcpy(out, &r, sizeof(F2));
    return 0;
  }
  static int fp12_op(int op, const void* a, const void* b, void* out) {
    F12 x, y, r;
    memcpy(&x, a, sizeof(F12));
    if (b) memcpy(&y, b, sizeof(F12));
    switch (op) {
      case 0: fp12_mul<C>(r, x, y); break;
      case 1: fp12_sqr<C>(r, x); break;
      case 2: fp12_inv<C>(r, x); break;
      case 3: fp12_frob<C, 1>(r, x); break;
      case 4: fp12_frob<C, 2>(r, x); break;
      case 5: fp12_frob<C, 3>(r, x); break;
      case 6: fp12_cyclo_sqr<C>(r, x); break;
      case 7: fp12_conj<C>(r, x); break;
      case 8: fp12_expt<C>(r, x); break;
      case 9: final_exp<C>(r, x); break;
      default: return -1;
    }
    memcpy(out, &r, sizeof(F12));
    return 0;
  }
  // sum_i (+/-) P_i by mixed additions into one XYZZ accumulator
  static int g1_sum(const void* pts, const uint8_t* neg, int n, void* out) {
    X1 acc;
    xyzz_set_inf<FpField<C>>(acc);
    const A1* p = (const A1*)pts;
    for (int i = 0; i < n; i++) xyzz_madd<FpField<C>>(acc, p[i], neg && neg[i]);
    A1 r;
    xyzz_to_affine<FpField<C>>(r, acc);
    memcpy(out, &r, sizeof(A1));
    return 0;
  }
  static int g2_sum(const void* pts, const uint8_t* neg, int n, void* out) {
    X2 acc;
    xyzz_set_inf<Fp2Field<C>>(acc);
    const A2* p = (const A2*)pts;
    for (int i = 0; i < n; i++) xyzz_madd<Fp2Field<C>>(acc, p[i], neg && neg[i]);
    A2 r;
    xyzz_to_affine<Fp2Field<C>>(r, acc);
    memcpy(out, &r, sizeof(A2));
    return 0;
  }
  // tree sum through xyzz_add (exercises the XYZZ+XYZZ path incl. doubling / inverse branches)
  static int g1_tree(const void* pts, int n, void* out) {
    const A1* p = (const A1*)pts;
    X1* v = new X1[n > 0 ? n : 1];
    for (int i = 0; i < n; i++) xyzz_from_affine<FpField<C>>(v[i], p[i]);
    int m = n;
    while (m > 1) {
      int h = (m + 1) / 2;
      for (int i = 0; i + h < m; i++) xyzz_add<FpField<C>>(v[i], v[i + h]);
      m = h;
    }
    X1 acc;
    if (n > 0) acc = v[0]; else xyzz_set_inf<FpField<C>>(acc);
    A1 r;
    xyzz_to_affine<FpField<C>>(r, acc);
    memcpy(out, &r, sizeof(A1));
    delete[] v;
    return 0;
  }
  static int g2_tree(const void* pts, int n, void* out) {
    const A2* p = (const A2*)pts;
    X2* v = new X2[n > 0 ? n : 1];
    for (int i = 0; i < n; i++) xyzz_from_affine<Fp2Field<C>>(v[i], p[i]);
    int m = n;
    while (m > 1) {
      int h = (m + 1) / 2;
      for (int i = 0; i + h < m; i++) xyzz_add<Fp2Field<C>>(v[i], v[i + h]);
      m = h;
    }
    X2 acc;
    if (n > 0) acc = v[0]; else xyzz_set_inf<Fp2Field<C>>(acc);
    A2 r;
    xyzz_to_affine<Fp2Field<C>>(r, acc);
    memcpy(out, &r, sizeof(A2));
    delete[] v;
    return 0;
  }
  // window digits of one scalar, exactly as k_digits computes them (n = 1 layout: digits[w])
  static int digits(const void* scalar, int mont, int c, uint32_t* out, int cap) {
    int W = msm_num_windows(C::FR_BITS, c);
    if (W > cap) return -3;
    msm_digits_body<C>(0, 1, (const uint32_t*)scalar, mont != 0, c, W, out);
    return W;
  }
  // level-1 bucket reduction body over an array of n_chunks*L affine "buckets"; returns A and W0 as affine
  static int chunks(const void* pts, int n_chunks, void* outA, void* outW0) {
    const A1* p = (const A1*)pts;
    std::vector<X1> b(n_chunks * 8), A(n_chunks), W0(n_chunks);
    for (int i = 0; i < n_chunks * 8; i++) xyzz_from_affine<FpField<C>>(b[i], p[i]);
    for (int g = 0; g < n_chunks; g++) msm_chunk_body<FpField<C>>(g, b.data(), A.data(), W0.data(), 8,
                                                                  [](X1& a, const X1& q) { xyzz_add<FpField<C>>(a, q); });
    for (int g = 0; g < n_chunks; g++) {
      A1 r;
      xyzz_to_affine<FpField<C>>(r, A[g]);
      memcpy((char*)outA + g * sizeof(A1), &r, sizeof(A1));
      xyzz_to_affine<FpField<C>>(r, W0[g]);
      memcpy((char*)outW0 + g * sizeof(A1), &r, sizeof(A1));
    }
    return 0;
  }
  // the Horner pass of the host tail (msm_plan.h: host_tail): sum_w 2^off[w] V[w] over affine inputs V[w], in XYZZ (which = 0,
  // rounds 1-3) or Jacobian coordinates (which = 1, ec_jac.h); group 1 or 2; affine result
  static int horner(int group, const void* pts, int W, const int* down, int which, void* out) {
    if (group == 1) {
      const A1* p = (const A1*)pts;
      std::vector<X1> V(W);
      for (int i = 0; i < W; i++) xyzz_from_affine<FpField<C>>(V[i], p[i]);
      X1 t;
      if (which) horner_jac<FpField<C>>(t, V.data(), W, down); else horner_xyzz<FpField<C>>(t, V.data(), W, down);
      A1 r;
      xyzz_to_affine<FpField<C>>(r, t);
      memcpy(out, &r, sizeof(A1));
    } else {
      const A2* p = (const A2*)pts;
      std::vector<X2> V(W);
      for (int i = 0; i < W; i++) xyzz_from_affine<Fp2Field<C>>(V[i], p[i]);
      X2 t;
      if (which) horner_jac<Fp2Field<C>>(t, V.data(), W, down); else horner_xyzz<Fp2Field<C>>(t, V.data(), W, down);
      A2 r;
      xyzz_to_affine<Fp2Field<C>>(r, t);
      memcpy(out, &r, sizeof(A2));
    }
    return 0;
  }
  // A whole MSM over shifted-base tables (msm_fold.h), replayed on the host with the kernels' own bodies: the table rows
  // (fold_rows_body), the signed digits of the even layout into ONE bucket set (msm_window_digit, xyzz_madd), the bucket
  // groups reduced like windows (msm_chunk_body, the plain / masked sums of k_masked_sums), the groups combined into one
  // window (fold_combine_src, k_group_combine_q) and that window's host tail (host_tail_window).  Digit width c, groups
  // of 2^lgM buckets, chunks of 2^lgL.  group 1 or 2; affine result.
  template <class F>
  static int fold_msm_t(const void* pts, const void* scalars, int n, int c, int lgM, int lgL, void* out) {
    typedef Affine<F> A;
    typedef XYZZ<F> X;
    const WinLayout wl = msm_win_layout(C::FR_BITS, c);
    const int Wd = wl.W;
    const size_t nbuckets = (size_t)1 << (c - 1);
    const size_t M = std::min(nbuckets, (size_t)1 << lgM), W = nbuckets / M;
    const size_t L = (size_t)1 << lgL, T = M / L;
    if (T < 1 || (T & (T - 1)) || W > 32) return -4;
    int nb = 0;
    while (((size_t)1 << nb) < T) nb++;
    const int nsel = 4 + nb;
    const A* P = (const A*)pts;
    std::vector<A> rows((size_t)Wd * n);
    for (int i = 0; i < n; i++) fold_rows_body<F>(P[i], C::FR_BITS, c, (size_t)n, rows.data() + i);
    std::vector<X> B(nbuckets);
    for (auto& b : B) xyzz_set_inf<F>(b);
    for (int i = 0; i < n; i++) {
      uint32_t s[8];
      fr_canonical<C>(s, (const uint32_t*)scalars + 8 * i, false);
      uint32_t carry = 0, neg = 0;
      for (int w = 0; w < Wd; w++) {
        const uint32_t mag = msm_window_digit(s, msm_win_off(wl.base, wl.rem, w), msm_win_bits(wl.base, wl.rem, w), carry, neg);
        if (mag) xyzz_madd<F>(B[mag - 1], rows[(size_t)w * n + i], neg != 0);
      }
    }
    // per group: chunk sums, then the nsel sums of k_masked_sums
    std::vector<X> outg(W * nsel);
    for (size_t g = 0; g < W; g++) {
      std::vector<X> Ach(T), W0(T);
      for (size_t t = 0; t < T; t++)
        msm_chunk_body<F>(t, B.data() + g * M, Ach.data(), W0.data(), (int)L, [](X& a, const X& q) { xyzz_add<F>(a, q); });
      X* o = outg.data() + g * nsel;
      for (int k = 0; k < nsel; k++) xyzz_set_inf<F>(o[k]);
      const size_t half = (T + 1) / 2;
      for (size_t t = 0; t < T; t++) {
        xyzz_add<F>(o[t < half ? 0 : 1], W0[t]);
        xyzz_add<F>(o[t < half ? 2 : 3], Ach[t]);
        for (int k = 0; k < nb; k++)
          if ((t >> k) & 1) xyzz_add<F>(o[4 + k], Ach[t]);
      }
    }
    // the groups combined into one window of W T chunks
    int lgW = 0;
    while (((size_t)1 << lgW) < W) lgW++;
    const int nsel2 = 4 + nb + lgW;
    std::vector<unsigned char> blob(sizeof(HostTailHeader) + (size_t)nsel2 * sizeof(X));
    const HostTailHeader h{nb + lgW, lgL, nsel2, 0};
    memcpy(blob.data(), &h, sizeof(h));
    X* comb = reinterpret_cast<X*>(blob.data() + sizeof(h));
    for (int o = 0; o < nsel2; o++) {
      X acc;
      xyzz_set_inf<F>(acc);
      for (size_t g = 0; g < W; g++)
        for (int hh = 0; hh < 2; hh++) {
          const int src = fold_combine_src(o, (int)g, hh, nb);
          if (src >= 0) xyzz_add<F>(acc, outg[g * nsel + src]);
        }
      memcpy(&comb[o], &acc, sizeof(X));
    }
    X total;
    host_tail_window<F>(blob.data(), 0, &total);
    A r;
    xyzz_to_affine<F>(r, total);
    memcpy(out, &r, sizeof(A));
    return Wd;
  }
  static int fold_msm(int group, const void* pts, const void* scalars, int n, int c, int lgM, int lgL, void* out) {
    return group == 1 ? fold_msm_t<FpField<C>>(pts, scalars, n, c, lgM, lgL, out) : fold_msm_t<Fp2Field<C>>(pts, scalars, n, c, lgM, lgL, out);
  }
  static int g1dec(const uint8_t* w, int compressed, int subgroup, void* out) {
    A1 p;
    int st = g1_decode<C>(p, w, compressed != 0, subgroup);
    memcpy(out, &p, sizeof(A1));
    return st;
  }
  static int g1enc(const void* pt, int compressed, uint8_t* w) {
    A1 p;
    memcpy(&p, pt, sizeof(A1));
    g1_encode<C>(w, p, compressed != 0);
    return 0;
  }
  // carry-free 28-bit-limb form (fp28.h): op 0 round trip, 1 mul, 2 sqr, 3 dual product a b + c d, all through the
  // boundary conversions; op 4: (a - b) is zero mod p ?  returns the exact test's answer in out[0]
  static int fp28_op(int op, const void* a, const void* b, const void* c, const void* d, void* out) {
    F x[4], r;
    const void* in[4] = {a, b, c, d};
    Fp28<C> v[4], w, t;
    for (int i = 0; i < 4; i++) {
      if (!in[i]) { fp_zero<C>(x[i]); } else memcpy(&x[i], in[i], sizeof(F));
      fp28_from_fp<C>(v[i], x[i]);
    }
    switch (op) {
      case 0: w = v[0]; break;
      case 1: fp28_mul<C>(w, v[0], v[1]); break;
      case 2: fp28_sqr<C>(w, v[0]); break;
      case 3: fp28_mul2<C>(w, v[0], v[1], v[2], v[3]); break;
      case 4: {
        fp28_sub<C>(w, v[0], v[1]);
        bool z = fp28_maybe_zero<C>(w) && fp28_is_zero_exact<C>(w);
        *(int*)out = z ? 1 : 0;
        return 0;
      }
      case 5:  // stress the weight limits: (a + b) (c - d) with weights 2 x 2, then squared sum (weight 2)
        fp28_add<C>(w, v[0], v[1]);
        fp28_sub<C>(t, v[2], v[3]);
        fp28_mul<C>(w, w, t);
        break;
      case 6:  // ((a - b) - c - c) normalized (weight 4 -> 1), times d
        fp28_sub<C>(w, v[0], v[1]);
        fp28_sub<C>(w, w, v[2]);
        fp28_sub<C>(w, w, v[2]);
        fp28_normalize<C>(w, w);
        fp28_mul<C>(w, w, v[3]);
        break;
      case 7:  // (a - b)^2
        fp28_sub<C>(w, v[0], v[1]);
        fp28_sqr<C>(w, w);
        break;
      default: return -1;
    }
    fp28_to_fp<C>(r, w);
    memcpy(out, &r, sizeof(F));
    return 0;
  }
  // bucket accumulation through xyzz28_madd: sum of +-points, returned affine in the boundary form
  static int madd28_chain(const void* pts, const uint8_t* neg, int n, void* out) {
    const A1* p = (const A1*)pts;
    XYZZ28<C> acc;
    bool inf = true;
    for (int i = 0; i < n; i++) {
      Affine28<C> q;
      affine28_from<C>(q, p[i]);
      xyzz28_madd<C>(acc, inf, q, neg[i] != 0);
    }
    X1 a;
    xyzz28_to<C>(a, acc, inf);
    A1 r;
    xyzz_to_affine<FpField<C>>(r, a);
    memcpy(out, &r, sizeof(A1));
    return 0;
  }
  // quad-lane XYZZ addition (ec_quad.h) through the host emulation backend: fold a list of XYZZ points given as
  // affine inputs scaled by per-point z (so that ZZ != 1): out = affine sum
  static int quad_chain(const void* pts, const void* zs, int n, void* out) {
    typedef QuadHost<C> B;
    const A1* p = (const A1*)pts;
    const F* z = (const F*)zs;
    typename B::V acc;
    X1 inf;
    xyzz_set_inf<FpField<C>>(inf);
    B::scatter(acc, inf);
    for (int i = 0; i < n; i++) {
      X1 q;
      xyzz_from_affine<FpField<C>>(q, p[i]);
      if (!xyzz_is_inf<FpField<C>>(q) && !fp_is_zero<C>(z[i])) {  // (x z^2, y z^3, z^2, z^3) is the same point
        F z2, z3;
        fp_sqr<C>(z2, z[i]);
        fp_mul<C>(z3, z2, z[i]);
        fp_mul<C>(q.x, q.x, z2);
        fp_mul<C>(q.y, q.y, z3);
        q.zz = z2;
        q.zzz = z3;
      }
      typename B::V b;
      B::scatter(b, q);
      quad_xyzz_add<C, B>(acc, b);
    }
    X1 r;
    B::gather(r, acc);
    A1 a;
    xyzz_to_affine<FpField<C>>(a, r);
    memcpy(out, &a, sizeof(A1));
    return 0;
  }
  // the same fold through the carry-free quad addition (ec_quad28.h, QuadHost28)
  static int quad28_chain(const void* pts, const void* zs, int n, void* out) {
    typedef QuadHost28<C> B;
    const A1* p = (const A1*)pts;
    const F* z = (const F*)zs;
    typename B::V acc;
    for (int i = 0; i < 4; i++) fp28_zero<C>(acc.v[i]);
    for (int i = 0; i < n; i++) {
      X1 q;
      xyzz_from_affine<FpField<C>>(q, p[i]);
      typename B::V b;
      if (xyzz_is_inf<FpField<C>>(q)) {
        for (int k = 0; k < 4; k++) fp28_zero<C>(b.v[k]);
      } else {
        if (!fp_is_zero<C>(z[i])) {
          F z2, z3;
          fp_sqr<C>(z2, z[i]);
          fp_mul<C>(z3, z2, z[i]);
          fp_mul<C>(q.x, q.x, z2);
          fp_mul<C>(q.y, q.y, z3);
          q.zz = z2;
          q.zzz = z3;
        }
        fp28_from_fp<C>(b.v[0], q.x);
        fp28_from_fp<C>(b.v[1], q.y);
        fp28_from_fp<C>(b.v[2], q.zz);
        fp28_from_fp<C>(b.v[3], q.zzz);
      }
      quad28_xyzz_add<C, B>(acc, b);
      for (int k = 0; k < 4; k++)  // every stored coordinate stays normalized
        for (int j = 0; j < C::N28 - 1; j++)
          if (acc.v[k].l[j] < 0 || acc.v[k].l[j] >= (1 << 28)) return -3;
    }
    XYZZ28<C> r28;
    B::gather(r28, acc);
    X1 r;
    xyzz28_to<C>(r, r28, fp28_all_zero<C>(r28.zz));
    A1 a;
    xyzz_to_affine<FpField<C>>(a, r);
    memcpy(out, &a, sizeof(A1));
    return 0;
  }
  // G2 full additions in the carry-free lane-pair form (ec28_lp.h: xyzz28_lp_add): fold XYZZ points given as affine
  // inputs scaled by per-point z in Fp2 (so that ZZ != 1)
  static int add28_lp_chain(const void* pts, const void* zs, int n, void* out) {
    if constexpr (C::N28 > 0) {  // every curve: u^2 = -1 and, since round 3, u^2 = -5
      typedef PairHost<C> B;
      typedef Fp2Field<C> F2;
      const A2* p = (const A2*)pts;
      const Fp2<C>* z = (const Fp2<C>*)zs;
      XYZZ28L<typename B::V> acc;
      bool inf = true;
      for (int i = 0; i < n; i++) {
        X2 q;
        xyzz_from_affine<F2>(q, p[i]);
        const bool q_inf = xyzz_is_inf<F2>(q);
        XYZZ28L<typename B::V> b;
        if (!q_inf) {
          if (!F2::is_zero(z[i])) {
            Fp2<C> z2, z3;
            F2::sqr(z2, z[i]);
            F2::mul(z3, z2, z[i]);
            F2::mul(q.x, q.x, z2);
            F2::mul(q.y, q.y, z3);
            q.zz = z2;
            q.zzz = z3;
          }
          fp28_from_fp<C>(b.x.v[0], q.x.c0);
          fp28_from_fp<C>(b.x.v[1], q.x.c1);
          fp28_from_fp<C>(b.y.v[0], q.y.c0);
          fp28_from_fp<C>(b.y.v[1], q.y.c1);
          fp28_from_fp<C>(b.zz.v[0], q.zz.c0);
          fp28_from_fp<C>(b.zz.v[1], q.zz.c1);
          fp28_from_fp<C>(b.zzz.v[0], q.zzz.c0);
          fp28_from_fp<C>(b.zzz.v[1], q.zzz.c1);
        }
        xyzz28_lp_add<C, B>(acc, inf, b, q_inf);
      }
      X2 a;
      if (inf) {
        xyzz_set_inf<F2>(a);
      } else {
        fp28_to_fp<C>(a.x.c0, acc.x.v[0]);
        fp28_to_fp<C>(a.x.c1, acc.x.v[1]);
        fp28_to_fp<C>(a.y.c0, acc.y.v[0]);
        fp28_to_fp<C>(a.y.c1, acc.y.v[1]);
        fp28_to_fp<C>(a.zz.c0, acc.zz.v[0]);
        fp28_to_fp<C>(a.zz.c1, acc.zz.v[1]);
        fp28_to_fp<C>(a.zzz.c0, acc.zzz.v[0]);
        fp28_to_fp<C>(a.zzz.c1, acc.zzz.v[1]);
      }
      A2 r;
      xyzz_to_affine<F2>(r, a);
      memcpy(out, &r, sizeof(A2));
      return 0;
    } else {
      return -2;
    }
  }
  // G2 bucket accumulation in the carry-free lane-pair form (ec28_lp.h) through the host emulation backend
  static int madd28_lp_chain(const void* pts, const uint8_t* neg, int n, void* out) {
    if constexpr (C::N28 > 0) {  // every curve: u^2 = -1 and, since round 3, u^2 = -5
      typedef PairHost<C> B;
      const A2* p = (const A2*)pts;
      XYZZ28L<typename B::V> acc;
      bool inf = true;
      for (int i = 0; i < n; i++) {
        Affine28L<typename B::V> q;
        fp28_from_fp<C>(q.x.v[0], p[i].x.c0);
        fp28_from_fp<C>(q.x.v[1], p[i].x.c1);
        fp28_from_fp<C>(q.y.v[0], p[i].y.c0);
        fp28_from_fp<C>(q.y.v[1], p[i].y.c1);
        xyzz28_lp_madd<C, B>(acc, inf, q, neg[i] != 0);
      }
      X2 a;
      if (inf) {
        xyzz_set_inf<Fp2Field<C>>(a);
      } else {
        fp28_to_fp<C>(a.x.c0, acc.x.v[0]);
        fp28_to_fp<C>(a.x.c1, acc.x.v[1]);
        fp28_to_fp<C>(a.y.c0, acc.y.v[0]);
        fp28_to_fp<C>(a.y.c1, acc.y.v[1]);
        fp28_to_fp<C>(a.zz.c0, acc.zz.v[0]);
        fp28_to_fp<C>(a.zz.c1, acc.zz.v[1]);
        fp28_to_fp<C>(a.zzz.c0, acc.zzz.v[0]);
        fp28_to_fp<C>(a.zzz.c1, acc.zzz.v[1]);
      }
      A2 r;
      xyzz_to_affine<Fp2Field<C>>(r, a);
      memcpy(out, &r, sizeof(A2));
      return 0;
    } else {
      return -2;  // u^2 != -1: this path is not built for the curve
    }
  }
  // the same accumulation with the pair split by coordinate (ec28_kc.h: lane A owns X and ZZ, lane B Y and ZZZ, one-lane
  // Karatsuba products); besides the sum it checks that the state is bit-identical to the component split's after every
  // addition (returns -4 otherwise) and that every stored coordinate has weight 1 (-3)
  static int madd28_kc_chain(const void* pts, const uint8_t* neg, int n, void* out) {
    if constexpr (C::BETA == -1) {
      typedef KcHost<C> B;
      typedef PairHost<C> BL;
      const A2* p = (const A2*)pts;
      typename B::V u, z;
      XYZZ28L<typename BL::V> ref;
      bool inf = true, ref_inf = true;
      for (int i = 0; i < n; i++) {
        typename B::V q;
        fp28_from_fp<C>(q.v[0].c0, p[i].x.c0);
        fp28_from_fp<C>(q.v[0].c1, p[i].x.c1);
        fp28_from_fp<C>(q.v[1].c0, p[i].y.c0);
        fp28_from_fp<C>(q.v[1].c1, p[i].y.c1);
        xyzz28_kc_madd<C, B>(u, z, inf, q, neg[i] != 0);
        Affine28L<typename BL::V> ql;
        ql.x.v[0] = q.v[0].c0;
        ql.x.v[1] = q.v[0].c1;
        ql.y.v[0] = q.v[1].c0;
        ql.y.v[1] = q.v[1].c1;
        xyzz28_lp_madd<C, BL>(ref, ref_inf, ql, neg[i] != 0);
        if (inf != ref_inf) return -4;
        if (!inf) {
          const Fp28<C>* mine[8] = {&u.v[0].c0, &u.v[0].c1, &u.v[1].c0, &u.v[1].c1, &z.v[0].c0, &z.v[0].c1, &z.v[1].c0, &z.v[1].c1};
          const Fp28<C>* theirs[8] = {&ref.x.v[0], &ref.x.v[1], &ref.y.v[0], &ref.y.v[1], &ref.zz.v[0], &ref.zz.v[1], &ref.zzz.v[0], &ref.zzz.v[1]};
          for (int k = 0; k < 8; k++) {
            if (memcmp(mine[k], theirs[k], sizeof(Fp28<C>)) != 0) return -4;
            for (int j = 0; j < C::N28 - 1; j++)
              if (mine[k]->l[j] <= -(1 << 28) || mine[k]->l[j] >= (1 << 28)) return -3;  // weight 1 (a negated first point keeps its sign)
          }
        }
      }
      X2 a;
      if (inf) {
        xyzz_set_inf<Fp2Field<C>>(a);
      } else {
        fp28_to_fp<C>(a.x.c0, u.v[0].c0);
        fp28_to_fp<C>(a.x.c1, u.v[0].c1);
        fp28_to_fp<C>(a.y.c0, u.v[1].c0);
        fp28_to_fp<C>(a.y.c1, u.v[1].c1);
        fp28_to_fp<C>(a.zz.c0, z.v[0].c0);
        fp28_to_fp<C>(a.zz.c1, z.v[0].c1);
        fp28_to_fp<C>(a.zzz.c0, z.v[1].c0);
        fp28_to_fp<C>(a.zzz.c1, z.v[1].c1);
      }
      A2 r;
      xyzz_to_affine<Fp2Field<C>>(r, a);
      memcpy(out, &r, sizeof(A2));
      return 0;
    } else {
      return -2;
    }
  }
  // G1 bucket accumulation in extended twisted Edwards coordinates (ed28.h; curves with the model: BLS12-377).  mode 0:
  // points converted in batches of four (one shared inversion), ed28_madd chain; mode 1: every point converted alone to
  // extended coordinates and folded with the full addition ed28_add.  The sum returns through ed28_to_xyzz28.
  static int ed28_chain(const void* pts, const uint8_t* neg, int n, int mode, void* out) {
    if constexpr (C::HAS_EDWARDS) {
      const A1* p = (const A1*)pts;
      EdExt28<C> acc;
      ed28_set_identity<C>(acc);
      if (mode == 0) {
        for (int i0 = 0; i0 < n; i0 += 4) {
          A1 in[4];
          Fp<C> xh[4], yh[4];
          for (int j = 0; j < 4; j++) {
            if (i0 + j < n) {
              in[j] = p[i0 + j];
            } else {
              fp_zero<C>(in[j].x);
              fp_zero<C>(in[j].y);
            }
          }
          ed_affine_halves_batch<C, 4>(xh, yh, in);
          for (int j = 0; j < 4 && i0 + j < n; j++) {
            EdNiels28<C> q;
            ed_niels_from_halves<C>(q, xh[j], yh[j]);
            ed28_madd<C>(acc, q, neg[i0 + j] != 0);
            const Fp28<C>* co[4] = {&acc.x, &acc.y, &acc.z, &acc.t};
            for (int k = 0; k < 4; k++)  // every stored coordinate stays normalized
              for (int l = 0; l < C::N28 - 1; l++)
                if (co[k]->l[l] < 0 || co[k]->l[l] >= (1 << 28)) return -3;
          }
        }
      } else if (mode == 1) {
        for (int i = 0; i < n; i++) {
          A1 q = p[i];
          if (neg[i]) fp_neg<C>(q.y, q.y);
          EdExt28<C> e;
          ed28_from_affine<C>(e, q);
          ed28_add<C>(acc, e);
        }
      } else {  // the quad-lane schedule of the reduction (ed_quad28_add) through the host emulation of the permutes;
                // mode 3 also adds every partial sum to itself once (equal operands) and takes that back out
        typedef QuadHost28<C> B;
        typename B::V qa;
        qa.v[0] = acc.x;
        qa.v[1] = acc.y;
        qa.v[2] = acc.z;
        qa.v[3] = acc.t;
        for (int i = 0; i < n; i++) {
          A1 q = p[i];
          if (neg[i]) fp_neg<C>(q.y, q.y);
          EdExt28<C> e;
          ed28_from_affine<C>(e, q);
          typename B::V qb;
          qb.v[0] = e.x;
          qb.v[1] = e.y;
          qb.v[2] = e.z;
          qb.v[3] = e.t;
          ed_quad28_add<C, B>(qa, qb);
          if (mode == 3) {
            typename B::V dbl = qa, neg1 = qa;
            ed_quad28_add<C, B>(dbl, qa);  // 2 S
            fp28_neg<C>(neg1.v[0], qa.v[0]);  // -S: (-X : Y : Z : -T)
            fp28_neg<C>(neg1.v[3], qa.v[3]);
            ed_quad28_add<C, B>(dbl, neg1);  // 2 S - S
            qa = dbl;
          }
          for (int k = 0; k < 4; k++)
            for (int l = 0; l < C::N28 - 1; l++)
              if (qa.v[k].l[l] < 0 || qa.v[k].l[l] >= (1 << 28)) return -3;
        }
        acc.x = qa.v[0];
        acc.y = qa.v[1];
        acc.z = qa.v[2];
        acc.t = qa.v[3];
      }
      XYZZ28<C> w;
      bool inf;
      ed28_to_xyzz28<C>(w, inf, acc);
      X1 r;
      xyzz28_to<C>(r, w, inf);
      A1 a;
      xyzz_to_affine<FpField<C>>(a, r);
      memcpy(out, &a, sizeof(A1));
      return 0;
    } else {
      return -2;
    }
  }
  static int g2dec(const uint8_t* w, int compressed, int subgroup, void* out) {
    A2 p;
    int st = g2_decode<C>(p, w, compressed != 0, subgroup);
    memcpy(out, &p, sizeof(A2));
    return st;
  }
  static int g2enc(const void* pt, int compressed, uint8_t* w) {
    A2 p;
    memcpy(&p, pt, sizeof(A2));
    g2_encode<C>(w, p, compressed != 0);
    return 0;
  }
  static int miller(const void* g1s, const void* g2s, int n_pairs, void* out) {
    F12 f;
    miller_loop<C, 4>(f, (const A1*)g1s, (const A2*)g2s, n_pairs);
    memcpy(out, &f, sizeof(F12));
    return 0;
  }
};

// ---- the carry-free lane-pair element (fp2_lanes28.h) through its host model Fp2H28: same tower / pairing templates
// as the kernels, every operation checking its weight budget (aborts with a message when one is exceeded)
template <class CC>
struct Lp28T {
  typedef CC C;
  typedef Fp2H28<C> E;
  typedef Fp12<C, E> F12h;
  static void to_h(E& r, const Fp2<C>& a) {
    fp28_from_fp<C>(r.c[0], a.c0);
    fp28_from_fp<C>(r.c[1], a.c1);
    r.wt = 1;
  }
  static void from_h(Fp2<C>& r, const E& a) {
    fp28_to_fp<C>(r.c0, a.c[0]);
    fp28_to_fp<C>(r.c1, a.c[1]);
  }
  static void to_h12(F12h& r, const Fp12<C>& a) {
    const Fp2<C>* s = &a.c0.c0;
    E* d = &r.c0.c0;
    for (int i = 0; i < 6; i++) to_h(d[i], s[i]);
  }
  static void from_h12(Fp12<C>& r, const F12h& a) {
    Fp2<C>* d = &r.c0.c0;
    const E* s = &a.c0.c0;
    for (int i = 0; i < 6; i++) from_h(d[i], s[i]);
  }
  static int max_w(const F12h& a) {
    const E* s = &a.c0.c0;
    int w = 0;
    for (int i = 0; i < 6; i++) w = s[i].wt > w ? s[i].wt : w;
    return w;
  }
  // returns the largest weight left in the result's coefficients (the formulas promise 1)
  static int fp12_op(int op, const void* a, const void* b, void* out) {
    Fp12<C> x, y, r;
    memcpy(&x, a, sizeof(x));
    if (b) memcpy(&y, b, sizeof(y));
    F12h hx, hy, hr;
    to_h12(hx, x);
    if (b && op != 12) to_h12(hy, y);
    switch (op) {
      case 0: fp12_mul<C>(hr, hx, hy); break;
      case 1: fp12_sqr<C>(hr, hx); break;
      case 2: fp12_inv<C>(hr, hx); break;
      case 3: fp12_frob<C, 1>(hr, hx); break;
      case 4: fp12_frob<C, 2>(hr, hx); break;
      case 5: fp12_frob<C, 3>(hr, hx); break;
      case 6: fp12_cyclo_sqr<C>(hr, hx); break;
      case 7: fp12_conj<C>(hr, hx); break;
      case 8: fp12_expt<C>(hr, hx); break;
      case 9: final_exp<C>(hr, hx); break;
      case 10: hr = hx; fp12_mul<C>(hr, hr, hy); break;  // in place
      case 11: hr = hx; fp12_sqr<C>(hr, hr); break;
      case 12: {  // one compressed cyclotomic squaring, decompressed with its own inversion
        CycloComp<C, E> k;
        k.b0 = hx.c1.c0;
        k.b1 = hx.c0.c2;
        k.d0 = hx.c0.c1;
        k.d1 = hx.c1.c2;
        cyclo_sqr_compressed<C>(k);
        if (b) for (int i = 1; i < ((const uint8_t*)b)[0]; i++) cyclo_sqr_compressed<C>(k);
        E num, den, inv, a1;
        cyclo_a1_fraction<C>(num, den, k);
        fp2_inv<C>(inv, den);
        fp2_mul<C>(a1, num, inv);
        cyclo_decompress<C>(hr, k, a1);
        break;
      }
      default: return -1;
    }
    from_h12(r, hr);
    memcpy(out, &r, sizeof(r));
    return max_w(hr);
  }
  static int pairing(const void* g1s, const void* g2s, int n_pairs, int with_fexp, void* out) {
    typedef Affine<FpField<C>> A1;
    typedef Affine<Fp2Field<C>> A2;
    const A1* P = (const A1*)g1s;
    const A2* Q = (const A2*)g2s;
    Fp28<C> px[4], py[4];
    E qx[4], qy[4];
    bool live[4];
    for (int k = 0; k < n_pairs && k < 4; k++) {
      live[k] = !(affine_is_inf<FpField<C>>(P[k]) | affine_is_inf<Fp2Field<C>>(Q[k]));
      fp28_from_fp<C>(px[k], P[k].x);
      fp28_from_fp<C>(py[k], P[k].y);
      to_h(qx[k], Q[k].x);
      to_h(qy[k], Q[k].y);
    }
    F12h f, r;
    miller_loop_core<C, 4, E, Fp28<C>>(f, px, py, qx, qy, live, n_pairs);
    if (with_fexp) {
      final_exp<C>(r, f);
      f = r;
    }
    Fp12<C> o;
    from_h12(o, f);
    memcpy(out, &o, sizeof(o));
    return max_w(f);
  }
};

typedef Lp28T<Bls381> Lp28;

// the quad-lane pairing (pairing_quad.h) through its host model Fp2Q28H: pair A / pair B of a quad (BLS12-381, BLS12-377)
template <class CC>
struct Q28T {
  typedef CC C;
  typedef Fp2Q28H<C> E;
  typedef Fp12Q<C, E> F12q;
  static void to_q(F12q& r, const Fp12<C>& a) {
    const Fp2<C>* lo = &a.c0.c0;
    const Fp2<C>* up = &a.c1.c0;
    E* d = &r.v.c0;
    for (int j = 0; j < 3; j++) {
      fp28_from_fp<C>(d[j].c[0], lo[j].c0);
      fp28_from_fp<C>(d[j].c[1], lo[j].c1);
      fp28_from_fp<C>(d[j].c[2], up[j].c0);
      fp28_from_fp<C>(d[j].c[3], up[j].c1);
      d[j].wt = 1;
      d[j].vbound = 1;
    }
  }
  static void from_q(Fp12<C>& r, const F12q& a) {
    Fp2<C>* lo = &r.c0.c0;
    Fp2<C>* up = &r.c1.c0;
    const E* s = &a.v.c0;
    for (int j = 0; j < 3; j++) {
      fp28_to_fp<C>(lo[j].c0, s[j].c[0]);
      fp28_to_fp<C>(lo[j].c1, s[j].c[1]);
      fp28_to_fp<C>(up[j].c0, s[j].c[2]);
      fp28_to_fp<C>(up[j].c1, s[j].c[3]);
    }
  }
  static void rep(E& r, const Fp2<C>& a) {  // an Fp2 value replicated on both pairs
    fp28_from_fp<C>(r.c[0], a.c0);
    fp28_from_fp<C>(r.c[1], a.c1);
    r.c[2] = r.c[0];
    r.c[3] = r.c[1];
    r.wt = 1;
    r.vbound = 1;
  }
  static int max_w(const F12q& a) {
    const E* s = &a.v.c0;
    int w = 0;
    for (int j = 0; j < 3; j++) w = s[j].wt > w ? s[j].wt : w;
    return w;
  }
  static int fp12_op(int op, const void* a, const void* b, void* out) {
    Fp12<C> x, y, r;
    memcpy(&x, a, sizeof(x));
    if (b && op != 12 && op != 13) memcpy(&y, b, sizeof(y));
    F12q hx, hy, hr;
    to_q(hx, x);
    if (b && op != 12 && op != 13) to_q(hy, y);
    switch (op) {
      case 0: fp12q_mul<C>(hr, hx, hy); break;
      case 1: fp12q_sqr<C>(hr, hx); break;
      case 2: fp12q_inv<C>(hr, hx); break;
      case 3: fp12q_frob<C, 1>(hr, hx); break;
      case 4: fp12q_frob<C, 2>(hr, hx); break;
      case 5: fp12q_frob<C, 3>(hr, hx); break;
      case 7: fp12q_conj<C>(hr, hx); break;
      case 8: fp12q_expt<C>(hr, hx); break;
      case 9: final_exp_q<C>(hr, hx); break;
      case 10: hr = hx; fp12q_mul<C>(hr, hr, hy); break;  // in place
      case 11: hr = hx; fp12q_sqr<C>(hr, hr); break;
      case 12: {  // compressed cyclotomic squarings on the quad, decompressed (replicated) with their own inversion
        CycloCompQ<C, E> k;
        cyclo_compress_q<C>(k, hx);
        const int reps = b ? ((const uint8_t*)b)[0] : 1;
        for (int i = 0; i < (reps < 1 ? 1 : reps); i++) cyclo_sqr_compressed_q<C>(k);
        CycloComp<C, E> kc;
        cyclo_replicate_q<C>(kc, k);
        E num, den, inv, a1;
        cyclo_a1_fraction<C>(num, den, kc);
        fp2_inv<C>(inv, den);
        fp2_mul<C>(a1, num, inv);
        Fp12<C, E> v;
        cyclo_decompress<C>(v, kc, a1);
        fp12q_from_replicated<C>(hr, v);
        break;
      }
      case 13: {  // f *= line (c0, c1, c4) [M-twist] / (c0, c3, c4) [D-twist]: b holds the three Fp2 coefficients
        const Fp2<C>* l = (const Fp2<C>*)b;
        E c0, c1, c4;
        rep(c0, l[0]);
        rep(c1, l[1]);
        rep(c4, l[2]);
        hr = hx;
        if constexpr (C::MTWIST)
          fp12q_mul_by_014<C>(hr, c0, c1, c4);
        else
          fp12q_mul_by_034<C>(hr, c0, c1, c4);
        break;
      }
      default: return -1;
    }
    from_q(r, hr);
    memcpy(out, &r, sizeof(r));
    return max_w(hr);
  }
  static int pairing(const void* g1, const void* g2, int n_pairs, int with_fexp, void* out) {
    typedef Affine<FpField<C>> A1;
    typedef Affine<Fp2Field<C>> A2;
    const A1* P = (const A1*)g1;
    const A2* Q = (const A2*)g2;
    Fp28<C> px[4], py[4];
    E qx[4], qy[4];
    bool live[4];
    for (int k = 0; k < n_pairs && k < 4; k++) {
      live[k] = !(affine_is_inf<FpField<C>>(P[k]) | affine_is_inf<Fp2Field<C>>(Q[k]));
      fp28_from_fp<C>(px[k], P[k].x);
      fp28_from_fp<C>(py[k], P[k].y);
      rep(qx[k], Q[k].x);
      rep(qy[k], Q[k].y);
    }
    F12q f, r;
    miller_loop_q<C, 4, E, Fp28<C>>(f, px, py, qx, qy, live, n_pairs);
    if (with_fexp) {
      final_exp_q<C>(r, f);
      f = r;
    }
    Fp12<C> o;
    from_q(o, f);
    memcpy(out, &o, sizeof(o));
    return max_w(f);
  }
};
typedef Q28T<Bls381> Q28;

#define DISPATCH(curve, call)                 \
  switch (curve) {                            \
    case 0: return Ops<Bn254>::call;          \
    case 1: return Ops<Bls381>::call;         \
    case 2: return Ops<Bls377>::call;         \
    default: return -2;                       \
  }

extern "C" {
int hm_fp_op(int curve, int op, const void* a, const void* b, void* out) { DISPATCH(curve, fp_op(op, a, b, out)) }
int hm_fp2_op(int curve, int op, const void* a, const void* b, void* out) { DISPATCH(curve, fp2_op(op, a, b, out)) }
int hm_fp12_op(int curve, int op, const void* a, const void* b, void* out) { DISPATCH(curve, fp12_op(op, a, b, out)) }
int hm_g1_sum(int curve, const void* pts, const uint8_t* neg, int n, void* out) { DISPATCH(curve, g1_sum(pts, neg, n, out)) }
int hm_g2_sum(int curve, const void* pts, const uint8_t* neg, int n, void* out) { DISPATCH(curve, g2_sum(pts, neg, n, out)) }
int hm_g1_tree(int curve, const void* pts, int n, void* out) { DISPATCH(curve, g1_tree(pts, n, out)) }
int hm_g2_tree(int curve, const void* pts, int n, void* out) { DISPATCH(curve, g2_tree(pts, n, out)) }
int hm_digits(int curve, const void* scalar, int mont, int c, uint32_t* out, int cap) { DISPATCH(curve, digits(scalar, mont, c, out, cap)) }
int hm_horner(int curve, int group, const void* pts, int W, const int* down, int which, void* out) { DISPATCH(curve, horner(group, pts, W, down, which, out)) }
int hm_fold_msm(int curve, int group, const void* pts, const void* scalars, int n, int c, int lgM, int lgL, void* out) {
  DISPATCH(curve, fold_msm(group, pts, scalars, n, c, lgM, lgL, out))
}
int hm_chunks(int curve, const void* pts, int n_chunks, void* outA, void* outW0) { DISPATCH(curve, chunks(pts, n_chunks, outA, outW0)) }
int hm_g1_decode(int curve, const uint8_t* w, int compressed, int subgroup, void* out) { DISPATCH(curve, g1dec(w, compressed, subgroup, out)) }
int hm_g1_encode(int curve, const void* pt, int compressed, uint8_t* w) { DISPATCH(curve, g1enc(pt, compressed, w)) }
int hm_fp28_op(int curve, int op, const void* a, const void* b, const void* c, const void* d, void* out) { DISPATCH(curve, fp28_op(op, a, b, c, d, out)) }
int hm_madd28_chain(int curve, const void* pts, const uint8_t* neg, int n, void* out) { DISPATCH(curve, madd28_chain(pts, neg, n, out)) }
int hm_quad_chain(int curve, const void* pts, const void* zs, int n, void* out) { DISPATCH(curve, quad_chain(pts, zs, n, out)) }
int hm_quad28_chain(int curve, const void* pts, const void* zs, int n, void* out) { DISPATCH(curve, quad28_chain(pts, zs, n, out)) }
int hm_add28_lp_chain(int curve, const void* pts, const void* zs, int n, void* out) { DISPATCH(curve, add28_lp_chain(pts, zs, n, out)) }
int hm_madd28_lp_chain(int curve, const void* pts, const uint8_t* neg, int n, void* out) { DISPATCH(curve, madd28_lp_chain(pts, neg, n, out)) }
int hm_ed28_chain(int curve, const void* pts, const uint8_t* neg, int n, int mode, void* out) { DISPATCH(curve, ed28_chain(pts, neg, n, mode, out)) }
int hm_madd28_kc_chain(int curve, const void* pts, const uint8_t* neg, int n, void* out) { DISPATCH(curve, madd28_kc_chain(pts, neg, n, out)) }
int hm_g2_decode(int curve, const uint8_t* w, int compressed, int subgroup, void* out) { DISPATCH(curve, g2dec(w, compressed, subgroup, out)) }
int hm_g2_encode(int curve, const void* pt, int compressed, uint8_t* w) { DISPATCH(curve, g2enc(pt, compressed, w)) }
int hm_lp28_fp12_op(int op, const void* a, const void* b, void* out) { return Lp28::fp12_op(op, a, b, out); }
int hm_lp28_pairing(const void* g1s, const void* g2s, int n_pairs, int with_fexp, void* out) { return Lp28::pairing(g1s, g2s, n_pairs, with_fexp, out); }
// the same for a curve id: BLS12-381 (1) or BLS12-377 (2: u^2 = -5, xi = u, D-twist)
int hm_lp28c_fp12_op(int curve, int op, const void* a, const void* b, void* out) {
  if (curve == 1) return Lp28T<Bls381>::fp12_op(op, a, b, out);
  if (curve == 2) return Lp28T<Bls377>::fp12_op(op, a, b, out);
  if (curve == 0) return Lp28T<Bn254>::fp12_op(op, a, b, out);
  return -2;
}
int hm_lp28c_pairing(int curve, const void* g1s, const void* g2s, int n_pairs, int with_fexp, void* out) {
  if (curve == 1) return Lp28T<Bls381>::pairing(g1s, g2s, n_pairs, with_fexp, out);
  if (curve == 2) return Lp28T<Bls377>::pairing(g1s, g2s, n_pairs, with_fexp, out);
  if (curve == 0) return Lp28T<Bn254>::pairing(g1s, g2s, n_pairs, with_fexp, out);
  return -2;
}
int hm_q28_fp12_op(int op, const void* a, const void* b, void* out) { return Q28::fp12_op(op, a, b, out); }
int hm_q28_pairing(const void* g1, const void* g2, int n_pairs, int with_fexp, void* out) { return Q28::pairing(g1, g2, n_pairs, with_fexp, out); }
// the same for a curve id: BLS12-381 (1) or BLS12-377 (2: D-twist line, u^2 = -5, xi = u)
int hm_q28c_fp12_op(int curve, int op, const void* a, const void* b, void* out) {
  if (curve == 1) return Q28T<Bls381>::fp12_op(op, a, b, out);
  if (curve == 2) return Q28T<Bls377>::fp12_op(op, a, b, out);
  if (curve == 0) return Q28T<Bn254>::fp12_op(op, a, b, out);
  return -2;
}
int hm_q28c_pairing(int curve, const void* g1, const void* g2, int n_pairs, int with_fexp, void* out) {
  if (curve == 1) return Q28T<Bls381>::pairing(g1, g2, n_pairs, with_fexp, out);
  if (curve == 2) return Q28T<Bls377>::pairing(g1, g2, n_pairs, with_fexp, out);
  if (curve == 0) return Q28T<Bn254>::pairing(g1, g2, n_pairs, with_fexp, out);
  return -2;
}
int hm_fp28_reduce(int curve, const int32_t* in, int32_t* out) {
  if (curve != 1) return -2;
  Fp28<Bls381> a, r;
  memcpy(&a, in, sizeof(a));
  fp28_reduce<Bls381>(r, a);
  memcpy(out, &r, sizeof(r));
  return 0;
}
int hm_miller(int curve, const void* g1s, const void* g2s, int n_pairs, void* out) { DISPATCH(curve, miller(g1s, g2s, n_pairs, out)) }
}
