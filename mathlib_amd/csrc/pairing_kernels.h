// pairing_kernels.h -- batched pairing kernels (one pairing per lane; Fp12 state lives in scratch).
// Included by tu_pairing_<curve>.hip.  Replaces MillerLoop / FinalExponentiation behind the reference's
// Pairing, Pairing2 and FExp (driver/gurvy/bls12381/bls12-381.go:448-468, bn254.go:247-267, bls12-377.go:244-264).
#pragma once
#include <cstdlib>

#include "mlhip_internal.h"
#include "msm_body.h"
#include "fp2_lanes.h"
#include "fp2_lanes28.h"
#include "pairing_quad.h"
#include "pairing.h"

namespace mlhip {

template <class C>
__global__ void __launch_bounds__(64) k_miller(const Affine<FpField<C>>* __restrict__ g1,
                                               const Affine<Fp2Field<C>>* __restrict__ g2, int ppp, size_t n,
                                               Fp12<C>* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fp12<C> f;
  miller_loop<C, 4>(f, g1 + i * ppp, g2 + i * ppp, ppp);
  out[i] = f;
}

template <class C>
__global__ void __launch_bounds__(64) k_final_exp(const Fp12<C>* __restrict__ in, size_t n, Fp12<C>* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fp12<C> f = in[i], r;
  final_exp<C>(r, f);
  out[i] = r;
}

template <class C>
__global__ void __launch_bounds__(64) k_pairing(const Affine<FpField<C>>* __restrict__ g1,
                                                const Affine<Fp2Field<C>>* __restrict__ g2, size_t n,
                                                Fp12<C>* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fp12<C> f, r;
  miller_loop<C, 1>(f, g1 + i, g2 + i, 1);
  final_exp<C>(r, f);
  out[i] = r;
}

// ---- lane-pair kernels: two adjacent lanes per pairing, one Fp2 component each (fp2_lanes.h) ----------
template <class C>
MLHIP_HD void fp2_halve(Fp2L<C>& r, const Fp2L<C>& a) {
  fp_halve<C>(r.v, a.v);
}

template <class C>
__device__ __forceinline__ void lp_store_gt(Fp12<C>* out, size_t i, const Fp12<C, Fp2L<C>>& f) {
  // Fp12 memory order: c0.c0 c0.c1 c0.c2 c1.c0 c1.c1 c1.c2, each {c0, c1}; this lane owns component hi
  Fp<C>* o = reinterpret_cast<Fp<C>*>(out + i) + (lane_is_hi() ? 1 : 0);
  o[0] = f.c0.c0.v;
  o[2] = f.c0.c1.v;
  o[4] = f.c0.c2.v;
  o[6] = f.c1.c0.v;
  o[8] = f.c1.c1.v;
  o[10] = f.c1.c2.v;
}
template <class C>
__device__ __forceinline__ void lp_load_gt(Fp12<C, Fp2L<C>>& f, const Fp12<C>* in, size_t i) {
  const Fp<C>* o = reinterpret_cast<const Fp<C>*>(in + i) + (lane_is_hi() ? 1 : 0);
  f.c0.c0.v = o[0];
  f.c0.c1.v = o[2];
  f.c0.c2.v = o[4];
  f.c1.c0.v = o[6];
  f.c1.c1.v = o[8];
  f.c1.c2.v = o[10];
}

// what: 0 = Miller loop of ppp pairs per product, 1 = final exponentiation, 2 = Miller (1 pair) + final exp
// Two waves per SIMD: left alone, the register allocator takes ~36 AGPRs on top of the 256 VGPRs, which halves the
// occupancy to one wave per SIMD -- and a lone wave can issue at most every other VALU slot on gfx950
// (profiles/r01_ubench_int.txt, wps=1 vs wps=2).  The cap costs no extra scratch.
#ifndef MLHIP_LP_OCC
#define MLHIP_LP_OCC __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
// MAXP: pairs per product the instance can hold (1 for the plain pairing: the per-pair arrays then live in
// registers instead of dynamically indexed scratch)
template <class C, int WHAT, int MAXP>
__global__ void __launch_bounds__(64) MLHIP_LP_OCC k_pairing_lp(const Affine<FpField<C>>* __restrict__ g1,
                                                   const Affine<Fp2Field<C>>* __restrict__ g2, int ppp, size_t n,
                                                   const Fp12<C>* __restrict__ in, Fp12<C>* __restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = t >> 1;  // both lanes of a pair share i, so this exit is pair-uniform
  if (i >= n) return;
  typedef Fp2L<C> E2;
  // the Miller accumulator / final-exponentiation input lives in LDS (one padded slot per lane, 18 KB per block:
  // eight blocks fill a CU's 160 KB) instead of scratch -- the most re-read Fp12 of the kernel; the out-of-line
  // tower functions reach it through flat addresses.  Miller loop -2.6 %.
  constexpr int SLOT = sizeof(Fp12<C, E2>) / 4 + 1;  // odd word stride: conflict-free
  __shared__ uint32_t f_slots[64 * SLOT];
  Fp12<C, E2>& f = *reinterpret_cast<Fp12<C, E2>*>(&f_slots[threadIdx.x * SLOT]);
  Fp12<C, E2> r;
  if (WHAT == 1) {
    lp_load_gt<C>(f, in, i);
  } else {
    Fp<C> px[MAXP], py[MAXP];
    E2 qx[MAXP], qy[MAXP];
    bool live[MAXP];
    const int hi = lane_is_hi() ? 1 : 0;
    for (int k = 0; k < ppp && k < MAXP; k++) {
      const Affine<FpField<C>> P = g1[i * ppp + k];
      const Fp<C>* q = reinterpret_cast<const Fp<C>*>(g2 + i * ppp + k);
      px[k] = P.x;
      py[k] = P.y;
      qx[k].v = q[hi];
      qy[k].v = q[2 + hi];
      // Q is the point at infinity iff all four of its Fp components are zero
      uint32_t zq = (fp_is_zero<C>(qx[k].v) & fp_is_zero<C>(qy[k].v)) ? 1u : 0u;
      zq &= pair_xchg_u32(zq);
      live[k] = !(affine_is_inf<FpField<C>>(P) | (zq != 0));
    }
    miller_loop_core<C, MAXP, E2, Fp<C>>(f, px, py, qx, qy, live, ppp);
  }
  if (WHAT == 0) {
    lp_store_gt<C>(out, i, f);
  } else {
    final_exp<C>(r, f);
    lp_store_gt<C>(out, i, r);
  }
}

// ---- the same over lane pairs in the CARRY-FREE form (fp2_lanes28.h; BLS12-381): inputs are converted on entry (one
// product per coordinate), the Fp12 result is brought back to the boundary form on exit (one product per coefficient +
// the canonical representative), everything between runs on 28-bit limbs with v_mad_i64_i32 only.
template <class C>
__device__ __forceinline__ void lp28_store_gt(Fp12<C>* out, size_t i, const Fp12<C, Fp2L28<C>>& f) {
  Fp<C>* o = reinterpret_cast<Fp<C>*>(out + i) + (lane_is_hi() ? 1 : 0);
  const Fp2L28<C>* c = &f.c0.c0;
#pragma unroll 1
  for (int k = 0; k < 6; k++) {
    Fp<C> t;
    fp28_to_fp<C>(t, c[k].v);
    o[2 * k] = t;
  }
}
template <class C>
__device__ __forceinline__ void lp28_load_gt(Fp12<C, Fp2L28<C>>& f, const Fp12<C>* in, size_t i) {
  const Fp<C>* o = reinterpret_cast<const Fp<C>*>(in + i) + (lane_is_hi() ? 1 : 0);
  Fp2L28<C>* c = &f.c0.c0;
#pragma unroll 1
  for (int k = 0; k < 6; k++) fp28_from_fp<C>(c[k].v, o[2 * k]);
}

// ---- the Miller loop of ONE pair with the point T and P's coordinates in LDS (round 3) ------------------------------------
// In miller_loop_core T, P and the line are locals of the kernel that must survive two out-of-line Fp12 calls per
// iteration: the register allocator spills them (about 110 words per lane stored and reloaded through scratch every
// iteration, beside the 84-word f the callees exchange), and with two waves per SIMD those round trips are exposed
// (profiles/r03_pmc_waits.txt: the lane-pair Miller loop runs at 80 % of the rate its instruction mix allows, the
// inlined quad loop at 97 %).  Here T (three Fp2 = 42 words per lane) and px, py (28 words) live in a 71-word LDS slot per
// lane (odd stride: conflict-free; 18 KB per wave, eight waves fill the CU), are read where a step starts and written
// where it ends -- nothing but f's address is live across the calls.  Same formulas, same order: bit-identical results.
template <class C>
struct MillerLds {
  static constexpr int STRIDE = 5 * C::N28 + 1;
  int32_t* slot;
  __device__ __forceinline__ void put(int k, const Fp28<C>& v) const {
#pragma unroll
    for (int i = 0; i < C::N28; i++) slot[k * C::N28 + i] = v.l[i];
  }
  __device__ __forceinline__ void get(Fp28<C>& v, int k) const {
#pragma unroll
    for (int i = 0; i < C::N28; i++) v.l[i] = slot[k * C::N28 + i];
  }
};
template <class C>
__device__ __forceinline__ void miller_loop_lp28_lds(Fp12<C, Fp2L28<C>>& f, const Fp28<C>& px, const Fp28<C>& py,
                                                     const Fp2L28<C>& qx, const Fp2L28<C>& qy, bool live,
                                                     const MillerLds<C>& lds) {
  // (M-twist line: BLS12-381; D-twist line: BLS12-377, BN254 -- the latter with its two Frobenius lines after the loop)
  typedef Fp2L28<C> E2;
  fp12_one<C>(f);
  if (!live) return;
  {
    E2 one;
    fp2_one<C>(one);
    lds.put(0, qx.v);
    lds.put(1, qy.v);
    lds.put(2, one.v);
    lds.put(3, px);
    lds.put(4, py);
  }
  bool first = true;
#pragma unroll 1
  for (int i = C::ATE_BITS - 2; i >= 0; i--) {
    const bool bit = (i >= 64) ? ((C::ATE_HI >> (i - 64)) & 1) : ((C::ATE_LO >> i) & 1);
    {
      // the doubling step needs T only: the line is ready before f is touched, and f^2 * line is one call
      // (fp12_sqr_mul_by_014: f crosses memory once per iteration)
      G2Proj<C, E2> T;
      Line<C, E2> l;
      lds.get(T.x.v, 0);
      lds.get(T.y.v, 1);
      lds.get(T.z.v, 2);
      g2_double_step<C>(T, l);
      lds.put(0, T.x.v);
      lds.put(1, T.y.v);
      lds.put(2, T.z.v);
      Fp28<C> x, y;
      lds.get(x, 3);
      lds.get(y, 4);
      // (inlining the two bodies into the loop instead -- f a local the compiler spills piecewise -- is slower: 19.1 ms
      // against 17.8 for the fused batch, profiles/r03_pairing_ab.txt)
      E2 a, b, c = l.r2;
      fp2_mul_fp<C>(a, l.r0, y);
      fp2_mul_fp<C>(b, l.r1, x);
      fp2_norm<C>(c);
      if constexpr (C::MTWIST) {
        if (first)
          fp12_mul_by_014<C>(f, c, b, a);
        else
          fp12_sqr_mul_by_014<C>(f, c, b, a);
      } else {
        if (first)
          fp12_mul_by_034<C>(f, a, b, c);
        else
          fp12_sqr_mul_by_034<C>(f, a, b, c);
      }
      first = false;
    }
    if (bit) {
      G2Proj<C, E2> T;
      Line<C, E2> l;
      lds.get(T.x.v, 0);
      lds.get(T.y.v, 1);
      lds.get(T.z.v, 2);
      g2_add_step<C>(T, qx, qy, l);
      lds.put(0, T.x.v);
      lds.put(1, T.y.v);
      lds.put(2, T.z.v);
      Fp28<C> x, y;
      lds.get(x, 3);
      lds.get(y, 4);
      mul_by_line<C>(f, l, x, y);
    }
  }
  if constexpr (C::IS_BN) {
    // lines through pi(Q) and -pi^2(Q) (miller_loop_core's tail)
    E2 x1, y1, x2, y2, g;
    fp2_conj<C>(x1, qx);
    fp2_from_const<C>(g, C::GAMMA1[2]);
    fp2_mul<C>(x1, x1, g);
    fp2_conj<C>(y1, qy);
    fp2_from_const<C>(g, C::GAMMA1[3]);
    fp2_mul<C>(y1, y1, g);
    fp2_mul_by_real_const<C>(x2, qx, C::GAMMA2[2]);
    fp2_mul_by_real_const<C>(y2, qy, C::GAMMA2[3]);
    fp2_neg<C>(y2, y2);
    G2Proj<C, E2> T;
    Line<C, E2> l;
    lds.get(T.x.v, 0);
    lds.get(T.y.v, 1);
    lds.get(T.z.v, 2);
    Fp28<C> x, y;
    lds.get(x, 3);
    lds.get(y, 4);
    g2_add_step<C>(T, x1, y1, l);
    mul_by_line<C>(f, l, x, y);
    g2_add_step<C>(T, x2, y2, l);
    mul_by_line<C>(f, l, x, y);
  }
  if (C::X_NEG) fp12_conj<C>(f, f);
}

template <class C, int WHAT, int MAXP>
__global__ void __launch_bounds__(64) MLHIP_LP_OCC k_pairing_lp28(const Affine<FpField<C>>* __restrict__ g1,
                                                                  const Affine<Fp2Field<C>>* __restrict__ g2, int ppp,
                                                                  size_t n, const Fp12<C>* __restrict__ in,
                                                                  Fp12<C>* __restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = t >> 1;  // both lanes of a pair share i, so this exit is pair-uniform
  if (i >= n) return;
  typedef Fp2L28<C> E2;
  Fp12<C, E2> f, r;
  if (WHAT == 1) {
    lp28_load_gt<C>(f, in, i);
  } else {
    Fp28<C> px[MAXP], py[MAXP];
    E2 qx[MAXP], qy[MAXP];
    bool live[MAXP];
    const int hi = lane_is_hi() ? 1 : 0;
    for (int k = 0; k < ppp && k < MAXP; k++) {
      const Affine<FpField<C>> P = g1[i * ppp + k];
      const Fp<C>* q = reinterpret_cast<const Fp<C>*>(g2 + i * ppp + k);
      const Fp<C> qxc = q[hi], qyc = q[2 + hi];
      // Q is the point at infinity iff all four of its Fp components are zero (tested on the boundary form)
      uint32_t zq = (fp_is_zero<C>(qxc) & fp_is_zero<C>(qyc)) ? 1u : 0u;
      zq &= pair_xchg_u32(zq);
      live[k] = !(affine_is_inf<FpField<C>>(P) | (zq != 0));
      fp28_from_fp<C>(px[k], P.x);
      fp28_from_fp<C>(py[k], P.y);
      fp28_from_fp<C>(qx[k].v, qxc);
      fp28_from_fp<C>(qy[k].v, qyc);
    }
    if constexpr (MAXP == 1) {
      __shared__ int32_t t_slots[64 * MillerLds<C>::STRIDE];
      const MillerLds<C> lds{t_slots + threadIdx.x * MillerLds<C>::STRIDE};
      miller_loop_lp28_lds<C>(f, px[0], py[0], qx[0], qy[0], live[0], lds);
    } else {
      miller_loop_core<C, MAXP, E2, Fp28<C>>(f, px, py, qx, qy, live, ppp);
    }
  }
  if (WHAT == 0) {
    lp28_store_gt<C>(out, i, f);
  } else {
    final_exp<C>(r, f);
    lp28_store_gt<C>(out, i, r);
  }
}

// Fp12 in memory (c0.{c0,c1,c2}.{c0,c1}, then c1) <-> the quad form: this lane moves coefficient j of half `pair B ? c1 : c0`,
// component `hi ? imaginary : real`
template <class C>
__device__ __forceinline__ void q28_load_gt(Fp12Q<C, Fp2L28<C>>& f, const Fp12<C>* in, size_t i) {
  const Fp<C>* o = reinterpret_cast<const Fp<C>*>(in + i) + ((threadIdx.x & 2u) ? 6 : 0) + (lane_is_hi() ? 1 : 0);
  fp28_from_fp<C>(f.v.c0.v, o[0]);
  fp28_from_fp<C>(f.v.c1.v, o[2]);
  fp28_from_fp<C>(f.v.c2.v, o[4]);
}
template <class C>
__device__ __forceinline__ void q28_store_gt(Fp12<C>* out, size_t i, const Fp12Q<C, Fp2L28<C>>& f) {
  Fp<C>* o = reinterpret_cast<Fp<C>*>(out + i) + ((threadIdx.x & 2u) ? 6 : 0) + (lane_is_hi() ? 1 : 0);
  Fp<C> w;
  fp28_to_fp<C>(w, f.v.c0.v);
  o[0] = w;
  fp28_to_fp<C>(w, f.v.c1.v);
  o[2] = w;
  fp28_to_fp<C>(w, f.v.c2.v);
  o[4] = w;
}

// ---- one pairing per QUAD of lanes (pairing_quad.h; BLS12-381, carry-free element): pair A of a quad carries the c0 half
// and pair B the c1 half of every Fp12 value.  WHAT: 0 = Miller loop of ppp <= MAXP pairs per product, 1 = final
// exponentiation, 2 = Miller loop of one pair + final exponentiation.
template <class C, int WHAT, int MAXP>
__global__ void __launch_bounds__(64) MLHIP_LP_OCC k_pairing_q28(const Affine<FpField<C>>* __restrict__ g1,
                                                                 const Affine<Fp2Field<C>>* __restrict__ g2, int ppp, size_t n,
                                                                 const Fp12<C>* __restrict__ in, Fp12<C>* __restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = t >> 2;  // the four lanes of a quad share i: quad-uniform exit
  if (i >= n) return;
  typedef Fp2L28<C> E2;
  const int hi = lane_is_hi() ? 1 : 0;  // which Fp2 component this lane holds
  Fp12Q<C, E2> f, r;
  if (WHAT == 1) {
    q28_load_gt<C>(f, in, i);
  } else {
    Fp28<C> px[MAXP], py[MAXP];
    E2 qx[MAXP], qy[MAXP];
    bool live[MAXP];
    for (int k = 0; k < ppp && k < MAXP; k++) {
      const Affine<FpField<C>> P = g1[i * ppp + k];
      const Fp<C>* q = reinterpret_cast<const Fp<C>*>(g2 + i * ppp + k);
      const Fp<C> qxc = q[hi], qyc = q[2 + hi];
      uint32_t zq = (fp_is_zero<C>(qxc) & fp_is_zero<C>(qyc)) ? 1u : 0u;
      zq &= pair_xchg_u32(zq);  // Q at infinity: all four Fp components zero (both pairs hold the same Q)
      live[k] = !(affine_is_inf<FpField<C>>(P) | (zq != 0));
      fp28_from_fp<C>(px[k], P.x);
      fp28_from_fp<C>(py[k], P.y);
      fp28_from_fp<C>(qx[k].v, qxc);
      fp28_from_fp<C>(qy[k].v, qyc);
    }
    miller_loop_q<C, MAXP, E2, Fp28<C>>(f, px, py, qx, qy, live, ppp);
  }
  if (WHAT != 0) {
    final_exp_q<C>(r, f);
    f = r;
  }
  q28_store_gt<C>(out, i, f);
}

template <class C>
__global__ void __launch_bounds__(256) k_fp_mul(const Fp<C>* __restrict__ a, const Fp<C>* __restrict__ b, size_t n,
                                                int repeat, Fp<C>* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fp<C> x = a[i], y = b[i], r;
  fp_mul<C>(r, x, y);
  for (int k = 1; k < repeat; k++) fp_mul<C>(r, r, y);
  out[i] = r;
}

// carry-free lane pairs on every curve (BLS12-377's u^2 = -5: fp2_lanes28.h carry-propagates every product operand first;
// BN254: 10 limbs, xi = 9 + u) unless MLHIP_PAIRING_SAT=1 (read per batch so a test can switch paths)
template <class C>
bool lp28_enabled() {
  (void)C::ID;  // all three curves
  return !mlhip_alt_switch("MLHIP_PAIRING_SAT");  // (the saturated lane-pair kernels: test build only, except BN254's Miller loop)
}

// The batched entry points run the lane-pair kernels (two lanes per pairing); the one-lane-per-pairing
// kernels above stay as the reference shape (MLHIP_PAIRING_ONE_LANE=1 selects them; the tests run both).
template <class C>
int pairing_device(int what, const void* d_g1, const void* d_g2, size_t ppp, size_t n, const void* d_in, void* d_out,
                   hipStream_t st) {
  if (n == 0) return 0;
  typedef Affine<FpField<C>> A1;
  typedef Affine<Fp2Field<C>> A2;
  const bool one_lane = mlhip_alt_switch("MLHIP_PAIRING_ONE_LANE");  // read per batch so a test can switch paths (test build only)
  // BN254's Miller loop ALONE is a product-bound kernel and stays on the saturated lane pairs (7.3 against 7.45 ms per 65 536)
  // -- except where a quad's shorter chain decides: batches that leave the chip under-filled (round 4)
  bool bn_miller_saturated = C::IS_BN && what == 0;
  if (bn_miller_saturated && ppp <= 4) {
    const char* qe = getenv("MLHIP_PAIRING_QUAD");
    if (qe ? qe[0] == '1' : n <= ((size_t)1 << 14)) bn_miller_saturated = false;
  }
  if (one_lane) {
    if constexpr (kBuildAlt) {
      unsigned blocks = (unsigned)((n + 63) / 64);
      switch (what) {
        case 0:
          k_miller<C><<<dim3(blocks), dim3(64), 0, st>>>((const A1*)d_g1, (const A2*)d_g2, (int)ppp, n, (Fp12<C>*)d_out);
          break;
        case 1:
          k_final_exp<C><<<dim3(blocks), dim3(64), 0, st>>>((const Fp12<C>*)d_in, n, (Fp12<C>*)d_out);
          break;
        default:
          k_pairing<C><<<dim3(blocks), dim3(64), 0, st>>>((const A1*)d_g1, (const A2*)d_g2, n, (Fp12<C>*)d_out);
          break;
      }
    }
  } else if (lp28_enabled<C>() && !bn_miller_saturated) {
    // lane pairs in the carry-free form (MLHIP_PAIRING_SAT=1 selects the saturated lane-pair kernels below; BN254's Miller
    // loop alone stays on them: 7.3 against 7.45 ms per 65 536 -- its 10-limb products gain nothing, the final exponentiation's
    // additions and squarings do: 8.6 -> 7.3 ms, the fused pairing 15.9 -> 14.1 ms)
    unsigned blocks = (unsigned)((2 * n + 63) / 64);
    {  // BLS12-381 and, since round 4, BLS12-377 (D-twist line product, u^2 = -5) and BN254 (Frobenius lines, BN hard part)
      // One pairing per QUAD of lanes (pairing_quad.h) while the batch leaves the chip under-filled: up to 2^14 elements
      // (65 536 lanes = one wave per SIMD) a batch takes the time of ONE pairing's dependent chain, which is 1.5 x
      // shorter on a quad (1 024 pairings: 5.7 ms instead of 8.5; single Pairing 2.6 / FExp 3.1 ms instead of 3.9 / 4.6);
      // a full chip is bound by instruction issue, where the pairs' fewer instructions win (65 536: 18.6 vs 23.4 ms).
      // MLHIP_PAIRING_QUAD=1 / 0 forces / forbids the quads (products of up to 4 pairs; longer ones stay on lane pairs).
      const char* qe = getenv("MLHIP_PAIRING_QUAD");
      // (round 2 also ran the Miller loop of single pairs on quads at every size -- 65 536 loops 9.0 ms against the pairs'
      // 9.2; since round 3 the pairs' loop keeps T and P in LDS and squares and multiplies by the line in one call: 8.6 ms)
      // BLS12-377 (profiles/r04_pairing_quad_bls377.txt): quads win up to 2^15 elements (16 384 pairings 7.4 ms against 13.5,
      // 32 768: 12.9 / 14.9, 65 536: 25.2 / 23.9), its Miller loop alone at every size
      const size_t quad_max = (size_t)1 << (C::ID == 2 ? 15 : 14);
      const bool quads = qe ? qe[0] == '1' : (n <= quad_max || (C::ID == 2 && what == 0));
      if (quads && (what != 0 || ppp <= 4)) {
        const unsigned qblocks = (unsigned)((4 * n + 63) / 64);
        if (what == 0 && ppp == 1)
          k_pairing_q28<C, 0, 1><<<dim3(qblocks), dim3(64), 0, st>>>((const A1*)d_g1, (const A2*)d_g2, 1, n, nullptr, (Fp12<C>*)d_out);
        else if (what == 0)
          k_pairing_q28<C, 0, 4><<<dim3(qblocks), dim3(64), 0, st>>>((const A1*)d_g1, (const A2*)d_g2, (int)ppp, n, nullptr, (Fp12<C>*)d_out);
        else if (what == 1)
          k_pairing_q28<C, 1, 1><<<dim3(qblocks), dim3(64), 0, st>>>(nullptr, nullptr, 1, n, (const Fp12<C>*)d_in, (Fp12<C>*)d_out);
        else
          k_pairing_q28<C, 2, 1><<<dim3(qblocks), dim3(64), 0, st>>>((const A1*)d_g1, (const A2*)d_g2, 1, n, nullptr, (Fp12<C>*)d_out);
        HIPCHK(hipGetLastError());
        return 0;
      }
    }
    {
      switch (what) {
        case 0:
          if (ppp == 1)
            k_pairing_lp28<C, 0, 1><<<dim3(blocks), dim3(64), 0, st>>>((const A1*)d_g1, (const A2*)d_g2, 1, n, nullptr,
                                                                      (Fp12<C>*)d_out);
          else
            k_pairing_lp28<C, 0, 4><<<dim3(blocks), dim3(64), 0, st>>>((const A1*)d_g1, (const A2*)d_g2, (int)ppp, n,
                                                                      nullptr, (Fp12<C>*)d_out);
          break;
        case 1:
          k_pairing_lp28<C, 1, 1><<<dim3(blocks), dim3(64), 0, st>>>(nullptr, nullptr, 1, n, (const Fp12<C>*)d_in,
                                                                    (Fp12<C>*)d_out);
          break;
        default:
          k_pairing_lp28<C, 2, 1><<<dim3(blocks), dim3(64), 0, st>>>((const A1*)d_g1, (const A2*)d_g2, 1, n, nullptr,
                                                                    (Fp12<C>*)d_out);
          break;
      }
    }
  } else {
    // saturated lane pairs: BN254's Miller loop (the default there) and, in the test build, everything else (MLHIP_PAIRING_SAT=1)
    unsigned blocks = (unsigned)((2 * n + 63) / 64);
    switch (what) {
      case 0:
        if constexpr (C::IS_BN || kBuildAlt) {
          if (ppp == 1)
            k_pairing_lp<C, 0, 1><<<dim3(blocks), dim3(64), 0, st>>>((const A1*)d_g1, (const A2*)d_g2, 1, n, nullptr,
                                                                    (Fp12<C>*)d_out);
          else
            k_pairing_lp<C, 0, 4><<<dim3(blocks), dim3(64), 0, st>>>((const A1*)d_g1, (const A2*)d_g2, (int)ppp, n, nullptr,
                                                                    (Fp12<C>*)d_out);
        }
        break;
      case 1:
        if constexpr (kBuildAlt)
          k_pairing_lp<C, 1, 1><<<dim3(blocks), dim3(64), 0, st>>>(nullptr, nullptr, 1, n, (const Fp12<C>*)d_in,
                                                               (Fp12<C>*)d_out);
        break;
      default:
        if constexpr (kBuildAlt)
          k_pairing_lp<C, 2, 1><<<dim3(blocks), dim3(64), 0, st>>>((const A1*)d_g1, (const A2*)d_g2, 1, n, nullptr,
                                                               (Fp12<C>*)d_out);
        break;
    }
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// out[i] = a[i] * b[i] in Fp12 (Gt.Mul, reference driver/gurvy/bls12381/bls12-381.go:417-419)
template <class C>
__global__ void __launch_bounds__(64) k_gt_mul(const Fp12<C>* __restrict__ a, const Fp12<C>* __restrict__ b, size_t n,
                                               Fp12<C>* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fp12<C> x = a[i], y = b[i], r;
  fp12_mul<C>(r, x, y);
  out[i] = r;
}

// out[i] = in[i]^(scalars[i]) in Fp12 (Gt.Exp, reference driver/gurvy/bls12381/bls12-381.go:399-407): plain
// square-and-multiply with the generic Fp12 squaring, so it is valid for ANY Gt value (raw Miller-loop
// outputs included), like gnark's GT.Exp.
template <class C>
__global__ void __launch_bounds__(64) k_gt_exp(const Fp12<C>* __restrict__ in, const uint32_t* __restrict__ scalars, int mont,
                                               size_t n, Fp12<C>* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s[8];
  fr_canonical<C>(s, scalars + 8 * i, mont != 0);
  Fp12<C> base = in[i], acc;
  fp12_one<C>(acc);
  bool started = false;
  for (int b = 255; b >= 0; b--) {
    if (started) fp12_sqr<C>(acc, acc);
    if ((s[b >> 5] >> (b & 31)) & 1u) {
      if (started)
        fp12_mul<C>(acc, acc, base);
      else {
        acc = base;
        started = true;
      }
    }
  }
  out[i] = acc;
}

// the same over lane pairs (two lanes per exponentiation; the running power lives in the LDS slot like the Miller
// accumulator): the batched entry point runs this one, MLHIP_PAIRING_ONE_LANE=1 the kernel above
template <class C>
__global__ void __launch_bounds__(64) MLHIP_LP_OCC k_gt_exp_lp(const Fp12<C>* __restrict__ in,
                                                               const uint32_t* __restrict__ scalars, int mont, size_t n,
                                                               Fp12<C>* __restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = t >> 1;  // pair-uniform exit
  if (i >= n) return;
  typedef Fp2L<C> E2;
  constexpr int SLOT = sizeof(Fp12<C, E2>) / 4 + 1;
  __shared__ uint32_t f_slots[64 * SLOT];
  Fp12<C, E2>& acc = *reinterpret_cast<Fp12<C, E2>*>(&f_slots[threadIdx.x * SLOT]);
  uint32_t s[8];
  fr_canonical<C>(s, scalars + 8 * i, mont != 0);
  // 4-bit fixed windows: base^1 .. base^15 in scratch (4.3 KB per lane, one 288-byte read per window), then 4 squarings
  // and at most one product per window -- 255 squarings + <= 78 products instead of ~128
  Fp12<C, E2> tab[15];
  lp_load_gt<C>(tab[0], in, i);
#pragma unroll 1
  for (int k = 1; k < 15; k++) fp12_mul<C>(tab[k], tab[k - 1], tab[0]);
  fp12_one<C>(acc);
  bool started = false;
#pragma unroll 1
  for (int w = 63; w >= 0; w--) {  // the scalar is the same on both lanes of a pair: every branch is pair-uniform
    if (started) {
#pragma unroll 1
      for (int d = 0; d < 4; d++) fp12_sqr<C>(acc, acc);
    }
    const uint32_t nib = (s[w >> 3] >> ((w & 7) * 4)) & 15u;
    if (nib) {
      if (started)
        fp12_mul<C>(acc, acc, tab[nib - 1]);
      else {
        acc = tab[nib - 1];
        started = true;
      }
    }
  }
  lp_store_gt<C>(out, i, acc);
}

// the same in the carry-free form (BLS12-381): 4-bit windows, the table of powers in scratch as 28-bit-limb values, the
// accumulator too (an 84-word slot per lane would cost the eighth wave of a CU its LDS)
// Gt.Exp with one exponentiation per quad of lanes (pairing_quad.h): the 4-bit windowed chain of k_gt_exp_lp28 with one
// Fp6 product per squaring instead of two and two per multiplication instead of three -- for batches that leave the chip
// under-filled (the chain's depth is what they wait for)
template <class C>
__global__ void __launch_bounds__(64) MLHIP_LP_OCC k_gt_exp_q28(const Fp12<C>* __restrict__ in,
                                                                const uint32_t* __restrict__ scalars, int mont, size_t n,
                                                                Fp12<C>* __restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = t >> 2;  // quad-uniform exit
  if (i >= n) return;
  typedef Fp2L28<C> E2;
  uint32_t s[8];
  fr_canonical<C>(s, scalars + 8 * i, mont != 0);
  Fp12Q<C, E2> tab[15], acc;
  q28_load_gt<C>(tab[0], in, i);
#pragma unroll 1
  for (int k = 1; k < 15; k++) fp12q_mul<C>(tab[k], tab[k - 1], tab[0]);
  fp12q_one<C>(acc);
  bool started = false;
#pragma unroll 1
  for (int w = 63; w >= 0; w--) {  // the scalar is the same on the four lanes of a quad: every branch is quad-uniform
    if (started) {
#pragma unroll 1
      for (int d = 0; d < 4; d++) fp12q_sqr<C>(acc, acc);
    }
    const uint32_t nib = (s[w >> 3] >> ((w & 7) * 4)) & 15u;
    if (nib) {
      if (started)
        fp12q_mul<C>(acc, acc, tab[nib - 1]);
      else {
        acc = tab[nib - 1];
        started = true;
      }
    }
  }
  q28_store_gt<C>(out, i, acc);
}

template <class C>
__global__ void __launch_bounds__(64) MLHIP_LP_OCC k_gt_exp_lp28(const Fp12<C>* __restrict__ in,
                                                                 const uint32_t* __restrict__ scalars, int mont, size_t n,
                                                                 Fp12<C>* __restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = t >> 1;  // pair-uniform exit
  if (i >= n) return;
  typedef Fp2L28<C> E2;
  uint32_t s[8];
  fr_canonical<C>(s, scalars + 8 * i, mont != 0);
  Fp12<C, E2> tab[15], acc;
  lp28_load_gt<C>(tab[0], in, i);
#pragma unroll 1
  for (int k = 1; k < 15; k++) fp12_mul<C>(tab[k], tab[k - 1], tab[0]);
  fp12_one<C>(acc);
  bool started = false;
#pragma unroll 1
  for (int w = 63; w >= 0; w--) {  // the scalar is the same on both lanes of a pair: every branch is pair-uniform
    if (started) {
#pragma unroll 1
      for (int d = 0; d < 4; d++) fp12_sqr<C>(acc, acc);
    }
    const uint32_t nib = (s[w >> 3] >> ((w & 7) * 4)) & 15u;
    if (nib) {
      if (started)
        fp12_mul<C>(acc, acc, tab[nib - 1]);
      else {
        acc = tab[nib - 1];
        started = true;
      }
    }
  }
  lp28_store_gt<C>(out, i, acc);
}

template <class C>
int gt_exp_device(const void* d_in, const void* d_scalars, int mont, size_t n, void* d_out, hipStream_t st) {
  if (mlhip_alt_switch("MLHIP_PAIRING_ONE_LANE")) {
    if constexpr (kBuildAlt)
      k_gt_exp<C><<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st>>>((const Fp12<C>*)d_in, (const uint32_t*)d_scalars, mont, n,
                                                                     (Fp12<C>*)d_out);
  } else if (lp28_enabled<C>()) {
    {
      // quads at every size: the windowed chain is generic squarings and products, where a quad does the lane pair's work
      // in half the rounds without the 84-word operands crossing scratch (65 536: 15.7 ms against 19.2; 1 024: 4.0 / 7.9);
      // MLHIP_PAIRING_QUAD=0 keeps the lane-pair kernel.  BLS12-377 (round 4): profiles/r04_pairing_quad_bls377.txt
      const char* qe = getenv("MLHIP_PAIRING_QUAD");
      if (!(qe && qe[0] == '0'))
        k_gt_exp_q28<C><<<dim3((unsigned)((4 * n + 63) / 64)), dim3(64), 0, st>>>((const Fp12<C>*)d_in, (const uint32_t*)d_scalars,
                                                                               mont, n, (Fp12<C>*)d_out);
      else
        k_gt_exp_lp28<C><<<dim3((unsigned)((2 * n + 63) / 64)), dim3(64), 0, st>>>((const Fp12<C>*)d_in,
                                                                                (const uint32_t*)d_scalars, mont, n, (Fp12<C>*)d_out);
    }
  } else if constexpr (kBuildAlt) {
    k_gt_exp_lp<C><<<dim3((unsigned)((2 * n + 63) / 64)), dim3(64), 0, st>>>((const Fp12<C>*)d_in, (const uint32_t*)d_scalars,
                                                                          mont, n, (Fp12<C>*)d_out);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

template <class C>
int gt_mul_device(const void* d_a, const void* d_b, size_t n, void* d_out, hipStream_t st) {
  k_gt_mul<C><<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st>>>((const Fp12<C>*)d_a, (const Fp12<C>*)d_b, n,
                                                                 (Fp12<C>*)d_out);
  HIPCHK(hipGetLastError());
  return 0;
}

template <class C>
int fp_mul_device(const void* d_a, const void* d_b, size_t n, int repeat, void* d_out, hipStream_t st) {
  k_fp_mul<C><<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>((const Fp<C>*)d_a, (const Fp<C>*)d_b, n, repeat,
                                                                    (Fp<C>*)d_out);
  HIPCHK(hipGetLastError());
  return 0;
}

}  // namespace mlhip
