// fp.h -- prime-field arithmetic for the MI355X kernels (gfx950).
//
// Element = N little-endian 32-bit limbs in Montgomery form, R = 2^(32 N): the same bytes as
// gnark-crypto's fp.Element / kilic's Fe ([6]uint64 resp. [4]uint64, R = 2^384 / 2^256), the
// types the reference's drivers hold (driver/kilic/custom.go:24, driver/gurvy/custom.go:24-40), so
// points and Gt values cross the C ABI with no conversion.
//
// fp_mul replaces the reference's only in-tree field multiply, driver/kilic/custom_generic.go:57-175
// (6x64-bit CIOS).  On CDNA4 the primitive is v_mad_u64_u32 (32x32+64 -> 64, ~4.7 cycles per
// wave64, measured: profiles/r01_ubench_int.txt), so the limbs are 32-bit and the loop below is a
// 32-bit CIOS; the result is fully reduced (< p) like the reference's (custom_generic.go:166-174).
//
// All functions are __host__ __device__: the same source is unit-tested on the CPU against the
// oracle (tests/test_host_math.py) and runs in the kernels.  The host build is a TEST artifact;
// the product path (libmlhip.so) calls these only from device code and from the O(1) host tail.
#pragma once
#include <stdint.h>
#include "curve_constants.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MLHIP_HD __host__ __device__ __forceinline__
#define MLHIP_HD_NOINLINE __host__ __device__ __noinline__
#else
#define MLHIP_HD inline
#define MLHIP_HD_NOINLINE inline
#endif

namespace mlhip {

template <class C>
struct Fp {
  uint32_t l[C::N];
};

template <class C>
MLHIP_HD void fp_zero(Fp<C>& r) {
#pragma unroll
  for (int i = 0; i < C::N; i++) r.l[i] = 0;
}

template <class C>
MLHIP_HD void fp_one(Fp<C>& r) {
#pragma unroll
  for (int i = 0; i < C::N; i++) r.l[i] = C::ONE[i];
}

template <class C>
MLHIP_HD void fp_from_const(Fp<C>& r, const uint32_t (&k)[C::N]) {
#pragma unroll
  for (int i = 0; i < C::N; i++) r.l[i] = k[i];
}

template <class C>
MLHIP_HD bool fp_is_zero(const Fp<C>& a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < C::N; i++) o |= a.l[i];
  return o == 0;
}

template <class C>
MLHIP_HD bool fp_eq(const Fp<C>& a, const Fp<C>& b) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < C::N; i++) o |= a.l[i] ^ b.l[i];
  return o == 0;
}

// Carry-chain primitives.  clang's __builtin_addc / __builtin_subc lower to v_addc_co_u32 / v_subb_co_u32
// chains on gfx950 (hipcc pads their VCC hazards itself); other compilers (the g++ host-test build) get
// the portable 64-bit form.
#if defined(__clang__)
MLHIP_HD uint32_t mlhip_addc(uint32_t a, uint32_t b, uint32_t& c) {
  unsigned co;
  uint32_t r = __builtin_addc(a, b, c, &co);
  c = co;
  return r;
}
MLHIP_HD uint32_t mlhip_subb(uint32_t a, uint32_t b, uint32_t& br) {
  unsigned bo;
  uint32_t r = __builtin_subc(a, b, br, &bo);
  br = bo;
  return r;
}
#else
MLHIP_HD uint32_t mlhip_addc(uint32_t a, uint32_t b, uint32_t& c) {
  uint64_t s = (uint64_t)a + b + c;
  c = (uint32_t)(s >> 32);
  return (uint32_t)s;
}
MLHIP_HD uint32_t mlhip_subb(uint32_t a, uint32_t b, uint32_t& br) {
  uint64_t s = (uint64_t)a - b - br;
  br = (uint32_t)((s >> 32) & 1);
  return (uint32_t)s;
}
#endif

// r = t - p if t >= p else t   (t < 2p)
template <class C>
MLHIP_HD void fp_reduce_once(Fp<C>& r, const uint32_t (&t)[C::N]) {
  uint32_t d[C::N];
  uint32_t br = 0;
#pragma unroll
  for (int i = 0; i < C::N; i++) d[i] = mlhip_subb(t[i], C::P[i], br);
#pragma unroll
  for (int i = 0; i < C::N; i++) r.l[i] = br ? t[i] : d[i];
}

template <class C>
MLHIP_HD void fp_add(Fp<C>& r, const Fp<C>& a, const Fp<C>& b) {
  uint32_t t[C::N];
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < C::N; i++) t[i] = mlhip_addc(a.l[i], b.l[i], c);
  // p < 2^(32N-1) for all three curves, so a + b < 2p never carries out of N limbs
  fp_reduce_once<C>(r, t);
}

template <class C>
MLHIP_HD void fp_dbl(Fp<C>& r, const Fp<C>& a) {
  fp_add<C>(r, a, a);
}

template <class C>
MLHIP_HD void fp_sub(Fp<C>& r, const Fp<C>& a, const Fp<C>& b) {
  uint32_t d[C::N];
  uint32_t br = 0;
#pragma unroll
  for (int i = 0; i < C::N; i++) d[i] = mlhip_subb(a.l[i], b.l[i], br);
  // add p back when the subtraction borrowed
  uint32_t mask = (uint32_t)0 - br;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < C::N; i++) r.l[i] = mlhip_addc(d[i], C::P[i] & mask, c);
}

template <class C>
MLHIP_HD void fp_neg(Fp<C>& r, const Fp<C>& a) {
  // -a mod p, with -0 = 0
  uint32_t nz = 0;
#pragma unroll
  for (int i = 0; i < C::N; i++) nz |= a.l[i];
  uint32_t mask = nz ? 0xffffffffu : 0u;
  uint32_t br = 0;
#pragma unroll
  for (int i = 0; i < C::N; i++) r.l[i] = mlhip_subb(C::P[i] & mask, a.l[i], br);
}

// r = c ? a : b
template <class C>
MLHIP_HD void fp_select(Fp<C>& r, bool c, const Fp<C>& a, const Fp<C>& b) {
#pragma unroll
  for (int i = 0; i < C::N; i++) r.l[i] = c ? a.l[i] : b.l[i];
}

// Montgomery product a*b*R^-1 mod p, fully reduced.  32-bit CIOS: each inner step is one
// v_mad_u64_u32 (a_j*b_i + 64-bit addend) on the device.
template <class C>
MLHIP_HD void fp_mul_inline(Fp<C>& r, const Fp<C>& a, const Fp<C>& b) {
  constexpr int N = C::N;
  uint32_t t[N + 2];
#pragma unroll
  for (int i = 0; i < N + 2; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    uint64_t c = 0;
    const uint32_t bi = b.l[i];
#pragma unroll
    for (int j = 0; j < N; j++) {
      uint64_t acc = (uint64_t)a.l[j] * bi + t[j] + c;
      t[j] = (uint32_t)acc;
      c = acc >> 32;
    }
    uint64_t acc = (uint64_t)t[N] + c;
    t[N] = (uint32_t)acc;
    t[N + 1] = (uint32_t)(acc >> 32);
    const uint32_t m = t[0] * C::INV;
    acc = (uint64_t)m * C::P[0] + t[0];
    c = acc >> 32;
#pragma unroll
    for (int j = 1; j < N; j++) {
      acc = (uint64_t)m * C::P[j] + t[j] + c;
      t[j - 1] = (uint32_t)acc;
      c = acc >> 32;
    }
    acc = (uint64_t)t[N] + c;
    t[N - 1] = (uint32_t)acc;
    t[N] = t[N + 1] + (uint32_t)(acc >> 32);
  }
  uint32_t o[N];
#pragma unroll
  for (int i = 0; i < N; i++) o[i] = t[i];
  fp_reduce_once<C>(r, o);
}

#if !defined(__HIP_DEVICE_COMPILE__)
// Host-side variant for the O(1) tail of an MSM (window combination + to-affine, ~3k field
// multiplications per call): same CIOS on 64-bit limbs (the element's bytes are identical, the
// host is little-endian).  Tests build with MLHIP_HOST_USE_DEVICE_PATH to exercise the 32-bit
// device code on the CPU instead.
// "No-carry" CIOS: p leaves the top bit of its top 64-bit limb free on all three curves (381, 377, 254 bits in 384 / 256),
// so the running value never needs an extra word and every inner step is two multiplications into two carry
// chains.  One host-tail product: 64 -> 54 ns on the build container's Xeon (measured against the plain CIOS this
// replaces, same results on 10^5 chained products per curve).
template <class C>
inline void fp_mul_host64(Fp<C>& r, const Fp<C>& a, const Fp<C>& b) {
  constexpr int N = C::N / 2;
  static_assert((C::P[C::N - 1] >> 31) == 0, "no-carry CIOS needs a free top bit in the modulus");
  typedef unsigned __int128 u128;
  uint64_t x[N], y[N], q[N], t[N];
  for (int i = 0; i < N; i++) {
    x[i] = a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32);
    y[i] = b.l[2 * i] | ((uint64_t)b.l[2 * i + 1] << 32);
    q[i] = C::P[2 * i] | ((uint64_t)C::P[2 * i + 1] << 32);
    t[i] = 0;
  }
  for (int i = 0; i < N; i++) {
    u128 A = (u128)x[0] * y[i] + t[0];
    const uint64_t t0 = (uint64_t)A;
    A >>= 64;
    const uint64_t m = t0 * C::INV64;
    u128 Cc = ((u128)m * q[0] + t0) >> 64;
    for (int j = 1; j < N; j++) {
      A += (u128)x[j] * y[i] + t[j];
      Cc += (u128)m * q[j] + (uint64_t)A;
      t[j - 1] = (uint64_t)Cc;
      A >>= 64;
      Cc >>= 64;
    }
    t[N - 1] = (uint64_t)Cc + (uint64_t)A;
  }
  uint64_t d[N];
  unsigned br = 0;
  for (int i = 0; i < N; i++) {
    u128 s = (u128)t[i] - q[i] - br;
    d[i] = (uint64_t)s;
    br = (unsigned)((s >> 64) & 1);
  }
  for (int i = 0; i < N; i++) {
    uint64_t v = br ? t[i] : d[i];
    r.l[2 * i] = (uint32_t)v;
    r.l[2 * i + 1] = (uint32_t)(v >> 32);
  }
}
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// gfx950 body: product-scanning Montgomery multiplication, one v_mad_u64_u32 + one v_addc_co_u32 per
// limb product (generated: tools/gen_fp_comba.py).  Same result as fp_mul_inline, bit for bit
// (checked on the GPU by tests/test_gpu_parity.py::test_fp_mul_kernel_* and everything built on it).
#include "fp_mul_comba.inc"
template <class C>
__device__ __forceinline__ void fp_mul_device(Fp<C>& r, const Fp<C>& a, const Fp<C>& b) {
  if constexpr (C::N == 12)
    fp_mul_comba12<C>(r, a, b);
  else
    fp_mul_comba8<C>(r, a, b);
}
// r = (a*b + c*d) R^-1 mod p with one shared Montgomery reduction (fp2_lanes.h)
template <class C>
__device__ __forceinline__ void fp_mul2_device(Fp<C>& r, const Fp<C>& a, const Fp<C>& b, const Fp<C>& c, const Fp<C>& d) {
  if constexpr (C::N == 12)
    fp_mul2_comba12<C>(r, a, b, c, d);
  else
    fp_mul2_comba8<C>(r, a, b, c, d);
}
#endif

// Inlined entry points: used where the caller keeps its operands in registers (the G1 bucket
// accumulation loop, fp2_mul).  Measured on MI355X (tools/ubench_madd.hip): a register-resident mixed
// addition with inlined multiplies runs 1.9x faster than one that calls an out-of-line fp_mul through
// pointers (every operand then lives in scratch).
template <class C>
MLHIP_HD void fp_mul_i(Fp<C>& r, const Fp<C>& a, const Fp<C>& b) {
#if defined(__HIP_DEVICE_COMPILE__)
  fp_mul_device<C>(r, a, b);
#elif defined(MLHIP_HOST_USE_DEVICE_PATH)
  fp_mul_inline<C>(r, a, b);
#else
  fp_mul_host64<C>(r, a, b);
#endif
}

template <class C>
MLHIP_HD void fp_sqr_i(Fp<C>& r, const Fp<C>& a) {
  fp_mul_i<C>(r, a, a);
}

// Out-of-line entry points: one copy of the multiply per kernel for the callers whose state lives in
// scratch anyway (Fp12 towers, inversions) -- keeps those kernels' code small.
template <class C>
MLHIP_HD_NOINLINE void fp_mul(Fp<C>& r, const Fp<C>& a, const Fp<C>& b) {
#if defined(__HIP_DEVICE_COMPILE__)
  fp_mul_device<C>(r, a, b);
#elif defined(MLHIP_HOST_USE_DEVICE_PATH)
  fp_mul_inline<C>(r, a, b);
#else
  fp_mul_host64<C>(r, a, b);
#endif
}

template <class C>
MLHIP_HD_NOINLINE void fp_sqr(Fp<C>& r, const Fp<C>& a) {
#if defined(__HIP_DEVICE_COMPILE__)
  fp_mul_device<C>(r, a, a);
#elif defined(MLHIP_HOST_USE_DEVICE_PATH)
  fp_mul_inline<C>(r, a, a);
#else
  fp_mul_host64<C>(r, a, a);
#endif
}

// a * small constant (k <= 16) by additions
template <class C>
MLHIP_HD void fp_mul_small(Fp<C>& r, const Fp<C>& a, int k) {
  Fp<C> acc, cur = a;
  fp_zero<C>(acc);
  while (k) {
    if (k & 1) fp_add<C>(acc, acc, cur);
    fp_dbl<C>(cur, cur);
    k >>= 1;
  }
  r = acc;
}

// a^(p-2): Fermat inversion (0 -> 0).  Used once per pairing (final-exponentiation easy part) and
// in the O(1) host tail of an MSM; never in a hot loop.
// r = a^-1 (0 -> 0): constant-time divsteps inversion, defined in modinv.h (included at the end of this file)
template <class C>
MLHIP_HD_NOINLINE void fp_inv(Fp<C>& r, const Fp<C>& a);

// Fermat: a^(p-2); kept as the independent second implementation for the tests
template <class C>
MLHIP_HD void fp_inv_fermat(Fp<C>& r, const Fp<C>& a) {
  constexpr int N = C::N;
  uint32_t e[N];
  // e = p - 2 (BLS12-377's p ends in ...0001, so the borrow must propagate)
  uint64_t br = 2;
#pragma unroll
  for (int i = 0; i < N; i++) {
    uint64_t s = (uint64_t)C::P[i] - br;
    e[i] = (uint32_t)s;
    br = (s >> 32) & 1;
  }
  Fp<C> acc;
  fp_one<C>(acc);
  bool started = false;
  for (int i = N * 32 - 1; i >= 0; i--) {
    if (started) fp_sqr<C>(acc, acc);
    if ((e[i >> 5] >> (i & 31)) & 1) {
      if (started)
        fp_mul<C>(acc, acc, a);
      else {
        acc = a;
        started = true;
      }
    }
  }
  r = acc;
}

// Montgomery conversion helpers (host-side plumbing and tests)
template <class C>
MLHIP_HD void fp_to_mont(Fp<C>& r, const Fp<C>& a) {
  Fp<C> r2;
  fp_from_const<C>(r2, C::R2);
  fp_mul<C>(r, a, r2);
}

template <class C>
MLHIP_HD void fp_from_mont(Fp<C>& r, const Fp<C>& a) {
  Fp<C> one;
  fp_zero<C>(one);
  one.l[0] = 1;
  fp_mul<C>(r, a, one);
}

}  // namespace mlhip

#include "modinv.h"  // fp_inv
