// msm_scalar_mul.h -- batched single-scalar multiplication (G1.Mul / G2.Mul): windowed double-and-add per pair, and the
// fixed-base table path for one base and many scalars.  Part of msm_kernels.h.
#pragma once
// (included by msm_kernels.h after its common headers and constants)

namespace mlhip {

// Signed 4-bit windows of a 256-bit integer: s + 0x88..8 has the nibbles d_w + 8 with d_w in [-8, 7] and
// s = sum_w d_w 16^w + t[8] 16^64 (t[8] = the carry out of the addition: 0 for a scalar below 2^255).  The carries of the
// recoding are the carries of one 256-bit addition -- no per-window branch; the table holds {1..8}P, half of the
// unsigned form's, and every lane of a wave adds at the same loop positions (a windowed NAF would not: its non-zero
// digits sit at data-dependent positions, which serialises the lanes).
__device__ __forceinline__ void signed_windows4(uint32_t t[9], const uint32_t s[8]) {
  uint64_t c = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    c += (uint64_t)s[k] + 0x88888888u;
    t[k] = (uint32_t)c;
    c >>= 32;
  }
  t[8] = (uint32_t)c;
}
__device__ __forceinline__ int signed_window4_digit(const uint32_t t[9], int w) {
  return w == 64 ? (int)t[8] : (int)((t[w >> 3] >> ((w & 7) * 4)) & 15u) - 8;
}

// out[i] = [s_i] P_i: batched single-scalar multiplication (the reference's G1.Mul / G2.Mul,
// driver/gurvy/bls12381/bls12-381.go:238-247, :342-351; double-and-add shape of :920-932), one lane per
// product, signed 4-bit fixed windows: 8-entry table in scratch, 4 doublings + 1 addition per window.
template <class C, class F>
__global__ void __launch_bounds__(64) k_scalar_mul(const Affine<F>* __restrict__ points, size_t point_stride,
                                                   const uint32_t* __restrict__ scalars, int mont, size_t n,
                                                   Affine<F>* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s[8];
  if (mont < 0) {  // plain 256-bit integers, not reduced mod r (the fixed-base table: [d 2^(8j)]P for ANY P on the curve)
#pragma unroll
    for (int k = 0; k < 8; k++) s[k] = scalars[8 * i + k];
  } else {
    fr_canonical<C>(s, scalars + 8 * i, mont != 0);
  }
  const Affine<F> P = points[i * point_stride];
  uint32_t sw[9];
  signed_windows4(sw, s);
  XYZZ<F> tab[8];
  xyzz_from_affine<F>(tab[0], P);
  for (int k = 1; k < 8; k++) {
    tab[k] = tab[k - 1];
    xyzz_madd_ool<F>(tab[k], P);
  }
  XYZZ<F> acc;
  xyzz_set_inf<F>(acc);
  bool started = false;
  constexpr bool kInline = std::is_same<F, FpField<C>>::value;  // G1: the running point stays in registers
#pragma unroll 1
  for (int w = 64; w >= 0; w--) {
    if (started) {
#pragma unroll 1
      for (int d = 0; d < 4; d++) {
        XYZZ<F> t;
        if constexpr (kInline)
          xyzz_dbl<F>(t, acc);
        else
          xyzz_dbl_ool<F>(t, acc);
        acc = t;
      }
    }
    const int d = signed_window4_digit(sw, w);
    if (d) {
      XYZZ<F> q = tab[(d < 0 ? -d : d) - 1];
      typename F::T ny;
      F::neg(ny, q.y);
      F::select(q.y, d < 0, ny, q.y);
      if constexpr (kInline)
        xyzz_add<F>(acc, q);
      else
        xyzz_add_ool<F>(acc, q);
      started = true;
    }
  }
  Affine<F> r;
  xyzz_to_affine<F>(r, acc);
  out[i] = r;
}

// ---- one base, many scalars (point_stride = 0: [s_i]G for generators, Pedersen bases ...) ------------------------
// From FIXED_BASE_MIN scalars on, the table T[j][d-1] = [d 2^(8j)]P (32 windows x 255 affine points, built by
// k_scalar_mul itself from 8160 plain-integer scalars) turns every product into <= 32 mixed additions and no doubling:
// 2^20 G1 products in 10 ms instead of 70 ms, G2 in 26 ms instead of 209 ms (profiles/r01_perf_scalar_mul.txt).  G1 runs the additions in the carry-free form (ec28.h).
constexpr int FB_WINDOWS = 32, FB_ROW = 255;
constexpr size_t FIXED_BASE_MIN = (size_t)1 << 16;  // the table costs one double-and-add wave time (G1 ~4 ms, G2 ~7 ms)

struct FixedBaseScratch {
  char* buf = nullptr;
  size_t cap = 0;
  hipEvent_t last = nullptr;  // recorded after the last kernel that reads the buffer
};
static std::mutex g_fb_mu;
static FixedBaseScratch g_fb[64];  // per device

static __global__ void __launch_bounds__(256) k_fb_scalars(uint32_t* __restrict__ out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= FB_WINDOWS * FB_ROW) return;
  const uint32_t j = t / FB_ROW, d = t % FB_ROW + 1;
#pragma unroll
  for (int k = 0; k < 8; k++) out[8 * t + k] = 0;
  out[8 * t + (j >> 2)] = d << ((j & 3) * 8);
}

template <class C>
__global__ void __launch_bounds__(64) k_fixed_base_g1(const Affine28<C>* __restrict__ table,
                                                      const uint32_t* __restrict__ scalars, int mont, size_t n,
                                                      Affine<FpField<C>>* __restrict__ out) {
  typedef FpField<C> F;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s[8];
  fr_canonical<C>(s, scalars + 8 * i, mont != 0);
  XYZZ28<C> acc;
  bool inf = true;
#pragma unroll 1
  for (int j = 0; j < FB_WINDOWS; j++) {
    const uint32_t d = (s[j >> 2] >> ((j & 3) * 8)) & 255u;
    if (d) {
      const Affine28<C> q = table[j * FB_ROW + d - 1];
      xyzz28_madd<C>(acc, inf, q, false);
    }
  }
  XYZZ<F> r;
  xyzz28_to<C>(r, acc, inf);
  Affine<F> a;
  xyzz_to_affine<F>(a, r);
  out[i] = a;
}

template <class C, class F>
__global__ void __launch_bounds__(64) k_fixed_base(const Affine<F>* __restrict__ table, const uint32_t* __restrict__ scalars,
                                                   int mont, size_t n, Affine<F>* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s[8];
  fr_canonical<C>(s, scalars + 8 * i, mont != 0);
  XYZZ<F> acc;
  xyzz_set_inf<F>(acc);
#pragma unroll 1
  for (int j = 0; j < FB_WINDOWS; j++) {
    const uint32_t d = (s[j >> 2] >> ((j & 3) * 8)) & 255u;
    if (d) xyzz_madd_ool<F>(acc, table[j * FB_ROW + d - 1]);
  }
  Affine<F> a;
  xyzz_to_affine<F>(a, acc);
  out[i] = a;
}

// ---- the same two kernels for G2 over lane pairs (two adjacent lanes per product, one Fp2 component each): the one-lane
// Fp2 forms above keep ~340 live words per lane and run out of registers; MLHIP_SCALAR_MUL_ONE_LANE=1 selects them
template <class C>
__global__ void __launch_bounds__(64) k_scalar_mul_lp(const Affine<Fp2Field<C>>* __restrict__ points, size_t point_stride,
                                                      const uint32_t* __restrict__ scalars, int mont, size_t n,
                                                      Affine<Fp2Field<C>>* __restrict__ out) {
  typedef Fp2LField<C> FL;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = t >> 1;  // both lanes of a pair share i and the scalar: every branch below is pair-uniform
  if (i >= n) return;
  const int hi = (int)(threadIdx.x & 1u);
  uint32_t s[8];
  if (mont < 0) {
#pragma unroll
    for (int k = 0; k < 8; k++) s[k] = scalars[8 * i + k];
  } else {
    fr_canonical<C>(s, scalars + 8 * i, mont != 0);
  }
  Affine<FL> P;
  lp_load_affine<C>(P, points, i * point_stride, hi);
  uint32_t sw[9];
  signed_windows4(sw, s);
  XYZZ<FL> tab[8];
  xyzz_from_affine<FL>(tab[0], P);
#pragma unroll 1
  for (int k = 1; k < 8; k++) {
    tab[k] = tab[k - 1];
    xyzz_madd<FL>(tab[k], P, false);
  }
  XYZZ<FL> acc;
  xyzz_set_inf<FL>(acc);
  bool started = false;
#pragma unroll 1
  for (int w = 64; w >= 0; w--) {
    if (started) {
#pragma unroll 1
      for (int d = 0; d < 4; d++) {
        XYZZ<FL> d2;
        xyzz_dbl<FL>(d2, acc);
        acc = d2;
      }
    }
    const int d = signed_window4_digit(sw, w);  // pair-uniform
    if (d) {
      XYZZ<FL> q = tab[(d < 0 ? -d : d) - 1];
      typename FL::T ny;
      FL::neg(ny, q.y);
      FL::select(q.y, d < 0, ny, q.y);
      xyzz_add_lp_ool<C>(acc, q);
      started = true;
    }
  }
  Affine<FL> r;
  xyzz_to_affine<FL>(r, acc);
  Fp<C>* o = reinterpret_cast<Fp<C>*>(out + i);
  o[hi] = r.x.v;
  o[2 + hi] = r.y.v;
}

template <class C>
__global__ void __launch_bounds__(64) k_fixed_base_lp(const Affine<Fp2Field<C>>* __restrict__ table,
                                                      const uint32_t* __restrict__ scalars, int mont, size_t n,
                                                      Affine<Fp2Field<C>>* __restrict__ out) {
  typedef Fp2LField<C> FL;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = t >> 1;
  if (i >= n) return;
  const int hi = (int)(threadIdx.x & 1u);
  uint32_t s[8];
  fr_canonical<C>(s, scalars + 8 * i, mont != 0);
  XYZZ<FL> acc;
  xyzz_set_inf<FL>(acc);
#pragma unroll 1
  for (int j = 0; j < FB_WINDOWS; j++) {
    const uint32_t d = (s[j >> 2] >> ((j & 3) * 8)) & 255u;
    if (d) {
      Affine<FL> q;
      lp_load_affine<C>(q, table, (size_t)j * FB_ROW + d - 1, hi);
      xyzz_madd<FL>(acc, q, false);
    }
  }
  Affine<FL> r;
  xyzz_to_affine<FL>(r, acc);
  Fp<C>* o = reinterpret_cast<Fp<C>*>(out + i);
  o[hi] = r.x.v;
  o[2 + hi] = r.y.v;
}

// the same on the carry-free lane-pair form (ec28_lp.h, every curve since round 3): the table converted once by
// k_points_to28_g2, <= 32 carry-free mixed additions per scalar, one conversion back per result
template <class C>
__global__ void __launch_bounds__(64) k_fixed_base_lp28(const AffineG2_28<C>* __restrict__ table,
                                                        const uint32_t* __restrict__ scalars, int mont, size_t n,
                                                        Affine<Fp2Field<C>>* __restrict__ out) {
  typedef Fp2LField<C> FL;
  typedef PairDevice<C> B;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = t >> 1;
  if (i >= n) return;
  const int hi = (int)(threadIdx.x & 1u);
  uint32_t s[8];
  fr_canonical<C>(s, scalars + 8 * i, mont != 0);
  XYZZ28L<Fp28<C>> acc;
  bool inf = true;
#pragma unroll 1
  for (int j = 0; j < FB_WINDOWS; j++) {
    const uint32_t d = (s[j >> 2] >> ((j & 3) * 8)) & 255u;
    if (d) {  // pair-uniform: both lanes hold the same scalar
      const AffineG2_28<C>* e = table + ((size_t)j * FB_ROW + d - 1);
      Affine28L<Fp28<C>> q;
      q.x = e->c[hi];
      q.y = e->c[2 + hi];
      xyzz28_lp_madd<C, B>(acc, inf, q, false);
    }
  }
  XYZZ<FL> r;
  if (inf) {
    xyzz_set_inf<FL>(r);
  } else {
    fp28_to_fp<C>(r.x.v, acc.x);
    fp28_to_fp<C>(r.y.v, acc.y);
    fp28_to_fp<C>(r.zz.v, acc.zz);
    fp28_to_fp<C>(r.zzz.v, acc.zzz);
  }
  Affine<FL> a;
  xyzz_to_affine<FL>(a, r);
  Fp<C>* o = reinterpret_cast<Fp<C>*>(out + i);
  o[hi] = a.x.v;
  o[2 + hi] = a.y.v;
}

template <class C, class F>
int scalar_mul_device(const void* d_points, size_t point_stride, const void* d_scalars, int mont, size_t n, void* d_out,
                      hipStream_t st) {
  if (n == 0) return 0;
  const char* ol = getenv("MLHIP_SCALAR_MUL_ONE_LANE");
  const bool one_lane = ol && ol[0] == '1';
  (void)one_lane;
  size_t fb_min = FIXED_BASE_MIN;  // MLHIP_FIXED_BASE_MIN overrides (0 = never: always the double-and-add kernel)
  if (const char* e = getenv("MLHIP_FIXED_BASE_MIN")) {
    const long long v = atoll(e);
    fb_min = v <= 0 ? SIZE_MAX : (size_t)v;
  }
  if (point_stride == 0 && n >= fb_min) {
    constexpr size_t kEntries = (size_t)FB_WINDOWS * FB_ROW;
    constexpr bool kG1 = std::is_same<F, FpField<C>>::value;
    // scratch: [table | its 8160 scalars | (G1) the table in the carry-free form] in one persistent buffer per device.
    // Calls on different streams reuse it in the order they take the lock: each waits for the event the previous
    // one recorded after its last kernel.  (hipMallocAsync here gave intermittently wrong results on this runtime.)
    const size_t tab_bytes = kEntries * sizeof(Affine<F>), sc_bytes = kEntries * 32;
    const size_t t28_bytes = kG1 ? kEntries * sizeof(Affine28<C>) : kEntries * sizeof(AffineG2_28<C>);
    const size_t need = tab_bytes + sc_bytes + t28_bytes;
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_fb_mu);
    FixedBaseScratch& fb = g_fb[dev & 63];
    if (need > fb.cap) {
      if (fb.buf) HIPCHK(hipFree(fb.buf));  // waits for the device: no earlier user is still reading it
      fb.buf = nullptr;
      fb.cap = 0;
      HIPCHK(hipMalloc((void**)&fb.buf, need));
      fb.cap = need;
    }
    if (!fb.last)
      HIPCHK(hipEventCreateWithFlags(&fb.last, hipEventDisableTiming));
    else
      HIPCHK(hipStreamWaitEvent(st, fb.last, 0));
    char* scratch = fb.buf;
    Affine<F>* table = (Affine<F>*)scratch;
    uint32_t* tsc = (uint32_t*)(scratch + tab_bytes);
    k_fb_scalars<<<dim3((unsigned)((kEntries + 255) / 256)), dim3(256), 0, st>>>(tsc);
    if constexpr (!kG1) {
      if (!one_lane)
        k_scalar_mul_lp<C><<<dim3((unsigned)((2 * kEntries + 63) / 64)), dim3(64), 0, st>>>((const Affine<F>*)d_points, 0, tsc,
                                                                                         -1, kEntries, table);
      else
        k_scalar_mul<C, F><<<dim3((unsigned)((kEntries + 63) / 64)), dim3(64), 0, st>>>((const Affine<F>*)d_points, 0, tsc, -1,
                                                                                        kEntries, table);
    } else {
      k_scalar_mul<C, F><<<dim3((unsigned)((kEntries + 63) / 64)), dim3(64), 0, st>>>((const Affine<F>*)d_points, 0, tsc, -1,
                                                                                      kEntries, table);
    }
    if constexpr (kG1) {
      Affine28<C>* t28 = (Affine28<C>*)(scratch + tab_bytes + sc_bytes);
      k_points_to28<C><<<dim3((unsigned)((kEntries + 255) / 256)), dim3(256), 0, st>>>(table, kEntries, t28);
      k_fixed_base_g1<C><<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st>>>(t28, (const uint32_t*)d_scalars, mont, n,
                                                                             (Affine<F>*)d_out);
    } else {
      const char* a32 = getenv("MLHIP_ACC32");  // =1: the boundary-form lane-pair kernel (second implementation)
      if (!one_lane && !(a32 && a32[0] == '1')) {
        AffineG2_28<C>* t28 = (AffineG2_28<C>*)(scratch + tab_bytes + sc_bytes);
        k_points_to28_g2<C><<<dim3((unsigned)((4 * kEntries + 255) / 256)), dim3(256), 0, st>>>(table, kEntries, t28);
        k_fixed_base_lp28<C><<<dim3((unsigned)((2 * n + 63) / 64)), dim3(64), 0, st>>>(t28, (const uint32_t*)d_scalars, mont, n,
                                                                                    (Affine<F>*)d_out);
      } else if (!one_lane)
        k_fixed_base_lp<C><<<dim3((unsigned)((2 * n + 63) / 64)), dim3(64), 0, st>>>(table, (const uint32_t*)d_scalars, mont, n,
                                                                                  (Affine<F>*)d_out);
      else
        k_fixed_base<C, F><<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st>>>(table, (const uint32_t*)d_scalars, mont, n,
                                                                             (Affine<F>*)d_out);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(fb.last, st));
    return 0;
  }
  if constexpr (!std::is_same<F, FpField<C>>::value) {
    if (!one_lane) {
      k_scalar_mul_lp<C><<<dim3((unsigned)((2 * n + 63) / 64)), dim3(64), 0, st>>>((const Affine<F>*)d_points, point_stride,
                                                                                (const uint32_t*)d_scalars, mont, n,
                                                                                (Affine<F>*)d_out);
      HIPCHK(hipGetLastError());
      return 0;
    }
  }
  k_scalar_mul<C, F><<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st>>>((const Affine<F>*)d_points, point_stride,
                                                                          (const uint32_t*)d_scalars, mont, n,
                                                                          (Affine<F>*)d_out);
  HIPCHK(hipGetLastError());
  return 0;
}

}  // namespace mlhip
