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
                                                   Affine<F>* __restrict__ out, const uint32_t* __restrict__ skip = nullptr) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (skip && *skip) return;  // the fixed-base table of this base is still in `out` (k_fb_check)
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
// From FIXED_BASE_MIN scalars on, a table T[j][m-1] = [m 2^(wj)]P, m = 1 .. 2^(w-1), turns every product into
// ceil(256 / w) mixed additions of +-T[j][|d_j|] (signed w-bit digits, msm_window_digit) and no doubling.  The table is built
// by k_scalar_mul itself from plain-integer scalars -- one double-and-add wave time whatever its size, as long as its entries
// fit one wave per SIMD -- so w = 12 (22 windows x 2 048 points) instead of round 1's 32 unsigned bytes.  The table stays in
// the device's scratch buffer with its base as the key: a call with the same curve, group, width and base (k_fb_check
// compares on the device, the build kernels return at once) skips the build -- generators and Pedersen bases come back call
// after call.  2^20 products of one base, BLS12-381 (profiles/r03_perf_scalar_mul.txt): G1 10.2 -> 8.3 ms with the table built
// in the call, 4.1 ms with the table of an earlier call (double-and-add: 63 ms); G2 22.8 -> 19.6 / 13.2 ms (190 ms).
// MLHIP_FB_WINDOW = w (4 .. 14) overrides the width, MLHIP_FB_CACHE = 0 rebuilds the table in every call.
// The threshold: a first call costs what the double-and-add kernel costs (both are one wave time, ~4 ms G1 / ~6 ms G2), every
// later one a tenth of it.
constexpr size_t FIXED_BASE_MIN = (size_t)1 << 12;
constexpr int FB_W_G1 = 12, FB_W_G2 = 12, FB_W_MIN = 4, FB_W_MAX = 14;
constexpr size_t FB_HEADER = 512;  // bytes in front of the table: [flag | tag | base point (<= 192 bytes) | ... | FB_FLAG_ED]
// header word: 1 iff the base lies in the prime-order subgroup ([r]P = infinity, computed by the table build itself as one
// extra entry) on a curve with a twisted Edwards model -- the products then run on the 7-product Edwards addition (ed28.h)
constexpr uint32_t FB_FLAG_ED = 120;
MLHIP_HD int fb_windows(int w) { return (256 + w - 1) / w; }

struct FixedBaseScratch {
  char* buf = nullptr;
  size_t cap = 0;
  hipEvent_t last = nullptr;  // recorded after the last kernel that reads the buffer
};
static std::mutex g_fb_mu;
static FixedBaseScratch g_fb[64];  // per device (and per curve: this header is compiled into one translation unit per curve)

// mlhip_release_cache / mlhip_shutdown: the tables go too (20 MB at w = 12 for G2, 70 MB at w = 14); the next fixed-base
// call builds its table again.  Waits for the last kernel that reads each buffer.
static inline void fixed_base_release() {
  std::lock_guard<std::mutex> lk(g_fb_mu);
  int cur = 0;
  const bool have_cur = hipGetDevice(&cur) == hipSuccess;
  for (int dev = 0; dev < 64; dev++) {
    FixedBaseScratch& fb = g_fb[dev];
    if (!fb.buf && !fb.last) continue;
    (void)hipSetDevice(dev);
    if (fb.last) {
      (void)hipEventSynchronize(fb.last);
      (void)hipEventDestroy(fb.last);
      fb.last = nullptr;
    }
    if (fb.buf) (void)hipFree(fb.buf);
    fb.buf = nullptr;
    fb.cap = 0;
  }
  if (have_cur) (void)hipSetDevice(cur);
}

// header[0] = 1 iff the buffer holds the table of this (tag, base) already; otherwise header[0] = 0 and the tag is cleared
// until k_fb_commit, queued behind the kernels that build the table, stores the new key -- a call that fails between the
// two leaves no key, so the next one builds again
static __global__ void __launch_bounds__(64) k_fb_check(uint32_t* __restrict__ header, const uint32_t* __restrict__ base,
                                                        uint32_t nwords, uint32_t tag, int use_cache) {
  const uint32_t lane = threadIdx.x;
  bool same = header[1] == tag;
  if (lane < nwords) same = same && header[2 + lane] == base[lane];
  const bool all = __all(same) != 0 && use_cache != 0;
  if (lane == 0) {
    header[0] = all ? 1u : 0u;
    if (!all) header[1] = 0u;
  }
}
static __global__ void __launch_bounds__(64) k_fb_commit(uint32_t* __restrict__ header, const uint32_t* __restrict__ base,
                                                         uint32_t nwords, uint32_t tag, const uint32_t* __restrict__ r_times_base) {
  if (header[0]) return;  // the table was there already (and so is its subgroup flag)
  const uint32_t lane = threadIdx.x;
  if (lane < nwords) header[2 + lane] = base[lane];
  // r_times_base: the table build's extra entry [r]P (affine, (0, 0) = infinity) -- nullptr on curves without the Edwards path
  bool zero = true;
  if (r_times_base && lane < nwords) zero = r_times_base[lane] == 0u;
  const bool in_subgroup = r_times_base != nullptr && __all(zero) != 0;
  __syncthreads();
  if (lane == 0) {
    header[FB_FLAG_ED] = in_subgroup ? 1u : 0u;
    header[1] = tag;
  }
}

// the table's scalars: entry t = j * row + (m - 1) is the plain integer m 2^(wj); where that does not fit 256 bits (the
// upper part of the top window's row, which no scalar below 2^255 reaches) the entry is 0 = the point at infinity
struct FbOrder {
  uint32_t w[8];  // the group order r as a plain integer (the extra entry [r]P of a table on a curve with the Edwards path)
};
static __global__ void __launch_bounds__(256) k_fb_scalars(uint32_t* __restrict__ out, int w, uint32_t entries,
                                                           const uint32_t* __restrict__ skip, FbOrder order, int extra) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (*skip) return;
  if (extra && t == entries) {
#pragma unroll
    for (int k = 0; k < 8; k++) out[8 * t + k] = order.w[k];
    return;
  }
  if (t >= entries) return;
  const uint32_t row = 1u << (w - 1);
  const uint32_t j = t / row, m = t % row + 1;
  const uint32_t off = j * (uint32_t)w, word = off >> 5, sh = off & 31u;
  const uint64_t v = (uint64_t)m << sh;
  uint32_t o[8];
#pragma unroll
  for (int k = 0; k < 8; k++) o[k] = 0;
  bool fits = word < 8;
  if (fits) {
    o[word] = (uint32_t)v;
    if ((uint32_t)(v >> 32)) {
      if (word + 1 < 8)
        o[word + 1] = (uint32_t)(v >> 32);
      else
        fits = false;
    }
  }
#pragma unroll
  for (int k = 0; k < 8; k++) out[8 * t + k] = fits ? o[k] : 0u;
}

// the table in the form the product kernel of this call wants: Weierstrass rows, or (subgroup flag set) halved Niels triples
template <class C>
__global__ void __launch_bounds__(256) k_fb_table_to28(const Affine<FpField<C>>* __restrict__ points, size_t n,
                                                       Affine28<C>* __restrict__ out, const uint32_t* __restrict__ header, int ed_on) {
  if constexpr (C::HAS_EDWARDS)
    if (ed_on && header[FB_FLAG_ED]) return;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Affine28<C> q;
  affine28_from<C>(q, points[i]);
  out[i] = q;
}
template <class C>
__global__ void __launch_bounds__(256) k_fb_table_to_ed28(const Affine<FpField<C>>* __restrict__ points, size_t n,
                                                          EdNiels28<C>* __restrict__ out, const uint32_t* __restrict__ header, int ed_on) {
  if (!ed_on || !header[FB_FLAG_ED]) return;
  constexpr int K = 4;  // four points share one inversion (k_points_to_ed28)
  const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * K;
  if (i0 >= n) return;
  Affine<FpField<C>> in[K];
  Fp<C> xh[K], yh[K];
#pragma unroll
  for (int j = 0; j < K; j++) {
    if (i0 + j < n) {
      in[j] = points[i0 + j];
    } else {
      fp_zero<C>(in[j].x);
      fp_zero<C>(in[j].y);
    }
  }
  ed_affine_halves_batch<C, K>(xh, yh, in);
#pragma unroll
  for (int j = 0; j < K; j++) {
    if (i0 + j < n) {
      EdNiels28<C> q;
      ed_niels_from_halves<C>(q, xh[j], yh[j]);
      out[i0 + j] = q;
    }
  }
}

// [s_i]P from the Niels table of a base in the prime-order subgroup (BLS12-377 G1): ceil(256 / w) unified mixed additions of
// 7 products each -- no square, no carry propagation, no P = +-Q test (ed28_madd) -- against 10 products + the filter of the
// XYZZ form below; one conversion back to the Weierstrass curve (10 products) and one inversion per result.
template <class C>
__global__ void __launch_bounds__(64) k_fixed_base_ed(const EdNiels28<C>* __restrict__ table, const uint32_t* __restrict__ scalars,
                                                      int mont, size_t n, Affine<FpField<C>>* __restrict__ out, int w,
                                                      const uint32_t* __restrict__ header, int ed_on) {
  typedef FpField<C> F;
  if (!ed_on || !header[FB_FLAG_ED]) return;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s[8];
  fr_canonical<C>(s, scalars + 8 * i, mont != 0);
  EdExt28<C> acc;
  ed28_set_identity<C>(acc);
  const int nw = fb_windows(w);
  uint32_t carry = 0, neg = 0;
#pragma unroll 1
  for (int j = 0; j < nw; j++) {
    const uint32_t m = msm_window_digit(s, j * w, w, carry, neg);
    if (m) {
      const EdNiels28<C> q = table[((size_t)j << (w - 1)) + m - 1];
      ed28_madd<C>(acc, q, neg != 0);
    }
  }
  XYZZ28<C> x28;
  bool inf;
  ed28_to_xyzz28<C>(x28, inf, acc);
  XYZZ<F> r;
  xyzz28_to<C>(r, x28, inf);
  Affine<F> a;
  xyzz_to_affine<F>(a, r);
  out[i] = a;
}

template <class C>
__global__ void __launch_bounds__(64) k_fixed_base_g1(const Affine28<C>* __restrict__ table,
                                                      const uint32_t* __restrict__ scalars, int mont, size_t n,
                                                      Affine<FpField<C>>* __restrict__ out, int w,
                                                      const uint32_t* __restrict__ header, int ed_on) {
  typedef FpField<C> F;
  if constexpr (C::HAS_EDWARDS)
    if (ed_on && header[FB_FLAG_ED]) return;  // k_fixed_base_ed computes this call's products
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s[8];
  fr_canonical<C>(s, scalars + 8 * i, mont != 0);
  XYZZ28<C> acc;
  bool inf = true;
  const int nw = fb_windows(w);
  uint32_t carry = 0, neg = 0;
#pragma unroll 1
  for (int j = 0; j < nw; j++) {
    const uint32_t m = msm_window_digit(s, j * w, w, carry, neg);
    if (m) {
      const Affine28<C> q = table[((size_t)j << (w - 1)) + m - 1];
      xyzz28_madd<C>(acc, inf, q, neg != 0);
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
                                                   int mont, size_t n, Affine<F>* __restrict__ out, int w) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s[8];
  fr_canonical<C>(s, scalars + 8 * i, mont != 0);
  XYZZ<F> acc;
  xyzz_set_inf<F>(acc);
  const int nw = fb_windows(w);
  uint32_t carry = 0, neg = 0;
#pragma unroll 1
  for (int j = 0; j < nw; j++) {
    const uint32_t m = msm_window_digit(s, j * w, w, carry, neg);
    if (m) {
      Affine<F> q = table[((size_t)j << (w - 1)) + m - 1];
      typename F::T ny;
      F::neg(ny, q.y);
      F::select(q.y, neg != 0, ny, q.y);
      xyzz_madd_ool<F>(acc, q);
    }
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
                                                      Affine<Fp2Field<C>>* __restrict__ out,
                                                      const uint32_t* __restrict__ skip = nullptr) {
  typedef Fp2LField<C> FL;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = t >> 1;  // both lanes of a pair share i and the scalar: every branch below is pair-uniform
  if (i >= n) return;
  if (skip && *skip) return;
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
                                                      Affine<Fp2Field<C>>* __restrict__ out, int w) {
  typedef Fp2LField<C> FL;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = t >> 1;
  if (i >= n) return;
  const int hi = (int)(threadIdx.x & 1u);
  uint32_t s[8];
  fr_canonical<C>(s, scalars + 8 * i, mont != 0);
  XYZZ<FL> acc;
  xyzz_set_inf<FL>(acc);
  const int nw = fb_windows(w);
  uint32_t carry = 0, neg = 0;
#pragma unroll 1
  for (int j = 0; j < nw; j++) {
    const uint32_t m = msm_window_digit(s, j * w, w, carry, neg);  // pair-uniform
    if (m) {
      Affine<FL> q;
      lp_load_affine<C>(q, table, ((size_t)j << (w - 1)) + m - 1, hi);
      xyzz_madd<FL>(acc, q, neg != 0);
    }
  }
  Affine<FL> r;
  xyzz_to_affine<FL>(r, acc);
  Fp<C>* o = reinterpret_cast<Fp<C>*>(out + i);
  o[hi] = r.x.v;
  o[2 + hi] = r.y.v;
}

// the same on the carry-free lane-pair form (ec28_lp.h, every curve since round 3): the table converted once by
// k_points_to28_g2, one carry-free mixed addition per window, one conversion back per result
template <class C>
__global__ void __launch_bounds__(64) k_fixed_base_lp28(const AffineG2_28<C>* __restrict__ table,
                                                        const uint32_t* __restrict__ scalars, int mont, size_t n,
                                                        Affine<Fp2Field<C>>* __restrict__ out, int w) {
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
  const int nw = fb_windows(w);
  uint32_t carry = 0, neg = 0;
#pragma unroll 1
  for (int j = 0; j < nw; j++) {
    const uint32_t m = msm_window_digit(s, j * w, w, carry, neg);
    if (m) {  // pair-uniform: both lanes hold the same scalar
      const AffineG2_28<C>* e = table + (((size_t)j << (w - 1)) + m - 1);
      Affine28L<Fp28<C>> q;
      q.x = e->c[hi];
      q.y = e->c[2 + hi];
      xyzz28_lp_madd<C, B>(acc, inf, q, neg != 0);
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
  const bool one_lane = mlhip_alt_switch("MLHIP_SCALAR_MUL_ONE_LANE");  // G2 on one lane per product: test build only
  (void)one_lane;
  size_t fb_min = FIXED_BASE_MIN;  // MLHIP_FIXED_BASE_MIN overrides (0 = never: always the double-and-add kernel)
  if (const char* e = getenv("MLHIP_FIXED_BASE_MIN")) {
    const long long v = atoll(e);
    fb_min = v <= 0 ? SIZE_MAX : (size_t)v;
  }
  if (point_stride == 0 && n >= fb_min) {
    constexpr bool kG1 = std::is_same<F, FpField<C>>::value;
    int w = kG1 ? FB_W_G1 : FB_W_G2;
    if (const char* e = getenv("MLHIP_FB_WINDOW")) {
      const int v = atoi(e);
      if (v >= FB_W_MIN && v <= FB_W_MAX) w = v;
    }
    const char* ce = getenv("MLHIP_FB_CACHE");
    const int use_cache = !(ce && ce[0] == '0');
    const size_t entries = (size_t)fb_windows(w) << (w - 1);
    // a curve with the twisted Edwards model builds one entry more, [r]P: (0, 0) iff the base is in the prime-order subgroup
    constexpr bool kEd = kG1 && C::HAS_EDWARDS;
    const size_t built = entries + (kEd ? 1 : 0);
    // scratch: [header | table | its scalars | the table in the carry-free form] in one persistent buffer per device.
    // Calls on different streams reuse it in the order they take the lock: each waits for the event the previous
    // one recorded after its last kernel.  (hipMallocAsync here gave intermittently wrong results on this runtime.)
    const size_t tab_bytes = built * sizeof(Affine<F>), sc_bytes = built * 32;
    const size_t t28_bytes = kG1 ? entries * (kEd ? sizeof(EdNiels28<C>) : sizeof(Affine28<C>)) : entries * sizeof(AffineG2_28<C>);
    const size_t need = FB_HEADER + tab_bytes + sc_bytes + t28_bytes;
    static_assert(8 + sizeof(Affine<F>) <= FB_HEADER && sizeof(Affine<F>) / 4 <= 64, "the key fits the header and one wave");
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
      HIPCHK(hipMemsetAsync(fb.buf, 0, FB_HEADER, st));  // tag 0: no table yet -- on the call's stream, ahead of k_fb_check
    }
    if (!fb.last)
      HIPCHK(hipEventCreateWithFlags(&fb.last, hipEventDisableTiming));
    else
      HIPCHK(hipStreamWaitEvent(st, fb.last, 0));
    uint32_t* header = (uint32_t*)fb.buf;
    char* scratch = fb.buf + FB_HEADER;
    Affine<F>* table = (Affine<F>*)scratch;
    uint32_t* tsc = (uint32_t*)(scratch + tab_bytes);
    // what the table depends on beside the base: curve, group, width, and which kernel built it (the one-lane G2 kernel
    // and the lane-pair one give the same bytes; the bit only keeps an A/B honest)
    const uint32_t tag = 0x80000000u | ((uint32_t)C::ID << 16) | ((kG1 ? 1u : 2u) << 8) | (uint32_t)w | (one_lane ? 0x4000u : 0u);
    k_fb_check<<<dim3(1), dim3(64), 0, st>>>(header, (const uint32_t*)d_points, (uint32_t)(sizeof(Affine<F>) / 4), tag, use_cache);
    FbOrder order;
    for (int k = 0; k < 8; k++) order.w[k] = C::FR[k];
    k_fb_scalars<<<dim3((unsigned)((built + 255) / 256)), dim3(256), 0, st>>>(tsc, w, (uint32_t)entries, header, order, kEd ? 1 : 0);
    if constexpr (!kG1) {
      if (!one_lane)
        k_scalar_mul_lp<C><<<dim3((unsigned)((2 * entries + 63) / 64)), dim3(64), 0, st>>>((const Affine<F>*)d_points, 0, tsc,
                                                                                        -1, entries, table, header);
      else if constexpr (kBuildAlt)
        k_scalar_mul<C, F><<<dim3((unsigned)((entries + 63) / 64)), dim3(64), 0, st>>>((const Affine<F>*)d_points, 0, tsc, -1,
                                                                                       entries, table, header);
    } else {
      k_scalar_mul<C, F><<<dim3((unsigned)((built + 63) / 64)), dim3(64), 0, st>>>((const Affine<F>*)d_points, 0, tsc, -1,
                                                                                   built, table, header);
    }
    k_fb_commit<<<dim3(1), dim3(64), 0, st>>>(header, (const uint32_t*)d_points, (uint32_t)(sizeof(Affine<F>) / 4), tag,
                                              kEd ? (const uint32_t*)(table + entries) : nullptr);
    if constexpr (kG1) {
      // the header's subgroup flag picks ONE of the two product kernels (the other returns at once): the Edwards form for a
      // base in the prime-order subgroup -- generators, Pedersen bases -- the XYZZ form for any other curve point
      char* t28raw = scratch + tab_bytes + sc_bytes;
      const char* ede = getenv("MLHIP_EDWARDS");  // =0: the XYZZ products for every base (second implementation, A/B)
      const int ed_on = (ede && ede[0] == '0') ? 0 : 1;
      k_fb_table_to28<C><<<dim3((unsigned)((entries + 255) / 256)), dim3(256), 0, st>>>(table, entries, (Affine28<C>*)t28raw, header, ed_on);
      if constexpr (kEd) {
        k_fb_table_to_ed28<C><<<dim3((unsigned)(((entries + 3) / 4 + 255) / 256)), dim3(256), 0, st>>>(table, entries,
                                                                                                  (EdNiels28<C>*)t28raw, header, ed_on);
        k_fixed_base_ed<C><<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st>>>((const EdNiels28<C>*)t28raw, (const uint32_t*)d_scalars,
                                                                               mont, n, (Affine<F>*)d_out, w, header, ed_on);
      }
      k_fixed_base_g1<C><<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st>>>((const Affine28<C>*)t28raw, (const uint32_t*)d_scalars,
                                                                             mont, n, (Affine<F>*)d_out, w, header, ed_on);
    } else {
      const bool a32 = mlhip_alt_switch("MLHIP_ACC32");  // =1: the boundary-form lane-pair kernel (second implementation)
      if (!one_lane && !a32) {
        AffineG2_28<C>* t28 = (AffineG2_28<C>*)(scratch + tab_bytes + sc_bytes);
        k_points_to28_g2<C><<<dim3((unsigned)((4 * entries + 255) / 256)), dim3(256), 0, st>>>(table, entries, t28);
        k_fixed_base_lp28<C><<<dim3((unsigned)((2 * n + 63) / 64)), dim3(64), 0, st>>>(t28, (const uint32_t*)d_scalars, mont, n,
                                                                                    (Affine<F>*)d_out, w);
      } else if constexpr (kBuildAlt) {
        if (!one_lane)
          k_fixed_base_lp<C><<<dim3((unsigned)((2 * n + 63) / 64)), dim3(64), 0, st>>>(table, (const uint32_t*)d_scalars, mont, n,
                                                                                    (Affine<F>*)d_out, w);
        else
          k_fixed_base<C, F><<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st>>>(table, (const uint32_t*)d_scalars, mont, n,
                                                                               (Affine<F>*)d_out, w);
      }
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
  // G1, or (test build) G2 on one lane per product
  if constexpr (std::is_same<F, FpField<C>>::value || kBuildAlt)
    k_scalar_mul<C, F><<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st>>>((const Affine<F>*)d_points, point_stride,
                                                                            (const uint32_t*)d_scalars, mont, n,
                                                                            (Affine<F>*)d_out);
  HIPCHK(hipGetLastError());
  return 0;
}

}  // namespace mlhip
