// msm_g2.h -- G2 over lane pairs: accumulation (boundary form and carry-free form), segmented variants, bucket
// reduction.  Part of msm_kernels.h.
#pragma once
// (included by msm_kernels.h after its common headers and constants)

namespace mlhip {

// ---- G2 over lane pairs -----------------------------------------------------------------------------
// A G2 bucket is owned by two adjacent lanes, one Fp2 component each (fp2_lanes.h): the XYZZ accumulator is
// 4 x 12 words per lane -- the G1 footprint -- so the mixed addition stays in registers (one Fp2 element per
// lane needs ~340 live words and spills), and every Fp2 product is one fused dual Montgomery product.
template <class C>
struct Fp2LField {
  using Curve = C;
  using T = Fp2L<C>;
  MLHIP_HD static void zero(T& r) { fp2_zero<C>(r); }
  MLHIP_HD static void one(T& r) { fp2_one<C>(r); }
  MLHIP_HD static bool is_zero(const T& a) { return fp2_is_zero<C>(a); }
  MLHIP_HD static bool eq(const T& a, const T& b) { return fp2_eq<C>(a, b); }
  MLHIP_HD static void add(T& r, const T& a, const T& b) { fp2_add<C>(r, a, b); }
  MLHIP_HD static void sub(T& r, const T& a, const T& b) { fp2_sub<C>(r, a, b); }
  MLHIP_HD static void dbl(T& r, const T& a) { fp2_dbl<C>(r, a); }
  MLHIP_HD static void neg(T& r, const T& a) { fp2_neg<C>(r, a); }
  MLHIP_HD static void mul(T& r, const T& a, const T& b) { fp2_mul<C>(r, a, b); }
  MLHIP_HD static void sqr(T& r, const T& a) { fp2_sqr<C>(r, a); }
  MLHIP_HD static void inv(T& r, const T& a) { fp2_inv<C>(r, a); }
  MLHIP_HD static void select(T& r, bool c, const T& a, const T& b) { fp2_select<C>(r, c, a, b); }
};

// component loads / stores between the AoS Fp2 layout in memory and the lane-pair registers
template <class C>
__device__ __forceinline__ void lp_load_affine(Affine<Fp2LField<C>>& p, const Affine<Fp2Field<C>>* pts, size_t idx, int hi) {
  const Fp<C>* q = reinterpret_cast<const Fp<C>*>(pts + idx);
  p.x.v = q[hi];
  p.y.v = q[2 + hi];
}
template <class C>
__device__ __forceinline__ void lp_load_xyzz(XYZZ<Fp2LField<C>>& r, const XYZZ<Fp2Field<C>>* src, size_t idx, int hi) {
  const Fp<C>* q = reinterpret_cast<const Fp<C>*>(src + idx);
  r.x.v = q[hi];
  r.y.v = q[2 + hi];
  r.zz.v = q[4 + hi];
  r.zzz.v = q[6 + hi];
}
template <class C>
__device__ __forceinline__ void lp_store_xyzz(XYZZ<Fp2Field<C>>* dst, size_t idx, const XYZZ<Fp2LField<C>>& r, int hi) {
  Fp<C>* q = reinterpret_cast<Fp<C>*>(dst + idx);
  q[hi] = r.x.v;
  q[2 + hi] = r.y.v;
  q[4 + hi] = r.zz.v;
  q[6 + hi] = r.zzz.v;
}
template <class C>
__device__ __noinline__ void xyzz_add_lp_ool(XYZZ<Fp2LField<C>>& acc, const XYZZ<Fp2LField<C>>& q) {
  xyzz_add<Fp2LField<C>>(acc, q);
}

template <class C>
__global__ void __launch_bounds__(256) k_accumulate_lp(const Affine<Fp2Field<C>>* __restrict__ points,
                                                       const uint32_t* __restrict__ sorted,
                                                       const uint32_t* __restrict__ offsets,
                                                       const uint32_t* __restrict__ counts, size_t n_buckets,
                                                       const uint32_t* __restrict__ order, uint32_t big_threshold,
                                                       uint32_t* __restrict__ big_list, uint32_t* __restrict__ big_count,
                                                       XYZZ<Fp2Field<C>>* __restrict__ buckets) {
  typedef Fp2LField<C> FL;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t pair = t >> 1;  // both lanes of a pair share the bucket: every branch below is pair-uniform
  if (pair >= n_buckets) return;
  const int hi = lane_is_hi() ? 1 : 0;
  const size_t g = order[pair];
  const uint32_t cnt = counts[g];
  if (cnt > big_threshold) {
    if (!hi) {
      uint32_t pos = atomicAdd(big_count, 1u);
      big_list[pos] = (uint32_t)g;
    }
    return;
  }
  XYZZ<FL> acc;
  xyzz_set_inf<FL>(acc);
  const size_t begin = offsets[g], end = begin + cnt;
  if (begin < end) {
    uint32_t e = sorted[begin];
    Affine<FL> p;
    lp_load_affine<C>(p, points, e & 0x7fffffffu, hi);
    for (size_t k = begin; k < end; k++) {
      uint32_t en = e;
      Affine<FL> pn = p;
      if (k + 1 < end) {
        en = sorted[k + 1];
        lp_load_affine<C>(pn, points, en & 0x7fffffffu, hi);
      }
      xyzz_madd<FL>(acc, p, (e >> 31) != 0);
      e = en;
      p = pn;
    }
  }
  lp_store_xyzz<C>(buckets, g, acc, hi);
}

// ---- slice sums of long G2 buckets over lane pairs (k_big_slices is the one-lane form: ~340 live words per lane) ------
// BLOCK / 2 pairs stride over the slice's entries with the lane-pair mixed addition, then an LDS tree of lane-pair
// additions; slice bookkeeping as in k_big_slices (msm_accumulate.h).
template <class C, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_big_slices_lp(const Affine<Fp2Field<C>>* __restrict__ points,
                                                         const uint32_t* __restrict__ sorted,
                                                         const uint32_t* __restrict__ offsets,
                                                         const uint32_t* __restrict__ counts,
                                                         const uint32_t* __restrict__ big_list,
                                                         const uint32_t* __restrict__ big_count,
                                                         const uint32_t* __restrict__ prefix,
                                                         XYZZ<Fp2Field<C>>* __restrict__ partials) {
  typedef Fp2LField<C> FL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  XYZZ<Fp2Field<C>>* sh = reinterpret_cast<XYZZ<Fp2Field<C>>*>(smem);
  __shared__ uint32_t s_bi;
  constexpr uint32_t PAIRS = BLOCK / 2;
  const uint32_t pid = threadIdx.x >> 1;
  const int hi = lane_is_hi() ? 1 : 0;
  const uint32_t nbig = *big_count;
  if (nbig == 0) return;
  const uint32_t total = prefix[nbig];
  for (uint32_t sid = blockIdx.x; sid < total; sid += gridDim.x) {
    if (threadIdx.x == 0) {
      uint32_t lo = 0, up = nbig - 1;
      while (lo < up) {
        const uint32_t mid = (lo + up + 1) >> 1;
        if (prefix[mid] <= sid)
          lo = mid;
        else
          up = mid - 1;
      }
      s_bi = lo;
    }
    __syncthreads();
    const uint32_t bi = s_bi;
    const uint32_t g = big_list[bi];
    const size_t first = offsets[g], last = first + counts[g];
    const size_t begin = first + (size_t)(sid - prefix[bi]) * BIG_SLICE;
    const size_t end = begin + BIG_SLICE < last ? begin + BIG_SLICE : last;
    XYZZ<FL> acc, b;
    xyzz_set_inf<FL>(acc);
    for (size_t k = begin + pid; k < end; k += PAIRS) {  // pair-uniform bounds
      const uint32_t e = sorted[k];
      Affine<FL> q;
      lp_load_affine<C>(q, points, e & 0x7fffffffu, hi);
      xyzz_madd<FL>(acc, q, (e >> 31) != 0);
    }
    lp_store_xyzz<C>(sh, pid, acc, hi);
    __syncthreads();
    for (uint32_t s2 = PAIRS / 2; s2 > 0; s2 >>= 1) {
      if (pid < s2) {  // pair-uniform
        XYZZ<FL> a;
        lp_load_xyzz<C>(a, sh, pid, hi);
        lp_load_xyzz<C>(b, sh, pid + s2, hi);
        xyzz_add_lp_ool<C>(a, b);
        lp_store_xyzz<C>(sh, pid, a, hi);
      }
      __syncthreads();
    }
    if (pid == 0) {
      XYZZ<FL> a;
      lp_load_xyzz<C>(a, sh, 0, hi);
      lp_store_xyzz<C>(partials, sid, a, hi);
    }
    __syncthreads();
  }
}

// ---- G2 accumulation in the carry-free form over lane pairs (ec28_lp.h; curves with u^2 = -1) -----------------
template <class C>
struct alignas(8) AffineG2_28 {  // x.c0 | x.c1 | y.c0 | y.c1, 56 (40) bytes each
  Fp28<C> c[4];
};

template <class C>
__global__ void __launch_bounds__(256) k_points_to28_g2(const Affine<Fp2Field<C>>* __restrict__ points, size_t n,
                                                        AffineG2_28<C>* __restrict__ out) {
  // one coordinate component per thread: 4 threads per point
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 4 * n) return;
  const Fp<C>* src = reinterpret_cast<const Fp<C>*>(points);
  Fp28<C> v;
  fp28_from_fp<C>(v, src[t]);
  reinterpret_cast<Fp28<C>*>(out)[t] = v;
}

template <class C>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_accumulate28_lp(const AffineG2_28<C>* __restrict__ points,
                                                         const uint32_t* __restrict__ sorted,
                                                         const uint32_t* __restrict__ offsets,
                                                         const uint32_t* __restrict__ counts, size_t n_buckets,
                                                         const uint32_t* __restrict__ order, uint32_t big_threshold,
                                                         uint32_t* __restrict__ big_list, uint32_t* __restrict__ big_count,
                                                         XYZZ<Fp2Field<C>>* __restrict__ buckets) {
  typedef PairDevice<C> B;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t pair = t >> 1;  // both lanes of a pair share the bucket: every branch below is pair-uniform
  if (pair >= n_buckets) return;
  const int hi = (int)(threadIdx.x & 1u);
  const size_t g = order[pair];
  const uint32_t cnt = counts[g];
  if (cnt > big_threshold) {  // summed in slices by k_big_slices / k_accumulate_big (boundary form)
    if (!hi) {
      uint32_t pos = atomicAdd(big_count, 1u);
      big_list[pos] = (uint32_t)g;
    }
    return;
  }
  XYZZ28L<Fp28<C>> acc;
  bool inf = true;
  const size_t begin = offsets[g], end = begin + cnt;
  if (cnt != 0) {
    uint32_t e = sorted[begin];
    Affine28L<Fp28<C>> p, pn;
    p.x = points[e & 0x7fffffffu].c[hi];
    p.y = points[e & 0x7fffffffu].c[2 + hi];
    for (size_t k = begin; k < end; k++) {
      uint32_t en = e;
      pn = p;
      if (k + 1 < end) {  // prefetch the next index and point under this addition
        en = sorted[k + 1];
        pn.x = points[en & 0x7fffffffu].c[hi];
        pn.y = points[en & 0x7fffffffu].c[2 + hi];
      }
      xyzz28_lp_madd<C, B>(acc, inf, p, (e >> 31) != 0);
      e = en;
      p = pn;
    }
  }
  // back to the boundary form, one Fp2 component per lane
  XYZZ<Fp2LField<C>> r;
  if (inf) {
    xyzz_set_inf<Fp2LField<C>>(r);
  } else {
    fp28_to_fp<C>(r.x.v, acc.x);
    fp28_to_fp<C>(r.y.v, acc.y);
    fp28_to_fp<C>(r.zz.v, acc.zz);
    fp28_to_fp<C>(r.zzz.v, acc.zzz);
  }
  lp_store_xyzz<C>(buckets, g, r, hi);
}

// ---- segmented G2 accumulation (see k_accumulate28_seg): the state of bucket g is two XYZZ28L, one per lane ------
template <class C>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_accumulate28_lp_seg(
    const AffineG2_28<C>* __restrict__ points, const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ offsets,
    const uint32_t* __restrict__ counts, size_t n_buckets, const uint32_t* __restrict__ order, uint32_t big_threshold,
    uint32_t* __restrict__ big_list, uint32_t* __restrict__ big_count, XYZZ28L<Fp28<C>>* __restrict__ state, int flags,
    XYZZ<Fp2Field<C>>* __restrict__ buckets) {
  typedef PairDevice<C> B;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t pair = t >> 1;  // both lanes of a pair share the bucket: every branch below is pair-uniform
  if (pair >= n_buckets) return;
  const int hi = (int)(threadIdx.x & 1u);
  const size_t g = order[pair];
  const uint32_t cnt = counts[g];
  const bool first = (flags & MLHIP_SEG_FIRST) != 0, last = (flags & MLHIP_SEG_LAST) != 0;
  if (cnt > big_threshold) {
    if (!hi) {
      uint32_t pos = atomicAdd(big_count, 1u);
      big_list[pos] = (uint32_t)g;
    }
    return;
  }
  const bool to_boundary = last && !(flags & MLHIP_SEG_KEEP28);
  if (cnt == 0 && !first && !to_boundary) return;
  XYZZ28L<Fp28<C>> acc;
  bool inf = true;
  if (!first) {
    acc = state[2 * g + hi];
    const uint32_t z = fp28_all_zero<C>(acc.zz) ? 1u : 0u;
    inf = (z & pair_xchg_u32(z)) != 0;  // ZZ = 0 in Fp2: both components
  }
  const size_t begin = offsets[g], end = begin + cnt;
  if (cnt != 0) {
    uint32_t e = sorted[begin];
    Affine28L<Fp28<C>> p, pn;
    p.x = points[e & 0x7fffffffu].c[hi];
    p.y = points[e & 0x7fffffffu].c[2 + hi];
    for (size_t k = begin; k < end; k++) {
      uint32_t en = e;
      pn = p;
      if (k + 1 < end) {
        en = sorted[k + 1];
        pn.x = points[en & 0x7fffffffu].c[hi];
        pn.y = points[en & 0x7fffffffu].c[2 + hi];
      }
      xyzz28_lp_madd<C, B>(acc, inf, p, (e >> 31) != 0);
      e = en;
      p = pn;
    }
  }
  if (to_boundary) {
    XYZZ<Fp2LField<C>> r;
    if (inf) {
      xyzz_set_inf<Fp2LField<C>>(r);
    } else {
      fp28_to_fp<C>(r.x.v, acc.x);
      fp28_to_fp<C>(r.y.v, acc.y);
      fp28_to_fp<C>(r.zz.v, acc.zz);
      fp28_to_fp<C>(r.zzz.v, acc.zzz);
    }
    lp_store_xyzz<C>(buckets, g, r, hi);
  } else {
    if (inf) {
#pragma unroll
      for (int i = 0; i < C::N28; i++) acc.x.l[i] = acc.y.l[i] = acc.zz.l[i] = acc.zzz.l[i] = 0;
    }
    state[2 * g + hi] = acc;
  }
}

// ---- the same segment kernel with the pair split by coordinate (ec28_kc.h): lane A owns X and ZZ, lane B owns Y and ZZZ,
// every Fp2 product is a one-lane Karatsuba product.  State, points and buckets keep their layouts: a lane simply reads
// both components of its two coordinates (state[2 g] holds the real parts {x, y, zz, zzz}, state[2 g + 1] the imaginary
// parts) and the x or the y half of a point (56 + 56 contiguous bytes).  Bit-identical state to k_accumulate28_lp_seg.
template <class C>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_accumulate28_kc_seg(
    const AffineG2_28<C>* __restrict__ points, const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ offsets,
    const uint32_t* __restrict__ counts, size_t n_buckets, const uint32_t* __restrict__ order, uint32_t big_threshold,
    uint32_t* __restrict__ big_list, uint32_t* __restrict__ big_count, XYZZ28L<Fp28<C>>* __restrict__ state, int flags,
    XYZZ<Fp2Field<C>>* __restrict__ buckets) {
  typedef KcDevice<C> B;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t pair = t >> 1;  // both lanes of a pair share the bucket: every branch below is pair-uniform
  if (pair >= n_buckets) return;
  const int hi = (int)(threadIdx.x & 1u);  // 0: lane A (X, ZZ)   1: lane B (Y, ZZZ)
  const size_t g = order[pair];
  const uint32_t cnt = counts[g];
  const bool first = (flags & MLHIP_SEG_FIRST) != 0, last = (flags & MLHIP_SEG_LAST) != 0;
  if (cnt > big_threshold) {
    if (!hi) {
      uint32_t pos = atomicAdd(big_count, 1u);
      big_list[pos] = (uint32_t)g;
    }
    return;
  }
  const bool to_boundary = last && !(flags & MLHIP_SEG_KEEP28);
  if (cnt == 0 && !first && !to_boundary) return;
  Fp2x28<C> u, z;  // A: X, ZZ   B: Y, ZZZ
  bool inf = true;
  Fp28<C>* const st = reinterpret_cast<Fp28<C>*>(state + 2 * g);  // {x, y, zz, zzz} real parts, then the imaginary parts
  if (!first) {
    u.c0 = st[hi];
    u.c1 = st[4 + hi];
    z.c0 = st[2 + hi];
    z.c1 = st[6 + hi];
    const bool zf[1] = {fp28_all_zero<C>(z.c0) && fp28_all_zero<C>(z.c1)};
    inf = B::of_a(zf);  // ZZ = 0 in Fp2
  }
  const size_t begin = offsets[g], end = begin + cnt;
  if (cnt != 0) {
    uint32_t e = sorted[begin];
    Fp2x28<C> p, pn;
    p.c0 = points[e & 0x7fffffffu].c[2 * hi];
    p.c1 = points[e & 0x7fffffffu].c[2 * hi + 1];
    for (size_t k = begin; k < end; k++) {
      uint32_t en = e;
      pn = p;
      if (k + 1 < end) {  // prefetch the next index and point under this addition
        en = sorted[k + 1];
        pn.c0 = points[en & 0x7fffffffu].c[2 * hi];
        pn.c1 = points[en & 0x7fffffffu].c[2 * hi + 1];
      }
      xyzz28_kc_madd<C, B>(u, z, inf, p, (e >> 31) != 0);
      e = en;
      p = pn;
    }
  }
  if (to_boundary) {
    Fp<C>* const q = reinterpret_cast<Fp<C>*>(buckets + g);  // x.c0 x.c1 y.c0 y.c1 zz.c0 zz.c1 zzz.c0 zzz.c1
    Fp<C> r0, r1, r2, r3;
    if (inf) {  // as xyzz_set_inf: X = Y = 1, ZZ = ZZZ = 0
      fp_one<C>(r0);
      fp_zero<C>(r1);
      fp_zero<C>(r2);
      fp_zero<C>(r3);
    } else {
      fp28_to_fp<C>(r0, u.c0);
      fp28_to_fp<C>(r1, u.c1);
      fp28_to_fp<C>(r2, z.c0);
      fp28_to_fp<C>(r3, z.c1);
    }
    q[2 * hi] = r0;
    q[2 * hi + 1] = r1;
    q[4 + 2 * hi] = r2;
    q[5 + 2 * hi] = r3;
  } else {
    if (inf) {
#pragma unroll
      for (int i = 0; i < C::N28; i++) u.c0.l[i] = u.c1.l[i] = z.c0.l[i] = z.c1.l[i] = 0;
    }
    st[hi] = u.c0;
    st[4 + hi] = u.c1;
    st[2 + hi] = z.c0;
    st[6 + hi] = z.c1;
  }
}

template <class C, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_accumulate_big_seg_g2(const uint32_t* __restrict__ big_list,
                                                                 const uint32_t* __restrict__ big_count,
                                                                 const uint32_t* __restrict__ prefix,
                                                                 const XYZZ<Fp2Field<C>>* __restrict__ partials,
                                                                 XYZZ28L<Fp28<C>>* __restrict__ state, int flags,
                                                                 XYZZ<Fp2Field<C>>* __restrict__ buckets) {
  typedef Fp2Field<C> F;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  XYZZ<F>* sh = reinterpret_cast<XYZZ<F>*>(smem);
  const uint32_t nbig = *big_count;
  const bool first = (flags & MLHIP_SEG_FIRST) != 0;
  const bool last = (flags & MLHIP_SEG_LAST) != 0 && !(flags & MLHIP_SEG_KEEP28);
  for (uint32_t bi = blockIdx.x; bi < nbig; bi += gridDim.x) {
    const uint32_t g = big_list[bi];
    XYZZ<F> sum;
    big_bucket_total<F, BLOCK>(sum, sh, partials, prefix, bi);
    if (threadIdx.x == 0) {
      if (!first) {
        const XYZZ28L<Fp28<C>> lo = state[2 * g], up = state[2 * g + 1];
        if (!(fp28_all_zero<C>(lo.zz) && fp28_all_zero<C>(up.zz))) {
          XYZZ<F> prev;
          fp28_to_fp<C>(prev.x.c0, lo.x);
          fp28_to_fp<C>(prev.x.c1, up.x);
          fp28_to_fp<C>(prev.y.c0, lo.y);
          fp28_to_fp<C>(prev.y.c1, up.y);
          fp28_to_fp<C>(prev.zz.c0, lo.zz);
          fp28_to_fp<C>(prev.zz.c1, up.zz);
          fp28_to_fp<C>(prev.zzz.c0, lo.zzz);
          fp28_to_fp<C>(prev.zzz.c1, up.zzz);
          xyzz_add_ool<F>(sum, prev);
        }
      }
      if (last) {
        buckets[g] = sum;
      } else {
        XYZZ28L<Fp28<C>> lo, up;
        if (xyzz_is_inf<F>(sum)) {
#pragma unroll
          for (int i = 0; i < C::N28; i++) {
            lo.x.l[i] = lo.y.l[i] = lo.zz.l[i] = lo.zzz.l[i] = 0;
            up.x.l[i] = up.y.l[i] = up.zz.l[i] = up.zzz.l[i] = 0;
          }
        } else {
          fp28_from_fp<C>(lo.x, sum.x.c0);
          fp28_from_fp<C>(up.x, sum.x.c1);
          fp28_from_fp<C>(lo.y, sum.y.c0);
          fp28_from_fp<C>(up.y, sum.y.c1);
          fp28_from_fp<C>(lo.zz, sum.zz.c0);
          fp28_from_fp<C>(up.zz, sum.zz.c1);
          fp28_from_fp<C>(lo.zzz, sum.zzz.c0);
          fp28_from_fp<C>(up.zzz, sum.zzz.c1);
        }
        state[2 * g] = lo;
        state[2 * g + 1] = up;
      }
    }
    __syncthreads();
  }
}

template <class C>
__global__ void __launch_bounds__(256) k_chunks_lp(const XYZZ<Fp2Field<C>>* __restrict__ buckets, size_t n_chunks, int l_eff,
                                                   XYZZ<Fp2Field<C>>* __restrict__ A, XYZZ<Fp2Field<C>>* __restrict__ W0) {
  typedef Fp2LField<C> FL;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t g = t >> 1;
  if (g >= n_chunks) return;
  const int hi = lane_is_hi() ? 1 : 0;
  XYZZ<FL> acc, w0, b;
  xyzz_set_inf<FL>(acc);
  xyzz_set_inf<FL>(w0);
  for (int i = l_eff - 1; i >= 1; i--) {
    lp_load_xyzz<C>(b, buckets, g * (size_t)l_eff + i, hi);
    xyzz_add_lp_ool<C>(acc, b);
    xyzz_add_lp_ool<C>(w0, acc);
  }
  lp_load_xyzz<C>(b, buckets, g * (size_t)l_eff, hi);
  xyzz_add_lp_ool<C>(acc, b);
  lp_store_xyzz<C>(A, g, acc, hi);
  lp_store_xyzz<C>(W0, g, w0, hi);
}

// same selection scheme as k_masked_sums; BLOCK threads = BLOCK/2 lane pairs, LDS tree over pairs
template <class C, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_masked_sums_lp(const XYZZ<Fp2Field<C>>* __restrict__ A,
                                                          const XYZZ<Fp2Field<C>>* __restrict__ W0, uint32_t T, int nsel,
                                                          XYZZ<Fp2Field<C>>* __restrict__ out) {
  typedef Fp2LField<C> FL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  XYZZ<Fp2Field<C>>* sh = reinterpret_cast<XYZZ<Fp2Field<C>>*>(smem);
  constexpr uint32_t PAIRS = BLOCK / 2;
  const uint32_t pid = threadIdx.x >> 1;
  const int hi = lane_is_hi() ? 1 : 0;
  const uint32_t w = blockIdx.x / nsel;
  const int sel = blockIdx.x % nsel;
  const XYZZ<Fp2Field<C>>* src = (sel < 2 ? W0 : A) + (size_t)w * T;
  XYZZ<FL> acc, b;
  xyzz_set_inf<FL>(acc);
  if (sel < 4) {
    const uint32_t half = (T + 1) / 2;
    const uint32_t lo = (sel & 1) ? half : 0u, hi_t = (sel & 1) ? T : half;
    for (uint32_t t = lo + pid; t < hi_t; t += PAIRS) {
      lp_load_xyzz<C>(b, src, t, hi);
      xyzz_add_lp_ool<C>(acc, b);
    }
  } else {
    const int k = sel - 4;
    const uint32_t lowmask = (1u << k) - 1u;
    for (uint32_t j = pid; j < T / 2; j += PAIRS) {
      uint32_t t = ((j >> k) << (k + 1)) | (1u << k) | (j & lowmask);
      lp_load_xyzz<C>(b, src, t, hi);
      xyzz_add_lp_ool<C>(acc, b);
    }
  }
  // the upper half of the pairs hands its sums to the lower half first, so the tree needs PAIRS / 2 slots only (48 KB for
  // 512 threads: twice the pairs per block, half the strided additions per pair, one more tree level)
  if (pid >= PAIRS / 2) lp_store_xyzz<C>(sh, pid - PAIRS / 2, acc, hi);
  __syncthreads();
  if (pid < PAIRS / 2) {
    lp_load_xyzz<C>(b, sh, pid, hi);
    xyzz_add_lp_ool<C>(acc, b);
  }
  __syncthreads();
  if (pid < PAIRS / 2) lp_store_xyzz<C>(sh, pid, acc, hi);
  __syncthreads();
  for (uint32_t s = PAIRS / 4; s > 0; s >>= 1) {
    if (pid < s) {  // pair-uniform
      XYZZ<FL> a;
      lp_load_xyzz<C>(a, sh, pid, hi);
      lp_load_xyzz<C>(b, sh, pid + s, hi);
      xyzz_add_lp_ool<C>(a, b);
      lp_store_xyzz<C>(sh, pid, a, hi);
    }
    __syncthreads();
  }
  if (pid == 0) lp_store_xyzz<C>(out, blockIdx.x, [&] { XYZZ<FL> a; lp_load_xyzz<C>(a, sh, 0, hi); return a; }(), hi);
}

// ---- the G2 reduction on the carry-free bucket state (BLS12-381; ec28_lp.h: xyzz28_lp_add) -----------------------------
// As k_chunks_q28 / k_masked_sums_q28 for G1: the buckets come as the accumulation kernel keeps them (two XYZZ28L per
// bucket, one per lane; ZZ = 0 limbs in both for an empty bucket), the chunk sums stay in that form, the W x nsel sums that
// travel to the host leave in the boundary form.
template <class C>
__device__ __noinline__ void xyzz28_lp_add_ool(XYZZ28L<Fp28<C>>& acc, bool& inf, const XYZZ28L<Fp28<C>>& q, bool q_inf) {
  xyzz28_lp_add<C, PairDevice<C>>(acc, inf, q, q_inf);
}
template <class C>
__device__ __forceinline__ bool lp28_load_state(XYZZ28L<Fp28<C>>& r, const XYZZ28L<Fp28<C>>* src, size_t idx, int hi) {
  r = src[2 * idx + hi];
  return lp28_all_zero<C, PairDevice<C>>(r.zz);  // infinity
}
template <class C>
__device__ __forceinline__ void lp28_store_state(XYZZ28L<Fp28<C>>* dst, size_t idx, XYZZ28L<Fp28<C>> r, bool inf, int hi) {
  if (inf) {
#pragma unroll
    for (int i = 0; i < C::N28; i++) r.x.l[i] = r.y.l[i] = r.zz.l[i] = r.zzz.l[i] = 0;
  }
  dst[2 * idx + hi] = r;
}

template <class C>
__global__ void __launch_bounds__(256) k_chunks_lp28(const XYZZ28L<Fp28<C>>* __restrict__ buckets, size_t n_chunks, int l_eff,
                                                     XYZZ28L<Fp28<C>>* __restrict__ A, XYZZ28L<Fp28<C>>* __restrict__ W0) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t g = t >> 1;
  if (g >= n_chunks) return;
  const int hi = (int)(threadIdx.x & 1u);
  XYZZ28L<Fp28<C>> acc, w0, b;
  bool acc_inf = true, w0_inf = true;
  for (int i = l_eff - 1; i >= 1; i--) {
    const bool b_inf = lp28_load_state<C>(b, buckets, g * (size_t)l_eff + i, hi);
    xyzz28_lp_add_ool<C>(acc, acc_inf, b, b_inf);
    xyzz28_lp_add_ool<C>(w0, w0_inf, acc, acc_inf);
  }
  const bool b_inf = lp28_load_state<C>(b, buckets, g * (size_t)l_eff, hi);
  xyzz28_lp_add_ool<C>(acc, acc_inf, b, b_inf);
  lp28_store_state<C>(A, g, acc, acc_inf, hi);
  lp28_store_state<C>(W0, g, w0, w0_inf, hi);
}

template <class C, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_masked_sums_lp28(const XYZZ28L<Fp28<C>>* __restrict__ A,
                                                            const XYZZ28L<Fp28<C>>* __restrict__ W0, uint32_t T, int nsel,
                                                            XYZZ<Fp2Field<C>>* __restrict__ out) {
  typedef XYZZ28L<Fp28<C>> X;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  X* sh = reinterpret_cast<X*>(smem);  // PAIRS / 2 slots of two lane entries
  constexpr uint32_t PAIRS = BLOCK / 2;
  const uint32_t pid = threadIdx.x >> 1;
  const int hi = (int)(threadIdx.x & 1u);
  const uint32_t w = blockIdx.x / nsel;
  const int sel = blockIdx.x % nsel;
  const X* src = (sel < 2 ? W0 : A) + (size_t)2 * w * T;
  X acc, b;
  bool inf = true;
  if (sel < 4) {
    const uint32_t half = (T + 1) / 2;
    const uint32_t lo = (sel & 1) ? half : 0u, hi_t = (sel & 1) ? T : half;
    for (uint32_t t = lo + pid; t < hi_t; t += PAIRS) {
      const bool b_inf = lp28_load_state<C>(b, src, t, hi);
      xyzz28_lp_add_ool<C>(acc, inf, b, b_inf);
    }
  } else {
    const int k = sel - 4;
    const uint32_t lowmask = (1u << k) - 1u;
    for (uint32_t j = pid; j < T / 2; j += PAIRS) {
      const uint32_t t = ((j >> k) << (k + 1)) | (1u << k) | (j & lowmask);
      const bool b_inf = lp28_load_state<C>(b, src, t, hi);
      xyzz28_lp_add_ool<C>(acc, inf, b, b_inf);
    }
  }
  // the upper half of the pairs hands its sums to the lower half, then a tree over PAIRS / 2 slots
  if (pid >= PAIRS / 2) lp28_store_state<C>(sh, pid - PAIRS / 2, acc, inf, hi);
  __syncthreads();
  if (pid < PAIRS / 2) {
    const bool b_inf = lp28_load_state<C>(b, sh, pid, hi);
    xyzz28_lp_add_ool<C>(acc, inf, b, b_inf);
  }
  __syncthreads();
  if (pid < PAIRS / 2) lp28_store_state<C>(sh, pid, acc, inf, hi);
  __syncthreads();
  for (uint32_t s = PAIRS / 4; s > 0; s >>= 1) {
    if (pid < s) {  // pair-uniform
      const bool b_inf = lp28_load_state<C>(b, sh, pid + s, hi);
      xyzz28_lp_add_ool<C>(acc, inf, b, b_inf);
      lp28_store_state<C>(sh, pid, acc, inf, hi);
    }
    __syncthreads();
  }
  if (pid == 0) {
    XYZZ<Fp2LField<C>> r;
    if (inf) {
      xyzz_set_inf<Fp2LField<C>>(r);
    } else {
      fp28_to_fp<C>(r.x.v, acc.x);
      fp28_to_fp<C>(r.y.v, acc.y);
      fp28_to_fp<C>(r.zz.v, acc.zz);
      fp28_to_fp<C>(r.zzz.v, acc.zzz);
    }
    lp_store_xyzz<C>(out, blockIdx.x, r, hi);
  }
}

// ---- folded plans (msm_fold.h): the W bucket groups' sums combined into one window's, on lane pairs -----------------
// k_group_combine_q (msm_reduce.h) for G2: one block per output of the combined window, one lane PAIR per input slot
// (group g, half h; fold_combine_src), an LDS tree of lane-pair additions in the boundary form.  Launched with 4 W threads.
// Host tail of a 2^20-point G2 MSM over shifted-base tables: 0.30 ms (2 W jobs on the host pool, 3 W additions) -> one
// window's 19 doublings on the calling thread.
template <class C, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_group_combine_lp(const XYZZ<Fp2Field<C>>* __restrict__ in, int W, int nsel, int nb,
                                                            XYZZ<Fp2Field<C>>* __restrict__ out) {
  typedef Fp2LField<C> FL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  XYZZ<Fp2Field<C>>* sh = reinterpret_cast<XYZZ<Fp2Field<C>>*>(smem);
  const uint32_t pid = threadIdx.x >> 1;
  const int hi = lane_is_hi() ? 1 : 0;
  const int o = blockIdx.x;
  const uint32_t NP = 2u * (uint32_t)W;
  XYZZ<FL> acc, b;
  xyzz_set_inf<FL>(acc);
  if (pid < NP) {  // pair-uniform
    const int g = (int)(pid >> 1);
    const int src = fold_combine_src(o, g, (int)(pid & 1u), nb);
    if (src >= 0) lp_load_xyzz<C>(acc, in, (size_t)g * nsel + src, hi);
  }
  lp_store_xyzz<C>(sh, pid, acc, hi);
  __syncthreads();
#pragma unroll 1
  for (uint32_t s = NP / 2; s > 0; s >>= 1) {
    if (pid < s) {  // pair-uniform
      lp_load_xyzz<C>(b, sh, pid + s, hi);
      xyzz_add_lp_ool<C>(acc, b);
      lp_store_xyzz<C>(sh, pid, acc, hi);
    }
    __syncthreads();
  }
  if (pid == 0) lp_store_xyzz<C>(out, (size_t)o, acc, hi);
}

// k_big_slices_lp from the carry-free rows of a shifted-base table (AffineG2_28; see k_big_slices28, msm_ed.h): the pair's
// sum is accumulated with the lane-pair carry-free mixed addition and converted to the boundary form for the LDS tree.
template <class C, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_big_slices_lp28(const AffineG2_28<C>* __restrict__ points,
                                                           const uint32_t* __restrict__ sorted,
                                                           const uint32_t* __restrict__ offsets,
                                                           const uint32_t* __restrict__ counts,
                                                           const uint32_t* __restrict__ big_list,
                                                           const uint32_t* __restrict__ big_count,
                                                           const uint32_t* __restrict__ prefix,
                                                           XYZZ<Fp2Field<C>>* __restrict__ partials) {
  typedef Fp2LField<C> FL;
  typedef PairDevice<C> B;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  XYZZ<Fp2Field<C>>* sh = reinterpret_cast<XYZZ<Fp2Field<C>>*>(smem);
  __shared__ uint32_t s_bi;
  constexpr uint32_t PAIRS = BLOCK / 2;
  const uint32_t pid = threadIdx.x >> 1;
  const int hi = lane_is_hi() ? 1 : 0;
  const uint32_t nbig = *big_count;
  if (nbig == 0) return;
  const uint32_t total = prefix[nbig];
  for (uint32_t sid = blockIdx.x; sid < total; sid += gridDim.x) {
    if (threadIdx.x == 0) {
      uint32_t lo = 0, up = nbig - 1;
      while (lo < up) {
        const uint32_t mid = (lo + up + 1) >> 1;
        if (prefix[mid] <= sid)
          lo = mid;
        else
          up = mid - 1;
      }
      s_bi = lo;
    }
    __syncthreads();
    const uint32_t bi = s_bi;
    const uint32_t g = big_list[bi];
    const size_t first = offsets[g], last = first + counts[g];
    const size_t begin = first + (size_t)(sid - prefix[bi]) * BIG_SLICE;
    const size_t end = begin + BIG_SLICE < last ? begin + BIG_SLICE : last;
    XYZZ28L<Fp28<C>> a28;
    bool inf = true;
    for (size_t k = begin + pid; k < end; k += PAIRS) {  // pair-uniform bounds
      const uint32_t e = sorted[k];
      Affine28L<Fp28<C>> q;
      q.x = points[e & 0x7fffffffu].c[hi];
      q.y = points[e & 0x7fffffffu].c[2 + hi];
      xyzz28_lp_madd<C, B>(a28, inf, q, (e >> 31) != 0);
    }
    XYZZ<FL> acc, b;
    if (inf) {
      xyzz_set_inf<FL>(acc);
    } else {
      fp28_to_fp<C>(acc.x.v, a28.x);
      fp28_to_fp<C>(acc.y.v, a28.y);
      fp28_to_fp<C>(acc.zz.v, a28.zz);
      fp28_to_fp<C>(acc.zzz.v, a28.zzz);
    }
    lp_store_xyzz<C>(sh, pid, acc, hi);
    __syncthreads();
    for (uint32_t s2 = PAIRS / 2; s2 > 0; s2 >>= 1) {
      if (pid < s2) {  // pair-uniform
        XYZZ<FL> a;
        lp_load_xyzz<C>(a, sh, pid, hi);
        lp_load_xyzz<C>(b, sh, pid + s2, hi);
        xyzz_add_lp_ool<C>(a, b);
        lp_store_xyzz<C>(sh, pid, a, hi);
      }
      __syncthreads();
    }
    if (pid == 0) {
      XYZZ<FL> a;
      lp_load_xyzz<C>(a, sh, 0, hi);
      lp_store_xyzz<C>(partials, sid, a, hi);
    }
    __syncthreads();
  }
}

}  // namespace mlhip
