// msm_ed.h -- G1 bucket accumulation in extended twisted Edwards coordinates (ed28.h) for plans whose caller vouches for
// the prime-order subgroup (mlhip_msm_plan_assume_srs; curves with C::HAS_EDWARDS: BLS12-377).  Part of msm_kernels.h.
// The kernels mirror k_points_to28 / k_accumulate28_seg / k_accumulate_big_seg (msm_accumulate.h): same entry lists, same
// state buffer (an EdExt28 has the footprint of an XYZZ28).  The buckets STAY in extended Edwards coordinates: the quad-lane
// reduction kernels run their Edwards instantiation on them (msm_reduce.h: k_chunks_q28<C, true>, ed_quad28_add) and
// only the W x nsel window sums return to the Weierstrass curve, for the host tail.
#pragma once
// (included by msm_kernels.h after msm_accumulate.h)

namespace mlhip {

// Weierstrass affine (boundary form) -> halved Niels triples, four points per thread sharing one inversion
template <class C>
__global__ void __launch_bounds__(256) k_points_to_ed28(const Affine<FpField<C>>* __restrict__ points, size_t n,
                                                        EdNiels28<C>* __restrict__ out) {
  constexpr int K = 4;
  const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * K;
  if (i0 >= n) return;
  Affine<FpField<C>> in[K];
  Fp<C> xh[K], yh[K];
#pragma unroll
  for (int j = 0; j < K; j++) {
    if (i0 + j < n) {
      in[j] = points[i0 + j];
    } else {  // padding: the point at infinity
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

// one thread per bucket; state[g] is the bucket in extended coordinates, between segments and for the reduction
template <class C>
__global__ void __launch_bounds__(256) k_accumulate_ed28_seg(const EdNiels28<C>* __restrict__ points,
                                                             const uint32_t* __restrict__ sorted,
                                                             const uint32_t* __restrict__ offsets,
                                                             const uint32_t* __restrict__ counts, size_t n_buckets,
                                                             const uint32_t* __restrict__ order, uint32_t big_threshold,
                                                             uint32_t* __restrict__ big_list, uint32_t* __restrict__ big_count,
                                                             XYZZ28<C>* __restrict__ state, int flags) {
  static_assert(sizeof(EdExt28<C>) == sizeof(XYZZ28<C>), "the two bucket forms share the state buffer");
  size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n_buckets) return;
  const size_t g = order[tid];
  const uint32_t cnt = counts[g];
  const bool first = (flags & MLHIP_SEG_FIRST) != 0;
  if (cnt > big_threshold) {  // k_accumulate_big_seg_ed adds this segment's entries to the bucket's state
    uint32_t pos = atomicAdd(big_count, 1u);
    big_list[pos] = (uint32_t)g;
    return;
  }
  if (cnt == 0 && !first) return;
  EdExt28<C>* const st = reinterpret_cast<EdExt28<C>*>(state);
  EdExt28<C> acc;
  if (first)
    ed28_set_identity<C>(acc);
  else
    acc = st[g];
  const size_t begin = offsets[g], end = begin + cnt;
  if (cnt != 0) {
    uint32_t e = sorted[begin];
    EdNiels28<C> p = points[e & 0x7fffffffu];
    for (size_t k = begin; k < end; k++) {
      uint32_t en = e;
      EdNiels28<C> pn = p;
      if (k + 1 < end) {  // prefetch the next index and point under this addition
        en = sorted[k + 1];
        pn = points[en & 0x7fffffffu];
      }
      ed28_madd<C>(acc, p, (e >> 31) != 0);
      e = en;
      p = pn;
    }
  }
  st[g] = acc;
}

// the long buckets of a segment: their slice sums come from the Weierstrass long-bucket kernels on the original points
// (k_big_slices: boundary form); thread 0 adds the bucket's earlier state on the Weierstrass side and maps the total back
// to extended Edwards coordinates (two inversions; a handful of buckets per segment)
template <class C, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_accumulate_big_seg_ed(const uint32_t* __restrict__ big_list,
                                                                 const uint32_t* __restrict__ big_count,
                                                                 const uint32_t* __restrict__ prefix,
                                                                 const XYZZ<FpField<C>>* __restrict__ partials,
                                                                 XYZZ28<C>* __restrict__ state, int flags) {
  typedef FpField<C> F;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  XYZZ<F>* sh = reinterpret_cast<XYZZ<F>*>(smem);
  EdExt28<C>* const st = reinterpret_cast<EdExt28<C>*>(state);
  const uint32_t nbig = *big_count;
  const bool first = (flags & MLHIP_SEG_FIRST) != 0;
  for (uint32_t bi = blockIdx.x; bi < nbig; bi += gridDim.x) {
    const uint32_t g = big_list[bi];
    XYZZ<F> sum;
    big_bucket_total<F, BLOCK>(sum, sh, partials, prefix, bi);
    if (threadIdx.x == 0) {
      if (!first) {
        const EdExt28<C> e = st[g];
        XYZZ28<C> w;
        bool inf;
        ed28_to_xyzz28<C>(w, inf, e);
        XYZZ<F> prev;
        xyzz28_to<C>(prev, w, inf);
        xyzz_add_ool<F>(sum, prev);
      }
      Affine<F> a;
      xyzz_to_affine<F>(a, sum);  // (0, 0) for the point at infinity: ed28_from_affine maps it to the identity
      EdExt28<C> e;
      ed28_from_affine<C>(e, a);
      st[g] = e;
    }
    __syncthreads();
  }
}

// ---- the slice sums of long buckets from the CARRY-FREE rows (round 4: a shifted-base table keeps no boundary-form rows) --
// k_big_slices (msm_accumulate.h) with the per-thread loop of the accumulation kernels: Weierstrass rows (Affine28, xyzz28_madd)
// or, ED, Niels triples (ed28_madd); the thread's sum is converted to the boundary form for the LDS tree, so the partial sums
// have the format the combine kernels (k_accumulate_big_seg, k_accumulate_big_seg_ed) already read.
template <class C, int BLOCK, bool ED>
__global__ void __launch_bounds__(BLOCK) k_big_slices28(const void* __restrict__ rows, const uint32_t* __restrict__ sorted,
                                                        const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ counts,
                                                        const uint32_t* __restrict__ big_list, const uint32_t* __restrict__ big_count,
                                                        const uint32_t* __restrict__ prefix,
                                                        XYZZ<FpField<C>>* __restrict__ partials) {
  typedef FpField<C> F;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  XYZZ<F>* sh = reinterpret_cast<XYZZ<F>*>(smem);
  __shared__ uint32_t s_bi;
  const uint32_t nbig = *big_count;
  if (nbig == 0) return;
  const uint32_t total = prefix[nbig];
  for (uint32_t sid = blockIdx.x; sid < total; sid += gridDim.x) {
    if (threadIdx.x == 0) {  // the bucket this slice belongs to: last entry with prefix <= sid
      uint32_t lo = 0, hi = nbig - 1;
      while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (prefix[mid] <= sid)
          lo = mid;
        else
          hi = mid - 1;
      }
      s_bi = lo;
    }
    __syncthreads();
    const uint32_t bi = s_bi;
    const uint32_t g = big_list[bi];
    const size_t first = offsets[g], last = first + counts[g];
    const size_t begin = first + (size_t)(sid - prefix[bi]) * BIG_SLICE;
    const size_t end = begin + BIG_SLICE < last ? begin + BIG_SLICE : last;
    XYZZ28<C> w;
    bool inf = true;
    if constexpr (ED) {
      if constexpr (C::HAS_EDWARDS) {
        const EdNiels28<C>* pts = static_cast<const EdNiels28<C>*>(rows);
        EdExt28<C> acc;
        ed28_set_identity<C>(acc);
        for (size_t k = begin + threadIdx.x; k < end; k += BLOCK) {
          const uint32_t e = sorted[k];
          const EdNiels28<C> q = pts[e & 0x7fffffffu];
          ed28_madd<C>(acc, q, (e >> 31) != 0);
        }
        ed28_to_xyzz28<C>(w, inf, acc);
      }
    } else {
      const Affine28<C>* pts = static_cast<const Affine28<C>*>(rows);
      for (size_t k = begin + threadIdx.x; k < end; k += BLOCK) {
        const uint32_t e = sorted[k];
        const Affine28<C> q = pts[e & 0x7fffffffu];
        xyzz28_madd<C>(w, inf, q, (e >> 31) != 0);
      }
    }
    XYZZ<F> acc;
    xyzz28_to<C>(acc, w, inf);
    block_tree_sum_auto<F, BLOCK>(sh, acc);
    if (threadIdx.x == 0) partials[sid] = sh[0];
    __syncthreads();
  }
}

}  // namespace mlhip
