// msm_accumulate.h -- G1 bucket accumulation: boundary-form and carry-free kernels, the sliced sums of long buckets,
// and the segmented variants used when a host-buffer MSM is streamed over PCIe.  Part of msm_kernels.h.
#pragma once
// (included by msm_kernels.h after its common headers and constants)

namespace mlhip {

template <class F>
__global__ void __launch_bounds__(256) k_accumulate(const Affine<F>* __restrict__ points,
                                                    const uint32_t* __restrict__ sorted,
                                                    const uint32_t* __restrict__ offsets,
                                                    const uint32_t* __restrict__ counts, size_t n_buckets,
                                                    const uint32_t* __restrict__ order, uint32_t big_threshold,
                                                    uint32_t* __restrict__ big_list, uint32_t* __restrict__ big_count,
                                                    XYZZ<F>* __restrict__ buckets) {
  size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n_buckets) return;
  const size_t g = order[tid];  // buckets sorted by population: a wave's lanes run equally long loops
  uint32_t cnt = counts[g];
  if (cnt > big_threshold) {
    uint32_t pos = atomicAdd(big_count, 1u);
    big_list[pos] = (uint32_t)g;
    return;
  }
  XYZZ<F> acc;
  xyzz_set_inf<F>(acc);
  size_t begin = offsets[g];
  msm_accumulate_range<F>(acc, points, sorted, begin, begin + cnt, 1);
  buckets[g] = acc;
}

// ---- G1 accumulation in the carry-free 28-bit-limb form (fp28.h / ec28.h) -------------------------------------
// k_points_to28 rewrites the n input points once per MSM (2 products per point); k_accumulate28 is k_accumulate on
// that copy: ~14 % more mixed additions per second because a limb product is one v_mad_i64_i32 with no v_addc and
// field additions carry nothing.  Bucket sums are stored in the boundary form, so every later kernel is unchanged.
template <class C>
__global__ void __launch_bounds__(256) k_points_to28(const Affine<FpField<C>>* __restrict__ points, size_t n,
                                                     Affine28<C>* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Affine28<C> q;
  affine28_from<C>(q, points[i]);
  out[i] = q;
}

template <class C>
__global__ void __launch_bounds__(256) k_accumulate28(const Affine28<C>* __restrict__ points,
                                                      const uint32_t* __restrict__ sorted,
                                                      const uint32_t* __restrict__ offsets,
                                                      const uint32_t* __restrict__ counts, size_t n_buckets,
                                                      const uint32_t* __restrict__ order, uint32_t big_threshold,
                                                      uint32_t* __restrict__ big_list, uint32_t* __restrict__ big_count,
                                                      XYZZ<FpField<C>>* __restrict__ buckets) {
  size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n_buckets) return;
  const size_t g = order[tid];
  uint32_t cnt = counts[g];
  if (cnt > big_threshold) {  // summed in slices by k_big_slices / k_accumulate_big (boundary form)
    uint32_t pos = atomicAdd(big_count, 1u);
    big_list[pos] = (uint32_t)g;
    return;
  }
  XYZZ28<C> acc;
  bool inf = true;
  const size_t begin = offsets[g], end = begin + cnt;
  if (cnt != 0) {
    uint32_t e = sorted[begin];
    Affine28<C> p = points[e & 0x7fffffffu];
    for (size_t k = begin; k < end; k++) {
      uint32_t en = e;
      Affine28<C> pn = p;
      if (k + 1 < end) {  // prefetch the next index and point under this addition
        en = sorted[k + 1];
        pn = points[en & 0x7fffffffu];
      }
      xyzz28_madd<C>(acc, inf, p, (e >> 31) != 0);
      e = en;
      p = pn;
    }
  }
  XYZZ<FpField<C>> r;
  xyzz28_to<C>(r, acc, inf);
  buckets[g] = r;
}

// out-of-line group operations for kernels that use several of them (bounds the code size)
template <class F>
__device__ __noinline__ void xyzz_madd_ool(XYZZ<F>& acc, const Affine<F>& q) {
  xyzz_madd<F>(acc, q, false);
}
template <class F>
__device__ __noinline__ void xyzz_add_ool(XYZZ<F>& acc, const XYZZ<F>& q) {
  xyzz_add<F>(acc, q);
}
template <class F>
__device__ __noinline__ void xyzz_dbl_ool(XYZZ<F>& r, const XYZZ<F>& p) {
  xyzz_dbl<F>(r, p);
}

// LDS tree sum of one XYZZ per thread; result valid in sh[0] after return (all threads must call)
template <class F, int BLOCK>
__device__ void block_tree_sum(XYZZ<F>* sh, const XYZZ<F>& mine) {
  const int tid = threadIdx.x;
  sh[tid] = mine;
  __syncthreads();
  for (int s = BLOCK / 2; s > 0; s >>= 1) {
    if (tid < s) {
      XYZZ<F> a = sh[tid];
      xyzz_add_ool<F>(a, sh[tid + s]);
      sh[tid] = a;
    }
    __syncthreads();
  }
}

// one XYZZ per quad of lanes (ec_quad.h): lane q of a quad holds coordinate q (X, Y, ZZ, ZZZ)
template <class C>
__device__ __forceinline__ void quad_load(Fp<C>& v, const XYZZ<FpField<C>>* arr, size_t idx) {
  v = reinterpret_cast<const Fp<C>*>(arr + idx)[threadIdx.x & 3u];
}
template <class C>
__device__ __forceinline__ void quad_store(XYZZ<FpField<C>>* arr, size_t idx, const Fp<C>& v) {
  reinterpret_cast<Fp<C>*>(arr + idx)[threadIdx.x & 3u] = v;
}
template <class C>
__device__ __forceinline__ void quad_set_inf(Fp<C>& v) {  // (1, 1, 0, 0)
  Fp<C> one, zero;
  fp_one<C>(one);
  fp_zero<C>(zero);
  fp_select<C>(v, (threadIdx.x & 2u) != 0, zero, one);
}

// quad g of window w owns buckets [g L, (g+1) L): A = sum B_b, W0 = sum_i i B_{gL+i} (msm_chunk_body's order)

// The same tree for G1 with one point per QUAD of lanes: a group addition is 4 rounds of one field product per lane
// instead of 14 in a row, so the 8 dependent levels cost ~3x less (the tree is pure latency: one workgroup, one
// slice).  BLOCK / 4 quads first fold four entries each, then halve.
template <class C, int BLOCK>
__device__ void block_tree_sum_q(XYZZ<FpField<C>>* sh, const XYZZ<FpField<C>>& mine) {
  typedef QuadDevice<C> B;
  constexpr uint32_t NQ = BLOCK / 4;
  const uint32_t quad = threadIdx.x >> 2;
  sh[threadIdx.x] = mine;
  __syncthreads();
  Fp<C> acc, v;
  quad_load<C>(acc, sh, quad);
#pragma unroll 1
  for (uint32_t j = 1; j < 4; j++) {
    quad_load<C>(v, sh, quad + j * NQ);
    quad_xyzz_add<C, B>(acc, v);
  }
  __syncthreads();  // every entry has been read
  quad_store<C>(sh, quad, acc);
  __syncthreads();
#pragma unroll 1
  for (uint32_t s = NQ / 2; s > 0; s >>= 1) {
    if (quad < s) {  // quad-uniform
      quad_load<C>(v, sh, quad + s);
      quad_xyzz_add<C, B>(acc, v);
      quad_store<C>(sh, quad, acc);
    }
    __syncthreads();
  }
}

// the tree the long-bucket kernels use: quads for G1, one lane per point for G2
template <class F, int BLOCK>
__device__ __forceinline__ void block_tree_sum_auto(XYZZ<F>* sh, const XYZZ<F>& mine) {
  if constexpr (std::is_same<F, FpField<typename F::Curve>>::value)
    block_tree_sum_q<typename F::Curve, BLOCK>(sh, mine);
  else
    block_tree_sum<F, BLOCK>(sh, mine);
}


// ---- long buckets (skewed scalars: small values, equal values, plain sums of points) ---------------------------
// A bucket above the threshold is cut into slices of BIG_SLICE entries; k_big_slices sums every slice with one
// workgroup (so one bucket holding all n points still fills the GPU: 2^20 entries = 256 slices), the combine kernels
// (k_accumulate_big and its segment variants) add the slice sums of a bucket and store / fold the result.
constexpr uint32_t BIG_SLICE = 4096;

// prefix[i] = number of slices of the long buckets before entry i of big_list; prefix[nbig] = total
static __global__ void __launch_bounds__(1024) k_big_prefix(const uint32_t* __restrict__ counts,
                                                            const uint32_t* __restrict__ big_list,
                                                            const uint32_t* __restrict__ big_count,
                                                            uint32_t* __restrict__ prefix) {
  __shared__ uint32_t part[1024];
  __shared__ uint32_t base;
  const uint32_t nbig = *big_count, tid = threadIdx.x;
  if (tid == 0) base = 0;
  __syncthreads();
  for (uint32_t c0 = 0; c0 < nbig; c0 += 1024) {
    const uint32_t i = c0 + tid;
    const uint32_t v = i < nbig ? (counts[big_list[i]] + BIG_SLICE - 1) / BIG_SLICE : 0u;
    part[tid] = v;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
      const uint32_t t = tid >= d ? part[tid - d] : 0u;
      __syncthreads();
      part[tid] += t;
      __syncthreads();
    }
    if (i < nbig) prefix[i] = base + part[tid] - v;
    __syncthreads();
    if (tid == 1023) base += part[1023];
    __syncthreads();
  }
  if (tid == 0) prefix[nbig] = base;
}

template <class F, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_big_slices(const Affine<F>* __restrict__ points,
                                                      const uint32_t* __restrict__ sorted,
                                                      const uint32_t* __restrict__ offsets,
                                                      const uint32_t* __restrict__ counts,
                                                      const uint32_t* __restrict__ big_list,
                                                      const uint32_t* __restrict__ big_count,
                                                      const uint32_t* __restrict__ prefix, XYZZ<F>* __restrict__ partials) {
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
    XYZZ<F> acc;
    xyzz_set_inf<F>(acc);
    msm_accumulate_range<F>(acc, points, sorted, begin + threadIdx.x, end, BLOCK);
    block_tree_sum_auto<F, BLOCK>(sh, acc);
    if (threadIdx.x == 0) partials[sid] = sh[0];
    __syncthreads();
  }
}

// sum of the slice sums of long bucket number bi, valid on thread 0 (all threads must call)
template <class F, int BLOCK>
__device__ void big_bucket_total(XYZZ<F>& sum, XYZZ<F>* sh, const XYZZ<F>* __restrict__ partials,
                                 const uint32_t* __restrict__ prefix, uint32_t bi) {
  const uint32_t s0 = prefix[bi], s1 = prefix[bi + 1];
  if (s1 - s0 == 1) {  // the usual case: a bucket just above the threshold
    if (threadIdx.x == 0) sum = partials[s0];
    return;
  }
  XYZZ<F> acc;
  xyzz_set_inf<F>(acc);
  for (uint32_t k = s0 + threadIdx.x; k < s1; k += BLOCK) xyzz_add_ool<F>(acc, partials[k]);
  block_tree_sum_auto<F, BLOCK>(sh, acc);
  if (threadIdx.x == 0) sum = sh[0];
  __syncthreads();
}

template <class F, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_accumulate_big(const uint32_t* __restrict__ big_list,
                                                          const uint32_t* __restrict__ big_count,
                                                          const uint32_t* __restrict__ prefix,
                                                          const XYZZ<F>* __restrict__ partials,
                                                          XYZZ<F>* __restrict__ buckets) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  XYZZ<F>* sh = reinterpret_cast<XYZZ<F>*>(smem);
  const uint32_t nbig = *big_count;
  for (uint32_t bi = blockIdx.x; bi < nbig; bi += gridDim.x) {
    XYZZ<F> sum;
    big_bucket_total<F, BLOCK>(sum, sh, partials, prefix, bi);
    if (threadIdx.x == 0) buckets[big_list[bi]] = sum;
  }
}

// ---- segmented accumulation (host-buffer MSMs streamed over PCIe, plan_stream below) --------------------------
// The n pairs arrive in K segments; every segment is sorted by itself and added INTO the bucket sums of the segments
// before it, so the upload of segment s+1 runs under the kernels of segment s and the reduction runs once.  Between
// segments a bucket is kept as its raw carry-free accumulator (4 normalized coordinates; ZZ = 0 limbs <=> infinity),
// which makes the chain of additions identical to the unsegmented kernel's; the last segment writes the boundary form.
#define MLHIP_SEG_FIRST 1
#define MLHIP_SEG_LAST 2
#define MLHIP_SEG_KEEP28 4  // the last segment leaves the raw accumulators too: the reduction reads them (k_chunks_q28)

template <class C>
__global__ void __launch_bounds__(256) k_accumulate28_seg(const Affine28<C>* __restrict__ points,
                                                          const uint32_t* __restrict__ sorted,
                                                          const uint32_t* __restrict__ offsets,
                                                          const uint32_t* __restrict__ counts, size_t n_buckets,
                                                          const uint32_t* __restrict__ order, uint32_t big_threshold,
                                                          uint32_t* __restrict__ big_list, uint32_t* __restrict__ big_count,
                                                          XYZZ28<C>* __restrict__ state, int flags,
                                                          XYZZ<FpField<C>>* __restrict__ buckets) {
  size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n_buckets) return;
  const size_t g = order[tid];
  const uint32_t cnt = counts[g];
  const bool first = (flags & MLHIP_SEG_FIRST) != 0, last = (flags & MLHIP_SEG_LAST) != 0;
  const bool to_boundary = last && !(flags & MLHIP_SEG_KEEP28);
  if (cnt > big_threshold) {  // k_accumulate_big_seg adds this segment's entries to the bucket's state
    uint32_t pos = atomicAdd(big_count, 1u);
    big_list[pos] = (uint32_t)g;
    return;
  }
  if (cnt == 0 && !first && !to_boundary) return;  // nothing to add, nothing to convert
  XYZZ28<C> acc;
  bool inf = true;
  if (!first) {
    acc = state[g];
    inf = fp28_all_zero<C>(acc.zz);
  }
  const size_t begin = offsets[g], end = begin + cnt;
  if (cnt != 0) {
    uint32_t e = sorted[begin];
    Affine28<C> p = points[e & 0x7fffffffu];
    for (size_t k = begin; k < end; k++) {
      uint32_t en = e;
      Affine28<C> pn = p;
      if (k + 1 < end) {
        en = sorted[k + 1];
        pn = points[en & 0x7fffffffu];
      }
      xyzz28_madd<C>(acc, inf, p, (e >> 31) != 0);
      e = en;
      p = pn;
    }
  }
  if (to_boundary) {
    XYZZ<FpField<C>> r;
    xyzz28_to<C>(r, acc, inf);
    buckets[g] = r;
  } else {
    if (inf) {
#pragma unroll
      for (int i = 0; i < C::N28; i++) acc.x.l[i] = acc.y.l[i] = acc.zz.l[i] = acc.zzz.l[i] = 0;
    }
    state[g] = acc;
  }
}

// the long buckets of a segment: total of the slice sums (boundary form), then state <- state + total (thread 0)
template <class C, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_accumulate_big_seg(const uint32_t* __restrict__ big_list,
                                                              const uint32_t* __restrict__ big_count,
                                                              const uint32_t* __restrict__ prefix,
                                                              const XYZZ<FpField<C>>* __restrict__ partials,
                                                              XYZZ28<C>* __restrict__ state, int flags,
                                                              XYZZ<FpField<C>>* __restrict__ buckets) {
  typedef FpField<C> F;
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
        XYZZ28<C> s28 = state[g];
        XYZZ<F> prev;
        xyzz28_to<C>(prev, s28, fp28_all_zero<C>(s28.zz));
        xyzz_add_ool<F>(sum, prev);
      }
      if (last) {
        buckets[g] = sum;
      } else {
        XYZZ28<C> s28;
        if (xyzz_is_inf<F>(sum)) {
#pragma unroll
          for (int i = 0; i < C::N28; i++) s28.x.l[i] = s28.y.l[i] = s28.zz.l[i] = s28.zzz.l[i] = 0;
        } else {
          fp28_from_fp<C>(s28.x, sum.x);
          fp28_from_fp<C>(s28.y, sum.y);
          fp28_from_fp<C>(s28.zz, sum.zz);
          fp28_from_fp<C>(s28.zzz, sum.zzz);
        }
        state[g] = s28;
      }
    }
    __syncthreads();
  }
}

}  // namespace mlhip
