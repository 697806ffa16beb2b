// msm_reduce.h -- G1 bucket reduction: chunk sums and bit-masked sums, one point per lane and one point per quad of
// lanes (ec_quad.h).  Part of msm_kernels.h.
#pragma once
// (included by msm_kernels.h after its common headers and constants)

namespace mlhip {

template <class F>
__global__ void __launch_bounds__(256) k_chunks(const XYZZ<F>* __restrict__ buckets, size_t n_chunks, int l_eff,
                                                XYZZ<F>* __restrict__ A, XYZZ<F>* __restrict__ W0) {
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_chunks) return;
  msm_chunk_body<F>(g, buckets, A, W0, l_eff, [](XYZZ<F>& a, const XYZZ<F>& q) { xyzz_add_ool<F>(a, q); });
}

// block (w, sel): sel 0,1 -> the two halves of sum_t W0[w][t]; sel 2,3 -> the two halves of sum_t A[w][t];
// sel 4+k -> sum over t with bit k set of A[w][t].  Every block therefore sums T/2 elements (equal depth).
template <class F, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_masked_sums(const XYZZ<F>* __restrict__ A, const XYZZ<F>* __restrict__ W0,
                                                       uint32_t T, int nsel, XYZZ<F>* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  XYZZ<F>* sh = reinterpret_cast<XYZZ<F>*>(smem);
  const uint32_t w = blockIdx.x / nsel;
  const int sel = blockIdx.x % nsel;
  const XYZZ<F>* src = (sel < 2 ? W0 : A) + (size_t)w * T;
  XYZZ<F> acc;
  xyzz_set_inf<F>(acc);
  if (sel < 4) {
    const uint32_t half = (T + 1) / 2;
    const uint32_t lo = (sel & 1) ? half : 0u, hi = (sel & 1) ? T : half;
    for (uint32_t t = lo + threadIdx.x; t < hi; t += BLOCK) xyzz_add_ool<F>(acc, src[t]);
  } else {
    const int k = sel - 4;
    const uint32_t lowmask = (1u << k) - 1u;
    for (uint32_t j = threadIdx.x; j < T / 2; j += BLOCK) {
      uint32_t t = ((j >> k) << (k + 1)) | (1u << k) | (j & lowmask);
      xyzz_add_ool<F>(acc, src[t]);
    }
  }
  block_tree_sum<F, BLOCK>(sh, acc);
  if (threadIdx.x == 0) out[blockIdx.x] = sh[0];
}

// ---- the same two reduction levels with one point per QUAD of lanes (ec_quad.h): 3.5x shallower chains -------
template <class C>
__global__ void __launch_bounds__(256) k_chunks_q(const XYZZ<FpField<C>>* __restrict__ buckets, size_t n_chunks, int l_eff,
                                                  XYZZ<FpField<C>>* __restrict__ A, XYZZ<FpField<C>>* __restrict__ W0) {
  typedef QuadDevice<C> B;
  const size_t g = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  if (g >= n_chunks) return;  // quad-uniform
  const XYZZ<FpField<C>>* b = buckets + g * (size_t)l_eff;
  Fp<C> acc, w0, cur, x, y;
  quad_set_inf<C>(acc);
  quad_set_inf<C>(w0);
  quad_load<C>(cur, b, l_eff - 1);
  const int steps = 2 * (l_eff - 1) + 1;  // acc += b[i]; w0 += acc; ... ; acc += b[0]
#pragma unroll 1
  for (int s = 0; s < steps; s++) {
    const bool odd = (s & 1) != 0;
    const int i = l_eff - 1 - (s >> 1);
    fp_select<C>(x, odd, w0, acc);
    fp_select<C>(y, odd, acc, cur);
    if (!odd && i > 0) quad_load<C>(cur, b, i - 1);  // the next bucket arrives under this addition
    quad_xyzz_add<C, B>(x, y);
    fp_select<C>(w0, odd, x, w0);
    fp_select<C>(acc, odd, acc, x);
  }
  quad_store<C>(A, g, acc);
  quad_store<C>(W0, g, w0);
}

// same selections as k_masked_sums; BLOCK / 4 quads per block
template <class C, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_masked_sums_q(const XYZZ<FpField<C>>* __restrict__ A,
                                                         const XYZZ<FpField<C>>* __restrict__ W0, uint32_t T, int nsel,
                                                         XYZZ<FpField<C>>* __restrict__ out) {
  typedef QuadDevice<C> B;
  typedef XYZZ<FpField<C>> X;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  X* sh = reinterpret_cast<X*>(smem);
  constexpr uint32_t NQ = BLOCK / 4;
  const uint32_t quad = threadIdx.x >> 2;
  const uint32_t w = blockIdx.x / nsel;
  const int sel = blockIdx.x % nsel;
  const X* src = (sel < 2 ? W0 : A) + (size_t)w * T;
  Fp<C> acc, v;
  quad_set_inf<C>(acc);
  // element j of this block's list -> index t into src (plain halves or "bit k set")
  uint32_t count, lo = 0;
  int k = 0;
  if (sel < 4) {
    const uint32_t half = (T + 1) / 2;
    lo = (sel & 1) ? half : 0u;
    count = ((sel & 1) ? T : half) - lo;
  } else {
    k = sel - 4;
    count = T / 2;
  }
  const uint32_t lowmask = (1u << k) - 1u;
#pragma unroll 1
  for (uint32_t j = quad; j < count; j += NQ) {
    const uint32_t t = sel < 4 ? lo + j : (((j >> k) << (k + 1)) | (1u << k) | (j & lowmask));
    quad_load<C>(v, src, t);
    quad_xyzz_add<C, B>(acc, v);
  }
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
  if (quad == 0) quad_store<C>(out, blockIdx.x, acc);
}

// ---- the quad-lane reduction on the accumulation kernels' own bucket state (XYZZ28, ec_quad28.h) ------------------
// Buckets come as the carry-free accumulators (ZZ all-zero limbs = empty), the chunk sums stay in that form, and only
// the W x nsel sums that travel to the host are converted to the boundary form (one conversion per lane).
template <class C>
__device__ __forceinline__ void quad28_load(Fp28<C>& v, const XYZZ28<C>* arr, size_t idx) {
  v = reinterpret_cast<const Fp28<C>*>(arr + idx)[threadIdx.x & 3u];
}
template <class C>
__device__ __forceinline__ void quad28_store(XYZZ28<C>* arr, size_t idx, const Fp28<C>& v) {
  reinterpret_cast<Fp28<C>*>(arr + idx)[threadIdx.x & 3u] = v;
}

// Bucket accumulation with one bucket per QUAD of lanes, for MSMs too small to fill the chip with one lane per bucket:
// what such an MSM waits for is the chain of dependent additions of its fullest bucket, and on a quad an addition is four
// rounds of one product per lane (quad28_xyzz_add; the point enters as (x, +-y, 1, 1)) instead of ten products in a row.
// Leaves the carry-free bucket state the reduction reads (as k_accumulate28_seg with MLHIP_SEG_KEEP28).
template <class C>
__global__ void __launch_bounds__(256) k_accumulate_q28(const Affine28<C>* __restrict__ points,
                                                        const uint32_t* __restrict__ sorted,
                                                        const uint32_t* __restrict__ offsets,
                                                        const uint32_t* __restrict__ counts, size_t n_buckets,
                                                        uint32_t big_threshold, uint32_t* __restrict__ big_list,
                                                        uint32_t* __restrict__ big_count, XYZZ28<C>* __restrict__ state) {
  typedef QuadDevice28<C> B;
  const size_t g = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  if (g >= n_buckets) return;  // quad-uniform
  const unsigned lane = threadIdx.x & 3u;
  const uint32_t cnt = counts[g];
  if (cnt > big_threshold) {  // k_big_slices / k_accumulate_big_seg write this bucket's state
    if (lane == 0) {
      const uint32_t pos = atomicAdd(big_count, 1u);
      big_list[pos] = (uint32_t)g;
    }
    return;
  }
  Fp28<C> acc, one;
  fp28_zero<C>(acc);
  fp28_from_const<C>(one, C::ONE28);
  const size_t begin = offsets[g], end = begin + cnt;
#pragma unroll 1
  for (size_t k = begin; k < end; k++) {
    const uint32_t e = sorted[k];
    const Fp28<C>* q = reinterpret_cast<const Fp28<C>*>(points + (e & 0x7fffffffu));  // x | y
    Fp28<C> c = q[lane & 1u], n, b;
    // the point at infinity is (0, 0): skip it (quad-uniform: lanes 0 and 1 hold x and y)
    const unsigned z = B::quad_or((fp28_all_zero<C>(c) ? 1u : 0u) << lane);
    if ((z & 3u) == 3u) continue;
    fp28_neg<C>(n, c);
    fp28_select<C>(c, lane == 1 && (e >> 31) != 0, n, c);
    fp28_select<C>(b, lane >= 2, one, c);
    quad28_xyzz_add<C, B>(acc, b);
  }
  quad28_store<C>(state, g, acc);
}

// the empty sum of a quad: all-zero limbs (XYZZ: ZZ = 0), or the twisted Edwards identity (0 : 1 : 1 : 0) when the
// buckets arrive in extended Edwards coordinates (ED: a subgroup-trusted BLS12-377 plan, msm_ed.h) -- the same kernels
// then add with ed_quad28_add: three product rounds instead of four, no exceptional case
template <class C, bool ED>
__device__ __forceinline__ void quad28_empty(Fp28<C>& v) {
  fp28_zero<C>(v);
  if constexpr (ED) {
    Fp28<C> one;
    fp28_from_const<C>(one, C::ONE28);
    const unsigned q = threadIdx.x & 3u;
    fp28_select<C>(v, q == 1 || q == 2, one, v);
  }
}
template <class C, bool ED>
__device__ __forceinline__ void quad28_add_any(Fp28<C>& a, const Fp28<C>& b) {
  if constexpr (ED)
    ed_quad28_add<C, QuadDevice28<C>>(a, b);
  else
    quad28_xyzz_add<C, QuadDevice28<C>>(a, b);
}

// Per chunk of l_eff consecutive buckets: A = sum_i b[i] and W0 = sum_i i b[i] as a running sum from the top bucket down
// (acc += b[i]; w0 += acc).  The two additions of a bucket are written out (round 4; one loop body that selected its
// operands cost 56 selects per step: reduction 0.420 -> 0.410 ms at 2^20, profiles/r04_chunks_two_ab.txt).
template <class C, bool ED = false>
__global__ void __launch_bounds__(256) k_chunks_q28(const XYZZ28<C>* __restrict__ buckets, size_t n_chunks, int l_eff,
                                                    XYZZ28<C>* __restrict__ A, XYZZ28<C>* __restrict__ W0) {
  const size_t g = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  if (g >= n_chunks) return;  // quad-uniform
  const XYZZ28<C>* b = buckets + g * (size_t)l_eff;
  Fp28<C> acc, w0, cur;
  quad28_empty<C, ED>(acc);
  quad28_empty<C, ED>(w0);
  quad28_load<C>(cur, b, l_eff - 1);
#pragma unroll 1
  for (int i = l_eff - 1; i >= 0; i--) {
    Fp28<C> nxt = cur;
    if (i > 0) quad28_load<C>(nxt, b, i - 1);  // the next bucket arrives under these additions
    quad28_add_any<C, ED>(acc, cur);
    if (i > 0) quad28_add_any<C, ED>(w0, acc);
    cur = nxt;
  }
  quad28_store<C>(A, g, acc);
  quad28_store<C>(W0, g, w0);
}

// same selections as k_masked_sums; BLOCK / 4 quads per block; the result leaves in the boundary form
template <class C, int BLOCK, bool ED = false>
__global__ void __launch_bounds__(BLOCK) k_masked_sums_q28(const XYZZ28<C>* __restrict__ A, const XYZZ28<C>* __restrict__ W0,
                                                           uint32_t T, int nsel, XYZZ<FpField<C>>* __restrict__ out) {
  typedef QuadDevice28<C> B;
  typedef XYZZ28<C> X;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  X* sh = reinterpret_cast<X*>(smem);
  constexpr uint32_t NQ = BLOCK / 4;
  const uint32_t quad = threadIdx.x >> 2;
  const uint32_t w = blockIdx.x / nsel;
  const int sel = blockIdx.x % nsel;
  const X* src = (sel < 2 ? W0 : A) + (size_t)w * T;
  Fp28<C> acc, v;
  quad28_empty<C, ED>(acc);
  uint32_t count, lo = 0;
  int k = 0;
  if (sel < 4) {
    const uint32_t half = (T + 1) / 2;
    lo = (sel & 1) ? half : 0u;
    count = ((sel & 1) ? T : half) - lo;
  } else {
    k = sel - 4;
    count = T / 2;
  }
  const uint32_t lowmask = (1u << k) - 1u;
#pragma unroll 1
  for (uint32_t j = quad; j < count; j += NQ) {
    const uint32_t t = sel < 4 ? lo + j : (((j >> k) << (k + 1)) | (1u << k) | (j & lowmask));
    quad28_load<C>(v, src, t);
    quad28_add_any<C, ED>(acc, v);
  }
  quad28_store<C>(sh, quad, acc);
  __syncthreads();
#pragma unroll 1
  for (uint32_t s = NQ / 2; s > 0; s >>= 1) {
    if (quad < s) {  // quad-uniform
      quad28_load<C>(v, sh, quad + s);
      quad28_add_any<C, ED>(acc, v);
      quad28_store<C>(sh, quad, acc);
    }
    __syncthreads();
  }
  if (quad == 0) {
    if constexpr (ED) {
      // the W x nsel sums leave as Weierstrass points in the boundary form, like every other path's: each lane of the
      // quad rebuilds the point from the four coordinates and keeps its own
      XYZZ28<C> g4, w;
      B::gather(g4, acc);
      EdExt28<C> e;
      e.x = g4.x;
      e.y = g4.y;
      e.z = g4.zz;
      e.t = g4.zzz;
      bool inf;
      ed28_to_xyzz28<C>(w, inf, e);
      XYZZ<FpField<C>> r4;
      xyzz28_to<C>(r4, w, inf);
      const unsigned q = threadIdx.x & 3u;
      Fp<C> r, t2;
      fp_select<C>(t2, q == 0, r4.x, r4.y);
      fp_select<C>(r, q >= 2, r4.zz, t2);
      fp_select<C>(r, q == 3, r4.zzz, r);
      quad_store<C>(out, blockIdx.x, r);
    } else {
      // (an empty sum is ZZ = 0 limbs, which converts to the boundary form's ZZ = 0)
      Fp<C> r;
      fp28_to_fp<C>(r, acc);
      quad_store<C>(out, blockIdx.x, r);
    }
  }
}

// ---- folded plans (msm_fold.h): the W bucket groups' sums combined on the device --------------------------------
// The masked-sum kernels leave nsel sums per bucket group g (two halves of sum_t W0[g][t], two halves of sum_t A[g][t], nb
// bit-masked sums of A[g][.]).  For a folded plan the groups are consecutive ranges of ONE bucket set: with the global chunk
// index t' = g T + t the whole MSM is a single "window" of W T chunks,
//   total = sum W0 + sum A + L sum_k 2^k B_k,   B_k = sum of the A[t'] with bit k of t' set, k < nb + lg W,
// and every one of its 2 + nb + lg W sums is a sum of at most 2 W entries of `in`: B_k = sum_g in[g][4 + k] for k < nb, and
// the plain sums in[g][2] + in[g][3] of the groups g with bit k - nb set above.  One block per output sums its 2 W inputs in
// a quad-lane LDS tree; the host tail is then that of ONE window (nb + lg W + lg L doublings, no per-group work on the host
// pool): 0.083 -> 0.04 ms at c = 20.  out = [sum W0 | sum A | inf | inf | B_0 .. B_(nb + lgW - 1)]: the layout
// host_tail_window reads.  BLOCK = 8 W threads (W a power of two <= 32).
template <class C, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_group_combine_q(const XYZZ<FpField<C>>* __restrict__ in, int W, int nsel, int nb,
                                                           XYZZ<FpField<C>>* __restrict__ out) {
  typedef QuadDevice<C> B;
  typedef XYZZ<FpField<C>> X;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  X* sh = reinterpret_cast<X*>(smem);
  const uint32_t quad = threadIdx.x >> 2;  // input slot: group g = quad / 2, half h = quad & 1
  const int o = blockIdx.x;                // output index in the layout above
  const uint32_t NQ = 2u * (uint32_t)W;
  Fp<C> acc, v;
  quad_set_inf<C>(acc);
  if (quad < NQ) {  // quad-uniform
    const int g = (int)(quad >> 1);
    const int src = fold_combine_src(o, g, (int)(quad & 1u), nb);
    if (src >= 0) quad_load<C>(acc, in, (size_t)g * nsel + src);
  }
  quad_store<C>(sh, quad, acc);
  __syncthreads();
#pragma unroll 1
  for (uint32_t s = NQ / 2; s > 0; s >>= 1) {  // (launched with 8 W threads: one quad per input slot)
    if (quad < s) {  // quad-uniform
      quad_load<C>(v, sh, quad + s);
      quad_xyzz_add<C, B>(acc, v);
      quad_store<C>(sh, quad, acc);
    }
    __syncthreads();
  }
  if (quad == 0) quad_store<C>(out, (size_t)o, acc);
}

}  // namespace mlhip
