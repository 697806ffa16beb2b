// msm_kernels.h -- Pippenger MSM kernels and their launch sequence (gfx950).  Included by the per-curve
// translation units tu_msm_<curve>.hip.  Pipeline and data layout: DESIGN.md section 3.
//   k_coarse_hist / k_coarse_scatter / k_fine_sort   two-level LDS counting sort of the (window, bucket) keys
//                 (k_digits / k_scan / k_scatter: the global-atomic variant for n > 2^24, MLHIP_LEGACY_SORT=1)
//   k_order_*     buckets ordered by population, so that a wave's lanes run equally long loops
//   k_points_to28 / k_accumulate28      G1: points into the carry-free form (fp28.h), one thread per bucket:
//                 XYZZ += points[idx] (mixed additions, gathered reads); k_accumulate is the boundary-form variant
//   k_points_to28_g2 / k_accumulate28_lp / k_accumulate_lp   G2: two lanes per bucket (one Fp2 component each)
//   k_big_prefix / k_big_slices / k_accumulate_big  buckets longer than the threshold (skewed scalars): slices of 4096
//                 entries, one workgroup per slice (LDS tree), then the slice sums of each bucket
//   k_chunks_q / k_masked_sums_q   G1 bucket reduction, one point per quad of lanes (ec_quad.h): per 16 consecutive
//                 buckets sum and locally weighted sum, then per window the plain / bit-masked sums of the chunk sums
//                 (k_chunks / k_masked_sums: one lane per point; *_lp: G2 lane pairs)
//   host tail     Horner over <= W*c bit positions + one inversion (O(1) work, 64-bit limbs)
// Replaces gnark-crypto's MultiExp behind MultiScalarMul (reference
// driver/gurvy/bls12381/bls12-381.go:766-783, driver/gurvy/bn254.go:232-245, driver/gurvy/bls12-377.go:229-242).
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include <type_traits>

#include "ec28.h"
#include "ec28_lp.h"
#include "ec_quad.h"
#include "fp2_lanes.h"
#include "mlhip_internal.h"
#include "msm_body.h"

namespace mlhip {

constexpr uint32_t BIG_BUCKET_MIN = 256;  // a bucket is summed by a whole workgroup above max(this, 8 x the mean length)
constexpr int CHUNK_L = 8;            // buckets per level-1 reduction thread

// ------------------------------------------------------------------------------------ kernels
template <class C>
__global__ void __launch_bounds__(256) k_digits(const uint32_t* __restrict__ scalars, size_t n, int mont, int c, int W,
                                                uint32_t M, uint32_t* __restrict__ digits,
                                                uint32_t* __restrict__ counts) {
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    msm_digits_body<C>(i, n, scalars, mont != 0, c, W, digits);
    for (int w = 0; w < W; w++) {
      uint32_t d = digits[(size_t)w * n + i];
      if (d) atomicAdd(&counts[(size_t)w * M + (d >> 1) - 1], 1u);
    }
  }
}

// ---- exclusive scan of u32 counts: tile scan (1024 threads x 4) -> scan of tile sums -> add back
constexpr int SCAN_TILE = 4096;

static __global__ void __launch_bounds__(1024) k_scan_tile(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                            uint32_t* __restrict__ tile_sums, size_t total) {
  __shared__ uint32_t part[1024];
  const uint32_t tid = threadIdx.x;
  const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)tid * 4;
  uint32_t v[4];
#pragma unroll
  for (int k = 0; k < 4; k++) v[k] = base + k < total ? in[base + k] : 0u;
  const uint32_t s = v[0] + v[1] + v[2] + v[3];
  part[tid] = s;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {
    uint32_t x = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += x;
    __syncthreads();
  }
  uint32_t run = part[tid] - s;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (base + k < total) out[base + k] = run;
    run += v[k];
  }
  if (tid == 1023) tile_sums[blockIdx.x] = part[1023];
}

// one block: exclusive scan of the tile sums in place (n_tiles <= a few thousand)
static __global__ void __launch_bounds__(1024) k_scan_sums(uint32_t* __restrict__ tile_sums, size_t n_tiles) {
  __shared__ uint32_t part[1024];
  const uint32_t tid = threadIdx.x;
  size_t per = (n_tiles + 1023) / 1024;
  size_t lo = (size_t)tid * per, hi = lo + per;
  if (lo > n_tiles) lo = n_tiles;
  if (hi > n_tiles) hi = n_tiles;
  uint32_t s = 0;
  for (size_t k = lo; k < hi; k++) s += tile_sums[k];
  part[tid] = s;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {
    uint32_t x = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += x;
    __syncthreads();
  }
  uint32_t run = part[tid] - s;
  for (size_t k = lo; k < hi; k++) {
    uint32_t c = tile_sums[k];
    tile_sums[k] = run;
    run += c;
  }
}

static __global__ void __launch_bounds__(1024) k_scan_add(uint32_t* __restrict__ out, const uint32_t* __restrict__ tile_sums,
                                                           size_t total) {
  const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * 4;
  const uint32_t add = tile_sums[blockIdx.x];
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (base + k < total) out[base + k] += add;
}

static inline void launch_scan(const uint32_t* in, uint32_t* out, uint32_t* tile_sums, size_t total, hipStream_t st) {
  const size_t n_tiles = (total + SCAN_TILE - 1) / SCAN_TILE;
  k_scan_tile<<<dim3((unsigned)n_tiles), dim3(1024), 0, st>>>(in, out, tile_sums, total);
  k_scan_sums<<<dim3(1), dim3(1024), 0, st>>>(tile_sums, n_tiles);
  k_scan_add<<<dim3((unsigned)n_tiles), dim3(1024), 0, st>>>(out, tile_sums, total);
}

// ---- two-level LDS counting sort of the (window, bucket) keys -------------------------------------------
// Replaces k_digits + k_scatter (one global atomic per key, 2 x 16.7M at n = 2^20) by:
//   k_coarse_hist     per block of 1024 scalars: digits -> LDS histogram over NB coarse bins
//                     (bin = window * CB + bucket >> LOW), one global atomic per (block, bin)
//   scan of the NB coarse counts
//   k_coarse_scatter  same blocks: reserve a slice of every coarse bin per block (one global atomic per
//                     (block, bin)), rank inside the block with LDS atomics, write packed entries
//                     (fine bucket bits | sign | point index)
//   k_fine_sort       one block per coarse bin: LDS histogram of the 2^LOW fine buckets -> counts/offsets
//                     of the real buckets (coalesced), then LDS-ranked placement of the point indices
#ifndef MLHIP_SORT_TILE
#define MLHIP_SORT_TILE 1024
#endif
constexpr int SORT_TILE = MLHIP_SORT_TILE;  // scalars per block in the coarse passes

template <class C>
__global__ void __launch_bounds__(256) k_coarse_hist(const uint32_t* __restrict__ scalars, size_t n, int mont, int c, int W,
                                                     int low, uint32_t NB, uint32_t* __restrict__ coarse_count,
                                                     uint16_t* __restrict__ blockhist) {
  extern __shared__ uint32_t lds_u32[];
  uint32_t* hist = lds_u32;
  for (uint32_t b = threadIdx.x; b < NB; b += 256) hist[b] = 0;
  __syncthreads();
  const uint32_t cb_shift = (uint32_t)(c - 1 - low);  // coarse bins per window = 1 << cb_shift
  for (int k = 0; k < SORT_TILE / 256; k++) {
    size_t i = (size_t)blockIdx.x * SORT_TILE + (size_t)k * 256 + threadIdx.x;
    if (i >= n) break;
    // recompute the digit chain window by window (no per-lane array: keeps this in registers)
    uint32_t s[8];
    fr_canonical<C>(s, scalars + 8 * i, mont != 0);
    uint32_t carry = 0;
    const uint32_t half = 1u << (c - 1);
    for (int w = 0; w < W; w++) {
      int bit = w * c;
      uint32_t v = 0;
      if (bit < 256) {
        int word = bit >> 5, sh = bit & 31;
        uint64_t two = s[word];
        if (word + 1 < 8) two |= (uint64_t)s[word + 1] << 32;
        v = (uint32_t)((two >> sh) & ((1u << c) - 1));
      }
      v += carry;
      uint32_t mag;
      if (v > half) {
        mag = (1u << c) - v;
        carry = 1;
      } else {
        mag = v;
        carry = 0;
      }
      if (mag) atomicAdd(&hist[((uint32_t)w << cb_shift) + ((mag - 1) >> low)], 1u);
    }
  }
  __syncthreads();
  // the block's histogram also goes to memory (<= SORT_TILE per bin, one window each): k_coarse_scatter reloads it instead
  // of recomputing every digit a second time
  for (uint32_t b = threadIdx.x; b < NB; b += 256) {
    uint32_t h = hist[b];
    blockhist[(size_t)blockIdx.x * NB + b] = (uint16_t)h;
    if (h) atomicAdd(&coarse_count[b], h);
  }
}

template <class C>
__global__ void __launch_bounds__(256) k_coarse_scatter(const uint32_t* __restrict__ scalars, size_t n, int mont, int c, int W,
                                                        int low, int idx_bits, uint32_t NB,
                                                        const uint32_t* __restrict__ coarse_off,
                                                        uint32_t* __restrict__ coarse_cursor, uint32_t* __restrict__ tmp,
                                                        const uint16_t* __restrict__ blockhist) {
  extern __shared__ uint32_t lds_u32[];
  uint32_t* hist = lds_u32;       // per-block count, then running rank
  uint32_t* base = lds_u32 + NB;  // global position of this block's slice of each bin
  const uint32_t cb_shift = (uint32_t)(c - 1 - low);
  const uint32_t half = 1u << (c - 1);
  // pass 1: this block's counts, computed by k_coarse_hist
  for (uint32_t b = threadIdx.x; b < NB; b += 256) hist[b] = blockhist[(size_t)blockIdx.x * NB + b];
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < NB; b += 256) {
    uint32_t h = hist[b];
    base[b] = h ? coarse_off[b] + atomicAdd(&coarse_cursor[b], h) : 0u;
    hist[b] = 0;
  }
  __syncthreads();
  // pass 2: place
  const uint32_t low_mask = (1u << low) - 1u;
  for (int k = 0; k < SORT_TILE / 256; k++) {
    size_t i = (size_t)blockIdx.x * SORT_TILE + (size_t)k * 256 + threadIdx.x;
    if (i >= n) break;
    uint32_t s[8];
    fr_canonical<C>(s, scalars + 8 * i, mont != 0);
    uint32_t carry = 0;
    for (int w = 0; w < W; w++) {
      int bit = w * c;
      uint32_t v = 0;
      if (bit < 256) {
        int word = bit >> 5, sh = bit & 31;
        uint64_t two = s[word];
        if (word + 1 < 8) two |= (uint64_t)s[word + 1] << 32;
        v = (uint32_t)((two >> sh) & ((1u << c) - 1));
      }
      v += carry;
      uint32_t mag, neg;
      if (v > half) {
        mag = (1u << c) - v;
        neg = 1;
        carry = 1;
      } else {
        mag = v;
        neg = 0;
        carry = 0;
      }
      if (mag) {
        uint32_t bkt = mag - 1;
        uint32_t bin = ((uint32_t)w << cb_shift) + (bkt >> low);
        uint32_t pos = base[bin] + atomicAdd(&hist[bin], 1u);
        tmp[pos] = ((bkt & low_mask) << (idx_bits + 1)) | (neg << idx_bits) | (uint32_t)i;
      }
    }
  }
}

static __global__ void __launch_bounds__(256) k_fine_sort(const uint32_t* __restrict__ tmp, const uint32_t* __restrict__ coarse_off,
                                                   const uint32_t* __restrict__ coarse_count, int c, int low, int idx_bits,
                                                   uint32_t big_bin, uint32_t* __restrict__ counts,
                                                   uint32_t* __restrict__ offsets, uint32_t* __restrict__ sorted) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t fo[256];
  const uint32_t bin = blockIdx.x;
  const uint32_t F = 1u << low;
  const uint32_t begin = coarse_off[bin], cnt = coarse_count[bin];
  if (cnt > big_bin) return;  // sorted by several workgroups: k_bigbin_hist / k_bigbin_place
  const uint32_t idx_mask = (1u << idx_bits) - 1u;
  hist[threadIdx.x] = 0;
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < cnt; k += 1024) {  // four loads in flight per thread
    uint32_t e[4];
#pragma unroll
    for (int j = 0; j < 4; j++) e[j] = k + 256u * j < cnt ? tmp[begin + k + 256u * j] : 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (k + 256u * j < cnt) atomicAdd(&hist[e[j] >> (idx_bits + 1)], 1u);
  }
  __syncthreads();
  // exclusive scan of the F <= 256 fine counts
  uint32_t mine = threadIdx.x < F ? hist[threadIdx.x] : 0u;
  fo[threadIdx.x] = mine;
  __syncthreads();
  for (uint32_t off = 1; off < 256; off <<= 1) {
    uint32_t x = threadIdx.x >= off ? fo[threadIdx.x - off] : 0;
    __syncthreads();
    fo[threadIdx.x] += x;
    __syncthreads();
  }
  const uint32_t excl = fo[threadIdx.x] - mine;
  __syncthreads();
  fo[threadIdx.x] = excl;
  hist[threadIdx.x] = 0;  // becomes the running rank
  // real bucket id of (bin, fine): window-major layout g = w*M + (cb << low) + fine = bin << low + fine
  if (threadIdx.x < F) {
    size_t g = ((size_t)bin << low) + threadIdx.x;
    counts[g] = mine;
    offsets[g] = begin + excl;
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < cnt; k += 1024) {  // four loads, then four ranks, then four stores in flight
    uint32_t e[4], pos[4];
#pragma unroll
    for (int j = 0; j < 4; j++) e[j] = k + 256u * j < cnt ? tmp[begin + k + 256u * j] : 0u;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      pos[j] = 0xFFFFFFFFu;
      if (k + 256u * j < cnt) {
        uint32_t f = e[j] >> (idx_bits + 1);
        pos[j] = begin + fo[f] + atomicAdd(&hist[f], 1u);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (pos[j] != 0xFFFFFFFFu) sorted[pos[j]] = (e[j] & idx_mask) | (((e[j] >> idx_bits) & 1u) << 31);
  }
}

// ---- coarse bins far above the mean (skewed scalars put half of all entries into one bin) ------------------------
// k_fine_sort gives a bin to ONE workgroup; a bin of 2^20 entries then takes milliseconds.  Bins above `big_bin` are
// cut into slices of BIGBIN_SLICE entries: k_bigbin_hist counts the fine buckets per slice into the global bucket
// counts, k_bigbin_place ranks every slice inside the bucket ranges (one global atomic per slice and fine bucket,
// LDS ranks inside the slice).  Lanes of a wave that hold the same fine bucket -- the usual case in such a bin --
// share one LDS atomic.
constexpr uint32_t BIGBIN_SLICE = 16384;

static __global__ void __launch_bounds__(1024) k_bigbin_prefix(const uint32_t* __restrict__ coarse_count, uint32_t NB,
                                                               uint32_t big_bin, uint32_t* __restrict__ prefix) {
  __shared__ uint32_t part[1024];
  __shared__ uint32_t base;
  const uint32_t tid = threadIdx.x;
  if (tid == 0) base = 0;
  __syncthreads();
  for (uint32_t c0 = 0; c0 < NB; c0 += 1024) {
    const uint32_t i = c0 + tid;
    const uint32_t cnt = i < NB ? coarse_count[i] : 0u;
    const uint32_t v = cnt > big_bin ? (cnt + BIGBIN_SLICE - 1) / BIGBIN_SLICE : 0u;
    part[tid] = v;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
      const uint32_t t = tid >= d ? part[tid - d] : 0u;
      __syncthreads();
      part[tid] += t;
      __syncthreads();
    }
    if (i < NB) prefix[i] = base + part[tid] - v;
    __syncthreads();
    if (tid == 1023) base += part[1023];
    __syncthreads();
  }
  if (tid == 0) prefix[NB] = base;
}

// rank of this lane's entry among the entries of fine bucket f handled so far by the block (LDS counter cnt[f]);
// one atomic per wave when all active lanes hold the same f
__device__ __forceinline__ uint32_t lds_rank(uint32_t* cnt, uint32_t f) {
  const uint32_t f0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)f);
  const unsigned long long active = __ballot(1);
  const unsigned long long same = __ballot(f == f0);
  if (same == active) {
    const uint32_t lane = __lane_id();
    const uint32_t below = (uint32_t)__popcll(active & ((1ull << lane) - 1ull));
    uint32_t b = 0;
    if (below == 0) b = atomicAdd(&cnt[f0], (uint32_t)__popcll(active));
    b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
    return b + below;
  }
  return atomicAdd(&cnt[f], 1u);
}

// slice -> (bin, first entry, end) by binary search in the slice prefix; false when the slice id is past the end
__device__ __forceinline__ bool bigbin_slice(uint32_t sid, const uint32_t* __restrict__ prefix, uint32_t NB,
                                             const uint32_t* __restrict__ coarse_off,
                                             const uint32_t* __restrict__ coarse_count, uint32_t& bin, uint32_t& begin,
                                             uint32_t& end, uint32_t& bin_begin) {
  uint32_t lo = 0, hi = NB - 1;
  while (lo < hi) {  // last bin with prefix <= sid (bins without slices share their successor's prefix)
    const uint32_t mid = (lo + hi + 1) >> 1;
    if (prefix[mid] <= sid)
      lo = mid;
    else
      hi = mid - 1;
  }
  bin = lo;
  bin_begin = coarse_off[bin];
  const uint32_t cnt = coarse_count[bin];
  begin = bin_begin + (sid - prefix[bin]) * BIGBIN_SLICE;
  end = begin + BIGBIN_SLICE < bin_begin + cnt ? begin + BIGBIN_SLICE : bin_begin + cnt;
  return true;
}

static __global__ void __launch_bounds__(256) k_bigbin_hist(const uint32_t* __restrict__ tmp, const uint32_t* __restrict__ coarse_off,
                                                            const uint32_t* __restrict__ coarse_count,
                                                            const uint32_t* __restrict__ prefix, uint32_t NB, int low,
                                                            int idx_bits, uint32_t* __restrict__ counts) {
  __shared__ uint32_t hist[256];
  const uint32_t total = prefix[NB];
  for (uint32_t sid = blockIdx.x; sid < total; sid += gridDim.x) {
    uint32_t bin, begin, end, bin_begin;
    bigbin_slice(sid, prefix, NB, coarse_off, coarse_count, bin, begin, end, bin_begin);
    hist[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t k = begin + threadIdx.x; k < end; k += 256) (void)lds_rank(hist, tmp[k] >> (idx_bits + 1));
    __syncthreads();
    const uint32_t h = hist[threadIdx.x];
    if (threadIdx.x < (1u << low) && h) atomicAdd(&counts[((size_t)bin << low) + threadIdx.x], h);
    __syncthreads();
  }
}

static __global__ void __launch_bounds__(256) k_bigbin_place(const uint32_t* __restrict__ tmp, const uint32_t* __restrict__ coarse_off,
                                                             const uint32_t* __restrict__ coarse_count,
                                                             const uint32_t* __restrict__ prefix, uint32_t NB, int low,
                                                             int idx_bits, const uint32_t* __restrict__ counts,
                                                             uint32_t* __restrict__ cursor, uint32_t* __restrict__ offsets,
                                                             uint32_t* __restrict__ sorted) {
  __shared__ uint32_t hist[256];  // this slice's count per fine bucket, then the running rank
  __shared__ uint32_t fo[256];    // start of the bucket inside the bin, then this slice's reserved start
  const uint32_t total = prefix[NB];
  const uint32_t F = 1u << low;
  const uint32_t idx_mask = (1u << idx_bits) - 1u;
  for (uint32_t sid = blockIdx.x; sid < total; sid += gridDim.x) {
    uint32_t bin, begin, end, bin_begin;
    bigbin_slice(sid, prefix, NB, coarse_off, coarse_count, bin, begin, end, bin_begin);
    const size_t g = ((size_t)bin << low) + threadIdx.x;
    // exclusive scan of the bin's (complete) bucket counts: where each bucket starts
    const uint32_t mine = threadIdx.x < F ? counts[g] : 0u;
    fo[threadIdx.x] = mine;
    hist[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) {
      const uint32_t x = threadIdx.x >= off ? fo[threadIdx.x - off] : 0;
      __syncthreads();
      fo[threadIdx.x] += x;
      __syncthreads();
    }
    const uint32_t start = bin_begin + fo[threadIdx.x] - mine;
    if (threadIdx.x < F && sid == prefix[bin]) offsets[g] = start;  // the bin's first slice publishes the offsets
    // this slice's counts, then one reservation per fine bucket
    for (uint32_t k = begin + threadIdx.x; k < end; k += 256) (void)lds_rank(hist, tmp[k] >> (idx_bits + 1));
    __syncthreads();
    const uint32_t h = hist[threadIdx.x];
    fo[threadIdx.x] = (threadIdx.x < F && h) ? start + atomicAdd(&cursor[g], h) : 0u;
    hist[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t k = begin + threadIdx.x; k < end; k += 256) {
      const uint32_t e = tmp[k];
      const uint32_t f = e >> (idx_bits + 1);
      const uint32_t pos = fo[f] + lds_rank(hist, f);
      sorted[pos] = (e & idx_mask) | (((e >> idx_bits) & 1u) << 31);
    }
    __syncthreads();
  }
}

// ---- bucket ordering by population (largest first) so the 64 lanes of a wave own equally long buckets.
// Counting sort on key = 255 - min(count, 255) without global atomics: per-block LDS histogram written
// bin-major, scanned, then per-block placement with LDS cursors.
constexpr int ORDER_BINS = 256;

static __global__ void __launch_bounds__(256) k_order_hist(const uint32_t* __restrict__ counts, size_t n_buckets,
                                                            uint32_t* __restrict__ hist /* [ORDER_BINS][gridDim.x] */) {
  __shared__ uint32_t h[ORDER_BINS];
  h[threadIdx.x] = 0;
  __syncthreads();
  size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (g < n_buckets) {
    uint32_t c = counts[g];
    atomicAdd(&h[255u - (c < 255u ? c : 255u)], 1u);
  }
  __syncthreads();
  hist[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = h[threadIdx.x];
}

static __global__ void __launch_bounds__(256) k_order_place(const uint32_t* __restrict__ counts, size_t n_buckets,
                                                             const uint32_t* __restrict__ hist_scanned,
                                                             uint32_t* __restrict__ order) {
  __shared__ uint32_t cur[ORDER_BINS];
  cur[threadIdx.x] = hist_scanned[(size_t)threadIdx.x * gridDim.x + blockIdx.x];
  __syncthreads();
  size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (g < n_buckets) {
    uint32_t c = counts[g];
    uint32_t pos = atomicAdd(&cur[255u - (c < 255u ? c : 255u)], 1u);
    order[pos] = (uint32_t)g;
  }
}

static __global__ void __launch_bounds__(256) k_scatter(const uint32_t* __restrict__ digits, size_t n, int W, uint32_t M,
                                                 const uint32_t* __restrict__ offsets, uint32_t* __restrict__ cursor,
                                                 uint32_t* __restrict__ sorted) {
  size_t total = (size_t)W * n;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    uint32_t d = digits[idx];
    if (!d) continue;
    size_t w = idx / n;
    size_t i = idx - w * n;
    size_t g = w * M + (d >> 1) - 1;
    uint32_t pos = offsets[g] + atomicAdd(&cursor[g], 1u);
    sorted[pos] = (uint32_t)i | ((d & 1u) << 31);
  }
}

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
  if (cnt > big_threshold) {  // summed by a whole workgroup in k_accumulate_big (boundary form)
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
    block_tree_sum<F, BLOCK>(sh, acc);
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
  block_tree_sum<F, BLOCK>(sh, acc);
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
  if (cnt > big_threshold) {  // k_accumulate_big_seg adds this segment's entries to the bucket's state
    uint32_t pos = atomicAdd(big_count, 1u);
    big_list[pos] = (uint32_t)g;
    return;
  }
  if (cnt == 0 && !first && !last) return;  // nothing to add, nothing to convert
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
  if (last) {
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
  const bool first = (flags & MLHIP_SEG_FIRST) != 0, last = (flags & MLHIP_SEG_LAST) != 0;
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

// out[i] = [s_i] P_i: batched single-scalar multiplication (the reference's G1.Mul / G2.Mul,
// driver/gurvy/bls12381/bls12-381.go:238-247, :342-351; double-and-add shape of :920-932), one lane per
// product, 4-bit fixed windows: 15-entry table in scratch, 4 doublings + 1 addition per window.
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
  XYZZ<F> tab[15];
  xyzz_from_affine<F>(tab[0], P);
  for (int k = 1; k < 15; k++) {
    tab[k] = tab[k - 1];
    xyzz_madd_ool<F>(tab[k], P);
  }
  XYZZ<F> acc;
  xyzz_set_inf<F>(acc);
  bool started = false;
  constexpr bool kInline = std::is_same<F, FpField<C>>::value;  // G1: the running point stays in registers
#pragma unroll 1
  for (int w = 63; w >= 0; w--) {
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
    uint32_t nib = (s[w >> 3] >> ((w & 7) * 4)) & 15u;
    if (nib) {
      if constexpr (kInline) {
        const XYZZ<F> q = tab[nib - 1];
        xyzz_add<F>(acc, q);
      } else {
        xyzz_add_ool<F>(acc, tab[nib - 1]);
      }
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
// 2^20 G1 products in 10 ms instead of 70 ms, G2 in 42 ms instead of 326 ms (profiles/r01_perf_scalar_mul.txt).  G1 runs the additions in the carry-free form (ec28.h).
constexpr int FB_WINDOWS = 32, FB_ROW = 255;
constexpr size_t FIXED_BASE_MIN = (size_t)1 << 16;  // the table costs one double-and-add wave time (G1 ~4 ms, G2 ~14 ms)

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

template <class C, class F>
int scalar_mul_device(const void* d_points, size_t point_stride, const void* d_scalars, int mont, size_t n, void* d_out,
                      hipStream_t st) {
  if (n == 0) return 0;
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
    const size_t t28_bytes = kG1 ? kEntries * sizeof(Affine28<C>) : 0;
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
    k_scalar_mul<C, F><<<dim3((unsigned)((kEntries + 63) / 64)), dim3(64), 0, st>>>((const Affine<F>*)d_points, 0, tsc, -1,
                                                                                    kEntries, table);
    if constexpr (kG1) {
      Affine28<C>* t28 = (Affine28<C>*)(scratch + tab_bytes + sc_bytes);
      k_points_to28<C><<<dim3((unsigned)((kEntries + 255) / 256)), dim3(256), 0, st>>>(table, kEntries, t28);
      k_fixed_base_g1<C><<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st>>>(t28, (const uint32_t*)d_scalars, mont, n,
                                                                             (Affine<F>*)d_out);
    } else {
      k_fixed_base<C, F><<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st>>>(table, (const uint32_t*)d_scalars, mont, n,
                                                                             (Affine<F>*)d_out);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(fb.last, st));
    return 0;
  }
  k_scalar_mul<C, F><<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st>>>((const Affine<F>*)d_points, point_stride,
                                                                          (const uint32_t*)d_scalars, mont, n,
                                                                          (Affine<F>*)d_out);
  HIPCHK(hipGetLastError());
  return 0;
}

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
  if (cnt > big_threshold) {  // summed by a whole workgroup in k_accumulate_big (boundary form)
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
  if (cnt == 0 && !first && !last) return;
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
  if (last) {
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
  const bool first = (flags & MLHIP_SEG_FIRST) != 0, last = (flags & MLHIP_SEG_LAST) != 0;
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
  lp_store_xyzz<C>(sh, pid, acc, hi);
  __syncthreads();
  for (uint32_t s = PAIRS / 2; s > 0; s >>= 1) {
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

template <class F>
int plan_alloc(mlhip_msm_plan* p) {
  const size_t nbuckets = (size_t)p->W * p->M;
  p->pt_size = sizeof(Affine<F>);
  p->xyzz_size = sizeof(XYZZ<F>);
  HIPCHK(hipMalloc(&p->d_digits, (size_t)p->W * p->max_n * 4));
  HIPCHK(hipMalloc(&p->d_sorted, (size_t)p->W * p->max_n * 4));
  {
    // sort parameters: packed entry = fine bits | sign | index must fit 32 bits, coarse bins must fit LDS
    int idx_bits = 1;
    while (((size_t)1 << idx_bits) < p->max_n) idx_bits++;
    int low = p->c - 1 < 8 ? p->c - 1 : 8;
    if (low > 31 - idx_bits) low = 31 - idx_bits;
    const char* legacy = getenv("MLHIP_LEGACY_SORT");
    uint32_t nb = low >= 1 ? (uint32_t)p->W << (p->c - 1 - low) : 0;
    if (low < 1 || nb > 4096 || (legacy && legacy[0] == '1')) {
      p->sort_low = 0;
      p->sort_nb = 0;
    } else {
      p->sort_low = low;
      p->sort_nb = nb;
    }
    p->sort_idx_bits = idx_bits;
  }
  // zeroed every run: [counts | cursor | bigcount(4) | coarse_count | coarse_cursor]
  p->zero_bytes = (2 * nbuckets + 4 + 2 * (size_t)p->sort_nb) * 4;
  HIPCHK(hipMalloc(&p->d_zero, p->zero_bytes));
  p->d_counts = p->d_zero;
  p->d_cursor = p->d_zero + nbuckets;
  p->d_bigcount = p->d_zero + 2 * nbuckets;
  p->d_coarse_count = p->d_zero + 2 * nbuckets + 4;
  p->d_coarse_cursor = p->d_coarse_count + p->sort_nb;
  HIPCHK(hipMalloc(&p->d_coarse_off, ((size_t)p->sort_nb + 1) * 4));
  HIPCHK(hipMalloc(&p->d_binprefix, ((size_t)p->sort_nb + 2) * 4));
  if (p->sort_nb) {
    static_assert(SORT_TILE < 65536, "a block puts at most one entry per scalar into a coarse bin: the count fits 16 bits");
    const size_t blocks = (p->max_n + SORT_TILE - 1) / SORT_TILE;
    HIPCHK(hipMalloc(&p->d_blockhist, blocks * p->sort_nb * sizeof(uint16_t)));
  }
  HIPCHK(hipMalloc(&p->d_offsets, nbuckets * 4));
  HIPCHK(hipMalloc(&p->d_biglist, nbuckets * 4));
  {
    // long buckets: at most W n / BIG_BUCKET_MIN of them, and W n / BIG_SLICE + one more slice per bucket
    const size_t entries = (size_t)p->W * p->max_n;
    const size_t nbig_max = std::min(nbuckets, entries / BIG_BUCKET_MIN + 1);
    HIPCHK(hipMalloc(&p->d_bigprefix, (nbig_max + 2) * 4));
    HIPCHK(hipMalloc(&p->d_bigpart, (entries / BIG_SLICE + nbig_max + 2) * p->xyzz_size));
  }
  HIPCHK(hipMalloc(&p->d_order, nbuckets * 4));
  {
    const size_t nblk = (nbuckets + 255) / 256;
    const size_t hist_n = (size_t)ORDER_BINS * nblk;
    HIPCHK(hipMalloc(&p->d_hist, hist_n * 4));
    const size_t tiles = (std::max(nbuckets, hist_n) + SCAN_TILE - 1) / SCAN_TILE;
    HIPCHK(hipMalloc(&p->d_tilesums, (tiles + 1) * 4));
  }
  if (const char* e = getenv("MLHIP_RED_BLOCK")) {
    const int v = atoi(e);
    if (v == 64 || v == 128 || v == 256) p->red_block = v;
  }
  if (const char* e = getenv("MLHIP_ACC_BLOCK")) {
    const int v = atoi(e);
    if (v == 64 || v == 128 || v == 256) p->acc_block = v;
  }
  {
    const char* one_lane = getenv("MLHIP_REDUCE_ONE_LANE");  // =1: the one-point-per-lane reduction kernels
    p->reduce_one_lane = one_lane && one_lane[0] == '1';
  }
  if constexpr (std::is_same<F, FpField<typename F::Curve>>::value) {
    // G1 accumulation runs in the carry-free form (fp28.h): -24 % time for the 12-limb fields, -7 % for BN254.
    // MLHIP_ACC32=1 selects the boundary-form kernel (kept as the second implementation the tests compare with).
    const char* acc32 = getenv("MLHIP_ACC32");
    const bool want28 = !(acc32 && acc32[0] == '1');
    if (want28) HIPCHK(hipMalloc(&p->d_points28, p->max_n * sizeof(Affine28<typename F::Curve>)));
  }
  if constexpr (std::is_same<F, Fp2Field<typename F::Curve>>::value && F::Curve::BETA == -1 && F::Curve::N28 == 14) {
    // G2 in the carry-free form: BLS12-381 only (-14 % accumulation time); u^2 = -5 does not fit the weight budget
    // (BLS12-377) and the 10-limb BN254 form gains nothing over its 8 saturated limbs on lane pairs
    const char* acc32 = getenv("MLHIP_ACC32");
    if (!(acc32 && acc32[0] == '1')) HIPCHK(hipMalloc(&p->d_points28, p->max_n * sizeof(AffineG2_28<typename F::Curve>)));
  }
  HIPCHK(hipMalloc(&p->d_buckets, nbuckets * p->xyzz_size));
  HIPCHK(hipMalloc(&p->d_A, (size_t)p->W * p->T * p->xyzz_size));
  HIPCHK(hipMalloc(&p->d_W0, (size_t)p->W * p->T * p->xyzz_size));
  HIPCHK(hipMalloc(&p->d_out, (size_t)p->W * p->nsel * p->xyzz_size));
  HIPCHK(hipHostMalloc(&p->h_out, (size_t)p->W * p->nsel * p->xyzz_size, hipHostMallocDefault));
  for (int i = 0; i < 5; i++) HIPCHK(hipEventCreate(&p->ev[i]));
  HIPCHK(hipEventCreateWithFlags(&p->done, hipEventDisableTiming));
  if (p->d_points28) {
    HIPCHK(hipStreamCreateWithFlags(&p->aux, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
  }
  return 0;
}

// Horner over bit positions: total = sum_w 2^(cw) [ out[w][0..3] summed + L * sum_k 2^k out[w][4+k] ]
template <class F>
void host_tail(const mlhip_msm_plan* p, XYZZ<F>& total) {
  const XYZZ<F>* o = reinterpret_cast<const XYZZ<F>*>(p->h_out);
  const int npos = p->W * p->c;
  std::vector<XYZZ<F>> slot(npos);
  for (int i = 0; i < npos; i++) xyzz_set_inf<F>(slot[i]);
  for (int w = 0; w < p->W; w++) {
    XYZZ<F> s = o[w * p->nsel + 0];
    for (int h = 1; h < 4; h++) xyzz_add<F>(s, o[w * p->nsel + h]);
    slot[w * p->c] = s;
    for (int k = 0; k < p->nb; k++) slot[w * p->c + p->lgL + k] = o[w * p->nsel + 4 + k];
  }
  xyzz_set_inf<F>(total);
  bool started = false;
  for (int i = npos - 1; i >= 0; i--) {
    if (started) {
      XYZZ<F> d;
      xyzz_dbl<F>(d, total);
      total = d;
    }
    if (!xyzz_is_inf<F>(slot[i])) {
      xyzz_add<F>(total, slot[i]);
      started = true;
    }
  }
}

// slice sums of the long buckets listed by the accumulation kernel (nothing to do, two near-empty launches, when
// there are none)
template <class F, int BB>
void launch_big_slices(mlhip_msm_plan* p, const Affine<F>* d_points, hipStream_t st) {
  k_big_prefix<<<dim3(1), dim3(1024), 0, st>>>(p->d_counts, p->d_biglist, p->d_bigcount, p->d_bigprefix);
  k_big_slices<F, BB><<<dim3(1024), dim3(BB), BB * sizeof(XYZZ<F>), st>>>(d_points, p->d_sorted, p->d_offsets, p->d_counts,
                                                                        p->d_biglist, p->d_bigcount, p->d_bigprefix,
                                                                        (XYZZ<F>*)p->d_bigpart);
}

// digits -> entries sorted by (window, bucket) in d_sorted / d_offsets / d_counts, and the bucket order by population
template <class C>
int launch_sort(mlhip_msm_plan* p, const void* d_scalars, int mont, size_t n, hipStream_t st, bool prof) {
  const size_t nbuckets = (size_t)p->W * p->M;
  if (p->sort_low > 0) {
    // two-level LDS counting sort (no per-key global atomics)
    const unsigned blocks = (unsigned)((n + SORT_TILE - 1) / SORT_TILE);
    const uint32_t NB = p->sort_nb;
    k_coarse_hist<C><<<dim3(blocks), dim3(256), NB * 4, st>>>((const uint32_t*)d_scalars, n, mont, p->c, p->W, p->sort_low, NB,
                                                            p->d_coarse_count, p->d_blockhist);
    if (prof) HIPCHK(hipEventRecord(p->ev[1], st));
    launch_scan(p->d_coarse_count, p->d_coarse_off, p->d_tilesums, NB, st);
    k_coarse_scatter<C><<<dim3(blocks), dim3(256), NB * 8, st>>>((const uint32_t*)d_scalars, n, mont, p->c, p->W, p->sort_low,
                                                               p->sort_idx_bits, NB, p->d_coarse_off, p->d_coarse_cursor,
                                                               p->d_digits, p->d_blockhist);
    // bins more than 8x the mean (and at least 32768 entries) are sorted by many workgroups
    const uint32_t big_bin = (uint32_t)std::min<size_t>(std::max<size_t>(32768, 8 * ((size_t)p->W * n / NB)), 0x7fffffffu);
    k_fine_sort<<<dim3(NB), dim3(256), 0, st>>>(p->d_digits, p->d_coarse_off, p->d_coarse_count, p->c, p->sort_low,
                                               p->sort_idx_bits, big_bin, p->d_counts, p->d_offsets, p->d_sorted);
    k_bigbin_prefix<<<dim3(1), dim3(1024), 0, st>>>(p->d_coarse_count, NB, big_bin, p->d_binprefix);
    k_bigbin_hist<<<dim3(1024), dim3(256), 0, st>>>(p->d_digits, p->d_coarse_off, p->d_coarse_count, p->d_binprefix, NB,
                                                   p->sort_low, p->sort_idx_bits, p->d_counts);
    k_bigbin_place<<<dim3(1024), dim3(256), 0, st>>>(p->d_digits, p->d_coarse_off, p->d_coarse_count, p->d_binprefix, NB,
                                                    p->sort_low, p->sort_idx_bits, p->d_counts, p->d_cursor, p->d_offsets,
                                                    p->d_sorted);
  } else {
    // legacy path (very large n or MLHIP_LEGACY_SORT=1): digits array + global-atomic histogram / scatter
    {
      size_t blocks = (n + 255) / 256;
      if (blocks > 65536) blocks = 65536;
      k_digits<C><<<dim3((unsigned)blocks), dim3(256), 0, st>>>((const uint32_t*)d_scalars, n, mont, p->c, p->W, p->M,
                                                                 p->d_digits, p->d_counts);
    }
    if (prof) HIPCHK(hipEventRecord(p->ev[1], st));
    launch_scan(p->d_counts, p->d_offsets, p->d_tilesums, nbuckets, st);
    {
      size_t total_e = (size_t)p->W * n;
      size_t blocks = (total_e + 255) / 256;
      if (blocks > 262144) blocks = 262144;
      k_scatter<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(p->d_digits, n, p->W, p->M, p->d_offsets, p->d_cursor,
                                                               p->d_sorted);
    }
  }
  {
    const unsigned nblk = (unsigned)((nbuckets + 255) / 256);
    k_order_hist<<<dim3(nblk), dim3(256), 0, st>>>(p->d_counts, nbuckets, p->d_hist);
    launch_scan(p->d_hist, p->d_hist, p->d_tilesums, (size_t)ORDER_BINS * nblk, st);
    k_order_place<<<dim3(nblk), dim3(256), 0, st>>>(p->d_counts, nbuckets, p->d_hist, p->d_order);
  }
  return 0;
}

// bucket sums in d_buckets -> W x nsel partial sums in d_out (two levels: chunks of L buckets, bit-masked sums)
template <class C, class F>
int launch_reduce(mlhip_msm_plan* p, hipStream_t st) {
  typedef XYZZ<F> X;
  constexpr bool kLanePairs = std::is_same<F, Fp2Field<C>>::value;  // G2: two lanes per bucket
  {
    size_t n_chunks = (size_t)p->W * p->T;
    if constexpr (kLanePairs) {
      k_chunks_lp<C><<<dim3((unsigned)((2 * n_chunks + 255) / 256)), dim3(256), 0, st>>>((const X*)p->d_buckets, n_chunks,
                                                                                        p->L, (X*)p->d_A, (X*)p->d_W0);
      constexpr int RB = 256;  // 128 lane pairs x 384 B = 48 KB of LDS per block
      k_masked_sums_lp<C, RB><<<dim3((unsigned)(p->W * p->nsel)), dim3(RB), (RB / 2) * sizeof(X), st>>>(
          (const X*)p->d_A, (const X*)p->d_W0, p->T, p->nsel, (X*)p->d_out);
    } else {
      if (p->reduce_one_lane) {  // MLHIP_REDUCE_ONE_LANE=1 when the plan was created
        k_chunks<F><<<dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, st>>>((const X*)p->d_buckets, n_chunks,
                                                                                     p->L, (X*)p->d_A, (X*)p->d_W0);
        constexpr int RB = 256;  // 48 KB of LDS per block
        k_masked_sums<F, RB><<<dim3((unsigned)(p->W * p->nsel)), dim3(RB), RB * sizeof(X), st>>>(
            (const X*)p->d_A, (const X*)p->d_W0, p->T, p->nsel, (X*)p->d_out);
      } else {
        // one point per quad of lanes: a group addition is 4 rounds of one multiplication instead of 14 in a row
        k_chunks_q<C><<<dim3((unsigned)((4 * n_chunks + p->red_block - 1) / p->red_block)), dim3(p->red_block), 0, st>>>((const X*)p->d_buckets, n_chunks,
                                                                                         p->L, (X*)p->d_A, (X*)p->d_W0);
        constexpr int RB = 512;  // 128 quads, 24 KB of LDS per block
        k_masked_sums_q<C, RB><<<dim3((unsigned)(p->W * p->nsel)), dim3(RB), (RB / 4) * sizeof(X), st>>>(
            (const X*)p->d_A, (const X*)p->d_W0, p->T, p->nsel, (X*)p->d_out);
      }
    }
  }
  return 0;
}

template <class C, class F>
int plan_launch(mlhip_msm_plan* p, const void* d_points, const void* d_scalars, int mont, size_t n, hipStream_t st) {
  typedef Affine<F> A;
  typedef XYZZ<F> X;
  p->pending_n = n;
  p->pending = true;
  if (n != 0) {
    const size_t nbuckets = (size_t)p->W * p->M;
    const bool prof = p->profiling;
    // Buckets far longer than the mean (degenerate inputs: equal scalars, tiny scalars) are handed to a whole
    // workgroup each; the threshold scales with the mean length n / 2^(c-1) so that large n, and the sparser top
    // window (2-4x the mean for these group orders), stay on the one-thread-per-bucket path.
    uint32_t big_threshold = (uint32_t)std::min<size_t>((n >> (p->c - 1)) * 8, 1u << 30);
    if (big_threshold < BIG_BUCKET_MIN) big_threshold = BIG_BUCKET_MIN;
    if (p->upload_src && !p->d_points28) {  // no auxiliary stream on this path: plain upload first
      HIPCHK(hipMemcpy(const_cast<void*>(d_points), p->upload_src, p->upload_bytes, hipMemcpyHostToDevice));
      p->upload_src = nullptr;
    }
    // the conversion of the points is independent of the sort: it runs on the auxiliary stream beside the
    // (LDS-atomic bound) sort kernels.  The fork is recorded now (after the previous MSM's work on `st`); the work
    // itself is queued after the sort launches so that a host-blocking upload cannot delay them.
    if (p->d_points28) HIPCHK(hipEventRecord(p->ev_fork, st));
    HIPCHK(hipMemsetAsync(p->d_zero, 0, p->zero_bytes, st));
    if (prof) HIPCHK(hipEventRecord(p->ev[0], st));
    {
      int rc_sort = launch_sort<C>(p, d_scalars, mont, n, st, prof);
      if (rc_sort) return rc_sort;
    }
    // resident bases: the carry-free copy of the first conv_n points of this very buffer is already there
    const bool conv_cached = p->points_static && p->conv_src == d_points && n <= p->conv_n && !p->upload_src;
    if (p->d_points28 && conv_cached) {
      HIPCHK(hipEventRecord(p->ev_join, st));  // nothing to wait for
    } else if (p->d_points28) {
      HIPCHK(hipStreamWaitEvent(p->aux, p->ev_fork, 0));
      if (p->upload_src) {  // host-buffer call: the upload of the points rides the same stream, ahead of the conversion
        HIPCHK(hipMemcpyAsync(const_cast<void*>(d_points), p->upload_src, p->upload_bytes, hipMemcpyHostToDevice, p->aux));
        p->upload_src = nullptr;
      }
      if constexpr (std::is_same<F, Fp2Field<C>>::value) {
        if constexpr (C::BETA == -1)
          k_points_to28_g2<C><<<dim3((unsigned)((4 * n + 255) / 256)), dim3(256), 0, p->aux>>>(
              (const A*)d_points, n, (AffineG2_28<C>*)p->d_points28);
      } else {
        k_points_to28<C><<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p->aux>>>((const A*)d_points, n,
                                                                                     (Affine28<C>*)p->d_points28);
      }
      HIPCHK(hipEventRecord(p->ev_join, p->aux));
      p->conv_src = d_points;
      p->conv_n = n;
    }
    if (prof) HIPCHK(hipEventRecord(p->ev[2], st));
    constexpr bool kLanePairs = std::is_same<F, Fp2Field<C>>::value;  // G2: two lanes per bucket
    if constexpr (kLanePairs) {
      bool done28 = false;
      if constexpr (C::BETA == -1) {
        if (p->d_points28) {
          HIPCHK(hipStreamWaitEvent(st, p->ev_join, 0));
          k_accumulate28_lp<C><<<dim3((unsigned)((2 * nbuckets + p->acc_block - 1) / p->acc_block)), dim3(p->acc_block), 0, st>>>(
              (const AffineG2_28<C>*)p->d_points28, p->d_sorted, p->d_offsets, p->d_counts, nbuckets, p->d_order,
              big_threshold, p->d_biglist, p->d_bigcount, (X*)p->d_buckets);
          done28 = true;
        }
      }
      if (!done28)
        k_accumulate_lp<C><<<dim3((unsigned)((2 * nbuckets + p->acc_block - 1) / p->acc_block)), dim3(p->acc_block), 0, st>>>(
            (const A*)d_points, p->d_sorted, p->d_offsets, p->d_counts, nbuckets, p->d_order, big_threshold, p->d_biglist,
            p->d_bigcount, (X*)p->d_buckets);
    } else if (p->d_points28) {
      HIPCHK(hipStreamWaitEvent(st, p->ev_join, 0));
      k_accumulate28<C><<<dim3((unsigned)((nbuckets + p->acc_block - 1) / p->acc_block)), dim3(p->acc_block), 0, st>>>(
          (const Affine28<C>*)p->d_points28, p->d_sorted, p->d_offsets, p->d_counts, nbuckets, p->d_order, big_threshold,
          p->d_biglist, p->d_bigcount, (X*)p->d_buckets);
    } else {
      k_accumulate<F><<<dim3((unsigned)((nbuckets + p->acc_block - 1) / p->acc_block)), dim3(p->acc_block), 0, st>>>(
          (const A*)d_points, p->d_sorted, p->d_offsets, p->d_counts, nbuckets, p->d_order, big_threshold, p->d_biglist,
          p->d_bigcount, (X*)p->d_buckets);
    }
    if (prof) HIPCHK(hipEventRecord(p->ev[3], st));
    {
      constexpr int BB = sizeof(X) <= 192 ? 256 : 128;  // 48 KB of LDS per block
      launch_big_slices<F, BB>(p, (const A*)d_points, st);
      k_accumulate_big<F, BB><<<dim3(256), dim3(BB), BB * sizeof(X), st>>>(p->d_biglist, p->d_bigcount, p->d_bigprefix,
                                                                            (const X*)p->d_bigpart, (X*)p->d_buckets);
    }
    {
      int rc_red = launch_reduce<C, F>(p, st);
      if (rc_red) return rc_red;
    }
    if (prof) HIPCHK(hipEventRecord(p->ev[4], st));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(p->h_out, p->d_out, (size_t)p->W * p->nsel * sizeof(X), hipMemcpyDeviceToHost, st));
    HIPCHK(hipEventRecord(p->done, st));
  }
  return 0;
}

// Host-buffer G1 MSM streamed in K segments (see k_accumulate28_seg): h_points / h_scalars are the caller's pageable
// buffers, d_points / d_scalars the plan-sized device buffers they are staged through.  Uploads and the point
// conversion ride the auxiliary stream; the sort and the accumulation of segment s wait for its event on `st`.
// The host thread blocks inside the pageable copies, which is exactly what overlaps them with the kernels queued before.
template <class C, class F>
int plan_stream(mlhip_msm_plan* p, void* d_points, void* d_scalars, const void* h_points, const void* h_scalars, int mont,
                size_t n, int K, hipStream_t st) {
  typedef Affine<F> A;
  typedef XYZZ<F> X;
  constexpr bool kG2 = std::is_same<F, Fp2Field<C>>::value;
  if (!p->d_points28 || !p->aux) return mlhip_rt::fail(MLHIP_EINVAL, "streamed MSM needs the carry-free accumulation path");
  constexpr size_t kStateBytes = kG2 ? 2 * sizeof(XYZZ28L<Fp28<C>>) : sizeof(XYZZ28<C>);
  if (n == 0 || K < 2 || K > MLHIP_MAX_SEGMENTS) return mlhip_rt::fail(MLHIP_EINVAL, "bad segment count");
  const size_t nbuckets = (size_t)p->W * p->M;
  if (!p->d_state28) HIPCHK(hipMalloc(&p->d_state28, nbuckets * kStateBytes));
  for (int s = 0; s < K; s++)
    if (!p->ev_seg[s]) HIPCHK(hipEventCreateWithFlags(&p->ev_seg[s], hipEventDisableTiming));
  // resident bases (h_points == nullptr): only the scalars travel; the carry-free copy must already be there
  const bool resident = h_points == nullptr;
  if (resident && !(p->points_static && p->conv_src == d_points && n <= p->conv_n))
    return mlhip_rt::fail(MLHIP_EINVAL, "streamed MSM over resident bases needs their converted copy");
  p->pending_n = n;
  p->pending = true;
  if (!resident) p->conv_src = nullptr;  // the carry-free copy no longer matches any resident buffer
  const size_t seg = (n + K - 1) / K;
  const char* hp = (const char*)h_points;
  const char* hs = (const char*)h_scalars;
  HIPCHK(hipEventRecord(p->ev_fork, st));  // the staging buffers are free once the work queued before us is done
  HIPCHK(hipStreamWaitEvent(p->aux, p->ev_fork, 0));
  int s = 0;
  for (size_t off = 0; off < n; off += seg, s++) {
    const size_t len = std::min(seg, n - off);
    const bool first = off == 0, last = off + len >= n;
    const int flags = (first ? MLHIP_SEG_FIRST : 0) | (last ? MLHIP_SEG_LAST : 0);
    char* dsc = (char*)d_scalars + off * 32;
    A* dpt = (A*)d_points + off;
    HIPCHK(hipMemcpyAsync(dsc, hs + off * 32, len * 32, hipMemcpyHostToDevice, p->aux));
    if (!resident) {
      HIPCHK(hipMemcpyAsync(dpt, hp + off * sizeof(A), len * sizeof(A), hipMemcpyHostToDevice, p->aux));
      if constexpr (kG2)
        k_points_to28_g2<C><<<dim3((unsigned)((4 * len + 255) / 256)), dim3(256), 0, p->aux>>>(
            dpt, len, (AffineG2_28<C>*)p->d_points28 + off);
      else
        k_points_to28<C><<<dim3((unsigned)((len + 255) / 256)), dim3(256), 0, p->aux>>>(dpt, len,
                                                                                       (Affine28<C>*)p->d_points28 + off);
    }
    HIPCHK(hipEventRecord(p->ev_seg[s], p->aux));
    HIPCHK(hipStreamWaitEvent(st, p->ev_seg[s], 0));
    HIPCHK(hipMemsetAsync(p->d_zero, 0, p->zero_bytes, st));
    {
      int rc_sort = launch_sort<C>(p, dsc, mont, len, st, false);
      if (rc_sort) return rc_sort;
    }
    uint32_t big_threshold = (uint32_t)std::min<size_t>((len >> (p->c - 1)) * 8, 1u << 30);
    if (big_threshold < BIG_BUCKET_MIN) big_threshold = BIG_BUCKET_MIN;
    if constexpr (kG2) {
      k_accumulate28_lp_seg<C><<<dim3((unsigned)((2 * nbuckets + p->acc_block - 1) / p->acc_block)), dim3(p->acc_block), 0, st>>>(
          (const AffineG2_28<C>*)p->d_points28 + off, p->d_sorted, p->d_offsets, p->d_counts, nbuckets, p->d_order,
          big_threshold, p->d_biglist, p->d_bigcount, (XYZZ28L<Fp28<C>>*)p->d_state28, flags, (X*)p->d_buckets);
      constexpr int BB = 128;
      launch_big_slices<F, BB>(p, dpt, st);
      k_accumulate_big_seg_g2<C, BB><<<dim3(256), dim3(BB), BB * sizeof(X), st>>>(
          p->d_biglist, p->d_bigcount, p->d_bigprefix, (const X*)p->d_bigpart, (XYZZ28L<Fp28<C>>*)p->d_state28, flags,
          (X*)p->d_buckets);
    } else {
      k_accumulate28_seg<C><<<dim3((unsigned)((nbuckets + p->acc_block - 1) / p->acc_block)), dim3(p->acc_block), 0, st>>>(
          (const Affine28<C>*)p->d_points28 + off, p->d_sorted, p->d_offsets, p->d_counts, nbuckets, p->d_order,
          big_threshold, p->d_biglist, p->d_bigcount, (XYZZ28<C>*)p->d_state28, flags, (X*)p->d_buckets);
      constexpr int BB = 256;
      launch_big_slices<F, BB>(p, dpt, st);
      k_accumulate_big_seg<C, BB><<<dim3(256), dim3(BB), BB * sizeof(X), st>>>(
          p->d_biglist, p->d_bigcount, p->d_bigprefix, (const X*)p->d_bigpart, (XYZZ28<C>*)p->d_state28, flags,
          (X*)p->d_buckets);
    }
  }
  {
    int rc_red = launch_reduce<C, F>(p, st);
    if (rc_red) return rc_red;
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(p->h_out, p->d_out, (size_t)p->W * p->nsel * sizeof(X), hipMemcpyDeviceToHost, st));
  HIPCHK(hipEventRecord(p->done, st));
  return 0;
}

template <class C, class F>
int plan_finish(mlhip_msm_plan* p, void* out_affine, void* out_xyzz) {
  typedef Affine<F> A;
  typedef XYZZ<F> X;
  X total;
  if (!p->pending) return mlhip_rt::fail(MLHIP_EINVAL, "mlhip_msm_finish without a pending mlhip_msm_launch");
  p->pending = false;
  if (p->pending_n == 0) {
    xyzz_set_inf<F>(total);
  } else {
    HIPCHK(hipEventSynchronize(p->done));
    auto t0 = std::chrono::steady_clock::now();
    host_tail<F>(p, total);
    if (p->profiling) {
      for (int i = 0; i < 4; i++) HIPCHK(hipEventElapsedTime(&p->ms[i], p->ev[i], p->ev[i + 1]));
      HIPCHK(hipEventElapsedTime(&p->ms[4], p->ev[0], p->ev[4]));
      p->ms[5] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
  }
  A r;
  xyzz_to_affine<F>(r, total);
  memcpy(out_affine, &r, sizeof(A));
  if (out_xyzz) memcpy(out_xyzz, &total, sizeof(X));
  return 0;
}

}  // namespace mlhip
