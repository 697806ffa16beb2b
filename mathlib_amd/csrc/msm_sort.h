// msm_sort.h -- from scalars to bucket-sorted entries: signed window digits, the two-level LDS counting sort of the
// (window, bucket) keys (with the multi-workgroup path for oversized coarse bins), the global-atomic variant for
// very large n, and the bucket order by population.  Part of msm_kernels.h.
#pragma once
// (included by msm_kernels.h after its common headers and constants)

namespace mlhip {

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
  if (n_tiles <= 1) return;  // one tile is the whole scan (the coarse bins of the two-level sort: 2048 counts)
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
// Scalars per block in the coarse passes.  A block writes tile * W / NB entries into each coarse bin (8 at tile = 1024,
// c = 16): short runs, so the scatter's write traffic is several times the bytes of the entries
// (profiles/r01_pmc_traffic.json: 362 MB for 64 MB).  Large inputs have blocks to spare and take longer tiles
// (sort_tile_for); the per-block histogram handed from k_coarse_hist to k_coarse_scatter is 16-bit, so tile < 65536.
constexpr int SORT_TILE = 1024;      // smallest tile: sizes the per-block histogram buffer
constexpr int SORT_TILE_MAX = 8192;
static inline int sort_tile_for(size_t n) {
  if (const char* e = getenv("MLHIP_SORT_TILE")) {
    const int v = atoi(e);
    if (v == 1024 || v == 2048 || v == 4096 || v == 8192) return v;
  }
  // keep at least ~4 blocks per CU (256 CUs) in flight
  if (n >= ((size_t)1 << 23)) return 8192;
  if (n >= ((size_t)1 << 22)) return 4096;
  if (n >= ((size_t)1 << 20)) return 2048;  // 2^20: digits + sort 0.45 -> 0.42 ms; 2^19 is best at 1024, 2^21 is flat
  return 1024;
}

template <class C>
__global__ void __launch_bounds__(256) k_coarse_hist(const uint32_t* __restrict__ scalars, size_t n, int mont, int c, int W,
                                                     int low, uint32_t NB, uint32_t* __restrict__ coarse_count,
                                                     uint16_t* __restrict__ blockhist, int tile, uint32_t fold_stride) {
  extern __shared__ uint32_t lds_u32[];
  uint32_t* hist = lds_u32;
  for (uint32_t b = threadIdx.x; b < NB; b += 256) hist[b] = 0;
  __syncthreads();
  const uint32_t cb_shift = (uint32_t)(c - 1 - low);  // coarse bins per window = 1 << cb_shift
  // fold_stride != 0: a folded plan (msm_fold.h) -- the W digits of a scalar (same layout) share one bucket set (window 0 of
  // the bin numbering) and digit w's entry carries the table row index i + w fold_stride
  const WinLayout wl = msm_win_layout(C::FR_BITS, c);
  for (int k = 0; k < tile / 256; k++) {
    size_t i = (size_t)blockIdx.x * tile + (size_t)k * 256 + threadIdx.x;
    if (i >= n) break;
    // recompute the digit chain window by window (no per-lane array: keeps this in registers)
    uint32_t s[8];
    fr_canonical<C>(s, scalars + 8 * i, mont != 0);
    uint32_t carry = 0, neg = 0;
    for (int w = 0; w < W; w++) {
      const uint32_t mag = msm_window_digit(s, msm_win_off(wl.base, wl.rem, w), msm_win_bits(wl.base, wl.rem, w), carry, neg);
      if (mag) atomicAdd(&hist[((fold_stride ? 0u : (uint32_t)w) << cb_shift) + ((mag - 1) >> low)], 1u);
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
                                                        const uint16_t* __restrict__ blockhist, int tile, uint32_t fold_stride) {
  extern __shared__ uint32_t lds_u32[];
  uint32_t* hist = lds_u32;       // per-block count, then running rank
  uint32_t* base = lds_u32 + NB;  // global position of this block's slice of each bin
  const uint32_t cb_shift = (uint32_t)(c - 1 - low);
  const WinLayout wl = msm_win_layout(C::FR_BITS, c);
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
  for (int k = 0; k < tile / 256; k++) {
    size_t i = (size_t)blockIdx.x * tile + (size_t)k * 256 + threadIdx.x;
    if (i >= n) break;
    uint32_t s[8];
    fr_canonical<C>(s, scalars + 8 * i, mont != 0);
    uint32_t carry = 0, neg = 0;
    for (int w = 0; w < W; w++) {
      const uint32_t mag = msm_window_digit(s, msm_win_off(wl.base, wl.rem, w), msm_win_bits(wl.base, wl.rem, w), carry, neg);
      if (mag) {
        uint32_t bkt = mag - 1;
        uint32_t bin = ((fold_stride ? 0u : (uint32_t)w) << cb_shift) + (bkt >> low);
        uint32_t pos = base[bin] + atomicAdd(&hist[bin], 1u);
        tmp[pos] = ((bkt & low_mask) << (idx_bits + 1)) | (neg << idx_bits) | ((uint32_t)i + (uint32_t)w * fold_stride);
      }
    }
  }
}

// The same placement with the block's entries STAGED in LDS and written out bin by bin (round 3).  k_coarse_scatter stores
// every entry with its own 4-byte store to a random coarse bin: 64 different lines per wave instruction, 355 MB of
// memory-side writes for 64 MB of entries (profiles/r02_pmc_traffic.json), and the kernel takes four times as long as
// k_coarse_hist, which does the same digit and LDS-atomic work without the stores.  Here a block of 1024 threads ranks its
// tile * W entries into an LDS image ordered by bin (local offsets = an exclusive scan of the block's histogram), then
// groups of G = tile * W / NB lanes copy one bin's run each to its reserved slice: every store instruction writes whole
// runs of 32 - 64 contiguous bytes.  LDS: 3 NB + tile * W words (c = 16: 88 KB at tile 1024, 152 KB at 2048), one block per CU.
template <class C>
__global__ void __launch_bounds__(1024) k_coarse_scatter_staged(const uint32_t* __restrict__ scalars, size_t n, int mont, int c,
                                                                int W, int low, int idx_bits, uint32_t NB,
                                                                const uint32_t* __restrict__ coarse_off,
                                                                uint32_t* __restrict__ coarse_cursor, uint32_t* __restrict__ tmp,
                                                                const uint16_t* __restrict__ blockhist, int tile, int group,
                                                                uint32_t fold_stride) {
  extern __shared__ uint32_t lds_u32[];
  uint32_t* cnt = lds_u32;            // this block's count per bin, then the running rank
  uint32_t* loc = lds_u32 + NB;       // start of the bin's run in the staged image
  uint32_t* base = lds_u32 + 2 * NB;  // global position of this block's slice of the bin
  uint32_t* stage = lds_u32 + 3 * NB;
  __shared__ uint32_t wave_sum[16];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t cb_shift = (uint32_t)(c - 1 - low);
  const WinLayout wl = msm_win_layout(C::FR_BITS, c);
  // exclusive scan of the block's histogram: thread t owns bins [t per, (t + 1) per)
  const uint32_t per = (NB + 1023u) / 1024u;
  uint32_t mine = 0;
  for (uint32_t k = 0; k < per; k++) {
    const uint32_t b = tid * per + k;
    if (b < NB) mine += blockhist[(size_t)blockIdx.x * NB + b];
  }
  uint32_t incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
    if (lane >= (uint32_t)d) incl += up;
  }
  if (lane == 63) wave_sum[wave] = incl;
  __syncthreads();
  uint32_t before = 0;
  for (uint32_t w = 0; w < wave; w++) before += wave_sum[w];
  uint32_t run = before + incl - mine;
  for (uint32_t k = 0; k < per; k++) {
    const uint32_t b = tid * per + k;
    if (b < NB) {
      const uint32_t h = blockhist[(size_t)blockIdx.x * NB + b];
      loc[b] = run;
      run += h;
      base[b] = h ? coarse_off[b] + atomicAdd(&coarse_cursor[b], h) : 0u;
      cnt[b] = 0;
    }
  }
  __syncthreads();
  // rank into the staged image
  const uint32_t low_mask = (1u << low) - 1u;
  for (int k = 0; k < tile / 1024; k++) {
    const size_t i = (size_t)blockIdx.x * tile + (size_t)k * 1024 + tid;
    if (i < n) {
      uint32_t s[8];
      fr_canonical<C>(s, scalars + 8 * i, mont != 0);
      uint32_t carry = 0, neg = 0;
      for (int w = 0; w < W; w++) {
        const uint32_t mag = msm_window_digit(s, msm_win_off(wl.base, wl.rem, w), msm_win_bits(wl.base, wl.rem, w), carry, neg);
        if (mag) {
          const uint32_t bkt = mag - 1;
          const uint32_t bin = ((fold_stride ? 0u : (uint32_t)w) << cb_shift) + (bkt >> low);
          const uint32_t r = atomicAdd(&cnt[bin], 1u);
          stage[loc[bin] + r] = ((bkt & low_mask) << (idx_bits + 1)) | (neg << idx_bits) | ((uint32_t)i + (uint32_t)w * fold_stride);
        }
      }
    }
  }
  __syncthreads();
  // write out: `group` adjacent lanes per bin (a power of two <= 64)
  const uint32_t g = tid / (uint32_t)group, r0 = tid % (uint32_t)group, ngroups = 1024u / (uint32_t)group;
  for (uint32_t b = g; b < NB; b += ngroups) {
    const uint32_t h = cnt[b], l0 = loc[b], b0 = base[b];
    for (uint32_t k = r0; k < h; k += (uint32_t)group) tmp[b0 + k] = stage[l0 + k];
  }
}

static __global__ void __launch_bounds__(256) k_fine_sort(const uint32_t* __restrict__ tmp, const uint32_t* __restrict__ coarse_off,
                                                   const uint32_t* __restrict__ coarse_count, int c, int low, int idx_bits,
                                                   uint32_t big_bin, uint32_t* __restrict__ counts,
                                                   uint32_t* __restrict__ offsets, uint32_t* __restrict__ sorted) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t fo[256];
  // the bin's sorted image is assembled in LDS and copied out in whole lines when it fits (round 3: the placements are
  // single 4-byte stores scattered over the bin's range -- 297 MB of memory-side writes for 64 MB of entries); a bin holds
  // W n / NB entries on average (8192 at 2^20, c = 16) with a Poisson spread of one percent
  constexpr uint32_t STAGE = 10240;
  __shared__ uint32_t stage[STAGE];
  const uint32_t bin = blockIdx.x;
  const uint32_t F = 1u << low;
  const uint32_t begin = coarse_off[bin], cnt = coarse_count[bin];
  if (cnt > big_bin) return;  // sorted by several workgroups: k_bigbin_hist / k_bigbin_place
  const bool staged = cnt <= STAGE;
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
        pos[j] = fo[f] + atomicAdd(&hist[f], 1u);  // relative to the bin's start
      }
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (pos[j] != 0xFFFFFFFFu) {
        const uint32_t v = (e[j] & idx_mask) | (((e[j] >> idx_bits) & 1u) << 31);
        if (staged)
          stage[pos[j]] = v;
        else
          sorted[begin + pos[j]] = v;
      }
  }
  if (staged) {
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < cnt; k += 256) sorted[begin + k] = stage[k];
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

}  // namespace mlhip
