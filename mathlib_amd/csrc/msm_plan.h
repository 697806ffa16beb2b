// msm_plan.h -- host side of the MSM: plan allocation, the launch sequences (one pass and streamed in segments), the
// host tail.  Part of msm_kernels.h.
#pragma once
// (included by msm_kernels.h after its common headers and constants)

namespace mlhip {

// G2 runs in the carry-free lane-pair form (ec28_lp.h: bucket loop, segment / tile state, lane-pair reduction) on every curve
// since round 3 -- BLS12-381 first (-14 % accumulation), then BLS12-377 (u^2 = -5: 2^20 points 12.35 -> 9.45 ms) and BN254
// (10 limbs: 5.25 -> 4.95 ms); MLHIP_ACC32=1 keeps the boundary-form kernels as the second implementation
template <class C>
constexpr bool g2_carry_free_v = true;

// G2 bucket accumulation (carry-free form): lane pairs split by component with dual products (k_accumulate28_lp_seg, default)
// or by coordinate with one-lane Karatsuba Fp2 products (k_accumulate28_kc_seg, MLHIP_G2_KC=1: 16 % fewer multiplier
// instructions per addition, but three 64-bit column combinations per product column and 130 spilled registers instead of
// 63 -- measured 3-5 % SLOWER; kept as a second implementation for the parity tests).  Read per launch
// so that a test can switch paths; both leave bit-identical bucket states.
static inline bool g2_split_by_coordinate() { return mlhip_alt_switch("MLHIP_G2_KC"); }

// Does this plan sum its buckets in twisted Edwards coordinates (ed28.h)?  G1 of a curve with the model, the SRS promise
// (mlhip_msm_plan_assume_srs, or a table mlhip_bases_create has checked: every point in the subgroup, and the converted copy
// made once -- converting per launch costs what the cheaper additions save: 0.95 ms for 2^20 points), the carry-free
// state the reduction reads, and enough buckets for one lane each (the quad-lane kernel of small MSMs stays on XYZZ).
// MLHIP_EDWARDS=0: never (the Weierstrass kernels stay the second implementation).
template <class C, class F>
inline bool plan_use_edwards(const mlhip_msm_plan* p) {
  if constexpr (C::HAS_EDWARDS && std::is_same<F, FpField<C>>::value) {
    const char* e = getenv("MLHIP_EDWARDS");
    return p->trust_subgroup && p->points_static && p->reduce28 && p->d_points28 && (size_t)p->W * p->M > QUAD_ACC_MAX_BUCKETS &&
           p->points28_elem >= sizeof(EdNiels28<C>) && !(e && e[0] == '0');
  }
  return false;
}

// the buffers launch_sort works in (a plan's own, or those of a sort-ahead helper record: sort_ahead_prepare)
inline int plan_alloc_sort(mlhip_msm_plan* p) {
  const size_t nbuckets = (size_t)p->W * p->M;
  // a folded plan (msm_fold.h) sorts one tile of its table at a time: at most fold_tile scalars, Wd entries each, whose
  // indices run over the Wd fold_tile rows of the tile
  const size_t sort_n = p->fold ? std::min(p->max_n, p->fold_tile) : p->max_n;
  HIPCHK(hipMalloc(&p->d_digits, (size_t)p->Wd * sort_n * 4));
  HIPCHK(hipMalloc(&p->d_sorted, (size_t)p->Wd * sort_n * 4));
  {
    // sort parameters: packed entry = fine bits | sign | index must fit 32 bits, coarse bins must fit LDS
    int idx_bits = 1;
    while (((size_t)1 << idx_bits) < (p->fold ? (size_t)p->Wd * p->fold_tile : p->max_n)) idx_bits++;
    int low = p->c - 1 < 8 ? p->c - 1 : 8;
    if (low > 31 - idx_bits) low = 31 - idx_bits;
    const char* legacy = getenv("MLHIP_LEGACY_SORT");
    uint32_t nb = low >= 1 ? (uint32_t)(nbuckets >> low) : 0;  // (plain plan: W << (c - 1 - low))
    if (low < 1 || nb > 4096 || (legacy && legacy[0] == '1' && !p->fold)) {
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
    static_assert(SORT_TILE_MAX < 65536, "a block puts at most one entry per scalar into a coarse bin: the count fits 16 bits");
    const size_t blocks = (sort_n + SORT_TILE - 1) / SORT_TILE;
    HIPCHK(hipMalloc(&p->d_blockhist, blocks * p->sort_nb * sizeof(uint16_t)));
  }
  HIPCHK(hipMalloc(&p->d_offsets, nbuckets * 4));
  HIPCHK(hipMalloc(&p->d_order, nbuckets * 4));
  {
    const size_t nblk = (nbuckets + 255) / 256;
    const size_t hist_n = (size_t)ORDER_BINS * nblk;
    HIPCHK(hipMalloc(&p->d_hist, hist_n * 4));
    const size_t tiles = (std::max(nbuckets, hist_n) + SCAN_TILE - 1) / SCAN_TILE;
    HIPCHK(hipMalloc(&p->d_tilesums, (tiles + 1) * 4));
  }
  return 0;
}

template <class F>
int plan_alloc(mlhip_msm_plan* p) {
  const size_t nbuckets = (size_t)p->W * p->M;
  p->pt_size = sizeof(Affine<F>);
  p->xyzz_size = sizeof(XYZZ<F>);
  {
    int rc_sort = plan_alloc_sort(p);
    if (rc_sort) return rc_sort;
  }
  HIPCHK(hipMalloc(&p->d_biglist, nbuckets * 4));
  {
    // long buckets: at most W n / BIG_BUCKET_MIN of them, and W n / BIG_SLICE + one more slice per bucket
    const size_t entries = (size_t)p->Wd * (p->fold ? std::min(p->max_n, p->fold_tile) : p->max_n);
    const size_t nbig_max = std::min(nbuckets, entries / BIG_BUCKET_MIN + 1);
    HIPCHK(hipMalloc(&p->d_bigprefix, (nbig_max + 2) * 4));
    HIPCHK(hipMalloc(&p->d_bigpart, (entries / BIG_SLICE + nbig_max + 2) * p->xyzz_size));
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
    p->reduce_one_lane = mlhip_alt_switch("MLHIP_REDUCE_ONE_LANE");  // =1: the one-point-per-lane reduction kernels
  }
  if constexpr (std::is_same<F, FpField<typename F::Curve>>::value) {
    // G1 accumulation runs in the carry-free form (fp28.h): -24 % time for the 12-limb fields, -7 % for BN254.
    // MLHIP_ACC32=1 selects the boundary-form kernel (kept as the second implementation the tests compare with).
    const bool want28 = !mlhip_alt_switch("MLHIP_ACC32");
    // (a curve with a twisted Edwards model grows the buffer to the Niels triples' 168 B a point when the SRS promise is made)
    constexpr size_t kPoint28 = sizeof(Affine28<typename F::Curve>);
    if (want28 && p->fold) {
      p->points28_elem = 0;  // the table is allocated by plan_fold_build, in the form the plan will read
      p->points28_elem_ed = F::Curve::HAS_EDWARDS ? sizeof(EdNiels28<typename F::Curve>) : 0;
    } else if (want28) {
      HIPCHK(hipMalloc(&p->d_points28, p->max_n * kPoint28));
      p->points28_elem = kPoint28;
      p->points28_elem_ed = F::Curve::HAS_EDWARDS ? sizeof(EdNiels28<typename F::Curve>) : 0;
    }
    // ... and so does the quad-lane reduction, on the accumulators as the kernel leaves them (MLHIP_REDUCE32=1: the
    // boundary-form reduction kernels, kept as the second implementation)
    p->reduce28 = want28 && !p->reduce_one_lane && !mlhip_alt_switch("MLHIP_REDUCE32");
    if (p->reduce28) HIPCHK(hipMalloc(&p->d_state28, nbuckets * sizeof(XYZZ28<typename F::Curve>)));
  }
  if constexpr (std::is_same<F, Fp2Field<typename F::Curve>>::value && g2_carry_free_v<typename F::Curve>) {
    // G2 in the carry-free form (g2_carry_free_v above)
    if (!mlhip_alt_switch("MLHIP_ACC32")) {
      if (!p->fold) HIPCHK(hipMalloc(&p->d_points28, p->max_n * sizeof(AffineG2_28<typename F::Curve>)));
      // ... and the lane-pair reduction reads the accumulators as the kernel leaves them (MLHIP_REDUCE32=1: boundary form)
      p->reduce28 = !mlhip_alt_switch("MLHIP_REDUCE32");
      if (p->reduce28) HIPCHK(hipMalloc(&p->d_state28, nbuckets * 2 * sizeof(XYZZ28L<Fp28<typename F::Curve>>)));
    }
  }
  HIPCHK(hipMalloc(&p->d_buckets, nbuckets * p->xyzz_size));
  {
    size_t chunk_size = p->xyzz_size;
    if constexpr (std::is_same<F, FpField<typename F::Curve>>::value)
      chunk_size = std::max(chunk_size, sizeof(XYZZ28<typename F::Curve>));
    else
      chunk_size = std::max(chunk_size, 2 * sizeof(XYZZ28L<Fp28<typename F::Curve>>));
    HIPCHK(hipMalloc(&p->d_A, (size_t)p->W * p->T * chunk_size));
    HIPCHK(hipMalloc(&p->d_W0, (size_t)p->W * p->T * chunk_size));
  }
  if (p->fold && p->W > 1 && p->W <= 32 && p->reduce28) {  // (G1: k_group_combine_q on quads; G2: k_group_combine_lp on lane pairs)
    int lgW = 0;
    while ((1 << lgW) < p->W) lgW++;
    p->fold_nsel2 = 4 + p->nb + lgW;
  }
  HIPCHK(hipMalloc(&p->d_out, ((size_t)p->W * p->nsel + p->fold_nsel2) * p->xyzz_size));
  HIPCHK(hipHostMalloc(&p->h_out, ((size_t)p->W * p->nsel + p->fold_nsel2) * p->xyzz_size, hipHostMallocDefault));
  for (int i = 0; i < 5; i++) HIPCHK(hipEventCreate(&p->ev[i]));
  HIPCHK(hipEventCreateWithFlags(&p->done, hipEventDisableTiming));
  {  // the auxiliary stream: point conversion beside the sort, and the uploads of a streamed host-buffer MSM
    HIPCHK(hipStreamCreateWithFlags(&p->aux, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
  }
  return 0;
}

// total = sum_w 2^off(w) V_w,  V_w = out[w][0..3] summed + 2^lgL sum_k 2^k out[w][4+k]
// (window w starts at bit off(w): msm_win_layout, widths differ by at most one bit).  The V_w are independent chains of
// nb + lgL doublings and nb + 4 additions each -- two thirds of the tail's field products -- and run on the library's host
// workers (mlhip_rt::host_parallel); what stays sequential is the Horner pass over the windows, one doubling per scalar
// bit.  2^20 points, c = 16: 0.20 -> 0.10 ms.  MLHIP_HOST_THREADS=1 keeps everything on the calling thread.
// (host_parallel copies its input blob: {nb, lgL, nsel} then the W x nsel partial sums as the device left them)
// A folded plan (msm_fold.h): job g < W is the weighted sum V_g of bucket group g (as a window's), job W + g its PLAIN sum
// S_g = out[g][2] + out[g][3] (the two halves of sum_t A[g][t]).  Bucket b of group g has weight g M + b + 1, so
//   total = sum_g V_g + M sum_g g S_g
// -- lg M doublings in all, where the windows of a plain plan need one per scalar bit.
template <class F>
void host_tail_group(const void* in, int job, void* out) {
  const HostTailHeader& h = *static_cast<const HostTailHeader*>(in);
  const int W = h.pad;
  if (job < W) {
    host_tail_window<F>(in, job, out);
    return;
  }
  const XYZZ<F>* o = reinterpret_cast<const XYZZ<F>*>(static_cast<const unsigned char*>(in) + sizeof(HostTailHeader)) + (size_t)(job - W) * h.nsel;
  XYZZ<F> acc = o[2];
  xyzz_add<F>(acc, o[3]);
  memcpy(out, &acc, sizeof(acc));
}
template <class F>
void host_tail_fold(const mlhip_msm_plan* p, XYZZ<F>& total) {
  if (p->fold_nsel2) {
    // the device has combined the groups (k_group_combine_q): one window of W T chunks, summed on this thread
    const size_t sums2 = (size_t)p->fold_nsel2 * sizeof(XYZZ<F>);
    std::vector<unsigned char> blob2(sizeof(HostTailHeader) + sums2);
    const HostTailHeader h2{p->fold_nsel2 - 4, p->lgL, p->fold_nsel2, 0};
    memcpy(blob2.data(), &h2, sizeof(h2));
    memcpy(blob2.data() + sizeof(h2), static_cast<const XYZZ<F>*>(p->h_out) + (size_t)p->W * p->nsel, sums2);
    host_tail_window<F>(blob2.data(), 0, &total);
    return;
  }
  const size_t sums = (size_t)p->W * p->nsel * sizeof(XYZZ<F>);
  std::vector<unsigned char> blob(sizeof(HostTailHeader) + sums);
  const HostTailHeader h{p->nb, p->lgL, p->nsel, p->W};
  memcpy(blob.data(), &h, sizeof(h));
  memcpy(blob.data() + sizeof(h), p->h_out, sums);
  const int W = p->W;
  std::vector<XYZZ<F>> VS(2 * (size_t)W);
  mlhip_rt::host_parallel(W > 1 ? 2 * W : 1, host_tail_group<F>, blob.data(), blob.size(), VS.data(), sizeof(XYZZ<F>));
  // sum_g g S_g as a running sum from the top group down (run += S_g; hi += run), then the lg M doublings
  XYZZ<F> run, hi;
  xyzz_set_inf<F>(run);
  xyzz_set_inf<F>(hi);
  for (int g = W - 1; g >= 1; g--) {
    xyzz_add<F>(run, VS[(size_t)W + g]);
    xyzz_add<F>(hi, run);
  }
  int lgM = 0;
  while ((1u << lgM) < p->M) lgM++;
  const int down0 = lgM;
  horner_jac<F>(total, &hi, 1, &down0);  // total = 2^lgM hi
  for (int g = 0; g < W; g++) xyzz_add<F>(total, VS[g]);
}

template <class F>
void host_tail(const mlhip_msm_plan* p, XYZZ<F>& total) {
  if (p->fold) {
    host_tail_fold<F>(p, total);
    return;
  }
  const WinLayout wl = msm_win_layout(F::Curve::FR_BITS, p->c);
  const size_t sums = (size_t)p->W * p->nsel * sizeof(XYZZ<F>);
  std::vector<unsigned char> blob(sizeof(HostTailHeader) + sums);
  const HostTailHeader h{p->nb, p->lgL, p->nsel, 0};
  memcpy(blob.data(), &h, sizeof(h));
  memcpy(blob.data() + sizeof(h), p->h_out, sums);
  std::vector<XYZZ<F>> V(p->W);
  mlhip_rt::host_parallel(p->W, host_tail_window<F>, blob.data(), blob.size(), V.data(), sizeof(XYZZ<F>));
  // the sequential part: one doubling per scalar bit.  In Jacobian coordinates since round 4 (ec_jac.h: 2M + 5S a doubling
  // instead of 6M + 3S; the 16 additions are dearer and do not matter)
  std::vector<int> down(p->W, 0);
  for (int w = 1; w < p->W; w++) down[w] = msm_win_off(wl.base, wl.rem, w) - msm_win_off(wl.base, wl.rem, w - 1);
  horner_jac<F>(total, V.data(), p->W, down.data());
}

// slice sums of the long buckets listed by the accumulation kernel (nothing to do, two near-empty launches, when
// there are none)
// (`row0`: a folded plan keeps carry-free rows only -- the slices gather from d_points28 + row0, in the form the table holds)
template <class F, int BB>
void launch_big_slices(mlhip_msm_plan* p, const Affine<F>* d_points, hipStream_t st, const mlhip_msm_plan* sv = nullptr,
                       size_t row0 = 0) {
  typedef typename F::Curve C;
  if (!sv) sv = p;  // the plan whose entry lists (sorted / offsets / counts) describe this tile
  k_big_prefix<<<dim3(1), dim3(1024), 0, st>>>(sv->d_counts, p->d_biglist, p->d_bigcount, p->d_bigprefix);
  if (p->fold) {
    if constexpr (std::is_same<F, Fp2Field<C>>::value) {
      k_big_slices_lp28<C, 256><<<dim3(1024), dim3(256), 128 * sizeof(XYZZ<F>), st>>>(
          (const AffineG2_28<C>*)p->d_points28 + row0, sv->d_sorted, sv->d_offsets, sv->d_counts, p->d_biglist, p->d_bigcount,
          p->d_bigprefix, (XYZZ<F>*)p->d_bigpart);
    } else {
      bool done_ed = false;
      if constexpr (C::HAS_EDWARDS) {
        if (p->conv_ed) {
          k_big_slices28<C, BB, true><<<dim3(1024), dim3(BB), BB * sizeof(XYZZ<F>), st>>>(
              (const EdNiels28<C>*)p->d_points28 + row0, sv->d_sorted, sv->d_offsets, sv->d_counts, p->d_biglist, p->d_bigcount,
              p->d_bigprefix, (XYZZ<F>*)p->d_bigpart);
          done_ed = true;
        }
      }
      if (!done_ed)
        k_big_slices28<C, BB, false><<<dim3(1024), dim3(BB), BB * sizeof(XYZZ<F>), st>>>(
            (const Affine28<C>*)p->d_points28 + row0, sv->d_sorted, sv->d_offsets, sv->d_counts, p->d_biglist, p->d_bigcount,
            p->d_bigprefix, (XYZZ<F>*)p->d_bigpart);
    }
    return;
  }
  if constexpr (std::is_same<F, Fp2Field<C>>::value) {
    // G2: 128 lane pairs per slice (48 KB of LDS)
    k_big_slices_lp<C, 256><<<dim3(1024), dim3(256), 128 * sizeof(XYZZ<F>), st>>>(d_points, sv->d_sorted, sv->d_offsets,
                                                                                  sv->d_counts, p->d_biglist, p->d_bigcount,
                                                                                  p->d_bigprefix, (XYZZ<F>*)p->d_bigpart);
  } else {
    k_big_slices<F, BB><<<dim3(1024), dim3(BB), BB * sizeof(XYZZ<F>), st>>>(d_points, sv->d_sorted, sv->d_offsets, sv->d_counts,
                                                                          p->d_biglist, p->d_bigcount, p->d_bigprefix,
                                                                          (XYZZ<F>*)p->d_bigpart);
  }
}

// digits -> entries sorted by (window, bucket) in d_sorted / d_offsets / d_counts, and the bucket order by population
template <class C>
int launch_sort(mlhip_msm_plan* p, const void* d_scalars, int mont, size_t n, hipStream_t st, bool prof) {
  const size_t nbuckets = (size_t)p->W * p->M;
  const int Wd = p->Wd;  // entries per scalar (== W on a plain plan)
  const uint32_t fold_stride = p->fold ? (uint32_t)p->fold_tile : 0u;
  if (p->fold && (p->sort_low <= 0 || n > p->fold_tile))
    return mlhip_rt::fail(MLHIP_EINVAL, "folded plan: a sort covers at most one tile of the table, on the two-level path");
  if (p->sort_low > 0) {
    // two-level LDS counting sort (no per-key global atomics)
    int tile = sort_tile_for(n);
    // (a folded plan's digits share the bins: a block may put tile * Wd entries into one bin, and the count is 16-bit)
    // (plan_create_ex admits only digit counts for which the smallest tile fits: Wd SORT_TILE < 65536)
    while (p->fold && tile > SORT_TILE && (size_t)tile * Wd >= 65536) tile /= 2;
    const uint32_t NB = p->sort_nb;
    // entries staged in LDS and written bin by bin (k_coarse_scatter_staged) when a block's tile * W entries fit beside
    // the three bin tables -- the tile shrinks to make them fit; MLHIP_SCATTER_STAGED=0: one store per entry (round 1)
    const char* staged_env = getenv("MLHIP_SCATTER_STAGED");  // read per launch so that a test can switch paths
    const bool staged_on = !(staged_env && staged_env[0] == '0');
    constexpr size_t kLdsMax = 160 * 1024 - 256;
    auto staged_lds = [&](int t) { return (3 * (size_t)NB + (size_t)t * Wd) * 4; };
    bool staged = false;
    if (staged_on) {
      int t = tile;
      while (t > 1024 && staged_lds(t) > kLdsMax) t /= 2;
      if (staged_lds(t) <= kLdsMax) {
        tile = t;
        staged = true;
      }
    }
    const unsigned blocks = (unsigned)((n + tile - 1) / tile);
    k_coarse_hist<C><<<dim3(blocks), dim3(256), NB * 4, st>>>((const uint32_t*)d_scalars, n, mont, p->c, Wd, p->sort_low, NB,
                                                            p->d_coarse_count, p->d_blockhist, tile, fold_stride);
    if (prof) HIPCHK(hipEventRecord(p->ev[1], st));
    launch_scan(p->d_coarse_count, p->d_coarse_off, p->d_tilesums, NB, st);
    if (staged) {
      int group = 1;  // lanes per bin in the write-out: the mean run length, rounded down to a power of two
      while (group < 64 && (size_t)group * 2 * NB <= (size_t)tile * Wd) group *= 2;
      HIPCHK(hipFuncSetAttribute((const void*)k_coarse_scatter_staged<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax));
      k_coarse_scatter_staged<C><<<dim3(blocks), dim3(1024), staged_lds(tile), st>>>(
          (const uint32_t*)d_scalars, n, mont, p->c, Wd, p->sort_low, p->sort_idx_bits, NB, p->d_coarse_off, p->d_coarse_cursor,
          p->d_digits, p->d_blockhist, tile, group, fold_stride);
    } else {
      k_coarse_scatter<C><<<dim3(blocks), dim3(256), NB * 8, st>>>((const uint32_t*)d_scalars, n, mont, p->c, Wd, p->sort_low,
                                                                 p->sort_idx_bits, NB, p->d_coarse_off, p->d_coarse_cursor,
                                                                 p->d_digits, p->d_blockhist, tile, fold_stride);
    }
    // bins more than 8x the mean (and at least 32768 entries) are sorted by many workgroups
    const uint32_t big_bin = (uint32_t)std::min<size_t>(std::max<size_t>(32768, 8 * ((size_t)Wd * n / NB)), 0x7fffffffu);
    k_fine_sort<<<dim3(NB), dim3(256), 0, st>>>(p->d_digits, p->d_coarse_off, p->d_coarse_count, p->c, p->sort_low,
                                               p->sort_idx_bits, big_bin, p->d_counts, p->d_offsets, p->d_sorted);
    if ((size_t)Wd * n > big_bin) {  // a bin holds at most all W n entries: small MSMs skip three near-empty launches
      k_bigbin_prefix<<<dim3(1), dim3(1024), 0, st>>>(p->d_coarse_count, NB, big_bin, p->d_binprefix);
      k_bigbin_hist<<<dim3(1024), dim3(256), 0, st>>>(p->d_digits, p->d_coarse_off, p->d_coarse_count, p->d_binprefix, NB,
                                                     p->sort_low, p->sort_idx_bits, p->d_counts);
      k_bigbin_place<<<dim3(1024), dim3(256), 0, st>>>(p->d_digits, p->d_coarse_off, p->d_coarse_count, p->d_binprefix, NB,
                                                      p->sort_low, p->sort_idx_bits, p->d_counts, p->d_cursor, p->d_offsets,
                                                      p->d_sorted);
    }
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
      bool done28 = false;
      if constexpr (g2_carry_free_v<C>) {
        if (p->reduce28) {
          typedef XYZZ28L<Fp28<C>> X28;
          k_chunks_lp28<C><<<dim3((unsigned)((2 * n_chunks + 255) / 256)), dim3(256), 0, st>>>(
              (const X28*)p->d_state28, n_chunks, p->L, (X28*)p->d_A, (X28*)p->d_W0);
          constexpr int RB = 512;  // 256 lane pairs, 128 slots x 448 B = 56 KB of LDS per block
          k_masked_sums_lp28<C, RB><<<dim3((unsigned)(p->W * p->nsel)), dim3(RB), (RB / 4) * 2 * sizeof(X28), st>>>(
              (const X28*)p->d_A, (const X28*)p->d_W0, p->T, p->nsel, (X*)p->d_out);
          if (p->fold_nsel2)  // folded plan: the groups' sums combined into one window's
            k_group_combine_lp<C, 128><<<dim3((unsigned)p->fold_nsel2), dim3((unsigned)(4 * p->W)), 2 * p->W * sizeof(X), st>>>(
                (const X*)p->d_out, p->W, p->nsel, p->nb, (X*)p->d_out + (size_t)p->W * p->nsel);
          done28 = true;
        }
      }
      if constexpr (kBuildAlt) {  // (boundary-form lane pairs: MLHIP_REDUCE32 / MLHIP_ACC32, test build only)
        if (!done28) {
          k_chunks_lp<C><<<dim3((unsigned)((2 * n_chunks + 255) / 256)), dim3(256), 0, st>>>((const X*)p->d_buckets, n_chunks,
                                                                                            p->L, (X*)p->d_A, (X*)p->d_W0);
          constexpr int RB = 512;  // 256 lane pairs, 128 slots x 384 B = 48 KB of LDS per block
          k_masked_sums_lp<C, RB><<<dim3((unsigned)(p->W * p->nsel)), dim3(RB), (RB / 4) * sizeof(X), st>>>(
              (const X*)p->d_A, (const X*)p->d_W0, p->T, p->nsel, (X*)p->d_out);
        }
      } else if (!done28) {
        return mlhip_rt::fail(MLHIP_EINVAL, "G2 reduction: the carry-free bucket state is missing");
      }
    } else {
      if (p->reduce28) {
        // the accumulation left its carry-free bucket state in d_state28 (MLHIP_SEG_KEEP28)
        typedef XYZZ28<C> X28;
        constexpr int RB = 512;  // 128 quads, 28 KB of LDS per block
        const dim3 cgrid((unsigned)((4 * n_chunks + p->red_block - 1) / p->red_block)), cblock(p->red_block);
        bool done_ed = false;
        if constexpr (C::HAS_EDWARDS) {
          if (p->last_ed) {  // ... in extended twisted Edwards coordinates (msm_ed.h): the same kernels on that addition
            k_chunks_q28<C, true><<<cgrid, cblock, 0, st>>>((const X28*)p->d_state28, n_chunks, p->L, (X28*)p->d_A, (X28*)p->d_W0);
            k_masked_sums_q28<C, RB, true><<<dim3((unsigned)(p->W * p->nsel)), dim3(RB), (RB / 4) * sizeof(X28), st>>>(
                (const X28*)p->d_A, (const X28*)p->d_W0, p->T, p->nsel, (X*)p->d_out);
            done_ed = true;
          }
        }
        if (!done_ed) {
          k_chunks_q28<C><<<cgrid, cblock, 0, st>>>((const X28*)p->d_state28, n_chunks, p->L, (X28*)p->d_A, (X28*)p->d_W0);
          k_masked_sums_q28<C, RB><<<dim3((unsigned)(p->W * p->nsel)), dim3(RB), (RB / 4) * sizeof(X28), st>>>(
              (const X28*)p->d_A, (const X28*)p->d_W0, p->T, p->nsel, (X*)p->d_out);
        }
        if (p->fold_nsel2)  // folded plan: the groups' sums combined into one window's (both forms leave Weierstrass sums)
          k_group_combine_q<C, 256><<<dim3((unsigned)p->fold_nsel2), dim3((unsigned)std::max(64, 8 * p->W)), 64 * sizeof(X), st>>>(
              (const X*)p->d_out, p->W, p->nsel, p->nb, (X*)p->d_out + (size_t)p->W * p->nsel);
      } else if constexpr (kBuildAlt) {  // boundary-form reductions: test build only
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
      } else {
        return mlhip_rt::fail(MLHIP_EINVAL, "G1 reduction: the carry-free bucket state is missing");
      }
    }
  }
  return 0;
}

template <class C, class F>
int plan_stream(mlhip_msm_plan* p, void* d_points, void* d_scalars, const void* h_points, const void* h_scalars, int mont,
                size_t n, int K, hipStream_t st);

// Number of tiles a device-resident MSM is cut into (1 = one pass over all points); see plan_stream.  Measured
// (profiles/r02_tiles.txt): G1 from 2^22 points on in tiles of 2^21 (235 MB of points), G2 from 2^23 on in tiles of 2^20
// (also 235 MB); at most MLHIP_MAX_SEGMENTS tiles.
// MLHIP_TILE_LOG2 = t forces tiles of 2^t points for every n above that (0 = never tile).
template <class C, class F>
int resident_tiles(const mlhip_msm_plan* p, size_t n) {
  constexpr bool kG2 = std::is_same<F, Fp2Field<C>>::value;
  if (!p->aux || !p->d_points28) return 1;
  if (p->fold) {  // one pass per tile of the table (an entry index addresses the rows of one tile)
    const size_t k = (n + p->fold_tile - 1) / p->fold_tile;
    return k < 2 ? 1 : (int)std::min<size_t>(k, MLHIP_MAX_SEGMENTS);
  }
  int lg = kG2 ? 20 : 21;
  size_t from = (size_t)1 << (kG2 ? 23 : 22);
  if (plan_use_edwards<C, F>(p)) {  // 168-byte Niels triples: 2^20 of them are what 2^21 Weierstrass points weigh
    lg = 20;
    from = (size_t)1 << 21;
  }
  if (const char* e = getenv("MLHIP_TILE_LOG2")) {
    const int v = atoi(e);
    if (v <= 0) return 1;
    lg = v > 30 ? 30 : v;
    from = ((size_t)1 << lg) + 1;
  }
  if (n < from) return 1;
  size_t k = (n + ((size_t)1 << lg) - 1) >> lg;
  if (k > MLHIP_MAX_SEGMENTS) k = MLHIP_MAX_SEGMENTS;
  return k < 2 ? 1 : (int)k;
}

template <class C, class F>
int plan_launch(mlhip_msm_plan* p, const void* d_points, const void* d_scalars, int mont, size_t n, hipStream_t st) {
  typedef Affine<F> A;
  typedef XYZZ<F> X;
  if (p->fold) {
    // the points are the plan's own table (plan_fold_build): the caller's point argument only names the bases it was built from
    if (n > p->fold_n || !p->d_points28 || p->upload_src)
      return mlhip_rt::fail(MLHIP_EINVAL, "folded plan: more scalars than tabulated bases, or no table");
    d_points = p->d_points28;  // (never read as Affine<F>: every kernel of a folded plan gathers from the carry-free rows)
  }
  if (n != 0 && !p->upload_src) {
    const int K = resident_tiles<C, F>(p, n);
    if (K > 1)
      return plan_stream<C, F>(p, const_cast<void*>(d_points), const_cast<void*>(d_scalars), nullptr, nullptr, mont, n, K, st);
  }
  p->tiles_timed = 0;
  p->pending_n = n;
  p->pending = true;
  p->last_ed = false;
  if (n != 0) {
    const size_t nbuckets = (size_t)p->W * p->M;
    const bool prof = p->profiling;
    // Buckets far longer than the mean (degenerate inputs: equal scalars, tiny scalars) are handed to a whole
    // workgroup each; the threshold scales with the mean length n / 2^(c-1) so that large n, and the sparser top
    // window (2-4x the mean for these group orders), stay on the one-thread-per-bucket path.
    // (a folded plan's buckets collect the entries of all Wd digits: the mean is Wd n / 2^(c-1))
    uint32_t big_threshold = (uint32_t)std::min<size_t>((((size_t)(p->fold ? p->Wd : 1) * n) >> (p->c - 1)) * 8, 1u << 30);
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
    const bool use_ed = p->fold ? p->conv_ed : plan_use_edwards<C, F>(p);  // (a table is read in the form it was built in)
    p->last_ed = use_ed;
    const bool conv_cached = p->fold || (p->points_static && p->conv_src == d_points && n <= p->conv_n && !p->upload_src && p->conv_ed == use_ed);
    if (p->d_points28 && conv_cached) {
      HIPCHK(hipEventRecord(p->ev_join, st));  // nothing to wait for
    } else if (p->d_points28) {
      HIPCHK(hipStreamWaitEvent(p->aux, p->ev_fork, 0));
      if (p->upload_src) {  // host-buffer call: the upload of the points rides the same stream, ahead of the conversion
        HIPCHK(hipMemcpyAsync(const_cast<void*>(d_points), p->upload_src, p->upload_bytes, hipMemcpyHostToDevice, p->aux));
        p->upload_src = nullptr;
      }
      if constexpr (std::is_same<F, Fp2Field<C>>::value) {
        if constexpr (g2_carry_free_v<C>)
          k_points_to28_g2<C><<<dim3((unsigned)((4 * n + 255) / 256)), dim3(256), 0, p->aux>>>(
              (const A*)d_points, n, (AffineG2_28<C>*)p->d_points28);
      } else {
        bool converted = false;
        if constexpr (C::HAS_EDWARDS) {
          if (use_ed) {
            k_points_to_ed28<C><<<dim3((unsigned)(((n + 3) / 4 + 255) / 256)), dim3(256), 0, p->aux>>>(
                (const A*)d_points, n, (EdNiels28<C>*)p->d_points28);
            converted = true;
          }
        }
        if (!converted)
          k_points_to28<C><<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p->aux>>>((const A*)d_points, n,
                                                                                       (Affine28<C>*)p->d_points28);
      }
      HIPCHK(hipEventRecord(p->ev_join, p->aux));
      p->conv_src = d_points;
      p->conv_n = n;
      p->conv_ed = use_ed;
    }
    if (prof) HIPCHK(hipEventRecord(p->ev[2], st));
    constexpr bool kLanePairs = std::is_same<F, Fp2Field<C>>::value;  // G2: two lanes per bucket
    if constexpr (kLanePairs) {
      bool done28 = false;
      if constexpr (g2_carry_free_v<C>) {
        if (p->d_points28) {
          HIPCHK(hipStreamWaitEvent(st, p->ev_join, 0));
          if constexpr (g2_carry_free_v<C>) {
            if (p->reduce28) {  // one segment that is first and last, leaving the raw accumulators for k_chunks_lp28
              const dim3 grid((unsigned)((2 * nbuckets + p->acc_block - 1) / p->acc_block)), block(p->acc_block);
              bool kc = false;
              if constexpr (C::BETA == -1 && kBuildAlt) {
                if (g2_split_by_coordinate()) {
                  k_accumulate28_kc_seg<C><<<grid, block, 0, st>>>(
                      (const AffineG2_28<C>*)p->d_points28, p->d_sorted, p->d_offsets, p->d_counts, nbuckets, p->d_order,
                      big_threshold, p->d_biglist, p->d_bigcount, (XYZZ28L<Fp28<C>>*)p->d_state28,
                      MLHIP_SEG_FIRST | MLHIP_SEG_LAST | MLHIP_SEG_KEEP28, (X*)p->d_buckets);
                  kc = true;
                }
              }
              if (!kc)
                k_accumulate28_lp_seg<C><<<grid, block, 0, st>>>(
                    (const AffineG2_28<C>*)p->d_points28, p->d_sorted, p->d_offsets, p->d_counts, nbuckets, p->d_order,
                    big_threshold, p->d_biglist, p->d_bigcount, (XYZZ28L<Fp28<C>>*)p->d_state28,
                    MLHIP_SEG_FIRST | MLHIP_SEG_LAST | MLHIP_SEG_KEEP28, (X*)p->d_buckets);
              done28 = true;
            }
          }
          if constexpr (kBuildAlt) {  // (MLHIP_REDUCE32=1: the buckets leave in the boundary form)
            if (!done28)
              k_accumulate28_lp<C><<<dim3((unsigned)((2 * nbuckets + p->acc_block - 1) / p->acc_block)), dim3(p->acc_block), 0, st>>>(
                  (const AffineG2_28<C>*)p->d_points28, p->d_sorted, p->d_offsets, p->d_counts, nbuckets, p->d_order,
                  big_threshold, p->d_biglist, p->d_bigcount, (X*)p->d_buckets);
          }
          done28 = true;
        }
      }
      if constexpr (kBuildAlt) {  // (MLHIP_ACC32=1)
        if (!done28)
          k_accumulate_lp<C><<<dim3((unsigned)((2 * nbuckets + p->acc_block - 1) / p->acc_block)), dim3(p->acc_block), 0, st>>>(
              (const A*)d_points, p->d_sorted, p->d_offsets, p->d_counts, nbuckets, p->d_order, big_threshold, p->d_biglist,
              p->d_bigcount, (X*)p->d_buckets);
      }
    } else if (p->d_points28) {
      HIPCHK(hipStreamWaitEvent(st, p->ev_join, 0));
      if (p->reduce28 && nbuckets <= QUAD_ACC_MAX_BUCKETS && !getenv("MLHIP_NO_QUAD_ACC"))
        // too few buckets to fill the chip with one lane each: one bucket per quad of lanes (shorter dependent chains)
        k_accumulate_q28<C><<<dim3((unsigned)((4 * nbuckets + 255) / 256)), dim3(256), 0, st>>>(
            (const Affine28<C>*)p->d_points28, p->d_sorted, p->d_offsets, p->d_counts, nbuckets, big_threshold, p->d_biglist,
            p->d_bigcount, (XYZZ28<C>*)p->d_state28);
      else if (use_ed) {  // (implies reduce28) the same single segment in twisted Edwards coordinates
        if constexpr (C::HAS_EDWARDS)
          k_accumulate_ed28_seg<C><<<dim3((unsigned)((nbuckets + p->acc_block - 1) / p->acc_block)), dim3(p->acc_block), 0, st>>>(
              (const EdNiels28<C>*)p->d_points28, p->d_sorted, p->d_offsets, p->d_counts, nbuckets, p->d_order, big_threshold,
              p->d_biglist, p->d_bigcount, (XYZZ28<C>*)p->d_state28, MLHIP_SEG_FIRST | MLHIP_SEG_LAST);
      } else if (p->reduce28)  // one segment that is first and last, leaving the raw accumulators for k_chunks_q28
        k_accumulate28_seg<C><<<dim3((unsigned)((nbuckets + p->acc_block - 1) / p->acc_block)), dim3(p->acc_block), 0, st>>>(
            (const Affine28<C>*)p->d_points28, p->d_sorted, p->d_offsets, p->d_counts, nbuckets, p->d_order, big_threshold,
            p->d_biglist, p->d_bigcount, (XYZZ28<C>*)p->d_state28, MLHIP_SEG_FIRST | MLHIP_SEG_LAST | MLHIP_SEG_KEEP28,
            (X*)p->d_buckets);
      else if constexpr (kBuildAlt)  // (MLHIP_REDUCE32=1 / MLHIP_REDUCE_ONE_LANE=1: the buckets leave in the boundary form)
        k_accumulate28<C><<<dim3((unsigned)((nbuckets + p->acc_block - 1) / p->acc_block)), dim3(p->acc_block), 0, st>>>(
            (const Affine28<C>*)p->d_points28, p->d_sorted, p->d_offsets, p->d_counts, nbuckets, p->d_order, big_threshold,
            p->d_biglist, p->d_bigcount, (X*)p->d_buckets);
    } else if constexpr (kBuildAlt) {  // (MLHIP_ACC32=1)
      k_accumulate<F><<<dim3((unsigned)((nbuckets + p->acc_block - 1) / p->acc_block)), dim3(p->acc_block), 0, st>>>(
          (const A*)d_points, p->d_sorted, p->d_offsets, p->d_counts, nbuckets, p->d_order, big_threshold, p->d_biglist,
          p->d_bigcount, (X*)p->d_buckets);
    }
    if (prof) HIPCHK(hipEventRecord(p->ev[3], st));
    if ((size_t)(p->fold ? p->Wd : 1) * n > big_threshold) {  // a bucket holds at most all n entries (Wd n when folded): small MSMs skip three near-empty launches
      constexpr int BB = sizeof(X) <= 192 ? 256 : 128;  // 48 KB of LDS per block
      launch_big_slices<F, BB>(p, (const A*)d_points, st);
      bool folded = false;
      if constexpr (!kLanePairs) {
        if constexpr (C::HAS_EDWARDS) {
          if (use_ed) {
            k_accumulate_big_seg_ed<C, BB><<<dim3(256), dim3(BB), BB * sizeof(X), st>>>(
                p->d_biglist, p->d_bigcount, p->d_bigprefix, (const X*)p->d_bigpart, (XYZZ28<C>*)p->d_state28,
                MLHIP_SEG_FIRST | MLHIP_SEG_LAST);
            folded = true;
          }
        }
        if (!folded && p->reduce28) {
          k_accumulate_big_seg<C, BB><<<dim3(256), dim3(BB), BB * sizeof(X), st>>>(
              p->d_biglist, p->d_bigcount, p->d_bigprefix, (const X*)p->d_bigpart, (XYZZ28<C>*)p->d_state28,
              MLHIP_SEG_FIRST | MLHIP_SEG_LAST | MLHIP_SEG_KEEP28, (X*)p->d_buckets);
          folded = true;
        }
      } else if constexpr (g2_carry_free_v<C>) {
        if (p->reduce28 && p->d_points28) {
          k_accumulate_big_seg_g2<C, BB><<<dim3(256), dim3(BB), BB * sizeof(X), st>>>(
              p->d_biglist, p->d_bigcount, p->d_bigprefix, (const X*)p->d_bigpart, (XYZZ28L<Fp28<C>>*)p->d_state28,
              MLHIP_SEG_FIRST | MLHIP_SEG_LAST | MLHIP_SEG_KEEP28, (X*)p->d_buckets);
          folded = true;
        }
      }
      if constexpr (kBuildAlt) {  // (boundary-form buckets)
        if (!folded)
          k_accumulate_big<F, BB><<<dim3(256), dim3(BB), BB * sizeof(X), st>>>(p->d_biglist, p->d_bigcount, p->d_bigprefix,
                                                                                (const X*)p->d_bigpart, (X*)p->d_buckets);
      }
    }
    {
      int rc_red = launch_reduce<C, F>(p, st);
      if (rc_red) return rc_red;
    }
    if (prof) HIPCHK(hipEventRecord(p->ev[4], st));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(p->h_out, p->d_out, ((size_t)p->W * p->nsel + p->fold_nsel2) * sizeof(X), hipMemcpyDeviceToHost, st));
    HIPCHK(hipEventRecord(p->done, st));
  }
  return 0;
}

// Host-buffer G1 MSM streamed in K segments (see k_accumulate28_seg): h_points / h_scalars are the caller's pageable
// buffers, d_points / d_scalars the plan-sized device buffers they are staged through.  Uploads and the point
// conversion ride the auxiliary stream; the sort and the accumulation of segment s wait for its event on `st`.
// The host thread blocks inside the pageable copies, which is exactly what overlaps them with the kernels queued before.
//
// The same train serves device-resident inputs cut into TILES (h_scalars == nullptr: the scalars are already at
// d_scalars; h_points == nullptr: the points are already at d_points and are converted tile by tile unless the plan
// holds their carry-free copy).  Every window gathers its points in a different random order, so an MSM re-reads the
// whole point array once per window; once that array outgrows what the chip keeps close (Infinity Cache 256 MB, TLB
// reach) each gather goes to HBM and the additions wait: 0.139 ns per G1 addition at 2^20-2^21 points, 0.165 at 2^24
// (profiles/r02_tiles.txt).  A tile of 2^21-2^22 points keeps all W passes over its points near; the bucket
// accumulators travel through d_state28 between tiles (0.2-0.5 GB per tile, streamed once).
// one streamed / tiled MSM in flight on a plan: what stream_begin fixes for its tiles
struct StreamCtx {
  void* d_points = nullptr;
  void* d_scalars = nullptr;
  const void* h_points = nullptr;
  const void* h_scalars = nullptr;
  int mont = 0;
  size_t n = 0, seg = 0;  // seg: the longest segment (what a sort-ahead helper record must hold)
  int K = 0;
  size_t bound[MLHIP_MAX_SEGMENTS + 1] = {};  // segment s = pairs [bound[s], bound[s + 1])
  bool scheduled = false;                     // bound[] was filled by the caller (stream_schedule); else K equal segments
  bool resident = false, conv_cached = false, prof = false;
  bool ed = false;  // the buckets are summed in twisted Edwards coordinates (plan_use_edwards)
};

template <class C, class F>
int stream_begin(mlhip_msm_plan* p, StreamCtx& cx, hipStream_t st, int min_K) {
  constexpr bool kG2 = std::is_same<F, Fp2Field<C>>::value;
  if (!p->aux || !p->d_points28)
    return mlhip_rt::fail(MLHIP_EINVAL, "streamed MSM needs the auxiliary stream and the carry-free path");
  constexpr size_t kStateBytes = kG2 ? 2 * sizeof(XYZZ28L<Fp28<C>>) : sizeof(XYZZ28<C>);
  if (cx.n == 0 || cx.K < min_K || cx.K > MLHIP_MAX_SEGMENTS) return mlhip_rt::fail(MLHIP_EINVAL, "bad segment count");
  const size_t nbuckets = (size_t)p->W * p->M;
  if (!p->d_state28) HIPCHK(hipMalloc(&p->d_state28, nbuckets * kStateBytes));
  for (int s = 0; s < cx.K; s++) {
    if (!p->ev_seg[s]) HIPCHK(hipEventCreateWithFlags(&p->ev_seg[s], hipEventDisableTiming));
    if (cx.h_scalars && cx.h_points && !p->ev_seg_sc[s]) HIPCHK(hipEventCreateWithFlags(&p->ev_seg_sc[s], hipEventDisableTiming));
  }
  // resident points (h_points == nullptr): only the scalars travel (or nothing: h_scalars == nullptr); their carry-free
  // copy is either the plan's (resident bases) or made tile by tile
  cx.resident = cx.h_points == nullptr;
  cx.ed = p->fold ? p->conv_ed : plan_use_edwards<C, F>(p);
  p->last_ed = cx.ed;
  cx.conv_cached = cx.resident && p->points_static && p->conv_src == cx.d_points && cx.n <= p->conv_n && p->conv_ed == cx.ed;
  if (p->fold) {
    if (!cx.resident || cx.n > p->fold_n || !p->d_points28)
      return mlhip_rt::fail(MLHIP_EINVAL, "folded plan: the points are the plan's table; more scalars than tabulated bases");
    cx.conv_cached = true;
    cx.d_points = p->d_points28;  // (a token: never read as Affine<F>)
  }
  cx.prof = p->profiling && cx.h_scalars == nullptr;  // tiles of device-resident inputs: per-tile phase events
  if (cx.prof)
    for (int s = 0; s < cx.K; s++)
      for (int j = 0; j < 3; j++)
        if (!p->ev_tile[s][j]) HIPCHK(hipEventCreate(&p->ev_tile[s][j]));
  p->tiles_timed = cx.prof ? cx.K : -1;  // -1: a streamed host-buffer MSM records no phase events
  p->pending_n = cx.n;
  p->pending = true;
  if (!cx.conv_cached) p->conv_src = nullptr;  // the carry-free copy is being rewritten
  if (!cx.scheduled) {
    const size_t seg = (cx.n + cx.K - 1) / cx.K;
    int k = 0;
    for (size_t off = 0; off < cx.n; off += seg) cx.bound[k++] = off;
    cx.bound[k] = cx.n;
    cx.K = k;
  }
  if (p->fold) {
    // no segment may cross a tile of the table: cut at the tile boundaries; if that makes too many segments, fall back to
    // the tiles themselves
    if ((cx.n + p->fold_tile - 1) / p->fold_tile > MLHIP_MAX_SEGMENTS) return mlhip_rt::fail(MLHIP_EINVAL, "folded plan: too many tiles");
    size_t b[2 * MLHIP_MAX_SEGMENTS + 2];
    int m = 0;
    b[0] = 0;
    for (int s2 = 0; s2 < cx.K; s2++) {
      const size_t hi = cx.bound[s2 + 1];
      for (size_t t = (b[m] / p->fold_tile + 1) * p->fold_tile; t < hi; t += p->fold_tile) b[++m] = t;
      b[++m] = hi;
    }
    if (m > MLHIP_MAX_SEGMENTS) {
      m = 0;
      for (size_t t = p->fold_tile; t < cx.n; t += p->fold_tile) b[++m] = t;
      b[++m] = cx.n;
      if (m > MLHIP_MAX_SEGMENTS) return mlhip_rt::fail(MLHIP_EINVAL, "folded plan: too many tiles");
    }
    for (int s2 = 0; s2 <= m; s2++) cx.bound[s2] = b[s2];
    cx.K = m;
    for (int s2 = 0; s2 < cx.K; s2++) {  // (events of the segments the cut added)
      if (!p->ev_seg[s2]) HIPCHK(hipEventCreateWithFlags(&p->ev_seg[s2], hipEventDisableTiming));
      if (cx.prof)
        for (int j = 0; j < 3; j++)
          if (!p->ev_tile[s2][j]) HIPCHK(hipEventCreate(&p->ev_tile[s2][j]));
    }
    if (cx.prof) p->tiles_timed = cx.K;
  }
  cx.seg = 0;
  for (int s2 = 0; s2 < cx.K; s2++) cx.seg = std::max(cx.seg, cx.bound[s2 + 1] - cx.bound[s2]);
  HIPCHK(hipEventRecord(p->ev_fork, st));  // the staging buffers are free once the work queued before us is done
  HIPCHK(hipStreamWaitEvent(p->aux, p->ev_fork, 0));
  return 0;
}

// tile s = pairs [off, off + len): uploads / conversion on the auxiliary stream, then sort and accumulation on `st`.
// `sorter` != nullptr: the tile was already sorted in ANOTHER plan of the same window geometry over the same scalars
// (mlhip_msm_launch_shared: the G1 and the G2 MSM of one scalar vector) -- its entry lists are read, nothing is sorted.
template <class C, class F>
int stream_tile(mlhip_msm_plan* p, const StreamCtx& cx, int s, hipStream_t st, const mlhip_msm_plan* sorter) {
  typedef Affine<F> A;
  typedef XYZZ<F> X;
  constexpr bool kG2 = std::is_same<F, Fp2Field<C>>::value;
  const size_t nbuckets = (size_t)p->W * p->M;
  const size_t off = cx.bound[s];
  const size_t len = cx.bound[s + 1] - off;
  const bool first = off == 0, last = off + len >= cx.n;
  const int flags = (first ? MLHIP_SEG_FIRST : 0) | (last ? MLHIP_SEG_LAST : 0) | (p->reduce28 ? MLHIP_SEG_KEEP28 : 0);
  const char* hp = (const char*)cx.h_points;
  const char* hs = (const char*)cx.h_scalars;
  char* dsc = (char*)cx.d_scalars + off * 32;
  // where this segment's points start: plain arrays at `off`; a folded plan's table at the tile block that holds base `off`
  const size_t row0 = p->fold ? fold_row(p, off) : off;
  A* dpt = p->fold ? nullptr : (A*)cx.d_points + row0;  // (folded: the slices of long buckets gather from d_points28 + row0)
  const bool prof = cx.prof;
  if (hs) HIPCHK(hipMemcpyAsync(dsc, hs + off * 32, len * 32, hipMemcpyHostToDevice, p->aux));
  // points and scalars both travel: the sort starts when the segment's scalars are there, under the upload of its points
  const bool split = hs && !cx.resident && p->ev_seg_sc[s];
  if (split) HIPCHK(hipEventRecord(p->ev_seg_sc[s], p->aux));
  if (!cx.conv_cached) {
    if (!cx.resident) HIPCHK(hipMemcpyAsync(dpt, hp + off * sizeof(A), len * sizeof(A), hipMemcpyHostToDevice, p->aux));
    if constexpr (kG2)
      k_points_to28_g2<C><<<dim3((unsigned)((4 * len + 255) / 256)), dim3(256), 0, p->aux>>>(
          dpt, len, (AffineG2_28<C>*)p->d_points28 + off);
    else {
      bool converted = false;
      if constexpr (C::HAS_EDWARDS) {
        if (cx.ed) {
          k_points_to_ed28<C><<<dim3((unsigned)(((len + 3) / 4 + 255) / 256)), dim3(256), 0, p->aux>>>(
              dpt, len, (EdNiels28<C>*)p->d_points28 + off);
          converted = true;
        }
      }
      if (!converted)
        k_points_to28<C><<<dim3((unsigned)((len + 255) / 256)), dim3(256), 0, p->aux>>>(dpt, len,
                                                                                       (Affine28<C>*)p->d_points28 + off);
    }
  }
  HIPCHK(hipEventRecord(p->ev_seg[s], p->aux));
  if (prof && first) HIPCHK(hipEventRecord(p->ev[0], st));
  HIPCHK(hipMemsetAsync(p->d_zero, 0, p->zero_bytes, st));
  if (prof) HIPCHK(hipEventRecord(p->ev_tile[s][0], st));
  const bool uploads = hs || !cx.resident;
  if (uploads) HIPCHK(hipStreamWaitEvent(st, split ? p->ev_seg_sc[s] : p->ev_seg[s], 0));  // the sort needs the uploaded scalars
  if (!sorter) {
    int rc_sort = launch_sort<C>(p, dsc, cx.mont, len, st, false);
    if (rc_sort) return rc_sort;
  }
  if (!uploads || split) HIPCHK(hipStreamWaitEvent(st, p->ev_seg[s], 0));  // only the accumulation waits for points / conversion
  if (prof) HIPCHK(hipEventRecord(p->ev_tile[s][1], st));
  const mlhip_msm_plan* sv = sorter ? sorter : p;  // whose entry lists the kernels read
  uint32_t big_threshold = (uint32_t)std::min<size_t>((((size_t)(p->fold ? p->Wd : 1) * len) >> (p->c - 1)) * 8, 1u << 30);
  if (big_threshold < BIG_BUCKET_MIN) big_threshold = BIG_BUCKET_MIN;
  if constexpr (kG2) {
    const dim3 grid((unsigned)((2 * nbuckets + p->acc_block - 1) / p->acc_block)), block(p->acc_block);
    bool kc = false;
    if constexpr (C::BETA == -1 && kBuildAlt) {
      if (g2_split_by_coordinate()) {
        k_accumulate28_kc_seg<C><<<grid, block, 0, st>>>(
            (const AffineG2_28<C>*)p->d_points28 + row0, sv->d_sorted, sv->d_offsets, sv->d_counts, nbuckets, sv->d_order,
            big_threshold, p->d_biglist, p->d_bigcount, (XYZZ28L<Fp28<C>>*)p->d_state28, flags, (X*)p->d_buckets);
        kc = true;
      }
    }
    if (!kc)
      k_accumulate28_lp_seg<C><<<grid, block, 0, st>>>(
          (const AffineG2_28<C>*)p->d_points28 + row0, sv->d_sorted, sv->d_offsets, sv->d_counts, nbuckets, sv->d_order,
          big_threshold, p->d_biglist, p->d_bigcount, (XYZZ28L<Fp28<C>>*)p->d_state28, flags, (X*)p->d_buckets);
    constexpr int BB = 128;
    launch_big_slices<F, BB>(p, dpt, st, sv, row0);
    k_accumulate_big_seg_g2<C, BB><<<dim3(256), dim3(BB), BB * sizeof(X), st>>>(
        p->d_biglist, p->d_bigcount, p->d_bigprefix, (const X*)p->d_bigpart, (XYZZ28L<Fp28<C>>*)p->d_state28, flags,
        (X*)p->d_buckets);
  } else {
    constexpr int BB = 256;
    bool done_ed = false;
    if constexpr (C::HAS_EDWARDS) {
      if (cx.ed) {  // (implies MLHIP_SEG_KEEP28: the last tile leaves XYZZ28 for the reduction)
        k_accumulate_ed28_seg<C><<<dim3((unsigned)((nbuckets + p->acc_block - 1) / p->acc_block)), dim3(p->acc_block), 0, st>>>(
            (const EdNiels28<C>*)p->d_points28 + row0, sv->d_sorted, sv->d_offsets, sv->d_counts, nbuckets, sv->d_order,
            big_threshold, p->d_biglist, p->d_bigcount, (XYZZ28<C>*)p->d_state28, flags);
        launch_big_slices<F, BB>(p, dpt, st, sv, row0);
        k_accumulate_big_seg_ed<C, BB><<<dim3(256), dim3(BB), BB * sizeof(X), st>>>(
            p->d_biglist, p->d_bigcount, p->d_bigprefix, (const X*)p->d_bigpart, (XYZZ28<C>*)p->d_state28, flags);
        done_ed = true;
      }
    }
    if (!done_ed) {
      k_accumulate28_seg<C><<<dim3((unsigned)((nbuckets + p->acc_block - 1) / p->acc_block)), dim3(p->acc_block), 0, st>>>(
          (const Affine28<C>*)p->d_points28 + row0, sv->d_sorted, sv->d_offsets, sv->d_counts, nbuckets, sv->d_order,
          big_threshold, p->d_biglist, p->d_bigcount, (XYZZ28<C>*)p->d_state28, flags, (X*)p->d_buckets);
      launch_big_slices<F, BB>(p, dpt, st, sv, row0);
      k_accumulate_big_seg<C, BB><<<dim3(256), dim3(BB), BB * sizeof(X), st>>>(
          p->d_biglist, p->d_bigcount, p->d_bigprefix, (const X*)p->d_bigpart, (XYZZ28<C>*)p->d_state28, flags,
          (X*)p->d_buckets);
    }
  }
  if (prof) HIPCHK(hipEventRecord(p->ev_tile[s][2], st));
  return 0;
}

template <class C, class F>
int stream_end(mlhip_msm_plan* p, const StreamCtx& cx, hipStream_t st) {
  typedef XYZZ<F> X;
  if (cx.resident && !cx.conv_cached && p->points_static) {  // every tile was converted: the copy is whole again
    p->conv_src = cx.d_points;
    p->conv_n = cx.n;
    p->conv_ed = cx.ed;
  }
  if (cx.prof) HIPCHK(hipEventRecord(p->ev[3], st));
  {
    int rc_red = launch_reduce<C, F>(p, st);
    if (rc_red) return rc_red;
  }
  if (cx.prof) HIPCHK(hipEventRecord(p->ev[4], st));
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(p->h_out, p->d_out, ((size_t)p->W * p->nsel + p->fold_nsel2) * sizeof(X), hipMemcpyDeviceToHost, st));
  HIPCHK(hipEventRecord(p->done, st));
  return 0;
}

// ---- sorting ahead (round 3) ----------------------------------------------------------------------------------------------
// A tiled device-resident MSM ran sort(0), accumulate(0), sort(1), accumulate(1) ... on one stream: the sort kernels are
// bound by LDS atomics and leave the multipliers idle, the accumulation is the opposite.  The entry lists of tile s + 1
// are now sorted on a second stream, in one of two helper records that own sort buffers only, while tile s accumulates
// on the caller's stream from the other record's lists (the `sorter` argument of stream_tile, as the G2 plan of a
// shared-scalar MSM reads the G1 plan's lists).  Events: ev_sorted[b] (sort stream -> accumulation), ev_lists_free[b]
// (accumulation -> the next sort into record b; it also orders consecutive MSMs on the plan).  MLHIP_SORT_AHEAD=0: off.
template <class C>
int sort_ahead_prepare(mlhip_msm_plan* p, size_t seg, bool& on) {
  on = false;
  const char* e = getenv("MLHIP_SORT_AHEAD");
  if (e && e[0] == '0') return 0;
  if (p->sort_low <= 0) return 0;  // the legacy sort shares the digits buffer with the accumulation path: not split
  for (int b = 0; b < 2; b++) {
    mlhip_msm_plan* h = p->sort_helper[b];
    if (h && h->max_n < seg) {  // longer tiles than last time: rebuild
      (void)hipStreamSynchronize(p->sort_stream);
      mlhip_msm_plan_destroy(h);
      h = p->sort_helper[b] = nullptr;
    }
    if (!h) {
      h = new mlhip_msm_plan();
      h->curve = p->curve;
      h->group = p->group;
      h->device = p->device;
      h->c = p->c;
      h->W = p->W;
      h->Wd = p->Wd;
      h->fold = p->fold;
      h->fold_tile = p->fold_tile;
      h->M = p->M;
      h->L = p->L;
      h->lgL = p->lgL;
      h->T = p->T;
      h->nb = p->nb;
      h->nsel = p->nsel;
      h->max_n = seg;
      // The helper is an optimisation: if its buffers do not fit (two copies of the tile's entry lists; plausible at
      // 2^24 pairs) the MSM must still run, with the sort in line as before.  The record is registered only once it is
      // complete -- a half-allocated one left in p->sort_helper would pass the `max_n >= seg` test of the next launch
      // and hand null list pointers to launch_sort.  MLHIP_FAULT_INJECT=sort_helper_alloc makes the allocation "fail"
      // (tests/test_gpu_parity.py::test_sort_ahead_helper_allocation_failure).
      const char* fi = getenv("MLHIP_FAULT_INJECT");
      int rc = (fi && !strcmp(fi, "sort_helper_alloc")) ? MLHIP_ENOMEM : plan_alloc_sort(h);
      if (rc || h->sort_low <= 0) {
        mlhip_msm_plan_destroy(h);  // frees whatever was allocated
        (void)hipGetLastError();    // an out-of-memory error must not surface in the next HIPCHK
        return 0;                   // on = false: the in-line path
      }
      p->sort_helper[b] = h;  // owned by p from here on
    }
    if (!p->ev_sorted[b]) HIPCHK(hipEventCreateWithFlags(&p->ev_sorted[b], hipEventDisableTiming));
    if (!p->ev_lists_free[b]) HIPCHK(hipEventCreateWithFlags(&p->ev_lists_free[b], hipEventDisableTiming));
  }
  if (!p->sort_stream) {
    // the highest priority: a sort is a chain of a dozen short kernels that must find wave slots between the
    // accumulation's one-wave workgroups
    int lo = 0, hi = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    const char* pe = getenv("MLHIP_SORT_AHEAD_PRIO");
    if (pe && pe[0] == '0')
      HIPCHK(hipStreamCreateWithFlags(&p->sort_stream, hipStreamNonBlocking));
    else
      HIPCHK(hipStreamCreateWithPriority(&p->sort_stream, hipStreamNonBlocking, hi));
  }
  on = true;
  return 0;
}
// queue the sort of tile s into helper record s & 1 (after whatever still reads that record's lists)
template <class C>
int sort_ahead_tile(mlhip_msm_plan* p, const StreamCtx& cx, int s) {
  const int b = s & 1;
  mlhip_msm_plan* h = p->sort_helper[b];
  const size_t off = cx.bound[s];
  const size_t len = cx.bound[s + 1] - off;
  HIPCHK(hipStreamWaitEvent(p->sort_stream, p->ev_lists_free[b], 0));  // never recorded yet: no wait
  HIPCHK(hipMemsetAsync(h->d_zero, 0, h->zero_bytes, p->sort_stream));
  int rc = launch_sort<C>(h, (const char*)cx.d_scalars + off * 32, cx.mont, len, p->sort_stream, false);
  if (rc) return rc;
  HIPCHK(hipEventRecord(p->ev_sorted[b], p->sort_stream));
  return 0;
}

// Segment schedule of a host-buffer MSM (round 4; profiles/r04_hostapi.txt, same-box A/Bs at 2^20 pairs).  K equal
// segments expose the whole first upload and pay the per-segment costs (a sort train, one round trip of the bucket state,
// two pageable copies, shorter bucket lists) K times.  What bounds the call differs between the two host protocols:
//   (b) resident bases, only the 32-byte scalars travel -- a fifth of the kernels' time: TWO segments, 3 and 13 sixteenths of
//       the call; the second upload hides under the first segment's kernels and only one extra sort train is paid:
//       3.37 -> 3.23 ms against four equal segments, 0.02 ms above the resident MSM of the same box;
//   (c) points and scalars travel, 128 B a pair -- the copies (2.35 ms) run at about the kernels' rate (2.6 ms), so a later
//       segment may be at most ~1.1x the one before or the kernels wait for it, and every extra segment costs ~0.1 ms:
//       growing schedules LOSE (1,1,2,3,4,5: 4.28 ms; 2,3,5,6: 4.24; 1,2,2,3,4,4: 4.09) against four equal segments (3.99) --
//       equal segments of 2^18 pairs stay.  (What round 4 did gain for (c) is the split event: a segment's sort starts on
//       its scalars, under the upload of its points -- stream_tile.)
// MLHIP_STREAM_SCHEDULE="w0,w1,..." (weights, at most MLHIP_MAX_SEGMENTS) overrides; MLHIP_STREAM_SEGMENTS = K keeps K equal
// segments (what the tests use to force many segments on small inputs).
inline void stream_schedule(StreamCtx& cx, bool points_travel, size_t tile, bool fold_tiles = false) {
  int w[MLHIP_MAX_SEGMENTS];
  int k = 0;
  if (const char* e = getenv("MLHIP_STREAM_SCHEDULE")) {
    for (const char* q = e; *q && k < MLHIP_MAX_SEGMENTS;) {
      const int v = atoi(q);
      if (v > 0) w[k++] = v;
      while (*q && *q != ',') q++;
      if (*q == ',') q++;
    }
  } else if (getenv("MLHIP_STREAM_SEGMENTS")) {
    return;  // K equal segments
  } else if (fold_tiles && cx.n > tile && !points_travel) {
    // a folded plan with several tiles: 3 x 2^16 pairs, the rest of the first tile, then the tiles (a segment cannot cross one)
    const size_t first = (size_t)3 << 16;
    int m = 0;
    cx.bound[0] = 0;
    if (tile > 2 * first) cx.bound[++m] = first;
    for (size_t t = tile; t < cx.n && m + 1 < MLHIP_MAX_SEGMENTS; t += tile) cx.bound[++m] = t;
    cx.bound[++m] = cx.n;
    cx.K = m;
    cx.scheduled = true;
    return;
  } else if (cx.n >= ((size_t)1 << 20) && !points_travel) {
    // resident bases: 3 x 2^16 pairs first, then segments that grow fourfold (the scalars of the next segment -- 0.6 ns a
    // pair on the wire -- must arrive within the kernels of this one -- 2.5 ns a pair) up to one tile (2^21 pairs: what keeps
    // a segment's W passes over its points near the chip, resident_tiles); a remainder shorter than the first segment joins
    // the segment before it.  2^20: 3 | 13 sixteenths; 2^21: 0.19 | 0.75 | 1.06 M; 2^22: 0.19 | 0.75 | 2.0 | 1.06 M.
    const size_t first = (size_t)3 << 16;
    size_t off = 0, len = first;
    int m = 0;
    cx.bound[0] = 0;
    while (off < cx.n && m < MLHIP_MAX_SEGMENTS) {
      size_t take = std::min(len, cx.n - off);
      if (cx.n - off - take < first || m + 1 == MLHIP_MAX_SEGMENTS) take = cx.n - off;  // no crumb at the end
      off += take;
      cx.bound[++m] = off;
      len = std::min(len * 4, tile);
    }
    if (m >= 2) {
      cx.K = m;
      cx.scheduled = true;
    }
    return;
  }
  if (k < 2) return;
  long long total = 0;
  for (int i = 0; i < k; i++) total += w[i];
  size_t cum = 0;
  int m = 0;
  cx.bound[0] = 0;
  for (int i = 0; i < k; i++) {
    cum += (size_t)w[i];
    size_t b = i + 1 == k ? cx.n : ((size_t)((unsigned __int128)cx.n * cum / (size_t)total) + 1023) / 1024 * 1024;
    if (b > cx.n) b = cx.n;
    if (b > cx.bound[m]) cx.bound[++m] = b;
  }
  if (m < 2) return;
  cx.K = m;
  cx.scheduled = true;
}

template <class C, class F>
int plan_stream(mlhip_msm_plan* p, void* d_points, void* d_scalars, const void* h_points, const void* h_scalars, int mont,
                size_t n, int K, hipStream_t st) {
  StreamCtx cx;
  cx.d_points = d_points;
  cx.d_scalars = d_scalars;
  cx.h_points = h_points;
  cx.h_scalars = h_scalars;
  cx.mont = mont;
  cx.n = n;
  cx.K = K;
  if (h_scalars && std::is_same<F, FpField<C>>::value)
    stream_schedule(cx, h_points != nullptr,
                    p->fold ? p->fold_tile : (size_t)1 << (plan_use_edwards<C, F>(p) ? 20 : 21), p->fold != 0);  // the tile of resident_tiles
  int rc = stream_begin<C, F>(p, cx, st, 2);
  if (rc) return rc;
  bool ahead = false;
  if (!h_scalars && !h_points) {  // device-resident tiles: nothing to upload, the sorts can run ahead
    rc = sort_ahead_prepare<C>(p, cx.seg, ahead);
    if (rc) return rc;
  }
  if (ahead) HIPCHK(hipStreamWaitEvent(p->sort_stream, p->ev_fork, 0));  // the scalars are ready where `st` stood at launch
  for (int s = 0; s < cx.K; s++) {
    if (ahead) {
      rc = sort_ahead_tile<C>(p, cx, s);
      if (rc) return rc;
      HIPCHK(hipStreamWaitEvent(st, p->ev_sorted[s & 1], 0));
    }
    rc = stream_tile<C, F>(p, cx, s, st, ahead ? p->sort_helper[s & 1] : nullptr);
    if (rc) return rc;
    if (ahead) HIPCHK(hipEventRecord(p->ev_lists_free[s & 1], st));
  }
  return stream_end<C, F>(p, cx, st);
}

// The G1 MSM and the G2 MSM of ONE scalar vector (device-resident inputs; BASELINE configs[3], a Groth16 prover's
// B-query): the entry lists of a tile depend on the scalars and the window geometry only, so every tile is sorted once
// (in the G1 plan) and accumulated twice.  One stream, tile by tile: sort, G1 accumulation, G2 accumulation.
// Host buffers (h_* != nullptr: the caller's pageable memory, staged through d_*): the scalars and the G1 points of a
// tile travel on the G1 plan's auxiliary stream, the G2 points on the G2 plan's, under the kernels of the tile before.
template <class C>
int plan_stream_shared(mlhip_msm_plan* p1, mlhip_msm_plan* p2, void* d_points_g1, void* d_points_g2, void* d_scalars,
                       const void* h_points_g1, const void* h_points_g2, const void* h_scalars, int mont, size_t n,
                       hipStream_t st) {
  typedef FpField<C> F1;
  typedef Fp2Field<C> F2;
  if (p1->c != p2->c || p1->W != p2->W || p1->M != p2->M)
    return mlhip_rt::fail(MLHIP_EINVAL, "the two plans of a shared-scalar MSM need the same window width");
  if (p1->fold || p2->fold) return mlhip_rt::fail(MLHIP_EINVAL, "shared-scalar MSM: plans with shifted-base tables are not supported");
  int K = 1;
  if (h_scalars) {
    // uploads to hide: segments of 2^17 pairs, as a host-buffer G2 MSM (api.hip: stream_segments)
    K = (int)std::min<size_t>(std::max<size_t>(n >> 17, 1), MLHIP_MAX_SEGMENTS);
    if (const char* e = getenv("MLHIP_STREAM_SEGMENTS")) {
      const int v = atoi(e);
      K = v < 2 ? 1 : (int)std::min<size_t>(std::min<size_t>((size_t)v, n), MLHIP_MAX_SEGMENTS);
    }
  } else {
    // tiles of 2^20 pairs from 2^22 on (see resident_tiles: G1 gains from 2^22, G2 from 2^23, neither loses)
    if (n >= ((size_t)1 << 22)) K = (int)std::min<size_t>((n + ((size_t)1 << 20) - 1) >> 20, MLHIP_MAX_SEGMENTS);
    if (const char* e = getenv("MLHIP_TILE_LOG2")) {
      const int v = atoi(e);
      K = 1;
      if (v > 0 && v < 31 && n > ((size_t)1 << v)) K = (int)std::min<size_t>((n + ((size_t)1 << v) - 1) >> v, MLHIP_MAX_SEGMENTS);
    }
  }
  StreamCtx c1, c2;
  c1.d_points = d_points_g1;
  c2.d_points = d_points_g2;
  c1.d_scalars = c2.d_scalars = d_scalars;
  c1.h_points = h_points_g1;
  c2.h_points = h_points_g2;
  c1.h_scalars = h_scalars;  // uploaded once, by the G1 plan's tiles
  c1.mont = c2.mont = mont;
  c1.n = c2.n = n;
  c1.K = c2.K = K;
  int rc = stream_begin<C, F1>(p1, c1, st, 1);
  if (rc) return rc;
  rc = stream_begin<C, F2>(p2, c2, st, 1);
  if (rc) return rc;
  bool ahead = false;
  if (!h_scalars && !h_points_g1 && !h_points_g2 && K >= 2) {
    rc = sort_ahead_prepare<C>(p1, c1.seg, ahead);
    if (rc) return rc;
  }
  if (ahead) HIPCHK(hipStreamWaitEvent(p1->sort_stream, p1->ev_fork, 0));
  for (int s = 0; s < c1.K; s++) {
    const mlhip_msm_plan* lists = p1;  // whose entry lists the G2 tile reads
    if (ahead) {
      rc = sort_ahead_tile<C>(p1, c1, s);
      if (rc) return rc;
      HIPCHK(hipStreamWaitEvent(st, p1->ev_sorted[s & 1], 0));
      lists = p1->sort_helper[s & 1];
    }
    rc = stream_tile<C, F1>(p1, c1, s, st, ahead ? lists : nullptr);
    if (rc) return rc;
    if (s + 1 == c1.K) {
      // G1 is complete: its reduction and its copy to the host go ahead of G2's last accumulation, so that the host
      // tail of the G1 result runs under it
      rc = stream_end<C, F1>(p1, c1, st);
      if (rc) return rc;
    }
    rc = stream_tile<C, F2>(p2, c2, s, st, lists);
    if (rc) return rc;
    if (ahead) HIPCHK(hipEventRecord(p1->ev_lists_free[s & 1], st));
  }
  return stream_end<C, F2>(p2, c2, st);
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
    if (p->profiling && p->tiles_timed > 0) {
      // tiles: digits are part of each tile's sort; accumulate = the sum over the tiles' accumulation kernels
      p->ms[0] = p->ms[1] = p->ms[2] = 0;
      for (int s = 0; s < p->tiles_timed; s++) {
        float a = 0, b = 0;
        HIPCHK(hipEventElapsedTime(&a, p->ev_tile[s][0], p->ev_tile[s][1]));
        HIPCHK(hipEventElapsedTime(&b, p->ev_tile[s][1], p->ev_tile[s][2]));
        p->ms[1] += a;
        p->ms[2] += b;
      }
      HIPCHK(hipEventElapsedTime(&p->ms[3], p->ev[3], p->ev[4]));
      HIPCHK(hipEventElapsedTime(&p->ms[4], p->ev[0], p->ev[4]));
      p->ms[5] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    } else if (p->profiling && p->tiles_timed == 0) {
      for (int i = 0; i < 4; i++) HIPCHK(hipEventElapsedTime(&p->ms[i], p->ev[i], p->ev[i + 1]));
      HIPCHK(hipEventElapsedTime(&p->ms[4], p->ev[0], p->ev[4]));
      p->ms[5] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    } else if (p->profiling) {
      for (int i = 0; i < 6; i++) p->ms[i] = 0;  // a streamed host-buffer MSM records no phase events: nothing stale is left
    }
  }
  A r;
  xyzz_to_affine<F>(r, total);
  memcpy(out_affine, &r, sizeof(A));
  if (out_xyzz) memcpy(out_xyzz, &total, sizeof(X));
  return 0;
}

}  // namespace mlhip
