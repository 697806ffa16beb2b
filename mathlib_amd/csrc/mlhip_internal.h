// mlhip_internal.h -- shared declarations of libmlhip.so's translation units (one per curve and
// kernel family, so the build parallelises; see mathlib_amd/build.py).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <string>

#include "../../include/mlhip.h"

namespace mlhip_rt {
int fail(int code, const std::string& msg);
// out[j] = fn(in, j) for j = 0 .. njobs - 1 on the library's host worker threads and the calling thread; returns when
// all are there.  fn must be a pure function of the `in_bytes` bytes at `in` (the call copies them: a worker never touches
// the caller's memory) writing `out_stride` bytes.  The workers are started on first use (MLHIP_HOST_THREADS = threads per
// call incl. the caller, default min(8, cores); 1 = none); a call that finds them busy with another caller's jobs, or
// njobs < 2, computes everything itself, and a job a worker is slow with is computed by the caller as well (api.hip).
// Used by the host tail of an MSM (msm_plan.h: host_tail): the per-window sums are independent of each other.
void host_parallel(int njobs, void (*fn)(const void*, int, void*), const void* in, size_t in_bytes, void* out, size_t out_stride);
}

#define HIPCHK(x)                                                                                             \
  do {                                                                                                        \
    hipError_t e_ = (x);                                                                                      \
    if (e_ != hipSuccess) return mlhip_rt::fail(MLHIP_EHIP, std::string(#x) + ": " + hipGetErrorString(e_)); \
  } while (0)

#define MLHIP_MAX_SEGMENTS 24

// MLHIP_BUILD_ALT=1 (python -m mathlib_amd.build --alt -> libmlhip_alt.so): the test build.  It also contains the second
// implementations the parity tests compare the default kernels with -- boundary-form (32-bit limb) bucket accumulation and
// reduction, the one-point-per-lane reduction, G2 buckets split by coordinate, one-lane / saturated-limb pairing and G2.Mul
// kernels -- behind their MLHIP_* switches (include/mlhip.h, "2nd impl").  The product library is built without it: those
// kernels are not instantiated, their switches are ignored, and mlhip_version() has bit 16 clear.
#ifndef MLHIP_BUILD_ALT
#define MLHIP_BUILD_ALT 0
#endif
constexpr bool kBuildAlt = MLHIP_BUILD_ALT != 0;
// an alternate-implementation switch: set to '1' AND compiled in
static inline bool mlhip_alt_switch(const char* name) {
  if (!kBuildAlt) return false;
  const char* e = getenv(name);
  return e && e[0] == '1';
}

struct mlhip_msm_plan {
  int curve, group, device, c, W, L, lgL, nb, nsel;
  size_t max_n;
  uint32_t M, T;
  // Shifted-base tables (msm_fold.h; resident bases only).  Wd = digits per scalar.  A plain plan has one bucket set per
  // digit position: Wd == W windows of M = 2^(c-1) buckets.  A FOLDED plan (fold != 0) has ONE set of 2^(c-1) buckets for
  // all Wd digits -- digit j of scalar i adds row (j, i) = 2^off(j) P_i of the table (off(j) = first bit of digit j) instead of P_i -- cut into W groups of
  // M consecutive buckets for the reduction kernels; the host tail then has lg M doublings instead of one per scalar bit,
  // the reduction 2^(c-1) buckets instead of Wd 2^(c-1), and c can grow (fewer digits = fewer additions).
  // The table is tile-major: bases [k fold_tile, (k + 1) fold_tile) own rows [(k Wd + j) fold_tile + i_local], so that an
  // entry index (i_local + j fold_tile) stays below Wd fold_tile whatever the number of bases; a sorted segment of scalars
  // never crosses a tile.
  int Wd = 0, fold = 0;
  size_t fold_tile = 0, fold_n = 0;  // rows per digit block; bases tabulated
  size_t fold_rows = 0;              // rows allocated in d_points28 (tiles x Wd x fold_tile): carry-free rows or Niels triples
  // the W groups' sums are combined on the device into the sums of one window of W T chunks (k_group_combine_q / _lp);
  // d_out / h_out then hold fold_nsel2 = 4 + nb + lg W more entries behind the W x nsel ones, and the host tail reads those
  int fold_nsel2 = 0;
  size_t pt_size, xyzz_size;
  uint32_t *d_digits = nullptr, *d_sorted = nullptr, *d_zero = nullptr, *d_offsets = nullptr, *d_biglist = nullptr;
  int sort_low = 0, sort_idx_bits = 0;  // two-level sort: fine bits per coarse bin (0 = legacy path)
  uint32_t sort_nb = 0;
  uint32_t *d_coarse_count = nullptr, *d_coarse_cursor = nullptr, *d_coarse_off = nullptr;
  uint16_t* d_blockhist = nullptr;  // per-block coarse histograms (k_coarse_hist -> k_coarse_scatter)
  uint32_t *d_order = nullptr, *d_hist = nullptr, *d_tilesums = nullptr;
  uint32_t *d_counts = nullptr, *d_cursor = nullptr, *d_bigcount = nullptr;  // views into d_zero
  size_t zero_bytes = 0;
  void *d_buckets = nullptr, *d_A = nullptr, *d_W0 = nullptr, *d_out = nullptr;
  void* h_out = nullptr;
  uint32_t* d_binprefix = nullptr;  // coarse bins sorted by several workgroups: slice prefix
  uint32_t* d_bigprefix = nullptr;  // long buckets: slice counts (prefix) and slice sums
  void* d_bigpart = nullptr;
  void* d_points28 = nullptr;  // G1: the points in the carry-free 28-bit-limb form (ec28.h), rewritten every MSM
  // bytes per point d_points28 was allocated with, and what the twisted Edwards form (Niels triples, ed28.h) would need
  // (0: this curve / group has none).  A BLS12-377 G1 plan starts with the 112-byte Weierstrass rows and grows to 168 bytes
  // only when the caller makes the SRS promise (api.hip: plan_reserve_edwards) -- at 2^24 points that is 0.9 GB not spent
  // on plans that never take the Edwards path
  size_t points28_elem = 0, points28_elem_ed = 0;
  bool profiling = false;
  bool reduce_one_lane = false;
  bool reduce28 = false;  // G1: the reduction reads the carry-free bucket state (d_state28) -- k_chunks_q28 / k_masked_sums_q28
  int red_block = 256;  // workgroup size of k_chunks_q (MLHIP_RED_BLOCK)
  int acc_block = 64;  // workgroup size of the bucket-accumulation kernel (MLHIP_ACC_BLOCK)
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t done = nullptr;
  hipStream_t aux = nullptr;  // the point conversion runs here, beside the sort kernels
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // host-buffer entry points: points still in host memory, uploaded on `aux` beside the sort (cleared per launch)
  const void* upload_src = nullptr;
  size_t upload_bytes = 0;
  // resident bases (mlhip_bases_*): the library owns the point buffer, so its carry-free copy is made once
  bool points_static = false;
  const void* conv_src = nullptr;
  size_t conv_n = 0;
  bool conv_ed = false;  // ... and it holds the twisted Edwards form (ed28.h) rather than the Weierstrass one
  // the caller vouches that every point handed to this plan lies in the prime-order subgroup (or is the point at infinity):
  // curves with C::HAS_EDWARDS may then sum their G1 buckets in twisted Edwards coordinates (mlhip_msm_plan_assume_srs)
  bool trust_subgroup = false;
  bool last_ed = false;  // the last launch summed its buckets in twisted Edwards coordinates (mlhip_msm_plan_timings [9])
  // streamed host-buffer MSMs (plan_stream): raw carry-free bucket accumulators between segments, one event per segment
  void* d_state28 = nullptr;
  hipEvent_t ev_seg[MLHIP_MAX_SEGMENTS] = {};
  hipEvent_t ev_seg_sc[MLHIP_MAX_SEGMENTS] = {};  // ... and one after the segment's SCALARS alone (the sort needs only them)
  // tiles of device-resident inputs under profiling: before the sort / after it / after the accumulation of each tile
  hipEvent_t ev_tile[MLHIP_MAX_SEGMENTS][3] = {};
  int tiles_timed = 0;  // > 0: the last launch was tiled and recorded ev_tile[0 .. tiles_timed)
  bool pending = false;
  size_t pending_n = 0;
  float ms[6] = {0, 0, 0, 0, 0, 0};
  // tiles of device-resident inputs: the entry lists of tile s + 1 are sorted on `sort_stream`, in one of two helper
  // records that hold sort buffers only, while tile s accumulates (msm_plan.h: sort_ahead_*)
  mlhip_msm_plan* sort_helper[2] = {nullptr, nullptr};
  hipStream_t sort_stream = nullptr;
  hipEvent_t ev_sorted[2] = {nullptr, nullptr}, ev_lists_free[2] = {nullptr, nullptr};
};


// per-curve entry points, defined in tu_msm_<curve>.hip / tu_pairing_<curve>.hip
#define MLHIP_DECLARE_CURVE(NAME)                                                                                   \
  int mlhip_tu_plan_alloc_##NAME(mlhip_msm_plan* p);                                                                \
  int mlhip_tu_plan_launch_##NAME(mlhip_msm_plan* p, const void* d_points, const void* d_scalars, int mont,        \
                                  size_t n, hipStream_t st);                                                       \
  int mlhip_tu_plan_finish_##NAME(mlhip_msm_plan* p, void* out_affine, void* out_xyzz);                            \
  int mlhip_tu_plan_stream_##NAME(mlhip_msm_plan* p, void* d_points, void* d_scalars, const void* h_points,         \
                                  const void* h_scalars, int mont, size_t n, int segments, hipStream_t st);         \
  int mlhip_tu_plan_shared_##NAME(mlhip_msm_plan* g1, mlhip_msm_plan* g2, void* d_points_g1, void* d_points_g2,     \
                                  void* d_scalars, const void* h_points_g1, const void* h_points_g2,                \
                                  const void* h_scalars, int mont, size_t n, hipStream_t st);                       \
  int mlhip_tu_pairing_##NAME(int what, const void* d_g1, const void* d_g2, size_t ppp, size_t n, const void* d_in, \
                              void* d_out, hipStream_t st);                                                         \
  int mlhip_tu_fp_mul_##NAME(const void* d_a, const void* d_b, size_t n, int repeat, void* d_out, hipStream_t st);   \
  int mlhip_tu_gt_mul_##NAME(const void* d_a, const void* d_b, size_t n, void* d_out, hipStream_t st);             \
  int mlhip_tu_gt_exp_##NAME(const void* d_in, const void* d_scalars, int mont, size_t n, void* d_out,             \
                             hipStream_t st);                                                                      \
  int mlhip_tu_wire_codec_##NAME(int group, int encode, const void* d_in, size_t n, int compressed, int subgroup,   \
                                 void* d_out, void* d_status, hipStream_t st);                                     \
  int mlhip_tu_scalar_mul_##NAME(int group, const void* d_points, size_t point_stride, const void* d_scalars,     \
                                 int mont, size_t n, void* d_out, hipStream_t st);                                 \
  int mlhip_tu_plan_fold_build_##NAME(mlhip_msm_plan* p, const void* d_points, size_t n, hipStream_t st);          \
  void mlhip_tu_release_cache_##NAME(void);
// G1 points outside the prime-order subgroup (or off the curve) in an array of affine points: mlhip_bases_create's check
int mlhip_tu_g1_count_outside_subgroup_Bls377(const void* d_pts, size_t n, uint32_t* d_bad, hipStream_t st);
MLHIP_DECLARE_CURVE(Bn254)
MLHIP_DECLARE_CURVE(Bls381)
MLHIP_DECLARE_CURVE(Bls377)
