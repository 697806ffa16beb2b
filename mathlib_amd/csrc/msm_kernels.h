// msm_kernels.h -- Pippenger MSM kernels and their launch sequence (gfx950).  Included by the per-curve
// translation units tu_msm_<curve>.hip.  Pipeline and data layout: DESIGN.md section 3.
//   k_coarse_hist / k_coarse_scatter / k_fine_sort   two-level LDS counting sort of the (window, bucket) keys
//                 (k_digits / k_scan / k_scatter: the global-atomic variant for n > 2^24, MLHIP_LEGACY_SORT=1)
//   k_order_*     buckets ordered by population, so that a wave's lanes run equally long loops
//   k_points_to28 / k_accumulate28      G1: points into the carry-free form (fp28.h), one thread per bucket:
//                 XYZZ += points[idx] (mixed additions, gathered reads); k_accumulate is the boundary-form variant
//   k_points_to28_g2 / k_accumulate28_lp / k_accumulate_lp   G2: two lanes per bucket (one Fp2 component each)
//   k_points_to_ed28 / k_accumulate_ed28_seg   G1 of a subgroup-trusted plan on a curve with a twisted Edwards model
//                 (BLS12-377): 7-product unified mixed additions (ed28.h, msm_ed.h), buckets handed on as XYZZ28
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
#include "ec_jac.h"
#include "ed28.h"
#include "ec28_lp.h"
#include "ec28_kc.h"
#include "ec_quad.h"
#include "ec_quad28.h"
#include "fp2_lanes.h"
#include "mlhip_internal.h"
#include "msm_body.h"
#include "msm_fold_body.h"

namespace mlhip {

constexpr size_t QUAD_ACC_MAX_BUCKETS = 32768;  // up to here a G1 MSM accumulates one bucket per quad of lanes (k_accumulate_q28)
constexpr uint32_t BIG_BUCKET_MIN = 256;  // a bucket goes to the sliced long-bucket path above max(this, 8 x the mean length)
constexpr int CHUNK_L = 8;            // buckets per level-1 reduction thread

}  // namespace mlhip

#include "msm_sort.h"
#include "msm_accumulate.h"
#include "msm_ed.h"
#include "msm_reduce.h"
#include "msm_g2.h"
#include "msm_scalar_mul.h"
#include "msm_fold.h"
#include "msm_plan.h"
