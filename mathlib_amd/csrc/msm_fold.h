// msm_fold.h -- shifted-base tables for resident bases (mlhip_bases_*): row (j, i) = 2^off(j) P_i for every digit position
// j of a signed-digit scalar (digit j = bits [off(j), off(j + 1)), at most c wide).  With the table, digit j of scalar i adds +-row(j, i) into bucket |d| of ONE bucket set
// shared by all digit positions -- sum_i s_i P_i = sum_b b (sum over the (i, j) with |d_ij| = b of +-2^off(j) P_i) -- instead
// of +-P_i into bucket |d| of window j followed by sum_j 2^off(j) (window sum).  What that buys on a device with HBM to spare:
//   * the bucket reduction handles 2^(c-1) buckets instead of W 2^(c-1), so c can grow from 16 to 19-20 at the same reduction
//     cost: 13-14 digits instead of 16 = 13-19 % fewer bucket additions (the accumulation is 73 % of an MSM of 2^20 points);
//   * the host tail has lg M doublings (the group weights, see host_tail) instead of one per scalar bit (256).
// The gathers of the accumulation go to a table of W rows per base (1.5 GB for 2^20 BLS12-381 G1 bases, beyond the Infinity
// Cache) instead of re-reading 117 MB W times; measured (profiles/r04_fold_probe.txt) the additions do not wait for them:
// 0.138 ns per addition from the 1.5 GB table against 0.144 ns today.
// The reference has no counterpart (gnark's MultiExp takes fresh slices, driver/gurvy/bls12381/bls12-381.go:766-783); the
// results are the same group elements, byte for byte.  Part of msm_kernels.h.
#pragma once
// (included by msm_kernels.h after its common headers and constants)

namespace mlhip {

// rows[j stride + i] = 2^off(j) P_i, j = 0 .. Wd - 1: one lane per base (fold_rows_body, msm_fold_body.h)
template <class C, class F>
__global__ void __launch_bounds__(64) k_fold_rows(const Affine<F>* __restrict__ pts, size_t n, int c, size_t stride,
                                                  Affine<F>* __restrict__ rows) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fold_rows_body<F>(pts[i], C::FR_BITS, c, stride, rows + i);
}

// first row of the table block that holds base `off` (tile-major layout, mlhip_internal.h)
static inline size_t fold_row(const mlhip_msm_plan* p, size_t off) {
  return (off / p->fold_tile) * (size_t)p->Wd * p->fold_tile + off % p->fold_tile;
}

// Build the table of a folded plan from n affine bases in device memory: tile by tile, the boundary-form rows of a tile in a
// temporary buffer (Wd fold_tile rows: 1.3 GB for 2^20 G1 bases), then their carry-free copy -- Weierstrass rows or, for a
// subgroup-checked table on a curve with the model, Niels triples -- into d_points28, the only form the plan keeps (every
// kernel of a folded plan, the slices of long buckets included, gathers from it).  Runs on `st` and returns when it is done.
template <class C, class F>
int plan_fold_build(mlhip_msm_plan* p, const void* d_points, size_t n, hipStream_t st) {
  typedef Affine<F> A;
  constexpr bool kG2 = std::is_same<F, Fp2Field<C>>::value;
  if (!p->fold) return mlhip_rt::fail(MLHIP_EINVAL, "plan_fold_build on a plan without shifted-base tables");
  if (n == 0 || n > p->max_n) return mlhip_rt::fail(MLHIP_EINVAL, "plan_fold_build: n out of range");
  const size_t TL = p->fold_tile;
  const size_t tiles = (n + TL - 1) / TL;
  const size_t tile_rows = (size_t)p->Wd * TL;
  const size_t rows = tiles * tile_rows;
  const bool use_ed = [&] {
    if constexpr (C::HAS_EDWARDS && !kG2) {
      const char* e = getenv("MLHIP_EDWARDS");
      return p->trust_subgroup && p->reduce28 && (size_t)p->W * p->M > QUAD_ACC_MAX_BUCKETS && !(e && e[0] == '0');
    }
    return false;
  }();
  size_t elem = kG2 ? sizeof(AffineG2_28<C>) : sizeof(Affine28<C>);
  if constexpr (C::HAS_EDWARDS && !kG2) {
    if (use_ed) elem = sizeof(EdNiels28<C>);
  }
  if (p->d_points28 && (p->fold_rows < rows || p->points28_elem < elem)) {
    (void)hipFree(p->d_points28);
    p->d_points28 = nullptr;
  }
  if (!p->d_points28) {
    HIPCHK(hipMalloc(&p->d_points28, rows * elem));
    p->points28_elem = elem;
  }
  p->fold_rows = rows;
  void* tmp = nullptr;
  HIPCHK(hipMalloc(&tmp, tile_rows * sizeof(A)));
  int rc = 0;
  for (size_t k = 0; k < tiles && !rc; k++) {
    const size_t lo = k * TL, len = std::min(TL, n - lo);
    char* dst = (char*)p->d_points28 + k * tile_rows * elem;
    // rows past the last base of a partial tile: the point at infinity
    if (hipMemsetAsync(tmp, 0, tile_rows * sizeof(A), st) != hipSuccess) rc = MLHIP_EHIP;
    k_fold_rows<C, F><<<dim3((unsigned)((len + 63) / 64)), dim3(64), 0, st>>>((const A*)d_points + lo, len, p->c, TL, (A*)tmp);
    if constexpr (kG2) {
      k_points_to28_g2<C><<<dim3((unsigned)((4 * tile_rows + 255) / 256)), dim3(256), 0, st>>>((const A*)tmp, tile_rows,
                                                                                                (AffineG2_28<C>*)dst);
    } else {
      bool converted = false;
      if constexpr (C::HAS_EDWARDS) {
        if (use_ed) {
          k_points_to_ed28<C><<<dim3((unsigned)(((tile_rows + 3) / 4 + 255) / 256)), dim3(256), 0, st>>>((const A*)tmp, tile_rows,
                                                                                                       (EdNiels28<C>*)dst);
          converted = true;
        }
      }
      if (!converted)
        k_points_to28<C><<<dim3((unsigned)((tile_rows + 255) / 256)), dim3(256), 0, st>>>((const A*)tmp, tile_rows,
                                                                                          (Affine28<C>*)dst);
    }
  }
  if (!rc && hipGetLastError() != hipSuccess) rc = MLHIP_EHIP;
  if (hipStreamSynchronize(st) != hipSuccess) rc = MLHIP_EHIP;
  (void)hipFree(tmp);
  if (rc) return mlhip_rt::fail(rc, "building the shifted-base table failed");
  p->fold_n = n;
  p->points_static = true;
  p->conv_src = d_points;
  p->conv_n = n;
  p->conv_ed = use_ed;
  return 0;
}

}  // namespace mlhip
