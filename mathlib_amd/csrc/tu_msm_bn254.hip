// MSM kernels instantiated for Bn254 (G1 over Fp, G2 over Fp2).
#include "msm_kernels.h"
using namespace mlhip;
int mlhip_tu_plan_alloc_Bn254(mlhip_msm_plan* p) {
  return p->group == MLHIP_GROUP_G1 ? plan_alloc<FpField<Bn254>>(p) : plan_alloc<Fp2Field<Bn254>>(p);
}
int mlhip_tu_plan_launch_Bn254(mlhip_msm_plan* p, const void* d_points, const void* d_scalars, int mont, size_t n,
                            hipStream_t st) {
  if (p->group == MLHIP_GROUP_G1) return plan_launch<Bn254, FpField<Bn254>>(p, d_points, d_scalars, mont, n, st);
  return plan_launch<Bn254, Fp2Field<Bn254>>(p, d_points, d_scalars, mont, n, st);
}
int mlhip_tu_plan_finish_Bn254(mlhip_msm_plan* p, void* out_affine, void* out_xyzz) {
  if (p->group == MLHIP_GROUP_G1) return plan_finish<Bn254, FpField<Bn254>>(p, out_affine, out_xyzz);
  return plan_finish<Bn254, Fp2Field<Bn254>>(p, out_affine, out_xyzz);
}
int mlhip_tu_plan_stream_Bn254(mlhip_msm_plan* p, void* d_points, void* d_scalars, const void* h_points,
                            const void* h_scalars, int mont, size_t n, int segments, hipStream_t st) {
  if (p->group == MLHIP_GROUP_G1) return plan_stream<Bn254, FpField<Bn254>>(p, d_points, d_scalars, h_points, h_scalars, mont, n, segments, st);
  return plan_stream<Bn254, Fp2Field<Bn254>>(p, d_points, d_scalars, h_points, h_scalars, mont, n, segments, st);
}
int mlhip_tu_plan_shared_Bn254(mlhip_msm_plan* g1, mlhip_msm_plan* g2, void* d_points_g1, void* d_points_g2, void* d_scalars,
                            const void* h_points_g1, const void* h_points_g2, const void* h_scalars, int mont, size_t n,
                            hipStream_t st) {
  return plan_stream_shared<Bn254>(g1, g2, d_points_g1, d_points_g2, d_scalars, h_points_g1, h_points_g2, h_scalars, mont, n, st);
}
int mlhip_tu_scalar_mul_Bn254(int group, const void* d_points, size_t point_stride, const void* d_scalars, int mont,
                              size_t n, void* d_out, hipStream_t st) {
  if (group == MLHIP_GROUP_G1)
    return scalar_mul_device<Bn254, FpField<Bn254>>(d_points, point_stride, d_scalars, mont, n, d_out, st);
  return scalar_mul_device<Bn254, Fp2Field<Bn254>>(d_points, point_stride, d_scalars, mont, n, d_out, st);
}
int mlhip_tu_plan_fold_build_Bn254(mlhip_msm_plan* p, const void* d_points, size_t n, hipStream_t st) {
  if (p->group == MLHIP_GROUP_G1) return plan_fold_build<Bn254, FpField<Bn254>>(p, d_points, n, st);
  return plan_fold_build<Bn254, Fp2Field<Bn254>>(p, d_points, n, st);
}
void mlhip_tu_release_cache_Bn254(void) { fixed_base_release(); }
