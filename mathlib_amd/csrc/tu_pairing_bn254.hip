// Pairing kernels instantiated for Bn254.
#include "pairing_kernels.h"
using namespace mlhip;
int mlhip_tu_pairing_Bn254(int what, const void* d_g1, const void* d_g2, size_t ppp, size_t n, const void* d_in,
                        void* d_out, hipStream_t st) {
  return pairing_device<Bn254>(what, d_g1, d_g2, ppp, n, d_in, d_out, st);
}
int mlhip_tu_fp_mul_Bn254(const void* d_a, const void* d_b, size_t n, int repeat, void* d_out, hipStream_t st) {
  return fp_mul_device<Bn254>(d_a, d_b, n, repeat, d_out, st);
}
int mlhip_tu_gt_mul_Bn254(const void* d_a, const void* d_b, size_t n, void* d_out, hipStream_t st) {
  return gt_mul_device<Bn254>(d_a, d_b, n, d_out, st);
}
int mlhip_tu_gt_exp_Bn254(const void* d_in, const void* d_scalars, int mont, size_t n, void* d_out, hipStream_t st) {
  return gt_exp_device<Bn254>(d_in, d_scalars, mont, n, d_out, st);
}
