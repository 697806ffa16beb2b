// Pairing kernels instantiated for Bls377.
#include "pairing_kernels.h"
using namespace mlhip;
int mlhip_tu_pairing_Bls377(int what, const void* d_g1, const void* d_g2, size_t ppp, size_t n, const void* d_in,
                        void* d_out, hipStream_t st) {
  return pairing_device<Bls377>(what, d_g1, d_g2, ppp, n, d_in, d_out, st);
}
int mlhip_tu_fp_mul_Bls377(const void* d_a, const void* d_b, size_t n, int repeat, void* d_out, hipStream_t st) {
  return fp_mul_device<Bls377>(d_a, d_b, n, repeat, d_out, st);
}
int mlhip_tu_gt_mul_Bls377(const void* d_a, const void* d_b, size_t n, void* d_out, hipStream_t st) {
  return gt_mul_device<Bls377>(d_a, d_b, n, d_out, st);
}
int mlhip_tu_gt_exp_Bls377(const void* d_in, const void* d_scalars, int mont, size_t n, void* d_out, hipStream_t st) {
  return gt_exp_device<Bls377>(d_in, d_scalars, mont, n, d_out, st);
}
