// Wire-format codec kernels instantiated for Bn254.
#include "codec_kernels.h"
using namespace mlhip;
int mlhip_tu_g1_codec_Bn254(int encode, const void* d_in, size_t n, int compressed, int subgroup, void* d_out,
                           void* d_status, hipStream_t st) {
  return g1_codec_device<Bn254>(encode, d_in, n, compressed, subgroup, d_out, d_status, st);
}
