// Wire-format codec kernels instantiated for Bls377.
#include "codec_kernels.h"
using namespace mlhip;
int mlhip_tu_g1_codec_Bls377(int encode, const void* d_in, size_t n, int compressed, int subgroup, void* d_out,
                           void* d_status, hipStream_t st) {
  return g1_codec_device<Bls377>(encode, d_in, n, compressed, subgroup, d_out, d_status, st);
}
