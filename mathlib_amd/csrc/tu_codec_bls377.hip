// Wire-format codec kernels instantiated for Bls377.
#include "codec_kernels.h"
using namespace mlhip;
int mlhip_tu_wire_codec_Bls377(int group, int encode, const void* d_in, size_t n, int compressed, int subgroup, void* d_out,
                             void* d_status, hipStream_t st) {
  if (group == 2) return wire_codec_device<G2Wire<Bls377>>(encode, d_in, n, compressed, subgroup, d_out, d_status, st);
  return wire_codec_device<G1Wire<Bls377>>(encode, d_in, n, compressed, subgroup, d_out, d_status, st);
}
