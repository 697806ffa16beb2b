// Wire-format codec kernels instantiated for Bls377.
#include "codec_kernels.h"
using namespace mlhip;
int mlhip_tu_wire_codec_Bls377(int group, int encode, const void* d_in, size_t n, int compressed, int subgroup, void* d_out,
                             void* d_status, hipStream_t st) {
  if (group == 2) return wire_codec_device<G2Wire<Bls377>>(encode, d_in, n, compressed, subgroup, d_out, d_status, st);
  return wire_codec_device<G1Wire<Bls377>>(encode, d_in, n, compressed, subgroup, d_out, d_status, st);
}
int mlhip_tu_g1_count_outside_subgroup_Bls377(const void* d_pts, size_t n, uint32_t* d_bad, hipStream_t st) {
  return g1_count_outside_subgroup_device<Bls377>(d_pts, n, d_bad, st);
}
