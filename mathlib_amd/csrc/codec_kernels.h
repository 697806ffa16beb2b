// codec_kernels.h -- bulk G1 / G2 wire decode / encode kernels (one point per lane).  Included by tu_codec_<curve>.hip.
// Replaces NewG1FromBytes / NewG1FromCompressed and G1.Bytes / G1.Compressed of the reference drivers
// (driver/gurvy/bls12381/bls12-381.go:531-569, :286-296) for arrays of points.
#pragma once
#include "codec.h"
#include "mlhip_internal.h"

namespace mlhip {

template <class W>
__global__ void __launch_bounds__(64) k_wire_decode(const uint8_t* __restrict__ wire, size_t n, int compressed, int subgroup,
                                                    typename W::Aff* __restrict__ out, uint8_t* __restrict__ status) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int NB = W::XB;
  const int len = compressed ? NB : 2 * NB;
  // stage the point's bytes in registers/scratch through aligned 32-bit loads (len is a multiple of 4)
  uint8_t w[2 * NB];
  const uint32_t* src = reinterpret_cast<const uint32_t*>(wire + i * (size_t)len);
  for (int k = 0; k < len / 4; k++) {
    uint32_t v = src[k];
    w[4 * k] = (uint8_t)v;
    w[4 * k + 1] = (uint8_t)(v >> 8);
    w[4 * k + 2] = (uint8_t)(v >> 16);
    w[4 * k + 3] = (uint8_t)(v >> 24);
  }
  typename W::Aff p;
  int st = W::decode(p, w, compressed != 0, subgroup);
  out[i] = p;
  status[i] = (uint8_t)st;
}

template <class W>
__global__ void __launch_bounds__(256) k_wire_encode(const typename W::Aff* __restrict__ pts, size_t n, int compressed,
                                                     uint8_t* __restrict__ wire) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int NB = W::XB;
  const int len = compressed ? NB : 2 * NB;
  uint8_t w[2 * NB];
  W::encode(w, pts[i], compressed != 0);
  uint32_t* dst = reinterpret_cast<uint32_t*>(wire + i * (size_t)len);
  for (int k = 0; k < len / 4; k++)
    dst[k] = (uint32_t)w[4 * k] | ((uint32_t)w[4 * k + 1] << 8) | ((uint32_t)w[4 * k + 2] << 16) | ((uint32_t)w[4 * k + 3] << 24);
}

// every point of an array of affine G1 points (boundary form) is the point at infinity or a curve point of the prime-order
// subgroup?  `bad` counts the others.  mlhip_bases_create runs it once per table before it lets a curve with a twisted
// Edwards model (BLS12-377) sum its buckets in those coordinates (ed28.h: the addition law is complete on G1 only).
template <class C>
__global__ void __launch_bounds__(64) k_g1_count_outside_subgroup(const Affine<FpField<C>>* __restrict__ pts, size_t n,
                                                                  uint32_t* __restrict__ bad) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Affine<FpField<C>> P = pts[i];
  if (fp_is_zero<C>(P.x) && fp_is_zero<C>(P.y)) return;
  Fp<C> l, r, b;
  fp_sqr<C>(l, P.y);
  fp_sqr<C>(r, P.x);
  fp_mul<C>(r, r, P.x);
  fp_from_const<C>(b, C::B_G1);
  fp_add<C>(r, r, b);
  if (!fp_eq<C>(l, r) || !g1_in_subgroup<C>(P, 1)) atomicAdd(bad, 1u);
}
template <class C>
int g1_count_outside_subgroup_device(const void* d_pts, size_t n, uint32_t* d_bad, hipStream_t st) {
  if (n == 0) return 0;
  k_g1_count_outside_subgroup<C><<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st>>>((const Affine<FpField<C>>*)d_pts, n, d_bad);
  HIPCHK(hipGetLastError());
  return 0;
}

template <class W>
int wire_codec_device(int encode, const void* d_in, size_t n, int compressed, int subgroup, void* d_out, void* d_status,
                    hipStream_t st) {
  if (n == 0) return 0;
  if (encode)
    k_wire_encode<W><<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>((const typename W::Aff*)d_in, n, compressed,
                                                                          (uint8_t*)d_out);
  else
    k_wire_decode<W><<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st>>>((const uint8_t*)d_in, n, compressed, subgroup,
                                                                        (typename W::Aff*)d_out, (uint8_t*)d_status);
  HIPCHK(hipGetLastError());
  return 0;
}

}  // namespace mlhip
