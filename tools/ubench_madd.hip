// A/B microbenchmark of mixed-addition (XYZZ += affine) code shapes on gfx950; not product code.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench_madd.hip -o tools/ubench_madd
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "../mathlib_amd/csrc/msm_body.h"
#include "../mathlib_amd/csrc/ec28.h"
using namespace mlhip;
#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s line %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
typedef Bls381 C;
typedef FpField<C> F0;
struct FI {  // same layout as FpField<C>, multiply inlined
  using Curve = C; using T = Fp<C>; static constexpr int WORDS = 12;
  __host__ __device__ __forceinline__ static void zero(T& r) { fp_zero<C>(r); }
  __host__ __device__ __forceinline__ static void one(T& r) { fp_one<C>(r); }
  __host__ __device__ __forceinline__ static bool is_zero(const T& a) { return fp_is_zero<C>(a); }
  __host__ __device__ __forceinline__ static bool eq(const T& a, const T& b) { return fp_eq<C>(a, b); }
  __host__ __device__ __forceinline__ static void add(T& r, const T& a, const T& b) { fp_add<C>(r, a, b); }
  __host__ __device__ __forceinline__ static void sub(T& r, const T& a, const T& b) { fp_sub<C>(r, a, b); }
  __host__ __device__ __forceinline__ static void dbl(T& r, const T& a) { fp_dbl<C>(r, a); }
  __host__ __device__ __forceinline__ static void neg(T& r, const T& a) { fp_neg<C>(r, a); }
  __host__ __device__ __forceinline__ static void mul(T& r, const T& a, const T& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    fp_mul_device<C>(r, a, b);
#else
    fp_mul_inline<C>(r, a, b);
#endif
  }
  __host__ __device__ __forceinline__ static void sqr(T& r, const T& a) { mul(r, a, a); }
  __host__ __device__ __forceinline__ static void inv(T& r, const T& a) { fp_inv<C>(r, a); }
  __host__ __device__ __forceinline__ static void select(T& r, bool c, const T& a, const T& b) { fp_select<C>(r, c, a, b); }
};
__device__ __noinline__ void madd_ool(XYZZ<FI>& acc, const Affine<FI>& q, bool neg) { xyzz_madd<FI>(acc, q, neg); }

template <int V, int WPS>
__global__ void __launch_bounds__(256, WPS) kern(const Affine<F0>* pts, int npts, int iters, XYZZ<F0>* out) {
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (V == 0) {
    XYZZ<F0> acc; xyzz_set_inf<F0>(acc);
    for (int k = 0; k < iters; k++) { Affine<F0> p = pts[(g * 7 + k * 13) % npts]; xyzz_madd<F0>(acc, p, (k & 1) != 0); }
    out[g] = acc;
  } else if (V == 1) {
    XYZZ<FI> acc; xyzz_set_inf<FI>(acc);
    const Affine<FI>* q = (const Affine<FI>*)pts;
    for (int k = 0; k < iters; k++) { Affine<FI> p = q[(g * 7 + k * 13) % npts]; madd_ool(acc, p, (k & 1) != 0); }
    ((XYZZ<FI>*)out)[g] = acc;
  } else {
    XYZZ<FI> acc; xyzz_set_inf<FI>(acc);
    const Affine<FI>* q = (const Affine<FI>*)pts;
    for (int k = 0; k < iters; k++) { Affine<FI> p = q[(g * 7 + k * 13) % npts]; xyzz_madd<FI>(acc, p, (k & 1) != 0); }
    ((XYZZ<FI>*)out)[g] = acc;
  }
}

// V3: the carry-free 28-bit-limb form (fp28.h / ec28.h); out = the bucket converted back to the boundary form
template <int WPS>
__global__ void __launch_bounds__(256, WPS) kern28(const Affine28<C>* pts, int npts, int iters, XYZZ<F0>* out) {
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  XYZZ28<C> acc; bool inf = true;
  for (int k = 0; k < iters; k++) { Affine28<C> p = pts[(g * 7 + k * 13) % npts]; xyzz28_madd<C>(acc, inf, p, (k & 1) != 0); }
  XYZZ<F0> r; xyzz28_to<C>(r, acc, inf);
  out[g] = r;
}
template <int WPS> int run28(const char* name, const Affine28<C>* d_pts, int npts, XYZZ<F0>* d_out, int blocks, int iters, const Affine<F0>* want) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  kern28<WPS><<<blocks, 256>>>(d_pts, npts, 2, d_out); CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    CHECK(hipEventRecord(e0)); kern28<WPS><<<blocks, 256>>>(d_pts, npts, iters, d_out); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  XYZZ<F0> h[3]; CHECK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
  bool ok = true;
  for (int i = 0; i < 3; i++) { Affine<F0> a; xyzz_to_affine<F0>(a, h[i]); ok &= memcmp(&a, &want[i], sizeof(a)) == 0; }
  printf("%-44s blocks=%5d iters=%-4d %8.3f ms  %.3e madd/s  (same points as V2: %s)\n", name, blocks, iters, best, (double)blocks * 256 * iters / (best * 1e-3), ok ? "OK" : "MISMATCH");
  return 0;
}

template <int V, int WPS> int run(const char* name, const Affine<F0>* d_pts, int npts, XYZZ<F0>* d_out, int blocks, int iters) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  kern<V, WPS><<<blocks, 256>>>(d_pts, npts, 2, d_out); CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    CHECK(hipEventRecord(e0)); kern<V, WPS><<<blocks, 256>>>(d_pts, npts, iters, d_out); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  std::vector<uint32_t> h(48); CHECK(hipMemcpy(h.data(), d_out, 192, hipMemcpyDeviceToHost));
  double madds = (double)blocks * 256 * iters;
  printf("%-44s blocks=%5d iters=%d  %8.3f ms  %.3e madd/s  (out[0]=%08x)\n", name, blocks, iters, best, madds / (best * 1e-3), h[0]); fflush(stdout);
  return 0;
}

int main() {
  // points: [k]G for small k computed on the host with the same headers (valid curve points)
  const int npts = 4096;
  std::vector<Affine<F0>> pts(npts);
  Affine<F0> g; fp_from_const<C>(g.x, C::G1X); fp_from_const<C>(g.y, C::G1Y);
  XYZZ<F0> acc; xyzz_set_inf<F0>(acc);
  for (int i = 0; i < npts; i++) { xyzz_madd<F0>(acc, g, false); xyzz_to_affine<F0>(pts[i], acc); }
  Affine<F0>* d_pts; XYZZ<F0>* d_out;
  const int maxblocks = 256 * 16;
  CHECK(hipMalloc(&d_pts, npts * sizeof(Affine<F0>))); CHECK(hipMalloc(&d_out, (size_t)maxblocks * 256 * sizeof(XYZZ<F0>)));
  CHECK(hipMemcpy(d_pts, pts.data(), npts * sizeof(Affine<F0>), hipMemcpyHostToDevice));
  std::vector<Affine28<C>> pts28(npts);
  for (int i = 0; i < npts; i++) affine28_from<C>(pts28[i], pts[i]);
  Affine28<C>* d_pts28; CHECK(hipMalloc(&d_pts28, npts * sizeof(Affine28<C>)));
  CHECK(hipMemcpy(d_pts28, pts28.data(), npts * sizeof(Affine28<C>), hipMemcpyHostToDevice));
  for (int blocks : {256 * 4, 256 * 8}) {
    run<0, 1>("V0 fp_mul out-of-line (pointers)", d_pts, npts, d_out, blocks, 32);
    run<1, 1>("V1 madd out-of-line, mul inlined, no cap", d_pts, npts, d_out, blocks, 32);
    run<1, 3>("V1 madd out-of-line, mul inlined, 3 w/SIMD", d_pts, npts, d_out, blocks, 32);
    run<1, 4>("V1 madd out-of-line, mul inlined, 4 w/SIMD", d_pts, npts, d_out, blocks, 32);
    run<2, 1>("V2 fully inlined loop, no cap", d_pts, npts, d_out, blocks, 32);
    run<2, 3>("V2 fully inlined loop, 3 w/SIMD", d_pts, npts, d_out, blocks, 32);
    run<2, 4>("V2 fully inlined loop, 4 w/SIMD", d_pts, npts, d_out, blocks, 32);
    // reference result of the first three lanes from the host with the boundary-form code
    Affine<F0> want[3];
    for (int gid = 0; gid < 3; gid++) {
      XYZZ<F0> a; xyzz_set_inf<F0>(a);
      for (int k = 0; k < 32; k++) xyzz_madd<F0>(a, pts[(gid * 7 + k * 13) % npts], (k & 1) != 0);
      xyzz_to_affine<F0>(want[gid], a);
    }
    run28<1>("V3 carry-free 28-bit limbs, no cap", d_pts28, npts, d_out, blocks, 32, want);
    run28<2>("V3 carry-free 28-bit limbs, 2 w/SIMD", d_pts28, npts, d_out, blocks, 32, want);
    run28<3>("V3 carry-free 28-bit limbs, 3 w/SIMD", d_pts28, npts, d_out, blocks, 32, want);
  }
  return 0;
}
