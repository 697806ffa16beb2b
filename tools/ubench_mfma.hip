// Micro-benchmark behind DESIGN.md section 7 ("the Montgomery reduction on MFMA"): what the two
// primitive rates are on gfx950 --
//   A  v_mfma_i32_32x32x32_i8 back to back (the i8 matrix instruction a constant-Toeplitz reduction would use):
//      cycles per instruction per SIMD, one and two waves per SIMD;
//   B  v_mad_i64_i32 alone (eight chains per wave, two waves per SIMD): the rate the kernels run at today;
//   C  both on one SIMD: waves 0-3 of a 512-thread workgroup issue MFMAs, waves 4-7 the v_mad chains -- does the matrix
//      pipe run beside the multiplier, and what does the v_mad stream lose;
//   D  one wave interleaving 1 MFMA with K v_mad (the form a fused kernel would have).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma.hip -o tools/ubench_mfma ; run: ./tools/ubench_mfma
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// mode 0: MFMA only; 1: mad only; 2: waves 0-3 MFMA, waves 4-7 mad (blockDim 512); 3: interleaved, K mads per MFMA
template <int MODE, int K>
__global__ void __launch_bounds__(512) bench(int64_t* out, int iters, int seed) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int wave = threadIdx.x >> 6;
  v4i a = {tid + seed, tid * 3 + 1, tid * 5 + 2, tid * 7 + 3}, b = {tid * 11 + 4, tid * 13 + 5, tid ^ seed, tid + 9};
  v16i c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  int64_t acc[8];
  int32_t x[8], y[8];
  for (int i = 0; i < 8; i++) {
    acc[i] = tid * 977 + i + seed;
    x[i] = tid * 31 + i * 7 + seed;
    y[i] = tid * 13 + i + 3;
  }
  const bool do_mfma = MODE == 0 || MODE == 3 || (MODE == 2 && wave < 4);
  const bool do_mad = MODE == 1 || MODE == 3 || (MODE == 2 && wave >= 4);
  for (int it = 0; it < iters; it++) {
    if (MODE == 3) {
      c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; k++) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[k & 7]) : "v"(x[k & 7]), "v"(y[k & 7]) : "vcc");
      c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; k++) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[k & 7]) : "v"(x[k & 7]), "v"(y[k & 7]) : "vcc");
    } else {
      if (do_mfma) {  // four independent accumulators: the 32 x 32 x 32 result takes 16 passes to come back
        c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
      }
      if (do_mad) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
          for (int k = 0; k < 8; k++) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[k]) : "v"(x[k]), "v"(y[k]) : "vcc");
        }
      }
    }
  }
  int64_t s = 0;
  for (int i = 0; i < 8; i++) s += acc[i];
  for (int i = 0; i < 16; i++) s += c0[i] + c1[i] + c2[i] + c3[i];
  out[tid] = s;
}

template <int MODE, int K>
int run(const char* name, int blocks, int threads, int iters, double mfma_per_iter, double mad_per_iter, int cus, double ghz) {
  int64_t* out;
  CHECK(hipMalloc(&out, (size_t)blocks * threads * 8));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  bench<MODE, K><<<blocks, threads>>>(out, 64, 1);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  bench<MODE, K><<<blocks, threads>>>(out, iters, 2);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double waves = (double)blocks * threads / 64;
  const double simds = cus * 4.0;
  // instructions per SIMD, and the SIMD cycles (at `ghz`) spent per instruction of each kind
  const double cyc = ms * 1e-3 * ghz * 1e9;
  printf("%-58s %8.3f ms", name, ms);
  if (mfma_per_iter > 0) printf("  | %.1f cycles per MFMA per SIMD", cyc / (mfma_per_iter * iters * waves / simds));
  if (mad_per_iter > 0) printf("  | %.2f cycles per v_mad per SIMD", cyc / (mad_per_iter * iters * waves / simds));
  printf("\n");
  CHECK(hipFree(out));
  return 0;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double ghz = 2.15;  // what rocm-smi shows under these integer loads (profiles/r03_power_clocks.txt)
  printf("device %s, %d CUs; cycles quoted at %.2f GHz\n", prop.name, cus, ghz);
  const int iters = 20000;
  // per iteration and wave: MFMA-only 4 MFMAs; mad-only 32 mads
  if (run<0, 0>("A  MFMA only, 1 wave per SIMD (256-thread blocks)", cus, 256, iters, 4, 0, cus, ghz)) return 1;
  if (run<0, 0>("A  MFMA only, 2 waves per SIMD (512-thread blocks)", cus, 512, iters, 4, 0, cus, ghz)) return 1;
  if (run<1, 0>("B  v_mad_i64_i32 only, 1 wave per SIMD", cus, 256, iters, 0, 32, cus, ghz)) return 1;
  if (run<1, 0>("B  v_mad_i64_i32 only, 2 waves per SIMD", cus, 512, iters, 0, 32, cus, ghz)) return 1;
  // C: per SIMD one MFMA wave (4 per iteration) and one mad wave (32 per iteration): waves = 8 per block, half each
  if (run<2, 0>("C  one MFMA wave + one v_mad wave per SIMD", cus, 512, iters, 4 * 0.5, 32 * 0.5, cus, ghz)) return 1;
  if (run<3, 8>("D  one stream, 8 v_mad per MFMA, 2 waves per SIMD", cus, 512, iters, 2, 16, cus, ghz)) return 1;
  if (run<3, 16>("D  one stream, 16 v_mad per MFMA, 2 waves per SIMD", cus, 512, iters, 2, 32, cus, ghz)) return 1;
  if (run<3, 32>("D  one stream, 32 v_mad per MFMA, 2 waves per SIMD", cus, 512, iters, 2, 64, cus, ghz)) return 1;
  return 0;
}
