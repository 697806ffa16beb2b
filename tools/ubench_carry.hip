// Microbenchmark of carry-handling schedules for the 96-bit column accumulate (gfx950); not product code.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s line %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)

#define P12(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11)

template <int V>
__global__ void __launch_bounds__(256) kern(uint64_t* out, int iters, uint32_t seed) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t a[12], b[12];
  for (int i = 0; i < 12; i++) { a[i] = tid * 2654435761u + i * 40503u + seed; b[i] = tid * 2246822519u + i * 97u + 13u; }
  uint64_t acc = tid; uint32_t ext = 0; uint64_t c0, c1, c2;
  for (int it = 0; it < iters; it++) {
    if (V == 0) {        // adjacent, VCC (unsafe per the 2-wait-state rule)
      asm volatile(
#define X(i) "v_mad_u64_u32 %0, vcc, %" #i "+2, %" #i "+14, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, %2, %14, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, %3, %15, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, %4, %16, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, %5, %17, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, %6, %18, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, %7, %19, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, %8, %20, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, %9, %21, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, %10, %22, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, %11, %23, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, %12, %24, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, %13, %25, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
#undef X
        : "+v"(acc), "+v"(ext)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]), "v"(a[10]), "v"(a[11]),
          "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(b[8]), "v"(b[9]), "v"(b[10]), "v"(b[11]) : "vcc");
    } else if (V == 1) { // rotating SGPR pairs, addc (VOP3) two behind
      asm volatile(
        "v_mad_u64_u32 %0, %2, %5, %17, %0\n\t"
        "v_mad_u64_u32 %0, %3, %6, %18, %0\n\t"
        "v_mad_u64_u32 %0, %4, %7, %19, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, %2\n\t"
        "v_mad_u64_u32 %0, %2, %8, %20, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, %3\n\t"
        "v_mad_u64_u32 %0, %3, %9, %21, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, %4\n\t"
        "v_mad_u64_u32 %0, %4, %10, %22, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, %2\n\t"
        "v_mad_u64_u32 %0, %2, %11, %23, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, %3\n\t"
        "v_mad_u64_u32 %0, %3, %12, %24, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, %4\n\t"
        "v_mad_u64_u32 %0, %4, %13, %25, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, %2\n\t"
        "v_mad_u64_u32 %0, %2, %14, %26, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, %3\n\t"
        "v_mad_u64_u32 %0, %3, %15, %27, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, %4\n\t"
        "v_mad_u64_u32 %0, %4, %16, %28, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, %2\n\t"
        "v_addc_co_u32 %1, vcc, 0, %1, %3\n\tv_addc_co_u32 %1, vcc, 0, %1, %4"
        : "+v"(acc), "+v"(ext), "=&s"(c0), "=&s"(c1), "=&s"(c2)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]), "v"(a[10]), "v"(a[11]),
          "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(b[8]), "v"(b[9]), "v"(b[10]), "v"(b[11]) : "vcc");
    } else if (V == 2) { // VCC + explicit s_nop 1 (what hipcc would emit)
      asm volatile(
        "v_mad_u64_u32 %0, vcc, %2, %14, %0\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, %3, %15, %0\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, %4, %16, %0\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, %5, %17, %0\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, %6, %18, %0\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, %7, %19, %0\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, %8, %20, %0\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, %9, %21, %0\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, %10, %22, %0\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, %11, %23, %0\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, %12, %24, %0\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, %13, %25, %0\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(acc), "+v"(ext)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]), "v"(a[10]), "v"(a[11]),
          "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(b[8]), "v"(b[9]), "v"(b[10]), "v"(b[11]) : "vcc");
    } else if (V == 3) { // two interleaved columns (two accumulators), VCC for one and an SGPR pair for the other
      uint64_t acc2 = acc ^ 0x55; uint32_t ext2 = ext;
      asm volatile(
        "v_mad_u64_u32 %0, vcc, %5, %17, %0\n\t"
        "v_mad_u64_u32 %2, %4, %6, %18, %2\n\t"
#define STEP(x, y, x2, y2) "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %" #x ", %" #y ", %0\n\tv_addc_co_u32 %3, %4, 0, %3, %4\n\tv_mad_u64_u32 %2, %4, %" #x2 ", %" #y2 ", %2\n\t"
        STEP(7, 19, 8, 20) STEP(9, 21, 10, 22) STEP(11, 23, 12, 24) STEP(13, 25, 14, 26) STEP(15, 27, 16, 28)
#undef STEP
        "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\ts_nop 0\n\tv_addc_co_u32 %3, %4, 0, %3, %4"
        : "+v"(acc), "+v"(ext), "+v"(acc2), "+v"(ext2), "=&s"(c0)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]), "v"(a[10]), "v"(a[11]),
          "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(b[8]), "v"(b[9]), "v"(b[10]), "v"(b[11]) : "vcc");
      acc += acc2; ext += ext2;
    }
    a[it & 7] += ext;
  }
  out[tid] = acc + ext;
}

template <int V> int run(const char* name, uint64_t* d, int blocks, int iters) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  kern<V><<<blocks, 256>>>(d, 16, 1); CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0)); kern<V><<<blocks, 256>>>(d, iters, 2); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  uint64_t h; CHECK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
  double macs = (double)blocks * 256 * iters * 12;
  printf("%-52s blocks=%5d  %8.3f ms  %.3e MAC-with-carry/s  (%llx)\n", name, blocks, ms, macs / (ms * 1e-3), (unsigned long long)h); fflush(stdout);
  return 0;
}
int main() {
  uint64_t* d; CHECK(hipMalloc(&d, 256 * 16 * 256 * 8));
  for (int blocks : {256, 512, 1024, 2048}) {
    run<0>("V0 adjacent mad/addc via VCC (no wait states)", d, blocks, 4096);
    run<1>("V1 rotating SGPR pairs, addc two behind (safe)", d, blocks, 4096);
    run<2>("V2 VCC + s_nop 1 (hipcc's padding)", d, blocks, 4096);
    run<3>("V3 two interleaved accumulators (VCC + SGPR)", d, blocks, 4096);
  }
  return 0;
}
