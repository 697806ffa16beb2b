// Instruction-rate microbenchmark for gfx950: integer multiply / add / f64 fma issue cost.
// Used once to choose the Fp limb representation (see DESIGN.md "integer roofline").
// Build: hipcc -O3 --offload-arch=gfx950 ubench_int.hip -o ubench_int ; run: ./ubench_int
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <string>

#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template<int KIND>
__global__ void __launch_bounds__(256) bench(uint64_t* out, uint64_t* cyc, int iters, uint32_t seed) {
  uint32_t tid = blockIdx.x*blockDim.x+threadIdx.x;
  uint64_t acc[8]; uint32_t a[8], b[8]; double d[8];
  for (int i=0;i<8;i++){ acc[i]=tid*977u+i+seed; a[i]=tid*31u+i*7u+seed; b[i]=tid*13u+i+3u; d[i]=1.0+1e-9*(tid+i); }
  double fa = 1.0000001, fb=1e-12;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it=0; it<iters; ++it) {
    #pragma unroll
    for (int u=0;u<4;u++) {
      if (KIND==0) {
        #define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
        REP8(X)
        #undef X
      } else if (KIND==1) {
        #define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
        #undef X
      } else if (KIND==2) {
        #define X(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
        #undef X
      } else if (KIND==3) {
        #define X(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
        #undef X
      } else if (KIND==4) {
        #define X(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
        #undef X
      } else if (KIND==5) {
        #define X(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(fa), "v"(fb));
        REP8(X)
        #undef X
      } else if (KIND==6) {
        #define X(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(acc[(i+1)&7]));
        REP8(X)
        #undef X
      } else if (KIND==7) {
        #define X(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %2, vcc, %2, %3, vcc" : "+v"(a[i]), "+v"(b[i]) : "v"(a[(i+1)&7]), "v"(b[(i+3)&7]) : "vcc");
        REP8(X)
        #undef X
      } else if (KIND==8) {
        #define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
        #undef X
      } else if (KIND==9) {
        #define X(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
        #undef X
      } else if (KIND==10) {
        #define X(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i+1)&7]));
        REP8(X)
        #undef X
      } else if (KIND==11) {
        // mad + carry count (96-bit column accumulate)
        #define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc" : "+v"(acc[i]), "+v"(b[i]) : "v"(a[i]), "v"(a[(i+1)&7]), "v"(b[i]) : "vcc");
        REP8(X)
        #undef X
      } else if (KIND==12) {
        #define X(i) asm volatile("v_mad_u64_u32 %0, %3, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i]), "v"(b[i]), "s"((uint64_t)0) : );
        // not valid (sdst must be output) - replaced below
        #undef X
        #define X(i) { uint64_t sc; asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc[i]), "=s"(sc) : "v"(a[i]), "v"(b[i])); }
        REP8(X)
        #undef X
      } else if (KIND==13) {
        #define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(fa));
        REP8(X)
        #undef X
      } else if (KIND==14) {
        #define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(fb));
        REP8(X)
        #undef X
      } else if (KIND==15) {
        #define X(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
        #undef X
      } else if (KIND==16) {
        #define X(i) asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
        #undef X
      } else if (KIND==17) {
        #define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(acc[i]) : "v"(acc[(i+1)&7]));
        REP8(X)
        #undef X
      } else if (KIND==18) {
        #define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
        #undef X
      } else if (KIND==19) {
        #define X(i) asm volatile("v_alignbit_b32 %0, %0, %1, 28" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
        #undef X
      } else if (KIND==20) {
        #define X(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
        #undef X
      } else if (KIND==21) {
        #define X(i) asm volatile("v_lshrrev_b64 %0, 28, %0" : "+v"(acc[i]));
        REP8(X)
        #undef X
      }
    }
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  uint64_t s=0; for(int i=0;i<8;i++){ s+=acc[i]+a[i]+b[i]+(uint64_t)d[i]; }
  out[tid]=s;
  if ((threadIdx.x&63)==0) cyc[tid>>6]=t1-t0;
}

struct K { const char* name; int insts_per_slot; void (*fn)(uint64_t*,uint64_t*,int,uint32_t); };

int main(){
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop,0));
  printf("device %s CUs=%d clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  K ks[] = {
    {"v_mad_u64_u32(vcc)",1,bench<0>},{"v_mul_lo_u32",1,bench<1>},{"v_mul_hi_u32",1,bench<2>},
    {"v_mad_u32_u24",1,bench<3>},{"v_mul_hi_u32_u24",1,bench<4>},{"v_fma_f64",1,bench<5>},
    {"v_lshl_add_u64",1,bench<6>},{"v_add_co+v_addc_co (pair)",2,bench<7>},{"v_add_u32",1,bench<8>},
    {"v_mov_b32",1,bench<9>},{"v_add3_u32",1,bench<10>},{"mad_u64+addc (pair)",2,bench<11>},
    {"v_mad_u64_u32(sgpr carry)",1,bench<12>},{"v_mul_f64",1,bench<13>},{"v_add_f64",1,bench<14>},
    {"v_mul_u32_u24",1,bench<15>},{"v_mad_i32_i24",1,bench<16>},{"v_pk_fma_f32",1,bench<17>},{"v_fma_f32",1,bench<18>},
    {"v_alignbit_b32",1,bench<19>},{"v_and_b32",1,bench<20>},{"v_lshrrev_b64",1,bench<21>},
  };
  int nCU = prop.multiProcessorCount;
  const int iters = 4096;
  for (int wps : {1,2,4,8}) {
    int blocks = nCU * wps;           // 256 threads = 4 waves = 1 wave/SIMD per block
    size_t nthreads = (size_t)blocks*256;
    uint64_t *out,*cyc; CHECK(hipMalloc(&out,nthreads*8)); CHECK(hipMalloc(&cyc,nthreads/64*8));
    std::vector<uint64_t> h(nthreads/64);
    for (auto& k : ks) {
      hipEvent_t e0,e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
      k.fn<<<blocks,256>>>(out,cyc,64,1); CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      k.fn<<<blocks,256>>>(out,cyc,iters,2);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms,e0,e1));
      CHECK(hipMemcpy(h.data(),cyc,h.size()*8,hipMemcpyDeviceToHost));
      double avg=0; for(auto v:h) avg+=v; avg/=h.size();
      double slots = (double)iters*4*8;   // asm statements per wave
      double cyc_per_stmt = avg/slots;      // s_memtime ticks (100MHz? or shader clock) per statement per wave
      double stmts_per_s = slots*(double)(nthreads/64)/(ms*1e-3);
      printf("wps=%d %-28s %8.3f ms  memtime_ticks/stmt/wave=%7.3f  wave-stmts/s=%.3e  lane-ops/s=%.3e (x%d inst)\n",
             wps,k.name,ms,cyc_per_stmt,stmts_per_s,stmts_per_s*64,k.insts_per_slot);
      hipEventDestroy(e0); hipEventDestroy(e1);
    }
    hipFree(out); hipFree(cyc);
  }
  return 0;
}
