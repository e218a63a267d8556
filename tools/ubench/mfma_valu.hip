// Microbenchmark (dev tool): does VALU work hide under v_mfma_f32_16x16x4_f32?  One wavefront per SIMD (256 threads, 1 block per CU
// on a few CUs); loop of 12 MFMAs (3 independent accumulators, like a k-tile step of the N = 200 tile) with K filler
// instructions behind EVERY MFMA; cycles per MFMA from s_memtime.  Build: hipcc --offload-arch=gfx950 -O3 mfma_valu.hip -o mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;

#define REP4(x) x x x x
#define REP12(x) REP4(x) REP4(x) REP4(x)

template <int KIND, int K>
__global__ void __launch_bounds__(256, 1) k(float *out, unsigned long long *cyc, int iters) {
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0;
  float wa = threadIdx.x * 1e-3f, wb = 1.0f + threadIdx.x * 1e-4f;
  float v0 = wa, v1 = wb, v2 = wa + 1, v3 = wb + 1, v4 = 0.5f, v5 = 0.25f;
  double d0 = wa, d1 = wb, d2 = 1.5, d3 = 2.5;
  __shared__ float lds[4096];
  lds[threadIdx.x] = wa;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 12; ++m) {
      if (m % 3 == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(wa), "v"(wb));
      if (m % 3 == 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(wa), "v"(wb));
      if (m % 3 == 2) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a2) : "v"(wa), "v"(wb));
#pragma unroll
      for (int j = 0; j < K; ++j) {
        if (KIND == 0) {  // independent fp32 VALU (rotating registers)
          if (j % 4 == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(v4), "v"(v5));
          if (j % 4 == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v1) : "v"(v4), "v"(v5));
          if (j % 4 == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v2) : "v"(v4), "v"(v5));
          if (j % 4 == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v3) : "v"(v4), "v"(v5));
        } else if (KIND == 1) {  // dependent fp32 VALU chain
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(v4), "v"(v5));
        } else if (KIND == 2) {  // independent fp64 VALU
          if (j % 2 == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d0) : "v"(d2), "v"(d3));
          if (j % 2 == 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d1) : "v"(d2), "v"(d3));
        } else if (KIND == 3) {  // SALU
          asm volatile("s_add_u32 s90, s90, 1" ::: "s90", "scc");
        } else if (KIND == 4) {  // ds_read_b128
          f32x4 r;
          asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"((unsigned)(threadIdx.x * 16)));
        } else if (KIND == 5) {  // v_mov (cheapest VALU)
          asm volatile("v_mov_b32 %0, %1" : "=v"(v0) : "v"(v4));
        } else if (KIND == 6) {  // s_nop 0
          asm volatile("s_nop 0");
        }
      }
    }
    if (KIND == 4) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + v0 + v1 + v2 + v3 + (float)(d0 + d1);
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int K> void run(const char *name) {
  float *out; unsigned long long *cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  const int iters = 2000;
  hipLaunchKernelGGL((k<KIND, K>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((k<KIND, K>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(256);
  hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
  double s = 0; for (auto x : h) s += x;
  printf("%-22s K=%d: %.1f cycles per MFMA\n", name, K, s / 256 / iters / 12);
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0, 0>("bare");
  run<0, 1>("fp32 indep"); run<0, 2>("fp32 indep"); run<0, 4>("fp32 indep"); run<0, 6>("fp32 indep"); run<0, 8>("fp32 indep");
  run<1, 1>("fp32 dependent"); run<1, 2>("fp32 dependent"); run<1, 4>("fp32 dependent");
  run<2, 1>("fp64 indep"); run<2, 2>("fp64 indep"); run<2, 4>("fp64 indep");
  run<3, 2>("salu"); run<3, 6>("salu");
  run<4, 1>("ds_read_b128"); run<4, 2>("ds_read_b128");
  run<5, 2>("v_mov"); run<5, 4>("v_mov"); run<5, 8>("v_mov");
  run<6, 4>("s_nop"); run<6, 8>("s_nop");
  return 0;
}
