// Probe (GPU box): dependent-issue cost of the vector fp32 chain forms the one-trajectory tile could be built on (gfx950).
//   hipcc --offload-arch=gfx950 -O3 valu_chain.hip -o valu_chain && ./valu_chain
// Per form: cycles per chain link (s_memtime deltas / links) with ONE wavefront on the SIMD.
//   0 v_fmac_f32 (dependent)                      1 v_fmac_f32_dpp quad_perm (dependent)
//   2 v_fmac_f32_dpp + s_nop 0 between (what hipcc emits around inline asm)
//   3 two interleaved independent v_fmac_f32_dpp chains      4 v_pk_fma_f32 dependent (two chains per instruction)
//   5 four interleaved independent dpp chains                6 v_fma_f32 (VOP3, dependent)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int FORM> __global__ void chain(float *out, int n) {
  const int l = threadIdx.x;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  float h = 1.0f + l * 1e-3f, w = 0.5f + l * 1e-4f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p = {0.f, 0.f}, hh = {h, h}, ww = {w, w};
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (FORM == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a0) : "v"(h), "v"(w));
      if (FORM == 1) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "+v"(a0) : "v"(h), "v"(w));
      if (FORM == 2) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\ts_nop 0" : "+v"(a0) : "v"(h), "v"(w));
      if (FORM == 3) asm volatile("v_fmac_f32_dpp %0, %2, %3 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %1, %2, %3 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1) : "v"(h), "v"(w));
      if (FORM == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p) : "v"(hh), "v"(ww));
      if (FORM == 5) asm volatile("v_fmac_f32_dpp %0, %4, %5 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %1, %4, %5 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
                                  "v_fmac_f32_dpp %2, %4, %5 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %3, %4, %5 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(h), "v"(w));
      if (FORM == 6) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(h), "v"(w));
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  out[l] = a0 + a1 + a2 + a3 + p.x + p.y;
  if (l == 0) out[64 + FORM] = (float)(t1 - t0) / (16.0f * n);
}
int main() {
  float *d, h[80];
  hipMalloc(&d, sizeof h);
  const int n = 4096;
  chain<0><<<1, 64>>>(d, n); chain<1><<<1, 64>>>(d, n); chain<2><<<1, 64>>>(d, n); chain<3><<<1, 64>>>(d, n);
  chain<4><<<1, 64>>>(d, n); chain<5><<<1, 64>>>(d, n); chain<6><<<1, 64>>>(d, n);
  hipDeviceSynchronize();
  chain<0><<<1, 64>>>(d, n); chain<1><<<1, 64>>>(d, n); chain<2><<<1, 64>>>(d, n); chain<3><<<1, 64>>>(d, n);
  chain<4><<<1, 64>>>(d, n); chain<5><<<1, 64>>>(d, n); chain<6><<<1, 64>>>(d, n);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char *name[] = {"v_fmac_f32 dependent", "v_fmac_f32_dpp dependent", "v_fmac_f32_dpp + s_nop 0", "2 interleaved dpp chains (per pair)",
                        "v_pk_fma_f32 dependent", "4 interleaved dpp chains (per group of 4)", "v_fma_f32 dependent"};
  for (int f = 0; f < 7; ++f) printf("form %d %-44s %.2f s_memtime ticks per asm statement\n", f, name[f], h[64 + f]);
  printf("(s_memtime ticks at 100 MHz: multiply by shader clock / 100 MHz for cycles)\n");
  return 0;
}
