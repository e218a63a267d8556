// Probe (GPU box): operand / result layout and arithmetic of v_mfma_f32_4x4x1_16B_f32 on gfx950, and its dependent-issue rate.
//   hipcc --offload-arch=gfx950 -O3 mfma4x4.hip -o mfma4x4 && ./mfma4x4
// Expected (to be confirmed by this probe; the 4-trajectory tile is built on it):
//   A: lane l -> block l / 4, row i = l % 4;   B: lane l -> block l / 4, column j = l % 4;
//   D: VGPR v of lane l -> block l / 4, row i = v, column j = l % 4;     D = fma(A, B, C) exactly (K = 1: one fused multiply-add)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
using f32x4 = __attribute__((ext_vector_type(4))) float;
__global__ void probe(const float *a, const float *b, const float *c, float *d) {
  const int l = threadIdx.x;
  f32x4 acc = {c[l * 4 + 0], c[l * 4 + 1], c[l * 4 + 2], c[l * 4 + 3]};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
  for (int v = 0; v < 4; ++v) d[l * 4 + v] = acc[v];
}
__global__ void rate(float *out, int n, int dep) {
  const int l = threadIdx.x;
  f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  const float x = 1.0f + l * 1e-3f, y = 0.5f;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < n; ++i) {
    if (dep) {
#pragma unroll
      for (int u = 0; u < 16; ++u) a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0);
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) { a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, x, a1, 0, 0, 0); }
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  out[l] = a0[0] + a1[1];
  if (l == 0) out[64] = (float)(t1 - t0) / (16.0f * n);
}
int main() {
  float ha[64], hb[64], hc[256], hd[256], *a, *b, *c, *d;
  for (int l = 0; l < 64; ++l) { ha[l] = 1.0f + l; hb[l] = 100.0f + 3.0f * l; }
  for (int e = 0; e < 256; ++e) hc[e] = 0.25f * e;
  hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&c, 1024); hipMalloc(&d, 1100);
  hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice); hipMemcpy(c, hc, 1024, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(a, b, c, d);
  hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int v = 0; v < 4; ++v) {
      const int blk = l / 4, j = l % 4, i = v;
      const float want = fmaf(ha[4 * blk + i], hb[4 * blk + j], hc[l * 4 + v]);
      if (want != hd[l * 4 + v]) { if (bad < 8) printf("MISMATCH lane %d vgpr %d: got %g want %g\n", l, v, hd[l * 4 + v], want); ++bad; }
    }
  printf("layout/arith check: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
  // rounding: fma vs mul+add
  for (int l = 0; l < 64; ++l) { ha[l] = 1.0f + ldexpf(1.0f, -12) * (l + 1); hb[l] = 1.0f + ldexpf(1.0f, -13) * (3 * l + 1); }
  for (int e = 0; e < 256; ++e) hc[e] = -1.0f;
  hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice); hipMemcpy(c, hc, 1024, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(a, b, c, d);
  hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
  int nf = 0, nm = 0;
  for (int l = 0; l < 64; ++l)
    for (int v = 0; v < 4; ++v) {
      const int blk = l / 4, j = l % 4, i = v;
      volatile float p = ha[4 * blk + i] * hb[4 * blk + j];
      nf += (fmaf(ha[4 * blk + i], hb[4 * blk + j], -1.0f) == hd[l * 4 + v]);
      nm += ((p + -1.0f) == hd[l * 4 + v]);
    }
  printf("fused: %d / 256 equal fmaf, %d / 256 equal mul+add\n", nf, nm);
  rate<<<1, 64>>>(d, 2000, 1); hipMemcpy(hd, d, 260, hipMemcpyDeviceToHost); printf("dependent chain:   %.2f cycles per MFMA\n", hd[64]);
  rate<<<1, 64>>>(d, 2000, 0); hipMemcpy(hd, d, 260, hipMemcpyDeviceToHost); printf("two chains interleaved: %.2f cycles per MFMA\n", hd[64]);
  return bad != 0;
}
