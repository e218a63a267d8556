// Dev check (GPU box): det_exp_ldexp == det_exp bit for bit, including subnormal results and the overflow edge.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I include tools/check_det_exp.hip -o /tmp/check_det_exp && /tmp/check_det_exp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../neural-ode-ion-channels_amd/csrc/ionode_device.hpp"

__global__ void k(const double *x, double *a, double *b, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { a[i] = ionode::det_exp(x[i]); b[i] = ionode::det_exp_ldexp(x[i]); }
}

int main() {
  std::vector<double> x;
  for (int i = 0; i <= 2000000; ++i) x.push_back(-760.0 + 1470.0 * i / 2000000.0);          // whole range, both edges
  for (int i = 0; i <= 2000000; ++i) x.push_back(-745.2 + 37.5 * i / 2000000.0);            // subnormal results
  for (int i = 0; i <= 200000; ++i) x.push_back(709.0 + 0.8 * i / 200000.0);                // overflow edge
  const int n = (int)x.size();
  double *dx, *da, *db;
  hipMalloc(&dx, n * 8); hipMalloc(&da, n * 8); hipMalloc(&db, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, dx, da, db, n);
  std::vector<double> a(n), b(n);
  hipMemcpy(a.data(), da, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, n * 8, hipMemcpyDeviceToHost);
  long bad = 0, sub = 0;
  for (int i = 0; i < n; ++i) {
    if (memcmp(&a[i], &b[i], 8) != 0) { if (bad < 5) printf("x=%.17g  %a  %a\n", x[i], a[i], b[i]); ++bad; }
    if (a[i] > 0 && a[i] < 2.2250738585072014e-308) ++sub;
  }
  printf("%d values, %ld subnormal results, %ld mismatches\n", n, sub, bad);
  return bad != 0;
}
