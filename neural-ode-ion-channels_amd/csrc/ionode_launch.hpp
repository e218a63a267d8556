// ionode_launch.hpp -- host-side table of compiled kernel instantiations.
#pragma once
#include "ionode_device.hpp"

namespace ionode {

using LaunchFn = hipError_t (*)(const KArgs &, unsigned grid, size_t lds, hipStream_t);

struct Variant {
  int model, f32, G, RT;
  LaunchFn fn;
  const char *name;  // as rocprofv3 --kernel-trace prints it
};

template <int MODEL, typename S, int G, int RT>
hipError_t launch(const KArgs &a, unsigned grid, size_t lds, hipStream_t s) {
  auto kern = ionode_dopri5_kernel<MODEL, S, G, RT>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * G), lds, s, a);
  return hipGetLastError();
}

#define IONODE_VARIANT(MODEL, S, F32, G, RT) \
  Variant { MODEL, F32, G, RT, &launch<MODEL, S, G, RT>, "ionode_dopri5_kernel<" #MODEL ", " #S ", " #G ", " #RT ">" }

// one table per translation unit (they compile in parallel)
const Variant *variants_closed(int *n);
const Variant *variants_nnf_f64(int *n);
const Variant *variants_nnf_f32(int *n);
const Variant *variants_nnd_f64(int *n);
const Variant *variants_nnd_f32(int *n);

}  // namespace ionode
