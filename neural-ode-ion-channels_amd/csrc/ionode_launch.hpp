// ionode_launch.hpp -- host-side table of compiled kernel instantiations.
#pragma once
#include "ionode_device.hpp"

namespace ionode {

using LaunchFn = hipError_t (*)(const KArgs &, unsigned grid, size_t lds, hipStream_t);

struct Variant {
  int model, f32, G, RT, NT, PD;  // NT = k-tiles (16*NT = padded MLP width), PD = weight-ring depth
  int tail;                       // lane-wise kernels: 1 = lean variant (ionode_device.hpp LEAN: states only, verified uniform grids), 2 = epilogue through v_at_outputs
  LaunchFn fn;
  const char *name;  // as rocprofv3 --kernel-trace prints it
};

template <int MODEL, typename S, int G, int RT, int NT, int PD, int TAIL>
hipError_t launch(const KArgs &a, unsigned grid, size_t lds, hipStream_t s) {
  auto kern = ionode_dopri5_kernel<MODEL, S, G, RT, NT, PD, TAIL>;
  // the lane-wise kernels' LDS region is laid out from the SAME template constants the kernel uses: a plan that reserved less
  // (a host / device layout mismatch) is refused here instead of becoming an out-of-bounds LDS access on the device
  constexpr bool mlp = (MODEL == IONODE_MODEL_NNF || MODEL == IONODE_MODEL_NND);
  if constexpr (!mlp || RT == 64) {
    constexpr int D = (MODEL == IONODE_MODEL_MARKOV6) ? 6 : 2;
    const size_t need = (size_t)LwLds::bytes(D, mlp ? (TAIL == 1 ? 1 : 0) : TAIL);
    if ((size_t)a.lw_bytes < need || (a.lw_bytes & 15) || lds < (size_t)IONODE_LW_TILES_PER_WG * (size_t)a.lw_bytes) return hipErrorInvalidValue;
  }
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * (IONODE_IS_LW(MODEL, RT) ? IONODE_LW_TILES_PER_WG : G)), lds, s, a);
  return hipGetLastError();
}

#define IONODE_VARIANT(MODEL, S, F32, G, RT, NT, PD, TAIL) IONODE_VARIANT_(MODEL, S, F32, G, RT, NT, PD, TAIL)
#define IONODE_VARIANT_(MODEL, S, F32, G, RT, NT, PD, TAIL)                                   \
  Variant {                                                                                   \
    MODEL, F32, G, RT, NT, PD, TAIL, &launch<MODEL, S, G, RT, NT, PD, TAIL>,                          \
        "ionode_dopri5_kernel<" #MODEL ", " #S ", " #G ", " #RT ", " #NT ", " #PD ", " #TAIL ">" \
  }
// the MLP shapes of the reference's architectures/s00-s11.py: N = 10, 100, 200, 500
#define IONODE_MLP_VARIANTS(MODEL, S, F32)                                                      \
  IONODE_VARIANT(MODEL, S, F32, 1, 1, 1, 1, 0), IONODE_VARIANT(MODEL, S, F32, 4, 4, 7, 7, 0),        \
      /* N <= 16 at 64 trajectories per wavefront (RT slot 64), general and lean (TAIL 1) */ \
      IONODE_VARIANT(MODEL, S, F32, 1, 64, 1, 1, 0), IONODE_VARIANT(MODEL, S, F32, 1, 64, 1, 1, 1),      \
      /* ... and N = 10 with the net evaluated per lane on the vector ALU (PD slot 10: MlpLane) */      \
      IONODE_VARIANT(MODEL, S, F32, 1, 64, 1, 10, 0), IONODE_VARIANT(MODEL, S, F32, 1, 64, 1, 10, 1),    \
      IONODE_VARIANT(MODEL, S, F32, 4, 4, 13, 13, 0), IONODE_VARIANT(MODEL, S, F32, 4, 8, 32, 4, 0),     \
      /* N = 200 with two column sets per tile (TAIL slot 4: 32 trajectories per workgroup), launches of >= 512 such tiles' worth */ \
      IONODE_VARIANT(MODEL, S, F32, 4, 4, 13, 13, 4),                                                     \
      /* ... and both N = 200 tiles as LEAN variants (TAIL & 8: uniform protocol grid, verified output grid, no step log / checkpoints) */ \
      IONODE_VARIANT(MODEL, S, F32, 4, 4, 13, 13, 8), IONODE_VARIANT(MODEL, S, F32, 4, 4, 13, 13, 12),    \
      /* N = 200 at FOUR trajectories per tile (TAIL & 16: MlpTile4, small batches / single calls), general and lean */ \
      IONODE_VARIANT(MODEL, S, F32, 4, 4, 13, 13, 16), IONODE_VARIANT(MODEL, S, F32, 4, 4, 13, 13, 24),   \
      /* N = 200 at ONE trajectory per tile (TAIL & 32: MlpRow1, the reference's own odeint(func, y0, t) call shape), general and lean */ \
      IONODE_VARIANT(MODEL, S, F32, 4, 4, 13, 13, 32), IONODE_VARIANT(MODEL, S, F32, 4, 4, 13, 13, 40),   \
      /* ... and for stacks of 7 .. 15 hidden layers without LDS-resident steps (TAIL & 64: MlpRow1Deep) */ \
      IONODE_VARIANT(MODEL, S, F32, 4, 4, 13, 13, 96), IONODE_VARIANT(MODEL, S, F32, 4, 4, 13, 13, 104),  \
      /* lean variants of the N = 100 and N = 500 tiles */                                              \
      IONODE_VARIANT(MODEL, S, F32, 4, 4, 7, 7, 8), IONODE_VARIANT(MODEL, S, F32, 4, 8, 32, 4, 8),        \
      /* any other width up to 512 (NT slot 0: MlpGen, the k-tile count is a run-time value), general and lean */ \
      IONODE_VARIANT(MODEL, S, F32, 4, 1, 0, 1, 0), IONODE_VARIANT(MODEL, S, F32, 4, 1, 0, 1, 8)

// one table per translation unit (they compile in parallel)
const Variant *variants_closed(int *n);
const Variant *variants_nnf_f64(int *n);
const Variant *variants_nnf_f32(int *n);
const Variant *variants_nnd_f64(int *n);
const Variant *variants_nnd_f32(int *n);

}  // namespace ionode
