// ionode_grad_launch.hpp -- launchers of the backward sweep, shared by the translation units that instantiate it
// (ionode_grad_capi.hip: N = 10 / 100 / 200; inst_grad32.hip: N = 500, compiled with instruction sinking).
#pragma once
#include "ionode_grad.hpp"
#include "ionode_regress.hpp"

namespace ionode {

using SweepFn = void (*)(const GArgs &, unsigned grid, size_t lds, hipStream_t);

// a.phase 0: the one-phase sweep.  1: phase A of the two-phase sweep (ionode_grad_recompute_kernel, every (tile, step) at once);
// 2: phase B (ionode_grad_walk_kernel: adjoint algebra + backward products; + 8 KiB of LDS for the step's packets).
template <int MODEL, typename S, int NT>
void launch_sweep(const GArgs &a, unsigned grid, size_t lds, hipStream_t s) {
  if constexpr (ModelTraits<MODEL>::MLP) {
    if (a.phase == 1) {
      auto kern = ionode_grad_recompute_kernel<MODEL, S, NT>;
      if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      const unsigned nb = (unsigned)((a.it_end - a.it_begin + GRAD_RECOMPUTE_IB - 1) / GRAD_RECOMPUTE_IB);
      hipLaunchKernelGGL(kern, dim3(grid, nb), dim3(256), lds, s, a);
      return;
    }
    if (a.phase == 2) {
      auto kern = ionode_grad_walk_kernel<MODEL, S>;   // one wavefront per tile, the step's 16 packets in LDS
      hipLaunchKernelGGL(kern, dim3(grid), dim3(64), (size_t)16 * GRAD_PACKET * 8, s, a);
      return;
    }
  }
  auto kern = ionode_dopri5_backward_kernel<MODEL, S, NT>;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
}

template <int NT> SweepFn pick_sweep(int model, int f32) {
  if (model == IONODE_MODEL_NNF) return f32 ? &launch_sweep<IONODE_MODEL_NNF, float, NT> : &launch_sweep<IONODE_MODEL_NNF, double, NT>;
  return f32 ? &launch_sweep<IONODE_MODEL_NND, float, NT> : &launch_sweep<IONODE_MODEL_NND, double, NT>;
}

// inst_grad32.hip
SweepFn pick_sweep32(int model, int f32);
void launch_regress32(const RArgs &a, unsigned grid, hipStream_t s);

}  // namespace ionode
