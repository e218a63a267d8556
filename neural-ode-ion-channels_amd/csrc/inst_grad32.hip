// N = 500 (architectures s06-s08) backward sweep and regression tile: 512 registers per lane are all in use, so this unit is
// compiled with -sink-insts-to-avoid-spills on top of -disable-machine-licm (no scratch: tests/test_kernel_resources.py).  The
// same flag costs the N = 200 kernels 1.4 % (0.890 -> 0.903 s config-5 sweep, 1.96 -> 1.99 ms regression step, same box), hence
// the separate unit.
#define IONODE_GRAD_TEMPLATES_ONLY
#include "ionode_grad_launch.hpp"
namespace ionode {
SweepFn pick_sweep32(int model, int f32) { return pick_sweep<32>(model, f32); }
void launch_regress32(const RArgs &a, unsigned grid, hipStream_t s) { launch_regress<32>(a, grid, s); }
}  // namespace ionode
