// Closed-form RHS kernels: 6-state Markov and HH 2-state.  Compiled without machine-LICM (Makefile): their fp64 constants are scalar operands.
#include "ionode_launch.hpp"
namespace ionode {
// 4th parameter (RT slot) = trajectories per wavefront: 0 -> 64 (one per lane), 16 -> 16 (small batches)
static const Variant kTab[] = {
    IONODE_VARIANT(1, double, 0, 1, 0, 0, 0, 0), IONODE_VARIANT(1, float, 1, 1, 0, 0, 0, 0),
    IONODE_VARIANT(1, double, 0, 1, 16, 0, 0, 0), IONODE_VARIANT(1, float, 1, 1, 16, 0, 0, 0),
    // last parameter 1: states only on a verified uniform output grid (the lean variant)
    IONODE_VARIANT(1, double, 0, 1, 0, 0, 0, 1), IONODE_VARIANT(1, float, 1, 1, 0, 0, 0, 1),
    IONODE_VARIANT(1, double, 0, 1, 16, 0, 0, 1), IONODE_VARIANT(1, float, 1, 1, 16, 0, 0, 1),
    // last parameter 2: current / objective epilogue through the protocol-at-outputs table (ionode_desc.v_at_outputs)
    IONODE_VARIANT(1, double, 0, 1, 0, 0, 0, 2), IONODE_VARIANT(1, float, 1, 1, 0, 0, 0, 2),
    IONODE_VARIANT(1, double, 0, 1, 16, 0, 0, 2), IONODE_VARIANT(1, float, 1, 1, 16, 0, 0, 2),
    // HH 2-state.  __launch_bounds__ asks for TWO wavefronts per SIMD (a 256-register budget): hipcc then allocates 147-157 registers,
    // which the hardware runs at THREE per SIMD anyway (<= 168); asked for three, its scheduler fills the 168 and spills 2-6 (round 4)
    IONODE_VARIANT(0, double, 0, 1, 0, 0, 0, 0), IONODE_VARIANT(0, float, 1, 1, 0, 0, 0, 0),
    IONODE_VARIANT(0, double, 0, 1, 16, 0, 0, 0), IONODE_VARIANT(0, float, 1, 1, 16, 0, 0, 0),
    IONODE_VARIANT(0, double, 0, 1, 0, 0, 0, 1), IONODE_VARIANT(0, float, 1, 1, 0, 0, 0, 1),
    IONODE_VARIANT(0, double, 0, 1, 16, 0, 0, 1), IONODE_VARIANT(0, float, 1, 1, 16, 0, 0, 1),
    IONODE_VARIANT(0, double, 0, 1, 0, 0, 0, 2), IONODE_VARIANT(0, float, 1, 1, 0, 0, 0, 2),
    IONODE_VARIANT(0, double, 0, 1, 16, 0, 0, 2), IONODE_VARIANT(0, float, 1, 1, 16, 0, 0, 2),
};
const Variant *variants_closed(int *n) { *n = sizeof(kTab) / sizeof(kTab[0]); return kTab; }
}  // namespace ionode
