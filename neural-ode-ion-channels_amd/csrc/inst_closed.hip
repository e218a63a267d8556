// Closed-form RHS kernels: 6-state Markov, and the HH 2-state kernels at two wavefronts per SIMD (the three-per-SIMD builds: inst_closed3.hip).
#include "ionode_launch.hpp"
namespace ionode {
// 4th parameter (RT slot) = trajectories per wavefront: 0 -> 64 (one per lane), 16 -> 16 (small batches)
static const Variant kTab[] = {
    IONODE_VARIANT(1, double, 0, 1, 0, 0, 0, 0), IONODE_VARIANT(1, float, 1, 1, 0, 0, 0, 0),
    IONODE_VARIANT(1, double, 0, 1, 16, 0, 0, 0), IONODE_VARIANT(1, float, 1, 1, 16, 0, 0, 0),
    // last parameter 1: deferred aligned emission (2-state models, exact uniform output grid, no current trace)
    // last parameter 2: current / objective epilogue through the protocol-at-outputs table (ionode_desc.v_at_outputs)
    IONODE_VARIANT(1, double, 0, 1, 0, 0, 0, 2), IONODE_VARIANT(1, float, 1, 1, 0, 0, 0, 2),
    IONODE_VARIANT(1, double, 0, 1, 16, 0, 0, 2), IONODE_VARIANT(1, float, 1, 1, 16, 0, 0, 2),
    // NT slot 2: the 2-state kernels again at 2 wavefronts per SIMD (no register spill), dispatched when the launch is at most 2048 wavefronts
    IONODE_VARIANT(0, double, 0, 1, 0, 2, 0, 0), IONODE_VARIANT(0, float, 1, 1, 0, 2, 0, 0),
    IONODE_VARIANT(0, double, 0, 1, 16, 2, 0, 0), IONODE_VARIANT(0, float, 1, 1, 16, 2, 0, 0),
    IONODE_VARIANT(0, double, 0, 1, 0, 2, 0, 1), IONODE_VARIANT(0, float, 1, 1, 0, 2, 0, 1),
    IONODE_VARIANT(0, double, 0, 1, 16, 2, 0, 1), IONODE_VARIANT(0, float, 1, 1, 16, 2, 0, 1),
    IONODE_VARIANT(0, double, 0, 1, 0, 2, 0, 2), IONODE_VARIANT(0, float, 1, 1, 0, 2, 0, 2),
    IONODE_VARIANT(0, double, 0, 1, 16, 2, 0, 2), IONODE_VARIANT(0, float, 1, 1, 16, 2, 0, 2),
};
const Variant *variants_closed(int *n) { *n = sizeof(kTab) / sizeof(kTab[0]); return kTab; }
}  // namespace ionode
