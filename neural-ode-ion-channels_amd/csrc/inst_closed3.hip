// HH 2-state kernels at THREE wavefronts per SIMD (launches beyond one residency round).  Own translation unit because it is compiled
// with -mllvm -disable-machine-licm (Makefile): with machine-LICM hipcc hoists every materialised fp64 constant out of the attempt
// loop, exceeds the 168-register budget and spills 10-22 VGPRs to scratch inside the stage loop (72 B / lane, 11-22 GB of extra
// fetch per launch); without it the kernels need 154 registers and no scratch: -4 ... -6 % time.  The two-per-SIMD builds keep the
// hoisted constants (238 registers, no spill either; without LICM they lose 24 %).
#include "ionode_launch.hpp"
namespace ionode {
static const Variant kTab[] = {
    IONODE_VARIANT(0, double, 0, 1, 0, 0, 0, 0), IONODE_VARIANT(0, float, 1, 1, 0, 0, 0, 0),
    IONODE_VARIANT(0, double, 0, 1, 16, 0, 0, 0), IONODE_VARIANT(0, float, 1, 1, 16, 0, 0, 0),
    IONODE_VARIANT(0, double, 0, 1, 0, 0, 0, 1), IONODE_VARIANT(0, float, 1, 1, 0, 0, 0, 1),
    IONODE_VARIANT(0, double, 0, 1, 16, 0, 0, 1), IONODE_VARIANT(0, float, 1, 1, 16, 0, 0, 1),
    IONODE_VARIANT(0, double, 0, 1, 0, 0, 0, 2), IONODE_VARIANT(0, float, 1, 1, 0, 0, 0, 2),
    IONODE_VARIANT(0, double, 0, 1, 16, 0, 0, 2), IONODE_VARIANT(0, float, 1, 1, 16, 0, 0, 2),
};
const Variant *variants_closed3(int *n) { *n = sizeof(kTab) / sizeof(kTab[0]); return kTab; }
}  // namespace ionode
