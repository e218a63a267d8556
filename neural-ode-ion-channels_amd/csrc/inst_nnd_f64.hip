// MLP RHS kernels, model nnd (IONODE_MODEL id 3), double state.  (G, RT): wavefronts per tile, row tiles per wavefront.
#include "ionode_launch.hpp"
namespace ionode {
static const Variant kTab[] = {
    IONODE_VARIANT(3, double, 0, 1, 1), IONODE_VARIANT(3, double, 0, 4, 2),
    IONODE_VARIANT(3, double, 0, 4, 4), IONODE_VARIANT(3, double, 0, 4, 8),
};
const Variant *variants_nnd_f64(int *n) { *n = sizeof(kTab) / sizeof(kTab[0]); return kTab; }
}  // namespace ionode
