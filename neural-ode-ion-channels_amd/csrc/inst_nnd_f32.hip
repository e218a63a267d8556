// MLP RHS kernels, model nnd (IONODE_MODEL id 3), float state.  (G, RT): wavefronts per tile, row tiles per wavefront.
#include "ionode_launch.hpp"
namespace ionode {
static const Variant kTab[] = {
    IONODE_VARIANT(3, float, 1, 1, 1), IONODE_VARIANT(3, float, 1, 4, 2),
    IONODE_VARIANT(3, float, 1, 4, 4), IONODE_VARIANT(3, float, 1, 4, 8),
};
const Variant *variants_nnd_f32(int *n) { *n = sizeof(kTab) / sizeof(kTab[0]); return kTab; }
}  // namespace ionode
