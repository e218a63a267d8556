// MLP RHS kernels, model nnf (IONODE_MODEL id 2), double state.  (G, RT, NT, PD): wavefronts per tile, row tiles per wavefront, k-tiles, ring depth.
#include "ionode_launch.hpp"
namespace ionode {
static const Variant kTab[] = {IONODE_MLP_VARIANTS(2, double, 0)};
const Variant *variants_nnf_f64(int *n) { *n = sizeof(kTab) / sizeof(kTab[0]); return kTab; }
}  // namespace ionode
