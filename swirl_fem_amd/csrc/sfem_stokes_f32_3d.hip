// Instantiations of the fused Stokes divergence / gradient kernels:
// float, 3D, P = 4..12 (pressure on P - 2 Gauss nodes).
#include "sfem_stokes.h"
namespace sfem {
SFEM_DEFINE_STOKES_DISPATCH(float, 3)
}  // namespace sfem
