// Instantiations of the fused Helmholtz kernel: float, 2D, P = 2..12.
#include "sfem_helmholtz.h"
namespace sfem {
SFEM_DEFINE_HELMHOLTZ_DISPATCH(float, 2)
}  // namespace sfem
