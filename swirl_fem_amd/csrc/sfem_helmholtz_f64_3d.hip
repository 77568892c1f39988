// Instantiations of the fused Helmholtz kernel: double, 3D, P = 2..12.
#include "sfem_helmholtz.h"
namespace sfem {
SFEM_DEFINE_HELMHOLTZ_DISPATCH(double, 3)
}  // namespace sfem
