// Instantiations of the fused Helmholtz kernel: double, 2D, P = 2..12.
#include "sfem_helmholtz.h"
namespace sfem {
SFEM_DEFINE_HELMHOLTZ_DISPATCH(double, 2)
}  // namespace sfem
