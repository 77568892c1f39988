// Instantiations of the facet-table Helmholtz kernels: float, P = 6..8.
#include "sfem_helmholtz_facet.h"
namespace sfem {
SFEM_DEFINE_FACET_DISPATCH_LOW(float)
}  // namespace sfem
