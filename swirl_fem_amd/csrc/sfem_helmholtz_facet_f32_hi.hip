// Instantiations of the facet-table Helmholtz kernels: float, P = 9..12
// (several waves per element).
#include "sfem_helmholtz_facet.h"
namespace sfem {
SFEM_DEFINE_FACET_DISPATCH_HIGH(float)
}  // namespace sfem
