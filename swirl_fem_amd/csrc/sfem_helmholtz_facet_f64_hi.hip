// Instantiations of the facet-table Helmholtz kernels: double, P = 9..12
// (several waves per element).
#include "sfem_helmholtz_facet.h"
namespace sfem {
SFEM_DEFINE_FACET_DISPATCH_HIGH(double)
}  // namespace sfem
