// Instantiations of the facet-table Stokes kernels: float, P = 6..8.
#include "sfem_stokes_facet.h"
namespace sfem {
SFEM_DEFINE_STOKES_FACET_DISPATCH(float)
}  // namespace sfem
