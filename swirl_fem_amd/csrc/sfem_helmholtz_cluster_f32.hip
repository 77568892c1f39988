// Instantiations of the cluster-assembled Helmholtz kernel: float, P = 4..8.
#include "sfem_helmholtz_cluster.h"
namespace sfem {
SFEM_DEFINE_HELMHOLTZ_CLUSTER_DISPATCH(float)
}  // namespace sfem
