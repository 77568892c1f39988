// Instantiations and launcher of the matrix-core Helmholtz kernel (P = 12, fp32).
#include <stdlib.h>
#include <string.h>

#include "sfem_helmholtz_mfma.h"

namespace sfem {

int launch_helmholtz_mfma_p12(const HelmholtzParams<float>& prm,
                              hipStream_t stream) {
  constexpr int P = 12;
  if (prm.num_elements > 0x7fffffff) {
    set_error("helmholtz (mfma): too many elements (%lld)",
              (long long)prm.num_elements);
    return SFEM_EINVAL;
  }
  MfmaConsts<P> cst;
  memcpy(cst.d, prm.dmat_host, sizeof(cst.d));
  memcpy(cst.w, prm.weights_host, sizeof(cst.w));
  memcpy(cst.x, prm.nodes_host, sizeof(cst.x));
  const dim3 grid((unsigned)prm.num_elements), block(256);
  const bool mass = prm.lambda0 != 0.f;
#define SFEM_LAUNCH_MFMA(GMV)                                                 \
  do {                                                                        \
    if (mass)                                                                 \
      hipLaunchKernelGGL((helmholtz_mfma_p12_kernel<GMV, true>), grid, block, \
                         0, stream, cst, prm);                                \
    else                                                                      \
      hipLaunchKernelGGL((helmholtz_mfma_p12_kernel<GMV, false>), grid,       \
                         block, 0, stream, cst, prm);                         \
  } while (0)
  if (prm.geo_mode == GEO_AFFINE) SFEM_LAUNCH_MFMA(GEO_AFFINE);
  else SFEM_LAUNCH_MFMA(GEO_MULTILINEAR);
#undef SFEM_LAUNCH_MFMA
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

}  // namespace sfem
