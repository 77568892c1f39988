// fp64 instantiations of the compile-time-sized tensor interpolation
// (sfem_interp.h), dispatched from sfem_basis_eval / sfem_basis_eval_t.
#include "sfem_interp.h"

namespace sfem {
int launch_tensor_interp_f64(int ndim, int ni, int no, const void* in,
                             const void* mat, const void* weight, void* out,
                             int64_t E, int nc, bool trans, hipStream_t st) {
  return launch_tensor_interp_t<double>(ndim, ni, no, in, mat, weight, out, E,
                                        nc, trans, st);
}
}  // namespace sfem
