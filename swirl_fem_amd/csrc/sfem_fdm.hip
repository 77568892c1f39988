// Element-wise fast diagonalisation solve (the local part of the two-level
// Schwarz pressure preconditioner, navier_stokes/pressure_preconditioner.py):
//
//   z_e = (S_0 (x) .. (x) S_{d-1}) [ w_e .* (S_0 (x) .. (x) S_{d-1})^T r_e ]
//
// r_e: the Pp^d values of element e (the pressure nodes of an element belong
// to it alone), S_a = S[cases[a][e]] the Pp x Pp eigenvector matrix of the
// element's 1D generalised eigenproblem along its axis a (rows = nodes,
// columns = modes), w_e = the pseudo-inverted eigenvalues of the element
// block.  Beyond the reference (which ships no pressure preconditioner for its
// hook, navier_stokes/navier_stokes.py:354, :449-452).
//
// One workgroup per element: the tensor lives in LDS (two copies, Pp <= 10),
// each of the 2 d mode products is Pp^(d+1) multiply-adds spread over the 64
// lanes.  HBM-bound: r, w and z once = 3 s Pp^d bytes per element.
#include "sfem_common.h"

namespace sfem {

constexpr int FDM_MAX_P = 10;

template <typename T>
__global__ void __launch_bounds__(64)
fdm_solve_kernel(const T* __restrict__ r, T* __restrict__ z,
                 const int64_t* __restrict__ pel, const T* __restrict__ S,
                 const int32_t* __restrict__ cases, const T* __restrict__ w,
                 int64_t num_elements, int d, int Pp) {
  __shared__ T buf[2][FDM_MAX_P * FDM_MAX_P * FDM_MAX_P];
  __shared__ T mat[3][FDM_MAX_P * FDM_MAX_P];
  const int64_t e = blockIdx.x;
  const int lane = threadIdx.x;
  int n = 1;
  for (int a = 0; a < d; ++a) n *= Pp;
  const int pp2 = Pp * Pp;
  for (int a = 0; a < d; ++a) {
    const T* Sa = S + (int64_t)cases[a * num_elements + e] * pp2;
    for (int q = lane; q < pp2; q += 64) mat[a][q] = Sa[q];
  }
  const int64_t base = e * n;
  for (int q = lane; q < n; q += 64)
    buf[0][q] = r[pel ? pel[base + q] : base + q];
  __syncthreads();
  int cur = 0;
  // forward: t_m = sum_i S[i][m] r_i along every axis
  for (int a = 0; a < d; ++a) {
    int stride = 1;
    for (int b = a + 1; b < d; ++b) stride *= Pp;
    for (int o = lane; o < n; o += 64) {
      const int post = o % stride, m = (o / stride) % Pp;
      const int pre = o / (stride * Pp);
      const T* in = &buf[cur][pre * Pp * stride + post];
      T acc = T(0);
      for (int i = 0; i < Pp; ++i) acc += mat[a][i * Pp + m] * in[i * stride];
      buf[cur ^ 1][o] = acc;
    }
    cur ^= 1;
    __syncthreads();
  }
  for (int q = lane; q < n; q += 64) buf[cur][q] *= w[base + q];
  __syncthreads();
  // backward: z_i = sum_m S[i][m] t_m
  for (int a = 0; a < d; ++a) {
    int stride = 1;
    for (int b = a + 1; b < d; ++b) stride *= Pp;
    for (int o = lane; o < n; o += 64) {
      const int post = o % stride, i = (o / stride) % Pp;
      const int pre = o / (stride * Pp);
      const T* in = &buf[cur][pre * Pp * stride + post];
      T acc = T(0);
      for (int m = 0; m < Pp; ++m) acc += mat[a][i * Pp + m] * in[m * stride];
      buf[cur ^ 1][o] = acc;
    }
    cur ^= 1;
    __syncthreads();
  }
  for (int q = lane; q < n; q += 64)
    z[pel ? pel[base + q] : base + q] = buf[cur][q];
}

}  // namespace sfem

using namespace sfem;

extern "C" int sfem_fdm_solve(const void* r, void* z, const int64_t* pel,
                              const void* S, const int32_t* cases,
                              const void* inv_eigenvalues,
                              int64_t num_elements, int ndim, int Pp,
                              int dtype, sfem_stream_t stream) {
  SFEM_REQUIRE(num_elements >= 0 && ndim >= 1 && ndim <= 3 && Pp >= 1 &&
                   Pp <= FDM_MAX_P,
               "sfem_fdm_solve: ndim 1..3, 1 <= Pp <= %d", FDM_MAX_P);
  if (num_elements == 0) return SFEM_OK;
  SFEM_REQUIRE(r && z && S && cases && inv_eigenvalues,
               "sfem_fdm_solve: null pointer");
  SFEM_REQUIRE(num_elements <= 0x7fffffff, "sfem_fdm_solve: too many elements");
  const dim3 grid((unsigned)num_elements), block(64);
  if (dtype == SFEM_F64)
    hipLaunchKernelGGL(fdm_solve_kernel<double>, grid, block, 0,
                       as_stream(stream), (const double*)r, (double*)z, pel,
                       (const double*)S, cases, (const double*)inv_eigenvalues,
                       num_elements, ndim, Pp);
  else if (dtype == SFEM_F32)
    hipLaunchKernelGGL(fdm_solve_kernel<float>, grid, block, 0,
                       as_stream(stream), (const float*)r, (float*)z, pel,
                       (const float*)S, cases, (const float*)inv_eigenvalues,
                       num_elements, ndim, Pp);
  else {
    set_error("sfem_fdm_solve: unknown dtype %d", dtype);
    return SFEM_EINVAL;
  }
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}
