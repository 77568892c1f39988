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

// PP = Pp and D = ndim at compile time: the index arithmetic of the mode
// products (divisions by powers of Pp) folds into shifts / multiplies and the
// inner sums unroll (a run-time Pp measured 2.16 ms at 64^3 elements, p = 7:
// 5 x the kernel's memory time).
template <typename T, int D, int PP>
__global__ void __launch_bounds__(64)
fdm_solve_kernel(const T* __restrict__ r, T* __restrict__ z,
                 const int64_t* __restrict__ pel, const T* __restrict__ S,
                 const int32_t* __restrict__ cases, const T* __restrict__ w,
                 int64_t num_elements, const T* __restrict__ weights,
                 T* __restrict__ elem_sum, T* __restrict__ weighted_sum) {
  constexpr int N = D == 3 ? PP * PP * PP : (D == 2 ? PP * PP : PP);
  __shared__ T buf[2][N];
  __shared__ T mat[D][PP * PP];
  const int64_t e = blockIdx.x;
  const int lane = threadIdx.x;
#pragma unroll
  for (int a = 0; a < D; ++a) {
    const T* Sa = S + (int64_t)cases[a * num_elements + e] * (PP * PP);
    for (int q = lane; q < PP * PP; q += 64) mat[a][q] = Sa[q];
  }
  const int64_t base = e * N;
  T rsum = T(0);
  for (int q = lane; q < N; q += 64) {
    const T v = r[pel ? pel[base + q] : base + q];
    buf[0][q] = v;
    rsum += v;
  }
  if (elem_sum) {            // R_0 r of the coarse level, while r is here
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) rsum += __shfl_down(rsum, off, 64);
    if (lane == 0) elem_sum[e] = rsum;
  }
  __syncthreads();
  int cur = 0;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int a = 0; a < D; ++a) {
      constexpr int P2 = PP * PP;
      const int stride = a == D - 1 ? 1 : (a == D - 2 ? PP : P2);
      for (int o = lane; o < N; o += 64) {
        const int post = o % stride, m = (o / stride) % PP;
        const int pre = o / (stride * PP);
        const T* in = &buf[cur][pre * PP * stride + post];
        T acc = T(0);
        if (pass == 0) {      // t_m = sum_i S[i][m] r_i
#pragma unroll
          for (int i = 0; i < PP; ++i)
            acc += mat[a][i * PP + m] * in[i * stride];
        } else {              // z_i = sum_m S[i][m] t_m  (m plays i here)
#pragma unroll
          for (int i = 0; i < PP; ++i)
            acc += mat[a][m * PP + i] * in[i * stride];
        }
        buf[cur ^ 1][o] = acc;
      }
      cur ^= 1;
      __syncthreads();
    }
    if (pass == 0) {
      for (int q = lane; q < N; q += 64) buf[cur][q] *= w[base + q];
      __syncthreads();
    }
  }
  T wsum = T(0);
  for (int q = lane; q < N; q += 64) {
    const int64_t node = pel ? pel[base + q] : base + q;
    const T v = buf[cur][q];
    z[node] = v;
    if (weighted_sum) wsum += weights[node] * v;
  }
  if (weighted_sum) {        // this element's share of weights . z
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wsum += __shfl_down(wsum, off, 64);
    if (lane == 0) weighted_sum[e] = wsum;
  }
}

// z[e][i] += yc[e] - shift[e / elems_per_member]  (element e owns the nodes
// [e n, (e + 1) n)): the coarse correction and the mean projection of the
// Schwarz preconditioner in one pass
template <typename T>
__global__ void __launch_bounds__(256)
add_element_constants_kernel(T* __restrict__ z, const T* __restrict__ yc,
                             const T* __restrict__ shift, int64_t total, int n,
                             int64_t elems_per_member) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += stride) {
    const int64_t e = t / n;
    z[t] += yc[e] - shift[e / elems_per_member];
  }
}

template <typename T, int D>
static int launch_fdm(int Pp, dim3 grid, hipStream_t st, const T* r, T* z,
                      const int64_t* pel, const T* S, const int32_t* cases,
                      const T* w, int64_t E, const T* weights, T* elem_sum,
                      T* weighted_sum) {
#define SFEM_FDM_CASE(PPV)                                                    \
  case PPV:                                                                   \
    hipLaunchKernelGGL((fdm_solve_kernel<T, D, PPV>), grid, dim3(64), 0, st,  \
                       r, z, pel, S, cases, w, E, weights, elem_sum,          \
                       weighted_sum);                                         \
    return SFEM_OK;
  switch (Pp) {
    SFEM_FDM_CASE(1) SFEM_FDM_CASE(2) SFEM_FDM_CASE(3) SFEM_FDM_CASE(4)
    SFEM_FDM_CASE(5) SFEM_FDM_CASE(6) SFEM_FDM_CASE(7) SFEM_FDM_CASE(8)
    SFEM_FDM_CASE(9) SFEM_FDM_CASE(10)
  }
#undef SFEM_FDM_CASE
  set_error("sfem_fdm_solve: Pp=%d outside 1..%d", Pp, FDM_MAX_P);
  return SFEM_EUNSUPPORTED;
}

// ---------------------------------------------------------------------------
// Coarse level of the same preconditioner: a fixed Chebyshev polynomial in the
// Jacobi-scaled sparse matrix A (rows of equal length, stored column-major:
// entry k of row i at [k * n + i]) applied to b,  x = q_m(D^-1 A) D^-1 b  with
// the residual polynomial of the interval [lmin, lmax].  No inner products,
// one launch per step, exactly linear and symmetric positive definite for any
// 0 < lmin as long as lmax bounds the spectrum -- what a preconditioner inside
// CG needs (a truncated inner CG is neither).
//   d_0 = D^-1 b / theta;  x += d;  r -= A d;
//   d <- c1 d + c2 D^-1 r   (c1 = rho' rho, c2 = 2 rho' / delta)
template <typename T>
__global__ void __launch_bounds__(256)
cheb_init_kernel(const T* __restrict__ b, const T* __restrict__ dinv,
                 T* __restrict__ x, T* __restrict__ r, T* __restrict__ d,
                 T inv_theta, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  x[i] = T(0);
  r[i] = b[i];
  d[i] = inv_theta * dinv[i] * b[i];
}

template <typename T>
__global__ void __launch_bounds__(256)
cheb_step_kernel(const int32_t* __restrict__ cols, const T* __restrict__ vals,
                 const T* __restrict__ dinv, const T* __restrict__ d_in,
                 T* __restrict__ d_out, T* __restrict__ r, T* __restrict__ x,
                 T c1, T c2, int64_t n, int width) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  T ad = T(0);
  for (int k = 0; k < width; ++k)
    ad += vals[k * n + i] * d_in[cols[k * n + i]];
  const T di = d_in[i];
  const T rn = r[i] - ad;
  r[i] = rn;
  x[i] += di;
  d_out[i] = c1 * di + c2 * dinv[i] * rn;
}

}  // namespace sfem

using namespace sfem;

extern "C" int sfem_ell_chebyshev(const int32_t* cols, const void* vals,
                                  const void* dinv, const void* b, void* x,
                                  void* work, int64_t n, int width, int steps,
                                  double lmin, double lmax, int dtype,
                                  sfem_stream_t stream) {
  SFEM_REQUIRE(n >= 0 && width >= 1 && steps >= 1 && lmin > 0 && lmax > lmin,
               "sfem_ell_chebyshev: bad sizes or spectrum bounds");
  if (n == 0) return SFEM_OK;
  SFEM_REQUIRE(cols && vals && dinv && b && x && work,
               "sfem_ell_chebyshev: null pointer");
  SFEM_REQUIRE(dtype == SFEM_F32 || dtype == SFEM_F64,
               "sfem_ell_chebyshev: unknown dtype %d", dtype);
  const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin);
  const double sigma = theta / delta;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  hipStream_t st = as_stream(stream);
#define SFEM_CHEB(T)                                                          \
  do {                                                                        \
    T* r = (T*)work;                                                          \
    T* d0 = r + n;                                                            \
    T* d1 = d0 + n;                                                           \
    hipLaunchKernelGGL(cheb_init_kernel<T>, grid, block, 0, st, (const T*)b,  \
                       (const T*)dinv, (T*)x, r, d0, (T)(1.0 / theta), n);    \
    double rho = 1.0 / sigma;                                                 \
    for (int k = 0; k < steps; ++k) {                                         \
      const double rho_new = 1.0 / (2.0 * sigma - rho);                       \
      hipLaunchKernelGGL(cheb_step_kernel<T>, grid, block, 0, st, cols,       \
                         (const T*)vals, (const T*)dinv, (k & 1) ? d1 : d0,   \
                         (k & 1) ? d0 : d1, r, (T*)x, (T)(rho_new * rho),     \
                         (T)(2.0 * rho_new / delta), n, width);               \
      rho = rho_new;                                                          \
    }                                                                         \
  } while (0)
  if (dtype == SFEM_F64) SFEM_CHEB(double);
  else SFEM_CHEB(float);
#undef SFEM_CHEB
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

static int fdm_solve_impl(const void* r, void* z, const int64_t* pel,
                          const void* S, const int32_t* cases,
                          const void* inv_eigenvalues, const void* weights,
                          void* elem_sum, void* weighted_sum,
                          int64_t num_elements, int ndim, int Pp, int dtype,
                          sfem_stream_t stream);

extern "C" int sfem_fdm_solve(const void* r, void* z, const int64_t* pel,
                              const void* S, const int32_t* cases,
                              const void* inv_eigenvalues,
                              int64_t num_elements, int ndim, int Pp,
                              int dtype, sfem_stream_t stream) {
  return fdm_solve_impl(r, z, pel, S, cases, inv_eigenvalues, nullptr, nullptr,
                        nullptr, num_elements, ndim, Pp, dtype, stream);
}

extern "C" int sfem_fdm_solve_sums(const void* r, void* z, const int64_t* pel,
                                   const void* S, const int32_t* cases,
                                   const void* inv_eigenvalues,
                                   const void* weights, void* elem_sum,
                                   void* weighted_sum, int64_t num_elements,
                                   int ndim, int Pp, int dtype,
                                   sfem_stream_t stream) {
  SFEM_REQUIRE(num_elements == 0 || (elem_sum && (!weighted_sum || weights)),
               "sfem_fdm_solve_sums: elem_sum (and weights with "
               "weighted_sum) required");
  return fdm_solve_impl(r, z, pel, S, cases, inv_eigenvalues, weights,
                        elem_sum, weighted_sum, num_elements, ndim, Pp, dtype,
                        stream);
}

extern "C" int sfem_add_element_constants(void* z, const void* yc,
                                          const void* shift,
                                          int64_t num_elements, int n,
                                          int64_t elems_per_member, int dtype,
                                          sfem_stream_t stream) {
  SFEM_REQUIRE(num_elements >= 0 && n >= 1 && elems_per_member >= 1,
               "sfem_add_element_constants: bad sizes");
  SFEM_REQUIRE(dtype == SFEM_F32 || dtype == SFEM_F64,
               "sfem_add_element_constants: unknown dtype %d", dtype);
  if (num_elements == 0) return SFEM_OK;
  SFEM_REQUIRE(z && yc && shift, "sfem_add_element_constants: null pointer");
  const int64_t total = num_elements * n;
  int64_t blocks = (total + 1023) / 1024;
  if (blocks > 8192) blocks = 8192;
  if (dtype == SFEM_F64)
    hipLaunchKernelGGL(add_element_constants_kernel<double>,
                       dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                       (double*)z, (const double*)yc, (const double*)shift,
                       total, n, elems_per_member);
  else
    hipLaunchKernelGGL(add_element_constants_kernel<float>,
                       dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                       (float*)z, (const float*)yc, (const float*)shift, total,
                       n, elems_per_member);
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

static int fdm_solve_impl(const void* r, void* z, const int64_t* pel,
                          const void* S, const int32_t* cases,
                          const void* inv_eigenvalues, const void* weights,
                          void* elem_sum, void* weighted_sum,
                          int64_t num_elements, int ndim, int Pp, int dtype,
                          sfem_stream_t stream) {
  SFEM_REQUIRE(num_elements >= 0 && ndim >= 1 && ndim <= 3 && Pp >= 1 &&
                   Pp <= FDM_MAX_P,
               "sfem_fdm_solve: ndim 1..3, 1 <= Pp <= %d", FDM_MAX_P);
  if (num_elements == 0) return SFEM_OK;
  SFEM_REQUIRE(r && z && S && cases && inv_eigenvalues,
               "sfem_fdm_solve: null pointer");
  SFEM_REQUIRE(num_elements <= 0x7fffffff, "sfem_fdm_solve: too many elements");
  const dim3 grid((unsigned)num_elements);
  hipStream_t st = as_stream(stream);
  int rc;
#define SFEM_FDM_DIM(T, DV)                                                   \
  launch_fdm<T, DV>(Pp, grid, st, (const T*)r, (T*)z, pel, (const T*)S,       \
                    cases, (const T*)inv_eigenvalues, num_elements,           \
                    (const T*)weights, (T*)elem_sum, (T*)weighted_sum)
  if (dtype == SFEM_F64)
    rc = ndim == 3 ? SFEM_FDM_DIM(double, 3)
                   : (ndim == 2 ? SFEM_FDM_DIM(double, 2)
                                : SFEM_FDM_DIM(double, 1));
  else if (dtype == SFEM_F32)
    rc = ndim == 3 ? SFEM_FDM_DIM(float, 3)
                   : (ndim == 2 ? SFEM_FDM_DIM(float, 2)
                                : SFEM_FDM_DIM(float, 1));
  else {
    set_error("sfem_fdm_solve: unknown dtype %d", dtype);
    return SFEM_EINVAL;
  }
#undef SFEM_FDM_DIM
  if (rc != SFEM_OK) return rc;
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}
