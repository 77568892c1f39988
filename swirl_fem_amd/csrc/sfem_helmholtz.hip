// libsfem_hip: C-ABI entry points of the fused Helmholtz operator plus its
// setup kernels (symmetric geometric factors, encoded indices).
#include <stdlib.h>

#include "sfem_helmholtz.h"

namespace sfem {

// Per element the factors w detJ (J^-1 J^-T)[ik] (ref-direction indices) and
// w detJ are stored in the paired layout read by helmholtz_kernel:
//   3D: [3][Q][2] = (G00,G01) (G02,G11) (G12,G22), then W [Q]     (7 Q reals)
//   2D: [2][Q][2] = (G00,G01) (G11,W)                             (4 Q reals)
//   1D: [Q][2]    = (G00,W)
template <typename T>
__global__ void __launch_bounds__(256)
helmholtz_setup_kernel(const T* __restrict__ invjac,
                       const T* __restrict__ jacdet,
                       const T* __restrict__ weights, T* __restrict__ geo,
                       int64_t num_elements, int d, int Q) {
  const int ng = d * (d + 1) / 2;
  const int64_t total = num_elements * Q;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += stride) {
    const int64_t e = t / Q;
    const int q = (int)(t - e * Q);
    const T* ij = invjac + t * d * d;
    const T wd = weights[q] * jacdet[t];
    T vals[7];
    int f = 0;
    for (int i = 0; i < d; ++i)
      for (int k = i; k < d; ++k) {
        T acc = T(0);
        for (int j = 0; j < d; ++j) acc += ij[j * d + i] * ij[j * d + k];
        vals[f++] = wd * acc;
      }
    vals[ng] = wd;
    T* g = geo + e * (int64_t)(ng + 1) * Q;
    const int npair = d == 3 ? 3 : (ng + 1) / 2;
    for (int pi = 0; pi < npair; ++pi) {
      g[((int64_t)pi * Q + q) * 2 + 0] = vals[2 * pi];
      g[((int64_t)pi * Q + q) * 2 + 1] = vals[2 * pi + 1];
    }
    if (d == 3) g[(int64_t)6 * Q + q] = wd;
  }
}

__global__ void __launch_bounds__(256)
encode_kernel(const int32_t* __restrict__ elements,
              const uint8_t* __restrict__ dirichlet,
              const int32_t* __restrict__ multiplicity,
              int32_t* __restrict__ enc, int64_t count) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += stride) {
    const int32_t k = elements[i];
    uint32_t v;
    if (k < 0) {
      v = SFEM_IDX_PAD;
    } else {
      v = (uint32_t)k;
      if (dirichlet && dirichlet[k]) v |= SFEM_IDX_DIRICHLET;
      if (multiplicity[k] > 1) v |= SFEM_IDX_SHARED;
    }
    enc[i] = (int32_t)v;
  }
}

// geo_elem[e] = { detJ (J^-1 J^-T) upper triangle (ref-direction indices),
// padded to 6 entries, detJ, 0 } taken at the element's first quadrature point.
template <typename T>
__global__ void __launch_bounds__(256)
helmholtz_setup_affine_kernel(const T* __restrict__ invjac,
                              const T* __restrict__ jacdet,
                              T* __restrict__ geo_elem, int64_t num_elements,
                              int d, int Q) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= num_elements) return;
  const T* ij = invjac + e * Q * d * d;
  const T det = jacdet[e * Q];
  T* g = geo_elem + e * 8;
  int f = 0;
  for (int i = 0; i < d; ++i)
    for (int k = i; k < d; ++k) {
      T acc = T(0);
      for (int j = 0; j < d; ++j) acc += ij[j * d + i] * ij[j * d + k];
      g[f++] = det * acc;
    }
  for (; f < 6; ++f) g[f] = T(0);
  g[6] = det;
  g[7] = T(0);
}

template <typename T>
static int run_helmholtz(const void* u, void* out, const int32_t* enc,
                         const void* geo, const void* geo_elem,
                         const int32_t* geo_index, const void* dmat,
                         const void* weights, int64_t E, int ndim, int P,
                         int ncomp, double l0, double l1, bool gs,
                         hipStream_t stream) {
  const char* dbg = getenv("SFEM_DEBUG_FLAGS");
  HelmholtzParams<T> prm{(const T*)u,        (T*)out,         enc,
                         (const T*)geo,      (const T*)geo_elem, geo_index,
                         (const T*)dmat,     (const T*)weights, E,
                         ncomp,              (T)l0,           (T)l1,
                         dbg ? atoi(dbg) : 0};
  if (ndim == 3) return dispatch_helmholtz<T, 3>(prm, P, gs, stream);
  if (ndim == 2) return dispatch_helmholtz<T, 2>(prm, P, gs, stream);
  set_error("helmholtz: ndim=%d (fused kernel supports 2 and 3)", ndim);
  return SFEM_EUNSUPPORTED;
}

}  // namespace sfem

using namespace sfem;

extern "C" {

int sfem_helmholtz_setup(const void* invjac, const void* jacdet,
                         const void* weights_nd, void* geo,
                         int64_t num_elements, int ndim, int Q, int dtype,
                         sfem_stream_t stream) {
  SFEM_REQUIRE(num_elements >= 0 && ndim >= 1 && ndim <= 3 && Q >= 1,
               "sfem_helmholtz_setup: bad sizes");
  if (num_elements == 0) return SFEM_OK;
  SFEM_REQUIRE(invjac && jacdet && weights_nd && geo,
               "sfem_helmholtz_setup: null pointer");
  const unsigned grid = stream_grid(num_elements * Q, 256);
  if (dtype == SFEM_F64)
    hipLaunchKernelGGL(helmholtz_setup_kernel<double>, dim3(grid), dim3(256), 0,
                       as_stream(stream), (const double*)invjac,
                       (const double*)jacdet, (const double*)weights_nd,
                       (double*)geo, num_elements, ndim, Q);
  else if (dtype == SFEM_F32)
    hipLaunchKernelGGL(helmholtz_setup_kernel<float>, dim3(grid), dim3(256), 0,
                       as_stream(stream), (const float*)invjac,
                       (const float*)jacdet, (const float*)weights_nd,
                       (float*)geo, num_elements, ndim, Q);
  else {
    set_error("sfem_helmholtz_setup: unknown dtype %d", dtype);
    return SFEM_EINVAL;
  }
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_encode_elements(const int32_t* elements, const uint8_t* dirichlet,
                         const int32_t* multiplicity, int32_t* enc,
                         int64_t count, sfem_stream_t stream) {
  SFEM_REQUIRE(count >= 0, "sfem_encode_elements: negative count");
  if (count == 0) return SFEM_OK;
  SFEM_REQUIRE(elements && multiplicity && enc,
               "sfem_encode_elements: null pointer");
  hipLaunchKernelGGL(encode_kernel, dim3(stream_grid(count, 256)), dim3(256), 0,
                     as_stream(stream), elements, dirichlet, multiplicity, enc,
                     count);
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_helmholtz_apply(const sfem_helmholtz_args* a, sfem_stream_t stream) {
  SFEM_REQUIRE(a, "sfem_helmholtz_apply: null args");
  SFEM_REQUIRE(a->num_elements >= 0 && a->num_nodes >= 0 && a->ncomp >= 1 &&
                   a->ncomp <= 8,
               "sfem_helmholtz_apply: bad sizes");
  SFEM_REQUIRE(a->dtype == SFEM_F32 || a->dtype == SFEM_F64,
               "sfem_helmholtz_apply: unknown dtype %d", a->dtype);
  SFEM_REQUIRE(a->num_nodes <= SFEM_IDX_MASK,
               "sfem_helmholtz_apply: more than 2^30-1 nodes per device");
  SFEM_REQUIRE(0 <= a->zero_begin && a->zero_begin <= a->zero_end &&
                   a->zero_end <= a->num_nodes,
               "sfem_helmholtz_apply: bad zero range");
  if (a->num_nodes == 0) return SFEM_OK;
  SFEM_REQUIRE(a->out, "sfem_helmholtz_apply: null out");
  const size_t esz = a->dtype == SFEM_F64 ? 8 : 4;
  if (a->zero_end > a->zero_begin)
    SFEM_HIP(hipMemsetAsync(
        (char*)a->out + (size_t)a->zero_begin * a->ncomp * esz, 0,
        (size_t)(a->zero_end - a->zero_begin) * a->ncomp * esz,
        as_stream(stream)));
  if (a->num_elements == 0) return SFEM_OK;
  SFEM_REQUIRE(a->u && a->enc && a->dmat && (a->geo || a->geo_elem),
               "sfem_helmholtz_apply: null pointer");
  SFEM_REQUIRE(!a->geo_elem || a->weights,
               "sfem_helmholtz_apply: geo_elem needs the quadrature weights");
  SFEM_REQUIRE(!(a->geo_elem && a->geo) || a->geo_index,
               "sfem_helmholtz_apply: mixed geometry needs geo_index");
  if (a->dtype == SFEM_F64)
    return run_helmholtz<double>(a->u, a->out, a->enc, a->geo, a->geo_elem,
                                 a->geo_index, a->dmat, a->weights,
                                 a->num_elements, a->ndim, a->P, a->ncomp,
                                 a->lambda0, a->lambda1, true,
                                 as_stream(stream));
  return run_helmholtz<float>(a->u, a->out, a->enc, a->geo, a->geo_elem,
                              a->geo_index, a->dmat, a->weights,
                              a->num_elements, a->ndim, a->P, a->ncomp,
                              a->lambda0, a->lambda1, true, as_stream(stream));
}

int sfem_helmholtz_setup_affine(const void* invjac, const void* jacdet,
                                void* geo_elem, int64_t num_elements, int ndim,
                                int Q, int dtype, sfem_stream_t stream) {
  SFEM_REQUIRE(num_elements >= 0 && ndim >= 1 && ndim <= 3 && Q >= 1,
               "sfem_helmholtz_setup_affine: bad sizes");
  if (num_elements == 0) return SFEM_OK;
  SFEM_REQUIRE(invjac && jacdet && geo_elem,
               "sfem_helmholtz_setup_affine: null pointer");
  const unsigned grid = (unsigned)((num_elements + 255) / 256);
  if (dtype == SFEM_F64)
    hipLaunchKernelGGL(helmholtz_setup_affine_kernel<double>, dim3(grid),
                       dim3(256), 0, as_stream(stream), (const double*)invjac,
                       (const double*)jacdet, (double*)geo_elem, num_elements,
                       ndim, Q);
  else if (dtype == SFEM_F32)
    hipLaunchKernelGGL(helmholtz_setup_affine_kernel<float>, dim3(grid),
                       dim3(256), 0, as_stream(stream), (const float*)invjac,
                       (const float*)jacdet, (float*)geo_elem, num_elements,
                       ndim, Q);
  else {
    set_error("sfem_helmholtz_setup_affine: unknown dtype %d", dtype);
    return SFEM_EINVAL;
  }
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_helmholtz_local(const void* u_local, void* out_local, const void* geo,
                         const void* geo_elem, const int32_t* geo_index,
                         const void* dmat, const void* weights,
                         int64_t num_elements, int ndim, int P, int ncomp,
                         double lambda0, double lambda1, int dtype,
                         sfem_stream_t stream) {
  SFEM_REQUIRE(num_elements >= 0 && ncomp >= 1 && ncomp <= 8,
               "sfem_helmholtz_local: bad sizes");
  if (num_elements == 0) return SFEM_OK;
  SFEM_REQUIRE(u_local && out_local && dmat && (geo || geo_elem),
               "sfem_helmholtz_local: null pointer");
  SFEM_REQUIRE(!geo_elem || weights,
               "sfem_helmholtz_local: geo_elem needs the quadrature weights");
  SFEM_REQUIRE(!(geo_elem && geo) || geo_index,
               "sfem_helmholtz_local: mixed geometry needs geo_index");
  if (dtype == SFEM_F64)
    return run_helmholtz<double>(u_local, out_local, nullptr, geo, geo_elem,
                                 geo_index, dmat, weights, num_elements, ndim,
                                 P, ncomp, lambda0, lambda1, false,
                                 as_stream(stream));
  if (dtype == SFEM_F32)
    return run_helmholtz<float>(u_local, out_local, nullptr, geo, geo_elem,
                                geo_index, dmat, weights, num_elements, ndim,
                                P, ncomp, lambda0, lambda1, false,
                                as_stream(stream));
  set_error("sfem_helmholtz_local: unknown dtype %d", dtype);
  return SFEM_EINVAL;
}

}  // extern "C"
