// libsfem_hip: generic sum-factorised basis evaluation, its transpose and the
// geometric factors, for any ndim in {1,2,3}, P,q <= SFEM_MAX_P, ncomp <= 8.
//
// These kernels serve the general `local_covector` path (arbitrary pointwise
// forms, non-collocated quadrature, setup).  One workgroup owns one element;
// the element's tensor is staged in LDS and contracted with the 1D matrices
// axis by axis (O(d P^(d+1)) instead of the reference's dense O(P^2d) einsum,
// core/interpolation.py:260-263).  The tuned collocated Helmholtz/mass/
// stiffness operator lives in sfem_helmholtz.hip.
#include "sfem_common.h"
#include "sfem_interp.h"
#include <cstdlib>

namespace sfem {

struct Shape3 {
  int s[3];
  __host__ __device__ int size() const { return s[0] * s[1] * s[2]; }
};

// out = contract `in` along `axis` with mat (rows x cols, row-major; TRANS uses
// mat^T).  Both tensors live in LDS, C order, shapes padded to 3 axes.
template <typename T, bool TRANS>
__device__ inline void contract_axis(const T* __restrict__ in, T* __restrict__ out,
                                     Shape3 in_shape, int axis,
                                     const T* __restrict__ mat, int rows,
                                     int cols) {
  // non-transposed: out dim = rows, in dim = cols; transposed: the reverse
  const int out_n = TRANS ? cols : rows;
  const int in_n = TRANS ? rows : cols;
  Shape3 os = in_shape;
  os.s[axis] = out_n;
  const int stride = axis == 0 ? in_shape.s[1] * in_shape.s[2]
                               : (axis == 1 ? in_shape.s[2] : 1);
  const int total = os.size();
  for (int t = threadIdx.x; t < total; t += blockDim.x) {
    int i2 = t % os.s[2];
    int i1 = (t / os.s[2]) % os.s[1];
    int i0 = t / (os.s[2] * os.s[1]);
    int idx[3] = {i0, i1, i2};
    const int o = idx[axis];
    idx[axis] = 0;
    const int base = (idx[0] * in_shape.s[1] + idx[1]) * in_shape.s[2] + idx[2];
    T acc = T(0);
    for (int m = 0; m < in_n; ++m) {
      const T w = TRANS ? mat[m * cols + o] : mat[o * cols + m];
      acc += w * in[base + m * stride];
    }
    out[t] = acc;
  }
  __syncthreads();
}

template <typename T>
struct BasisArgs {
  const T* u_local;   // (E, n, nc)
  const T* interp1;   // (q, P)
  const T* grad1;     // (q, P)
  const T* invjac;    // (E, Q, d, d) or null
  T* val;             // (E, Q, nc) or null
  T* grad;            // (E, Q, d, nc) or null
  int64_t num_elements;
  int ndim, P, q, ncomp, collocated;
};

template <typename T>
__global__ void __launch_bounds__(256)
basis_eval_kernel(BasisArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  const int d = a.ndim, P = a.P, q = a.q, nc = a.ncomp;
  const int m = P > q ? P : q;
  int cap = 1, n = 1, Q = 1;
  for (int i = 0; i < d; ++i) { cap *= m; n *= P; Q *= q; }
  T* s_i = smem;                 // (q, P)
  T* s_g = s_i + q * P;          // (q, P)
  T* buf0 = s_g + q * P;
  T* buf1 = buf0 + cap;
  T* src = buf1 + cap;           // element nodal values of one component
  for (int t = threadIdx.x; t < q * P; t += blockDim.x) {
    s_i[t] = a.interp1[t];
    s_g[t] = a.grad1[t];
  }
  for (int64_t e = blockIdx.x; e < a.num_elements; e += gridDim.x) {
    for (int k = 0; k < nc; ++k) {
      __syncthreads();
      for (int t = threadIdx.x; t < n; t += blockDim.x)
        src[t] = a.u_local[(e * n + t) * nc + k];
      __syncthreads();
      // pass -1: values (I on every axis); pass i>=0: d/dxi_i (G on axis i)
      for (int pass = (a.val ? -1 : 0); pass < (a.grad ? d : 0); ++pass) {
        if (pass < 0 && a.collocated) {
          for (int t = threadIdx.x; t < n; t += blockDim.x)
            a.val[(e * Q + t) * nc + k] = src[t];
          continue;
        }
        Shape3 sh = {{1, 1, 1}};
        for (int i = 0; i < d; ++i) sh.s[3 - d + i] = P;
        const T* in = src;
        T* out = buf0;
        for (int ax = 0; ax < d; ++ax) {
          const int axis = 3 - d + ax;
          contract_axis<T, false>(in, out, sh, axis, ax == pass ? s_g : s_i, q,
                                  P);
          sh.s[axis] = q;
          in = out;
          out = (out == buf0) ? buf1 : buf0;
        }
        if (pass < 0) {
          for (int t = threadIdx.x; t < Q; t += blockDim.x)
            a.val[(e * Q + t) * nc + k] = in[t];
        } else {
          for (int t = threadIdx.x; t < Q; t += blockDim.x)
            a.grad[((e * Q + t) * d + pass) * nc + k] = in[t];
        }
      }
    }
    if (a.grad && a.invjac) {
      // reference-space -> physical gradient, in place:
      //   g[j][k] = sum_i invjac[j][i] * ref[i][k]   (fespace.py:193, :224)
      __syncthreads();
      for (int t = threadIdx.x; t < Q * nc; t += blockDim.x) {
        const int pt = t / nc, k = t - pt * nc;
        const T* ij = a.invjac + (e * Q + pt) * d * d;
        T* g = a.grad + (e * Q + pt) * d * nc + k;
        T ref[3], phys[3];
        for (int i = 0; i < d; ++i) ref[i] = g[i * nc];
        for (int j = 0; j < d; ++j) {
          T acc = T(0);
          for (int i = 0; i < d; ++i) acc += ij[j * d + i] * ref[i];
          phys[j] = acc;
        }
        for (int j = 0; j < d; ++j) g[j * nc] = phys[j];
      }
    }
  }
}

template <typename T>
struct BasisTArgs {
  const T* c0;        // (E, Q, nc) or null
  const T* c1;        // (E, Q, d, nc) or null
  const T* interp1;
  const T* grad1;
  const T* invjac;    // (E, Q, d, d), required with c1
  const T* wdet;      // (E, Q)
  T* out;             // (E, n, nc)
  int64_t num_elements;
  int ndim, P, q, ncomp, collocated;
};

template <typename T>
__global__ void __launch_bounds__(256)
basis_eval_t_kernel(BasisTArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  const int d = a.ndim, P = a.P, q = a.q, nc = a.ncomp;
  const int m = P > q ? P : q;
  int cap = 1, n = 1, Q = 1;
  for (int i = 0; i < d; ++i) { cap *= m; n *= P; Q *= q; }
  T* s_i = smem;
  T* s_g = s_i + q * P;
  T* buf0 = s_g + q * P;
  T* buf1 = buf0 + cap;
  T* acc = buf1 + cap;           // (n,) accumulated covector of one component
  for (int t = threadIdx.x; t < q * P; t += blockDim.x) {
    s_i[t] = a.interp1[t];
    s_g[t] = a.grad1[t];
  }
  for (int64_t e = blockIdx.x; e < a.num_elements; e += gridDim.x) {
    for (int k = 0; k < nc; ++k) {
      __syncthreads();
      for (int t = threadIdx.x; t < n; t += blockDim.x) acc[t] = T(0);
      for (int pass = (a.c0 ? -1 : 0); pass < (a.c1 ? d : 0); ++pass) {
        __syncthreads();
        // quadrature-point field feeding this pass
        for (int t = threadIdx.x; t < Q; t += blockDim.x) {
          const T w = a.wdet[e * Q + t];
          T v;
          if (pass < 0) {
            v = a.c0[(e * Q + t) * nc + k];
          } else {
            // ref[i] = sum_j invjac[j][i] * c1[j]   (transpose of the above)
            const T* ij = a.invjac + (e * Q + t) * d * d;
            const T* c = a.c1 + (e * Q + t) * d * nc + k;
            v = T(0);
            for (int j = 0; j < d; ++j) v += ij[j * d + pass] * c[j * nc];
          }
          buf0[t] = w * v;
        }
        __syncthreads();
        if (pass < 0 && a.collocated) {
          for (int t = threadIdx.x; t < n; t += blockDim.x) acc[t] += buf0[t];
          continue;
        }
        Shape3 sh = {{1, 1, 1}};
        for (int i = 0; i < d; ++i) sh.s[3 - d + i] = q;
        const T* in = buf0;
        T* out = buf1;
        for (int ax = 0; ax < d; ++ax) {
          const int axis = 3 - d + ax;
          contract_axis<T, true>(in, out, sh, axis, ax == pass ? s_g : s_i, q,
                                 P);
          sh.s[axis] = P;
          in = out;
          out = (out == buf0) ? buf1 : buf0;
        }
        for (int t = threadIdx.x; t < n; t += blockDim.x) acc[t] += in[t];
      }
      __syncthreads();
      for (int t = threadIdx.x; t < n; t += blockDim.x)
        a.out[(e * n + t) * nc + k] = acc[t];
    }
  }
}

// jac (E*Q, d, d) -> inverse in place, determinant to jacdet.
template <typename T>
__global__ void __launch_bounds__(256)
invert_jac_kernel(T* __restrict__ jac, T* __restrict__ jacdet, int64_t count,
                  int d) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += stride) {
    T* a = jac + i * d * d;
    if (d == 1) {
      const T det = a[0];
      a[0] = T(1) / det;
      jacdet[i] = det;
    } else if (d == 2) {
      const T a00 = a[0], a01 = a[1], a10 = a[2], a11 = a[3];
      const T det = a00 * a11 - a01 * a10;
      const T r = T(1) / det;
      a[0] = a11 * r; a[1] = -a01 * r; a[2] = -a10 * r; a[3] = a00 * r;
      jacdet[i] = det;
    } else {
      const T a00 = a[0], a01 = a[1], a02 = a[2], a10 = a[3], a11 = a[4],
              a12 = a[5], a20 = a[6], a21 = a[7], a22 = a[8];
      const T c00 = a11 * a22 - a12 * a21;
      const T c01 = a12 * a20 - a10 * a22;
      const T c02 = a10 * a21 - a11 * a20;
      const T det = a00 * c00 + a01 * c01 + a02 * c02;
      const T r = T(1) / det;
      a[0] = c00 * r; a[1] = (a02 * a21 - a01 * a22) * r;
      a[2] = (a01 * a12 - a02 * a11) * r;
      a[3] = c01 * r; a[4] = (a00 * a22 - a02 * a20) * r;
      a[5] = (a02 * a10 - a00 * a12) * r;
      a[6] = c02 * r; a[7] = (a01 * a20 - a00 * a21) * r;
      a[8] = (a00 * a11 - a01 * a10) * r;
      jacdet[i] = det;
    }
  }
}

template <typename T>
int launch_basis_eval(const BasisArgs<T>& a, hipStream_t stream) {
  const int m = a.P > a.q ? a.P : a.q;
  size_t cap = 1, n = 1;
  for (int i = 0; i < a.ndim; ++i) { cap *= m; n *= a.P; }
  const size_t lds = sizeof(T) * (2 * (size_t)a.q * a.P + 2 * cap + n);
  if (lds > 160 * 1024) {
    set_error("basis kernel needs %zu bytes of LDS (> 160 KiB)", lds);
    return SFEM_EUNSUPPORTED;
  }
  auto kern = basis_eval_kernel<T>;
  if (lds > 64 * 1024)
    SFEM_HIP(hipFuncSetAttribute((const void*)kern,
                                 hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds));
  const int64_t grid = a.num_elements < 256 * 8 ? a.num_elements : 256 * 8;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, stream, a);
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

template <typename T>
int launch_basis_eval_t(const BasisTArgs<T>& a, hipStream_t stream) {
  const int m = a.P > a.q ? a.P : a.q;
  size_t cap = 1, n = 1;
  for (int i = 0; i < a.ndim; ++i) { cap *= m; n *= a.P; }
  const size_t lds = sizeof(T) * (2 * (size_t)a.q * a.P + 2 * cap + n);
  if (lds > 160 * 1024) {
    set_error("basis kernel needs %zu bytes of LDS (> 160 KiB)", lds);
    return SFEM_EUNSUPPORTED;
  }
  auto kern = basis_eval_t_kernel<T>;
  if (lds > 64 * 1024)
    SFEM_HIP(hipFuncSetAttribute((const void*)kern,
                                 hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds));
  const int64_t grid = a.num_elements < 256 * 8 ? a.num_elements : 256 * 8;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, stream, a);
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

// SFEM_INTERP=0: the values-only cases run the generic kernels too (A/B)
static bool interp_enabled() {
  const char* v = std::getenv("SFEM_INTERP");
  return !(v && v[0] == '0');
}

static int check_shape(const char* who, int64_t E, int ndim, int P, int q,
                       int ncomp) {
  SFEM_REQUIRE(E >= 0, "%s: negative element count", who);
  SFEM_REQUIRE(ndim >= 1 && ndim <= 3, "%s: ndim=%d outside 1..3", who, ndim);
  SFEM_REQUIRE(P >= 1 && P <= SFEM_MAX_P && q >= 1 && q <= SFEM_MAX_P,
               "%s: P=%d q=%d outside 1..%d", who, P, q, SFEM_MAX_P);
  SFEM_REQUIRE(ncomp >= 1 && ncomp <= 8, "%s: ncomp=%d outside 1..8", who,
               ncomp);
  return SFEM_OK;
}

}  // namespace sfem

using namespace sfem;

extern "C" {

int sfem_basis_eval(const void* u_local, const void* interp1,
                    const void* grad1, const void* invjac, void* val,
                    void* grad, int64_t num_elements, int ndim, int P, int q,
                    int ncomp, int collocated, int dtype,
                    sfem_stream_t stream) {
  int rc = check_shape("sfem_basis_eval", num_elements, ndim, P, q, ncomp);
  if (rc) return rc;
  SFEM_REQUIRE(!collocated || P == q, "sfem_basis_eval: collocated needs P==q");
  if (num_elements == 0 || (!val && !grad)) return SFEM_OK;
  SFEM_REQUIRE(u_local && interp1 && grad1, "sfem_basis_eval: null pointer");
  if (val && !grad && !collocated && interp_enabled()) {
    // values only: the compile-time-sized interpolation (sfem_interp.h), same
    // bits as the generic kernel below
    const int rc =
        dtype == SFEM_F64
            ? launch_tensor_interp_f64(ndim, P, q, u_local, interp1, nullptr,
                                       val, num_elements, ncomp, false,
                                       as_stream(stream))
            : (dtype == SFEM_F32
                   ? launch_tensor_interp_f32(ndim, P, q, u_local, interp1,
                                              nullptr, val, num_elements, ncomp,
                                              false, as_stream(stream))
                   : SFEM_EUNSUPPORTED);
    if (rc == SFEM_OK) {
      SFEM_LAUNCH_CHECK();
      return SFEM_OK;
    }
  }
  if (dtype == SFEM_F64) {
    BasisArgs<double> a{(const double*)u_local, (const double*)interp1,
                        (const double*)grad1, (const double*)invjac,
                        (double*)val, (double*)grad, num_elements, ndim, P, q,
                        ncomp, collocated};
    return launch_basis_eval(a, as_stream(stream));
  } else if (dtype == SFEM_F32) {
    BasisArgs<float> a{(const float*)u_local, (const float*)interp1,
                       (const float*)grad1, (const float*)invjac, (float*)val,
                       (float*)grad, num_elements, ndim, P, q, ncomp,
                       collocated};
    return launch_basis_eval(a, as_stream(stream));
  }
  set_error("sfem_basis_eval: unknown dtype %d", dtype);
  return SFEM_EINVAL;
}

int sfem_basis_eval_t(const void* c0, const void* c1, const void* interp1,
                      const void* grad1, const void* invjac, const void* wdet,
                      void* out, int64_t num_elements, int ndim, int P, int q,
                      int ncomp, int collocated, int dtype,
                      sfem_stream_t stream) {
  int rc = check_shape("sfem_basis_eval_t", num_elements, ndim, P, q, ncomp);
  if (rc) return rc;
  SFEM_REQUIRE(!collocated || P == q,
               "sfem_basis_eval_t: collocated needs P==q");
  if (num_elements == 0) return SFEM_OK;
  SFEM_REQUIRE(interp1 && grad1 && wdet && out,
               "sfem_basis_eval_t: null pointer");
  SFEM_REQUIRE(!c1 || invjac, "sfem_basis_eval_t: c1 needs invjac");
  if (c0 && !c1 && !collocated && interp_enabled()) {
    // out = (I^T (x) .. (x) I^T)(wdet .* c0), compile-time sizes
    const int rc =
        dtype == SFEM_F64
            ? launch_tensor_interp_f64(ndim, q, P, c0, interp1, wdet, out,
                                       num_elements, ncomp, true,
                                       as_stream(stream))
            : (dtype == SFEM_F32
                   ? launch_tensor_interp_f32(ndim, q, P, c0, interp1, wdet, out,
                                              num_elements, ncomp, true,
                                              as_stream(stream))
                   : SFEM_EUNSUPPORTED);
    if (rc == SFEM_OK) {
      SFEM_LAUNCH_CHECK();
      return SFEM_OK;
    }
  }
  if (dtype == SFEM_F64) {
    BasisTArgs<double> a{(const double*)c0, (const double*)c1,
                         (const double*)interp1, (const double*)grad1,
                         (const double*)invjac, (const double*)wdet,
                         (double*)out, num_elements, ndim, P, q, ncomp,
                         collocated};
    return launch_basis_eval_t(a, as_stream(stream));
  } else if (dtype == SFEM_F32) {
    BasisTArgs<float> a{(const float*)c0, (const float*)c1,
                        (const float*)interp1, (const float*)grad1,
                        (const float*)invjac, (const float*)wdet, (float*)out,
                        num_elements, ndim, P, q, ncomp, collocated};
    return launch_basis_eval_t(a, as_stream(stream));
  }
  set_error("sfem_basis_eval_t: unknown dtype %d", dtype);
  return SFEM_EINVAL;
}

int sfem_geom_factors(const void* elem_coords, const void* interp1,
                      const void* grad1, int64_t num_elements, int ndim, int P,
                      int q, void* invjac, void* jacdet, void* quad_coords,
                      int dtype, sfem_stream_t stream) {
  int rc = check_shape("sfem_geom_factors", num_elements, ndim, P, q, ndim);
  if (rc) return rc;
  if (num_elements == 0) return SFEM_OK;
  SFEM_REQUIRE(elem_coords && invjac && jacdet,
               "sfem_geom_factors: null pointer");
  // jac[i][j] = d x_j / d xi_i is the reference gradient of the coordinate
  // field (ncomp = ndim), written straight into the invjac buffer ...
  rc = sfem_basis_eval(elem_coords, interp1, grad1, nullptr, quad_coords,
                       invjac, num_elements, ndim, P, q, ndim, 0, dtype,
                       stream);
  if (rc) return rc;
  // ... and inverted in place.
  int64_t Q = 1;
  for (int i = 0; i < ndim; ++i) Q *= q;
  const int64_t count = num_elements * Q;
  if (dtype == SFEM_F64)
    hipLaunchKernelGGL(invert_jac_kernel<double>, dim3(stream_grid(count, 256)),
                       dim3(256), 0, as_stream(stream), (double*)invjac,
                       (double*)jacdet, count, ndim);
  else
    hipLaunchKernelGGL(invert_jac_kernel<float>, dim3(stream_grid(count, 256)),
                       dim3(256), 0, as_stream(stream), (float*)invjac,
                       (float*)jacdet, count, ndim);
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

}  // extern "C"
