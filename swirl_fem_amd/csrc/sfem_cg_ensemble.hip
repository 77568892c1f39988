// Conjugate gradients for an ensemble: B independent recurrences in ONE set of
// launches.
//
// The reference trains on ensembles by `jax.vmap`-ing its solver step
// (niles/train.py:232, :262-264): B systems with the same operator, each with
// its own right-hand side, step lengths and iteration count.  Here the B
// members are B disjoint copies of the mesh seen as one mesh (`Mesh.replicate`:
// member m owns the contiguous range [m len, (m + 1) len) of every vector), so
// the operator kernels serve all members in a single launch as they are; what
// has to know about members is the recurrence -- inner products, alpha, beta
// and the stopping rule of linalg/cg.py:60-97 per member.  These kernels are
// that: blockIdx.y is the member, the scalars of member m are the
// SFEM_ENS_NSCALARS doubles at scalars + m SFEM_ENS_NSCALARS (same slots as
// the single solve: [0] gamma [1] p.Ap [3] alpha [4] beta [5] b.b [6] atol2
// [7] done [8] iterations [9] active in this iteration [10] status).  A member
// that has stopped is a no-op from then on (alpha = 0, nothing written), as a
// finished single solve is.
//
// Inner products are stored partial sums (SFEM_ENS_GROUPS per member and
// product, one per workgroup) added in index order by whoever needs the
// total: no atomics, nothing to clear.
//
// All of it is HBM- or launch-bound vector work; an ensemble pays where one
// member alone leaves the GPU waiting on launches (64 x 64 quads, order 8).
#include "sfem_common.h"

namespace sfem {

constexpr int ENS_NS = SFEM_ENS_NSCALARS;
constexpr int ENS_G = SFEM_ENS_GROUPS;

__device__ __forceinline__ bool ens_bad_gamma(double g) {
  return !(g >= 0.0) || !(g <= 1.7976931348623157e308);
}
__device__ __forceinline__ bool ens_bad_pap(double v) {
  return v == 0.0 || !(v >= -1.7976931348623157e308) ||
         !(v <= 1.7976931348623157e308);
}
// partial sums of member m, product `which`, in index order
__device__ __forceinline__ double ens_total(const double* __restrict__ partials,
                                            int m, int which) {
  const double* p = partials + ((int64_t)m * 2 + which) * ENS_G;
  double t = 0.0;
#pragma unroll
  for (int g = 0; g < ENS_G; ++g) t += p[g];
  return t;
}

// workgroups of the two summing kernels: 1024 threads (32 groups per member
// are few workgroups; they need their lanes to keep loads in flight)
constexpr int ENS_SUM_BLOCK = 1024;

__device__ __forceinline__ double ens_block_sum(double v) {
  __shared__ double partial[ENS_SUM_BLOCK / 64];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) partial[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int w = 0; w < ENS_SUM_BLOCK / 64; ++w) t += partial[w];
  return t;
}

// partials[m][which][g] = sum over chunk g of member m of a * b
template <typename T>
__global__ void __launch_bounds__(ENS_SUM_BLOCK)
ens_dot_kernel(const T* __restrict__ a, const T* __restrict__ b, int64_t len,
               double* __restrict__ partials, int which) {
  const int m = blockIdx.y, g = blockIdx.x;
  const int64_t chunk = (len + ENS_G - 1) / ENS_G;
  const int64_t lo = (int64_t)g * chunk;
  const int64_t hi = lo + chunk < len ? lo + chunk : len;
  const T* am = a + (int64_t)m * len;
  const T* bm = b + (int64_t)m * len;
  double v = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += ENS_SUM_BLOCK)
    v += (double)am[i] * (double)bm[i];
  const double total = ens_block_sum(v);
  if (threadIdx.x == 0)
    partials[((int64_t)m * 2 + which) * ENS_G + g] = total;
}

// after b.b (which = 0) and gamma_0 = r.z (which = 1): the stopping rule
__global__ void __launch_bounds__(256)
ens_init_kernel(double* __restrict__ scalars,
                const double* __restrict__ partials, int members,
                double maxiter, double tol, double atol) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= members) return;
  double* s = scalars + (int64_t)m * ENS_NS;
  const double bb = ens_total(partials, m, 0);
  const double gamma = ens_total(partials, m, 1);
  const double a = tol * tol * bb, b = atol * atol;
  for (int q = 0; q < ENS_NS; ++q) s[q] = 0.0;
  s[0] = gamma;
  s[5] = bb;
  s[6] = a > b ? a : b;
  s[10] = SFEM_CG_STATUS_RUNNING;
  if (ens_bad_gamma(gamma)) {
    s[10] = SFEM_CG_STATUS_BAD_GAMMA;
    s[7] = 1.0;
  } else if (!(gamma > s[6])) {
    s[10] = SFEM_CG_STATUS_CONVERGED;
    s[7] = 1.0;
  } else if (maxiter <= 0.0) {
    s[10] = SFEM_CG_STATUS_MAXITER;
    s[7] = 1.0;
  }
}

// r -= alpha Ap with alpha = gamma / p.Ap taken from the stored partial sums
// (a stopped member, or one whose p.Ap is unusable, keeps its r)
template <typename T>
__global__ void __launch_bounds__(256)
ens_update_r_kernel(T* __restrict__ r, const T* __restrict__ ap, int64_t len,
                    const double* __restrict__ scalars,
                    const double* __restrict__ partials) {
  const int m = blockIdx.y;
  const double* s = scalars + (int64_t)m * ENS_NS;
  if (s[7] != 0.0) return;
  const double pap = ens_total(partials, m, 0);
  if (ens_bad_pap(pap)) return;
  const T alpha = (T)(s[0] / pap);
  T* rm = r + (int64_t)m * len;
  const T* am = ap + (int64_t)m * len;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < len;
       i += stride)
    rm[i] -= alpha * am[i];
}

// end of the iteration, one thread per member: alpha again (for x), beta,
// gamma <- gamma_new, the counter and the stop test of cg.py:68-73
__global__ void __launch_bounds__(256)
ens_close_kernel(double* __restrict__ scalars,
                 const double* __restrict__ partials, int members,
                 double maxiter) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= members) return;
  double* s = scalars + (int64_t)m * ENS_NS;
  s[9] = 0.0;
  if (s[7] != 0.0) return;
  const double pap = ens_total(partials, m, 0);
  s[1] = pap;
  if (ens_bad_pap(pap)) {
    s[10] = SFEM_CG_STATUS_BAD_PAP;
    s[7] = 1.0;
    return;
  }
  const double g = ens_total(partials, m, 1);
  s[3] = s[0] / pap;
  s[4] = g / s[0];
  s[0] = g;
  s[8] += 1.0;
  s[9] = 1.0;
  if (ens_bad_gamma(g)) {
    s[10] = SFEM_CG_STATUS_BAD_GAMMA;
    s[7] = 1.0;
  } else if (!(g > s[6])) {
    s[10] = SFEM_CG_STATUS_CONVERGED;
    s[7] = 1.0;
  } else if (s[8] >= maxiter) {
    s[10] = SFEM_CG_STATUS_MAXITER;
    s[7] = 1.0;
  }
}

// x += alpha p;  p = z + beta p   for the members active in this iteration
template <typename T>
__global__ void __launch_bounds__(256)
ens_update_xp_kernel(T* __restrict__ x, T* __restrict__ p,
                     const T* __restrict__ z, int64_t len,
                     const double* __restrict__ scalars) {
  const int m = blockIdx.y;
  const double* s = scalars + (int64_t)m * ENS_NS;
  if (s[9] == 0.0) return;
  const T alpha = (T)s[3], beta = (T)s[4];
  T* xm = x + (int64_t)m * len;
  T* pm = p + (int64_t)m * len;
  const T* zm = z + (int64_t)m * len;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < len;
       i += stride) {
    const T pv = pm[i];
    xm[i] += alpha * pv;
    pm[i] = zm[i] + beta * pv;
  }
}

// out = w - (b . w / total) 1 per member (the mean projection of the pressure
// solve, navier_stokes.py:73-78, with b = B 1 of ONE member): one workgroup
// sums, all subtract -- two launches
template <typename T>
__global__ void __launch_bounds__(ENS_SUM_BLOCK)
ens_weighted_sum_kernel(const T* __restrict__ w, const T* __restrict__ b,
                        int64_t len, double* __restrict__ partials) {
  const int m = blockIdx.y, g = blockIdx.x;
  const int64_t chunk = (len + ENS_G - 1) / ENS_G;
  const int64_t lo = (int64_t)g * chunk;
  const int64_t hi = lo + chunk < len ? lo + chunk : len;
  const T* wm = w + (int64_t)m * len;
  double v = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += ENS_SUM_BLOCK)
    v += (double)wm[i] * (double)b[i];
  const double total = ens_block_sum(v);
  if (threadIdx.x == 0) partials[(int64_t)m * ENS_G + g] = total;
}

template <typename T>
__global__ void __launch_bounds__(256)
ens_subtract_mean_kernel(const T* __restrict__ w, T* __restrict__ out,
                         int64_t len, const double* __restrict__ partials,
                         double total) {
  const int m = blockIdx.y;
  const double* p = partials + (int64_t)m * ENS_G;
  double t = 0.0;
#pragma unroll
  for (int g = 0; g < ENS_G; ++g) t += p[g];
  const T c = (T)(t / total);
  const T* wm = w + (int64_t)m * len;
  T* om = out + (int64_t)m * len;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < len;
       i += stride)
    om[i] = wm[i] - c;
}

// ---- the pair for M r = r - (w . r / total) 1 per member (the mean
// projection of the pressure solve, navier_stokes.py:73-78) without storing
// z = M r: the r update also sums r.r, 1.r and w.r (sums[m][0..3][g]); the
// closing kernel forms c = w.r / total and gamma_new = r.r - c 1.r; the x / p
// update uses z = r - c on the fly.  Four launches per iteration besides the
// operator instead of seven (same arithmetic as sfem_cg_update_r_mean /
// sfem_cg_update_xp_mean of the single solve).
template <typename T>
__global__ void __launch_bounds__(ENS_SUM_BLOCK)
ens_update_r_mean_kernel(T* __restrict__ r, const T* __restrict__ ap,
                         const T* __restrict__ w, int64_t len,
                         const double* __restrict__ scalars,
                         const double* __restrict__ partials,
                         double* __restrict__ sums) {
  const int m = blockIdx.y, g = blockIdx.x;
  const double* s = scalars + (int64_t)m * ENS_NS;
  if (s[7] != 0.0) return;                 // (uniform over the workgroup)
  const double pap = ens_total(partials, m, 0);
  if (ens_bad_pap(pap)) return;
  const T alpha = (T)(s[0] / pap);
  const int64_t chunk = (len + ENS_G - 1) / ENS_G;
  const int64_t lo = (int64_t)g * chunk;
  const int64_t hi = lo + chunk < len ? lo + chunk : len;
  T* rm = r + (int64_t)m * len;
  const T* am = ap + (int64_t)m * len;
  double rr = 0.0, sr = 0.0, wr = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += ENS_SUM_BLOCK) {
    const T v = rm[i] - alpha * am[i];
    rm[i] = v;
    rr += (double)v * (double)v;
    sr += (double)v;
    wr += (double)w[i] * (double)v;
  }
  rr = ens_block_sum(rr);
  __syncthreads();
  sr = ens_block_sum(sr);
  __syncthreads();
  wr = ens_block_sum(wr);
  if (threadIdx.x == 0) {
    double* o = sums + (int64_t)m * 3 * ENS_G + g;
    o[0] = rr;
    o[ENS_G] = sr;
    o[2 * ENS_G] = wr;
  }
}

__global__ void __launch_bounds__(256)
ens_close_mean_kernel(double* __restrict__ scalars,
                      const double* __restrict__ partials,
                      const double* __restrict__ sums, double total,
                      int members, double maxiter) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= members) return;
  double* s = scalars + (int64_t)m * ENS_NS;
  s[9] = 0.0;
  if (s[7] != 0.0) return;
  const double pap = ens_total(partials, m, 0);
  s[1] = pap;
  if (ens_bad_pap(pap)) {
    s[10] = SFEM_CG_STATUS_BAD_PAP;
    s[7] = 1.0;
    return;
  }
  const double* q = sums + (int64_t)m * 3 * ENS_G;
  double rr = 0.0, sr = 0.0, wr = 0.0;
  for (int g = 0; g < ENS_G; ++g) {
    rr += q[g];
    sr += q[ENS_G + g];
    wr += q[2 * ENS_G + g];
  }
  const double c = wr / total;
  const double g = rr - c * sr;
  s[12] = c;
  s[3] = s[0] / pap;
  s[4] = g / s[0];
  s[0] = g;
  s[8] += 1.0;
  s[9] = 1.0;
  if (ens_bad_gamma(g)) {
    s[10] = SFEM_CG_STATUS_BAD_GAMMA;
    s[7] = 1.0;
  } else if (!(g > s[6])) {
    s[10] = SFEM_CG_STATUS_CONVERGED;
    s[7] = 1.0;
  } else if (s[8] >= maxiter) {
    s[10] = SFEM_CG_STATUS_MAXITER;
    s[7] = 1.0;
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
ens_update_xp_mean_kernel(T* __restrict__ x, T* __restrict__ p,
                          const T* __restrict__ r, int64_t len,
                          const double* __restrict__ scalars) {
  const int m = blockIdx.y;
  const double* s = scalars + (int64_t)m * ENS_NS;
  if (s[9] == 0.0) return;
  const T alpha = (T)s[3], beta = (T)s[4], c = (T)s[12];
  T* xm = x + (int64_t)m * len;
  T* pm = p + (int64_t)m * len;
  const T* rm = r + (int64_t)m * len;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < len;
       i += stride) {
    const T pv = pm[i];
    xm[i] += alpha * pv;
    pm[i] = (rm[i] - c) + beta * pv;
  }
}

inline unsigned ens_blocks(int64_t len) {
  const int64_t b = (len + 1023) / 1024;          // four values per thread
  return (unsigned)(b < 1 ? 1 : (b > 512 ? 512 : b));
}

}  // namespace sfem

using namespace sfem;

#define SFEM_ENS_CHECK(name)                                                  \
  SFEM_REQUIRE(len >= 0 && members >= 1 && members <= SFEM_ENS_MAX_MEMBERS,   \
               name ": len >= 0, 1 <= members <= %d", SFEM_ENS_MAX_MEMBERS);  \
  SFEM_REQUIRE(dtype == SFEM_F32 || dtype == SFEM_F64,                        \
               name ": unknown dtype %d", dtype)

extern "C" int sfem_ens_dot(const void* a, const void* b, int64_t len,
                            int members, double* partials, int which,
                            int dtype, sfem_stream_t stream) {
  SFEM_ENS_CHECK("sfem_ens_dot");
  SFEM_REQUIRE(a && b && partials && (which == 0 || which == 1),
               "sfem_ens_dot: null pointer or which not in {0, 1}");
  const dim3 grid(ENS_G, (unsigned)members);
  if (dtype == SFEM_F64)
    hipLaunchKernelGGL(ens_dot_kernel<double>, grid, dim3(ENS_SUM_BLOCK), 0,
                       as_stream(stream), (const double*)a, (const double*)b,
                       len, partials, which);
  else
    hipLaunchKernelGGL(ens_dot_kernel<float>, grid, dim3(ENS_SUM_BLOCK), 0,
                       as_stream(stream), (const float*)a, (const float*)b,
                       len, partials, which);
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

extern "C" int sfem_ens_init(double* scalars, const double* partials,
                             int members, double maxiter, double tol,
                             double atol, sfem_stream_t stream) {
  SFEM_REQUIRE(scalars && partials && members >= 1 &&
                   members <= SFEM_ENS_MAX_MEMBERS,
               "sfem_ens_init: null pointer or bad member count");
  hipLaunchKernelGGL(ens_init_kernel, dim3((members + 255) / 256), dim3(256),
                     0, as_stream(stream), scalars, partials, members, maxiter,
                     tol, atol);
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

extern "C" int sfem_ens_update_r(void* r, const void* ap, int64_t len,
                                 int members, const double* scalars,
                                 const double* partials, int dtype,
                                 sfem_stream_t stream) {
  SFEM_ENS_CHECK("sfem_ens_update_r");
  if (len == 0) return SFEM_OK;
  SFEM_REQUIRE(r && ap && scalars && partials,
               "sfem_ens_update_r: null pointer");
  const dim3 grid(ens_blocks(len), (unsigned)members);
  if (dtype == SFEM_F64)
    hipLaunchKernelGGL(ens_update_r_kernel<double>, grid, dim3(256), 0,
                       as_stream(stream), (double*)r, (const double*)ap, len,
                       scalars, partials);
  else
    hipLaunchKernelGGL(ens_update_r_kernel<float>, grid, dim3(256), 0,
                       as_stream(stream), (float*)r, (const float*)ap, len,
                       scalars, partials);
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

extern "C" int sfem_ens_close(double* scalars, const double* partials,
                              int members, double maxiter,
                              sfem_stream_t stream) {
  SFEM_REQUIRE(scalars && partials && members >= 1 &&
                   members <= SFEM_ENS_MAX_MEMBERS,
               "sfem_ens_close: null pointer or bad member count");
  hipLaunchKernelGGL(ens_close_kernel, dim3((members + 255) / 256), dim3(256),
                     0, as_stream(stream), scalars, partials, members,
                     maxiter);
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

extern "C" int sfem_ens_update_xp(void* x, void* p, const void* z, int64_t len,
                                  int members, const double* scalars,
                                  int dtype, sfem_stream_t stream) {
  SFEM_ENS_CHECK("sfem_ens_update_xp");
  if (len == 0) return SFEM_OK;
  SFEM_REQUIRE(x && p && z && scalars, "sfem_ens_update_xp: null pointer");
  const dim3 grid(ens_blocks(len), (unsigned)members);
  if (dtype == SFEM_F64)
    hipLaunchKernelGGL(ens_update_xp_kernel<double>, grid, dim3(256), 0,
                       as_stream(stream), (double*)x, (double*)p,
                       (const double*)z, len, scalars);
  else
    hipLaunchKernelGGL(ens_update_xp_kernel<float>, grid, dim3(256), 0,
                       as_stream(stream), (float*)x, (float*)p,
                       (const float*)z, len, scalars);
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

extern "C" int sfem_ens_update_r_mean(void* r, const void* ap, const void* w,
                                      int64_t len, int members,
                                      const double* scalars,
                                      const double* partials, double* sums,
                                      int dtype, sfem_stream_t stream) {
  SFEM_ENS_CHECK("sfem_ens_update_r_mean");
  if (len == 0) return SFEM_OK;
  SFEM_REQUIRE(r && ap && w && scalars && partials && sums,
               "sfem_ens_update_r_mean: null pointer");
  const dim3 grid(ENS_G, (unsigned)members);
  if (dtype == SFEM_F64)
    hipLaunchKernelGGL(ens_update_r_mean_kernel<double>, grid,
                       dim3(ENS_SUM_BLOCK), 0, as_stream(stream), (double*)r,
                       (const double*)ap, (const double*)w, len, scalars,
                       partials, sums);
  else
    hipLaunchKernelGGL(ens_update_r_mean_kernel<float>, grid,
                       dim3(ENS_SUM_BLOCK), 0, as_stream(stream), (float*)r,
                       (const float*)ap, (const float*)w, len, scalars,
                       partials, sums);
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

extern "C" int sfem_ens_close_mean(double* scalars, const double* partials,
                                   const double* sums, double total,
                                   int members, double maxiter,
                                   sfem_stream_t stream) {
  SFEM_REQUIRE(scalars && partials && sums && total != 0.0 && members >= 1 &&
                   members <= SFEM_ENS_MAX_MEMBERS,
               "sfem_ens_close_mean: null pointer, total = 0 or bad member "
               "count");
  hipLaunchKernelGGL(ens_close_mean_kernel, dim3((members + 255) / 256),
                     dim3(256), 0, as_stream(stream), scalars, partials, sums,
                     total, members, maxiter);
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

extern "C" int sfem_ens_update_xp_mean(void* x, void* p, const void* r,
                                       int64_t len, int members,
                                       const double* scalars, int dtype,
                                       sfem_stream_t stream) {
  SFEM_ENS_CHECK("sfem_ens_update_xp_mean");
  if (len == 0) return SFEM_OK;
  SFEM_REQUIRE(x && p && r && scalars, "sfem_ens_update_xp_mean: null pointer");
  const dim3 grid(ens_blocks(len), (unsigned)members);
  if (dtype == SFEM_F64)
    hipLaunchKernelGGL(ens_update_xp_mean_kernel<double>, grid, dim3(256), 0,
                       as_stream(stream), (double*)x, (double*)p,
                       (const double*)r, len, scalars);
  else
    hipLaunchKernelGGL(ens_update_xp_mean_kernel<float>, grid, dim3(256), 0,
                       as_stream(stream), (float*)x, (float*)p,
                       (const float*)r, len, scalars);
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

extern "C" int sfem_ens_subtract_weighted_mean(const void* w, const void* b,
                                               double total, void* out,
                                               double* partials, int64_t len,
                                               int members, int dtype,
                                               sfem_stream_t stream) {
  SFEM_ENS_CHECK("sfem_ens_subtract_weighted_mean");
  if (len == 0) return SFEM_OK;
  SFEM_REQUIRE(w && b && out && partials && total != 0.0,
               "sfem_ens_subtract_weighted_mean: null pointer or total = 0");
  const dim3 gsum(ENS_G, (unsigned)members);
  const dim3 gsub(ens_blocks(len), (unsigned)members);
  hipStream_t st = as_stream(stream);
  if (dtype == SFEM_F64) {
    hipLaunchKernelGGL(ens_weighted_sum_kernel<double>, gsum,
                       dim3(ENS_SUM_BLOCK), 0, st,
                       (const double*)w, (const double*)b, len, partials);
    hipLaunchKernelGGL(ens_subtract_mean_kernel<double>, gsub, dim3(256), 0,
                       st, (const double*)w, (double*)out, len, partials,
                       total);
  } else {
    hipLaunchKernelGGL(ens_weighted_sum_kernel<float>, gsum,
                       dim3(ENS_SUM_BLOCK), 0, st,
                       (const float*)w, (const float*)b, len, partials);
    hipLaunchKernelGGL(ens_subtract_mean_kernel<float>, gsub, dim3(256), 0, st,
                       (const float*)w, (float*)out, len, partials, total);
  }
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}
