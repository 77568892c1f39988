// libsfem_hip: gather / scatter / exchange / CG vector kernels for gfx950.
//
// All of these are HBM-bound streaming or indexed kernels: 64-wide waves,
// one element of work per lane per iteration with grid-stride loops, 8/16-byte
// accesses where the layout allows, float atomics only where a sum crosses
// workgroups.
#include <stdarg.h>
#include <string.h>

#include "sfem_common.h"

namespace sfem {

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}

// ---------------------------------------------------------------- gather ---
template <typename T>
__global__ void __launch_bounds__(256)
gather_kernel(const T* __restrict__ u, const int32_t* __restrict__ idx,
              T* __restrict__ out, int64_t count, T fill) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += stride) {
    const int32_t k = idx[i];
    out[i] = k < 0 ? fill : u[k];
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
gather_rows_kernel(const T* __restrict__ x, const int32_t* __restrict__ idx,
                   T* __restrict__ out, int64_t count, int ncomp) {
  const int64_t total = count * ncomp;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += stride) {
    const int64_t i = t / ncomp;
    const int c = (int)(t - i * ncomp);
    const int32_t k = idx[i];
    out[t] = k < 0 ? T(0) : x[(int64_t)k * ncomp + c];
  }
}

// --------------------------------------------------------------- scatter ---
template <typename T>
__global__ void __launch_bounds__(256)
scatter_add_kernel(const T* __restrict__ u_local,
                   const int32_t* __restrict__ idx, T* __restrict__ out,
                   int64_t count, int ncomp) {
  const int64_t total = count * ncomp;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += stride) {
    const int64_t i = t / ncomp;
    const int c = (int)(t - i * ncomp);
    const int32_t k = idx[i];
    if (k >= 0) unsafeAtomicAdd(&out[(int64_t)k * ncomp + c], u_local[t]);
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
scatter_csr_kernel(const T* __restrict__ u_local,
                   const int64_t* __restrict__ offsets,
                   const int32_t* __restrict__ slots, T* __restrict__ out,
                   int64_t num_nodes, int ncomp) {
  const int64_t total = num_nodes * ncomp;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += stride) {
    const int64_t v = t / ncomp;
    const int c = (int)(t - v * ncomp);
    T acc = T(0);
    for (int64_t s = offsets[v]; s < offsets[v + 1]; ++s)
      acc += u_local[(int64_t)slots[s] * ncomp + c];
    out[t] = acc;
  }
}

// -------------------------------------------------------------- exchange ---
template <typename T>
__global__ void __launch_bounds__(256)
exchange_sum_kernel(const T* __restrict__ u, const int32_t* __restrict__ gidx,
                    const int32_t* __restrict__ unique, T* __restrict__ sums,
                    int64_t count, int ncomp) {
  const int64_t total = count * ncomp;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += stride) {
    const int64_t i = t / ncomp;
    const int c = (int)(t - i * ncomp);
    const int32_t k = gidx[i];
    if (k >= 0)
      unsafeAtomicAdd(&sums[(int64_t)unique[i] * ncomp + c],
                      u[(int64_t)k * ncomp + c]);
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
exchange_expand_kernel(const T* __restrict__ sums,
                       const int32_t* __restrict__ gidx,
                       const int32_t* __restrict__ unique, T* __restrict__ out,
                       int64_t count, int ncomp) {
  const int64_t total = count * ncomp;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += stride) {
    const int64_t i = t / ncomp;
    const int c = (int)(t - i * ncomp);
    const int32_t k = gidx[i];
    if (k >= 0)
      out[(int64_t)k * ncomp + c] = sums[(int64_t)unique[i] * ncomp + c];
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
unpack_add_kernel(const T* __restrict__ buf, const int32_t* __restrict__ idx,
                  T* __restrict__ u, int64_t count, int ncomp) {
  const int64_t total = count * ncomp;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += stride) {
    const int64_t i = t / ncomp;
    const int c = (int)(t - i * ncomp);
    const int32_t k = idx[i];
    if (k >= 0) u[(int64_t)k * ncomp + c] += buf[t];
  }
}

// Strided variants for the single-launch partition exchange: `u` is any
// (N, ncomp) view (row-major or component-major); `idx` may name a node more
// than once (edge / corner nodes shared with several neighbours), hence atomics.
template <typename T>
__global__ void __launch_bounds__(256)
pack_strided_kernel(const T* __restrict__ u, const int32_t* __restrict__ idx,
                    T* __restrict__ buf, int64_t count, int ncomp,
                    int64_t node_stride, int64_t comp_stride) {
  const int64_t total = count * ncomp;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += stride) {
    const int64_t i = t / ncomp;
    const int c = (int)(t - i * ncomp);
    const int32_t k = idx[i];
    buf[t] = k >= 0 ? u[(int64_t)k * node_stride + c * comp_stride] : T(0);
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
unpack_add_atomic_kernel(const T* __restrict__ buf,
                         const int32_t* __restrict__ idx, T* __restrict__ u,
                         int64_t count, int ncomp, int64_t node_stride,
                         int64_t comp_stride) {
  const int64_t total = count * ncomp;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += stride) {
    const int64_t i = t / ncomp;
    const int c = (int)(t - i * ncomp);
    const int32_t k = idx[i];
    if (k >= 0)
      unsafeAtomicAdd(&u[(int64_t)k * node_stride + c * comp_stride], buf[t]);
  }
}

// QQ^T through the classes themselves: class c owns members[offsets[c] ..
// offsets[c+1]); one thread sums them (in member order: reproducible) and
// writes the sum back to each.  In place, no workspace, all components in
// one launch; `u[k * node_stride + c * comp_stride]` covers (N, nc) row-major
// and component-major fields alike.
template <typename T>
__global__ void __launch_bounds__(256)
exchange_classes_kernel(T* __restrict__ u, const int32_t* __restrict__ members,
                        const int32_t* __restrict__ offsets,
                        int64_t num_classes, int ncomp, int64_t node_stride,
                        int64_t comp_stride) {
  const int64_t total = num_classes * ncomp;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += stride) {
    int64_t cls;
    int c;
    if (comp_stride == 1) {          // interleaved: components adjacent
      cls = t / ncomp;
      c = (int)(t - cls * ncomp);
    } else {                         // strips: classes adjacent
      c = (int)(t / num_classes);
      cls = t - (int64_t)c * num_classes;
    }
    const int32_t lo = offsets[cls], hi = offsets[cls + 1];
    T sum = T(0);
    for (int32_t m = lo; m < hi; ++m)
      sum += u[(int64_t)members[m] * node_stride + c * comp_stride];
    for (int32_t m = lo; m < hi; ++m)
      u[(int64_t)members[m] * node_stride + c * comp_stride] = sum;
  }
}

// Clears `nstrips` equally long strips of a field in one launch (the shared
// node range of every component before an atomically assembled apply).
template <typename T>
__global__ void __launch_bounds__(256)
zero_strips_kernel(T* __restrict__ base, int64_t strip_len,
                   int64_t strip_stride, int nstrips) {
  const int64_t total = strip_len * nstrips;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += stride) {
    const int64_t s = t / strip_len;
    base[s * strip_stride + (t - s * strip_len)] = T(0);
  }
}

// ------------------------------------------------------------ CG kernels ---
// Block-level sum of one double per thread; one atomic per workgroup.
__device__ inline double block_sum(double v) {
  __shared__ double partial[16];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) partial[wave] = v;
  __syncthreads();
  double total = 0.0;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 0; w < nw; ++w) total += partial[w];
  }
  return total;  // valid on thread 0
}

// three sums at once (results valid on thread 0)
__device__ inline void block_sum3(double (&v)[3]) {
  __shared__ double partial3[3][16];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) partial3[k][wave] = v[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      double total = 0.0;
      for (int w = 0; w < nw; ++w) total += partial3[k][w];
      v[k] = total;
    }
  }
}

// Streaming vector kernels move 16 bytes per lane per access (dwordx4: the
// widest, 1 KiB per wave-instruction) with 4 independent accesses in flight per
// lane; the last < VEC elements are handled by a scalar tail.
template <typename T>
struct Vec16;
template <>
struct Vec16<double> {
  using type = double2;
  static constexpr int N = 2;
};
template <>
struct Vec16<float> {
  using type = float4;
  static constexpr int N = 4;
};

// Non-temporal 16-byte accesses for vectors that are streamed once per kernel.
template <typename T>
struct NativeVec16;
template <>
struct NativeVec16<double> {
  typedef double type __attribute__((ext_vector_type(2)));
};
template <>
struct NativeVec16<float> {
  typedef float type __attribute__((ext_vector_type(4)));
};

template <typename T>
__device__ __forceinline__ typename Vec16<T>::type nt_load(
    const typename Vec16<T>::type* p) {
  using NV = typename NativeVec16<T>::type;
  const NV v = __builtin_nontemporal_load(reinterpret_cast<const NV*>(p));
  typename Vec16<T>::type out;
  *reinterpret_cast<NV*>(&out) = v;
  return out;
}

template <typename T>
__device__ __forceinline__ void nt_store(const typename Vec16<T>::type& v,
                                         typename Vec16<T>::type* p) {
  using NV = typename NativeVec16<T>::type;
  __builtin_nontemporal_store(*reinterpret_cast<const NV*>(&v),
                              reinterpret_cast<NV*>(p));
}

// NT = the vectors are far larger than the caches (a launch-time decision): do
// not let them evict what the gathers of the next apply could still use.
// Measured at 90 M nodes: CG iteration 2.21 -> 2.08 ms; at 11 M nodes and
// below the cached form is the faster one.
template <typename T, bool NT>
__device__ __forceinline__ typename Vec16<T>::type ld16(
    const typename Vec16<T>::type* p) {
  if (NT) return nt_load<T>(p);
  return *p;
}

template <typename T, bool NT>
__device__ __forceinline__ void st16(const typename Vec16<T>::type& v,
                                     typename Vec16<T>::type* p) {
  if (NT) nt_store<T>(v, p);
  else *p = v;
}

template <typename T>
__device__ __forceinline__ T vget(const typename Vec16<T>::type& v, int i) {
  return reinterpret_cast<const T*>(&v)[i];
}

template <typename T>
__global__ void __launch_bounds__(512)
dot_kernel(const T* __restrict__ a, const T* __restrict__ b, int64_t count,
           double* __restrict__ result) {
  using V = typename Vec16<T>::type;
  constexpr int VN = Vec16<T>::N;
  const int64_t nvec = count / VN;
  const V* av = reinterpret_cast<const V*>(a);
  const V* bv = reinterpret_cast<const V*>(b);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += stride) {
    const V x = av[i], y = bv[i];
#pragma unroll
    for (int c = 0; c < VN; ++c) acc += (double)vget<T>(x, c) * (double)vget<T>(y, c);
  }
  if (blockIdx.x == 0 && threadIdx.x < count - nvec * VN) {
    const int64_t i = nvec * VN + threadIdx.x;
    acc += (double)a[i] * (double)b[i];
  }
  const double total = block_sum(acc);
  if (threadIdx.x == 0) unsafeAtomicAdd(result, total);
}

// out = w - (b . w / total) 1 in two launches without atomics or a cleared
// accumulator: every workgroup of the first kernel stores its partial sum, every
// workgroup of the second adds the (<= 1024) partials up again.
template <typename T>
__global__ void __launch_bounds__(512)
dot_partials_kernel(const T* __restrict__ a, const T* __restrict__ b,
                    int64_t count, double* __restrict__ partials) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += stride)
    acc += (double)a[i] * (double)b[i];
  const double total = block_sum(acc);
  if (threadIdx.x == 0) partials[blockIdx.x] = total;
}

template <typename T>
__global__ void __launch_bounds__(512)
subtract_mean_kernel(const T* __restrict__ w, const double* __restrict__ partials,
                     int num_partials, double inv_total, T* __restrict__ out,
                     int64_t count, double* __restrict__ dot_result) {
  __shared__ double mean;
  double acc = 0.0;
  for (int i = threadIdx.x; i < num_partials; i += blockDim.x)
    acc += partials[i];
  const double total = block_sum(acc);
  if (threadIdx.x == 0) mean = total * inv_total;
  __syncthreads();
  const T m = (T)mean;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double wz = 0.0;      // w . out: the r . z of a CG that uses this as M
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += stride) {
    const T wi = w[i], zi = wi - m;
    out[i] = zi;
    wz += (double)wi * (double)zi;
  }
  if (dot_result) {
    const double tot = block_sum(wz);
    if (threadIdx.x == 0) unsafeAtomicAdd(dot_result, tot);
  }
}

// scalars (16 doubles): [0] gamma [1] p.Ap [2] gamma_new [3] alpha [4] beta
//   [5] b.b [6] atol2 [7] done flag (0/1) [8] iterations
// result += scale * sum_i w[i] a[idx[i]] b[idx[i]]  over a short index list
// (the interface correction of the partitioned CG inner products).
template <typename T>
__global__ void __launch_bounds__(256)
dot_indexed_kernel(const T* __restrict__ a, const T* __restrict__ b,
                   const int64_t* __restrict__ idx,
                   const double* __restrict__ w, int64_t count, int ncomp,
                   int64_t node_stride, int64_t comp_stride, double scale,
                   double* __restrict__ result) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += stride) {
    const int64_t k = idx[i] * node_stride;
    double s = 0.0;
    for (int c = 0; c < ncomp; ++c)
      s += (double)a[k + c * comp_stride] * (double)b[k + c * comp_stride];
    acc += w[i] * s;
  }
  const double total = block_sum(acc);
  if (threadIdx.x == 0 && total != 0.0) unsafeAtomicAdd(result, scale * total);
}

// gamma_new = scalars[2] + the SFEM_CG_RR_SLOTS partial sums behind the named
// scalars.  `cg_update_r` with fuse_rr = 2 spreads its per-workgroup r.r sums
// over those slots: one address would serialise the atomics of all workgroups
// (the reason the fused kernels were held to 16 workgroups per CU), 64 let the
// update stream with 128 per CU.  Every wave sums the slots for itself.
__device__ __forceinline__ double cg_gamma_new(const double* scalars) {
  double v = scalars[SFEM_CG_NSCALARS_NAMED + (threadIdx.x & 63)];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return scalars[2] + v;
}

template <typename T, bool FUSE_RR>
__global__ void __launch_bounds__(512)
cg_update_xr_kernel(T* __restrict__ x, T* __restrict__ r,
                    const T* __restrict__ p, const T* __restrict__ ap,
                    int64_t count, double* __restrict__ scalars) {
  if (scalars[7] != 0.0) return;  // converged: iteration is a no-op
  using V = typename Vec16<T>::type;
  constexpr int VN = Vec16<T>::N;
  const T alpha = (T)(scalars[0] / scalars[1]);
  const int64_t nvec = count / VN;
  V* xv = reinterpret_cast<V*>(x);
  V* rv = reinterpret_cast<V*>(r);
  const V* pv = reinterpret_cast<const V*>(p);
  const V* apv = reinterpret_cast<const V*>(ap);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += stride) {
    V xx = xv[i], rr = rv[i];
    const V pp = pv[i], aa = apv[i];
#pragma unroll
    for (int c = 0; c < VN; ++c) {
      T* xe = reinterpret_cast<T*>(&xx) + c;
      T* re = reinterpret_cast<T*>(&rr) + c;
      *xe += alpha * vget<T>(pp, c);
      *re -= alpha * vget<T>(aa, c);
      if (FUSE_RR) acc += (double)*re * (double)*re;
    }
    xv[i] = xx;
    rv[i] = rr;
  }
  if (blockIdx.x == 0 && threadIdx.x < count - nvec * VN) {
    const int64_t i = nvec * VN + threadIdx.x;
    x[i] += alpha * p[i];
    const T rn = r[i] - alpha * ap[i];
    r[i] = rn;
    if (FUSE_RR) acc += (double)rn * (double)rn;
  }
  if (FUSE_RR) {
    const double total = block_sum(acc);
    if (threadIdx.x == 0) unsafeAtomicAdd(&scalars[2], total);
  }
}

template <typename T>
__global__ void __launch_bounds__(512)
cg_update_p_kernel(T* __restrict__ p, const T* __restrict__ z, int64_t count,
                   const double* __restrict__ scalars) {
  if (scalars[7] != 0.0) return;
  using V = typename Vec16<T>::type;
  constexpr int VN = Vec16<T>::N;
  const T beta = (T)(cg_gamma_new(scalars) / scalars[0]);
  const int64_t nvec = count / VN;
  V* pv = reinterpret_cast<V*>(p);
  const V* zv = reinterpret_cast<const V*>(z);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += stride) {
    V pp = pv[i];
    const V zz = zv[i];
#pragma unroll
    for (int c = 0; c < VN; ++c) {
      T* pe = reinterpret_cast<T*>(&pp) + c;
      *pe = vget<T>(zz, c) + beta * *pe;
    }
    pv[i] = pp;
  }
  if (blockIdx.x == 0 && threadIdx.x < count - nvec * VN) {
    const int64_t i = nvec * VN + threadIdx.x;
    p[i] = z[i] + beta * p[i];
  }
}

// 8-pass split of the two updates (instead of 9): the x update rides with the
// p update, where p is in registers anyway.
//   update_r : r -= alpha Ap (+ gamma_new += r.r)        reads r, Ap; writes r
//   update_xp: x += alpha p;  p = z + beta p             reads x, p, z; writes x, p
template <typename T, bool FUSE_RR, bool NT, bool STRIPED = false>
__global__ void __launch_bounds__(512)
cg_update_r_kernel(T* __restrict__ r, const T* __restrict__ ap, int64_t count,
                   double* __restrict__ scalars) {
  if (scalars[7] != 0.0) return;
  using V = typename Vec16<T>::type;
  constexpr int VN = Vec16<T>::N;
  const T alpha = (T)(scalars[0] / scalars[1]);
  const int64_t nvec = count / VN;
  V* rv = reinterpret_cast<V*>(r);
  const V* apv = reinterpret_cast<const V*>(ap);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += stride) {
    V rr = ld16<T, NT>(&rv[i]);
    const V aa = ld16<T, NT>(&apv[i]);
#pragma unroll
    for (int c = 0; c < VN; ++c) {
      T* re = reinterpret_cast<T*>(&rr) + c;
      *re -= alpha * vget<T>(aa, c);
      if (FUSE_RR) acc += (double)*re * (double)*re;
    }
    st16<T, NT>(rr, &rv[i]);
  }
  if (blockIdx.x == 0 && threadIdx.x < count - nvec * VN) {
    const int64_t i = nvec * VN + threadIdx.x;
    const T rn = r[i] - alpha * ap[i];
    r[i] = rn;
    if (FUSE_RR) acc += (double)rn * (double)rn;
  }
  if (FUSE_RR) {
    const double total = block_sum(acc);
    if (threadIdx.x == 0)
      unsafeAtomicAdd(STRIPED ? &scalars[SFEM_CG_NSCALARS_NAMED +
                                         (blockIdx.x & (SFEM_CG_RR_SLOTS - 1))]
                              : &scalars[2], total);
  }
}

template <typename T, bool NT>
__global__ void __launch_bounds__(512)
cg_update_xp_kernel(T* __restrict__ x, T* __restrict__ p,
                    const T* __restrict__ z, int64_t count,
                    const double* __restrict__ scalars) {
  if (scalars[7] != 0.0) return;
  using V = typename Vec16<T>::type;
  constexpr int VN = Vec16<T>::N;
  const T alpha = (T)(scalars[0] / scalars[1]);
  const T beta = (T)(cg_gamma_new(scalars) / scalars[0]);
  const int64_t nvec = count / VN;
  V* xv = reinterpret_cast<V*>(x);
  V* pv = reinterpret_cast<V*>(p);
  const V* zv = reinterpret_cast<const V*>(z);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += stride) {
    V xx = ld16<T, NT>(&xv[i]);
    V pp = ld16<T, NT>(&pv[i]);
    const V zz = ld16<T, NT>(&zv[i]);
#pragma unroll
    for (int c = 0; c < VN; ++c) {
      T* xe = reinterpret_cast<T*>(&xx) + c;
      T* pe = reinterpret_cast<T*>(&pp) + c;
      *xe += alpha * *pe;
      *pe = vget<T>(zz, c) + beta * *pe;
    }
    st16<T, NT>(xx, &xv[i]);
    st16<T, NT>(pp, &pv[i]);
  }
  if (blockIdx.x == 0 && threadIdx.x < count - nvec * VN) {
    const int64_t i = nvec * VN + threadIdx.x;
    const T pi = p[i];
    x[i] += alpha * pi;
    p[i] = z[i] + beta * pi;
  }
}

// Lazy solution update.  x is only ever read by x += alpha p, so it need not
// be touched every iteration: the last m directions are kept in a ring
// (p_k lives in slot k mod m of `pring`, p_{k+1} = z + beta p_k is written to
// the next slot) and every m-th iteration adds the m pending terms
//     x = (((x + alpha_{k-m+1} p_{k-m+1}) + ...) + alpha_k p_k)
// in the order, and with the roundings, the iteration-by-iteration update
// would have used: bitwise the same x.  Vector passes of this kernel per
// iteration: (3 (m - 1) + (m + 4)) / m instead of 5 -- 4.25 at m = 4.
//   lazy[0]      = iterations whose term is already in x (raised only by
//                  sfem_cg_flush_x, which the host calls before it reads x)
//   lazy[1 + s]  = alpha of the direction in slot s
// k = scalars[8] (iterations closed so far) picks the slots.
template <typename T, bool NT>
__global__ void __launch_bounds__(512)
cg_update_xp_lazy_kernel(T* __restrict__ x, T* __restrict__ pring,
                         int64_t ring_stride, const T* __restrict__ z,
                         int64_t count, const double* __restrict__ scalars,
                         double* __restrict__ lazy, int m) {
  if (scalars[7] != 0.0) return;
  using V = typename Vec16<T>::type;
  constexpr int VN = Vec16<T>::N;
  const int64_t k = (int64_t)scalars[8];
  const int slot = (int)(k % m), next = (int)((k + 1) % m);
  const T alpha = (T)(scalars[0] / scalars[1]);
  const T beta = (T)(cg_gamma_new(scalars) / scalars[0]);
  const bool flush = next == 0;
  int64_t lo = (int64_t)lazy[0];
  if (lo < k + 1 - m) lo = k + 1 - m;
  // alphas of the older pending directions (written by earlier launches)
  T a_old[SFEM_CG_LAZY_MAX];
  int s_old[SFEM_CG_LAZY_MAX];
  int n_old = 0;
  if (flush) {
    for (int64_t j = lo; j < k; ++j) {
      s_old[n_old] = (int)(j % m);
      a_old[n_old++] = (T)lazy[1 + (j % m)];
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) lazy[1 + slot] = (double)alpha;
  const int64_t nvec = count / VN;
  V* xv = reinterpret_cast<V*>(x);
  const V* pk = reinterpret_cast<const V*>(pring + slot * ring_stride);
  V* pn = reinterpret_cast<V*>(pring + next * ring_stride);
  const V* zv = reinterpret_cast<const V*>(z);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += stride) {
    const V pp = ld16<T, NT>(&pk[i]);
    const V zz = ld16<T, NT>(&zv[i]);
    if (flush) {
      V xx = ld16<T, NT>(&xv[i]);
      for (int q = 0; q < n_old; ++q) {
        const V po = ld16<T, NT>(
            reinterpret_cast<const V*>(pring + s_old[q] * ring_stride) + i);
#pragma unroll
        for (int c = 0; c < VN; ++c)
          reinterpret_cast<T*>(&xx)[c] += a_old[q] * vget<T>(po, c);
      }
#pragma unroll
      for (int c = 0; c < VN; ++c)
        reinterpret_cast<T*>(&xx)[c] += alpha * vget<T>(pp, c);
      st16<T, NT>(xx, &xv[i]);
    }
    V np;
#pragma unroll
    for (int c = 0; c < VN; ++c)
      reinterpret_cast<T*>(&np)[c] = vget<T>(zz, c) + beta * vget<T>(pp, c);
    st16<T, NT>(np, &pn[i]);
  }
  if (blockIdx.x == 0 && threadIdx.x < count - nvec * VN) {
    const int64_t i = nvec * VN + threadIdx.x;
    const T pi = pring[slot * ring_stride + i];
    if (flush) {
      T xx = x[i];
      for (int q = 0; q < n_old; ++q)
        xx += a_old[q] * pring[s_old[q] * ring_stride + i];
      xx += alpha * pi;
      x[i] = xx;
    }
    pring[next * ring_stride + i] = z[i] + beta * pi;
  }
}

// x += the terms of the iterations [max(lazy[0], iters - iters mod m), iters)
// that the lazy update has not added yet (iters = scalars[8]); the caller
// then records lazy[0] = iters (cg_lazy_mark_kernel).
template <typename T>
__global__ void __launch_bounds__(512)
cg_flush_x_kernel(T* __restrict__ x, const T* __restrict__ pring,
                  int64_t ring_stride, int64_t count,
                  const double* __restrict__ scalars,
                  const double* __restrict__ lazy, int m) {
  const int64_t iters = (int64_t)scalars[8];
  int64_t lo = (int64_t)lazy[0];
  if (lo < iters - iters % m) lo = iters - iters % m;
  if (lo >= iters) return;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += stride) {
    T xx = x[i];
    for (int64_t j = lo; j < iters; ++j)
      xx += (T)lazy[1 + (j % m)] * pring[(j % m) * ring_stride + i];
    x[i] = xx;
  }
}

__global__ void cg_lazy_mark_kernel(const double* __restrict__ scalars,
                                    double* __restrict__ lazy) {
  if (threadIdx.x == 0 && blockIdx.x == 0) lazy[0] = scalars[8];
}

// The same two updates for the preconditioner  M r = r - (w . r / total) 1
// (the mean projection of the pressure solve, navier_stokes.py:73-78): z = M r
// is never stored.
//   update_r_mean : r -= alpha Ap; sums r.r, 1.r and w.r   reads r, Ap, w; writes r
//   update_xp_mean: c = w.r / total;  gamma_new = r.z = r.r - c 1.r;
//                   x += alpha p;  p = (r - c) + beta p     reads x, p, r; writes x, p
// 9 vector passes per iteration instead of 12 (dot, subtraction and the plain
// updates).  r.r goes to the striped slots of `scalars`; 1.r and w.r to
// `sums` = 2 sets x (SFEM_CG_RR_SLOTS for 1.r, SFEM_CG_RR_SLOTS for w.r): the
// set of the iteration's parity is accumulated and read, the other is cleared
// by update_xp_mean for the next iteration.  scalars[11] receives -c 1.r, which
// the closing phase adds to gamma_new; scalars[12] = c.
template <typename T, bool NT>
__global__ void __launch_bounds__(512)
cg_update_r_mean_kernel(T* __restrict__ r, const T* __restrict__ ap,
                        const T* __restrict__ w, int64_t count,
                        double* __restrict__ scalars,
                        double* __restrict__ sums) {
  if (scalars[7] != 0.0) return;
  using V = typename Vec16<T>::type;
  constexpr int VN = Vec16<T>::N;
  const T alpha = (T)(scalars[0] / scalars[1]);
  double* set = sums + (((int64_t)scalars[8]) & 1) * (2 * SFEM_CG_RR_SLOTS);
  const int64_t nvec = count / VN;
  V* rv = reinterpret_cast<V*>(r);
  const V* apv = reinterpret_cast<const V*>(ap);
  const V* wv = reinterpret_cast<const V*>(w);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double rr_acc = 0.0, one_acc = 0.0, w_acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += stride) {
    V rr = ld16<T, NT>(&rv[i]);
    const V aa = ld16<T, NT>(&apv[i]);
    const V ww = ld16<T, NT>(&wv[i]);
#pragma unroll
    for (int c = 0; c < VN; ++c) {
      T* re = reinterpret_cast<T*>(&rr) + c;
      *re -= alpha * vget<T>(aa, c);
      const double v = (double)*re;
      rr_acc += v * v;
      one_acc += v;
      w_acc += v * (double)vget<T>(ww, c);
    }
    st16<T, NT>(rr, &rv[i]);
  }
  if (blockIdx.x == 0 && threadIdx.x < count - nvec * VN) {
    const int64_t i = nvec * VN + threadIdx.x;
    const T rn = r[i] - alpha * ap[i];
    r[i] = rn;
    rr_acc += (double)rn * (double)rn;
    one_acc += (double)rn;
    w_acc += (double)rn * (double)w[i];
  }
  double t[3] = {rr_acc, one_acc, w_acc};
  block_sum3(t);
  if (threadIdx.x == 0) {
    const int q = blockIdx.x & (SFEM_CG_RR_SLOTS - 1);
    unsafeAtomicAdd(&scalars[SFEM_CG_NSCALARS_NAMED + q], t[0]);
    unsafeAtomicAdd(&set[q], t[1]);
    unsafeAtomicAdd(&set[SFEM_CG_RR_SLOTS + q], t[2]);
  }
}

template <typename T, bool NT>
__global__ void __launch_bounds__(512)
cg_update_xp_mean_kernel(T* __restrict__ x, T* __restrict__ p,
                         const T* __restrict__ r, int64_t count,
                         double* __restrict__ scalars,
                         double* __restrict__ sums, double total) {
  if (scalars[7] != 0.0) return;
  using V = typename Vec16<T>::type;
  constexpr int VN = Vec16<T>::N;
  const int par = (int)(((int64_t)scalars[8]) & 1);
  const double* set = sums + par * (2 * SFEM_CG_RR_SLOTS);
  double one_r = set[threadIdx.x & 63];
  double w_r = set[SFEM_CG_RR_SLOTS + (threadIdx.x & 63)];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    one_r += __shfl_xor(one_r, off, 64);
    w_r += __shfl_xor(w_r, off, 64);
  }
  const double mean = w_r / total;
  const double corr = -mean * one_r;
  const double gamma_new = cg_gamma_new(scalars) + corr;
  const T alpha = (T)(scalars[0] / scalars[1]);
  const T beta = (T)(gamma_new / scalars[0]);
  const T cm = (T)mean;
  const int64_t nvec = count / VN;
  V* xv = reinterpret_cast<V*>(x);
  V* pv = reinterpret_cast<V*>(p);
  const V* rv = reinterpret_cast<const V*>(r);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += stride) {
    V xx = ld16<T, NT>(&xv[i]);
    V pp = ld16<T, NT>(&pv[i]);
    const V rr = ld16<T, NT>(&rv[i]);
#pragma unroll
    for (int c = 0; c < VN; ++c) {
      T* xe = reinterpret_cast<T*>(&xx) + c;
      T* pe = reinterpret_cast<T*>(&pp) + c;
      *xe += alpha * *pe;
      *pe = (vget<T>(rr, c) - cm) + beta * *pe;
    }
    st16<T, NT>(xx, &xv[i]);
    st16<T, NT>(pp, &pv[i]);
  }
  if (blockIdx.x == 0) {
    if (threadIdx.x < count - nvec * VN) {
      const int64_t i = nvec * VN + threadIdx.x;
      const T pi = p[i];
      x[i] += alpha * pi;
      p[i] = (r[i] - cm) + beta * pi;
    }
    // nobody reads these in this kernel: the correction of gamma_new for the
    // closing phase, and the other parity's sums for the next iteration
    if (threadIdx.x == 0) {
      scalars[11] = corr;
      scalars[12] = mean;
    }
    if (threadIdx.x < 2 * SFEM_CG_RR_SLOTS)
      sums[(par ^ 1) * (2 * SFEM_CG_RR_SLOTS) + threadIdx.x] = 0.0;
  }
}

// Layered assembly (sfem_helmholtz_args.layered_extent): `ap` is an extended
// vector [ N nodal values | layer 1 | layer 2 | ... ]; layer k covers the nodes
// [0, len[k]) and starts at element off[k].  The operator wrote every
// contribution of a shared node to a layer of its own with a plain store; here,
// where Ap is streamed anyway, they are added up in layer order (a fixed order:
// the result is bitwise reproducible) -- no atomics, no cleared range.
struct LayerDesc {
  int nl;
  int64_t len[SFEM_MAX_LAYERS];   // multiples of the 16-byte vector width
  int64_t off[SFEM_MAX_LAYERS];
  // optional: one byte per SFEM_LAYER_CHUNK nodes of the layer, 0 = no element
  // writes into that chunk (it holds zeros: not read).  The refiner numbers
  // runs of edge blocks between runs of face blocks, so most of the second and
  // third layer of a hexahedral mesh is such chunks.
  const uint8_t* mask[SFEM_MAX_LAYERS];
};

template <typename T, bool NT>
__device__ __forceinline__ typename Vec16<T>::type add_layers(
    typename Vec16<T>::type aa, const T* __restrict__ ap, int64_t i,
    const LayerDesc& ld) {
  using V = typename Vec16<T>::type;
  constexpr int VN = Vec16<T>::N;
  for (int k = 0; k < ld.nl; ++k) {
    if (i * VN >= ld.len[k]) break;            // lengths do not increase
    if (ld.mask[k] && !ld.mask[k][(i * VN) / SFEM_LAYER_CHUNK]) continue;
    const V l = ld16<T, NT>(reinterpret_cast<const V*>(ap + ld.off[k]) + i);
#pragma unroll
    for (int c = 0; c < VN; ++c)
      reinterpret_cast<T*>(&aa)[c] += vget<T>(l, c);
  }
  return aa;
}

template <typename T, bool FUSE_RR, bool NT, bool STRIPED>
__global__ void __launch_bounds__(512)
cg_update_r_layered_kernel(T* __restrict__ r, const T* __restrict__ ap,
                           int64_t count, LayerDesc ld,
                           double* __restrict__ scalars,
                           double* __restrict__ rr_partials) {
  if (scalars[7] != 0.0) return;
  using V = typename Vec16<T>::type;
  constexpr int VN = Vec16<T>::N;
  const T alpha = (T)(scalars[0] / scalars[1]);
  const int64_t nvec = count / VN;
  V* rv = reinterpret_cast<V*>(r);
  const V* apv = reinterpret_cast<const V*>(ap);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += stride) {
    V rr = ld16<T, NT>(&rv[i]);
    const V aa = add_layers<T, NT>(ld16<T, NT>(&apv[i]), ap, i, ld);
#pragma unroll
    for (int c = 0; c < VN; ++c) {
      T* re = reinterpret_cast<T*>(&rr) + c;
      *re -= alpha * vget<T>(aa, c);
      if (FUSE_RR) acc += (double)*re * (double)*re;
    }
    st16<T, NT>(rr, &rv[i]);
  }
  if (blockIdx.x == 0 && threadIdx.x < count - nvec * VN) {
    const int64_t i = nvec * VN + threadIdx.x;
    T a = ap[i];
    for (int k = 0; k < ld.nl; ++k)
      if (i < ld.len[k]) a += ap[ld.off[k] + i];
    const T rn = r[i] - alpha * a;
    r[i] = rn;
    if (FUSE_RR) acc += (double)rn * (double)rn;
  }
  if (FUSE_RR) {
    const double total = block_sum(acc);
    if (threadIdx.x == 0) {
      if (rr_partials)        // stored, summed in index order by phase 8
        rr_partials[blockIdx.x] = total;
      else
        unsafeAtomicAdd(STRIPED ? &scalars[SFEM_CG_NSCALARS_NAMED +
                                           (blockIdx.x & (SFEM_CG_RR_SLOTS - 1))]
                                : &scalars[2], total);
    }
  }
}

// out[i] += layer_1[i] + layer_2[i] + ...  for i < len[0]: the assembled vector
// for consumers that do not add the layers up themselves.
template <typename T>
__global__ void __launch_bounds__(512)
fold_layers_kernel(T* __restrict__ out, int64_t count, LayerDesc ld) {
  using V = typename Vec16<T>::type;
  constexpr int VN = Vec16<T>::N;
  const int64_t top = ld.nl ? (ld.len[0] < count ? ld.len[0] : count) : 0;
  const int64_t nvec = top / VN;
  V* ov = reinterpret_cast<V*>(out);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += stride)
    ov[i] = add_layers<T, false>(ov[i], out, i, ld);
  if (blockIdx.x == 0 && threadIdx.x < top - nvec * VN) {
    const int64_t i = nvec * VN + threadIdx.x;
    T a = out[i];
    for (int k = 0; k < ld.nl; ++k)
      if (i < ld.len[k]) a += out[ld.off[k] + i];
    out[i] = a;
  }
}

// The same at a list of distinct nodes, and the folded layer slots are cleared:
// a partitioned operator makes the values of its interface nodes whole before
// it packs them for the neighbours, and whoever adds the layers up afterwards
// (`r -= alpha Ap`) finds zeros there.
template <typename T>
__global__ void __launch_bounds__(256)
fold_layers_at_kernel(T* __restrict__ ext, const int64_t* __restrict__ idx,
                      int64_t count, LayerDesc ld) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < count;
       t += stride) {
    const int64_t n = idx[t];
    if (n < 0) continue;
    T a = ext[n];
    for (int k = 0; k < ld.nl; ++k) {
      if (n >= ld.len[k]) break;
      T* slot = ext + ld.off[k] + n;
      a += *slot;
      *slot = T(0);
    }
    ext[n] = a;
  }
}

static int make_layer_desc(const char* who, const int64_t* layer_len,
                           const int64_t* layer_off, int num_layers,
                           int64_t count, int vn, LayerDesc* ld,
                           const uint8_t* masks = nullptr,
                           const int64_t* mask_off = nullptr) {
  SFEM_REQUIRE(num_layers >= 0 && num_layers <= SFEM_MAX_LAYERS,
               "%s: %d layers (at most %d)", who, num_layers, SFEM_MAX_LAYERS);
  SFEM_REQUIRE(num_layers == 0 || (layer_len && layer_off),
               "%s: null layer description", who);
  ld->nl = num_layers;
  for (int k = 0; k < num_layers; ++k) {
    SFEM_REQUIRE(layer_len[k] >= 0 && layer_len[k] % vn == 0 &&
                     layer_off[k] >= count && layer_off[k] % vn == 0 &&
                     (k == 0 || layer_len[k] <= layer_len[k - 1]),
                 "%s: layer %d (length %lld, offset %lld) must be 16-byte "
                 "aligned, behind the %lld nodal values and not longer than "
                 "the layer before it", who, k, (long long)layer_len[k],
                 (long long)layer_off[k], (long long)count);
    ld->len[k] = layer_len[k];
    ld->off[k] = layer_off[k];
    ld->mask[k] = masks && mask_off && mask_off[k] >= 0 ? masks + mask_off[k]
                                                        : nullptr;
  }
  for (int k = num_layers; k < SFEM_MAX_LAYERS; ++k) ld->mask[k] = nullptr;
  return SFEM_OK;
}

template <typename T>
static void launch_update_r_layered(int fuse_rr, bool nt, int grid,
                                    hipStream_t st, T* rr, const T* aa,
                                    int64_t count, const LayerDesc& ld,
                                    double* scalars,
                                    double* rr_partials = nullptr) {
#define SFEM_UPDATE_RL(FUSE, NTV, STR)                                        \
  hipLaunchKernelGGL((cg_update_r_layered_kernel<T, FUSE, NTV, STR>),         \
                     dim3(grid), dim3(512), 0, st, rr, aa, count, ld, scalars, \
                     rr_partials)
  if (fuse_rr == 2) {
    if (nt) SFEM_UPDATE_RL(true, true, true);
    else SFEM_UPDATE_RL(true, false, true);
  } else if (fuse_rr) {
    if (nt) SFEM_UPDATE_RL(true, true, false);
    else SFEM_UPDATE_RL(true, false, false);
  } else {
    if (nt) SFEM_UPDATE_RL(false, true, false);
    else SFEM_UPDATE_RL(false, false, false);
  }
#undef SFEM_UPDATE_RL
}

template <typename T>
static void launch_update_r(int fuse_rr, bool nt, int grid, hipStream_t stream,
                            T* r, const T* ap, int64_t count, double* scalars) {
#define SFEM_UPDATE_R(FUSE, NTV)                                             \
  hipLaunchKernelGGL((cg_update_r_kernel<T, FUSE, NTV>), dim3(grid),         \
                     dim3(512), 0, stream, r, ap, count, scalars)
  if (fuse_rr == 2) {
    if (nt)
      hipLaunchKernelGGL((cg_update_r_kernel<T, true, true, true>), dim3(grid),
                         dim3(512), 0, stream, r, ap, count, scalars);
    else
      hipLaunchKernelGGL((cg_update_r_kernel<T, true, false, true>), dim3(grid),
                         dim3(512), 0, stream, r, ap, count, scalars);
  } else if (fuse_rr) {
    if (nt) SFEM_UPDATE_R(true, true);
    else SFEM_UPDATE_R(true, false);
  } else {
    if (nt) SFEM_UPDATE_R(false, true);
    else SFEM_UPDATE_R(false, false);
  }
#undef SFEM_UPDATE_R
}

// One-thread bookkeeping between the vector kernels of an iteration.
// phase 2 (init, after b.b -> [5] and gamma0 -> [0]):
//     atol2 = max(tol^2 b.b, atol^2); clear pAp, iterations;
//     done <- !(gamma0 > atol2) or maxiter <= 0          (cg.py:65-73)
// phase 0 (p.Ap accumulated in [1]): alpha = gamma / pAp; clear gamma_new
// phase 3: p.Ap <- sum of the partial sums an operator left in `partials`;
// phase 4 = phase 3 + phase 0 in one launch (no all-reduce in between)
// phase 5 / 6: see the kernel (one scalar launch per iteration)
// phase 1 (gamma_new accumulated in [2]): beta = gamma_new / gamma;
//     gamma <- gamma_new; clear pAp; ++iterations;
//     done <- !(gamma > atol2) or iterations >= maxiter   (cg.py:68-73)
// Breakdown guards (scalars[10], SFEM_CG_STATUS_*): the reference's stop rule
// `!(gamma > atol2)` (cg.py:68-73) reads a negative or NaN r.Mr as converged
// and divides by any p.Ap.  Both happen for real in the reference-convention
// partitioned solve, whose r.QQ^T r is a sum over ranks of terms that only
// cancel across ranks (-1e-19 |b|^2 is typical at tight tolerances): the solve
// then stops and says why instead of returning a wrong answer as converged.
__device__ __forceinline__ bool cg_bad_gamma(double g) {
  return !(g >= 0.0) || !(g <= 1.7976931348623157e308);
}
// p.Ap: any finite non-zero value is divided by, as in the reference
// (cg.py:78-79; a negative definite A converges like its negative); zero and
// non-finite values stop the solve.
__device__ __forceinline__ bool cg_bad_pap(double v) {
  return v == 0.0 || !(v >= -1.7976931348623157e308) ||
         !(v <= 1.7976931348623157e308);
}
// closes an iteration: beta, gamma <- gamma_new, counter, stop test
__device__ __forceinline__ void cg_close_iteration(double* scalars,
                                                   double maxiter) {
  double g = scalars[2];
  for (int q = 0; q < SFEM_CG_RR_SLOTS; ++q) {     // striped r.r (fuse_rr = 2)
    g += scalars[SFEM_CG_NSCALARS_NAMED + q];
    scalars[SFEM_CG_NSCALARS_NAMED + q] = 0.0;
  }
  g += scalars[11];       // r.z - r.r of a fused preconditioner (update_xp_mean)
  scalars[11] = 0.0;
  scalars[4] = g / scalars[0];
  scalars[0] = g;
  scalars[8] += 1.0;
  if (cg_bad_gamma(g)) {
    scalars[10] = SFEM_CG_STATUS_BAD_GAMMA;
    scalars[7] = 1.0;
  } else if (!(g > scalars[6])) {
    scalars[10] = SFEM_CG_STATUS_CONVERGED;
    scalars[7] = 1.0;
  } else if (scalars[8] >= maxiter) {
    scalars[10] = SFEM_CG_STATUS_MAXITER;
    scalars[7] = 1.0;
  }
}
// alpha = gamma / p.Ap, or stop (before this iteration's updates) when p.Ap
// is zero or not finite
__device__ __forceinline__ void cg_set_alpha(double* scalars, double pap) {
  if (cg_bad_pap(pap)) {
    scalars[10] = SFEM_CG_STATUS_BAD_PAP;
    scalars[7] = 1.0;
    return;
  }
  scalars[3] = scalars[0] / pap;
  scalars[2] = 0.0;
}

// sum of partials[tid], partials[tid + bd], ... in that order, eight loads in
// flight at a time (one workgroup sums tens of thousands of stored values: a
// load per dependent add would cost a memory round trip each)
__device__ __forceinline__ double strided_sum(const double* __restrict__ partials,
                                              int64_t n) {
  const int64_t bd = blockDim.x;
  double v = 0.0;
  for (int64_t q = threadIdx.x; q < n; q += 8 * bd) {
    double t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      t[u] = q + u * bd < n ? partials[q + u * bd] : 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) v += t[u];
  }
  return v;
}

// Stage one of a long fixed-order sum: workgroup g adds up its contiguous
// chunk of `partials` and stores the result at scratch[g].
__global__ void __launch_bounds__(256)
fold_partials_kernel(const double* __restrict__ partials, int64_t n,
                     int64_t chunk, double* __restrict__ scratch) {
  const int64_t lo = (int64_t)blockIdx.x * chunk;
  const int64_t len = n - lo < chunk ? n - lo : chunk;
  const double total = block_sum(len > 0 ? strided_sum(partials + lo, len)
                                         : 0.0);
  if (threadIdx.x == 0) scratch[blockIdx.x] = total;
}

__global__ void __launch_bounds__(1024)
cg_scalar_kernel(double* scalars, int phase, double maxiter, double tol,
                 double atol, double* partials, int64_t num_partials,
                 bool stored) {
  const int tid = threadIdx.x;
  if (phase == 8) {
    // gamma_new <- stored per-workgroup sums of r.r, in index order
    const double total = block_sum(strided_sum(partials, num_partials));
    if (tid == 0 && scalars[7] == 0.0) scalars[2] = total;
    return;
  }
  if (phase == 7) {
    // fold the striped r.r into the named slot NOW: the caller is about to
    // correct / all-reduce gamma_new (partitioned solves)
    if (tid == 0 && scalars[7] == 0.0) {
      double g = scalars[2];
      for (int q = 0; q < SFEM_CG_RR_SLOTS; ++q) {
        g += scalars[SFEM_CG_NSCALARS_NAMED + q];
        scalars[SFEM_CG_NSCALARS_NAMED + q] = 0.0;
      }
      scalars[2] = g;
    }
    return;
  }
  if (phase == 5 || phase == 6) {
    // Deferred bookkeeping: phase 5 sits between the apply and the updates
    // of iteration k+1 and first closes iteration k (what phase 1 does), so
    // an iteration needs one scalar launch.  The convergence flag is then
    // raised one apply late -- that apply only writes scratch -- and phase 6
    // closes the open iteration when the host wants to look ([9] = open).
    double total = 0.0;
    if (phase == 5) {
      double v;
      if (stored) {
        v = strided_sum(partials, num_partials);
      } else {
        v = 0.0;
        for (int64_t q = tid; q < num_partials; q += blockDim.x) {
          v += partials[q];
          partials[q] = 0.0;        // the next apply accumulates again
        }
      }
      total = block_sum(v);
    }
    if (tid != 0 || scalars[7] != 0.0) return;
    if (scalars[9] != 0.0) {
      scalars[9] = 0.0;
      cg_close_iteration(scalars, maxiter);
      if (scalars[7] != 0.0) return;
    }
    if (phase == 5) {
      scalars[1] = total;
      cg_set_alpha(scalars, total);
      if (scalars[7] == 0.0) scalars[9] = 1.0;
    }
    return;
  }
  if (phase == 3 || phase == 4) {   // p.Ap <- sum of the fused partial sums
    double v = 0.0;
    for (int64_t q = tid; q < num_partials; q += blockDim.x) v += partials[q];
    const double total = block_sum(v);
    if (tid == 0 && scalars[7] == 0.0) {
      scalars[1] = total;
      if (phase == 4) cg_set_alpha(scalars, total);   // ... and phase 0
    }
    return;
  }
  if (partials && !stored && (phase == 1 || phase == 2))
    for (int q = tid; q < SFEM_DOT_SLOTS; q += blockDim.x) partials[q] = 0.0;
  if (tid != 0) return;
  if (phase == 2) {
    const double a = tol * tol * scalars[5], b = atol * atol;
    scalars[6] = a > b ? a : b;
    scalars[1] = 0.0;
    scalars[2] = 0.0;
    scalars[8] = 0.0;
    scalars[9] = 0.0;
    scalars[11] = 0.0;
    scalars[12] = 0.0;
    for (int q = 0; q < SFEM_CG_RR_SLOTS; ++q)
      scalars[SFEM_CG_NSCALARS_NAMED + q] = 0.0;
    scalars[10] = SFEM_CG_STATUS_RUNNING;
    scalars[7] = 0.0;
    if (cg_bad_gamma(scalars[0])) {
      scalars[10] = SFEM_CG_STATUS_BAD_GAMMA;
      scalars[7] = 1.0;
    } else if (!(scalars[0] > scalars[6])) {
      scalars[10] = SFEM_CG_STATUS_CONVERGED;
      scalars[7] = 1.0;
    } else if (maxiter <= 0.0) {
      scalars[10] = SFEM_CG_STATUS_MAXITER;
      scalars[7] = 1.0;
    }
    return;
  }
  if (scalars[7] != 0.0) return;
  if (phase == 0) {
    cg_set_alpha(scalars, scalars[1]);
  } else {
    scalars[1] = 0.0;
    cg_close_iteration(scalars, maxiter);
  }
}

template <typename T>
__global__ void __launch_bounds__(512)
axpby_kernel(T a, const T* __restrict__ x, T b, T* __restrict__ y,
             int64_t count) {
  using V = typename Vec16<T>::type;
  constexpr int VN = Vec16<T>::N;
  const int64_t nvec = count / VN;
  const V* xv = reinterpret_cast<const V*>(x);
  V* yv = reinterpret_cast<V*>(y);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const bool use_y = b != T(0);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += stride) {
    const V xx = xv[i];
    V yy;
    if (use_y) yy = yv[i];
#pragma unroll
    for (int c = 0; c < VN; ++c) {
      T* ye = reinterpret_cast<T*>(&yy) + c;
      *ye = a * vget<T>(xx, c) + (use_y ? b * *ye : T(0));
    }
    yv[i] = yy;
  }
  if (blockIdx.x == 0 && threadIdx.x < count - nvec * VN) {
    const int64_t i = nvec * VN + threadIdx.x;
    y[i] = a * x[i] + (use_y ? b * y[i] : T(0));
  }
}

}  // namespace sfem

using namespace sfem;

#define DISPATCH_DTYPE(dtype, ...)                      \
  if ((dtype) == SFEM_F64) {                            \
    using T = double;                                   \
    __VA_ARGS__;                                        \
  } else if ((dtype) == SFEM_F32) {                     \
    using T = float;                                    \
    __VA_ARGS__;                                        \
  } else {                                              \
    set_error("unknown dtype %d", (int)(dtype));        \
    return SFEM_EINVAL;                                 \
  }

extern "C" {

int sfem_abi_version(void) { return SFEM_ABI_VERSION; }
const char* sfem_last_error(void) { return g_error; }

int sfem_gather(const void* u, const int32_t* indices, void* out,
                int64_t count, double fill, int dtype, sfem_stream_t stream) {
  SFEM_REQUIRE(count >= 0, "sfem_gather: negative count");
  if (count == 0) return SFEM_OK;
  SFEM_REQUIRE(u && indices && out, "sfem_gather: null pointer");
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(
      gather_kernel<T>, dim3(stream_grid(count, 256)), dim3(256), 0,
      as_stream(stream), (const T*)u, indices, (T*)out, count, (T)fill));
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_gather_rows(const void* x, const int32_t* indices, void* out,
                     int64_t count, int ncomp, int dtype,
                     sfem_stream_t stream) {
  SFEM_REQUIRE(count >= 0 && ncomp >= 1, "sfem_gather_rows: bad sizes");
  if (count == 0) return SFEM_OK;
  SFEM_REQUIRE(x && indices && out, "sfem_gather_rows: null pointer");
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(
      gather_rows_kernel<T>, dim3(stream_grid(count * ncomp, 256)), dim3(256),
      0, as_stream(stream), (const T*)x, indices, (T*)out, count, ncomp));
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_scatter_add(const void* u_local, const int32_t* indices, void* out,
                     int64_t count, int64_t num_nodes, int ncomp, int dtype,
                     sfem_stream_t stream) {
  SFEM_REQUIRE(count >= 0 && num_nodes >= 0 && ncomp >= 1,
               "sfem_scatter_add: bad sizes");
  SFEM_REQUIRE(dtype == SFEM_F32 || dtype == SFEM_F64,
               "sfem_scatter_add: unknown dtype %d", dtype);
  if (num_nodes == 0) return SFEM_OK;
  SFEM_REQUIRE(out, "sfem_scatter_add: null out");
  const size_t esz = dtype == SFEM_F64 ? 8 : 4;
  SFEM_HIP(hipMemsetAsync(out, 0, (size_t)num_nodes * ncomp * esz,
                          as_stream(stream)));
  if (count == 0) return SFEM_OK;
  SFEM_REQUIRE(u_local && indices, "sfem_scatter_add: null pointer");
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(
      scatter_add_kernel<T>, dim3(stream_grid(count * ncomp, 256)), dim3(256),
      0, as_stream(stream), (const T*)u_local, indices, (T*)out, count, ncomp));
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_scatter_csr(const void* u_local, const int64_t* offsets,
                     const int32_t* slots, void* out, int64_t num_nodes,
                     int ncomp, int dtype, sfem_stream_t stream) {
  SFEM_REQUIRE(num_nodes >= 0 && ncomp >= 1, "sfem_scatter_csr: bad sizes");
  if (num_nodes == 0) return SFEM_OK;
  SFEM_REQUIRE(u_local && offsets && slots && out,
               "sfem_scatter_csr: null pointer");
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(
      scatter_csr_kernel<T>, dim3(stream_grid(num_nodes * ncomp, 256)),
      dim3(256), 0, as_stream(stream), (const T*)u_local, offsets, slots,
      (T*)out, num_nodes, ncomp));
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_exchange_local(const void* u, void* out, const int32_t* gidx,
                        const int32_t* unique, int64_t count,
                        int64_t num_nodes, void* sums, int64_t num_unique,
                        int ncomp, int dtype, sfem_stream_t stream) {
  SFEM_REQUIRE(count >= 0 && num_nodes >= 0 && num_unique >= 0 && ncomp >= 1,
               "sfem_exchange_local: bad sizes");
  SFEM_REQUIRE(dtype == SFEM_F32 || dtype == SFEM_F64,
               "sfem_exchange_local: unknown dtype %d", dtype);
  if (num_nodes == 0) return SFEM_OK;
  SFEM_REQUIRE(u && out, "sfem_exchange_local: null pointer");
  const size_t esz = dtype == SFEM_F64 ? 8 : 4;
  if (out != u)
    SFEM_HIP(hipMemcpyAsync(out, u, (size_t)num_nodes * ncomp * esz,
                            hipMemcpyDeviceToDevice, as_stream(stream)));
  if (count == 0) return SFEM_OK;
  SFEM_REQUIRE(gidx && unique && sums, "sfem_exchange_local: null pointer");
  SFEM_HIP(hipMemsetAsync(sums, 0, (size_t)num_unique * ncomp * esz,
                          as_stream(stream)));
  const unsigned grid = stream_grid(count * ncomp, 256);
  DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL(exchange_sum_kernel<T>, dim3(grid), dim3(256), 0,
                       as_stream(stream), (const T*)u, gidx, unique, (T*)sums,
                       count, ncomp);
    hipLaunchKernelGGL(exchange_expand_kernel<T>, dim3(grid), dim3(256), 0,
                       as_stream(stream), (const T*)sums, gidx, unique,
                       (T*)out, count, ncomp);
  });
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_exchange_classes(void* u, const int32_t* members,
                          const int32_t* offsets, int64_t num_classes,
                          int ncomp, int64_t node_stride, int64_t comp_stride,
                          int dtype, sfem_stream_t stream) {
  SFEM_REQUIRE(num_classes >= 0 && ncomp >= 1 && node_stride >= 1 &&
               comp_stride >= 1, "sfem_exchange_classes: bad sizes");
  SFEM_REQUIRE(dtype == SFEM_F32 || dtype == SFEM_F64,
               "sfem_exchange_classes: unknown dtype %d", dtype);
  if (num_classes == 0) return SFEM_OK;
  SFEM_REQUIRE(u && members && offsets, "sfem_exchange_classes: null pointer");
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(
      exchange_classes_kernel<T>, dim3(stream_grid(num_classes * ncomp, 256)),
      dim3(256), 0, as_stream(stream), (T*)u, members, offsets, num_classes,
      ncomp, node_stride, comp_stride));
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_zero_strips(void* base, int64_t strip_len, int64_t strip_stride,
                     int nstrips, int dtype, sfem_stream_t stream) {
  SFEM_REQUIRE(strip_len >= 0 && nstrips >= 0 && strip_stride >= 0,
               "sfem_zero_strips: bad sizes");
  SFEM_REQUIRE(dtype == SFEM_F32 || dtype == SFEM_F64,
               "sfem_zero_strips: unknown dtype %d", dtype);
  if (strip_len == 0 || nstrips == 0) return SFEM_OK;
  SFEM_REQUIRE(base, "sfem_zero_strips: null pointer");
  const size_t esz = dtype == SFEM_F64 ? 8 : 4;
  // large ranges: the runtime's fill runs at stream bandwidth; short ones are
  // bound by the number of launches, so all strips go into one
  // (measured, 3 strips: 34 MB 16.4 vs 20.6 us, 64 MB 27 vs 31, 128 MB 73 vs 65)
  if ((size_t)strip_len * esz > ((size_t)64 << 20)) {
    for (int s = 0; s < nstrips; ++s)
      SFEM_HIP(hipMemsetAsync((char*)base + (size_t)s * strip_stride * esz, 0,
                              (size_t)strip_len * esz, as_stream(stream)));
    return SFEM_OK;
  }
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(
      zero_strips_kernel<T>, dim3(stream_grid(strip_len * nstrips, 256)),
      dim3(256), 0, as_stream(stream), (T*)base, strip_len, strip_stride,
      nstrips));
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_subtract_weighted_mean(const void* w, const void* b, double total,
                                void* out, double* partials, int64_t count,
                                double* dot_result, int dtype,
                                sfem_stream_t stream) {
  SFEM_REQUIRE(count >= 0, "sfem_subtract_weighted_mean: bad count");
  SFEM_REQUIRE(dtype == SFEM_F32 || dtype == SFEM_F64,
               "sfem_subtract_weighted_mean: unknown dtype %d", dtype);
  SFEM_REQUIRE(total != 0.0, "sfem_subtract_weighted_mean: total is zero");
  if (count == 0) return SFEM_OK;
  SFEM_REQUIRE(w && b && out && partials,
               "sfem_subtract_weighted_mean: null pointer");
  int64_t nblk = (count + 512 * 4 - 1) / (512 * 4);
  if (nblk > SFEM_DOT_SLOTS) nblk = SFEM_DOT_SLOTS;
  DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL(dot_partials_kernel<T>, dim3((unsigned)nblk), dim3(512),
                       0, as_stream(stream), (const T*)w, (const T*)b, count,
                       partials);
    hipLaunchKernelGGL(subtract_mean_kernel<T>,
                       dim3(dot_result ? reduce_grid(count, 512 * 4)
                                       : stream_grid(count, 512)),
                       dim3(512), 0, as_stream(stream), (const T*)w, partials,
                       (int)nblk, 1.0 / total, (T*)out, count, dot_result);
  });
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_pack(const void* u, const int32_t* idx, void* buf, int64_t count,
              int ncomp, int dtype, sfem_stream_t stream) {
  return sfem_gather_rows(u, idx, buf, count, ncomp, dtype, stream);
}

int sfem_unpack_add(const void* buf, const int32_t* idx, void* u,
                    int64_t count, int ncomp, int dtype,
                    sfem_stream_t stream) {
  SFEM_REQUIRE(count >= 0 && ncomp >= 1, "sfem_unpack_add: bad sizes");
  if (count == 0) return SFEM_OK;
  SFEM_REQUIRE(buf && idx && u, "sfem_unpack_add: null pointer");
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(
      unpack_add_kernel<T>, dim3(stream_grid(count * ncomp, 256)), dim3(256),
      0, as_stream(stream), (const T*)buf, idx, (T*)u, count, ncomp));
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_pack_strided(const void* u, const int32_t* idx, void* buf,
                      int64_t count, int ncomp, int64_t node_stride,
                      int64_t comp_stride, int dtype, sfem_stream_t stream) {
  SFEM_REQUIRE(count >= 0 && ncomp >= 1, "sfem_pack_strided: bad sizes");
  if (count == 0) return SFEM_OK;
  SFEM_REQUIRE(u && idx && buf, "sfem_pack_strided: null pointer");
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(
      pack_strided_kernel<T>, dim3(stream_grid(count * ncomp, 256)), dim3(256),
      0, as_stream(stream), (const T*)u, idx, (T*)buf, count, ncomp,
      node_stride, comp_stride));
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_unpack_add_atomic(const void* buf, const int32_t* idx, void* u,
                           int64_t count, int ncomp, int64_t node_stride,
                           int64_t comp_stride, int dtype,
                           sfem_stream_t stream) {
  SFEM_REQUIRE(count >= 0 && ncomp >= 1, "sfem_unpack_add_atomic: bad sizes");
  if (count == 0) return SFEM_OK;
  SFEM_REQUIRE(buf && idx && u, "sfem_unpack_add_atomic: null pointer");
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(
      unpack_add_atomic_kernel<T>, dim3(stream_grid(count * ncomp, 256)),
      dim3(256), 0, as_stream(stream), (const T*)buf, idx, (T*)u, count, ncomp,
      node_stride, comp_stride));
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_dot(const void* a, const void* b, int64_t count, double* result,
             int dtype, sfem_stream_t stream) {
  SFEM_REQUIRE(count >= 0 && result, "sfem_dot: bad arguments");
  SFEM_HIP(hipMemsetAsync(result, 0, sizeof(double), as_stream(stream)));
  if (count == 0) return SFEM_OK;
  SFEM_REQUIRE(a && b, "sfem_dot: null pointer");
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(
      dot_kernel<T>, dim3(reduce_grid(count, 512 * 4)), dim3(512), 0,
      as_stream(stream), (const T*)a, (const T*)b, count, result));
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_dot_accumulate(const void* a, const void* b, int64_t count,
                        double* result, int dtype, sfem_stream_t stream) {
  SFEM_REQUIRE(count >= 0 && result, "sfem_dot_accumulate: bad arguments");
  if (count == 0) return SFEM_OK;
  SFEM_REQUIRE(a && b, "sfem_dot_accumulate: null pointer");
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(
      dot_kernel<T>, dim3(reduce_grid(count, 512 * 4)), dim3(512), 0,
      as_stream(stream), (const T*)a, (const T*)b, count, result));
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_dot_indexed(const void* a, const void* b, const int64_t* idx,
                     const double* w, int64_t count, int ncomp,
                     int64_t node_stride, int64_t comp_stride, double scale,
                     double* result, int dtype, sfem_stream_t stream) {
  SFEM_REQUIRE(count >= 0 && ncomp >= 1 && result,
               "sfem_dot_indexed: bad arguments");
  if (count == 0) return SFEM_OK;
  SFEM_REQUIRE(a && b && idx && w, "sfem_dot_indexed: null pointer");
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(
      dot_indexed_kernel<T>, dim3(reduce_grid(count, 256)), dim3(256), 0,
      as_stream(stream), (const T*)a, (const T*)b, idx, w, count, ncomp,
      node_stride, comp_stride, scale, result));
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_cg_update_xr(void* x, void* r, const void* p, const void* ap,
                      int64_t count, double* scalars, int fuse_rr, int dtype,
                      sfem_stream_t stream) {
  SFEM_REQUIRE(count >= 0 && scalars, "sfem_cg_update_xr: bad arguments");
  if (count == 0) return SFEM_OK;
  SFEM_REQUIRE(x && r && p && ap, "sfem_cg_update_xr: null pointer");
  const unsigned grid = reduce_grid(count, 512 * 2);
  DISPATCH_DTYPE(dtype, {
    if (fuse_rr)
      hipLaunchKernelGGL((cg_update_xr_kernel<T, true>), dim3(grid), dim3(512),
                         0, as_stream(stream), (T*)x, (T*)r, (const T*)p,
                         (const T*)ap, count, scalars);
    else
      hipLaunchKernelGGL((cg_update_xr_kernel<T, false>), dim3(grid),
                         dim3(512), 0, as_stream(stream), (T*)x, (T*)r,
                         (const T*)p, (const T*)ap, count, scalars);
  });
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_cg_update_p(void* p, const void* z, int64_t count, double* scalars,
                     int dtype, sfem_stream_t stream) {
  SFEM_REQUIRE(count >= 0 && scalars, "sfem_cg_update_p: bad arguments");
  if (count == 0) return SFEM_OK;
  SFEM_REQUIRE(p && z, "sfem_cg_update_p: null pointer");
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(
      cg_update_p_kernel<T>, dim3(stream_grid(count, 512 * 2)), dim3(512), 0,
      as_stream(stream), (T*)p, (const T*)z, count, scalars));
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_cg_update_r(void* r, const void* ap, int64_t count, double* scalars,
                     int fuse_rr, int dtype, sfem_stream_t stream) {
  SFEM_REQUIRE(count >= 0 && scalars, "sfem_cg_update_r: bad arguments");
  if (count == 0) return SFEM_OK;
  SFEM_REQUIRE(r && ap, "sfem_cg_update_r: null pointer");
  SFEM_REQUIRE(fuse_rr >= 0 && fuse_rr <= 2, "sfem_cg_update_r: bad fuse_rr");
  DISPATCH_DTYPE(dtype, {
    // fuse_rr = 2: the per-workgroup sums are spread over SFEM_CG_RR_SLOTS
    // addresses, so the update can stream with 128 workgroups per CU
    const int grid = fuse_rr == 2 ? stream_grid(count, 512 * 2)
                                  : reduce_grid(count, 512 * 2);
    launch_update_r<T>(fuse_rr, streams_past_caches(count, sizeof(T)), grid,
                       as_stream(stream), (T*)r, (const T*)ap, count, scalars);
  });
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_cg_update_r_layered(void* r, const void* ap_ext, int64_t count,
                             const int64_t* layer_len,
                             const int64_t* layer_off, int num_layers,
                             const uint8_t* layer_masks,
                             const int64_t* mask_off, double* scalars,
                             int fuse_rr, int dtype, sfem_stream_t stream) {
  SFEM_REQUIRE(count >= 0 && scalars, "sfem_cg_update_r_layered: bad arguments");
  if (count == 0) return SFEM_OK;
  SFEM_REQUIRE(r && ap_ext, "sfem_cg_update_r_layered: null pointer");
  SFEM_REQUIRE(fuse_rr >= 0 && fuse_rr <= 2,
               "sfem_cg_update_r_layered: bad fuse_rr");
  SFEM_REQUIRE(dtype == SFEM_F32 || dtype == SFEM_F64,
               "sfem_cg_update_r_layered: unknown dtype %d", dtype);
  LayerDesc ld;
  const int rc = make_layer_desc("sfem_cg_update_r_layered", layer_len,
                                 layer_off, num_layers, count,
                                 dtype == SFEM_F64 ? 2 : 4, &ld, layer_masks,
                                 mask_off);
  if (rc != SFEM_OK) return rc;
  DISPATCH_DTYPE(dtype, {
    const int grid = fuse_rr == 2 ? stream_grid(count, 512 * 2)
                                  : reduce_grid(count, 512 * 2);
    launch_update_r_layered<T>(fuse_rr, streams_past_caches(count, sizeof(T)),
                               grid, as_stream(stream), (T*)r,
                               (const T*)ap_ext, count, ld, scalars);
  });
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_cg_update_r_layered_det(void* r, const void* ap_ext, int64_t count,
                                 const int64_t* layer_len,
                                 const int64_t* layer_off, int num_layers,
                                 const uint8_t* layer_masks,
                                 const int64_t* mask_off,
                                 double* scalars, double* rr_partials,
                                 int64_t rr_capacity, int64_t* num_rr,
                                 int dtype, sfem_stream_t stream) {
  SFEM_REQUIRE(count >= 0 && scalars && rr_partials && num_rr &&
                   rr_capacity >= 1,
               "sfem_cg_update_r_layered_det: bad arguments");
  *num_rr = 0;
  if (count == 0) return SFEM_OK;
  SFEM_REQUIRE(r && ap_ext, "sfem_cg_update_r_layered_det: null pointer");
  SFEM_REQUIRE(dtype == SFEM_F32 || dtype == SFEM_F64,
               "sfem_cg_update_r_layered_det: unknown dtype %d", dtype);
  LayerDesc ld;
  const int rc = make_layer_desc("sfem_cg_update_r_layered_det", layer_len,
                                 layer_off, num_layers, count,
                                 dtype == SFEM_F64 ? 2 : 4, &ld, layer_masks,
                                 mask_off);
  if (rc != SFEM_OK) return rc;
  DISPATCH_DTYPE(dtype, {
    int grid = stream_grid(count, 512 * 2);
    if (grid > rr_capacity) grid = (int)rr_capacity;
    *num_rr = grid;
    launch_update_r_layered<T>(1, streams_past_caches(count, sizeof(T)), grid,
                               as_stream(stream), (T*)r, (const T*)ap_ext,
                               count, ld, scalars, rr_partials);
  });
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_fold_layers_at(void* out_ext, const int64_t* idx, int64_t num_idx,
                        int64_t count, const int64_t* layer_len,
                        const int64_t* layer_off, int num_layers, int dtype,
                        sfem_stream_t stream) {
  SFEM_REQUIRE(count >= 0 && num_idx >= 0, "sfem_fold_layers_at: bad sizes");
  if (num_idx == 0 || num_layers == 0) return SFEM_OK;
  SFEM_REQUIRE(out_ext && idx, "sfem_fold_layers_at: null pointer");
  SFEM_REQUIRE(dtype == SFEM_F32 || dtype == SFEM_F64,
               "sfem_fold_layers_at: unknown dtype %d", dtype);
  LayerDesc ld;
  const int rc = make_layer_desc("sfem_fold_layers_at", layer_len, layer_off,
                                 num_layers, count, dtype == SFEM_F64 ? 2 : 4,
                                 &ld);
  if (rc != SFEM_OK) return rc;
  DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL((fold_layers_at_kernel<T>),
                       dim3(stream_grid(num_idx, 256)), dim3(256), 0,
                       as_stream(stream), (T*)out_ext, idx, num_idx, ld);
  });
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_fold_layers(void* out_ext, int64_t count, const int64_t* layer_len,
                     const int64_t* layer_off, int num_layers, int dtype,
                     sfem_stream_t stream) {
  SFEM_REQUIRE(count >= 0, "sfem_fold_layers: negative count");
  if (count == 0 || num_layers == 0) return SFEM_OK;
  SFEM_REQUIRE(out_ext, "sfem_fold_layers: null pointer");
  SFEM_REQUIRE(dtype == SFEM_F32 || dtype == SFEM_F64,
               "sfem_fold_layers: unknown dtype %d", dtype);
  LayerDesc ld;
  const int rc = make_layer_desc("sfem_fold_layers", layer_len, layer_off,
                                 num_layers, count, dtype == SFEM_F64 ? 2 : 4,
                                 &ld);
  if (rc != SFEM_OK) return rc;
  DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL((fold_layers_kernel<T>),
                       dim3(stream_grid(ld.len[0], 512 * 2)), dim3(512), 0,
                       as_stream(stream), (T*)out_ext, count, ld);
  });
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_cg_update_xp(void* x, void* p, const void* z, int64_t count,
                      double* scalars, int dtype, sfem_stream_t stream) {
  SFEM_REQUIRE(count >= 0 && scalars, "sfem_cg_update_xp: bad arguments");
  if (count == 0) return SFEM_OK;
  SFEM_REQUIRE(x && p && z, "sfem_cg_update_xp: null pointer");
  DISPATCH_DTYPE(dtype, {
    if (streams_past_caches(count, sizeof(T)))
      hipLaunchKernelGGL((cg_update_xp_kernel<T, true>),
                         dim3(stream_grid(count, 512 * 2)), dim3(512), 0,
                         as_stream(stream), (T*)x, (T*)p, (const T*)z, count,
                         scalars);
    else
      hipLaunchKernelGGL((cg_update_xp_kernel<T, false>),
                         dim3(stream_grid(count, 512 * 2)), dim3(512), 0,
                         as_stream(stream), (T*)x, (T*)p, (const T*)z, count,
                         scalars);
  });
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_cg_update_xp_lazy(void* x, void* pring, int64_t ring_stride,
                           const void* z, int64_t count, double* scalars,
                           double* lazy, int m, int dtype,
                           sfem_stream_t stream) {
  SFEM_REQUIRE(count >= 0 && scalars && lazy,
               "sfem_cg_update_xp_lazy: bad arguments");
  SFEM_REQUIRE(m >= 2 && m <= SFEM_CG_LAZY_MAX,
               "sfem_cg_update_xp_lazy: ring of %d directions (2..%d)", m,
               SFEM_CG_LAZY_MAX);
  if (count == 0) return SFEM_OK;
  SFEM_REQUIRE(x && pring && z && ring_stride >= count,
               "sfem_cg_update_xp_lazy: null pointer or ring stride < count");
  SFEM_REQUIRE(dtype == SFEM_F32 || dtype == SFEM_F64,
               "sfem_cg_update_xp_lazy: unknown dtype %d", dtype);
  SFEM_REQUIRE(ring_stride % (dtype == SFEM_F64 ? 2 : 4) == 0,
               "sfem_cg_update_xp_lazy: ring stride must keep 16-byte "
               "alignment");
  DISPATCH_DTYPE(dtype, {
    const int grid = stream_grid(count, 512 * 2);
    if (streams_past_caches(count, sizeof(T)))
      hipLaunchKernelGGL((cg_update_xp_lazy_kernel<T, true>), dim3(grid),
                         dim3(512), 0, as_stream(stream), (T*)x, (T*)pring,
                         ring_stride, (const T*)z, count, scalars, lazy, m);
    else
      hipLaunchKernelGGL((cg_update_xp_lazy_kernel<T, false>), dim3(grid),
                         dim3(512), 0, as_stream(stream), (T*)x, (T*)pring,
                         ring_stride, (const T*)z, count, scalars, lazy, m);
  });
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_cg_flush_x(void* x, const void* pring, int64_t ring_stride,
                    int64_t count, const double* scalars, double* lazy, int m,
                    int dtype, sfem_stream_t stream) {
  SFEM_REQUIRE(count >= 0 && scalars && lazy && m >= 2 &&
                   m <= SFEM_CG_LAZY_MAX,
               "sfem_cg_flush_x: bad arguments");
  SFEM_REQUIRE(dtype == SFEM_F32 || dtype == SFEM_F64,
               "sfem_cg_flush_x: unknown dtype %d", dtype);
  if (count > 0) {
    SFEM_REQUIRE(x && pring && ring_stride >= count,
                 "sfem_cg_flush_x: null pointer or ring stride < count");
    DISPATCH_DTYPE(dtype, {
      hipLaunchKernelGGL((cg_flush_x_kernel<T>),
                         dim3(stream_grid(count, 512)), dim3(512), 0,
                         as_stream(stream), (T*)x, (const T*)pring,
                         ring_stride, count, scalars, lazy, m);
    });
  }
  hipLaunchKernelGGL(cg_lazy_mark_kernel, dim3(1), dim3(1), 0,
                     as_stream(stream), scalars, lazy);
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_cg_update_r_mean(void* r, const void* ap, const void* w,
                          int64_t count, double* scalars, double* sums,
                          int dtype, sfem_stream_t stream) {
  SFEM_REQUIRE(count >= 0 && scalars && sums,
               "sfem_cg_update_r_mean: bad arguments");
  if (count == 0) return SFEM_OK;
  SFEM_REQUIRE(r && ap && w, "sfem_cg_update_r_mean: null pointer");
  DISPATCH_DTYPE(dtype, {
    const int grid = stream_grid(count, 512 * 2);
    if (streams_past_caches(count, sizeof(T)))
      hipLaunchKernelGGL((cg_update_r_mean_kernel<T, true>), dim3(grid),
                         dim3(512), 0, as_stream(stream), (T*)r, (const T*)ap,
                         (const T*)w, count, scalars, sums);
    else
      hipLaunchKernelGGL((cg_update_r_mean_kernel<T, false>), dim3(grid),
                         dim3(512), 0, as_stream(stream), (T*)r, (const T*)ap,
                         (const T*)w, count, scalars, sums);
  });
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_cg_update_xp_mean(void* x, void* p, const void* r, int64_t count,
                           double* scalars, double* sums, double total,
                           int dtype, sfem_stream_t stream) {
  SFEM_REQUIRE(count >= 0 && scalars && sums && total != 0.0,
               "sfem_cg_update_xp_mean: bad arguments");
  if (count == 0) return SFEM_OK;
  SFEM_REQUIRE(x && p && r, "sfem_cg_update_xp_mean: null pointer");
  DISPATCH_DTYPE(dtype, {
    const int grid = stream_grid(count, 512 * 2);
    if (streams_past_caches(count, sizeof(T)))
      hipLaunchKernelGGL((cg_update_xp_mean_kernel<T, true>), dim3(grid),
                         dim3(512), 0, as_stream(stream), (T*)x, (T*)p,
                         (const T*)r, count, scalars, sums, total);
    else
      hipLaunchKernelGGL((cg_update_xp_mean_kernel<T, false>), dim3(grid),
                         dim3(512), 0, as_stream(stream), (T*)x, (T*)p,
                         (const T*)r, count, scalars, sums, total);
  });
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_cg_scalars(double* scalars, int phase, double maxiter, double tol,
                    double atol, double* partials, sfem_stream_t stream) {
  SFEM_REQUIRE(scalars && phase >= 0 && phase <= 7 &&
                   (phase < 3 || phase >= 6 || partials),
               "sfem_cg_scalars: bad arguments");
  hipLaunchKernelGGL(cg_scalar_kernel, dim3(1), dim3(256), 0,
                     as_stream(stream), scalars, phase, maxiter, tol, atol,
                     partials, (int64_t)SFEM_DOT_SLOTS, false);
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_cg_scalars_n(double* scalars, int phase, double maxiter, double tol,
                      double atol, double* partials, int64_t num_partials,
                      sfem_stream_t stream) {
  SFEM_REQUIRE(scalars && partials && num_partials >= 1 &&
                   (phase == 3 || phase == 4 || phase == 5 || phase == 8),
               "sfem_cg_scalars_n: phases 3, 4, 5, 8 over stored partial sums");
  // long sums in two fixed-order stages: SFEM_FOLD_GROUPS workgroups reduce
  // contiguous chunks into the scratch behind the partial sums, the scalar
  // kernel adds those up (one workgroup over 786 k values -- p = 11, three
  // waves per element -- took 170 us)
  if (num_partials > 4096) {
    const int64_t chunk = (num_partials + SFEM_FOLD_GROUPS - 1) /
                          SFEM_FOLD_GROUPS;
    hipLaunchKernelGGL(fold_partials_kernel, dim3(SFEM_FOLD_GROUPS), dim3(256),
                       0, as_stream(stream), partials, num_partials, chunk,
                       partials + num_partials);
    partials += num_partials;
    num_partials = SFEM_FOLD_GROUPS;
  }
  hipLaunchKernelGGL(cg_scalar_kernel, dim3(1), dim3(256), 0,
                     as_stream(stream), scalars, phase, maxiter, tol, atol,
                     partials, num_partials, true);
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_axpby(double a, const void* x, double b, void* y, int64_t count,
               int dtype, sfem_stream_t stream) {
  SFEM_REQUIRE(count >= 0, "sfem_axpby: negative count");
  if (count == 0) return SFEM_OK;
  SFEM_REQUIRE(x && y, "sfem_axpby: null pointer");
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(
      axpby_kernel<T>, dim3(stream_grid(count, 512 * 2)), dim3(512), 0,
      as_stream(stream), (T)a, (const T*)x, (T)b, (T*)y, count));
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

}  // extern "C"
