// Tensor-product interpolation between two grids of one element with the
// sizes known at compile time:
//
//   out = (M (x) .. (x) M) (w .* in),     M: NO x NI,  in: NI^D,  out: NO^D
//
// per element and component -- the values-only case of `sfem_basis_eval`
// (M = I, the (q, P) interpolation matrix of core/interpolation.py:260-263)
// and the values-only case of `sfem_basis_eval_t` (M = I^T, w = detJ x
// quadrature weights).  The stepper spends its generic-kernel time exactly
// there: the over-integrated convection term interpolates to and from its
// quadrature grid and the filter interpolates down and up one order
// (navier_stokes.py:238-245, :460-482).  The generic kernels in
// sfem_basis.hip contract with run-time shapes (index arithmetic by division
// per output, two LDS reads per multiply-add): at 64^3 elements, three
// components, 8 -> 10 points 9.6 ms forward / 9.8 ms transposed (here 5.6 /
// 6.7), 8 -> 7 points 6.1 / 6.6 (here 2.8 / 2.8); 2D, 8 x 64^3 elements,
// 9 -> 11 points 15.1 / 16.3 (here 2.9 / 2.9)  [scripts/time_interp.py].
//
// Here: one wave per element, all components of the element in LDS at once,
// one thread per LINE of the axis being contracted: NI values into registers,
// NO results out of them with the matrix read as a wave-wide broadcast.  The
// axes are contracted in the generic kernel's order with the generic kernel's
// summation order, so the results are the same to the last bit.
#ifndef SFEM_INTERP_H_
#define SFEM_INTERP_H_
#include "sfem_common.h"

namespace sfem {

constexpr int ipow(int b, int e) { return e == 0 ? 1 : b * ipow(b, e - 1); }

template <typename T, int D, int NI, int NO>
__global__ void __launch_bounds__(64)
tensor_interp_kernel(const T* __restrict__ in, const T* __restrict__ mat,
                     const T* __restrict__ weight, T* __restrict__ out,
                     int64_t num_elements, int nc, int group, int trans) {
  constexpr int NM = NI > NO ? NI : NO;
  constexpr int CAP = ipow(NM, D);
  constexpr int NIN = ipow(NI, D), NOUT = ipow(NO, D);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* m = reinterpret_cast<T*>(smem_raw);          // [NO][NI]
  T* buf = m + NO * NI;                           // [2][group][CAP]
  const int lane = threadIdx.x;
  const int64_t e = blockIdx.x;
  // interp1 is (q, P): forward M = interp1 (NO = q), transposed M = interp1^T
  for (int t = lane; t < NO * NI; t += 64) {
    const int o = t / NI, i = t - o * NI;
    m[t] = trans ? mat[i * NO + o] : mat[t];
  }
  for (int k0 = 0; k0 < nc; k0 += group) {
    const int g = nc - k0 < group ? nc - k0 : group;
    __syncthreads();
    if (g == nc) {                                // the whole element, in order
      for (int t = lane; t < NIN * nc; t += 64) {
        const int q = t / nc, k = t - q * nc;
        T v = in[e * NIN * nc + t];
        if (weight) v *= weight[e * NIN + q];
        buf[k * CAP + q] = v;
      }
    } else {
      for (int t = lane; t < NIN * g; t += 64) {
        const int q = t / g, k = t - q * g;
        T v = in[(e * NIN + q) * nc + k0 + k];
        if (weight) v *= weight[e * NIN + q];
        buf[k * CAP + q] = v;
      }
    }
    __syncthreads();
    int cur = 0;
#pragma unroll
    for (int a = 0; a < D; ++a) {
      // layout [pre = NO^a][n][post = NI^(D-1-a)]
      constexpr int dummy = 0; (void)dummy;
      const int pre = a == 0 ? 1 : (a == 1 ? NO : NO * NO);
      const int post = (D - 1 - a) == 0 ? 1 : ((D - 1 - a) == 1 ? NI : NI * NI);
      const int lines = pre * post;
      const T* src = buf + cur * group * CAP;
      T* dst = buf + (cur ^ 1) * group * CAP;
      for (int L = lane; L < lines * g; L += 64) {
        const int k = L / lines, l = L - k * lines;
        const int pi = l / post, qi = l - pi * post;
        const T* x0 = src + k * CAP + pi * NI * post + qi;
        T x[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) x[i] = x0[i * post];
        T* y0 = dst + k * CAP + pi * NO * post + qi;
#pragma unroll
        for (int o = 0; o < NO; ++o) {
          T acc = T(0);
#pragma unroll
          for (int i = 0; i < NI; ++i) acc += m[o * NI + i] * x[i];
          y0[o * post] = acc;
        }
      }
      cur ^= 1;
      __syncthreads();
    }
    const T* res = buf + cur * group * CAP;
    if (g == nc) {
      for (int t = lane; t < NOUT * nc; t += 64) {
        const int q = t / nc, k = t - q * nc;
        out[e * NOUT * nc + t] = res[k * CAP + q];
      }
    } else {
      for (int t = lane; t < NOUT * g; t += 64) {
        const int q = t / g, k = t - q * g;
        out[(e * NOUT + q) * nc + k0 + k] = res[k * CAP + q];
      }
    }
  }
}

template <typename T, int D, int NI, int NO>
int launch_tensor_interp_one(const T* in, const T* mat, const T* weight, T* out,
                             int64_t E, int nc, bool trans, hipStream_t st) {
  constexpr int NM = NI > NO ? NI : NO;
  constexpr int CAP = ipow(NM, D);
  // components per pass: what fits in 60 KB next to the matrix
  const size_t per = 2 * (size_t)CAP * sizeof(T);
  // components per pass: as many as fit in 20 KB of LDS (then 8 workgroups
  // share a CU; 64^3 elements, 8 -> 10 points, three components: 5.6 ms at
  // 20 KB against 7.8 ms with all three in 48 KB), at least one
  const size_t room = 20 * 1024 - (size_t)NO * NI * sizeof(T);
  int group = (int)(room / per);
  if (group < 1 && per + (size_t)NO * NI * sizeof(T) <= 60 * 1024) group = 1;
  if (group < 1) return SFEM_EUNSUPPORTED;
  if (group > nc) group = nc;
  const size_t lds = (size_t)NO * NI * sizeof(T) + per * group;
  hipLaunchKernelGGL((tensor_interp_kernel<T, D, NI, NO>), dim3((unsigned)E),
                     dim3(64), lds, st, in, mat, weight, out, E, nc, group,
                     (int)trans);
  return SFEM_OK;
}

// NO - NI in [-2, 2], both in [2, 14]; anything else: SFEM_EUNSUPPORTED
template <typename T, int D, int NI>
int launch_tensor_interp_ni(int no, const T* in, const T* mat, const T* weight,
                            T* out, int64_t E, int nc, bool trans,
                            hipStream_t st) {
#define SFEM_INTERP_CASE(DELTA)                                               \
  if constexpr (NI + (DELTA) >= 2 && NI + (DELTA) <= 14) {                    \
    if (no == NI + (DELTA))                                                   \
      return launch_tensor_interp_one<T, D, NI, NI + (DELTA)>(                \
          in, mat, weight, out, E, nc, trans, st);                            \
  }
  SFEM_INTERP_CASE(-2) SFEM_INTERP_CASE(-1) SFEM_INTERP_CASE(0)
  SFEM_INTERP_CASE(1) SFEM_INTERP_CASE(2)
#undef SFEM_INTERP_CASE
  return SFEM_EUNSUPPORTED;
}

template <typename T, int D>
int launch_tensor_interp_d(int ni, int no, const T* in, const T* mat,
                           const T* weight, T* out, int64_t E, int nc,
                           bool trans, hipStream_t st) {
#define SFEM_INTERP_NI(V)                                                     \
  case V:                                                                     \
    return launch_tensor_interp_ni<T, D, V>(no, in, mat, weight, out, E, nc,  \
                                            trans, st);
  switch (ni) {
    SFEM_INTERP_NI(2) SFEM_INTERP_NI(3) SFEM_INTERP_NI(4) SFEM_INTERP_NI(5)
    SFEM_INTERP_NI(6) SFEM_INTERP_NI(7) SFEM_INTERP_NI(8) SFEM_INTERP_NI(9)
    SFEM_INTERP_NI(10) SFEM_INTERP_NI(11) SFEM_INTERP_NI(12)
    SFEM_INTERP_NI(13) SFEM_INTERP_NI(14)
  }
#undef SFEM_INTERP_NI
  return SFEM_EUNSUPPORTED;
}

template <typename T>
int launch_tensor_interp_t(int ndim, int ni, int no, const void* in,
                           const void* mat, const void* weight, void* out,
                           int64_t E, int nc, bool trans, hipStream_t st) {
  if (E > 0x7fffffff) return SFEM_EUNSUPPORTED;
  if (ndim == 3)
    return launch_tensor_interp_d<T, 3>(ni, no, (const T*)in, (const T*)mat,
                                        (const T*)weight, (T*)out, E, nc, trans,
                                        st);
  if (ndim == 2)
    return launch_tensor_interp_d<T, 2>(ni, no, (const T*)in, (const T*)mat,
                                        (const T*)weight, (T*)out, E, nc, trans,
                                        st);
  return SFEM_EUNSUPPORTED;
}

// defined in sfem_interp_f64.hip / sfem_interp_f32.hip; SFEM_EUNSUPPORTED when
// there is no instantiation for the sizes (the caller runs the generic kernel)
int launch_tensor_interp_f64(int ndim, int ni, int no, const void* in,
                             const void* mat, const void* weight, void* out,
                             int64_t E, int nc, bool trans, hipStream_t st);
int launch_tensor_interp_f32(int ndim, int ni, int no, const void* in,
                             const void* mat, const void* weight, void* out,
                             int64_t E, int nc, bool trans, hipStream_t st);

}  // namespace sfem
#endif  // SFEM_INTERP_H_
