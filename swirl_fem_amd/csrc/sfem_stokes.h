// Fused divergence / pressure-gradient operators of the P_N - P_{N-2} Stokes
// discretisation for gfx950.
//
//   D  u = pressure.scatter( D_local( velocity.gather(u) ) )
//        D_local(u)_k = sum_q phi_k(x_q) w_q detJ_q div u(x_q)
//   D^T p = mask * velocity.scatter( Dt_local( pressure.gather(p) ) )
//        Dt_local(p)_{i,c} = sum_q w_q detJ_q p(x_q) d phi_i / d x_c (x_q)
//
// Reference: navier_stokes/navier_stokes.py:313-338 (D_local, Dt_local, D, Dt);
// both are evaluated on the velocity GLL points (`quadrature` of :279-282), the
// pressure lives on PP = P - 2 Gauss nodes per direction (:113-120).  The
// reference differentiates / integrates through dense Kronecker matrices; here
//
//   div u * w detJ = w * sum_{a,c} K[a][c] d u_c / d xi_a,   K = cofactors of J
//
// so the Jacobian determinant cancels and one element costs 3 d sum-factorised
// derivative lines + the projection onto (resp. interpolation from) the
// pressure basis, all in the line-per-lane layout of sfem_helmholtz.h: lane
// (i, j) owns the nodes (*, i, j), axis-0 contractions run in registers with
// the 1D matrices as scalar (kernel-argument) operands, the other axes go
// through the two padded LDS copies of the element.
#pragma once
#include "sfem_helmholtz.h"

namespace sfem {

template <typename T>
struct StokesParams {
  const T* u;            // div: (N, d) velocity
  T* out;                // grad_t: (N, d) result
  const T* p_in;         // grad_t: (Np,) pressure nodal values
  T* p_out;              // div: (Np,) result
  const T* scale;        // div: optional (N, d) factor applied to u when gathered
  const int32_t* enc;    // (E, n) encoded velocity indices
  const int32_t* penc;   // (E, np) pressure node ids, or null = e * np + k
  const T* kfac;         // GEO_POINT: (slots, d*d, Q)  w * cofactor planes
  const T* geo_elem;     // (E, 24) multilinear map coefficients
  const int32_t* geo_index;
  const int32_t* elem_list;
  const T* dmat_host;
  const T* weights_host;
  const T* nodes_host;
  const T* interp_host;  // (P, PP): phi_k(x_q), pressure basis at the GLL points
  int64_t num_elements;
  int geo_mode;
  int64_t node_stride, comp_stride;
  int64_t scale_node_stride, scale_comp_stride;   // comp stride 0: one factor per node
  const uint16_t* shared_order;   // as in HelmholtzParams, or null
  int shared_stride;
  double* dot_out;       // div: partial sums of p_in . p_out, or null
};

// Pressure-basis values at the velocity points, by value in the kernel
// arguments (scalar operands, like DMat): m[q * PP + k] = phi_k(x_q).
template <typename T, int P, int PP>
struct IMat {
  T m[P * PP];
};

// y[q] = sum_k M[q][k] x[k]
template <typename T, int P, int PP>
__device__ __forceinline__ void interp_fwd(const IMat<T, P, PP>& im,
                                           const T (&x)[PP], T (&y)[P]) {
#pragma unroll
  for (int q = 0; q < P; ++q) {
    T s = T(0);
#pragma unroll
    for (int k = 0; k < PP; ++k) s += im.m[q * PP + k] * x[k];
    y[q] = s;
  }
}

// y[k] = sum_q M[q][k] x[q]
template <typename T, int P, int PP>
__device__ __forceinline__ void interp_t(const IMat<T, P, PP>& im,
                                         const T (&x)[P], T (&y)[PP]) {
#pragma unroll
  for (int k = 0; k < PP; ++k) {
    T s = T(0);
#pragma unroll
    for (int q = 0; q < P; ++q) s += im.m[q * PP + k] * x[q];
    y[k] = s;
  }
}

// The same two products with the matrix re-read from the kernarg segment per
// product (scalar loads): 48 entries at P = 8 fp64 are 96 SGPRs otherwise.
template <typename T, int P, int PP>
__device__ __forceinline__ void interp_fwd_mem(
    const SFEM_CONSTANT_AS IMat<T, P, PP>* km, const T (&x)[PP], T (&y)[P]) {
  asm volatile("" : "+s"(km));
#pragma unroll
  for (int q = 0; q < P; ++q) {
    T s = T(0);
#pragma unroll
    for (int k = 0; k < PP; ++k) s += km->m[q * PP + k] * x[k];
    y[q] = s;
  }
}
template <typename T, int P, int PP>
__device__ __forceinline__ void interp_t_mem(
    const SFEM_CONSTANT_AS IMat<T, P, PP>* km, const T (&x)[P], T (&y)[PP]) {
  asm volatile("" : "+s"(km));
#pragma unroll
  for (int k = 0; k < PP; ++k) {
    T s = T(0);
#pragma unroll
    for (int q = 0; q < P; ++q) s += km->m[q * PP + k] * x[q];
    y[k] = s;
  }
}

// Per-lane cofactor state of one element:  Kw[a][c] = w_q detJ d xi_a / d x_c.
template <typename T, int P, int DIM, int GM>
struct ElemCof {
  static constexpr int NPT = DIM == 3 ? P * P * P : P * P;
  static constexpr int TPE = DIM == 3 ? P * P : P;
  static constexpr bool HAS_POINT = GM == GEO_POINT;
  static constexpr bool HAS_AFFINE = GM == GEO_AFFINE;
  static constexpr bool HAS_MULTI = GM == GEO_MULTILINEAR;

  const char* base;
  uint32_t lane_off;
  T wbc;                                  // in-plane quadrature weight
  T kc[HAS_AFFINE ? DIM * DIM : 1];       // affine: constant cofactors
  T r0[HAS_MULTI ? DIM : 1], p1[HAS_MULTI ? DIM : 1], q1[HAS_MULTI ? DIM : 1],
      p2[HAS_MULTI && DIM == 3 ? 3 : 1], q2[HAS_MULTI && DIM == 3 ? 3 : 1];

  __device__ __forceinline__ void init(const StokesParams<T>& prm,
                                       const DMat<T, P>& dm, int64_t e,
                                       bool active, int i, int j, int t) {
    int64_t slot = e;
    if (HAS_POINT && prm.geo_index) slot = active ? prm.geo_index[e] : 0;
    base = reinterpret_cast<const char*>(prm.kfac) +
           (HAS_POINT ? slot * (int64_t)(DIM * DIM) * NPT * sizeof(T) : 0);
    lane_off = (uint32_t)(t * sizeof(T));
    wbc = T(0);
    if (HAS_POINT || !active) return;
    const T wj = lane_pick<T, P>(dm.w, j);
    wbc = DIM == 3 ? lane_pick<T, P>(dm.w, i) * wj : wj;
    const T* A = prm.geo_elem + e * 24;
    if (DIM == 3) {
      const T s = lane_pick<T, P>(dm.x, i), tt = lane_pick<T, P>(dm.x, j);
      T a0[3], a1[3], a2[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const T A1 = A[c], A2 = A[3 + c], A3 = A[6 + c], A4 = A[9 + c],
                A5 = A[12 + c], A6 = A[15 + c], A7 = A[18 + c];
        a0[c] = A1 + A4 * s + (A6 + A7 * s) * tt;
        a1[c] = A2 + A5 * tt;
        a2[c] = A3 + A5 * s;
        if (HAS_MULTI) {
          r0[c] = a0[c];
          p1[c] = a1[c];
          q1[c] = A4 + A7 * tt;
          p2[c] = a2[c];
          q2[c] = A6 + A7 * s;
        }
      }
      if (HAS_AFFINE) {
        kc[0] = a1[1] * a2[2] - a1[2] * a2[1];
        kc[1] = a1[2] * a2[0] - a1[0] * a2[2];
        kc[2] = a1[0] * a2[1] - a1[1] * a2[0];
        kc[3] = a2[1] * a0[2] - a2[2] * a0[1];
        kc[4] = a2[2] * a0[0] - a2[0] * a0[2];
        kc[5] = a2[0] * a0[1] - a2[1] * a0[0];
        kc[6] = a0[1] * a1[2] - a0[2] * a1[1];
        kc[7] = a0[2] * a1[0] - a0[0] * a1[2];
        kc[8] = a0[0] * a1[1] - a0[1] * a1[0];
      }
    } else {
      const T s = lane_pick<T, P>(dm.x, j);
      T a0[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        a0[c] = A[c] + A[4 + c] * s;          // d x_c / d r
        if (HAS_MULTI) { r0[c] = a0[c]; p1[c] = A[2 + c]; q1[c] = A[4 + c]; }
      }
      if (HAS_AFFINE) {
        kc[0] = A[3];        //  y_s
        kc[1] = -A[2];       // -x_s
        kc[2] = -a0[1];      // -y_r
        kc[3] = a0[0];       //  x_r
      }
    }
  }

  // The same for a lane that already holds its in-plane weight and node
  // coordinates (3D; kernels that visit many elements per wave).
  __device__ __forceinline__ void init_lane(const StokesParams<T>& prm,
                                            int64_t e, int t, T wbc_, T s,
                                            T tt) {
    int64_t slot = e;
    if (HAS_POINT && prm.geo_index) slot = prm.geo_index[e];
    base = reinterpret_cast<const char*>(prm.kfac) +
           (HAS_POINT ? slot * (int64_t)(DIM * DIM) * NPT * sizeof(T) : 0);
    lane_off = (uint32_t)(t * sizeof(T));
    wbc = wbc_;
    if (HAS_POINT) return;
    const T* A = prm.geo_elem + e * 24;
    T a0[3], a1[3], a2[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const T A1 = A[c], A2 = A[3 + c], A3 = A[6 + c], A4 = A[9 + c],
              A5 = A[12 + c], A6 = A[15 + c], A7 = A[18 + c];
      a0[c] = A1 + A4 * s + (A6 + A7 * s) * tt;
      a1[c] = A2 + A5 * tt;
      a2[c] = A3 + A5 * s;
      if (HAS_MULTI) {
        r0[c] = a0[c];
        p1[c] = a1[c];
        q1[c] = A4 + A7 * tt;
        p2[c] = a2[c];
        q2[c] = A6 + A7 * s;
      }
    }
    if (HAS_AFFINE) {
      kc[0] = a1[1] * a2[2] - a1[2] * a2[1];
      kc[1] = a1[2] * a2[0] - a1[0] * a2[2];
      kc[2] = a1[0] * a2[1] - a1[1] * a2[0];
      kc[3] = a2[1] * a0[2] - a2[2] * a0[1];
      kc[4] = a2[2] * a0[0] - a2[0] * a0[2];
      kc[5] = a2[0] * a0[1] - a2[1] * a0[0];
      kc[6] = a0[1] * a1[2] - a0[2] * a1[1];
      kc[7] = a0[2] * a1[0] - a0[0] * a1[2];
      kc[8] = a0[0] * a1[1] - a0[1] * a1[0];
    }
  }

  // K[a' * DIM + c] at the lane's node of slice a, times the quadrature weight
  __device__ __forceinline__ void cof(const DMat<T, P>& dm, int a,
                                      T (&K)[DIM * DIM]) const {
    cof_wx(dm.w[a], dm.x[a], a, K);
  }
  // ... with the 1D weight and node value of slice a handed in
  __device__ __forceinline__ void cof_wx(T w_a, T x_a, int a,
                                         T (&K)[DIM * DIM]) const {
    if (HAS_POINT) {
#pragma unroll
      for (int f = 0; f < DIM * DIM; ++f)
        K[f] = *reinterpret_cast<const T*>(
            base + ((size_t)f * NPT * sizeof(T) + lane_off +
                    (uint32_t)(a * TPE * sizeof(T))));
      return;
    }
    const T wq = wbc * w_a;
    if (HAS_AFFINE) {
#pragma unroll
      for (int f = 0; f < DIM * DIM; ++f) K[f] = wq * kc[f];
      return;
    }
    const T r = x_a;
    if (DIM == 3) {
      T R1[3], R2[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        R1[c] = wq * (p1[c] + r * q1[c]);     // weight folded into one factor
        R2[c] = p2[c] + r * q2[c];
      }
      K[0] = R1[1] * R2[2] - R1[2] * R2[1];
      K[1] = R1[2] * R2[0] - R1[0] * R2[2];
      K[2] = R1[0] * R2[1] - R1[1] * R2[0];
      T W0[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) W0[c] = wq * r0[c];
      K[3] = R2[1] * W0[2] - R2[2] * W0[1];
      K[4] = R2[2] * W0[0] - R2[0] * W0[2];
      K[5] = R2[0] * W0[1] - R2[1] * W0[0];
      K[6] = r0[1] * R1[2] - r0[2] * R1[1];
      K[7] = r0[2] * R1[0] - r0[0] * R1[2];
      K[8] = r0[0] * R1[1] - r0[1] * R1[0];
    } else {
      const T R1x = p1[0] + r * q1[0], R1y = p1[1] + r * q1[1];
      K[0] = wq * R1y;
      K[1] = -wq * R1x;
      K[2] = -wq * r0[1];
      K[3] = wq * r0[0];
    }
  }
};

// Opaque-value fences: keep per-element geometry state from being expanded
// and hoisted out of the component loop (see helmholtz_kernel).
template <typename T, int P, int DIM, int GM>
__device__ __forceinline__ void cof_fence(ElemCof<T, P, DIM, GM>& g) {
  asm volatile("" : "+v"(g.lane_off));
  if (GM != GEO_POINT) asm volatile("" : "+v"(g.wbc));
  if (GM == GEO_MULTILINEAR) {
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
      asm volatile("" : "+v"(g.r0[c]), "+v"(g.p1[c]), "+v"(g.q1[c]));
      if (DIM == 3) asm volatile("" : "+v"(g.p2[c]), "+v"(g.q2[c]));
    }
  }
  if (GM == GEO_AFFINE) {
#pragma unroll
    for (int f = 0; f < DIM * DIM; ++f) asm volatile("" : "+v"(g.kc[f]));
  }
}

#define SFEM_STOKES_PROLOGUE                                                  \
  using Tile = HelmholtzTile<T, P, DIM>;                                      \
  constexpr int TPE = Tile::TPE, SA = Tile::SA, SB = Tile::SB;                \
  constexpr int EPB = Tile::EPB, W = Tile::ELEM_WORDS;                        \
  constexpr int N = DIM == 3 ? P * P * P : P * P;                             \
  constexpr int NP = DIM == 3 ? PP * PP * PP : PP * PP;                       \
  __shared__ T lds[2 * EPB * W];                                              \
  const int tid = threadIdx.x;                                                \
  const int el = tid / TPE;                                                   \
  const int t = tid - el * TPE;                                               \
  const int i = DIM == 3 ? t / P : 0;                                         \
  const int j = DIM == 3 ? t - i * P : t;                                     \
  const bool lane_ok = el < EPB;                                              \
  const int64_t work = (int64_t)blockIdx.x * EPB + (lane_ok ? el : 0);        \
  const bool active = lane_ok && work < prm.num_elements;                     \
  const int64_t e =                                                           \
      prm.elem_list ? (active ? (int64_t)prm.elem_list[work] : 0) : work;     \
  T* s0 = lds + (lane_ok ? el : 0) * 2 * W;                                   \
  T* s1 = s0 + W;                                                             \
  const int64_t ns = prm.node_stride, ks = prm.comp_stride;                   \
  ElemCof<T, P, DIM, GM> geom;                                                \
  geom.init(prm, dm, e, active, i, j, t);                                     \
  uint32_t enc[P];                                                            \
  {                                                                           \
    const int32_t* enc0 = prm.enc + e * N;                                    \
    _Pragma("unroll") for (int a = 0; a < P; ++a)                             \
      enc[a] = active ? (uint32_t)enc0[t + a * TPE] : (uint32_t)SFEM_IDX_PAD; \
  }                                                                           \
  const int32_t* penc0 = prm.penc ? prm.penc + e * NP : nullptr;              \
  const int64_t pbase = e * NP;                                               \
  (void)s1; (void)N; (void)penc0; (void)pbase

// SECOND: the second half of the split E = D Q D^T (see stokes_e_first_kernel):
// only the SHARED slots are gathered (the others were consumed in registers by
// the first half) and the projection is added to what the first half stored.
template <typename T, int P, int PP, int DIM, int GM, bool SECOND = false>
__global__ void __launch_bounds__((HelmholtzTile<T, P, DIM>::BLOCK),
                                  (HelmholtzTile<T, P, DIM>::MINW))
stokes_div_kernel(StokesParams<T> prm, DMat<T, P> dm, IMat<T, P, PP> im) {
  SFEM_STOKES_PROLOGUE;
  double pdot = 0.0;     // this lane's share of p_in . p_out
  T tq[P];
#pragma unroll
  for (int a = 0; a < P; ++a) tq[a] = T(0);
#pragma unroll
  for (int c = 0; c < DIM; ++c) {
    cof_fence(geom);
    T ua[P], d0[P];
#pragma unroll
    for (int a = 0; a < P; ++a) {
      uint32_t ea = enc[a];
      asm volatile("" : "+v"(ea));
      const uint32_t id = ea & SFEM_IDX_MASK;
      T v = T(0);
      if (id != SFEM_IDX_PAD && (!SECOND || (ea & SFEM_IDX_SHARED))) {
        v = prm.u[(int64_t)id * ns + c * ks];
        if (prm.scale)
          v *= prm.scale[(int64_t)id * prm.scale_node_stride +
                         c * prm.scale_comp_stride];
      }
      ua[a] = v;
    }
    line_apply<T, P, false>(dm, ua, d0);
    if (lane_ok) {
#pragma unroll
      for (int a = 0; a < P; ++a) {
        s0[a * SA + i * SB + j] = ua[a];
        if (DIM == 3) s1[a * SA + i * SB + j] = ua[a];
      }
    }
    __syncthreads();
    if (lane_ok) {  // last axis
      T* line = (DIM == 3 ? s1 + i * SA + j * SB : s0 + j * SA);
      T x[P], y[P];
#pragma unroll
      for (int m = 0; m < P; ++m) x[m] = line[m];
      line_apply<T, P, false>(dm, x, y);
#pragma unroll
      for (int m = 0; m < P; ++m) line[m] = y[m];
    }
    if (DIM == 3 && lane_ok) {  // middle axis
      T* line = s0 + i * SA + j;
      T x[P], y[P];
#pragma unroll
      for (int m = 0; m < P; ++m) x[m] = line[m * SB];
      line_apply<T, P, false>(dm, x, y);
#pragma unroll
      for (int m = 0; m < P; ++m) line[m * SB] = y[m];
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int a = 0; a < P; ++a) {
        const int o = a * SA + i * SB + j;
        T K[DIM * DIM];
        geom.cof(dm, a, K);
        T v = K[c] * d0[a] + K[DIM + c] * s0[o];
        if (DIM == 3) v += K[2 * DIM + c] * s1[o];
        tq[a] += v;
      }
    }
    __syncthreads();
  }
  // projection onto the pressure basis: axis 0 in registers, then LDS lines
  T r[PP];
  interp_t<T, P, PP>(im, tq, r);
  if (lane_ok) {
#pragma unroll
    for (int k = 0; k < PP; ++k) s0[k * SA + i * SB + j] = r[k];
  }
  __syncthreads();
  if (DIM == 3) {
    if (lane_ok && i < PP) {  // line [k0 = i, *, j]
      T* line = s0 + i * SA + j;
      T x[P], y[PP];
#pragma unroll
      for (int m = 0; m < P; ++m) x[m] = line[m * SB];
      interp_t<T, P, PP>(im, x, y);
#pragma unroll
      for (int k = 0; k < PP; ++k) line[k * SB] = y[k];
    }
    __syncthreads();
    if (active && i < PP && j < PP) {  // line [k0 = i, k1 = j, *]
      const T* line = s0 + i * SA + j * SB;
      T x[P], y[PP];
#pragma unroll
      for (int m = 0; m < P; ++m) x[m] = line[m];
      interp_t<T, P, PP>(im, x, y);
#pragma unroll
      for (int k = 0; k < PP; ++k) {
        const int slot = (i * PP + j) * PP + k;
        const int64_t pid = penc0 ? (int64_t)penc0[slot] : pbase + slot;
        if (pid >= 0) {
          prm.p_out[pid] = SECOND ? prm.p_out[pid] + y[k] : y[k];
          if (!SECOND && prm.dot_out)
            pdot += (double)y[k] * (double)prm.p_in[pid];
        }
      }
    }
  } else {
    if (active && j < PP) {  // line [k0 = j, *]
      const T* line = s0 + j * SA;
      T x[P], y[PP];
#pragma unroll
      for (int m = 0; m < P; ++m) x[m] = line[m];
      interp_t<T, P, PP>(im, x, y);
#pragma unroll
      for (int k = 0; k < PP; ++k) {
        const int slot = j * PP + k;
        const int64_t pid = penc0 ? (int64_t)penc0[slot] : pbase + slot;
        if (pid >= 0) {
          prm.p_out[pid] = SECOND ? prm.p_out[pid] + y[k] : y[k];
          if (!SECOND && prm.dot_out)
            pdot += (double)y[k] * (double)prm.p_in[pid];
        }
      }
    }
  }
  if (!SECOND && prm.dot_out) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) pdot += __shfl_down(pdot, off, 64);
    if ((tid & 63) == 0)
      unsafeAtomicAdd(&prm.dot_out[(blockIdx.x * (Tile::BLOCK / 64) +
                                    (tid >> 6)) & (SFEM_DOT_SLOTS - 1)],
                      pdot);
  }
}

template <typename T, int P, int PP, int DIM, int GM, bool SORTED = false>
__global__ void __launch_bounds__((HelmholtzTile<T, P, DIM>::BLOCK),
                                  (HelmholtzTile<T, P, DIM>::MINW))
stokes_grad_t_kernel(StokesParams<T> prm, DMat<T, P> dm, IMat<T, P, PP> im) {
  SFEM_STOKES_PROLOGUE;
  // pressure at the velocity points of the lane's line: interpolate the last
  // axis first (lines [k0, k1, *]), then the middle one, axis 0 in registers
  T tq[P];
  if (DIM == 3) {
    if (lane_ok && i < PP && j < PP) {
      T x[PP], y[P];
#pragma unroll
      for (int k = 0; k < PP; ++k) {
        const int slot = (i * PP + j) * PP + k;
        int64_t pid = -1;
        if (active) pid = penc0 ? (int64_t)penc0[slot] : pbase + slot;
        x[k] = pid >= 0 ? prm.p_in[pid] : T(0);
      }
      interp_fwd<T, P, PP>(im, x, y);
      T* line = s0 + i * SA + j * SB;
#pragma unroll
      for (int m = 0; m < P; ++m) line[m] = y[m];
    }
    __syncthreads();
    if (lane_ok && i < PP) {
      T* line = s0 + i * SA + j;
      T x[PP], y[P];
#pragma unroll
      for (int k = 0; k < PP; ++k) x[k] = line[k * SB];
      interp_fwd<T, P, PP>(im, x, y);
#pragma unroll
      for (int m = 0; m < P; ++m) line[m * SB] = y[m];
    }
    __syncthreads();
  } else {
    if (lane_ok && j < PP) {
      T x[PP], y[P];
#pragma unroll
      for (int k = 0; k < PP; ++k) {
        const int slot = j * PP + k;
        int64_t pid = -1;
        if (active) pid = penc0 ? (int64_t)penc0[slot] : pbase + slot;
        x[k] = pid >= 0 ? prm.p_in[pid] : T(0);
      }
      interp_fwd<T, P, PP>(im, x, y);
      T* line = s0 + j * SA;
#pragma unroll
      for (int m = 0; m < P; ++m) line[m] = y[m];
    }
    __syncthreads();
  }
  {
    T x[PP];
#pragma unroll
    for (int k = 0; k < PP; ++k)
      x[k] = lane_ok ? s0[k * SA + i * SB + j] : T(0);
    interp_fwd<T, P, PP>(im, x, tq);
  }
  __syncthreads();

#pragma unroll
  for (int c = 0; c < DIM; ++c) {
    cof_fence(geom);
    T w0[P];
#pragma unroll
    for (int a = 0; a < P; ++a) w0[a] = T(0);
    if (lane_ok) {
#pragma unroll
      for (int a = 0; a < P; ++a) {
        const int o = a * SA + i * SB + j;
        T K[DIM * DIM];
#pragma unroll
        for (int f = 0; f < DIM * DIM; ++f) K[f] = T(0);
        if (active) geom.cof(dm, a, K);
        w0[a] = K[c] * tq[a];
        s0[o] = K[DIM + c] * tq[a];
        if (DIM == 3) s1[o] = K[2 * DIM + c] * tq[a];
      }
    }
    __syncthreads();
    if (lane_ok) {
      T* line = (DIM == 3 ? s1 + i * SA + j * SB : s0 + j * SA);
      T x[P], y[P];
#pragma unroll
      for (int m = 0; m < P; ++m) x[m] = line[m];
      line_apply<T, P, true>(dm, x, y);
#pragma unroll
      for (int m = 0; m < P; ++m) line[m] = y[m];
    }
    if (DIM == 3 && lane_ok) {
      T* line = s0 + i * SA + j;
      T x[P], y[P];
#pragma unroll
      for (int m = 0; m < P; ++m) x[m] = line[m * SB];
      line_apply<T, P, true>(dm, x, y);
#pragma unroll
      for (int m = 0; m < P; ++m) line[m * SB] = y[m];
    }
    T dt0[P];
    line_apply<T, P, true>(dm, w0, dt0);
    __syncthreads();
    constexpr bool sorted = SORTED;        // own instantiation, see helmholtz_kernel
#pragma unroll
    for (int a = 0; a < P; ++a) {
      uint32_t ea = enc[a];
      asm volatile("" : "+v"(ea));
      const uint32_t id = ea & SFEM_IDX_MASK;
      const int o = a * SA + i * SB + j;
      T v = T(0);
      if (lane_ok) {
        v = dt0[a] + s0[o];
        if (DIM == 3) v += s1[o];
      }
      // a diagonal factor that is the same on every copy of a node commutes
      // with the assembly: E = D QQ^T (Q . D^T) saves the gather of Q in D
      if (prm.scale && id != SFEM_IDX_PAD)
        v *= prm.scale[(int64_t)id * prm.scale_node_stride +
                       c * prm.scale_comp_stride];
      dt0[a] = v;
      if (id != SFEM_IDX_PAD) {
        T* dst = prm.out + (int64_t)id * ns + c * ks;
        const bool dirichlet = ea & SFEM_IDX_DIRICHLET;
        if (ea & SFEM_IDX_SHARED) {
          if (!dirichlet && !sorted) unsafeAtomicAdd(dst, v);
        } else {
          *dst = dirichlet ? T(0) : v;
        }
      }
    }
    if constexpr (sorted) {   // shared slots in ascending node order
      uint32_t* codes = reinterpret_cast<uint32_t*>(s1);
      if (Tile::BLOCK > 64) __syncthreads();
      if (lane_ok) {
#pragma unroll
        for (int a = 0; a < P; ++a) {
          s0[a * SA + i * SB + j] = dt0[a];
          codes[a * TPE + t] = enc[a];
        }
      }
      __syncthreads();
      if (active)
        scatter_shared_sorted<T, P, DIM>(
            prm.shared_order + e * prm.shared_stride, prm.shared_stride, t, s0,
            codes, prm.out + c * ks, ns);
    }
    __syncthreads();
  }
}

// First half of the split pressure operator  E p = D [ Q . QQ^T ( D^T p ) ]
// (navier_stokes.py:340-348) for a diagonal Q.  A velocity node that belongs
// to one element only -- and takes no part in the periodic / partition
// exchange: `enc` flags those SHARED as well -- is complete after this
// element's D^T, so its value never has to leave the registers: it is scaled
// by Q and fed to this element's D at once.  Only the SHARED slots are
// accumulated in memory (atomics into `out`, as in stokes_grad_t_kernel) and
// picked up again by the second half (stokes_div_kernel<SECOND>) after the
// exchange.  At P = 8 in 3D that removes 216 of the 512 nodes of every element
// from both the scatter and the gather, and the field in between shrinks to
// the element surfaces.  p_out receives D of the complete part.
template <typename T, int P, int PP, int DIM, int GM, bool SORTED = false>
__global__ void __launch_bounds__((HelmholtzTile<T, P, DIM>::BLOCK),
                                  (HelmholtzTile<T, P, DIM>::MINW))
stokes_e_first_kernel(StokesParams<T> prm, DMat<T, P> dm, IMat<T, P, PP> im) {
  SFEM_STOKES_PROLOGUE;
  T tq[P];
  if (DIM == 3) {
    if (lane_ok && i < PP && j < PP) {
      T x[PP], y[P];
#pragma unroll
      for (int k = 0; k < PP; ++k) {
        const int slot = (i * PP + j) * PP + k;
        int64_t pid = -1;
        if (active) pid = penc0 ? (int64_t)penc0[slot] : pbase + slot;
        x[k] = pid >= 0 ? prm.p_in[pid] : T(0);
      }
      interp_fwd<T, P, PP>(im, x, y);
      T* line = s0 + i * SA + j * SB;
#pragma unroll
      for (int m = 0; m < P; ++m) line[m] = y[m];
    }
    __syncthreads();
    if (lane_ok && i < PP) {
      T* line = s0 + i * SA + j;
      T x[PP], y[P];
#pragma unroll
      for (int k = 0; k < PP; ++k) x[k] = line[k * SB];
      interp_fwd<T, P, PP>(im, x, y);
#pragma unroll
      for (int m = 0; m < P; ++m) line[m * SB] = y[m];
    }
    __syncthreads();
  } else {
    if (lane_ok && j < PP) {
      T x[PP], y[P];
#pragma unroll
      for (int k = 0; k < PP; ++k) {
        const int slot = j * PP + k;
        int64_t pid = -1;
        if (active) pid = penc0 ? (int64_t)penc0[slot] : pbase + slot;
        x[k] = pid >= 0 ? prm.p_in[pid] : T(0);
      }
      interp_fwd<T, P, PP>(im, x, y);
      T* line = s0 + j * SA;
#pragma unroll
      for (int m = 0; m < P; ++m) line[m] = y[m];
    }
    __syncthreads();
  }
  {
    T x[PP];
#pragma unroll
    for (int k = 0; k < PP; ++k)
      x[k] = lane_ok ? s0[k * SA + i * SB + j] : T(0);
    interp_fwd<T, P, PP>(im, x, tq);
  }
  __syncthreads();

  T dq[P];          // w div(complete part) at the velocity points
#pragma unroll
  for (int a = 0; a < P; ++a) dq[a] = T(0);
#pragma unroll
  for (int c = 0; c < DIM; ++c) {
    cof_fence(geom);
    // ---- D^T p, component c (stokes_grad_t_kernel)
    T w0[P];
#pragma unroll
    for (int a = 0; a < P; ++a) w0[a] = T(0);
    if (lane_ok) {
#pragma unroll
      for (int a = 0; a < P; ++a) {
        const int o = a * SA + i * SB + j;
        T K[DIM * DIM];
#pragma unroll
        for (int f = 0; f < DIM * DIM; ++f) K[f] = T(0);
        if (active) geom.cof(dm, a, K);
        w0[a] = K[c] * tq[a];
        s0[o] = K[DIM + c] * tq[a];
        if (DIM == 3) s1[o] = K[2 * DIM + c] * tq[a];
      }
    }
    __syncthreads();
    if (lane_ok) {
      T* line = (DIM == 3 ? s1 + i * SA + j * SB : s0 + j * SA);
      T x[P], y[P];
#pragma unroll
      for (int m = 0; m < P; ++m) x[m] = line[m];
      line_apply<T, P, true>(dm, x, y);
#pragma unroll
      for (int m = 0; m < P; ++m) line[m] = y[m];
    }
    if (DIM == 3 && lane_ok) {
      T* line = s0 + i * SA + j;
      T x[P], y[P];
#pragma unroll
      for (int m = 0; m < P; ++m) x[m] = line[m * SB];
      line_apply<T, P, true>(dm, x, y);
#pragma unroll
      for (int m = 0; m < P; ++m) line[m * SB] = y[m];
    }
    T ua[P];
    line_apply<T, P, true>(dm, w0, ua);
    __syncthreads();
    // ---- shared slots go to memory, complete ones stay (scaled by Q)
#pragma unroll
    for (int a = 0; a < P; ++a) {
      uint32_t ea = enc[a];
      asm volatile("" : "+v"(ea));
      const uint32_t id = ea & SFEM_IDX_MASK;
      const int o = a * SA + i * SB + j;
      T v = T(0);
      if (lane_ok) {
        v = ua[a] + s0[o];
        if (DIM == 3) v += s1[o];
      }
      // (own position: staged for the sorted scatter of the shared slots)
      if (SORTED && lane_ok) s0[o] = v;
      ua[a] = T(0);
      if (id != SFEM_IDX_PAD && !(ea & SFEM_IDX_DIRICHLET)) {
        if (ea & SFEM_IDX_SHARED) {
          if (!SORTED)
            unsafeAtomicAdd(prm.out + (int64_t)id * ns + c * ks, v);
        } else {
          if (prm.scale)
            v *= prm.scale[(int64_t)id * prm.scale_node_stride +
                           c * prm.scale_comp_stride];
          ua[a] = v;
        }
      }
    }
    if constexpr (SORTED) {
      uint32_t* codes = reinterpret_cast<uint32_t*>(s1);
      if (Tile::BLOCK > 64) __syncthreads();
      if (lane_ok) {
#pragma unroll
        for (int a = 0; a < P; ++a) codes[a * TPE + t] = enc[a];
      }
      __syncthreads();
      if (active)
        scatter_shared_sorted<T, P, DIM>(
            prm.shared_order + e * prm.shared_stride, prm.shared_stride, t, s0,
            codes, prm.out + c * ks, ns);
    }
    __syncthreads();
    // ---- D of the complete part, component c (stokes_div_kernel)
    T d0[P];
    line_apply<T, P, false>(dm, ua, d0);
    if (lane_ok) {
#pragma unroll
      for (int a = 0; a < P; ++a) {
        s0[a * SA + i * SB + j] = ua[a];
        if (DIM == 3) s1[a * SA + i * SB + j] = ua[a];
      }
    }
    __syncthreads();
    if (lane_ok) {
      T* line = (DIM == 3 ? s1 + i * SA + j * SB : s0 + j * SA);
      T x[P], y[P];
#pragma unroll
      for (int m = 0; m < P; ++m) x[m] = line[m];
      line_apply<T, P, false>(dm, x, y);
#pragma unroll
      for (int m = 0; m < P; ++m) line[m] = y[m];
    }
    if (DIM == 3 && lane_ok) {
      T* line = s0 + i * SA + j;
      T x[P], y[P];
#pragma unroll
      for (int m = 0; m < P; ++m) x[m] = line[m * SB];
      line_apply<T, P, false>(dm, x, y);
#pragma unroll
      for (int m = 0; m < P; ++m) line[m * SB] = y[m];
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int a = 0; a < P; ++a) {
        const int o = a * SA + i * SB + j;
        T K[DIM * DIM];
        geom.cof(dm, a, K);
        T v = K[c] * d0[a] + K[DIM + c] * s0[o];
        if (DIM == 3) v += K[2 * DIM + c] * s1[o];
        dq[a] += v;
      }
    }
    __syncthreads();
  }
  // ---- projection onto the pressure basis (stokes_div_kernel)
  T r[PP];
  interp_t<T, P, PP>(im, dq, r);
  if (lane_ok) {
#pragma unroll
    for (int k = 0; k < PP; ++k) s0[k * SA + i * SB + j] = r[k];
  }
  __syncthreads();
  if (DIM == 3) {
    if (lane_ok && i < PP) {
      T* line = s0 + i * SA + j;
      T x[P], y[PP];
#pragma unroll
      for (int m = 0; m < P; ++m) x[m] = line[m * SB];
      interp_t<T, P, PP>(im, x, y);
#pragma unroll
      for (int k = 0; k < PP; ++k) line[k * SB] = y[k];
    }
    __syncthreads();
    if (active && i < PP && j < PP) {
      const T* line = s0 + i * SA + j * SB;
      T x[P], y[PP];
#pragma unroll
      for (int m = 0; m < P; ++m) x[m] = line[m];
      interp_t<T, P, PP>(im, x, y);
#pragma unroll
      for (int k = 0; k < PP; ++k) {
        const int slot = (i * PP + j) * PP + k;
        const int64_t pid = penc0 ? (int64_t)penc0[slot] : pbase + slot;
        if (pid >= 0) prm.p_out[pid] = y[k];
      }
    }
  } else {
    if (active && j < PP) {
      const T* line = s0 + j * SA;
      T x[P], y[PP];
#pragma unroll
      for (int m = 0; m < P; ++m) x[m] = line[m];
      interp_t<T, P, PP>(im, x, y);
#pragma unroll
      for (int k = 0; k < PP; ++k) {
        const int slot = j * PP + k;
        const int64_t pid = penc0 ? (int64_t)penc0[slot] : pbase + slot;
        if (pid >= 0) prm.p_out[pid] = y[k];
      }
    }
  }
}

// Convection on a collocated grid (the over-integration grid of
// StokesVelocity.C_local, navier_stokes.py:238-245, reached by interpolation):
//   out[e, q, c] = w_q detJ_q  sum_j u_j(x_q) d u_c / d x_j (x_q)
//                = w_q sum_a ( sum_j K[a][j] u_j ) d u_c / d xi_a
// Element-local in and out, (E, n, DIM) row-major; the transposed
// interpolation back to the nodes and the scatter follow in separate kernels.
template <typename T, int P, int DIM, int GM>
__global__ void __launch_bounds__((HelmholtzTile<T, P, DIM>::BLOCK),
                                  (HelmholtzTile<T, P, DIM>::MINW))
stokes_convect_kernel(StokesParams<T> prm, DMat<T, P> dm) {
  using Tile = HelmholtzTile<T, P, DIM>;
  constexpr int TPE = Tile::TPE, SA = Tile::SA, SB = Tile::SB;
  constexpr int EPB = Tile::EPB, W = Tile::ELEM_WORDS;
  constexpr int N = DIM == 3 ? P * P * P : P * P;
  __shared__ T lds[2 * EPB * W];
  const int tid = threadIdx.x;
  const int el = tid / TPE;
  const int t = tid - el * TPE;
  const int i = DIM == 3 ? t / P : 0;
  const int j = DIM == 3 ? t - i * P : t;
  const bool lane_ok = el < EPB;
  const int64_t work = (int64_t)blockIdx.x * EPB + (lane_ok ? el : 0);
  const bool active = lane_ok && work < prm.num_elements;
  const int64_t e =
      prm.elem_list ? (active ? (int64_t)prm.elem_list[work] : 0) : work;
  T* s0 = lds + (lane_ok ? el : 0) * 2 * W;
  T* s1 = s0 + W;
  (void)s1;
  ElemCof<T, P, DIM, GM> geom;
  geom.init(prm, dm, e, active, i, j, t);
  const T* ue = prm.u + e * N * DIM;
  T* oe = prm.out + e * N * DIM;

  T uall[DIM][P];
#pragma unroll
  for (int c = 0; c < DIM; ++c)
#pragma unroll
    for (int a = 0; a < P; ++a)
      uall[c][a] = active ? ue[(int64_t)(t + a * TPE) * DIM + c] : T(0);

#pragma unroll
  for (int c = 0; c < DIM; ++c) {
    cof_fence(geom);
    T d0[P];
    line_apply<T, P, false>(dm, uall[c], d0);
    if (lane_ok) {
#pragma unroll
      for (int a = 0; a < P; ++a) {
        s0[a * SA + i * SB + j] = uall[c][a];
        if (DIM == 3) s1[a * SA + i * SB + j] = uall[c][a];
      }
    }
    __syncthreads();
    if (lane_ok) {  // last axis
      T* line = (DIM == 3 ? s1 + i * SA + j * SB : s0 + j * SA);
      T x[P], y[P];
#pragma unroll
      for (int m = 0; m < P; ++m) x[m] = line[m];
      line_apply<T, P, false>(dm, x, y);
#pragma unroll
      for (int m = 0; m < P; ++m) line[m] = y[m];
    }
    if (DIM == 3 && lane_ok) {  // middle axis
      T* line = s0 + i * SA + j;
      T x[P], y[P];
#pragma unroll
      for (int m = 0; m < P; ++m) x[m] = line[m * SB];
      line_apply<T, P, false>(dm, x, y);
#pragma unroll
      for (int m = 0; m < P; ++m) line[m * SB] = y[m];
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int a = 0; a < P; ++a) {
        const int o = a * SA + i * SB + j;
        T K[DIM * DIM];
        geom.cof(dm, a, K);
        T v = T(0);
#pragma unroll
        for (int ax = 0; ax < DIM; ++ax) {
          T U = T(0);                 // contravariant velocity along xi_ax
#pragma unroll
          for (int jj = 0; jj < DIM; ++jj) U += K[ax * DIM + jj] * uall[jj][a];
          const T g = ax == 0 ? d0[a] : (ax == 1 ? s0[o] : s1[o]);
          v += U * g;
        }
        oe[(int64_t)(t + a * TPE) * DIM + c] = v;
      }
    }
    __syncthreads();
  }
}

template <typename T, int P, int DIM>
int launch_stokes(const StokesParams<T>& prm, int mode, hipStream_t stream) {
  const bool grad_t = mode == 1;
  constexpr int PP = P - 2;
  using Tile = HelmholtzTile<T, P, DIM>;
  const int64_t groups = (prm.num_elements + Tile::EPB - 1) / Tile::EPB;
  if (groups > 0x7fffffff) {
    set_error("stokes: too many workgroups (%lld)", (long long)groups);
    return SFEM_EINVAL;
  }
  const DMat<T, P> dm =
      make_dmat<T, P>(prm.dmat_host, prm.weights_host, prm.nodes_host);
  IMat<T, P, PP> im;
  for (int q = 0; q < P * PP; ++q)
    im.m[q] = prm.interp_host ? prm.interp_host[q] : T(0);
  const dim3 grid((unsigned)groups), block(Tile::BLOCK);
  // sorted shared scatter: own instantiations, one-wave elements in 3D
  constexpr bool CAN_SORT = DIM == 3 && Tile::TPE <= 64;
  const bool sorted = CAN_SORT && prm.shared_order != nullptr;
#define SFEM_LAUNCH_STOKES(GMV)                                               \
  if (mode == 2)                                                              \
    hipLaunchKernelGGL((stokes_convect_kernel<T, P, DIM, GMV>), grid, block,  \
                       0, stream, prm, dm);                                   \
  else if (mode == 3 && sorted)                                               \
    hipLaunchKernelGGL((stokes_e_first_kernel<T, P, PP, DIM, GMV, CAN_SORT>), \
                       grid, block, 0, stream, prm, dm, im);                  \
  else if (mode == 3)                                                         \
    hipLaunchKernelGGL((stokes_e_first_kernel<T, P, PP, DIM, GMV, false>),    \
                       grid, block, 0, stream, prm, dm, im);                  \
  else if (mode == 4)                                                         \
    hipLaunchKernelGGL((stokes_div_kernel<T, P, PP, DIM, GMV, true>), grid,   \
                       block, 0, stream, prm, dm, im);                        \
  else if (grad_t && sorted)                                                  \
    hipLaunchKernelGGL((stokes_grad_t_kernel<T, P, PP, DIM, GMV, CAN_SORT>),  \
                       grid, block, 0, stream, prm, dm, im);                  \
  else if (grad_t)                                                            \
    hipLaunchKernelGGL((stokes_grad_t_kernel<T, P, PP, DIM, GMV, false>),     \
                       grid, block, 0, stream, prm, dm, im);                  \
  else                                                                        \
    hipLaunchKernelGGL((stokes_div_kernel<T, P, PP, DIM, GMV>), grid, block,  \
                       0, stream, prm, dm, im)
  switch (prm.geo_mode) {
    case GEO_POINT: SFEM_LAUNCH_STOKES(GEO_POINT); break;
    case GEO_AFFINE: SFEM_LAUNCH_STOKES(GEO_AFFINE); break;
    default: SFEM_LAUNCH_STOKES(GEO_MULTILINEAR); break;
  }
#undef SFEM_LAUNCH_STOKES
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

// Defined once per (dtype, ndim) translation unit.
template <typename T, int DIM>
int dispatch_stokes(const StokesParams<T>& prm, int P, int mode,
                    hipStream_t stream);

#define SFEM_STOKES_CASE(PP_) \
  case PP_: return launch_stokes<T, PP_, DIM>(prm, mode, stream);

#define SFEM_DEFINE_STOKES_DISPATCH(TYPE, DIMV)                              \
  template <>                                                                \
  int dispatch_stokes<TYPE, DIMV>(const StokesParams<TYPE>& prm, int P,      \
                                  int mode, hipStream_t stream) {            \
    using T = TYPE;                                                          \
    constexpr int DIM = DIMV;                                                \
    switch (P) {                                                             \
      SFEM_STOKES_CASE(4) SFEM_STOKES_CASE(5)                                \
      SFEM_STOKES_CASE(6) SFEM_STOKES_CASE(7) SFEM_STOKES_CASE(8)            \
      SFEM_STOKES_CASE(9) SFEM_STOKES_CASE(10) SFEM_STOKES_CASE(11)          \
      SFEM_STOKES_CASE(12)                                                   \
      default:                                                               \
        set_error("stokes: P=%d outside the compiled range 4..12", P);       \
        return SFEM_EUNSUPPORTED;                                            \
    }                                                                        \
  }

}  // namespace sfem
