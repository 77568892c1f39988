// Fused collocated Helmholtz / mass / stiffness operator for gfx950.
//
//   out = mask * scatter( lambda0 * B_loc(g) + lambda1 * A_loc(g) ),  g = gather(u)
//
// Reference call sites: examples/poisson.py:141-154 (A, B),
// navier_stokes/navier_stokes.py:220-236 (A_local, B_local), :295-307 (B, A),
// :431 (H = beta_k/dt B + mu A).  The reference applies the dense
// (Q, n, d) Kronecker matrix per element (core/interpolation.py:288-292) and
// its linear transpose (core/fespace.py:471); here each element is contracted
// axis by axis with the 1D differentiation matrix.
//
// Mapping onto CDNA4
//   * One *line* of P nodes per lane.  In 3D an element uses P^2 lanes: lane
//     (i, j) first owns the line along axis 0 (nodes [*, i, j]), so the global
//     gather / scatter and the geometric-factor reads are 64-lane coalesced
//     (consecutive lanes <-> consecutive nodes), and the axis-0 derivative is a
//     P x P register contraction whose D entries are wave-uniform (scalar
//     loads, SGPR operands -- no VGPRs and no LDS reads for the matrix).
//   * The other two axes are done the same way after a transpose through LDS:
//     lane (i, j) re-reads the element as the line [i, *, j] resp. [i, j, *],
//     contracts it in registers and writes the result back in place.  Per
//     element that is ~15 P LDS accesses per lane instead of the ~4 P^2 of a
//     slice-by-slice kernel; rows are padded (SB = P | 1) so the three access
//     patterns are bank-conflict free for 8-byte words.
//   * P = 8 (the p = 7 headline case): P^2 = 64 = one wavefront per element,
//     one-wave workgroups, so every barrier below is elided by the compiler
//     and an element never waits on another wave.
//   * HBM traffic per element is the algorithmic minimum of the stored-factor
//     model: n encoded indices (4 B), n gathered values, 6 (+1 with mass)
//     geometric factors per point, n results.  Dirichlet mask and the
//     shared/owned classification ride in the top bits of the index, shared
//     nodes are accumulated with HBM atomics, owned nodes with plain stores.
//   * fp64/fp32 MFMA on gfx950 runs at the VALU rate (MI355X_MICROARCH.md,
//     "Matrix cores"), the P x P contractions would half-fill a 16x16x4 tile,
//     and the kernel is HBM-bound at ~20 % VALU occupancy, so the contractions
//     stay on the vector ALU.
#pragma once
#include "sfem_common.h"

namespace sfem {

template <typename T>
struct HelmholtzParams {
  const T* u;            // (N, nc) or (E, n, nc) when !GS
  T* out;
  const int32_t* enc;    // (E, n) encoded indices (GS only)
  const T* geo;          // per-point factors of the non-affine elements
  const T* geo_elem;     // (E, 8) per-element constants, or null
  const int32_t* geo_index;  // (E,) slot in `geo`, -1 = affine; or null
  const T* dmat_host;    // (P, P) on the HOST; travels as a kernel argument
  const T* weights_host; // (P,) quadrature weights on the HOST (affine path)
  int64_t num_elements;
  int ncomp;
  T lambda0, lambda1;
  int debug_flags;       // reserved for A/B experiments (SFEM_DEBUG_FLAGS)
};

__host__ __device__ constexpr int round_up(int a, int b) {
  return (a + b - 1) / b * b;
}

template <typename T, int P, int DIM>
struct HelmholtzTile {
  static constexpr int TPE = DIM == 3 ? P * P : P;       // lanes per element
  static constexpr int SB = P | 1;                       // padded row stride
  static constexpr int SA = DIM == 3 ? P * SB : SB;      // axis-0 stride
  static constexpr int ELEM_WORDS = P * SA;              // one padded tensor
  static constexpr int LDS_PER_ELEM = 2 * ELEM_WORDS * (int)sizeof(T);
  // elements per workgroup: fill whole waves, stay under ~40 KiB of LDS and
  // 512 threads; a single wave when an element divides a wave evenly.
  static constexpr int pick_epb() {
    if (64 % TPE == 0) return 64 / TPE;
    int best = 1;
    double best_util = 0.0;
    for (int epb = 1; epb <= 16; ++epb) {
      const int thr = round_up(epb * TPE, 64);
      if (thr > 512 || epb * LDS_PER_ELEM > 40 * 1024) break;
      const double util = double(epb * TPE) / thr;
      if (util > best_util + 1e-9) { best_util = util; best = epb; }
    }
    return best;
  }
  static constexpr int EPB = pick_epb();
  static constexpr int BLOCK = round_up(EPB * TPE, 64);
  // register budget: ask for >= MINW waves per SIMD (latency hiding for an
  // HBM-bound kernel); larger lines need more registers per lane.
  static constexpr int MINW = (P * (int)sizeof(T) <= 64) ? 4 : 2;
  static constexpr int NGEO = DIM == 3 ? 6 : (DIM == 2 ? 3 : 1);
};

// The 1D differentiation matrix travels BY VALUE as a kernel argument in its
// even/odd split form.  The kernarg segment is constant address space, so every
// entry (compile-time index after unrolling) is a scalar load into SGPRs and
// feeds v_fma as the scalar operand: no VGPRs and no LDS traffic for the matrix.
//
// Even/odd split: for node sets symmetric about 0 (GLL, GL, Newton-Cotes) D is
// centro-antisymmetric, D[P-1-r][P-1-m] = -D[r][m].  With PH = P/2, PC = P-PH,
//   xe[m] = x[m] + x[P-1-m],  xo[m] = x[m] - x[P-1-m]          (m < PH)
//   E[r][m] = (D[r][m] + D[r][P-1-m]) / 2   (r < PH, m < PH),  E[r][PH] = D[r][PH]
//   O[r][m] = (D[r][m] - D[r][P-1-m]) / 2   (r < PH, m < PH),  O[PH][m] = D[PH][m]
//   se = E xe, so = O xo;   y[r] = so[r] + se[r],  y[P-1-r] = so[r] - se[r]
// halves both the multiplies and the matrix (2 PH PC entries: 32 doubles = 64
// SGPRs at P = 8).  D^T is centro-antisymmetric too and its split is the same
// pair with the roles swapped: E' = O^T, O' = E^T.
template <typename T, int P>
struct DMat {
  static constexpr int PH = P / 2, PC = P - P / 2;
  T e[PH * PC];   // E[r][m], r < PH, m < PC
  T o[PC * PH];   // O[r][m], r < PC, m < PH
  T w[P];         // 1D quadrature weights (affine elements)
};

template <typename T, int P>
inline DMat<T, P> make_dmat(const T* d, const T* w) {
  constexpr int PH = P / 2, PC = P - P / 2;
  DMat<T, P> dm;
  for (int r = 0; r < P; ++r) dm.w[r] = w ? w[r] : T(0);
  for (int r = 0; r < PH; ++r)
    for (int m = 0; m < PC; ++m)
      dm.e[r * PC + m] = m < PH ? (d[r * P + m] + d[r * P + P - 1 - m]) / 2
                                : d[r * P + m];
  for (int r = 0; r < PC; ++r)
    for (int m = 0; m < PH; ++m)
      dm.o[r * PH + m] = r < PH ? (d[r * P + m] - d[r * P + P - 1 - m]) / 2
                                : d[r * P + m];
  return dm;
}

// y = D x (TRANS: y = D^T x) for one line of P values held in registers.
template <typename T, int P, bool TRANS>
__device__ __forceinline__ void line_apply(const DMat<T, P>& dm,
                                           const T (&x)[P], T (&y)[P]) {
  constexpr int PH = P / 2, PC = P - P / 2;
  T xe[PC], xo[PH > 0 ? PH : 1];
#pragma unroll
  for (int m = 0; m < PH; ++m) {
    xe[m] = x[m] + x[P - 1 - m];
    xo[m] = x[m] - x[P - 1 - m];
  }
  if (PC > PH) xe[PH] = x[PH];
#pragma unroll
  for (int r = 0; r < PC; ++r) {
    T se = T(0), so = T(0);
    if (r < PH) {
#pragma unroll
      for (int m = 0; m < PC; ++m)
        se += (TRANS ? dm.o[m * PH + r] : dm.e[r * PC + m]) * xe[m];
    }
#pragma unroll
    for (int m = 0; m < PH; ++m)
      so += (TRANS ? dm.e[m * PC + r] : dm.o[r * PH + m]) * xo[m];
    if (r < PH) {
      y[r] = so + se;
      y[P - 1 - r] = so - se;
    } else {
      y[r] = so;
    }
  }
}

// XCD-aware block -> work-group-of-elements map: blocks b and b+8 share an
// XCD (and its L2), so give every XCD one contiguous range of the mesh; the
// faces shared by consecutive elements are then re-read from the same L2.
__device__ __forceinline__ int64_t xcd_remap(int64_t b, int64_t nblocks) {
  const int64_t per = nblocks / 8, rem = nblocks % 8;
  const int64_t x = b % 8, k = b / 8;
  // XCD x owns `per + (x < rem)` consecutive groups
  const int64_t start = x * per + (x < rem ? x : rem);
  return start + k;
}

// GM (geometry mode): 0 = every element reads per-point factors, 1 = every
// element is affine, 2 = mixed (per element, via geo_index).
template <typename T, int P, int DIM, bool GS, bool SCALAR, int GM>
__global__ void __launch_bounds__((HelmholtzTile<T, P, DIM>::BLOCK),
                                  (HelmholtzTile<T, P, DIM>::MINW))
helmholtz_kernel(HelmholtzParams<T> prm, DMat<T, P> dm) {
  using Tile = HelmholtzTile<T, P, DIM>;
  constexpr int TPE = Tile::TPE, SA = Tile::SA, SB = Tile::SB;
  constexpr int EPB = Tile::EPB, W = Tile::ELEM_WORDS, NG = Tile::NGEO;
  constexpr int N = DIM == 3 ? P * P * P : P * P;        // nodes per element
  __shared__ T lds[2 * EPB * W];

  const int tid = threadIdx.x;
  const int el = tid / TPE;                 // element within the workgroup
  const int t = tid - el * TPE;             // lane within the element
  const int i = DIM == 3 ? t / P : 0;
  const int j = DIM == 3 ? t - i * P : t;
  const int64_t e0 = (int64_t)blockIdx.x * EPB;   // wave-uniform
  const bool lane_ok = el < EPB;            // tail lanes of a padded block
  const bool active = lane_ok && e0 + el < prm.num_elements;

  T* s0 = lds + (lane_ok ? el : 0) * 2 * W;    // becomes the axis-1 result
  T* s1 = s0 + W;                              // becomes the axis-2 result
  const DMat<T, P>& dmat = dm;
  // SCALAR: one component known at compile time (no component loop, so the
  // compiler has nothing to hoist out of it and spill)
  const int nc = SCALAR ? 1 : prm.ncomp;
  const bool has_mass = prm.lambda0 != T(0);
  const bool has_stiff = prm.lambda1 != T(0);

  // Owner layout: this lane holds nodes (a, i, j), a = 0..P-1, i.e. element
  // slots a*TPE + t.  All per-element arrays are addressed as a wave-uniform
  // base (SGPR pair) + one 32-bit per-lane byte offset + compile-time constant.
  // Affine elements (constant Jacobian) keep 7 numbers per ELEMENT instead of
  // per point: G(q) = w_q * (detJ J^-1 J^-T), W(q) = w_q detJ with w_q the
  // tensor quadrature weight.  `geo_index` maps the other elements to their
  // slot in the per-point array.
  const int64_t e_lane = e0 + (lane_ok ? el : 0);
  int64_t gslot = e_lane;
  if (GM == 2) gslot = active ? prm.geo_index[e_lane] : 0;
  const bool affine = GM == 1 || (GM == 2 && gslot < 0);
  T cg[GM == 0 ? 1 : 7];
  T wbc = T(0);
  if (GM != 0 && affine) {
    const T* ce = prm.geo_elem + e_lane * 8;
#pragma unroll
    for (int f = 0; f < 7; ++f) cg[f] = ce[f];
    // per-lane lookup by select chain (a runtime index into a by-value kernel
    // argument would force the struct into scratch)
    T wi = T(0), wj = T(0);
#pragma unroll
    for (int r = 0; r < P; ++r) {
      wi = i == r ? dm.w[r] : wi;
      wj = j == r ? dm.w[r] : wj;
    }
    wbc = DIM == 3 ? wi * wj : wj;
  }
  const char* geo0 = reinterpret_cast<const char*>(
      prm.geo + (GM == 1 || affine ? 0 : gslot) * (int64_t)(NG + 1) * N);
  uint32_t geo_off_v = (uint32_t)(t * sizeof(T));
  uint32_t slot_off_v = (uint32_t)(el * N + t);
  const uint32_t& geo_off = geo_off_v;
  const uint32_t& slot_off = slot_off_v;
  // Geometric factors are stored in pairs so that fp64 reads are 16 bytes per
  // lane (1 KiB per wave-instruction):
  //   3D: [pair 0..2][Q][2] = (G00,G01) (G02,G11) (G12,G22), then W [Q]
  //   2D: [pair 0..1][Q][2] = (G00,G01) (G11,W)
  // Each access is a wave-uniform (SGPR) base + one shared lane offset + a
  // small immediate.
  typedef T Pair __attribute__((ext_vector_type(2)));
  auto geo_pair = [&](int pi, int a) -> Pair {
    const char* base = geo0 + (size_t)pi * 2 * N * sizeof(T);
    const Pair* ptr = reinterpret_cast<const Pair*>(
        base + (2 * geo_off + (uint32_t)(2 * a * TPE * sizeof(T))));
    // streamed once: non-temporal so the factors do not evict the gathered
    // nodal values that neighbouring elements re-read from L2 / MALL
    return *ptr;
  };
  auto geo_mass = [&](int a) -> T {
    if (DIM == 2) return geo_pair(1, a).y;
    const char* base = geo0 + (size_t)6 * N * sizeof(T);
    return *reinterpret_cast<const T*>(
        base + (geo_off + (uint32_t)(a * TPE * sizeof(T))));
  };

  uint32_t enc[P];
  if (GS) {
    const int32_t* enc0 = prm.enc + e0 * N;
#pragma unroll
    for (int a = 0; a < P; ++a)
      enc[a] = active ? (uint32_t)enc0[slot_off + a * TPE]
                      : (uint32_t)SFEM_IDX_PAD;
  }
  const T* ul0 = GS ? nullptr : prm.u + e0 * N * nc;
  T* ol0 = GS ? nullptr : prm.out + e0 * N * nc;

  for (int k = 0; k < nc; ++k) {
    if (!SCALAR) {
      // keep address arithmetic and flag tests inside the component loop:
      // hoisted out of it they occupy ~100 registers and spill
      asm volatile("" : "+v"(geo_off_v), "+v"(slot_off_v));
      if (GM != 0) asm volatile("" : "+v"(wbc));
    }
    T ua[P], acc[P];
#pragma unroll
    for (int a = 0; a < P; ++a) {
      if (GS) {
        const uint32_t id = enc[a] & SFEM_IDX_MASK;
        ua[a] = id == SFEM_IDX_PAD ? T(0) : prm.u[(int64_t)id * nc + k];
      } else {
        ua[a] = active ? ul0[(slot_off + a * TPE) * nc + k] : T(0);
      }
    }
    if (has_stiff) {
      T d0[P];   // derivative along axis 0 at (a, i, j)
      line_apply<T, P, false>(dmat, ua, d0);
      if (lane_ok) {
#pragma unroll
        for (int a = 0; a < P; ++a) {
          s0[a * SA + i * SB + j] = ua[a];
          if (DIM == 3) s1[a * SA + i * SB + j] = ua[a];
        }
      }
      __syncthreads();
      if (lane_ok) {  // last axis: lane owns the line [i, j, *] (3D) / [j, *]
        T* line = (DIM == 3 ? s1 + i * SA + j * SB : s0 + j * SA);
        T x[P], y[P];
#pragma unroll
        for (int m = 0; m < P; ++m) x[m] = line[m];
        line_apply<T, P, false>(dmat, x, y);
#pragma unroll
        for (int m = 0; m < P; ++m) line[m] = y[m];
      }
      if (DIM == 3 && lane_ok) {  // middle axis: lane owns the line [i, *, j]
        T* line = s0 + i * SA + j;
        T x[P], y[P];
#pragma unroll
        for (int m = 0; m < P; ++m) x[m] = line[m * SB];
        line_apply<T, P, false>(dmat, x, y);
#pragma unroll
        for (int m = 0; m < P; ++m) line[m * SB] = y[m];
      }
      __syncthreads();
      // pointwise: w = G * (reference gradient), G symmetric
      T w0[P];
#pragma unroll
      for (int a = 0; a < P; ++a) w0[a] = T(0);
      if (active) {
#pragma unroll
        for (int a = 0; a < P; ++a) {
          const int o = a * SA + i * SB + j;
          if (DIM == 3) {
            const T g0 = d0[a], g1 = s0[o], g2 = s1[o];
            T G00 = T(0), G01 = T(0), G02 = T(0), G11 = T(0), G12 = T(0),
              G22 = T(0);
            if (GM != 0 && affine) {
              const T sc = wbc * dm.w[a];
              G00 = cg[0] * sc; G01 = cg[1] * sc; G02 = cg[2] * sc;
              G11 = cg[3] * sc; G12 = cg[4] * sc; G22 = cg[5] * sc;
            } else if (GM != 1) {
              const Pair p0 = geo_pair(0, a), p1 = geo_pair(1, a),
                         p2 = geo_pair(2, a);
              G00 = p0.x; G01 = p0.y; G02 = p1.x; G11 = p1.y;
              G12 = p2.x; G22 = p2.y;
            }
            w0[a] = G00 * g0 + G01 * g1 + G02 * g2;
            s0[o] = G01 * g0 + G11 * g1 + G12 * g2;
            s1[o] = G02 * g0 + G12 * g1 + G22 * g2;
          } else {
            const T g0 = d0[a], g1 = s0[o];
            T G00 = T(0), G01 = T(0), G11 = T(0);
            if (GM != 0 && affine) {
              const T sc = wbc * dm.w[a];
              G00 = cg[0] * sc; G01 = cg[1] * sc; G11 = cg[2] * sc;
            } else if (GM != 1) {
              const Pair p0 = geo_pair(0, a), p1 = geo_pair(1, a);
              G00 = p0.x; G01 = p0.y; G11 = p1.x;
            }
            w0[a] = G00 * g0 + G01 * g1;
            s0[o] = G01 * g0 + G11 * g1;
          }
        }
      }
      __syncthreads();
      if (lane_ok) {  // transposed derivative along the last axis, in place
        T* line = (DIM == 3 ? s1 + i * SA + j * SB : s0 + j * SA);
        T x[P], y[P];
#pragma unroll
        for (int m = 0; m < P; ++m) x[m] = line[m];
        line_apply<T, P, true>(dmat, x, y);
#pragma unroll
        for (int m = 0; m < P; ++m) line[m] = y[m];
      }
      if (DIM == 3 && lane_ok) {
        T* line = s0 + i * SA + j;
        T x[P], y[P];
#pragma unroll
        for (int m = 0; m < P; ++m) x[m] = line[m * SB];
        line_apply<T, P, true>(dmat, x, y);
#pragma unroll
        for (int m = 0; m < P; ++m) line[m * SB] = y[m];
      }
      line_apply<T, P, true>(dmat, w0, acc);
      __syncthreads();
      if (lane_ok) {
#pragma unroll
        for (int a = 0; a < P; ++a) {
          const int o = a * SA + i * SB + j;
          acc[a] += s0[o];
          if (DIM == 3) acc[a] += s1[o];
          acc[a] *= prm.lambda1;
        }
      }
    } else {
#pragma unroll
      for (int a = 0; a < P; ++a) acc[a] = T(0);
    }
    if (has_mass && active) {
#pragma unroll
      for (int a = 0; a < P; ++a)
        acc[a] += prm.lambda0 *
                  ((GM != 0 && affine) ? cg[GM == 0 ? 0 : 6] * wbc * dm.w[a]
                                       : (GM != 1 ? geo_mass(a) : T(0))) *
                  ua[a];
    }
    // direct-stiffness summation
#pragma unroll
    for (int a = 0; a < P; ++a) {
      if (GS) {
        uint32_t ea = enc[a];
        // keep the flag tests inside the component loop (hoisting them costs
        // 3 lane masks per a in SGPRs and spills the D matrix)
        asm volatile("" : "+v"(ea));
        const uint32_t id = ea & SFEM_IDX_MASK;
        if (id != SFEM_IDX_PAD) {
          T* dst = prm.out + (int64_t)id * nc + k;
          const bool dirichlet = ea & SFEM_IDX_DIRICHLET;
          if (ea & SFEM_IDX_SHARED) {
            if (!dirichlet) unsafeAtomicAdd(dst, acc[a]);
          } else {
            *dst = dirichlet ? T(0) : acc[a];
          }
        }
      } else if (active) {
        ol0[(slot_off + a * TPE) * nc + k] = acc[a];
      }
    }
    if (k + 1 < nc) __syncthreads();
  }
}

template <typename T, int P, int DIM, bool GS>
int launch_helmholtz(const HelmholtzParams<T>& prm, hipStream_t stream) {
  using Tile = HelmholtzTile<T, P, DIM>;
  const int64_t groups = (prm.num_elements + Tile::EPB - 1) / Tile::EPB;
  if (groups > 0x7fffffff) {
    set_error("helmholtz: too many workgroups (%lld)", (long long)groups);
    return SFEM_EINVAL;
  }
  const DMat<T, P> dm = make_dmat<T, P>(prm.dmat_host, prm.weights_host);
  const dim3 grid((unsigned)groups), block(Tile::BLOCK);
  const int gm = prm.geo_elem == nullptr ? 0 : (prm.geo == nullptr ? 1 : 2);
#define SFEM_LAUNCH_GM(SC, GMV)                                              \
  hipLaunchKernelGGL((helmholtz_kernel<T, P, DIM, GS, SC, GMV>), grid, block, \
                     0, stream, prm, dm)
  if (prm.ncomp == 1) {
    if (gm == 0) SFEM_LAUNCH_GM(true, 0);
    else if (gm == 1) SFEM_LAUNCH_GM(true, 1);
    else SFEM_LAUNCH_GM(true, 2);
  } else {
    if (gm == 0) SFEM_LAUNCH_GM(false, 0);
    else if (gm == 1) SFEM_LAUNCH_GM(false, 1);
    else SFEM_LAUNCH_GM(false, 2);
  }
#undef SFEM_LAUNCH_GM
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

// Defined once per (dtype, ndim) translation unit (compiled in parallel).
template <typename T, int DIM>
int dispatch_helmholtz(const HelmholtzParams<T>& prm, int P, bool gs,
                       hipStream_t stream);

#define SFEM_HELMHOLTZ_CASE(PP)                                             \
  case PP:                                                                  \
    return gs ? launch_helmholtz<T, PP, DIM, true>(prm, stream)             \
              : launch_helmholtz<T, PP, DIM, false>(prm, stream);

#define SFEM_DEFINE_HELMHOLTZ_DISPATCH(TYPE, DIMV)                          \
  template <>                                                               \
  int dispatch_helmholtz<TYPE, DIMV>(const HelmholtzParams<TYPE>& prm,      \
                                     int P, bool gs, hipStream_t stream) {  \
    using T = TYPE;                                                         \
    constexpr int DIM = DIMV;                                               \
    switch (P) {                                                            \
      SFEM_HELMHOLTZ_CASE(2) SFEM_HELMHOLTZ_CASE(3) SFEM_HELMHOLTZ_CASE(4)  \
      SFEM_HELMHOLTZ_CASE(5) SFEM_HELMHOLTZ_CASE(6) SFEM_HELMHOLTZ_CASE(7)  \
      SFEM_HELMHOLTZ_CASE(8) SFEM_HELMHOLTZ_CASE(9) SFEM_HELMHOLTZ_CASE(10) \
      SFEM_HELMHOLTZ_CASE(11) SFEM_HELMHOLTZ_CASE(12)                       \
      default:                                                              \
        set_error("helmholtz: P=%d outside the compiled range 2..12", P);   \
        return SFEM_EUNSUPPORTED;                                           \
    }                                                                       \
  }

}  // namespace sfem
