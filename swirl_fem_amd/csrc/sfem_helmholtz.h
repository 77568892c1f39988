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

#ifndef SFEM_DMAT_MEM
#define SFEM_DMAT_MEM 1
#endif
#ifndef SFEM_PK_F32
#define SFEM_PK_F32 1
#endif
#ifndef SFEM_KERNARG_PICK
#define SFEM_KERNARG_PICK 1
#endif
#ifndef SFEM_CL_SWIZZLE
#define SFEM_CL_SWIZZLE 1
#endif
#ifndef SFEM_CL_SCOPE
#define SFEM_CL_SCOPE "wavefront"
#endif
namespace sfem {

template <typename T>
struct HelmholtzParams {
  const T* u;            // (N, nc) or (E, n, nc) when !GS
  T* out;
  const int32_t* enc;    // (E, n) encoded indices (GS only)
  const T* geo;          // per-point factors (elements with geo_index >= 0)
  const T* geo_elem;     // (E, 24) multilinear map coefficients, or null
  const int32_t* geo_index;  // (E,) slot in `geo`, -1 affine, -2 multilinear
  int geo_mode;          // GeoMode
  const T* dmat_host;    // (P, P) on the HOST; travels as a kernel argument
  const T* weights_host; // (P,) quadrature weights on the HOST
  const T* nodes_host;   // (P,) 1D node values on the HOST
  int64_t num_elements;  // elements processed by this launch
  const int32_t* elem_list;  // their ids, or null = 0..num_elements-1
  int ncomp;             // components handled inside the kernel
  int64_t node_stride;   // u / out: element offset = node * node_stride +
  int64_t comp_stride;   //                           component * comp_stride
  int comp;              // first component
  T lambda0, lambda1;
  double* dot_out;       // SFEM_DOT_SLOTS partial sums of u . out, or null
  int colored;           // launches are conflict-free colour classes: SHARED
                         // slots read-modify-write instead of atomics
  // (E, shared_stride) slots of each element's SHARED, non-Dirichlet nodes in
  // ascending node order, 0xFFFF padded; null = scatter in slot order
  const uint16_t* shared_order;
  int shared_stride;
};

__host__ __device__ constexpr int round_up(int a, int b) {
  return (a + b - 1) / b * b;
}

// PAD: rows of the LDS tensor padded to an odd length (no bank conflicts in the
// transposes).  Unpadded rows save 1/9 of the LDS at P = 8: 20 instead of 17
// one-wave workgroups per CU, which pays where the registers allow 5 waves per
// SIMD (the affine kernels: 0.765 -> 0.730 ms) and costs 1 % where they do
// not (stored factors), hence a per-kernel choice.
template <typename T, int P, int DIM, bool PAD = true>
struct HelmholtzTile {
  static constexpr int TPE = DIM == 3 ? P * P : P;       // lanes per element
  static constexpr int SB = PAD ? (P | 1) : P;           // row stride
  static constexpr int SA = DIM == 3 ? P * SB : SB;      // axis-0 stride
  static constexpr int ELEM_WORDS = P * SA;              // one padded tensor
  static constexpr int LDS_PER_ELEM = 2 * ELEM_WORDS * (int)sizeof(T);
  // elements per workgroup (TPE < 64): fill whole waves, stay under ~40 KiB of
  // LDS and 512 threads; a single wave when an element divides a wave evenly.
  static constexpr int pick_epb() {
    if (64 % TPE == 0) return 64 / TPE;
    // More than half a wave (P = 6, 7 in 3D): one element per one-wave
    // workgroup -- idle lanes cost less than barriers across 4-7 waves
    // (measured +13..24 % at p = 5, 6).
    if (2 * TPE > 64 && TPE < 64) return 1;
    // An element that already spans several waves gets its own workgroup:
    // packing two (P = 12: 288 of 320 lanes) fills lanes better but couples 5
    // waves at every barrier; measured 26-32 % slower (p = 9, 11 fp32).
    if (TPE > 64) return 1;
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
  static constexpr int MINW = P <= 8 ? 4 : 2;
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
  T w[P];         // 1D quadrature weights
  T x[P];         // 1D node values (on-the-fly geometry)
};

template <typename T, int P>
inline DMat<T, P> make_dmat(const T* d, const T* w, const T* x) {
  constexpr int PH = P / 2, PC = P - P / 2;
  DMat<T, P> dm;
  for (int r = 0; r < P; ++r) {
    dm.w[r] = w ? w[r] : T(0);
    dm.x[r] = x ? x[r] : T(0);
  }
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
#if SFEM_PK_F32
template <int P, bool TRANS>
__device__ __forceinline__ void line_apply_pk(const DMat<float, P>& dm,
                                              const float (&x)[P],
                                              float (&y)[P]);
#endif

template <typename T, int P, bool TRANS>
__device__ __forceinline__ void line_apply(const DMat<T, P>& dm,
                                           const T (&x)[P], T (&y)[P]) {
#if SFEM_PK_F32
  // fp32, one-wave elements: packed arithmetic, see below (measured at 64^3 /
  // 48^3: p = 7 apply 0.605 -> 0.568 ms; p = 11 1.03 -> 1.09 ms although it has
  // 21 % fewer instructions and one more wave per SIMD, so P > 8 keeps the
  // scalar form)
  if constexpr (sizeof(T) == 4 && P <= 8) {
    line_apply_pk<P, TRANS>(dm, x, y);
    return;
  }
#endif
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

// The same product with the matrix read from MEMORY (the kernarg segment, a
// constant address space: scalar loads, scalar-cache hits) instead of from the
// by-value copy.  For P >= 9 the 72-entry even/odd matrix plus weights do not
// fit the 102 SGPRs next to everything else, and the compiler spills scalars
// into VGPR lanes: 250 v_readlane / v_writelane per wave at P = 12, 18 % of the
// vector instructions of a kernel that is bound by them.  Reloading the entries
// per product keeps them in SGPRs only while they are used.
#if defined(__HIP_DEVICE_COMPILE__)
#define SFEM_CONSTANT_AS __attribute__((address_space(4)))
#else
#define SFEM_CONSTANT_AS
#endif
template <typename T, int P, bool TRANS>
__device__ __forceinline__ void line_apply_mem(
    const SFEM_CONSTANT_AS DMat<T, P>* km, const T (&x)[P], T (&y)[P]) {
  constexpr int PH = P / 2, PC = P - P / 2;
  // a fresh pointer per product: loads are not merged across products (which
  // would bring the register pressure back)
  asm volatile("" : "+s"(km));
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
        se += (TRANS ? km->o[m * PH + r] : km->e[r * PC + m]) * xe[m];
    }
#pragma unroll
    for (int m = 0; m < PH; ++m)
      so += (TRANS ? km->e[m * PC + r] : km->o[r * PH + m]) * xo[m];
    if (r < PH) {
      y[r] = so + se;
      y[P - 1 - r] = so - se;
    } else {
      y[r] = so;
    }
  }
}

#if SFEM_PK_F32
// fp32: the same sums with packed arithmetic (v_pk_fma_f32: two fp32 FMAs per
// lane per instruction, fewer instructions to issue).
// Matrix entries that sit next to each other in the by-value DMat form the
// packed scalar operand: for y = D x two consecutive m of one row (two partial
// sums per row, added at the end), for y = D^T x two consecutive rows.
typedef float pk_f2 __attribute__((ext_vector_type(2)));

template <int P, bool TRANS>
__device__ __forceinline__ void line_apply_pk(const DMat<float, P>& dm,
                                              const float (&x)[P],
                                              float (&y)[P]) {
  constexpr int PH = P / 2, PC = P - P / 2;
  float xe[PC], xo[PH > 0 ? PH : 1];
#pragma unroll
  for (int m = 0; m < PH; ++m) {
    xe[m] = x[m] + x[P - 1 - m];
    xo[m] = x[m] - x[P - 1 - m];
  }
  if (PC > PH) xe[PH] = x[PH];
  float se[PH > 0 ? PH : 1], so[PC];
  if (!TRANS) {
    // se[r] = sum_m e[r PC + m] xe[m], so[r] = sum_m o[r PH + m] xo[m]
#pragma unroll
    for (int r = 0; r < PH; ++r) {
      pk_f2 acc = {0.f, 0.f};
#pragma unroll
      for (int m = 0; m + 1 < PC; m += 2) {
        const pk_f2 a = {dm.e[r * PC + m], dm.e[r * PC + m + 1]};
        const pk_f2 b = {xe[m], xe[m + 1]};
        acc = __builtin_elementwise_fma(a, b, acc);
      }
      se[r] = acc.x + acc.y;
      if (PC & 1) se[r] += dm.e[r * PC + PC - 1] * xe[PC - 1];
    }
#pragma unroll
    for (int r = 0; r < PC; ++r) {
      pk_f2 acc = {0.f, 0.f};
#pragma unroll
      for (int m = 0; m + 1 < PH; m += 2) {
        const pk_f2 a = {dm.o[r * PH + m], dm.o[r * PH + m + 1]};
        const pk_f2 b = {xo[m], xo[m + 1]};
        acc = __builtin_elementwise_fma(a, b, acc);
      }
      so[r] = acc.x + acc.y;
      if (PH & 1) so[r] += dm.o[r * PH + PH - 1] * xo[PH - 1];
    }
  } else {
    // se[r] = sum_m o[m PH + r] xe[m], so[r] = sum_m e[m PC + r] xo[m]
#pragma unroll
    for (int r = 0; r + 1 < PH; r += 2) {
      pk_f2 acc = {0.f, 0.f};
#pragma unroll
      for (int m = 0; m < PC; ++m) {
        const pk_f2 a = {dm.o[m * PH + r], dm.o[m * PH + r + 1]};
        const pk_f2 b = {xe[m], xe[m]};
        acc = __builtin_elementwise_fma(a, b, acc);
      }
      se[r] = acc.x;
      se[r + 1] = acc.y;
    }
    if (PH & 1) {
      float acc = 0.f;
#pragma unroll
      for (int m = 0; m < PC; ++m) acc += dm.o[m * PH + PH - 1] * xe[m];
      se[PH - 1] = acc;
    }
#pragma unroll
    for (int r = 0; r + 1 < PC; r += 2) {
      pk_f2 acc = {0.f, 0.f};
#pragma unroll
      for (int m = 0; m < PH; ++m) {
        const pk_f2 a = {dm.e[m * PC + r], dm.e[m * PC + r + 1]};
        const pk_f2 b = {xo[m], xo[m]};
        acc = __builtin_elementwise_fma(a, b, acc);
      }
      so[r] = acc.x;
      so[r + 1] = acc.y;
    }
    if (PC & 1) {
      float acc = 0.f;
#pragma unroll
      for (int m = 0; m < PH; ++m) acc += dm.e[m * PC + PC - 1] * xo[m];
      so[PC - 1] = acc;
    }
  }
#pragma unroll
  for (int r = 0; r < PC; ++r) {
    if (r < PH) {
      y[r] = so[r] + se[r];
      y[P - 1 - r] = so[r] - se[r];
    } else {
      y[r] = so[r];
    }
  }
}
#endif

// Geometry modes.  The symmetric factors G = w detJ (J^-1 J^-T) (and W = w detJ)
// of a quadrature point come from one of three sources:
//   GEO_POINT        6 (+1) stored values per point, streamed from HBM
//   GEO_MULTILINEAR  the element is the image of the reference cube under a
//                    multilinear map (every mesh produced by refine_premesh
//                    is): 7 coefficient vectors per ELEMENT; the Jacobian,
//                    its adjugate and G are evaluated in registers per point
//                    (~65 flops) -- bytes traded for flops on a kernel whose
//                    VALU is ~25 % busy
//   GEO_AFFINE       multilinear with constant Jacobian: G is a per-element
//                    constant times the tensor quadrature weight
// A mesh that mixes the three kinds is processed by up to three launches, each
// over its own element list (`elem_list`), so every kernel stays specialised.
enum GeoMode { GEO_POINT = 0, GEO_AFFINE = 1, GEO_MULTILINEAR = 3 };

// w / d for the per-point geometric factors: hardware reciprocal estimate + two
// Newton steps (5 instructions) instead of the IEEE division sequence (~14 for
// fp64: v_div_scale x2, v_rcp, 6 fma, v_div_fmas, v_div_fixup).  Relative error
// < 2^-50 for normal d (the estimate is good to >= 2^-20), far inside the 1e-10
// parity bound; a Jacobian determinant is never 0 / inf / denormal on a valid
// element, the only inputs the full sequence treats differently.
__device__ __forceinline__ double fast_div(double w, double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  return w * r;
}
__device__ __forceinline__ float fast_div(float w, float d) {
  float r = __builtin_amdgcn_rcpf(d);
  r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
  return w * r;
}

template <typename T, int P>
__device__ __forceinline__ T lane_pick(const T (&v)[P], int idx) {
  // runtime index into a by-value kernel argument would push the struct to
  // scratch; a select chain keeps it in SGPRs
  T r = T(0);
#pragma unroll
  for (int q = 0; q < P; ++q) r = idx == q ? v[q] : r;
  return r;
}

// Per-lane geometry state of one element.
template <typename T, int P, int DIM, int GM>
struct ElemGeom {
  static constexpr int NG = DIM == 3 ? 6 : 3;
  static constexpr int NPT = DIM == 3 ? P * P * P : P * P;
  static constexpr int TPE = DIM == 3 ? P * P : P;
  static constexpr bool HAS_POINT = GM == GEO_POINT;
  static constexpr bool HAS_AFFINE = GM == GEO_AFFINE;
  static constexpr bool HAS_MULTI = GM == GEO_MULTILINEAR;
  typedef T Pair __attribute__((ext_vector_type(2)));

  const char* base;    // per-point factors of this element
  uint32_t lane_off;   // byte offset of this lane inside one factor plane
  T wbc;               // product of the in-plane quadrature weights
  T cst[HAS_AFFINE ? 7 : 1];       // affine: G upper triangle / w, det
  // multilinear: rows of the Jacobian are R0 (constant along the line),
  // R1 = p1 + r q1, R2 = p2 + r q2 with r the axis-0 node coordinate
  T r0[HAS_MULTI ? DIM : 1], p1[HAS_MULTI ? DIM : 1], q1[HAS_MULTI ? DIM : 1],
      p2[HAS_MULTI && DIM == 3 ? 3 : 1], q2[HAS_MULTI && DIM == 3 ? 3 : 1];

  __device__ __forceinline__ bool is_affine() const { return HAS_AFFINE; }
  __device__ __forceinline__ bool is_multi() const { return HAS_MULTI; }

  // `mem`: the same DMat in memory (a kernel that takes it as its FIRST
  // argument passes the kernarg segment): the per-lane weights and nodes are
  // then four small cached loads instead of four 8-deep select chains
  // (64 v_cndmask for fp64); null = select from the by-value copy.
  template <bool MEM = false>
  __device__ __forceinline__ void init(const HelmholtzParams<T>& prm,
                                       const DMat<T, P>& dm, int64_t e,
                                       bool active, int i, int j, int t,
                                       const DMat<T, P>* mem = nullptr) {
    int64_t slot = e;
    if (HAS_POINT && prm.geo_index) slot = active ? prm.geo_index[e] : 0;
    base = reinterpret_cast<const char*>(prm.geo) +
           (HAS_POINT ? slot * (int64_t)(NG + 1) * NPT * sizeof(T) : 0);
    lane_off = (uint32_t)(t * sizeof(T));
    wbc = T(0);
    if (GM == GEO_POINT || !active) return;
    if (HAS_AFFINE || HAS_MULTI) {
      {
        const T wj = MEM ? mem->w[j] : lane_pick<T, P>(dm.w, j);
        wbc = DIM == 3 ? (MEM ? mem->w[i] : lane_pick<T, P>(dm.w, i)) * wj : wj;
        const T* A = prm.geo_elem + e * 24;   // A1..A7 (3D) / A1..A3 (2D)
        if (DIM == 3) {
          const T s = MEM ? mem->x[i] : lane_pick<T, P>(dm.x, i);
          const T tt = MEM ? mem->x[j] : lane_pick<T, P>(dm.x, j);
          T a0[3], a1[3], a2[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const T A1 = A[c], A2 = A[3 + c], A3 = A[6 + c], A4 = A[9 + c],
                    A5 = A[12 + c], A6 = A[15 + c], A7 = A[18 + c];
            // d/dr = A1 + A4 s + A6 t + A7 s t;  d/ds = (A2 + A5 t) + r (A4 + A7 t)
            // d/dt = (A3 + A5 s) + r (A6 + A7 s)
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
          if (HAS_AFFINE && is_affine()) {
            // constant Jacobian rows a0, a1, a2
            T c0[3] = {a1[1] * a2[2] - a1[2] * a2[1],
                       a1[2] * a2[0] - a1[0] * a2[2],
                       a1[0] * a2[1] - a1[1] * a2[0]};
            T c1[3] = {a2[1] * a0[2] - a2[2] * a0[1],
                       a2[2] * a0[0] - a2[0] * a0[2],
                       a2[0] * a0[1] - a2[1] * a0[0]};
            T c2[3] = {a0[1] * a1[2] - a0[2] * a1[1],
                       a0[2] * a1[0] - a0[0] * a1[2],
                       a0[0] * a1[1] - a0[1] * a1[0]};
            const T det = a0[0] * c0[0] + a0[1] * c0[1] + a0[2] * c0[2];
            const T inv = T(1) / det;
            cst[0] = inv * (c0[0] * c0[0] + c0[1] * c0[1] + c0[2] * c0[2]);
            cst[1] = inv * (c0[0] * c1[0] + c0[1] * c1[1] + c0[2] * c1[2]);
            cst[2] = inv * (c0[0] * c2[0] + c0[1] * c2[1] + c0[2] * c2[2]);
            cst[3] = inv * (c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2]);
            cst[4] = inv * (c1[0] * c2[0] + c1[1] * c2[1] + c1[2] * c2[2]);
            cst[5] = inv * (c2[0] * c2[0] + c2[1] * c2[1] + c2[2] * c2[2]);
            cst[6] = det;
          }
        } else {
          const T s = lane_pick<T, P>(dm.x, j);   // axis-1 coordinate
          T a0[2], a1[2];
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const T A1 = A[c], A2 = A[2 + c], A3 = A[4 + c];
            a0[c] = A1 + A3 * s;      // d/dr, constant along the line
            a1[c] = A2;               // d/ds = A2 + r A3
            if (HAS_MULTI) { r0[c] = a0[c]; p1[c] = A2; q1[c] = A3; }
          }
          if (HAS_AFFINE && is_affine()) {
            const T det = a0[0] * a1[1] - a0[1] * a1[0];
            const T inv = T(1) / det;
            // columns of the adjugate: C0 = (d, -c), C1 = (-b, a)
            cst[0] = inv * (a1[1] * a1[1] + a1[0] * a1[0]);
            cst[1] = -inv * (a1[1] * a0[1] + a1[0] * a0[0]);
            cst[2] = inv * (a0[1] * a0[1] + a0[0] * a0[0]);
            cst[6] = det;
          }
        }
      }
    }
  }

  __device__ __forceinline__ Pair pair_at(int pi, int a) const {
    const char* b = base + (size_t)pi * 2 * NPT * sizeof(T);
    return *reinterpret_cast<const Pair*>(
        b + (2 * lane_off + (uint32_t)(2 * a * TPE * sizeof(T))));
  }

  // Multilinear 3D: (w / det) C C^T g with C = rows of cofactors, applied as
  // C (C^T g) without forming the six G entries (21 instead of 39 operations
  // per point).  Returns the three components and, optionally, W = w det.
  __device__ __forceinline__ void apply_multilinear3(
      const DMat<T, P>& dm, int a, bool want_w, T g0, T g1, T g2, T& o0, T& o1,
      T& o2, T& Wm) const {
    const T r = dm.x[a];
    const T wq = wbc * dm.w[a];
    T R1[3], R2[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      R1[c] = p1[c] + r * q1[c];
      R2[c] = p2[c] + r * q2[c];
    }
    const T c0[3] = {R1[1] * R2[2] - R1[2] * R2[1],
                     R1[2] * R2[0] - R1[0] * R2[2],
                     R1[0] * R2[1] - R1[1] * R2[0]};
    const T c1[3] = {R2[1] * r0[2] - R2[2] * r0[1],
                     R2[2] * r0[0] - R2[0] * r0[2],
                     R2[0] * r0[1] - R2[1] * r0[0]};
    const T c2[3] = {r0[1] * R1[2] - r0[2] * R1[1],
                     r0[2] * R1[0] - r0[0] * R1[2],
                     r0[0] * R1[1] - r0[1] * R1[0]};
    const T det = r0[0] * c0[0] + r0[1] * c0[1] + r0[2] * c0[2];
    Wm = want_w ? wq * det : T(0);
    const T sc = fast_div(wq, det);
    T y[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
      y[k] = sc * (c0[k] * g0 + c1[k] * g1 + c2[k] * g2);
    o0 = c0[0] * y[0] + c0[1] * y[1] + c0[2] * y[2];
    o1 = c1[0] * y[0] + c1[1] * y[1] + c1[2] * y[2];
    o2 = c2[0] * y[0] + c2[1] * y[1] + c2[2] * y[2];
  }

  // Factors at the lane's node of slice a: G[0..NG) upper triangle, W.
  // want_g / want_w are compile-time after inlining in the callers.
  __device__ __forceinline__ void factors(const DMat<T, P>& dm, int a,
                                          bool want_g, bool want_w,
                                          T (&G)[6], T& Wm) const {
#pragma unroll
    for (int f = 0; f < 6; ++f) G[f] = T(0);
    Wm = T(0);
    if (HAS_AFFINE && is_affine()) {
      const T sc = wbc * dm.w[a];
      if (want_g) {
        if (DIM == 3) {
#pragma unroll
          for (int f = 0; f < 6; ++f) G[f] = cst[f] * sc;
        } else {
          G[0] = cst[0] * sc; G[1] = cst[1] * sc; G[3] = cst[2] * sc;
        }
      }
      if (want_w) Wm = cst[6] * sc;
      return;
    }
    if (HAS_MULTI && is_multi()) {
      const T r = dm.x[a];
      const T wq = wbc * dm.w[a];
      if (DIM == 3) {
        T R1[3], R2[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          R1[c] = p1[c] + r * q1[c];
          R2[c] = p2[c] + r * q2[c];
        }
        const T c0[3] = {R1[1] * R2[2] - R1[2] * R2[1],
                         R1[2] * R2[0] - R1[0] * R2[2],
                         R1[0] * R2[1] - R1[1] * R2[0]};
        const T det = r0[0] * c0[0] + r0[1] * c0[1] + r0[2] * c0[2];
        if (want_w) Wm = wq * det;
        if (want_g) {
          const T c1[3] = {R2[1] * r0[2] - R2[2] * r0[1],
                           R2[2] * r0[0] - R2[0] * r0[2],
                           R2[0] * r0[1] - R2[1] * r0[0]};
          const T c2[3] = {r0[1] * R1[2] - r0[2] * R1[1],
                           r0[2] * R1[0] - r0[0] * R1[2],
                           r0[0] * R1[1] - r0[1] * R1[0]};
          const T sc = fast_div(wq, det);
          G[0] = sc * (c0[0] * c0[0] + c0[1] * c0[1] + c0[2] * c0[2]);
          G[1] = sc * (c0[0] * c1[0] + c0[1] * c1[1] + c0[2] * c1[2]);
          G[2] = sc * (c0[0] * c2[0] + c0[1] * c2[1] + c0[2] * c2[2]);
          G[3] = sc * (c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2]);
          G[4] = sc * (c1[0] * c2[0] + c1[1] * c2[1] + c1[2] * c2[2]);
          G[5] = sc * (c2[0] * c2[0] + c2[1] * c2[1] + c2[2] * c2[2]);
        }
      } else {
        const T R1x = p1[0] + r * q1[0], R1y = p1[1] + r * q1[1];
        const T det = r0[0] * R1y - r0[1] * R1x;
        if (want_w) Wm = wq * det;
        if (want_g) {
          const T sc = fast_div(wq, det);
          G[0] = sc * (R1y * R1y + R1x * R1x);
          G[1] = -sc * (R1y * r0[1] + R1x * r0[0]);
          G[3] = sc * (r0[1] * r0[1] + r0[0] * r0[0]);
        }
      }
      return;
    }
    if (HAS_POINT) {
      // stored in pairs so that fp64 reads are 16 bytes per lane:
      //   3D: (G00,G01) (G02,G11) (G12,G22), then W;  2D: (G00,G01) (G11,W)
      if (DIM == 3) {
        if (want_g) {
          const Pair p0 = pair_at(0, a), pa = pair_at(1, a), pb = pair_at(2, a);
          G[0] = p0.x; G[1] = p0.y; G[2] = pa.x; G[3] = pa.y;
          G[4] = pb.x; G[5] = pb.y;
        }
        if (want_w) {
          const char* b = base + (size_t)6 * NPT * sizeof(T);
          Wm = *reinterpret_cast<const T*>(
              b + (lane_off + (uint32_t)(a * TPE * sizeof(T))));
        }
      } else {
        const Pair pa = pair_at(1, a);
        if (want_g) {
          const Pair p0 = pair_at(0, a);
          G[0] = p0.x; G[1] = p0.y; G[3] = pa.x;
        }
        Wm = pa.y;
      }
    }
  }
};

// The DMat a kernel received as its FIRST argument, as memory (kernarg segment,
// offset 0): lane-indexed reads of its weights / nodes become cached loads.
template <typename T, int P>
__device__ __forceinline__ const DMat<T, P>* kernarg_dmat(int offset = 0) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (const DMat<T, P>*)((const __attribute__((address_space(4))) char*)
                                 __builtin_amdgcn_kernarg_segment_ptr() +
                             offset);
#else
  return nullptr;
#endif
}

// Workgroup barrier that orders LDS traffic ONLY.  __syncthreads() carries a
// workgroup-scope fence over all address spaces, i.e. s_waitcnt vmcnt(0): every
// wave then also waits for the acknowledgement of its outstanding global stores
// and atomics (~4 k cycles under load) although the barrier only protects LDS.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Orders LDS traffic among the lanes of ONE wave: no instruction (LDS executes
// a wave's operations in order), only a fence for the compiler.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, SFEM_CL_SCOPE);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, SFEM_CL_SCOPE);
}

// Position of node (a, r, c) of an element's P^3 tensor inside its LDS copy,
// for the three ways the kernel walks it:
//   own(a)    lane (i, j) touches (a, i, j)      a = compile-time index
//   last(m)   lane (i, j) touches (i, j, m)      line along the last axis
//   mid(m)    lane (i, j) touches (i, m, j)      line along the middle axis
// P = 8: rows are NOT padded (8 KB per fp64 element instead of 9.2: that is
// what lets two 8-element workgroups share a CU); instead
//   phys(a, r, c) = 64 a + 8 (r ^ (a & 3)) + (c ^ r)
// which makes the 32 lanes of a ds_read_b64 / ds_write_b64 group hit 32
// different 8-byte banks in all three patterns (unswizzled: 4- and 8-way
// conflicts, the LDS pipe then bounds the kernel).  The XOR constants reduce
// to one v_xor_b32 per access: own(a) = 64 a + L1[a & 3], last(m) = L2 ^ m,
// mid(m) = L3 ^ 9 m.  Other P: rows padded to an odd length as before.
template <typename T, int P>
struct ClusterLayout {
  static constexpr bool SWIZZLE = P == 8 && SFEM_CL_SWIZZLE;
  static constexpr int SB = SWIZZLE ? P : (P | 1);
  static constexpr int SA = P * SB;
  static constexpr int WORDS = P * SA;
  static constexpr int S = (int)sizeof(T);
  // BYTE offsets from the start of the workgroup's LDS array (which is 512-byte
  // aligned, as is every element's pair of copies: the XORs below then never
  // reach the bits of the base and one v_xor_b32 per access is all it costs)
  char* lds;
  uint32_t b1[SWIZZLE ? 4 : 1], b2, b3;
  __device__ __forceinline__ void init(T* lds_base, int el, int i, int j) {
    lds = reinterpret_cast<char*>(lds_base);
    const uint32_t base = (uint32_t)el * 2 * WORDS * S;   // s0; s1 = + WORDS*S
    if (SWIZZLE) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        b1[q] = base + S * (((i ^ q) << 3) + (j ^ i));
      b2 = base + WORDS * S + S * ((i << 6) + ((j ^ (i & 3)) << 3) + j);
      b3 = base + S * ((i << 6) + ((i & 3) << 3) + j);
    } else {
      b1[0] = base + S * (i * SB + j);
      b2 = base + WORDS * S + S * (i * SA + j * SB);
      b3 = base + S * (i * SA + j);
    }
  }
  __device__ __forceinline__ T& at(uint32_t off) const {
    return *reinterpret_cast<T*>(lds + off);
  }
  // copy 0 / copy 1 at the lane's own node of slice a
  __device__ __forceinline__ T& own0(int a) const {
    return at(SWIZZLE ? b1[a & 3] + a * 64 * S : b1[0] + a * SA * S);
  }
  __device__ __forceinline__ T& own1(int a) const {
    return at((SWIZZLE ? b1[a & 3] + a * 64 * S : b1[0] + a * SA * S) +
              WORDS * S);
  }
  // copy 1 along the last axis, copy 0 along the middle axis
  __device__ __forceinline__ T& last1(int m) const {
    return at(SWIZZLE ? (b2 ^ (uint32_t)(m * S)) : b2 + m * S);
  }
  __device__ __forceinline__ T& mid0(int m) const {
    return at(SWIZZLE ? (b3 ^ (uint32_t)(9 * m * S)) : b3 + m * SB * S);
  }
};

// (lambda0 B + lambda1 A)_local of one element held by (part of) one wave:
// ua[a] = nodal values of the lane's line, acc[a] = result.  s0 / s1: the
// element's two LDS copies.  Same arithmetic as helmholtz_kernel.
template <typename T, int P, int GM, bool MASS>
__device__ __forceinline__ void cluster_element_apply(
    const HelmholtzParams<T>& prm, const DMat<T, P>& dm,
    const ElemGeom<T, P, 3, GM>& geom, const ClusterLayout<T, P>& lay,
    bool lane_ok, bool active, const T (&ua)[P], T (&acc)[P]) {
  const bool has_stiff = prm.lambda1 != T(0);
  if (has_stiff) {
    T d0[P];
    line_apply<T, P, false>(dm, ua, d0);
    if (lane_ok) {
#pragma unroll
      for (int a = 0; a < P; ++a) {
        lay.own0(a) = ua[a];
        lay.own1(a) = ua[a];
      }
    }
    wave_sync();
    if (lane_ok) {   // last axis: the line [i, j, *]
      T x[P], y[P];
#pragma unroll
      for (int m = 0; m < P; ++m) x[m] = lay.last1(m);
      line_apply<T, P, false>(dm, x, y);
#pragma unroll
      for (int m = 0; m < P; ++m) lay.last1(m) = y[m];
    }
    if (lane_ok) {   // middle axis: the line [i, *, j]
      T x[P], y[P];
#pragma unroll
      for (int m = 0; m < P; ++m) x[m] = lay.mid0(m);
      line_apply<T, P, false>(dm, x, y);
#pragma unroll
      for (int m = 0; m < P; ++m) lay.mid0(m) = y[m];
    }
    wave_sync();
    T w0[P];
#pragma unroll
    for (int a = 0; a < P; ++a) { w0[a] = T(0); acc[a] = T(0); }
    if (active) {
#pragma unroll
      for (int a = 0; a < P; ++a) {
        T& r0 = lay.own0(a);
        T& r1 = lay.own1(a);
        T G[6], Wm;
        constexpr bool FUSE_W = GM != GEO_POINT;
        if constexpr (GM == GEO_MULTILINEAR) {
          T o0, o1, o2;
          geom.apply_multilinear3(dm, a, MASS, d0[a], r0, r1, o0, o1, o2, Wm);
          if (MASS) acc[a] = prm.lambda0 * Wm * ua[a];
          w0[a] = o0; r0 = o1; r1 = o2;
          continue;
        }
        geom.factors(dm, a, true, FUSE_W && MASS, G, Wm);
        if (FUSE_W && MASS) acc[a] = prm.lambda0 * Wm * ua[a];
        const T g0 = d0[a], g1 = r0, g2 = r1;
        w0[a] = G[0] * g0 + G[1] * g1 + G[2] * g2;
        r0 = G[1] * g0 + G[3] * g1 + G[4] * g2;
        r1 = G[2] * g0 + G[4] * g1 + G[5] * g2;
      }
    }
    wave_sync();
    if (lane_ok) {   // transposed derivative along the last axis, in place
      T x[P], y[P];
#pragma unroll
      for (int m = 0; m < P; ++m) x[m] = lay.last1(m);
      line_apply<T, P, true>(dm, x, y);
#pragma unroll
      for (int m = 0; m < P; ++m) lay.last1(m) = y[m];
    }
    if (lane_ok) {
      T x[P], y[P];
#pragma unroll
      for (int m = 0; m < P; ++m) x[m] = lay.mid0(m);
      line_apply<T, P, true>(dm, x, y);
#pragma unroll
      for (int m = 0; m < P; ++m) lay.mid0(m) = y[m];
    }
    T dt0[P];
    line_apply<T, P, true>(dm, w0, dt0);
    wave_sync();
    if (lane_ok) {
#pragma unroll
      for (int a = 0; a < P; ++a) {
        acc[a] += prm.lambda1 * (dt0[a] + lay.own0(a) + lay.own1(a));
      }
    }
    wave_sync();   // the copies are free again (next component)
    if (GM == GEO_POINT && MASS && active) {
#pragma unroll
      for (int a = 0; a < P; ++a) {
        T G[6], Wm;
        geom.factors(dm, a, false, true, G, Wm);
        acc[a] += prm.lambda0 * Wm * ua[a];
      }
    }
  } else {
#pragma unroll
    for (int a = 0; a < P; ++a) acc[a] = T(0);
    if (MASS && active) {
#pragma unroll
      for (int a = 0; a < P; ++a) {
        T G[6], Wm;
        geom.factors(dm, a, false, true, G, Wm);
        acc[a] = prm.lambda0 * Wm * ua[a];
      }
    }
  }
}

// Second half of the sorted shared scatter (see helmholtz_kernel): the lanes of
// one element walk its SHARED slots in ascending node order; `vals` is the
// padded element tensor in LDS, `codes` the element's `enc` row in slot order.
template <typename T, int P, int DIM, bool PAD = true>
__device__ __forceinline__ void scatter_shared_sorted(
    const uint16_t* so, int stride, int t, const T* vals,
    const uint32_t* codes, T* dst, int64_t ns) {
  using Tile = HelmholtzTile<T, P, DIM, PAD>;
  constexpr int TPE = Tile::TPE, SA = Tile::SA, SB = Tile::SB;
  for (int q = t; q < stride; q += TPE) {
    const uint32_t slot = so[q];
    if (slot != 0xFFFFu) {
      const uint32_t id = codes[slot] & SFEM_IDX_MASK;
      const int a2 = slot / TPE, t2 = slot - a2 * TPE;
      const int i2 = DIM == 3 ? t2 / P : 0;
      const int j2 = DIM == 3 ? t2 - i2 * P : t2;
      unsafeAtomicAdd(dst + (int64_t)id * ns, vals[a2 * SA + i2 * SB + j2]);
    }
  }
}

// SORTED: the sorted shared scatter below, its own instantiation (compiled
// together with the slot-order scatter it cost registers in both).
// MASS: lambda0 != 0 (decided at launch).  The pure stiffness kernels carry no
// mass-term code: 15-20 VGPRs less, which is what lets the multilinear kernel
// take the sorted path without spilling.
template <typename T, int P, int DIM, bool GS, bool SCALAR, int GM,
          bool SORTED = false, bool MASS = true>
__global__ void __launch_bounds__((HelmholtzTile<T, P, DIM>::BLOCK),
                                  (HelmholtzTile<T, P, DIM>::MINW))
helmholtz_kernel(HelmholtzParams<T> prm, DMat<T, P> dm) {
  // unpadded LDS rows for the light (affine) kernels at P = 8, see the tile
  constexpr bool PAD =
      !(GS && SCALAR && DIM == 3 && P == 8 && GM == GEO_AFFINE);
  using Tile = HelmholtzTile<T, P, DIM, PAD>;
  constexpr int TPE = Tile::TPE, SA = Tile::SA, SB = Tile::SB;
  constexpr int EPB = Tile::EPB, W = Tile::ELEM_WORDS;
  constexpr int N = DIM == 3 ? P * P * P : P * P;        // nodes per element
  __shared__ T lds[2 * EPB * W];

  const int tid = threadIdx.x;
  const int el = tid / TPE;                 // element within the workgroup
  const int t = tid - el * TPE;             // lane within the element
  const int i = DIM == 3 ? t / P : 0;
  const int j = DIM == 3 ? t - i * P : t;
  const bool lane_ok = el < EPB;            // tail lanes of a padded block
  const int64_t work = (int64_t)blockIdx.x * EPB + (lane_ok ? el : 0);
  const bool active = lane_ok && work < prm.num_elements;
  // element id: wave-uniform when an element fills whole waves
  const int64_t e =
      prm.elem_list ? (active ? (int64_t)prm.elem_list[work] : 0) : work;

  T* s0 = lds + (lane_ok ? el : 0) * 2 * W;    // becomes the axis-1 result
  T* s1 = s0 + W;                              // becomes the axis-2 result
  const DMat<T, P>& dmat = dm;
  // fp32, P >= 9: matrix entries from the kernarg segment (line_apply_mem;
  // measured p = 11 fp32 1.02 -> 0.99 ms at 48^3, p = 9 fp64 0.83 -> 0.85 ms)
#if SFEM_DMAT_MEM
#define SFEM_LINE_APPLY(TR, X, Y)                                             \
  do {                                                                        \
    if constexpr (P >= 9 && sizeof(T) == 4)                                   \
      line_apply_mem<T, P, TR>(                                               \
          (const SFEM_CONSTANT_AS DMat<T, P>*)((                              \
              const SFEM_CONSTANT_AS char*)                                   \
                  __builtin_amdgcn_kernarg_segment_ptr() +                    \
              sizeof(HelmholtzParams<T>)),                                    \
          X, Y);                                                              \
    else                                                                      \
      line_apply<T, P, TR>(dmat, X, Y);                                       \
  } while (0)
#else
#define SFEM_LINE_APPLY(TR, X, Y) line_apply<T, P, TR>(dmat, X, Y)
#endif
  // SCALAR: one component known at compile time (no component loop, so the
  // compiler has nothing to hoist out of it and spill)
  const int nc = SCALAR ? 1 : prm.ncomp;
  const int64_t ns = prm.node_stride, ks = prm.comp_stride;
  constexpr bool has_mass = MASS;
  const bool has_stiff = prm.lambda1 != T(0);

  // Owner layout: this lane holds nodes (a, i, j), a = 0..P-1, i.e. element
  // slots a*TPE + t.  Per-element arrays are addressed as a wave-uniform base
  // + one 32-bit per-lane offset + compile-time constant.
  ElemGeom<T, P, DIM, GM> geom;
#if SFEM_KERNARG_PICK
  // the second kernel argument as memory: per-lane weights / nodes are loads
  static_assert(sizeof(HelmholtzParams<T>) % alignof(DMat<T, P>) == 0, "");
  geom.template init<true>(prm, dm, e, active, i, j, t,
                           kernarg_dmat<T, P>(sizeof(HelmholtzParams<T>)));
#else
  geom.init(prm, dm, e, active, i, j, t);
#endif
  uint32_t slot_off_v = (uint32_t)t;
  const uint32_t& slot_off = slot_off_v;

  uint32_t enc[P];
  if (GS) {
    const int32_t* enc0 = prm.enc + e * N;
#pragma unroll
    for (int a = 0; a < P; ++a)
      // the index rows are streamed once per apply: keep them out of the L2
      enc[a] = active ? (uint32_t)__builtin_nontemporal_load(
                            &enc0[slot_off + a * TPE])
                      : (uint32_t)SFEM_IDX_PAD;
  }
  const T* ul0 = GS ? nullptr : prm.u + e * N * ns + prm.comp * ks;
  T* ol0 = GS ? nullptr : prm.out + e * N * ns + prm.comp * ks;
  const T* ug = prm.u + prm.comp * ks;
  T* og = prm.out + prm.comp * ks;

  double udot = 0.0;   // this lane's share of u . out (fused p.Ap of CG)
  for (int k = 0; k < nc; ++k) {
    if (!SCALAR) {
      // keep address arithmetic, flag tests and geometry products inside the
      // component loop: hoisted out of it they occupy ~100 registers and spill
      asm volatile("" : "+v"(geom.lane_off), "+v"(slot_off_v));
      if (GM != GEO_POINT) asm volatile("" : "+v"(geom.wbc));
      if (GM == GEO_MULTILINEAR) {
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
          asm volatile("" : "+v"(geom.r0[c]), "+v"(geom.p1[c]),
                       "+v"(geom.q1[c]));
          if (DIM == 3) asm volatile("" : "+v"(geom.p2[c]), "+v"(geom.q2[c]));
        }
      }
    }
    T ua[P], acc[P];
#pragma unroll
    for (int a = 0; a < P; ++a) {
      if (GS) {
        const uint32_t id = enc[a] & SFEM_IDX_MASK;
        ua[a] = id == SFEM_IDX_PAD ? T(0) : ug[(int64_t)id * ns + k * ks];
      } else {
        ua[a] = active ? ul0[(int64_t)(slot_off + a * TPE) * ns + k * ks] : T(0);
      }
    }
    if (has_stiff) {
      T d0[P];   // derivative along axis 0 at (a, i, j)
      SFEM_LINE_APPLY(false, ua, d0);
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
        SFEM_LINE_APPLY(false, x, y);
#pragma unroll
        for (int m = 0; m < P; ++m) line[m] = y[m];
      }
      if (DIM == 3 && lane_ok) {  // middle axis: lane owns the line [i, *, j]
        T* line = s0 + i * SA + j;
        T x[P], y[P];
#pragma unroll
        for (int m = 0; m < P; ++m) x[m] = line[m * SB];
        SFEM_LINE_APPLY(false, x, y);
#pragma unroll
        for (int m = 0; m < P; ++m) line[m * SB] = y[m];
      }
      __syncthreads();
      // pointwise: w = G * (reference gradient), G symmetric; the mass term
      // rides along when both are requested
      T w0[P];
#pragma unroll
      for (int a = 0; a < P; ++a) { w0[a] = T(0); acc[a] = T(0); }
      if (active) {
#pragma unroll
        for (int a = 0; a < P; ++a) {
          const int o = a * SA + i * SB + j;
          T G[6], Wm;
          constexpr bool FUSE_W = GM != GEO_POINT;
          if constexpr (GM == GEO_MULTILINEAR && DIM == 3) {
            T o0, o1, o2;
            geom.apply_multilinear3(dm, a, has_mass, d0[a], s0[o], s1[o], o0,
                                    o1, o2, Wm);
            if (has_mass) acc[a] = prm.lambda0 * Wm * ua[a];
            w0[a] = o0; s0[o] = o1; s1[o] = o2;
            continue;
          }
          geom.factors(dm, a, true, FUSE_W && has_mass, G, Wm);
          if (FUSE_W && has_mass) acc[a] = prm.lambda0 * Wm * ua[a];
          if (DIM == 3) {
            const T g0 = d0[a], g1 = s0[o], g2 = s1[o];
            w0[a] = G[0] * g0 + G[1] * g1 + G[2] * g2;
            s0[o] = G[1] * g0 + G[3] * g1 + G[4] * g2;
            s1[o] = G[2] * g0 + G[4] * g1 + G[5] * g2;
          } else {
            const T g0 = d0[a], g1 = s0[o];
            w0[a] = G[0] * g0 + G[1] * g1;
            s0[o] = G[1] * g0 + G[3] * g1;
          }
        }
      }
      __syncthreads();
      if (lane_ok) {  // transposed derivative along the last axis, in place
        T* line = (DIM == 3 ? s1 + i * SA + j * SB : s0 + j * SA);
        T x[P], y[P];
#pragma unroll
        for (int m = 0; m < P; ++m) x[m] = line[m];
        SFEM_LINE_APPLY(true, x, y);
#pragma unroll
        for (int m = 0; m < P; ++m) line[m] = y[m];
      }
      if (DIM == 3 && lane_ok) {
        T* line = s0 + i * SA + j;
        T x[P], y[P];
#pragma unroll
        for (int m = 0; m < P; ++m) x[m] = line[m * SB];
        SFEM_LINE_APPLY(true, x, y);
#pragma unroll
        for (int m = 0; m < P; ++m) line[m * SB] = y[m];
      }
      T dt0[P];
      SFEM_LINE_APPLY(true, w0, dt0);
      __syncthreads();
      if (lane_ok) {
#pragma unroll
        for (int a = 0; a < P; ++a) {
          const int o = a * SA + i * SB + j;
          T v = dt0[a] + s0[o];
          if (DIM == 3) v += s1[o];
          acc[a] += prm.lambda1 * v;
        }
      }
      if (GM == GEO_POINT && has_mass && active) {
#pragma unroll
        for (int a = 0; a < P; ++a) {
          T G[6], Wm;
          geom.factors(dm, a, false, true, G, Wm);
          acc[a] += prm.lambda0 * Wm * ua[a];
        }
      }
    } else {
#pragma unroll
      for (int a = 0; a < P; ++a) acc[a] = T(0);
      if (has_mass && active) {
#pragma unroll
        for (int a = 0; a < P; ++a) {
          T G[6], Wm;
          geom.factors(dm, a, false, true, G, Wm);
          acc[a] = prm.lambda0 * Wm * ua[a];
        }
      }
    }
    // direct-stiffness summation
    // SORTED is instantiated for one-wave elements in 3D only (see
    // launch_helmholtz): with several waves per element (P >= 9) the extra
    // barriers cost more than the coalescing gains (p = 11 fp32: 3.34 vs
    // 3.16 ms), and compiled into the same kernel as the slot-order scatter
    // the code slowed both down (p = 11: 2.31 -> 3.16 ms)
    constexpr bool sorted = GS && SORTED;
    if constexpr (sorted) {
      // The atomics are bound by the number of memory-side requests, i.e. of
      // 64-byte lines an instruction touches.  In slot order an instruction
      // (fixed a, lanes (i, j)) meets 4 faces + 4 edges: ~3 lanes per line.
      // In ascending NODE order consecutive lanes walk along a face or an edge
      // (the refiner numbers facet interiors contiguously): 6-8 lanes per line.
      // Values and codes change lanes through LDS (both copies are free here).
#pragma unroll
      for (int a = 0; a < P; ++a) {
        uint32_t ea = enc[a];
        asm volatile("" : "+v"(ea));
        const uint32_t id = ea & SFEM_IDX_MASK;
        if (id != SFEM_IDX_PAD) {
          const bool dirichlet = ea & SFEM_IDX_DIRICHLET;
          if (!dirichlet) udot += (double)acc[a] * (double)ua[a];
          if (!(ea & SFEM_IDX_SHARED))
            og[(int64_t)id * ns + k * ks] = dirichlet ? T(0) : acc[a];
        }
      }
      uint32_t* codes = reinterpret_cast<uint32_t*>(s1);
      if (Tile::BLOCK > 64) __syncthreads();   // other waves still read s1
      if (lane_ok) {
#pragma unroll
        for (int a = 0; a < P; ++a) {
          s0[a * SA + i * SB + j] = acc[a];
          codes[a * TPE + t] = enc[a];
        }
      }
      __syncthreads();
      if (active)
        scatter_shared_sorted<T, P, DIM, PAD>(
            prm.shared_order + e * prm.shared_stride, prm.shared_stride, t, s0,
            codes, og + k * ks, ns);
    }
    if (!sorted) {
#pragma unroll
    for (int a = 0; a < P; ++a) {
      if (GS) {
        uint32_t ea = enc[a];
        // keep the flag tests inside the component loop (hoisting them costs
        // 3 lane masks per a in SGPRs and spills the D matrix)
        asm volatile("" : "+v"(ea));
        const uint32_t id = ea & SFEM_IDX_MASK;
        if (id != SFEM_IDX_PAD) {
          T* dst = og + (int64_t)id * ns + k * ks;
          const bool dirichlet = ea & SFEM_IDX_DIRICHLET;
          if (!dirichlet) udot += (double)acc[a] * (double)ua[a];
          if (ea & SFEM_IDX_SHARED) {
            if (!dirichlet) {
              if (prm.colored) *dst = *dst + acc[a];
              else unsafeAtomicAdd(dst, acc[a]);
            }
          } else {
            *dst = dirichlet ? T(0) : acc[a];
          }
        }
      } else if (active) {
        ol0[(int64_t)(slot_off + a * TPE) * ns + k * ks] = acc[a];
      }
    }
    }
    if (k + 1 < nc) __syncthreads();
  }
  if (GS && prm.dot_out) {
    // u . out = sum over element slots of u_slot * (local result)_slot: one
    // wave reduction and one atomic per wave into a strip of partial sums
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) udot += __shfl_down(udot, off, 64);
    if ((tid & 63) == 0)
      unsafeAtomicAdd(&prm.dot_out[(blockIdx.x * (Tile::BLOCK / 64) +
                                    (tid >> 6)) & (SFEM_DOT_SLOTS - 1)],
                      udot);
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
  const DMat<T, P> dm =
      make_dmat<T, P>(prm.dmat_host, prm.weights_host, prm.nodes_host);
  const dim3 grid((unsigned)groups), block(Tile::BLOCK);
  // One-wave elements in 3D can issue their atomics in node order.  Not the
  // multilinear kernels WITH a mass term: they sit at the 128-VGPR budget of
  // 4 waves per SIMD and the sorted path spills (1.09 vs 1.04 ms; 1.13 ms
  // with 3 waves).
  constexpr bool CAN_SORT = GS && DIM == 3 && Tile::TPE <= 64;
  const bool sorted = CAN_SORT && prm.shared_order && !prm.colored;
  const bool mass = prm.lambda0 != T(0);
#define SFEM_LAUNCH_SM(SC, GMV, PRM, MASSV)                                   \
  do {                                                                        \
    constexpr bool SORT_GM =                                                  \
        CAN_SORT && (GMV != GEO_MULTILINEAR || !(MASSV));                     \
    if (sorted && SORT_GM)                                                    \
      hipLaunchKernelGGL(                                                     \
          (helmholtz_kernel<T, P, DIM, GS, SC, GMV, SORT_GM, MASSV>), grid,   \
          block, 0, stream, PRM, dm);                                         \
    else                                                                      \
      hipLaunchKernelGGL(                                                     \
          (helmholtz_kernel<T, P, DIM, GS, SC, GMV, false, MASSV>), grid,     \
          block, 0, stream, PRM, dm);                                         \
  } while (0)
#define SFEM_LAUNCH_GM(SC, GMV, PRM)                                          \
  do {                                                                        \
    if (mass) SFEM_LAUNCH_SM(SC, GMV, PRM, true);                             \
    else SFEM_LAUNCH_SM(SC, GMV, PRM, false);                                 \
  } while (0)
  if (prm.ncomp == 1) {
    switch (prm.geo_mode) {
      case GEO_POINT: SFEM_LAUNCH_GM(true, GEO_POINT, prm); break;
      case GEO_AFFINE: SFEM_LAUNCH_GM(true, GEO_AFFINE, prm); break;
      default: SFEM_LAUNCH_GM(true, GEO_MULTILINEAR, prm); break;
    }
  } else if (prm.geo_mode == GEO_POINT) {
    SFEM_LAUNCH_GM(false, GEO_POINT, prm);     // factors read once per element
  } else if (prm.geo_mode == GEO_AFFINE) {
    SFEM_LAUNCH_GM(false, GEO_AFFINE, prm);
  } else {
    SFEM_LAUNCH_GM(false, GEO_MULTILINEAR, prm);
  }
#undef SFEM_LAUNCH_GM
#undef SFEM_LAUNCH_SM
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
