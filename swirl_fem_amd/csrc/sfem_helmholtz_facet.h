// Fused collocated Helmholtz / mass / stiffness operator on COMPACT (facet)
// connectivity, 3D, one wavefront per element (P = 6..8).
//
//   out = mask * scatter( lambda0 * B_loc(g) + lambda1 * A_loc(g) ),  g = gather(u)
//
// Same operator and reference call sites as sfem_helmholtz.h
// (examples/poisson.py:141-154, navier_stokes/navier_stokes.py:220-236,
// :295-307, :431; element action core/interpolation.py:288-292 and its
// transpose core/fespace.py:471).  What differs is how an element finds its
// nodes and how the wave moves them:
//
//   * Connectivity.  `refine_premesh` (reference core/mesh_refiner.py:143-251)
//     numbers the interior nodes of every facet of the premesh (vertex, edge,
//     face, element interior) as one contiguous block, read by an element
//     through one of the 2^k k! orientations of the facet.  An element's P^3
//     ids are therefore 27 affine maps
//         id(a, i, j) = id0_f + sa_f (a - 1) + si_f (i - 1) + sj_f (j - 1)
//     one per facet f = (class(a), class(i), class(j)), class = FIRST / INNER /
//     LAST.  The kernel reads that 27 x 16-byte table (432 B per element)
//     instead of the P^3 x 4-byte index row (2 KB at P = 8) and the 0.6 KB
//     sorted-slot list: -26 % of the bytes an apply must move at config 2, 3
//     loads per lane instead of 8 + 5, Dirichlet / shared flags per facet
//     instead of per node.  Tables are built and verified per element by
//     `sfem_facet_table_build`; elements whose ids are not of this form (hand
//     numbered meshes, padding, node masks that cut through a facet) keep the
//     index-row kernel.
//   * LDS layout.  Counters of the index-row kernel (profiles/r02, pmc_issue)
//     show the LDS pipe 75 % busy, 47 % of that in bank conflicts (unpadded
//     rows: the mid-axis lines of the four i's of a 32-lane group sit on the
//     same banks).  Here node (x, y, z) of the element tensor lives at word
//     A x + B y + z with (A, B) = (67, 8) at P = 8 and the lines of the two
//     transposed passes are dealt to lanes the other way round (lane (i, j)
//     takes line x = j); every address is still `lane base + compile-time
//     offset`, i.e. three base registers and no swizzle arithmetic.  Modelled
//     bank cycles per element: 1216 (unpadded) / 576 (rows padded to 9) ->
//     480, the conflict-free bound being 384.
//   * Box elements (GEO_BOX: constant diagonal J^-1 J^-T, every Cartesian
//     mesh).  A_loc = sum_k c_k (K along axis k) (x) (w along the others) with
//     the 1D stiffness matrix K = D^T diag(w) D: three line products instead
//     of six and half the LDS traffic (`helmholtz_box_kernel`).
//   * No per-slot predicates: one element per workgroup, the grid is exactly
//     the element list, tables hold no padding slots.
#pragma once
#include <stdlib.h>

#include "sfem_helmholtz.h"

namespace sfem {

constexpr int GEO_BOX = 5;
constexpr int FACET_ROW = 27 * 4;   // int32 words per element table

template <typename T>
struct FacetParams {
  const T* u;
  T* out;
  const int32_t* tab;        // (E, 27, 4): code = id0 | flags, sa, si, sj
  const T* geo_const;        // (E, 8): G00 G01 G02 G11 G12 G22 (= detJ J^-1
                             //   J^-T, no quadrature weight), detJ, unused
  const T* geo_elem;         // (E, 24) multilinear coefficients
  const T* geo;              // per-point factors (GEO_POINT)
  const int32_t* geo_index;  // (E,) slot in `geo` or null
  const int32_t* elem_list;  // ids of the elements of this launch, or null
  const int32_t* chain_off;   // chain launches: (S + 1,) segment bounds in
  const int32_t* chain_elems; //   chain_elems (element ids in walking order)
  int64_t comp_stride;       // component k of node n at u[n + k comp_stride]
  int ncomp;
  T lambda0, lambda1;
  double* dot_out;
  // Layered assembly (no atomics, no cleared range, bitwise reproducible):
  // `tab` is then the (E, 27, 4) table in its layered form
  //   x = id0 | flags (as above: where the element READS),
  //   y = position in the EXTENDED output (layer base + id0) | DIRICHLET flag,
  //   z = sa, w = (si & 0xffff) | (sj << 16)
  // and `out` an extended vector [ N nodal values | layer 1 | layer 2 | ... ]:
  // every (element, facet) that writes has a layer of its own for that facet,
  // so all results leave as plain stores; the consumer adds the layers up
  // (`sfem_cg_update_r_layered`, `sfem_fold_layers`).  Scalar fields.
  int layered;
  // layered launches: > 0 = `dot_out` holds this many doubles, one per wave of
  // the launch; every wave STORES its partial sum of u . out at its own index
  // (nothing accumulated, nothing to clear: the sum over the slots in index
  // order is bitwise reproducible); 0 = SFEM_DOT_SLOTS atomically
  // accumulated slots
  int64_t dot_slots;
};

constexpr uint32_t FACET_SKIP = 0x40000000u;   // layered: class is not written

// Where node (x, y, z) of an element sits in its LDS copy (in words of T), and
// the shape of the workgroup: one element, ceil(P^2 / 64) waves.
template <int P>
struct FacetLayout {
  // P = 8: see the header comment; P >= 9 (several waves per element): rows
  // padded to an odd length as in the index-row kernel
  static constexpr int B = P >= 9 ? (P | 1) : P;
  static constexpr int A = P == 8 ? 67 : (P == 7 ? 52 : P * B);
  static constexpr bool SWAP = P == 8;   // transposed passes: lane (i, j) takes
                                         // the line with first index j
  static constexpr int WORDS = (P - 1) * (A + B + 1) + 1;
  static constexpr int COPY = (WORDS + 3) & ~3;
  static constexpr int TPE = P * P;
  static constexpr int WAVES = (TPE + 63) / 64;
  static constexpr int BLOCK = 64 * WAVES;
  __host__ __device__ static constexpr int word(int x, int y, int z) {
    return A * x + B * y + z;
  }
};

// Orders the LDS traffic of the element's lanes: a compiler fence inside one
// wave, a workgroup barrier (LDS only, see lds_barrier) across several.
template <int P>
__device__ __forceinline__ void facet_sync() {
  if (FacetLayout<P>::WAVES == 1) wave_sync();
  else lds_barrier();
}

// Shared scatter: in round q the lanes of the element take one face interior
// ((P-2)^2 lanes), two edge interiors (2 (P-2) lanes) and -- round 0 -- the
// eight vertices, so that one atomic instruction covers whole contiguous node
// blocks.  s[lane][q] = LDS word of the node the lane handles, 0xFFFF none.
template <int P>
struct FacetSlots {
  uint16_t s[FacetLayout<P>::BLOCK][6];
};

template <int P>
constexpr FacetSlots<P> make_facet_slots() {
  using L = FacetLayout<P>;
  FacetSlots<P> t{};
  for (int l = 0; l < L::BLOCK; ++l)
    for (int q = 0; q < 6; ++q) t.s[l][q] = 0xFFFF;
  constexpr int M = P - 2, NF = M * M;
  static_assert(NF + 2 * M + 8 <= L::TPE, "rounds must fit the element's lanes");
  // edges: 4 along j, 4 along i, 4 along a
  int edges[12][3] = {};   // fixed coordinates, -1 = running
  int ne = 0;
  for (int d = 2; d >= 0; --d)
    for (int pp = 0; pp < 2; ++pp)
      for (int r = 0; r < 2; ++r) {
        int c[3] = {0, 0, 0};
        c[d] = -1;
        c[(d + 1) % 3] = pp ? P - 1 : 0;
        c[(d + 2) % 3] = r ? P - 1 : 0;
        for (int k = 0; k < 3; ++k) edges[ne][k] = c[k];
        ++ne;
      }
  for (int q = 0; q < 6; ++q) {
    const int d = q / 2, fixed = (q & 1) ? P - 1 : 0;   // face normal axis d
    for (int l = 0; l < NF; ++l) {
      const int x = l / M + 1, y = l % M + 1;
      int c[3] = {0, 0, 0};
      c[d] = fixed;
      c[(d + 1) % 3] = d == 1 ? y : x;   // keep (x, y) in ascending axis order
      c[(d + 2) % 3] = d == 1 ? x : y;
      t.s[l][q] = (uint16_t)L::word(c[0], c[1], c[2]);
    }
    for (int h = 0; h < 2; ++h)
      for (int z = 0; z < M; ++z) {
        int c[3] = {0, 0, 0};
        for (int k = 0; k < 3; ++k)
          c[k] = edges[2 * q + h][k] < 0 ? z + 1 : edges[2 * q + h][k];
        t.s[NF + h * M + z][q] = (uint16_t)L::word(c[0], c[1], c[2]);
      }
  }
  for (int v = 0; v < 8; ++v)
    t.s[NF + 2 * M + v][0] = (uint16_t)L::word(
        (v & 4) ? P - 1 : 0, (v & 2) ? P - 1 : 0, (v & 1) ? P - 1 : 0);
  return t;
}

template <int P>
__device__ const FacetSlots<P> g_facet_slots = make_facet_slots<P>();

// Centrosymmetric P x P matrix (K[P-1-r][P-1-m] = K[r][m]) in even/odd form:
//   xe = x[m] + x[P-1-m], xo = x[m] - x[P-1-m]  (m < PH),  xe[PH] = x[PH]
//   se = KE xe (PC x PC), so = KO xo (PH x PH)
//   y[r] = se[r] + so[r], y[P-1-r] = se[r] - so[r], y[PH] = se[PH]
template <typename T, int P>
struct SMat {
  static constexpr int PH = P / 2, PC = P - P / 2;
  T e[PC * PC];
  T o[PH * PH > 0 ? PH * PH : 1];
  T w[P];
};

// K = D^T diag(w) D of the 1D nodes (the 1D stiffness matrix).
template <typename T, int P>
inline SMat<T, P> make_smat(const T* d, const T* w) {
  constexpr int PH = P / 2, PC = P - P / 2;
  double K[P][P];
  for (int r = 0; r < P; ++r)
    for (int m = 0; m < P; ++m) {
      double acc = 0;
      for (int q = 0; q < P; ++q)
        acc += (double)d[q * P + r] * (double)w[q] * (double)d[q * P + m];
      K[r][m] = acc;
    }
  SMat<T, P> sm;
  for (int r = 0; r < P; ++r) sm.w[r] = w[r];
  for (int r = 0; r < PC; ++r)
    for (int m = 0; m < PC; ++m)
      sm.e[r * PC + m] = (T)(m < PH ? (K[r][m] + K[r][P - 1 - m]) / 2
                                    : K[r][m]);
  for (int r = 0; r < PH; ++r)
    for (int m = 0; m < PH; ++m)
      sm.o[r * PH + m] = (T)((K[r][m] - K[r][P - 1 - m]) / 2);
  return sm;
}

template <typename T, int P>
__device__ __forceinline__ void sym_line_apply(const SMat<T, P>& sm,
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
#pragma unroll
    for (int m = 0; m < PC; ++m) se += sm.e[r * PC + m] * xe[m];
    if (r < PH) {
#pragma unroll
      for (int m = 0; m < PH; ++m) so += sm.o[r * PH + m] * xo[m];
      y[r] = se + so;
      y[P - 1 - r] = se - so;
    } else {
      y[r] = se;
    }
  }
}

// The same product with the matrix re-read from the kernarg segment per product
// (fp32, P >= 9: see line_apply_mem).
template <typename T, int P>
__device__ __forceinline__ void sym_line_apply_mem(
    const SFEM_CONSTANT_AS SMat<T, P>* km, const T (&x)[P], T (&y)[P]) {
  constexpr int PH = P / 2, PC = P - P / 2;
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
#pragma unroll
    for (int m = 0; m < PC; ++m) se += km->e[r * PC + m] * xe[m];
    if (r < PH) {
#pragma unroll
      for (int m = 0; m < PH; ++m) so += km->o[r * PH + m] * xo[m];
      y[r] = se + so;
      y[P - 1 - r] = se - so;
    } else {
      y[r] = se;
    }
  }
}

// Address of node `code` (flags in the two top bits) in a field.  OFF32: the
// field spans < 4 GiB, byte offsets fit 32 bits and the shift drops the flags
// (saddr + 32-bit voffset addressing, no 64-bit vector arithmetic).
template <typename T, bool OFF32>
__device__ __forceinline__ T* facet_node(T* base, uint32_t code) {
  if (OFF32) {
    const uint32_t off = code << (sizeof(T) == 8 ? 3 : 2);
    return (T*)((const char*)base + off);
  }
  return base + (code & SFEM_IDX_MASK);
}

// Per-lane connectivity of one element: the lane's nodes in the three slice
// classes (FIRST a = 0, INNER 0 < a < P-1, LAST a = P-1) and the a-stride of
// the INNER facet.  Flags ride in the two top bits of t[].
template <int P, bool LAY = false>
struct FacetLane {
  typedef int32_t I4 __attribute__((ext_vector_type(4)));
  struct Raw { I4 en[3]; };   // the lane's three table entries as loaded
  uint32_t t[3];
  uint32_t to[LAY ? 3 : 1];   // layered: where the lane WRITES (| DIRICHLET,
                              // | FACET_SKIP), same strides as t[]
  int32_t sa;
  static __device__ __forceinline__ int cls(int a) {
    return a == 0 ? 0 : (a == P - 1 ? 2 : 1);
  }
  // requests the entries (3 x 16 bytes per lane); no wait
  static __device__ __forceinline__ void issue(Raw& raw, const int32_t* tab,
                                               int64_t e, int i, int j) {
    const int lc = cls(i) * 3 + cls(j);
    const I4* row = reinterpret_cast<const I4*>(tab + e * FACET_ROW);
#pragma unroll
    for (int c = 0; c < 3; ++c) raw.en[c] = row[c * 9 + lc];
  }
  __device__ __forceinline__ void finish(const Raw& raw, int i, int j) {
    if constexpr (LAY) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int32_t si = (raw.en[c].w << 16) >> 16, sj = raw.en[c].w >> 16;
        const uint32_t d = (uint32_t)__mul24(si, i - 1) +
                           (uint32_t)__mul24(sj, j - 1);
        t[c] = (uint32_t)raw.en[c].x + d;
        to[c] = (uint32_t)raw.en[c].y + d;
      }
      sa = raw.en[1].z;
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c)
        t[c] = (uint32_t)raw.en[c].x + (uint32_t)__mul24(raw.en[c].z, i - 1) +
               (uint32_t)__mul24(raw.en[c].w, j - 1);
      sa = raw.en[1].y;
    }
  }
  __device__ __forceinline__ void load(const int32_t* tab, int64_t e, int i,
                                       int j) {
    Raw raw;
    issue(raw, tab, e, i, j);
    finish(raw, i, j);
  }
  // node id | flags of the lane's node in slice a (compile-time a)
  __device__ __forceinline__ uint32_t code(int a) const {
    return a == 0 ? t[0]
                  : (a == P - 1 ? t[2] : t[1] + (uint32_t)(sa * (a - 1)));
  }
  __device__ __forceinline__ uint32_t flags(int a) const {
    return LAY ? to[cls(a)] : t[cls(a)];
  }
  // layered: position | flags of the lane's result for slice a
  __device__ __forceinline__ uint32_t ocode(int a) const {
    return a == 0 ? to[0]
                  : (a == P - 1 ? to[LAY ? 2 : 0]
                                : to[LAY ? 1 : 0] + (uint32_t)(sa * (a - 1)));
  }
};

// Direct-stiffness summation of one element's results `acc` (slot layout:
// lane (i, j) holds nodes (a, i, j)), in two halves so that a kernel can issue
// other memory traffic between them.  Head: Dirichlet rows and u . out.
template <typename T, int P, bool LAY = false>
__device__ __forceinline__ void facet_scatter_head(const FacetLane<P, LAY>& fl,
                                                   T (&acc)[P],
                                                   const T (&ua)[P],
                                                   bool want_dot,
                                                   double& udot) {
#pragma unroll
  for (int a = 0; a < P; ++a)
    if (fl.flags(a) & SFEM_IDX_DIRICHLET) acc[a] = T(0);
  if (want_dot) {
#pragma unroll
    for (int a = 0; a < P; ++a) udot += (double)acc[a] * (double)ua[a];
  }
}

// Tail: plain stores for the nodes of this element only, atomics for the
// shared ones.  `vals` / `codes`: free LDS of FacetLayout<P>::WORDS words of T
// resp. uint32.
template <typename T, int P, bool OFF32>
__device__ __forceinline__ void facet_scatter_tail(
    const FacetLane<P>& fl, const uint16_t (&slots)[6], const T (&acc)[P],
    T* og, T* vals, uint32_t* codes, uint32_t own_w, bool lane_ok) {
  using L = FacetLayout<P>;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (lane_ok && !(fl.t[c] & SFEM_IDX_SHARED)) {
#pragma unroll
      for (int a = 0; a < P; ++a)
        if (FacetLane<P>::cls(a) == c) {
          *facet_node<T, OFF32>(og, fl.code(a)) = acc[a];
        }
    }
  }
  // shared nodes: values and codes change lanes through LDS
  if (lane_ok) {
#pragma unroll
    for (int a = 0; a < P; ++a) {
      vals[own_w + a * L::A] = acc[a];
      codes[own_w + a * L::A] = fl.code(a);
    }
  }
  facet_sync<P>();
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const uint32_t w = slots[q];
    if (w != 0xFFFFu) {
      const uint32_t code = codes[w];
      if ((code & ~(uint32_t)SFEM_IDX_MASK) == SFEM_IDX_SHARED) {
        unsafeAtomicAdd(facet_node<T, OFF32>(og, code), vals[w]);
      }
    }
  }
}

// Layered assembly: every class of the lane that is written goes out as plain
// stores from the lane's own slots -- no LDS round trip, no atomics.
template <typename T, int P, bool OFF32>
__device__ __forceinline__ void facet_store_layered(
    const FacetLane<P, true>& fl, const T (&acc)[P], T* og, bool lane_ok) {
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (lane_ok && !(fl.to[c] & FACET_SKIP)) {
#pragma unroll
      for (int a = 0; a < P; ++a)
        if (FacetLane<P, true>::cls(a) == c)
          *facet_node<T, OFF32>(og, fl.ocode(a)) = acc[a];
    }
  }
}

// The same with the element's surface in SLOT order: face, edge and vertex
// results change lanes through LDS (as for the atomics above), so that one
// store instruction covers whole contiguous facet blocks; the element interior
// leaves from the lanes' own slots.  Which order is faster depends on what else
// the kernel keeps busy (config 2, timing builds with plain stores in place of
// the atomics: box 0.494 ms in slot order / 0.574 own order, stored factors
// 1.753 / 1.793, multilinear 0.820 / 0.766, p = 11 fp32 1.433 / 1.249):
// ELEM::LAYERED_SLOTS.
template <typename T, int P, bool OFF32>
__device__ __forceinline__ void facet_store_layered_slots(
    const FacetLane<P, true>& fl, const uint16_t (&slots)[6], const T (&acc)[P],
    T* og, T* vals, uint32_t* codes, uint32_t own_w, bool lane_ok,
    bool inner_lane) {
  using L = FacetLayout<P>;
  if (lane_ok && inner_lane) {
#pragma unroll
    for (int a = 1; a < P - 1; ++a)
      *facet_node<T, OFF32>(og, fl.ocode(a)) = acc[a];
  }
  if (lane_ok) {
#pragma unroll
    for (int a = 0; a < P; ++a) {
      vals[own_w + a * L::A] = acc[a];
      codes[own_w + a * L::A] = fl.ocode(a);
    }
  }
  facet_sync<P>();
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const uint32_t w = slots[q];
    if (w != 0xFFFFu) {
      const uint32_t code = codes[w];
      if (!(code & FACET_SKIP)) *facet_node<T, OFF32>(og, code) = vals[w];
    }
  }
}

// Kernel-argument offset of the matrices (read through the kernarg segment
// with lane-dependent indices, and as scalar memory for P >= 9 in fp32).
template <typename PRM, typename MAT>
struct FacetKernarg {
  static constexpr size_t MAT_OFF =
      (sizeof(PRM) + alignof(MAT) - 1) / alignof(MAT) * alignof(MAT);
};

__device__ __forceinline__ const char* kernarg_bytes() {
#if defined(__HIP_DEVICE_COMPILE__)
  return (const char*)(const __attribute__((address_space(4))) char*)
      __builtin_amdgcn_kernarg_segment_ptr();
#else
  return nullptr;
#endif
}

#ifndef SFEM_FACET_AFFINE_MINW
#define SFEM_FACET_AFFINE_MINW 5
#endif
#ifndef SFEM_FACET_BOX_MINW
#define SFEM_FACET_BOX_MINW 5
#endif
#ifndef SFEM_FACET_BOX_MINW_HI
#define SFEM_FACET_BOX_MINW_HI 3
#endif
#ifndef SFEM_FACET_CHAIN_MINW_HI
#define SFEM_FACET_CHAIN_MINW_HI 2
#endif
// multilinear / stored factors: the look-ahead of the chain loop needs a 168-
// register budget (at 4 waves per SIMD: 116-144 B of spills, 1.28 ms; at 3:
// multilinear 0.865 vs 0.919 ms one element per wave, stored 1.795 vs 1.855)
#ifndef SFEM_FACET_CHAIN_MINW_GEN
#define SFEM_FACET_CHAIN_MINW_GEN 3
#endif
#ifndef SFEM_FACET_CHAIN_MINW_BOX
#define SFEM_FACET_CHAIN_MINW_BOX 4
#endif
#ifndef SFEM_FACET_CHAIN_MINW_AFFINE
#define SFEM_FACET_CHAIN_MINW_AFFINE 4
#endif

// 1D quadrature weight of slice a; the node sets are symmetric (supports_fused),
// so only the first half of the by-value copy is ever read (fewer SGPRs).
template <typename T, int P, typename M>
__device__ __forceinline__ T sym_w(const M& m, int a) {
  return m.w[a < P - 1 - a ? a : P - 1 - a];
}

// y = D x / D^T x with the matrix taken from the by-value kernel argument, or
// (fp32, P >= 9: 72 entries do not fit the SGPRs next to everything else, see
// line_apply_mem) re-read from the kernarg segment per product.
template <typename T, int P, bool TRANS, typename PRM>
__device__ __forceinline__ void facet_line(const DMat<T, P>& dm,
                                           const T (&x)[P], T (&y)[P]) {
  if constexpr (P >= 9 && sizeof(T) == 4) {
    using KA = FacetKernarg<PRM, DMat<T, P>>;
    line_apply_mem<T, P, TRANS>(
        (const SFEM_CONSTANT_AS DMat<T, P>*)((const SFEM_CONSTANT_AS char*)
                                                 __builtin_amdgcn_kernarg_segment_ptr() +
                                             KA::MAT_OFF),
        x, y);
  } else {
    line_apply<T, P, TRANS>(dm, x, y);
  }
}

template <typename T, int P, typename PRM>
__device__ __forceinline__ void facet_sym_line(const SMat<T, P>& sm,
                                               const T (&x)[P], T (&y)[P]) {
  if constexpr (P >= 9 && sizeof(T) == 4) {
    using KA = FacetKernarg<PRM, SMat<T, P>>;
    sym_line_apply_mem<T, P>(
        (const SFEM_CONSTANT_AS SMat<T, P>*)((const SFEM_CONSTANT_AS char*)
                                                 __builtin_amdgcn_kernarg_segment_ptr() +
                                             KA::MAT_OFF),
        x, y);
  } else {
    sym_line_apply<T, P>(sm, x, y);
  }
}

// Lane constants of one element's workgroup.
template <int P>
struct FacetWave {
  using L = FacetLayout<P>;
  static constexpr int TPE = P * P;
  int lane, i, j;                   // lane = thread of the element's workgroup
  bool ok;                          // lane holds a line of the element
  uint32_t own_w, mid_w, last_w;    // LDS words of the three access patterns
  template <typename T, typename M>
  __device__ __forceinline__ T wij(const M* kmat) const {
    return kmat->w[i] * kmat->w[j];
  }
  __device__ __forceinline__ void init() {
    lane = threadIdx.x;
    ok = TPE == L::BLOCK || lane < TPE;
    const int t = ok ? lane : 0;
    i = t / P;
    j = t - i * P;
    own_w = L::B * i + j;
    mid_w = L::SWAP ? L::A * j + i : L::A * i + j;
    last_w = L::SWAP ? L::A * j + L::B * i : L::A * i + L::B * j;
  }
};

// (lambda0 B + lambda1 A)_local of one element, general geometry
// (GM = GEO_POINT / GEO_AFFINE / GEO_MULTILINEAR); two LDS copies.
template <typename T, int P, int GM, bool MASS>
struct FacetElem {
  using L = FacetLayout<P>;
  using Mat = DMat<T, P>;
  static constexpr int LDS_WORDS = 2 * L::COPY;
  // P >= 9: 12 nodes per lane and several waves per element: the 256-register
  // budget of the index-row kernels
  static constexpr int MINW =
      P >= 9 ? 2 : (GM == GEO_AFFINE ? SFEM_FACET_AFFINE_MINW : 4);
  static constexpr int CHAIN_MINW =
      P >= 9 ? 2 : (GM == GEO_AFFINE ? SFEM_FACET_CHAIN_MINW_AFFINE
                                    : SFEM_FACET_CHAIN_MINW_GEN);
  // chain launches are compiled where they pay (operators.py: box / affine)
  static constexpr bool CHAINS = GM == GEO_AFFINE || P <= 8;
  // layered assembly: surface results in slot order (facet_store_layered_slots)
  static constexpr bool LAYERED_SLOTS = P <= 8 && GM != GEO_MULTILINEAR;
  struct Raw { T c[GM == GEO_AFFINE ? 7 : 1]; int64_t e; };
  T cst[GM == GEO_AFFINE ? 6 : 1];
  T lw, Wm0;
  ElemGeom<T, P, 3, (GM == GEO_AFFINE ? GEO_POINT : GM)> geom;

  // requests the element's constants (wave-uniform addresses); no wait
  static __device__ __forceinline__ void fetch(Raw& raw,
                                               const FacetParams<T>& prm,
                                               int64_t e) {
    raw.e = e;
    if constexpr (GM == GEO_AFFINE) {
      const T* g = prm.geo_const + e * 8;
#pragma unroll
      for (int q = 0; q < 7; ++q) raw.c[q] = g[q];
    }
  }
  __device__ __forceinline__ void finish(const Raw& raw,
                                         const FacetParams<T>& prm,
                                         const Mat& dm, const Mat* kdm,
                                         const FacetWave<P>& w, T wij) {
    lw = T(0);
    Wm0 = T(0);
    if constexpr (GM == GEO_AFFINE) {
#pragma unroll
      for (int q = 0; q < 6; ++q) cst[q] = raw.c[q];
      lw = prm.lambda1 * wij;
      if (MASS) Wm0 = prm.lambda0 * raw.c[6] * wij;
    } else {
      HelmholtzParams<T> hp{};
      hp.geo = prm.geo;
      hp.geo_elem = prm.geo_elem;
      hp.geo_index = prm.geo_index;
      geom.template init<true>(hp, dm, raw.e, true, w.i, w.j,
                               w.ok ? w.lane : 0, kdm);
    }
  }

  __device__ __forceinline__ void apply(const FacetParams<T>& prm,
                                        const Mat& dm, const FacetWave<P>& w,
                                        T* lds, const T (&ua)[P],
                                        T (&acc)[P]) const {
    T* s0 = lds;
    T* s1 = lds + L::COPY;
    const bool lane_ok = w.ok;
    const uint32_t own_w = w.own_w, mid_w = w.mid_w, last_w = w.last_w;
    if (prm.lambda1 != T(0)) {
      if (lane_ok) {
#pragma unroll
        for (int a = 0; a < P; ++a) {
          s0[own_w + a * L::A] = ua[a];
          s1[own_w + a * L::A] = ua[a];
        }
      }
      facet_sync<P>();
      if (lane_ok) {   // last axis, copy 1
        T x[P], y[P];
#pragma unroll
        for (int m = 0; m < P; ++m) x[m] = s1[last_w + m];
        facet_line<T, P, false, FacetParams<T>>(dm, x, y);
#pragma unroll
        for (int m = 0; m < P; ++m) s1[last_w + m] = y[m];
      }
      if (lane_ok) {   // middle axis, copy 0
        T x[P], y[P];
#pragma unroll
        for (int m = 0; m < P; ++m) x[m] = s0[mid_w + m * L::B];
        facet_line<T, P, false, FacetParams<T>>(dm, x, y);
#pragma unroll
        for (int m = 0; m < P; ++m) s0[mid_w + m * L::B] = y[m];
      }
      // axis 0 in registers; w0 takes the place of d0 slice by slice
      T w0[P];
      facet_line<T, P, false, FacetParams<T>>(dm, ua, w0);
      facet_sync<P>();
      if (MASS || !lane_ok) {
#pragma unroll
        for (int a = 0; a < P; ++a) acc[a] = T(0);
      }
      if (lane_ok) {
        T chain = T(0);
        (void)chain;
#pragma unroll
        for (int a = 0; a < P; ++a) {
          T& r0 = s0[own_w + a * L::A];
          T& r1 = s1[own_w + a * L::A];
          if constexpr (GM == GEO_AFFINE) {
            const T g0 = w0[a], g1 = r0, g2 = r1;
            const T wa = sym_w<T, P>(dm, a);
            // (pinned: eight hoisted products would hold 16 registers)
            T lwa = lw;
            asm volatile("" : "+v"(lwa));
            const T sc = lwa * wa;
            w0[a] = sc * (cst[0] * g0 + cst[1] * g1 + cst[2] * g2);
            r0 = sc * (cst[1] * g0 + cst[3] * g1 + cst[4] * g2);
            r1 = sc * (cst[2] * g0 + cst[4] * g1 + cst[5] * g2);
            if (MASS) acc[a] = (Wm0 * ua[a]) * wa;
          } else if constexpr (GM == GEO_MULTILINEAR) {
            T o0, o1, o2, Wm;
            // the factors of a slice do not depend on loaded data: keep the
            // scheduler from evaluating all eight ahead of the passes above
            // (160 more live registers, measured 218 VGPRs instead of 126)
            T g0 = w0[a];
            ElemGeom<T, P, 3, GM> gs = geom;
            // ... and the slices one after the other (each waits for the
            // previous slice's result `chain`)
#pragma unroll
            for (int c = 0; c < 3; ++c)
              asm volatile("" : "+v"(gs.p1[c]), "+v"(gs.p2[c]), "+v"(g0)
                           : "v"(chain));
            gs.apply_multilinear3(dm, a, MASS, g0, r0, r1, o0, o1, o2, Wm);
            chain = o2;
            if (MASS) acc[a] = prm.lambda0 * Wm * ua[a];
            w0[a] = o0; r0 = o1; r1 = o2;
          } else {
            T G[6], Wm;
            geom.factors(dm, a, true, false, G, Wm);
            const T g0 = w0[a], g1 = r0, g2 = r1;
            w0[a] = G[0] * g0 + G[1] * g1 + G[2] * g2;
            r0 = G[1] * g0 + G[3] * g1 + G[4] * g2;
            r1 = G[2] * g0 + G[4] * g1 + G[5] * g2;
          }
        }
      }
      {   // transposed axis 0 at once: w0 dies here
        T dt0[P];
        facet_line<T, P, true, FacetParams<T>>(dm, w0, dt0);
#pragma unroll
        for (int a = 0; a < P; ++a) {
          // lambda1 is inside lw for affine elements
          const T v = GM == GEO_AFFINE ? dt0[a] : prm.lambda1 * dt0[a];
          if (MASS || !lane_ok) acc[a] += v;
          else acc[a] = v;
        }
      }
      facet_sync<P>();
      if (lane_ok) {
        T x[P], y[P];
#pragma unroll
        for (int m = 0; m < P; ++m) x[m] = s1[last_w + m];
        facet_line<T, P, true, FacetParams<T>>(dm, x, y);
#pragma unroll
        for (int m = 0; m < P; ++m) s1[last_w + m] = y[m];
      }
      if (lane_ok) {
        T x[P], y[P];
#pragma unroll
        for (int m = 0; m < P; ++m) x[m] = s0[mid_w + m * L::B];
        facet_line<T, P, true, FacetParams<T>>(dm, x, y);
#pragma unroll
        for (int m = 0; m < P; ++m) s0[mid_w + m * L::B] = y[m];
      }
      facet_sync<P>();
      if (lane_ok) {
#pragma unroll
        for (int a = 0; a < P; ++a) {
          const T v = s0[own_w + a * L::A] + s1[own_w + a * L::A];
          if (GM == GEO_AFFINE) acc[a] += v;
          else acc[a] += prm.lambda1 * v;
        }
      }
      if (GM == GEO_POINT && MASS && lane_ok) {
#pragma unroll
        for (int a = 0; a < P; ++a) {
          T G[6], Wm;
          geom.factors(dm, a, false, true, G, Wm);
          acc[a] += prm.lambda0 * Wm * ua[a];
        }
      }
      facet_sync<P>();
    } else {
#pragma unroll
      for (int a = 0; a < P; ++a) acc[a] = T(0);
      if (MASS && lane_ok) {
#pragma unroll
        for (int a = 0; a < P; ++a) {
          if constexpr (GM == GEO_AFFINE) {
            acc[a] = (Wm0 * sym_w<T, P>(dm, a)) * ua[a];
          } else {
            T G[6], Wm;
            geom.factors(dm, a, false, true, G, Wm);
            acc[a] = prm.lambda0 * Wm * ua[a];
          }
        }
      }
    }
  }
};

// Box elements: J^-1 J^-T diagonal and constant.  With K = D^T diag(w) D,
//   A_loc u (a,i,j) = c0 w_i w_j (K u)_a + w_a ( c1 w_j (K u)_i + c2 w_i (K u)_j )
//   B_loc u = detJ w_a w_i w_j u
// One LDS copy (both transposed passes read it before either writes back) plus
// room for the codes of the shared scatter.
template <typename T, int P, bool MASS>
struct BoxElem {
  using L = FacetLayout<P>;
  using Mat = SMat<T, P>;
  static constexpr int CODE_WORDS =
      (L::COPY * 4 + (int)sizeof(T) - 1) / (int)sizeof(T);
  static constexpr int LDS_WORDS = L::COPY + CODE_WORDS;
  static constexpr int MINW = P >= 9 ? SFEM_FACET_BOX_MINW_HI : SFEM_FACET_BOX_MINW;
  static constexpr int CHAIN_MINW =
      P >= 9 ? SFEM_FACET_CHAIN_MINW_HI : SFEM_FACET_CHAIN_MINW_BOX;
  static constexpr bool CHAINS = true;
  static constexpr bool LAYERED_SLOTS = P <= 8;
  struct Raw { T c[4]; };
  T P0, P1, P2, Wm;

  // requests c0, c1, c2, detJ (wave-uniform addresses); no wait
  static __device__ __forceinline__ void fetch(Raw& raw,
                                               const FacetParams<T>& prm,
                                               int64_t e) {
    const T* g = prm.geo_const + e * 8;
    raw.c[0] = g[0]; raw.c[1] = g[3]; raw.c[2] = g[5]; raw.c[3] = g[6];
  }
  __device__ __forceinline__ void finish(const Raw& raw,
                                         const FacetParams<T>& prm,
                                         const Mat& sm, const Mat* ksm,
                                         const FacetWave<P>& w, T wij) {
    const T wi = ksm->w[w.i], wj = ksm->w[w.j];
    P0 = prm.lambda1 * raw.c[0] * wij;
    P1 = prm.lambda1 * raw.c[1] * wj;
    P2 = prm.lambda1 * raw.c[2] * wi;
    Wm = MASS ? prm.lambda0 * raw.c[3] * wij : T(0);
  }

  __device__ __forceinline__ void apply(const FacetParams<T>& prm,
                                        const Mat& sm, const FacetWave<P>& w,
                                        T* lds, const T (&ua)[P],
                                        T (&acc)[P]) const {
    T* s0 = lds;
    const bool lane_ok = w.ok;
    const uint32_t own_w = w.own_w, mid_w = w.mid_w, last_w = w.last_w;
    if (prm.lambda1 != T(0)) {
      if (lane_ok) {
#pragma unroll
        for (int a = 0; a < P; ++a) s0[own_w + a * L::A] = ua[a];
      }
      {   // axis 0 in registers
        T r0[P];
        facet_sym_line<T, P, FacetParams<T>>(sm, ua, r0);
#pragma unroll
        for (int a = 0; a < P; ++a) {
          acc[a] = P0 * r0[a];
          if (MASS) acc[a] += (Wm * ua[a]) * sym_w<T, P>(sm, a);
        }
      }
      facet_sync<P>();
      T y1[P];
      {
        T x[P], y2[P];
#pragma unroll
        for (int m = 0; m < P; ++m) x[m] = lane_ok ? s0[last_w + m] : T(0);
        facet_sym_line<T, P, FacetParams<T>>(sm, x, y2);
#pragma unroll
        for (int m = 0; m < P; ++m)
          x[m] = lane_ok ? s0[mid_w + m * L::B] : T(0);
        facet_sync<P>();       // both passes have read the element
        if (lane_ok) {
#pragma unroll
          for (int m = 0; m < P; ++m) s0[last_w + m] = y2[m];
        }
        facet_sym_line<T, P, FacetParams<T>>(sm, x, y1);
      }
      facet_sync<P>();
#pragma unroll
      for (int a = 0; a < P; ++a) {
        const T t2 = lane_ok ? s0[own_w + a * L::A] : T(0);
        acc[a] += (P2 * t2) * sym_w<T, P>(sm, a);
      }
      facet_sync<P>();
      if (lane_ok) {
#pragma unroll
        for (int m = 0; m < P; ++m) s0[mid_w + m * L::B] = y1[m];
      }
      facet_sync<P>();
#pragma unroll
      for (int a = 0; a < P; ++a) {
        const T t1 = lane_ok ? s0[own_w + a * L::A] : T(0);
        acc[a] += (P1 * t1) * sym_w<T, P>(sm, a);
      }
      facet_sync<P>();
    } else {
#pragma unroll
      for (int a = 0; a < P; ++a)
        acc[a] = MASS ? (Wm * sym_w<T, P>(sm, a)) * ua[a] : T(0);
    }
  }
};

// One element per one-wave workgroup.  ELEM = FacetElem<..> / BoxElem<..>.
template <typename T, int P, typename ELEM, bool SCALAR, bool OFF32,
          bool LAY = false>
__global__ void __launch_bounds__(FacetLayout<P>::BLOCK, ELEM::MINW)
helmholtz_facet_kernel(FacetParams<T> prm, typename ELEM::Mat dm) {
  using L = FacetLayout<P>;
  using Mat = typename ELEM::Mat;
  using KA = FacetKernarg<FacetParams<T>, Mat>;
  static_assert(!LAY || SCALAR, "layered assembly takes scalar fields");
  __shared__ T lds[ELEM::LDS_WORDS];
  T* s0 = lds;
  uint32_t* codes = reinterpret_cast<uint32_t*>(lds + L::COPY);

  FacetWave<P> w;
  w.init();
  const uint32_t work = blockIdx.x;
  const int64_t e = prm.elem_list ? (int64_t)prm.elem_list[work]
                                  : (int64_t)work;
  const Mat* kdm = reinterpret_cast<const Mat*>(kernarg_bytes() + KA::MAT_OFF);

  FacetLane<P, LAY> fl;
  fl.load(prm.tab, e, w.i, w.j);
  ELEM el;
  {
    typename ELEM::Raw raw;
    ELEM::fetch(raw, prm, e);
    el.finish(raw, prm, dm, kdm, w, w.template wij<T>(kdm));
  }

  const int nc = SCALAR ? 1 : prm.ncomp;
  double udot = 0.0;
  for (int k = 0; k < nc; ++k) {
    const T* ug = prm.u + (SCALAR ? 0 : k * prm.comp_stride);
    T* og = prm.out + (SCALAR ? 0 : k * prm.comp_stride);
    T ua[P], acc[P];
    if (w.ok) {
#pragma unroll
      for (int a = 0; a < P; ++a) {
        ua[a] = *facet_node<const T, OFF32>(ug, fl.code(a));
      }
    } else {
#pragma unroll
      for (int a = 0; a < P; ++a) ua[a] = T(0);
    }
    el.apply(prm, dm, w, lds, ua, acc);
    if (w.ok)
      facet_scatter_head<T, P, LAY>(fl, acc, ua, prm.dot_out != nullptr, udot);
    if constexpr (LAY && !ELEM::LAYERED_SLOTS) {
      facet_store_layered<T, P, OFF32>(fl, acc, og, w.ok);
    } else {
      uint16_t slots[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) slots[q] = g_facet_slots<P>.s[w.lane][q];
      if constexpr (LAY)
        facet_store_layered_slots<T, P, OFF32>(
            fl, slots, acc, og, s0, codes, w.own_w, w.ok,
            FacetLane<P>::cls(w.i) == 1 && FacetLane<P>::cls(w.j) == 1);
      else
        facet_scatter_tail<T, P, OFF32>(fl, slots, acc, og, s0, codes, w.own_w,
                                        w.ok);
      facet_sync<P>();
    }
  }
  if (prm.dot_out) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) udot += __shfl_down(udot, off, 64);
    if ((w.lane & 63) == 0) {
      const int64_t wave = (int64_t)blockIdx.x * L::WAVES + (w.lane >> 6);
      if (LAY && prm.dot_slots > 0)
        prm.dot_out[wave] = udot;
      else
        unsafeAtomicAdd(&prm.dot_out[wave & (SFEM_DOT_SLOTS - 1)], udot);
    }
  }
}

// One CHAIN SEGMENT per one-wave workgroup: elements order[k0 .. k1) such that
// the face a = P-1 of each is the face a = 0 of the next, node for node in the
// lane layout (`sfem_facet_chains`).  The wave walks the segment:
//   * the shared face travels in registers: its nodal values are gathered
//     once, its two contributions are added before they leave the wave -- the
//     face interior is then complete (a face has two elements) and is stored
//     without an atomic, its edges / vertices take one atomic instead of two;
//   * the next element's table is requested together with this element's
//     gather and its gather is issued before this element's scatter, so the
//     dependent memory round trips of an element overlap with the previous
//     element's work instead of adding up.  (Gathering a whole element ahead
//     -- 34 more live registers -- measured 0.569 vs 0.580 ms at 3 waves per
//     SIMD and 0.655 with spills at 4: not kept.)
//   * layered assembly (LAY): the element that hands its face on does not
//     write it (FACET_SKIP), the receiver stores the sum into ITS layer of the
//     face's facets; everything leaves as plain stores from the lanes' own
//     slots (no LDS round trip, no slot table).
template <typename T, int P, typename ELEM, bool OFF32, bool LAY = false>
__global__ void __launch_bounds__(FacetLayout<P>::BLOCK, ELEM::CHAIN_MINW)
helmholtz_chain_kernel(FacetParams<T> prm, typename ELEM::Mat dm) {
  using L = FacetLayout<P>;
  using Mat = typename ELEM::Mat;
  using KA = FacetKernarg<FacetParams<T>, Mat>;
  using FL = FacetLane<P, LAY>;
  __shared__ T lds[ELEM::LDS_WORDS];
  T* s0 = lds;
  uint32_t* codes = reinterpret_cast<uint32_t*>(lds + L::COPY);

  FacetWave<P> w;
  w.init();
  const uint32_t seg = blockIdx.x;
  const int32_t k0 = prm.chain_off[seg];
  const int32_t k1 = prm.chain_off[seg + 1];
  const Mat* kdm = reinterpret_cast<const Mat*>(kernarg_bytes() + KA::MAT_OFF);
  constexpr bool OWN_ORDER = LAY && !ELEM::LAYERED_SLOTS;
  uint16_t slots[OWN_ORDER ? 1 : 6];
  if constexpr (!OWN_ORDER) {
#pragma unroll
    for (int q = 0; q < 6; ++q) slots[q] = g_facet_slots<P>.s[w.lane][q];
  }
  const bool face_inner = FL::cls(w.i) == 1 && FL::cls(w.j) == 1;
  const T* ug = prm.u;
  T* og = prm.out;

  const T wij = w.template wij<T>(kdm);
  double udot = 0.0;
  T carry = T(0);
  {
  FL fl, fn;
  typename FL::Raw traw;
  typename ELEM::Raw graw;
  fl.load(prm.tab, (int64_t)prm.chain_elems[k0], w.i, w.j);
  fn = fl;
  ELEM::fetch(graw, prm, (int64_t)prm.chain_elems[k0]);
  T ua[P];
#pragma unroll
  for (int a = 0; a < P; ++a)
    ua[a] = w.ok ? *facet_node<const T, OFF32>(ug, fl.code(a)) : T(0);
  if (k0 + 1 < k1)
    FL::issue(traw, prm.tab, (int64_t)prm.chain_elems[k0 + 1], w.i, w.j);
  for (int32_t k = k0; k < k1; ++k) {
    const bool has_pred = k > k0, has_succ = k + 1 < k1;
    ELEM el;
    el.finish(graw, prm, dm, kdm, w, wij);
    if (has_succ) {    // requested one element ago: arrives with the gather
      fn.finish(traw, w.i, w.j);
      ELEM::fetch(graw, prm, (int64_t)prm.chain_elems[k + 1]);
    }
    T acc[P];
    el.apply(prm, dm, w, lds, ua, acc);
    if (has_pred) acc[0] += carry;
    carry = acc[P - 1];
    // flags of this visit: the carried-in face interior is complete, the
    // carried-out face is left to the successor
    FL fe = fl;
    if constexpr (LAY) {
      if (has_succ) fe.to[2] |= FACET_SKIP | SFEM_IDX_DIRICHLET;
    } else {
      if (has_pred && face_inner) fe.t[0] &= ~(uint32_t)SFEM_IDX_SHARED;
      if (has_succ) fe.t[2] |= SFEM_IDX_SHARED | SFEM_IDX_DIRICHLET;
    }
    const T u_last = ua[P - 1];
    if (w.ok)
      facet_scatter_head<T, P, LAY>(fe, acc, ua, prm.dot_out != nullptr, udot);
    // next element: gather now, its successor's table with it
    fl = fn;
    if (has_succ) {
      ua[0] = u_last;
#pragma unroll
      for (int a = 1; a < P; ++a)
        ua[a] = w.ok ? *facet_node<const T, OFF32>(ug, fl.code(a)) : T(0);
      if (k + 2 < k1)
        FL::issue(traw, prm.tab, (int64_t)prm.chain_elems[k + 2], w.i, w.j);
    }
    if constexpr (OWN_ORDER) {
      facet_store_layered<T, P, OFF32>(fe, acc, og, w.ok);
    } else {
      if constexpr (LAY)
        facet_store_layered_slots<T, P, OFF32>(fe, slots, acc, og, s0, codes,
                                               w.own_w, w.ok, face_inner);
      else
        facet_scatter_tail<T, P, OFF32>(fe, slots, acc, og, s0, codes, w.own_w,
                                        w.ok);
      facet_sync<P>();
    }
  }
  }
  if (prm.dot_out) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) udot += __shfl_down(udot, off, 64);
    if ((w.lane & 63) == 0) {
      const int64_t wave = (int64_t)blockIdx.x * L::WAVES + (w.lane >> 6);
      if (LAY && prm.dot_slots > 0)
        prm.dot_out[wave] = udot;
      else
        unsafeAtomicAdd(&prm.dot_out[wave & (SFEM_DOT_SLOTS - 1)], udot);
    }
  }
}

// ----------------------------------------------------------------- launch ---
template <typename T, int P, typename ELEM>
int launch_facet_elem(const FacetParams<T>& prm, const typename ELEM::Mat& mat,
                      unsigned groups, bool off32, hipStream_t stream) {
  const dim3 grid(groups), block(FacetLayout<P>::BLOCK);
  if (prm.layered) {
    if (prm.ncomp != 1) {
      set_error("helmholtz (facet): layered assembly takes scalar fields");
      return SFEM_EINVAL;
    }
    if (prm.chain_off) {
      if constexpr (ELEM::CHAINS) {
        if (off32)
          hipLaunchKernelGGL((helmholtz_chain_kernel<T, P, ELEM, true, true>),
                             grid, block, 0, stream, prm, mat);
        else
          hipLaunchKernelGGL((helmholtz_chain_kernel<T, P, ELEM, false, true>),
                             grid, block, 0, stream, prm, mat);
      } else {
        set_error("helmholtz (facet): no chain kernel for this geometry at "
                  "P=%d", P);
        return SFEM_EUNSUPPORTED;
      }
    } else if (off32) {
      hipLaunchKernelGGL(
          (helmholtz_facet_kernel<T, P, ELEM, true, true, true>), grid, block,
          0, stream, prm, mat);
    } else {
      hipLaunchKernelGGL(
          (helmholtz_facet_kernel<T, P, ELEM, true, false, true>), grid, block,
          0, stream, prm, mat);
    }
    return SFEM_OK;
  }
  if (prm.chain_off) {
    if constexpr (ELEM::CHAINS) {
      if (off32)
        hipLaunchKernelGGL((helmholtz_chain_kernel<T, P, ELEM, true>), grid,
                           block, 0, stream, prm, mat);
      else
        hipLaunchKernelGGL((helmholtz_chain_kernel<T, P, ELEM, false>), grid,
                           block, 0, stream, prm, mat);
    } else {
      // (nothing was launched: the caller must not read `out`)
      set_error("helmholtz (facet): no chain kernel for this geometry at P=%d",
                P);
      return SFEM_EUNSUPPORTED;
    }
  } else if (prm.ncomp == 1) {
    if (off32)
      hipLaunchKernelGGL((helmholtz_facet_kernel<T, P, ELEM, true, true>),
                         grid, block, 0, stream, prm, mat);
    else
      hipLaunchKernelGGL((helmholtz_facet_kernel<T, P, ELEM, true, false>),
                         grid, block, 0, stream, prm, mat);
  } else {
    if (off32)
      hipLaunchKernelGGL((helmholtz_facet_kernel<T, P, ELEM, false, true>),
                         grid, block, 0, stream, prm, mat);
    else
      hipLaunchKernelGGL((helmholtz_facet_kernel<T, P, ELEM, false, false>),
                         grid, block, 0, stream, prm, mat);
  }
  return SFEM_OK;
}

// `groups`: workgroups of the launch = elements, or chain segments when
// prm.chain_off is set (scalar fields only).
template <typename T, int P>
int launch_helmholtz_facet(const FacetParams<T>& prm, int geo_mode,
                           int64_t groups, int64_t field_reals,
                           const T* dmat, const T* weights, const T* nodes,
                           hipStream_t stream) {
  if (groups > 0x7fffffff) {
    set_error("helmholtz (facet): too many workgroups (%lld)",
              (long long)groups);
    return SFEM_EINVAL;
  }
  if (prm.chain_off && prm.ncomp != 1) {
    set_error("helmholtz (facet): chain launches take scalar fields");
    return SFEM_EINVAL;
  }
  const unsigned g = (unsigned)groups;
  if (prm.dot_out && prm.layered && prm.dot_slots > 0 &&
      groups * FacetLayout<P>::WAVES > prm.dot_slots) {
    set_error("helmholtz (facet): %lld waves but %lld dot slots",
              (long long)(groups * FacetLayout<P>::WAVES),
              (long long)prm.dot_slots);
    return SFEM_EINVAL;
  }
  const bool mass = prm.lambda0 != T(0);
  // fields of 4 GiB and more take the 64-bit addressing instantiations
  // (SFEM_FACET_OFF64=1 forces them: tests)
  const char* force64 = getenv("SFEM_FACET_OFF64");
  const bool off32 = (uint64_t)field_reals * sizeof(T) < ((uint64_t)1 << 32) &&
                     !(force64 && force64[0] == '1');
  int rc = SFEM_OK;
  if (geo_mode == GEO_BOX) {
    const SMat<T, P> sm = make_smat<T, P>(dmat, weights);
    if (mass) rc = launch_facet_elem<T, P, BoxElem<T, P, true>>(prm, sm, g,
                                                                off32, stream);
    else rc = launch_facet_elem<T, P, BoxElem<T, P, false>>(prm, sm, g, off32,
                                                            stream);
  } else {
    const DMat<T, P> dm = make_dmat<T, P>(dmat, weights, nodes);
#define SFEM_FACET_GM(GMV)                                                    \
  do {                                                                        \
    if (mass) rc = launch_facet_elem<T, P, FacetElem<T, P, GMV, true>>(       \
        prm, dm, g, off32, stream);                                           \
    else rc = launch_facet_elem<T, P, FacetElem<T, P, GMV, false>>(           \
        prm, dm, g, off32, stream);                                           \
  } while (0)
    if (geo_mode == GEO_AFFINE) SFEM_FACET_GM(GEO_AFFINE);
    else if (geo_mode == GEO_MULTILINEAR) SFEM_FACET_GM(GEO_MULTILINEAR);
    else SFEM_FACET_GM(GEO_POINT);
#undef SFEM_FACET_GM
  }
  if (rc != SFEM_OK) return rc;
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

// P = 6..8 and P = 9..12 are instantiated in separate translation units per
// dtype (compile time); sfem_helmholtz.hip picks by P.
template <typename T>
int dispatch_helmholtz_facet_low(const FacetParams<T>& prm, int P,
                                 int geo_mode, int64_t groups,
                                 int64_t field_reals, const T* dmat,
                                 const T* weights, const T* nodes,
                                 hipStream_t stream);
template <typename T>
int dispatch_helmholtz_facet_high(const FacetParams<T>& prm, int P,
                                  int geo_mode, int64_t groups,
                                  int64_t field_reals, const T* dmat,
                                  const T* weights, const T* nodes,
                                  hipStream_t stream);

template <typename T>
int dispatch_helmholtz_facet(const FacetParams<T>& prm, int P, int geo_mode,
                             int64_t groups, int64_t field_reals,
                             const T* dmat, const T* weights, const T* nodes,
                             hipStream_t stream) {
  if (P <= 8)
    return dispatch_helmholtz_facet_low<T>(prm, P, geo_mode, groups,
                                           field_reals, dmat, weights, nodes,
                                           stream);
  return dispatch_helmholtz_facet_high<T>(prm, P, geo_mode, groups,
                                          field_reals, dmat, weights, nodes,
                                          stream);
}

inline bool facet_supported_p(int P) { return P >= 6 && P <= 12; }

#define SFEM_FACET_CASE(PP)                                                   \
  case PP:                                                                    \
    return launch_helmholtz_facet<T, PP>(prm, geo_mode, groups, field_reals,  \
                                         dmat, weights, nodes, stream);

#define SFEM_DEFINE_FACET_DISPATCH_LOW(TYPE)                                  \
  template <>                                                                 \
  int dispatch_helmholtz_facet_low<TYPE>(                                     \
      const FacetParams<TYPE>& prm, int P, int geo_mode, int64_t groups,      \
      int64_t field_reals, const TYPE* dmat, const TYPE* weights,             \
      const TYPE* nodes, hipStream_t stream) {                                \
    using T = TYPE;                                                           \
    switch (P) {                                                              \
      SFEM_FACET_CASE(6) SFEM_FACET_CASE(7) SFEM_FACET_CASE(8)                \
      default:                                                                \
        set_error("helmholtz (facet): P=%d outside 6..8", P);                 \
        return SFEM_EUNSUPPORTED;                                             \
    }                                                                         \
  }

#define SFEM_DEFINE_FACET_DISPATCH_HIGH(TYPE)                                 \
  template <>                                                                 \
  int dispatch_helmholtz_facet_high<TYPE>(                                    \
      const FacetParams<TYPE>& prm, int P, int geo_mode, int64_t groups,      \
      int64_t field_reals, const TYPE* dmat, const TYPE* weights,             \
      const TYPE* nodes, hipStream_t stream) {                                \
    using T = TYPE;                                                           \
    switch (P) {                                                              \
      SFEM_FACET_CASE(9) SFEM_FACET_CASE(10) SFEM_FACET_CASE(11)              \
      SFEM_FACET_CASE(12)                                                     \
      default:                                                                \
        set_error("helmholtz (facet): P=%d outside 9..12", P);                \
        return SFEM_EUNSUPPORTED;                                             \
    }                                                                         \
  }

}  // namespace sfem
