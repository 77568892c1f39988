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
  int64_t comp_stride;       // component k of node n at u[n + k comp_stride]
  int ncomp;
  T lambda0, lambda1;
  double* dot_out;
};

// Where node (x, y, z) of an element sits in its LDS copy (in words of T).
template <int P>
struct FacetLayout {
  static constexpr int A = P == 8 ? 67 : (P == 7 ? 52 : P * P);
  static constexpr int B = P;
  static constexpr bool SWAP = P == 8;   // transposed passes: lane (i, j) takes
                                         // the line with first index j
  static constexpr int WORDS = (P - 1) * (A + B + 1) + 1;
  static constexpr int COPY = (WORDS + 1) & ~1;
  __host__ __device__ static constexpr int word(int x, int y, int z) {
    return A * x + B * y + z;
  }
};

// Shared scatter: in round q the lanes of the wave take one face interior
// ((P-2)^2 lanes), two edge interiors (2 (P-2) lanes) and -- round 0 -- the
// eight vertices, so that one atomic instruction covers whole contiguous node
// blocks.  slot[lane][q] = LDS word of the node the lane handles, 0xFFFF none.
struct alignas(16) FacetSlots {
  uint32_t pk[64][4];   // slots 2k, 2k+1 of a lane in the halves of pk[lane][k]
  void set(int lane, int q, unsigned w) {
    const unsigned sh = 16 * (q & 1);
    pk[lane][q >> 1] = (pk[lane][q >> 1] & ~(0xFFFFu << sh)) | (w << sh);
  }
};

template <int P>
inline FacetSlots make_facet_slots() {
  using L = FacetLayout<P>;
  FacetSlots s;
  for (int l = 0; l < 64; ++l)
    for (int k = 0; k < 4; ++k) s.pk[l][k] = 0xFFFFFFFFu;
  constexpr int M = P - 2, NF = M * M;
  static_assert(NF + 2 * M + 8 <= 64, "one wave per element");
  // edges: 4 along j, 4 along i, 4 along a
  int edges[12][3];   // fixed coordinates, -1 = running
  int ne = 0;
  for (int d = 2; d >= 0; --d)
    for (int p = 0; p < 2; ++p)
      for (int r = 0; r < 2; ++r) {
        int c[3];
        c[d] = -1;
        c[(d + 1) % 3] = p ? P - 1 : 0;
        c[(d + 2) % 3] = r ? P - 1 : 0;
        for (int k = 0; k < 3; ++k) edges[ne][k] = c[k];
        ++ne;
      }
  for (int q = 0; q < 6; ++q) {
    const int d = q / 2, fixed = (q & 1) ? P - 1 : 0;   // face normal axis d
    for (int l = 0; l < NF; ++l) {
      const int x = l / M + 1, y = l % M + 1;
      int c[3];
      c[d] = fixed;
      c[(d + 1) % 3] = d == 1 ? y : x;   // keep (x, y) in ascending axis order
      c[(d + 2) % 3] = d == 1 ? x : y;
      s.set(l, q, (unsigned)L::word(c[0], c[1], c[2]));
    }
    for (int h = 0; h < 2; ++h)
      for (int z = 0; z < M; ++z) {
        const int* ed = edges[2 * q + h];
        int c[3];
        for (int k = 0; k < 3; ++k) c[k] = ed[k] < 0 ? z + 1 : ed[k];
        s.set(NF + h * M + z, q, (unsigned)L::word(c[0], c[1], c[2]));
      }
  }
  for (int v = 0; v < 8; ++v)
    s.set(NF + 2 * M + v, 0,
          (unsigned)L::word((v & 4) ? P - 1 : 0, (v & 2) ? P - 1 : 0,
                            (v & 1) ? P - 1 : 0));
  return s;
}

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
template <int P>
struct FacetLane {
  uint32_t t[3];
  int32_t sa;
  static __device__ __forceinline__ int cls(int a) {
    return a == 0 ? 0 : (a == P - 1 ? 2 : 1);
  }
  __device__ __forceinline__ void load(const int32_t* tab, int64_t e, int i,
                                       int j) {
    typedef int32_t I4 __attribute__((ext_vector_type(4)));
    const int lc = cls(i) * 3 + cls(j);
    const I4* row = reinterpret_cast<const I4*>(tab + e * FACET_ROW);
    I4 en[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) en[c] = row[c * 9 + lc];
#pragma unroll
    for (int c = 0; c < 3; ++c)
      t[c] = (uint32_t)en[c].x + (uint32_t)__mul24(en[c].z, i - 1) +
             (uint32_t)__mul24(en[c].w, j - 1);
    sa = en[1].y;
  }
  // node id | flags of the lane's node in slice a (compile-time a)
  __device__ __forceinline__ uint32_t code(int a) const {
    return a == 0 ? t[0]
                  : (a == P - 1 ? t[2] : t[1] + (uint32_t)(sa * (a - 1)));
  }
  __device__ __forceinline__ uint32_t flags(int a) const {
    return t[cls(a)];
  }
};

// Direct-stiffness summation of one element's results `acc` (slot layout:
// lane (i, j) holds nodes (a, i, j)).  `vals` / `codes`: free LDS of
// FacetLayout<P>::WORDS words of T resp. uint32.
template <typename T, int P, bool OFF32>
__device__ __forceinline__ void facet_scatter(
    const FacetLane<P>& fl, const uint32_t (&slots)[3], T (&acc)[P],
    const T (&ua)[P], T* og, T* vals, uint32_t* codes, uint32_t own_w,
    bool want_dot, double& udot) {
  using L = FacetLayout<P>;
  // Dirichlet rows are zero
#pragma unroll
  for (int a = 0; a < P; ++a)
    if (fl.flags(a) & SFEM_IDX_DIRICHLET) acc[a] = T(0);
  if (want_dot) {
#pragma unroll
    for (int a = 0; a < P; ++a) udot += (double)acc[a] * (double)ua[a];
  }
  // nodes of this element only: plain stores
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (!(fl.t[c] & SFEM_IDX_SHARED)) {
#pragma unroll
      for (int a = 0; a < P; ++a)
        if (FacetLane<P>::cls(a) == c) {
#if SFEM_FACET_TIMING == 5
          asm volatile("" :: "v"(acc[a]));
#else
          *facet_node<T, OFF32>(og, fl.code(a)) = acc[a];
#endif
        }
    }
  }
  // shared nodes: values and codes change lanes through LDS
#pragma unroll
  for (int a = 0; a < P; ++a) {
    vals[own_w + a * L::A] = acc[a];
    codes[own_w + a * L::A] = fl.code(a);
  }
  wave_sync();
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const uint32_t w = (slots[q >> 1] >> (16 * (q & 1))) & 0xFFFFu;
    if (w != 0xFFFFu) {
      const uint32_t code = codes[w];
      if ((code & ~(uint32_t)SFEM_IDX_MASK) == SFEM_IDX_SHARED) {
#if SFEM_FACET_TIMING == 3 || SFEM_FACET_TIMING == 6     // timing experiments only (wrong sums)
        asm volatile("" :: "v"(vals[w]), "v"(code));
#elif SFEM_FACET_TIMING == 1
        *facet_node<T, OFF32>(og, code) = vals[w];
#elif SFEM_FACET_TIMING == 2
        T* dst = facet_node<T, OFF32>(og, code);
        *dst = *dst + vals[w];
#else
        unsafeAtomicAdd(facet_node<T, OFF32>(og, code), vals[w]);
#endif
      }
    }
  }
}

// Kernel-argument offsets (the matrices and the slot table are read through
// the kernarg segment with lane-dependent indices).
template <typename PRM, typename MAT>
struct FacetKernarg {
  static constexpr size_t MAT_OFF =
      (sizeof(PRM) + alignof(MAT) - 1) / alignof(MAT) * alignof(MAT);
  static constexpr size_t SLOT_OFF =
      (MAT_OFF + sizeof(MAT) + alignof(FacetSlots) - 1) / alignof(FacetSlots) *
      alignof(FacetSlots);
};

__device__ __forceinline__ const char* kernarg_bytes() {
#if defined(__HIP_DEVICE_COMPILE__)
  return (const char*)(const __attribute__((address_space(4))) char*)
      __builtin_amdgcn_kernarg_segment_ptr();
#else
  return nullptr;
#endif
}

template <int P>
struct FacetWave {
  static constexpr int TPE = P * P;
  static constexpr int LDS_WORDS = 2 * FacetLayout<P>::COPY;
};

#ifndef SFEM_FACET_TIMING
#define SFEM_FACET_TIMING 0
#endif
#ifndef SFEM_FACET_XCD
#define SFEM_FACET_XCD 0
#endif
// Workgroups are dealt round-robin to the 8 XCDs (each with its own L2):
// blocks b and b + 8 share one.  The remap hands every XCD a contiguous run of
// the element list, so that elements that share faces (neighbours in the
// list) meet in one L2.
__device__ __forceinline__ uint32_t facet_work_item() {
#if SFEM_FACET_XCD
  const uint32_t b = blockIdx.x, g = gridDim.x;
  const uint32_t chunk = g >> 3, rem = g & 7;     // XCD x takes chunk (+1 if x < rem)
  const uint32_t x = b & 7, s = b >> 3;
  return x * chunk + (x < rem ? x : rem) + s;
#else
  return blockIdx.x;
#endif
}
#ifndef SFEM_FACET_AFFINE_MINW
#define SFEM_FACET_AFFINE_MINW 5
#endif
#ifndef SFEM_FACET_BOX_MINW
#define SFEM_FACET_BOX_MINW 5
#endif

// 1D quadrature weight of slice a; the node sets are symmetric (supports_fused),
// so only the first half of the by-value copy is ever read (fewer SGPRs).
template <typename T, int P, typename M>
__device__ __forceinline__ T sym_w(const M& m, int a) {
  return m.w[a < P - 1 - a ? a : P - 1 - a];
}

// General geometry: GM = GEO_POINT / GEO_AFFINE / GEO_MULTILINEAR.
template <typename T, int P, int GM, bool MASS, bool SCALAR, bool OFF32>
__global__ void __launch_bounds__(
    64, (GM == GEO_AFFINE ? SFEM_FACET_AFFINE_MINW : 4))
helmholtz_facet_kernel(FacetParams<T> prm, DMat<T, P> dm, FacetSlots st) {
  using L = FacetLayout<P>;
  using KA = FacetKernarg<FacetParams<T>, DMat<T, P>>;
  constexpr int TPE = P * P;
  __shared__ T lds[2 * L::COPY];
  T* s0 = lds;
  T* s1 = lds + L::COPY;

  const int lane = threadIdx.x;
  const bool lane_ok = TPE == 64 || lane < TPE;
  const int t = lane_ok ? lane : 0;
  const int i = t / P, j = t - i * P;
  const uint32_t work = facet_work_item();
  const int64_t e = prm.elem_list ? (int64_t)prm.elem_list[work]
                                  : (int64_t)work;
  const DMat<T, P>* kdm =
      reinterpret_cast<const DMat<T, P>*>(kernarg_bytes() + KA::MAT_OFF);
  const FacetSlots* kst =
      reinterpret_cast<const FacetSlots*>(kernarg_bytes() + KA::SLOT_OFF);

  FacetLane<P> fl;
  fl.load(prm.tab, e, i, j);

  // LDS words of the lane's three access patterns
  const uint32_t own_w = L::B * i + j;
  const uint32_t mid_w = L::SWAP ? L::A * j + i : L::A * i + j;
  const uint32_t last_w = L::SWAP ? L::A * j + L::B * i : L::A * i + L::B * j;

  // geometry
  const T* cst = prm.geo_const + e * 8;          // affine: wave-uniform
  T lw = T(0), Wm0 = T(0);
  ElemGeom<T, P, 3, (GM == GEO_AFFINE ? GEO_POINT : GM)> geom;
  if constexpr (GM == GEO_AFFINE) {
    const T wij = kdm->w[i] * kdm->w[j];
    lw = prm.lambda1 * wij;
    if (MASS) Wm0 = prm.lambda0 * cst[6] * wij;
  } else {
    HelmholtzParams<T> hp{};
    hp.geo = prm.geo;
    hp.geo_elem = prm.geo_elem;
    hp.geo_index = prm.geo_index;
    geom.template init<true>(hp, dm, e, true, i, j, t, kdm);
  }

  const int nc = SCALAR ? 1 : prm.ncomp;
  double udot = 0.0;
  for (int k = 0; k < nc; ++k) {
    const T* ug = prm.u + (SCALAR ? 0 : k * prm.comp_stride);
    T* og = prm.out + (SCALAR ? 0 : k * prm.comp_stride);
    T ua[P], acc[P];
    if (lane_ok) {
#pragma unroll
      for (int a = 0; a < P; ++a)
#if SFEM_FACET_TIMING == 4 || SFEM_FACET_TIMING == 6
        ua[a] = (T)fl.code(a);
#else
        ua[a] = *facet_node<const T, OFF32>(ug, fl.code(a));
#endif
    } else {
#pragma unroll
      for (int a = 0; a < P; ++a) ua[a] = T(0);
    }
    const bool has_stiff = prm.lambda1 != T(0);
    if (has_stiff) {
      if (lane_ok) {
#pragma unroll
        for (int a = 0; a < P; ++a) {
          s0[own_w + a * L::A] = ua[a];
          s1[own_w + a * L::A] = ua[a];
        }
      }
      wave_sync();
      if (lane_ok) {   // last axis, copy 1
        T x[P], y[P];
#pragma unroll
        for (int m = 0; m < P; ++m) x[m] = s1[last_w + m];
        line_apply<T, P, false>(dm, x, y);
#pragma unroll
        for (int m = 0; m < P; ++m) s1[last_w + m] = y[m];
      }
      if (lane_ok) {   // middle axis, copy 0
        T x[P], y[P];
#pragma unroll
        for (int m = 0; m < P; ++m) x[m] = s0[mid_w + m * L::B];
        line_apply<T, P, false>(dm, x, y);
#pragma unroll
        for (int m = 0; m < P; ++m) s0[mid_w + m * L::B] = y[m];
      }
      // axis 0 in registers; w0 takes the place of d0 slice by slice
      T w0[P];
      line_apply<T, P, false>(dm, ua, w0);
      wave_sync();
#pragma unroll
      for (int a = 0; a < P; ++a) acc[a] = T(0);
      if (lane_ok) {
#pragma unroll
        for (int a = 0; a < P; ++a) {
          T& r0 = s0[own_w + a * L::A];
          T& r1 = s1[own_w + a * L::A];
          if constexpr (GM == GEO_AFFINE) {
            const T g0 = w0[a], g1 = r0, g2 = r1;
            const T wa = sym_w<T, P>(dm, a);
            const T sc = lw * wa;
            w0[a] = sc * (cst[0] * g0 + cst[1] * g1 + cst[2] * g2);
            r0 = sc * (cst[1] * g0 + cst[3] * g1 + cst[4] * g2);
            r1 = sc * (cst[2] * g0 + cst[4] * g1 + cst[5] * g2);
            if (MASS) acc[a] = (Wm0 * wa) * ua[a];
          } else if constexpr (GM == GEO_MULTILINEAR) {
            T o0, o1, o2, Wm;
            geom.apply_multilinear3(dm, a, MASS, w0[a], r0, r1, o0, o1, o2, Wm);
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
        line_apply<T, P, true>(dm, w0, dt0);
#pragma unroll
        for (int a = 0; a < P; ++a) {
          if (GM == GEO_AFFINE) acc[a] += dt0[a];    // lambda1 is inside lw
          else acc[a] += prm.lambda1 * dt0[a];
        }
      }
      wave_sync();
      if (lane_ok) {
        T x[P], y[P];
#pragma unroll
        for (int m = 0; m < P; ++m) x[m] = s1[last_w + m];
        line_apply<T, P, true>(dm, x, y);
#pragma unroll
        for (int m = 0; m < P; ++m) s1[last_w + m] = y[m];
      }
      if (lane_ok) {
        T x[P], y[P];
#pragma unroll
        for (int m = 0; m < P; ++m) x[m] = s0[mid_w + m * L::B];
        line_apply<T, P, true>(dm, x, y);
#pragma unroll
        for (int m = 0; m < P; ++m) s0[mid_w + m * L::B] = y[m];
      }
      wave_sync();
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
      wave_sync();
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
    if (lane_ok) {
      uint32_t slots[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) slots[q] = kst->pk[lane][q];
      facet_scatter<T, P, OFF32>(fl, slots, acc, ua, og, s0,
                                 reinterpret_cast<uint32_t*>(s1), own_w,
                                 prm.dot_out != nullptr, udot);
    }
    wave_sync();
  }
  if (prm.dot_out) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) udot += __shfl_down(udot, off, 64);
    if (lane == 0)
      unsafeAtomicAdd(&prm.dot_out[blockIdx.x & (SFEM_DOT_SLOTS - 1)], udot);
  }
}

// Box elements: J^-1 J^-T diagonal and constant.  With K = D^T diag(w) D,
//   A_loc u (a,i,j) = c0 w_i w_j (K u)_a + w_a ( c1 w_j (K u)_i + c2 w_i (K u)_j )
//   B_loc u = detJ w_a w_i w_j u
// One LDS copy: both transposed passes read it before either writes back.
template <typename T, int P, bool MASS, bool SCALAR, bool OFF32>
__global__ void __launch_bounds__(64, SFEM_FACET_BOX_MINW)
helmholtz_box_kernel(FacetParams<T> prm, SMat<T, P> sm, FacetSlots st) {
  using L = FacetLayout<P>;
  using KA = FacetKernarg<FacetParams<T>, SMat<T, P>>;
  constexpr int TPE = P * P;
  // one copy of the element and the codes of the shared scatter
  constexpr int CODE_WORDS = (L::COPY * 4 + (int)sizeof(T) - 1) / (int)sizeof(T);
  __shared__ T lds[L::COPY + CODE_WORDS];
  T* s0 = lds;
  T* s1 = lds + L::COPY;

  const int lane = threadIdx.x;
  const bool lane_ok = TPE == 64 || lane < TPE;
  const int t = lane_ok ? lane : 0;
  const int i = t / P, j = t - i * P;
  const uint32_t work = facet_work_item();
  const int64_t e = prm.elem_list ? (int64_t)prm.elem_list[work]
                                  : (int64_t)work;
  const SMat<T, P>* ksm =
      reinterpret_cast<const SMat<T, P>*>(kernarg_bytes() + KA::MAT_OFF);
  const FacetSlots* kst =
      reinterpret_cast<const FacetSlots*>(kernarg_bytes() + KA::SLOT_OFF);

  FacetLane<P> fl;
  fl.load(prm.tab, e, i, j);

  const uint32_t own_w = L::B * i + j;
  const uint32_t mid_w = L::SWAP ? L::A * j + i : L::A * i + j;
  const uint32_t last_w = L::SWAP ? L::A * j + L::B * i : L::A * i + L::B * j;

  const T* cst = prm.geo_const + e * 8;
  const T wi = ksm->w[i], wj = ksm->w[j];
  const T P0 = prm.lambda1 * cst[0] * (wi * wj);
  const T P1 = prm.lambda1 * cst[3] * wj;
  const T P2 = prm.lambda1 * cst[5] * wi;
  const T Wm = MASS ? prm.lambda0 * cst[6] * (wi * wj) : T(0);

  const int nc = SCALAR ? 1 : prm.ncomp;
  double udot = 0.0;
  for (int k = 0; k < nc; ++k) {
    const T* ug = prm.u + (SCALAR ? 0 : k * prm.comp_stride);
    T* og = prm.out + (SCALAR ? 0 : k * prm.comp_stride);
    T ua[P], acc[P];
    if (lane_ok) {
#pragma unroll
      for (int a = 0; a < P; ++a)
#if SFEM_FACET_TIMING == 4 || SFEM_FACET_TIMING == 6
        ua[a] = (T)fl.code(a);
#else
        ua[a] = *facet_node<const T, OFF32>(ug, fl.code(a));
#endif
    } else {
#pragma unroll
      for (int a = 0; a < P; ++a) ua[a] = T(0);
    }
    if (prm.lambda1 != T(0)) {
      if (lane_ok) {
#pragma unroll
        for (int a = 0; a < P; ++a) s0[own_w + a * L::A] = ua[a];
      }
      {   // axis 0 in registers
        T r0[P];
        sym_line_apply<T, P>(sm, ua, r0);
#pragma unroll
        for (int a = 0; a < P; ++a) {
          acc[a] = P0 * r0[a];
          if (MASS) acc[a] += (Wm * sym_w<T, P>(sm, a)) * ua[a];
        }
      }
      wave_sync();
      T y1[P];
      {
        T x[P], y2[P];
#pragma unroll
        for (int m = 0; m < P; ++m) x[m] = lane_ok ? s0[last_w + m] : T(0);
        sym_line_apply<T, P>(sm, x, y2);
#pragma unroll
        for (int m = 0; m < P; ++m)
          x[m] = lane_ok ? s0[mid_w + m * L::B] : T(0);
        wave_sync();       // both passes have read the element
        if (lane_ok) {
#pragma unroll
          for (int m = 0; m < P; ++m) s0[last_w + m] = y2[m];
        }
        sym_line_apply<T, P>(sm, x, y1);
      }
      wave_sync();
#pragma unroll
      for (int a = 0; a < P; ++a) {
        const T t2 = lane_ok ? s0[own_w + a * L::A] : T(0);
        acc[a] += (P2 * sym_w<T, P>(sm, a)) * t2;
      }
      wave_sync();
      if (lane_ok) {
#pragma unroll
        for (int m = 0; m < P; ++m) s0[mid_w + m * L::B] = y1[m];
      }
      wave_sync();
#pragma unroll
      for (int a = 0; a < P; ++a) {
        const T t1 = lane_ok ? s0[own_w + a * L::A] : T(0);
        acc[a] += (P1 * sym_w<T, P>(sm, a)) * t1;
      }
      wave_sync();
    } else {
#pragma unroll
      for (int a = 0; a < P; ++a)
        acc[a] = MASS ? (Wm * sym_w<T, P>(sm, a)) * ua[a] : T(0);
    }
    if (lane_ok) {
      uint32_t slots[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) slots[q] = kst->pk[lane][q];
      facet_scatter<T, P, OFF32>(fl, slots, acc, ua, og, s0,
                                 reinterpret_cast<uint32_t*>(s1), own_w,
                                 prm.dot_out != nullptr, udot);
    }
    wave_sync();
  }
  if (prm.dot_out) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) udot += __shfl_down(udot, off, 64);
    if (lane == 0)
      unsafeAtomicAdd(&prm.dot_out[blockIdx.x & (SFEM_DOT_SLOTS - 1)], udot);
  }
}

// ----------------------------------------------------------------- launch ---
template <typename T, int P>
int launch_helmholtz_facet(const FacetParams<T>& prm, int geo_mode,
                           int64_t num_elements, int64_t field_reals,
                           const T* dmat, const T* weights, const T* nodes,
                           hipStream_t stream) {
  if (num_elements > 0x7fffffff) {
    set_error("helmholtz (facet): too many workgroups (%lld)",
              (long long)num_elements);
    return SFEM_EINVAL;
  }
  static const FacetSlots slots = make_facet_slots<P>();
  const dim3 grid((unsigned)num_elements), block(64);
  const bool mass = prm.lambda0 != T(0);
  const bool scalar = prm.ncomp == 1;
  const bool off32 = (uint64_t)field_reals * sizeof(T) < ((uint64_t)1 << 32);
#define SFEM_FACET_GO(KERNEL, MAT)                                            \
  do {                                                                        \
    if (mass) {                                                               \
      if (scalar) {                                                           \
        if (off32) hipLaunchKernelGGL((KERNEL(true, true, true)), grid,       \
                                      block, 0, stream, prm, MAT, slots);     \
        else hipLaunchKernelGGL((KERNEL(true, true, false)), grid, block, 0,  \
                                stream, prm, MAT, slots);                     \
      } else {                                                                \
        if (off32) hipLaunchKernelGGL((KERNEL(true, false, true)), grid,      \
                                      block, 0, stream, prm, MAT, slots);     \
        else hipLaunchKernelGGL((KERNEL(true, false, false)), grid, block, 0, \
                                stream, prm, MAT, slots);                     \
      }                                                                       \
    } else {                                                                  \
      if (scalar) {                                                           \
        if (off32) hipLaunchKernelGGL((KERNEL(false, true, true)), grid,      \
                                      block, 0, stream, prm, MAT, slots);     \
        else hipLaunchKernelGGL((KERNEL(false, true, false)), grid, block, 0, \
                                stream, prm, MAT, slots);                     \
      } else {                                                                \
        if (off32) hipLaunchKernelGGL((KERNEL(false, false, true)), grid,     \
                                      block, 0, stream, prm, MAT, slots);     \
        else hipLaunchKernelGGL((KERNEL(false, false, false)), grid, block,   \
                                0, stream, prm, MAT, slots);                  \
      }                                                                       \
    }                                                                         \
  } while (0)
#define SFEM_FACET_BOX(M, S, O) helmholtz_box_kernel<T, P, M, S, O>
#define SFEM_FACET_AFF(M, S, O) \
  helmholtz_facet_kernel<T, P, GEO_AFFINE, M, S, O>
#define SFEM_FACET_MUL(M, S, O) \
  helmholtz_facet_kernel<T, P, GEO_MULTILINEAR, M, S, O>
#define SFEM_FACET_PNT(M, S, O) \
  helmholtz_facet_kernel<T, P, GEO_POINT, M, S, O>
  if (geo_mode == GEO_BOX) {
    const SMat<T, P> sm = make_smat<T, P>(dmat, weights);
    SFEM_FACET_GO(SFEM_FACET_BOX, sm);
  } else {
    const DMat<T, P> dm = make_dmat<T, P>(dmat, weights, nodes);
    if (geo_mode == GEO_AFFINE) SFEM_FACET_GO(SFEM_FACET_AFF, dm);
    else if (geo_mode == GEO_MULTILINEAR) SFEM_FACET_GO(SFEM_FACET_MUL, dm);
    else SFEM_FACET_GO(SFEM_FACET_PNT, dm);
  }
#undef SFEM_FACET_GO
#undef SFEM_FACET_BOX
#undef SFEM_FACET_AFF
#undef SFEM_FACET_MUL
#undef SFEM_FACET_PNT
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

// Defined once per dtype translation unit.
template <typename T>
int dispatch_helmholtz_facet(const FacetParams<T>& prm, int P, int geo_mode,
                             int64_t num_elements, int64_t field_reals,
                             const T* dmat, const T* weights, const T* nodes,
                             hipStream_t stream);

inline bool facet_supported_p(int P) { return P >= 6 && P <= 8; }

#define SFEM_DEFINE_FACET_DISPATCH(TYPE)                                      \
  template <>                                                                 \
  int dispatch_helmholtz_facet<TYPE>(                                         \
      const FacetParams<TYPE>& prm, int P, int geo_mode,                      \
      int64_t num_elements, int64_t field_reals, const TYPE* dmat,            \
      const TYPE* weights, const TYPE* nodes, hipStream_t stream) {           \
    switch (P) {                                                              \
      case 6:                                                                 \
        return launch_helmholtz_facet<TYPE, 6>(prm, geo_mode, num_elements,   \
                                               field_reals, dmat, weights,    \
                                               nodes, stream);                \
      case 7:                                                                 \
        return launch_helmholtz_facet<TYPE, 7>(prm, geo_mode, num_elements,   \
                                               field_reals, dmat, weights,    \
                                               nodes, stream);                \
      case 8:                                                                 \
        return launch_helmholtz_facet<TYPE, 8>(prm, geo_mode, num_elements,   \
                                               field_reals, dmat, weights,    \
                                               nodes, stream);                \
      default:                                                                \
        set_error("helmholtz (facet): P=%d outside 6..8", P);                 \
        return SFEM_EUNSUPPORTED;                                             \
    }                                                                         \
  }

}  // namespace sfem
