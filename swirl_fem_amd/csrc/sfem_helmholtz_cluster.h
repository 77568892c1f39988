// Cluster-assembled variant of the fused Helmholtz operator (3D, P = 4..8).
//
//   out = mask * scatter( lambda0 * B_loc(g) + lambda1 * A_loc(g) ),  g = gather(u)
//
// Same operator, same reference call sites and the same element arithmetic as
// helmholtz_kernel (sfem_helmholtz.h); what changes is the direct-stiffness
// summation (core/gather_scatter.py:130-133).  There every shared slot of every
// element is one HBM atomic (296 of 512 slots at p = 7).  Here one workgroup
// owns a CLUSTER of up to 8 neighbouring elements (2x2x2 on a structured mesh;
// any 8 elements that sit together on an unstructured one) and sums the nodes
// they share in LDS first:
//
//   * the nodes on the lattice boundary of the cluster's elements (1647 for
//     2x2x2 at p = 7) are listed once, in ascending node order, in the cluster
//     table; element slots that refer to them carry the POSITION in that table
//     instead of the node id;
//   * the workgroup gathers u at those nodes once (a sorted, coalesced sweep of
//     the table) into an LDS strip, every element picks its values up from
//     there; lattice-interior nodes are gathered directly as before;
//   * after the element work the waves add their boundary-slot results into
//     the same strip (ds_add, no HBM traffic); a second sweep of the table
//     stores the nodes that are complete inside the cluster with plain stores
//     (469 of the 1647) and adds the cluster-surface nodes to HBM atomically
//     (1178 per cluster = 147 per element instead of 296).
//
// One wave per element (several small elements per wave for P <= 5), so all
// synchronisation inside the element work is wave-local; the workgroup meets at
// four barriers per component.  LDS: 2 x 4 KB per fp64 element at P = 8 (rows
// unpadded, XOR-swizzled so that the three access patterns of the transposes
// stay bank-conflict free, see ClusterLayout: conflict cycles 6 % of the LDS
// cycles instead of 47 %) + the strip = 79.5 KB, two workgroups per CU.
//
// MEASURED (MI355X, config 2, round 2; profiles/r02_cluster_notes.md): it halves
// the atomic requests but is SLOWER than helmholtz_kernel with its sorted
// scatter, 1.04 vs 0.78 ms per apply: once the atomics are issued in node order
// they cost that kernel only 4 % (plain stores instead: 0.81 -> 0.78 ms), and
// coupling eight waves through barriers lengthens every element's life (37 k
// instead of 24 k cycles per wave at the same 16 waves per CU).  Kept as
// `assembly='cluster'` (tested, not the default).
#pragma once
#include "sfem_helmholtz.h"

#ifndef SFEM_CL_LDS_BARRIER
#define SFEM_CL_LDS_BARRIER 1
#endif
#if SFEM_CL_LDS_BARRIER
#define SFEM_CL_BARRIER() lds_barrier()
#else
#define SFEM_CL_BARRIER() __syncthreads()
#endif
#ifndef SFEM_CL_SIZE
#define SFEM_CL_SIZE 8
#endif
#ifndef SFEM_CL_MINW
#define SFEM_CL_MINW 4
#endif
namespace sfem {

template <typename T>
struct ClusterParams {
  const int32_t* elems;     // (C, CL) element ids of each cluster, -1 = none
  const int32_t* offsets;   // (C + 1,) start of each cluster in `nodes`
  const uint32_t* nodes;    // cluster tables: node id | DIRICHLET | SHARED,
                            //   SHARED = also held outside the cluster (atomic)
  int64_t num_clusters;
};

template <typename T, int P>
struct ClusterTile {
  static constexpr int TPE = P * P;                 // lanes per element
  static constexpr int EPW = 64 / TPE;              // elements per wave
  static constexpr int CL = SFEM_CL_SIZE;           // elements per cluster
  static constexpr int NW = (CL + EPW - 1) / EPW;   // waves per workgroup
  static constexpr int BLOCK = NW * 64;
  static constexpr int N = P * P * P;
  static constexpr int WORDS = ClusterLayout<T, P>::WORDS;
  static constexpr int ELEM_BYTES = 2 * WORDS * (int)sizeof(T);
  // shared nodes of a cluster: at most CL * (n - (P-2)^3); the LDS strip is
  // also capped so that 16 waves (the 128-VGPR occupancy) fit the 160 KB of a
  // CU: 10 KB per wave of the workgroup
  static constexpr int SHARED_PER_ELEM = N - (P - 2) * (P - 2) * (P - 2);
  static constexpr int LDS_BUDGET = NW >= 4 ? NW * 10 * 1024 : 22 * 1024;
  static constexpr int BUDGET =
      (LDS_BUDGET - 64 - CL * ELEM_BYTES) / (int)sizeof(T);
  static constexpr int KRAW =
      CL * SHARED_PER_ELEM < BUDGET ? CL * SHARED_PER_ELEM : BUDGET;
  static constexpr int KPT = (KRAW + BLOCK - 1) / BLOCK;   // entries per thread
  static constexpr int KMAX = KRAW / 64 * 64;
  static_assert(EPW >= 1, "one element must fit a wave");
  static_assert(KMAX >= SHARED_PER_ELEM, "LDS strip smaller than one element");
};

// enc (cluster form).  A slot on the BOUNDARY of the element's node lattice
// (a, i or j equal to 0 or P-1: the only slots a conforming mesh can share)
// carries DIRICHLET bit | POSITION of its node in the cluster table, a slot
// inside the lattice carries DIRICHLET bit | node id.  Which kind a slot is
// follows from its place, so the kernel tests no per-slot flags: slices 0 and
// P-1 go through the table with all lanes, the slices between with the lanes
// of the lattice rim (`edge`), and Dirichlet rows are zeroed by a select.
// Elements with padding slots (-1) are not clustered (the host falls back to
// helmholtz_kernel for such a mesh).
// NS1: u / out have unit node stride (scalar fields, component-major vectors).
template <typename T, int P, bool SCALAR, bool NS1, int GM, bool MASS>
__global__ void __launch_bounds__((ClusterTile<T, P>::BLOCK),
                                  (SFEM_CL_MINW))
helmholtz_cluster_kernel(DMat<T, P> dm, HelmholtzParams<T> prm,
                         ClusterParams<T> cl) {
  using Tile = ClusterTile<T, P>;
  using Lay = ClusterLayout<T, P>;
  constexpr int TPE = Tile::TPE, EPW = Tile::EPW, CL = Tile::CL;
  constexpr int BLOCK = Tile::BLOCK, N = Tile::N, W = Tile::WORDS;
  constexpr int KPT = Tile::KPT, KMAX = Tile::KMAX;
  __shared__ __attribute__((aligned(512))) T lds[CL * 2 * W + KMAX];
  T* strip = lds + CL * 2 * W;   // shared nodes of the cluster: u, then sums

  const int tid = threadIdx.x;
  // wave-uniform by construction: say so, or every per-element address
  // (index row, geometry, LDS copies) is computed per lane in VGPRs
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sub = lane / TPE;               // element within the wave
  const int t = lane - sub * TPE;           // lane within the element
  const int i = t / P, j = t - i * P;
  const int el = wave * EPW + sub;          // element within the cluster
  const bool lane_ok = sub < EPW && el < CL;
  const int64_t c = blockIdx.x;
  const int32_t eid = lane_ok ? cl.elems[c * CL + el] : -1;
  const bool active = eid >= 0;
  const int64_t e = active ? eid : 0;
  const int off0 = cl.offsets[c];
  // a table longer than the strip (a caller that ignored
  // sfem_helmholtz_cluster_limits) must not run over the LDS: the surplus
  // entries are dropped (the result is then wrong, the memory stays intact)
  const int Kraw = cl.offsets[c + 1] - off0;
  const int K = Kraw < KMAX ? (Kraw > 0 ? Kraw : 0) : KMAX;
  // lanes on the rim of the (i, j) lattice; inner lanes own interior nodes in
  // the slices 1 .. P-2
  const bool edge = i == 0 || i == P - 1 || j == 0 || j == P - 1;
  const bool inner = active && !edge, rim = active && edge;

  Lay lay;
  lay.init(lds, lane_ok ? el : 0, i, j);
  const int nc = SCALAR ? 1 : prm.ncomp;
  const int64_t ns = NS1 ? 1 : prm.node_stride, ks = prm.comp_stride;

  // this thread's entries of the cluster table (ascending node order)
  uint32_t tab[KPT];
#pragma unroll
  for (int m = 0; m < KPT; ++m) {
    const int q = tid + m * BLOCK;
    tab[m] = q < K ? __builtin_nontemporal_load(&cl.nodes[off0 + q]) : 0u;
  }
  uint32_t enc[P];
  {
    const int32_t* enc0 = prm.enc + e * N;
#pragma unroll
    for (int a = 0; a < P; ++a)
      enc[a] = active ? (uint32_t)__builtin_nontemporal_load(&enc0[a * TPE + t])
                      : 0u;
  }

  double udot = 0.0;
  for (int k = 0; k < nc; ++k) {
    const T* ug = prm.u + (prm.comp + k) * ks;
    T* og = prm.out + (prm.comp + k) * ks;
    // ---- gather: the cluster's table nodes once, into the strip ...
#pragma unroll
    for (int m = 0; m < KPT; ++m) {
      const int q = tid + m * BLOCK;
      if (q < K) strip[q] = ug[(int64_t)(tab[m] & SFEM_IDX_MASK) * ns];
    }
    // ... lattice-interior nodes straight into registers
    T ua[P], acc[P];
#pragma unroll
    for (int a = 0; a < P; ++a) ua[a] = T(0);
    if (inner) {
#pragma unroll
      for (int a = 1; a < P - 1; ++a)
        ua[a] = ug[(int64_t)(enc[a] & SFEM_IDX_MASK) * ns];
    }
    SFEM_CL_BARRIER();
    if (active) {
      ua[0] = strip[enc[0] & SFEM_IDX_MASK];
      ua[P - 1] = strip[enc[P - 1] & SFEM_IDX_MASK];
    }
    if (rim) {
#pragma unroll
      for (int a = 1; a < P - 1; ++a) ua[a] = strip[enc[a] & SFEM_IDX_MASK];
    }
    SFEM_CL_BARRIER();
    // the strip now collects the sums
#pragma unroll
    for (int m = 0; m < KPT; ++m) {
      const int q = tid + m * BLOCK;
      if (q < K) strip[q] = T(0);
    }

    {
      // (after the barriers: set up before them, the geometry constants would
      // have to stay in registers across the whole gather stage)
      ElemGeom<T, P, 3, GM> geom;
      // dm is the first kernel argument: kernarg offset 0
      geom.template init<true>(prm, dm, e, active, i, j, t, kernarg_dmat<T, P>());
      cluster_element_apply<T, P, GM, MASS>(prm, dm, geom, lay, lane_ok, active,
                                            ua, acc);
    }

    // ---- scatter.  Dirichlet rows are zero (select, no branch)
#pragma unroll
    for (int a = 0; a < P; ++a) {
      acc[a] = (enc[a] & SFEM_IDX_DIRICHLET) ? T(0) : acc[a];
      udot += (double)acc[a] * (double)ua[a];
    }
    if (inner) {     // lattice-interior nodes are complete: plain stores
#pragma unroll
      for (int a = 1; a < P - 1; ++a)
        og[(int64_t)(enc[a] & SFEM_IDX_MASK) * ns] = acc[a];
    }
    SFEM_CL_BARRIER();   // the strip is cleared
    if (active) {      // ds_add: LDS only
      unsafeAtomicAdd(&strip[enc[0] & SFEM_IDX_MASK], acc[0]);
      unsafeAtomicAdd(&strip[enc[P - 1] & SFEM_IDX_MASK], acc[P - 1]);
    }
    if (rim) {
#pragma unroll
      for (int a = 1; a < P - 1; ++a)
        unsafeAtomicAdd(&strip[enc[a] & SFEM_IDX_MASK], acc[a]);
    }
    SFEM_CL_BARRIER();
    // nodes complete inside the cluster: plain stores; cluster surface: HBM
    // atomics, in ascending node order (whole 64-byte lines per request)
#pragma unroll
    for (int m = 0; m < KPT; ++m) {
      const int q = tid + m * BLOCK;
      if (q < K) {
        const uint32_t code = tab[m];
        T* dst = og + (int64_t)(code & SFEM_IDX_MASK) * ns;
        const T v = strip[q];
        if (code & SFEM_IDX_SHARED) {
          if (!(code & SFEM_IDX_DIRICHLET)) unsafeAtomicAdd(dst, v);
        } else {
          *dst = v;
        }
      }
    }
    if (k + 1 < nc) SFEM_CL_BARRIER();
  }
  if (prm.dot_out) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) udot += __shfl_down(udot, off, 64);
    if (lane == 0)
      unsafeAtomicAdd(&prm.dot_out[(blockIdx.x * Tile::NW + wave) &
                                   (SFEM_DOT_SLOTS - 1)], udot);
  }
}

template <typename T, int P>
int launch_helmholtz_cluster(const HelmholtzParams<T>& prm,
                             const ClusterParams<T>& cl, hipStream_t stream) {
  using Tile = ClusterTile<T, P>;
  if (cl.num_clusters > 0x7fffffff) {
    set_error("helmholtz: too many clusters (%lld)", (long long)cl.num_clusters);
    return SFEM_EINVAL;
  }
  const DMat<T, P> dm =
      make_dmat<T, P>(prm.dmat_host, prm.weights_host, prm.nodes_host);
  const dim3 grid((unsigned)cl.num_clusters), block(Tile::BLOCK);
  const bool mass = prm.lambda0 != T(0);
#define SFEM_LAUNCH_CL(SC, N1, GMV)                                          \
  do {                                                                        \
    if (mass)                                                                 \
      hipLaunchKernelGGL(                                                     \
          (helmholtz_cluster_kernel<T, P, SC, N1, GMV, true>), grid, block,   \
          0, stream, dm, prm, cl);                                            \
    else                                                                      \
      hipLaunchKernelGGL(                                                     \
          (helmholtz_cluster_kernel<T, P, SC, N1, GMV, false>), grid, block,  \
          0, stream, dm, prm, cl);                                            \
  } while (0)
#define SFEM_LAUNCH_CL_GM(SC, N1)                                             \
  do {                                                                        \
    switch (prm.geo_mode) {                                                   \
      case GEO_POINT: SFEM_LAUNCH_CL(SC, N1, GEO_POINT); break;               \
      case GEO_AFFINE: SFEM_LAUNCH_CL(SC, N1, GEO_AFFINE); break;             \
      default: SFEM_LAUNCH_CL(SC, N1, GEO_MULTILINEAR); break;                \
    }                                                                         \
  } while (0)
  if (prm.ncomp == 1) SFEM_LAUNCH_CL_GM(true, true);
  else if (prm.node_stride == 1) SFEM_LAUNCH_CL_GM(false, true);
  else SFEM_LAUNCH_CL_GM(false, false);
#undef SFEM_LAUNCH_CL_GM
#undef SFEM_LAUNCH_CL
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

// Limits the host side needs to build clusters (sfem_helmholtz_cluster_limits).
template <typename T>
inline int cluster_limits(int P, int* cluster_size, int* max_shared) {
  switch (P) {
#define SFEM_CL_LIMIT(PP)                              \
  case PP:                                             \
    *cluster_size = ClusterTile<T, PP>::CL;            \
    *max_shared = ClusterTile<T, PP>::KMAX;            \
    return SFEM_OK;
    SFEM_CL_LIMIT(4) SFEM_CL_LIMIT(5) SFEM_CL_LIMIT(6) SFEM_CL_LIMIT(7)
    SFEM_CL_LIMIT(8)
#undef SFEM_CL_LIMIT
    default:
      return SFEM_EUNSUPPORTED;
  }
}

template <typename T>
int dispatch_helmholtz_cluster(const HelmholtzParams<T>& prm,
                               const ClusterParams<T>& cl, int P,
                               hipStream_t stream);

#define SFEM_DEFINE_HELMHOLTZ_CLUSTER_DISPATCH(TYPE)                          \
  template <>                                                                 \
  int dispatch_helmholtz_cluster<TYPE>(const HelmholtzParams<TYPE>& prm,      \
                                       const ClusterParams<TYPE>& cl, int P,  \
                                       hipStream_t stream) {                  \
    switch (P) {                                                              \
      case 4: return launch_helmholtz_cluster<TYPE, 4>(prm, cl, stream);      \
      case 5: return launch_helmholtz_cluster<TYPE, 5>(prm, cl, stream);      \
      case 6: return launch_helmholtz_cluster<TYPE, 6>(prm, cl, stream);      \
      case 7: return launch_helmholtz_cluster<TYPE, 7>(prm, cl, stream);      \
      case 8: return launch_helmholtz_cluster<TYPE, 8>(prm, cl, stream);      \
      default:                                                                \
        set_error("helmholtz (clusters): P=%d outside 4..8", P);              \
        return SFEM_EUNSUPPORTED;                                             \
    }                                                                         \
  }

}  // namespace sfem
