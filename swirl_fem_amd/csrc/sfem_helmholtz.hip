// libsfem_hip: C-ABI entry points of the fused Helmholtz operator plus its
// setup kernels (symmetric geometric factors, encoded indices).
#include <stdlib.h>

#include "sfem_helmholtz_cluster.h"
#include "sfem_helmholtz_mfma.h"
#include "sfem_helmholtz_facet.h"

namespace sfem {

// Per element the factors w detJ (J^-1 J^-T)[ik] (ref-direction indices) and
// w detJ are stored in the paired layout read by helmholtz_kernel:
//   3D: [3][Q][2] = (G00,G01) (G02,G11) (G12,G22), then W [Q]     (7 Q reals)
//   2D: [2][Q][2] = (G00,G01) (G11,W)                             (4 Q reals)
//   1D: [Q][2]    = (G00,W)
template <typename T>
__global__ void __launch_bounds__(256)
helmholtz_setup_kernel(const T* __restrict__ invjac,
                       const T* __restrict__ jacdet,
                       const T* __restrict__ weights, T* __restrict__ geo,
                       int64_t num_elements, int d, int Q) {
  const int ng = d * (d + 1) / 2;
  const int64_t total = num_elements * Q;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += stride) {
    const int64_t e = t / Q;
    const int q = (int)(t - e * Q);
    const T* ij = invjac + t * d * d;
    const T wd = weights[q] * jacdet[t];
    T vals[7];
    int f = 0;
    for (int i = 0; i < d; ++i)
      for (int k = i; k < d; ++k) {
        T acc = T(0);
        for (int j = 0; j < d; ++j) acc += ij[j * d + i] * ij[j * d + k];
        vals[f++] = wd * acc;
      }
    vals[ng] = wd;
    T* g = geo + e * (int64_t)(ng + 1) * Q;
    const int npair = d == 3 ? 3 : (ng + 1) / 2;
    for (int pi = 0; pi < npair; ++pi) {
      g[((int64_t)pi * Q + q) * 2 + 0] = vals[2 * pi];
      g[((int64_t)pi * Q + q) * 2 + 1] = vals[2 * pi + 1];
    }
    if (d == 3) g[(int64_t)6 * Q + q] = wd;
  }
}

__global__ void __launch_bounds__(256)
encode_kernel(const int32_t* __restrict__ elements,
              const uint8_t* __restrict__ dirichlet,
              const int32_t* __restrict__ multiplicity,
              const uint8_t* __restrict__ slot_shared,
              int32_t* __restrict__ enc, int64_t count) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += stride) {
    const int32_t k = elements[i];
    uint32_t v;
    if (k < 0) {
      v = SFEM_IDX_PAD;
    } else {
      v = (uint32_t)k;
      if (dirichlet && dirichlet[k]) v |= SFEM_IDX_DIRICHLET;
      if (slot_shared ? slot_shared[i] != 0 : multiplicity[k] > 1)
        v |= SFEM_IDX_SHARED;
    }
    enc[i] = (int32_t)v;
  }
}

// geo_elem[e] (24 reals) = coefficient vectors of the multilinear map through
// the element's 2^d corner nodes, x(r,s,t) = c + A1 r + A2 s + A3 t + A4 rs +
// A5 st + A6 rt + A7 rst (3D: A1..A7, 21 reals; 2D: x = c + A1 r + A2 s + A3 rs,
// 6 reals), r = axis 0, s = axis 1, t = axis 2 of the element's node lattice.
template <typename T>
__global__ void __launch_bounds__(256)
helmholtz_setup_multilinear_kernel(const T* __restrict__ elem_coords,
                                   T* __restrict__ geo_elem,
                                   int64_t num_elements, int d, int P) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= num_elements) return;
  int n = 1;
  for (int q = 0; q < d; ++q) n *= P;
  const T* x = elem_coords + e * n * d;
  T* g = geo_elem + e * 24;
  for (int q = 0; q < 24; ++q) g[q] = T(0);
  const int L = P - 1;
  if (d == 3) {
    for (int c = 0; c < 3; ++c) {
      T A[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int cr = 0; cr < 2; ++cr)
        for (int cs = 0; cs < 2; ++cs)
          for (int ct = 0; ct < 2; ++ct) {
            const T v = x[((cr * L * P + cs * L) * P + ct * L) * 3 + c];
            const T sr = cr ? 1 : -1, ss = cs ? 1 : -1, st = ct ? 1 : -1;
            A[1] += sr * v; A[2] += ss * v; A[3] += st * v;
            A[4] += sr * ss * v; A[5] += ss * st * v; A[6] += sr * st * v;
            A[7] += sr * ss * st * v;
          }
      for (int q = 1; q < 8; ++q) g[(q - 1) * 3 + c] = A[q] / 8;
    }
  } else if (d == 2) {
    for (int c = 0; c < 2; ++c) {
      T A1 = 0, A2 = 0, A3 = 0;
      for (int cr = 0; cr < 2; ++cr)
        for (int cs = 0; cs < 2; ++cs) {
          const T v = x[(cr * L * P + cs * L) * 2 + c];
          const T sr = cr ? 1 : -1, ss = cs ? 1 : -1;
          A1 += sr * v; A2 += ss * v; A3 += sr * ss * v;
        }
      g[c] = A1 / 4; g[2 + c] = A2 / 4; g[4 + c] = A3 / 4;
    }
  }
}

// geo_elem (24 multilinear coefficients, affine elements: only A1..A3 matter)
// -> G = detJ J^-1 J^-T (upper triangle), detJ, box flag.
template <typename T>
__global__ void __launch_bounds__(256)
helmholtz_setup_affine_kernel(const T* __restrict__ geo_elem,
                              T* __restrict__ geo_const, int64_t num_elements,
                              T box_tol) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= num_elements) return;
  const T* A = geo_elem + e * 24;
  T a0[3], a1[3], a2[3];
  for (int c = 0; c < 3; ++c) { a0[c] = A[c]; a1[c] = A[3 + c]; a2[c] = A[6 + c]; }
  const T c0[3] = {a1[1] * a2[2] - a1[2] * a2[1], a1[2] * a2[0] - a1[0] * a2[2],
                   a1[0] * a2[1] - a1[1] * a2[0]};
  const T c1[3] = {a2[1] * a0[2] - a2[2] * a0[1], a2[2] * a0[0] - a2[0] * a0[2],
                   a2[0] * a0[1] - a2[1] * a0[0]};
  const T c2[3] = {a0[1] * a1[2] - a0[2] * a1[1], a0[2] * a1[0] - a0[0] * a1[2],
                   a0[0] * a1[1] - a0[1] * a1[0]};
  const T det = a0[0] * c0[0] + a0[1] * c0[1] + a0[2] * c0[2];
  const T inv = T(1) / det;
  T* g = geo_const + e * 8;
  g[0] = inv * (c0[0] * c0[0] + c0[1] * c0[1] + c0[2] * c0[2]);
  g[1] = inv * (c0[0] * c1[0] + c0[1] * c1[1] + c0[2] * c1[2]);
  g[2] = inv * (c0[0] * c2[0] + c0[1] * c2[1] + c0[2] * c2[2]);
  g[3] = inv * (c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2]);
  g[4] = inv * (c1[0] * c2[0] + c1[1] * c2[1] + c1[2] * c2[2]);
  g[5] = inv * (c2[0] * c2[0] + c2[1] * c2[1] + c2[2] * c2[2]);
  g[6] = det;
  const T dmax = fmax(fabs(g[0]), fmax(fabs(g[3]), fabs(g[5])));
  const T omax = fmax(fabs(g[1]), fmax(fabs(g[2]), fabs(g[4])));
  g[7] = omax <= box_tol * dmax ? T(1) : T(0);
}

// ---------------------------------------------------------------- builder ---
// One wave per element: fits the 27 affine maps to the element's index row and
// verifies every slot, the Dirichlet flag and the shared flag against them.
// tab[e] is written for every element; ok[e] = 1 iff the element qualifies.
__global__ void __launch_bounds__(64)
facet_table_kernel(const int32_t* __restrict__ elements,
                   const uint8_t* __restrict__ dirichlet,
                   const int32_t* __restrict__ multiplicity,
                   int32_t* __restrict__ tab, uint8_t* __restrict__ ok,
                   int64_t num_elements, int64_t num_nodes, int P) {
  __shared__ int32_t ids[1728];   // P <= 12
  __shared__ int32_t ent[27][4];
  __shared__ int bad;
  const int64_t e = blockIdx.x;
  const int n = P * P * P, lane = threadIdx.x;
  const int32_t* row = elements + e * n;
  if (lane == 0) bad = 0;
  for (int s = lane; s < n; s += 64) ids[s] = row[s];
  __syncthreads();
  auto cls = [P](int a) { return a == 0 ? 0 : (a == P - 1 ? 2 : 1); };
  if (lane < 27) {
    const int c[3] = {lane / 9, (lane / 3) % 3, lane % 3};
    int o[3], inner[3];
    bool exists = true;
    for (int d = 0; d < 3; ++d) {
      inner[d] = c[d] == 1;
      o[d] = c[d] == 0 ? 0 : (c[d] == 1 ? 1 : P - 1);
      if (inner[d] && P < 3) exists = false;
    }
    int32_t code = 0, sd[3] = {0, 0, 0};
    if (exists) {
      const int os = (o[0] * P + o[1]) * P + o[2];
      const int32_t id0 = ids[os];
      const int stride[3] = {P * P, P, 1};
      for (int d = 0; d < 3; ++d)
        if (inner[d] && P >= 4) sd[d] = ids[os + stride[d]] - id0;
      if (id0 < 0 || id0 >= num_nodes) {
        atomicOr(&bad, 1);
      } else {
        uint32_t v = (uint32_t)id0;
        if (dirichlet && dirichlet[id0]) v |= SFEM_IDX_DIRICHLET;
        if (multiplicity[id0] > 1) v |= SFEM_IDX_SHARED;
        code = (int32_t)v;
      }
      // strides must fit the 24-bit multiplies of the kernels
      for (int d = 0; d < 3; ++d)
        if (sd[d] > 0x7FFFFF || sd[d] < -0x7FFFFF) atomicOr(&bad, 1);
    }
    ent[lane][0] = code; ent[lane][1] = sd[0]; ent[lane][2] = sd[1];
    ent[lane][3] = sd[2];
  }
  __syncthreads();
  for (int s = lane; s < n; s += 64) {
    const int a = s / (P * P), i = (s / P) % P, j = s % P;
    const int f = cls(a) * 9 + cls(i) * 3 + cls(j);
    const uint32_t code = (uint32_t)ent[f][0];
    const int32_t pred = (int32_t)(code & SFEM_IDX_MASK) + ent[f][1] * (a - 1) +
                         ent[f][2] * (i - 1) + ent[f][3] * (j - 1);
    const int32_t id = ids[s];
    bool good = id == pred && id >= 0 && id < num_nodes;
    if (good) {
      const bool dir = dirichlet && dirichlet[id];
      const bool sh = multiplicity[id] > 1;
      good = dir == ((code & SFEM_IDX_DIRICHLET) != 0) &&
             sh == ((code & SFEM_IDX_SHARED) != 0);
    }
    if (!good) atomicOr(&bad, 1);
  }
  __syncthreads();
  if (lane < 27) {
    int32_t* dst = tab + e * FACET_ROW + lane * 4;
    for (int q = 0; q < 4; ++q) dst[q] = ent[lane][q];
  }
  if (lane == 0) ok[e] = bad ? 0 : 1;
}

struct HelmholtzCall {
  const void* u; void* out; const int32_t* enc; const void* geo;
  const void* geo_elem; const int32_t* geo_index; const int32_t* elem_list;
  const void* dmat; const void* weights; const void* nodes;
  int64_t num_elements; int ndim, P, ncomp, geo_mode; double l0, l1; bool gs;
  double* dot_out;
  int colored;
  int64_t node_stride, comp_stride;
  const uint16_t* shared_order = nullptr;
  int shared_stride = 0;
  const int32_t* cluster_elems = nullptr;
  const int32_t* cluster_offsets = nullptr;
  const uint32_t* cluster_nodes = nullptr;
  int64_t num_clusters = 0;
  const int32_t* facet_table = nullptr;
  const void* geo_const = nullptr;
  int64_t num_nodes = 0;
  const int32_t* chain_offsets = nullptr;
  const int32_t* chain_elems = nullptr;
  int64_t num_chains = 0;
  int64_t layered_extent = 0;
  int64_t dot_slots = 0;
};

template <typename T>
static int run_helmholtz(const HelmholtzCall& c, hipStream_t stream) {
  HelmholtzParams<T> prm{};
  prm.u = (const T*)c.u; prm.out = (T*)c.out; prm.enc = c.enc;
  prm.geo = (const T*)c.geo; prm.geo_elem = (const T*)c.geo_elem;
  prm.geo_index = c.geo_index; prm.geo_mode = c.geo_mode;
  prm.dmat_host = (const T*)c.dmat; prm.weights_host = (const T*)c.weights;
  prm.nodes_host = (const T*)c.nodes; prm.num_elements = c.num_elements;
  prm.elem_list = c.elem_list; prm.ncomp = c.ncomp;
  prm.node_stride = c.node_stride > 0 ? c.node_stride : c.ncomp;
  prm.comp_stride = c.node_stride > 0 ? c.comp_stride : 1;
  prm.comp = 0; prm.lambda0 = (T)c.l0; prm.lambda1 = (T)c.l1;
  prm.dot_out = c.dot_out;
  prm.colored = c.colored;
  prm.shared_order = c.shared_order;
  prm.shared_stride = c.shared_stride;
  if (c.cluster_elems) {
    ClusterParams<T> cl{c.cluster_elems, c.cluster_offsets, c.cluster_nodes,
                        c.num_clusters};
    return dispatch_helmholtz_cluster<T>(prm, cl, c.P, stream);
  }
  if (c.facet_table) {
    FacetParams<T> fp{};
    fp.u = prm.u; fp.out = prm.out; fp.tab = c.facet_table;
    fp.geo_const = (const T*)c.geo_const; fp.geo_elem = prm.geo_elem;
    fp.geo = prm.geo; fp.geo_index = prm.geo_index;
    fp.elem_list = prm.elem_list; fp.comp_stride = prm.comp_stride;
    fp.ncomp = prm.ncomp; fp.lambda0 = prm.lambda0; fp.lambda1 = prm.lambda1;
    fp.dot_out = prm.dot_out;
    fp.chain_off = c.chain_offsets;
    fp.chain_elems = c.chain_elems;
    fp.layered = c.layered_extent > 0;
    fp.dot_slots = c.dot_slots;
    if (fp.layered)       // positions in the extended vector decide the
      return dispatch_helmholtz_facet<T>(   // addressing width, scalar only
          fp, c.P, c.geo_mode,
          c.chain_offsets ? c.num_chains : c.num_elements, c.layered_extent,
          prm.dmat_host, prm.weights_host, prm.nodes_host, stream);
    if (c.chain_offsets && c.ncomp > 1) {
      // component-major vector field: every component is a scalar field of
      // its own strip, walked by the (scalar) chain kernel
      for (int k = 0; k < c.ncomp; ++k) {
        FacetParams<T> fk = fp;
        fk.u = prm.u + (int64_t)k * prm.comp_stride;
        fk.out = prm.out + (int64_t)k * prm.comp_stride;
        fk.ncomp = 1;
        const int rc = dispatch_helmholtz_facet<T>(
            fk, c.P, c.geo_mode, c.num_chains, c.num_nodes, prm.dmat_host,
            prm.weights_host, prm.nodes_host, stream);
        if (rc != SFEM_OK) return rc;
      }
      return SFEM_OK;
    }
    return dispatch_helmholtz_facet<T>(fp, c.P, c.geo_mode,
                                       c.chain_offsets ? c.num_chains
                                                       : c.num_elements,
                                       c.num_nodes, prm.dmat_host,
                                       prm.weights_host, prm.nodes_host,
                                       stream);
  }
  if constexpr (sizeof(T) == 4) {
    // p = 11 fp32: the contractions on the matrix cores, opt-in (SFEM_MFMA=1):
    // measured 1.25 vs 1.0 ms for the vector-ALU kernel at 48^3 elements
    // (profiles/r02_mfma_notes.md), so not the default
    // (read per launch: the tests switch it inside one process; ~0.1 us)
    const char* v = getenv("SFEM_MFMA");
    const bool mfma_on = v && v[0] == '1';
    if (mfma_on && helmholtz_mfma_applies(prm, c.P, c.ndim, c.gs))
      return launch_helmholtz_mfma_p12(prm, stream);
  }
  if (c.ndim == 3) return dispatch_helmholtz<T, 3>(prm, c.P, c.gs, stream);
  if (c.ndim == 2) return dispatch_helmholtz<T, 2>(prm, c.P, c.gs, stream);
  set_error("helmholtz: ndim=%d (fused kernel supports 2 and 3)", c.ndim);
  return SFEM_EUNSUPPORTED;
}

static int check_geometry(const char* who, int geo_mode, const void* geo,
                          const void* geo_elem, const void* weights,
                          const void* nodes) {
  if (geo_mode == SFEM_GEO_POINT) {
    SFEM_REQUIRE(geo, "%s: per-point geometry needs `geo`", who);
  } else if (geo_mode == SFEM_GEO_AFFINE || geo_mode == SFEM_GEO_MULTILINEAR ||
             geo_mode == SFEM_GEO_BOX) {
    SFEM_REQUIRE(geo_elem && weights && nodes,
                 "%s: on-the-fly geometry needs geo_elem, weights and nodes",
                 who);
  } else {
    set_error("%s: unknown geo_mode %d", who, geo_mode);
    return SFEM_EINVAL;
  }
  return SFEM_OK;
}

// The facet / chain kernels read their matrix argument through the kernarg
// segment at FacetKernarg::MAT_OFF = align(sizeof(params)) -- the offset the
// code-object ABI gives the second by-value argument today.  Nothing in the
// language promises it: this probe compares those bytes with the argument
// itself, so that a compiler or ABI change fails loudly instead of feeding the
// kernels garbage matrices.
template <typename PRM, typename MAT>
__global__ void kernarg_probe_kernel(PRM prm, MAT mat, int32_t* bad) {
  const char* seen = kernarg_bytes() + FacetKernarg<PRM, MAT>::MAT_OFF;
  const char* want = reinterpret_cast<const char*>(&mat);
  for (size_t q = threadIdx.x; q < sizeof(MAT); q += blockDim.x)
    if (seen[q] != want[q]) atomicOr(bad, 1);
  if (threadIdx.x == 0 && prm.ncomp != 12345) atomicOr(bad, 2);
}

template <typename PRM, typename MAT>
static void launch_kernarg_probe(int32_t* bad, hipStream_t st) {
  PRM prm{};
  prm.ncomp = 12345;
  MAT mat;
  unsigned char* bytes = reinterpret_cast<unsigned char*>(&mat);
  for (size_t q = 0; q < sizeof(MAT); ++q)
    bytes[q] = (unsigned char)(37 * q + 11);
  hipLaunchKernelGGL((kernarg_probe_kernel<PRM, MAT>), dim3(1), dim3(64), 0,
                     st, prm, mat, bad);
}

}  // namespace sfem

using namespace sfem;

extern "C" {

int sfem_helmholtz_setup(const void* invjac, const void* jacdet,
                         const void* weights_nd, void* geo,
                         int64_t num_elements, int ndim, int Q, int dtype,
                         sfem_stream_t stream) {
  SFEM_REQUIRE(num_elements >= 0 && ndim >= 1 && ndim <= 3 && Q >= 1,
               "sfem_helmholtz_setup: bad sizes");
  if (num_elements == 0) return SFEM_OK;
  SFEM_REQUIRE(invjac && jacdet && weights_nd && geo,
               "sfem_helmholtz_setup: null pointer");
  const unsigned grid = stream_grid(num_elements * Q, 256);
  if (dtype == SFEM_F64)
    hipLaunchKernelGGL(helmholtz_setup_kernel<double>, dim3(grid), dim3(256), 0,
                       as_stream(stream), (const double*)invjac,
                       (const double*)jacdet, (const double*)weights_nd,
                       (double*)geo, num_elements, ndim, Q);
  else if (dtype == SFEM_F32)
    hipLaunchKernelGGL(helmholtz_setup_kernel<float>, dim3(grid), dim3(256), 0,
                       as_stream(stream), (const float*)invjac,
                       (const float*)jacdet, (const float*)weights_nd,
                       (float*)geo, num_elements, ndim, Q);
  else {
    set_error("sfem_helmholtz_setup: unknown dtype %d", dtype);
    return SFEM_EINVAL;
  }
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_encode_elements(const int32_t* elements, const uint8_t* dirichlet,
                         const int32_t* multiplicity,
                         const uint8_t* slot_shared, int32_t* enc,
                         int64_t count, sfem_stream_t stream) {
  SFEM_REQUIRE(count >= 0, "sfem_encode_elements: negative count");
  if (count == 0) return SFEM_OK;
  SFEM_REQUIRE(elements && (multiplicity || slot_shared) && enc,
               "sfem_encode_elements: null pointer");
  hipLaunchKernelGGL(encode_kernel, dim3(stream_grid(count, 256)), dim3(256), 0,
                     as_stream(stream), elements, dirichlet, multiplicity,
                     slot_shared, enc, count);
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_helmholtz_apply(const sfem_helmholtz_args* a, sfem_stream_t stream) {
  SFEM_REQUIRE(a, "sfem_helmholtz_apply: null args");
  SFEM_REQUIRE(a->num_elements >= 0 && a->num_nodes >= 0 && a->ncomp >= 1 &&
                   a->ncomp <= 8,
               "sfem_helmholtz_apply: bad sizes");
  SFEM_REQUIRE(a->dtype == SFEM_F32 || a->dtype == SFEM_F64,
               "sfem_helmholtz_apply: unknown dtype %d", a->dtype);
  SFEM_REQUIRE(a->num_nodes <= SFEM_IDX_MASK,
               "sfem_helmholtz_apply: more than 2^30-1 nodes per device");
  SFEM_REQUIRE(0 <= a->zero_begin && a->zero_begin <= a->zero_end &&
                   a->zero_end <= a->num_nodes,
               "sfem_helmholtz_apply: bad zero range");
  if (a->num_nodes == 0) return SFEM_OK;
  SFEM_REQUIRE(a->out, "sfem_helmholtz_apply: null out");
  const size_t esz = a->dtype == SFEM_F64 ? 8 : 4;
  if (a->zero_end > a->zero_begin) {
    // shared nodes are accumulated with atomics: clear their range first
    int rc0;
    if (a->node_stride > 0 && a->node_stride != a->ncomp) {
      // component-major layout: one contiguous strip per component
      SFEM_REQUIRE(a->node_stride == 1,
                   "sfem_helmholtz_apply: unsupported vector layout");
      rc0 = sfem_zero_strips((char*)a->out + (size_t)a->zero_begin * esz,
                             a->zero_end - a->zero_begin, a->comp_stride,
                             a->ncomp, a->dtype, stream);
    } else {
      rc0 = sfem_zero_strips(
          (char*)a->out + (size_t)a->zero_begin * a->ncomp * esz,
          (a->zero_end - a->zero_begin) * a->ncomp, 0, 1, a->dtype, stream);
    }
    if (rc0) return rc0;
  }
  if (a->num_elements == 0) return SFEM_OK;
  SFEM_REQUIRE(a->u && (a->enc || a->facet_table) && a->dmat,
               "sfem_helmholtz_apply: null pointer");
  SFEM_REQUIRE(a->geo_mode != SFEM_GEO_BOX || a->facet_table,
               "sfem_helmholtz_apply: SFEM_GEO_BOX needs a facet table");
  int rc = check_geometry("sfem_helmholtz_apply", a->geo_mode, a->geo,
                          a->geo_elem, a->weights, a->nodes);
  if (rc) return rc;
  const int64_t work = a->elem_list ? a->num_listed : a->num_elements;
  SFEM_REQUIRE(work >= 0 && work <= a->num_elements,
               "sfem_helmholtz_apply: bad element list length");
  if (work == 0) return SFEM_OK;
  HelmholtzCall c{a->u, a->out, a->enc, a->geo, a->geo_elem, a->geo_index,
                  a->elem_list, a->dmat, a->weights, a->nodes, work, a->ndim,
                  a->P, a->ncomp, a->geo_mode, a->lambda0, a->lambda1, true,
                  a->dot_out, a->colored, a->node_stride, a->comp_stride,
                  a->shared_order, a->shared_stride};
  SFEM_REQUIRE(!a->shared_order || (a->shared_stride > 0 &&
                                    a->shared_stride <= 0xFFFF),
               "sfem_helmholtz_apply: bad shared_stride");
  if (a->cluster_elems) {
    SFEM_REQUIRE(a->ndim == 3 && a->P >= 4 && a->P <= 8,
                 "sfem_helmholtz_apply: cluster assembly is 3D, P = 4..8");
    SFEM_REQUIRE(a->cluster_offsets && a->cluster_nodes && a->num_clusters > 0,
                 "sfem_helmholtz_apply: incomplete cluster description");
    SFEM_REQUIRE(!a->colored && !a->shared_order,
                 "sfem_helmholtz_apply: clusters exclude colored / "
                 "shared_order");
    c.cluster_elems = a->cluster_elems;
    c.cluster_offsets = a->cluster_offsets;
    c.cluster_nodes = a->cluster_nodes;
    c.num_clusters = a->num_clusters;
  }
  if (a->facet_table) {
    SFEM_REQUIRE(a->ndim == 3 && facet_supported_p(a->P),
                 "sfem_helmholtz_apply: facet tables are 3D, P = 6..12");
    SFEM_REQUIRE(!a->colored && !a->cluster_elems,
                 "sfem_helmholtz_apply: facet tables exclude colored / "
                 "cluster assembly");
    SFEM_REQUIRE(a->ncomp == 1 || a->node_stride == 1,
                 "sfem_helmholtz_apply: facet tables need node_stride = 1");
    SFEM_REQUIRE(a->geo_const || (a->geo_mode != SFEM_GEO_AFFINE &&
                                  a->geo_mode != SFEM_GEO_BOX),
                 "sfem_helmholtz_apply: affine / box facet applies need "
                 "geo_const");
    c.facet_table = a->facet_table;
    c.geo_const = a->geo_const;
    c.num_nodes = a->num_nodes;
    if (a->layered_extent) {
      SFEM_REQUIRE(a->layered_extent >= a->num_nodes &&
                       a->layered_extent <= SFEM_IDX_MASK,
                   "sfem_helmholtz_apply: layered_extent must cover the "
                   "nodal values and stay below 2^30");
      SFEM_REQUIRE(a->ncomp == 1 && a->zero_end == a->zero_begin,
                   "sfem_helmholtz_apply: layered assembly takes scalar "
                   "fields and clears nothing");
      c.layered_extent = a->layered_extent;
      SFEM_REQUIRE(a->dot_slots >= 0, "sfem_helmholtz_apply: negative dot_slots");
      c.dot_slots = a->dot_slots;
    } else {
      SFEM_REQUIRE(a->dot_slots == 0,
                   "sfem_helmholtz_apply: dot_slots goes with layered "
                   "assembly");
    }
    if (a->chain_offsets) {
      SFEM_REQUIRE(a->chain_elems && a->num_chains > 0 &&
                       (a->ncomp == 1 || a->node_stride == 1),
                   "sfem_helmholtz_apply: chains need chain_elems, "
                   "num_chains > 0 and a scalar or component-major field");
      c.chain_offsets = a->chain_offsets;
      c.chain_elems = a->chain_elems;
      c.num_chains = a->num_chains;
    }
  }
  SFEM_REQUIRE(!a->layered_extent || a->facet_table,
               "sfem_helmholtz_apply: layered assembly needs a facet table");
  if (a->dtype == SFEM_F64) return run_helmholtz<double>(c, as_stream(stream));
  return run_helmholtz<float>(c, as_stream(stream));
}

int sfem_kernarg_selftest(int32_t* bad, sfem_stream_t stream) {
  SFEM_REQUIRE(bad, "sfem_kernarg_selftest: null pointer");
  hipStream_t st = as_stream(stream);
  launch_kernarg_probe<FacetParams<double>, SMat<double, 8>>(bad, st);
  launch_kernarg_probe<FacetParams<double>, DMat<double, 8>>(bad, st);
  launch_kernarg_probe<FacetParams<float>, DMat<float, 12>>(bad, st);
  launch_kernarg_probe<FacetParams<float>, SMat<float, 12>>(bad, st);
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_facet_table_build(const int32_t* elements, const uint8_t* dirichlet,
                           const int32_t* multiplicity, int32_t* table,
                           uint8_t* ok, int64_t num_elements,
                           int64_t num_nodes, int P, sfem_stream_t stream) {
  SFEM_REQUIRE(num_elements >= 0 && num_nodes >= 0 && P >= 2 && P <= 12,
               "sfem_facet_table_build: bad sizes");
  if (num_elements == 0) return SFEM_OK;
  SFEM_REQUIRE(elements && multiplicity && table && ok,
               "sfem_facet_table_build: null pointer");
  SFEM_REQUIRE(num_elements <= 0x7fffffff,
               "sfem_facet_table_build: too many elements");
  hipLaunchKernelGGL(facet_table_kernel, dim3((unsigned)num_elements), dim3(64),
                     0, as_stream(stream), elements, dirichlet, multiplicity,
                     table, ok, num_elements, num_nodes, P);
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_helmholtz_setup_affine(const void* geo_elem, void* geo_const,
                                int64_t num_elements, double box_tol,
                                int dtype, sfem_stream_t stream) {
  SFEM_REQUIRE(num_elements >= 0, "sfem_helmholtz_setup_affine: bad sizes");
  if (num_elements == 0) return SFEM_OK;
  SFEM_REQUIRE(geo_elem && geo_const,
               "sfem_helmholtz_setup_affine: null pointer");
  const unsigned grid = (unsigned)((num_elements + 255) / 256);
  if (dtype == SFEM_F64)
    hipLaunchKernelGGL(helmholtz_setup_affine_kernel<double>, dim3(grid),
                       dim3(256), 0, as_stream(stream),
                       (const double*)geo_elem, (double*)geo_const,
                       num_elements, box_tol);
  else if (dtype == SFEM_F32)
    hipLaunchKernelGGL(helmholtz_setup_affine_kernel<float>, dim3(grid),
                       dim3(256), 0, as_stream(stream), (const float*)geo_elem,
                       (float*)geo_const, num_elements, (float)box_tol);
  else {
    set_error("sfem_helmholtz_setup_affine: unknown dtype %d", dtype);
    return SFEM_EINVAL;
  }
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_helmholtz_cluster_limits(int P, int dtype, int* cluster_size,
                                  int* max_shared) {
  SFEM_REQUIRE(cluster_size && max_shared,
               "sfem_helmholtz_cluster_limits: null pointer");
  int rc;
  if (dtype == SFEM_F64) rc = cluster_limits<double>(P, cluster_size, max_shared);
  else if (dtype == SFEM_F32) rc = cluster_limits<float>(P, cluster_size, max_shared);
  else {
    set_error("sfem_helmholtz_cluster_limits: unknown dtype %d", dtype);
    return SFEM_EINVAL;
  }
  if (rc) set_error("sfem_helmholtz_cluster_limits: P=%d outside 4..8", P);
  return rc;
}

int sfem_helmholtz_setup_multilinear(const void* elem_coords, void* geo_elem,
                                     int64_t num_elements, int ndim, int P,
                                     int dtype, sfem_stream_t stream) {
  SFEM_REQUIRE(num_elements >= 0 && (ndim == 2 || ndim == 3) && P >= 2 &&
                   P <= SFEM_MAX_P,
               "sfem_helmholtz_setup_multilinear: bad sizes");
  if (num_elements == 0) return SFEM_OK;
  SFEM_REQUIRE(elem_coords && geo_elem,
               "sfem_helmholtz_setup_multilinear: null pointer");
  const unsigned grid = (unsigned)((num_elements + 255) / 256);
  if (dtype == SFEM_F64)
    hipLaunchKernelGGL(helmholtz_setup_multilinear_kernel<double>, dim3(grid),
                       dim3(256), 0, as_stream(stream),
                       (const double*)elem_coords, (double*)geo_elem,
                       num_elements, ndim, P);
  else if (dtype == SFEM_F32)
    hipLaunchKernelGGL(helmholtz_setup_multilinear_kernel<float>, dim3(grid),
                       dim3(256), 0, as_stream(stream),
                       (const float*)elem_coords, (float*)geo_elem,
                       num_elements, ndim, P);
  else {
    set_error("sfem_helmholtz_setup_multilinear: unknown dtype %d", dtype);
    return SFEM_EINVAL;
  }
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_helmholtz_local(const sfem_helmholtz_args* a, sfem_stream_t stream) {
  SFEM_REQUIRE(a, "sfem_helmholtz_local: null args");
  SFEM_REQUIRE(a->num_elements >= 0 && a->ncomp >= 1 && a->ncomp <= 8,
               "sfem_helmholtz_local: bad sizes");
  SFEM_REQUIRE(a->dtype == SFEM_F32 || a->dtype == SFEM_F64,
               "sfem_helmholtz_local: unknown dtype %d", a->dtype);
  if (a->num_elements == 0) return SFEM_OK;
  SFEM_REQUIRE(a->u && a->out && a->dmat, "sfem_helmholtz_local: null pointer");
  int rc = check_geometry("sfem_helmholtz_local", a->geo_mode, a->geo,
                          a->geo_elem, a->weights, a->nodes);
  if (rc) return rc;
  const int64_t work = a->elem_list ? a->num_listed : a->num_elements;
  SFEM_REQUIRE(work >= 0 && work <= a->num_elements,
               "sfem_helmholtz_local: bad element list length");
  if (work == 0) return SFEM_OK;
  HelmholtzCall c{a->u, a->out, nullptr, a->geo, a->geo_elem, a->geo_index,
                  a->elem_list, a->dmat, a->weights, a->nodes, work, a->ndim,
                  a->P, a->ncomp, a->geo_mode, a->lambda0, a->lambda1, false,
                  nullptr, 0, a->node_stride, a->comp_stride};
  if (a->dtype == SFEM_F64) return run_helmholtz<double>(c, as_stream(stream));
  return run_helmholtz<float>(c, as_stream(stream));
}

}  // extern "C"
