// libsfem_hip: C-ABI entry points of the fused Stokes divergence (D) and
// pressure-gradient (D^T) operators plus their setup kernel.
#include "sfem_stokes_facet.h"

namespace sfem {

// kfac[e][a * d + c][q] = w_q detJ_q invjac[e][q][c][a]   (= w * cofactor)
template <typename T>
__global__ void __launch_bounds__(256)
stokes_setup_kernel(const T* __restrict__ invjac, const T* __restrict__ jacdet,
                    const T* __restrict__ weights, T* __restrict__ kfac,
                    int64_t num_elements, int d, int Q) {
  const int64_t total = num_elements * Q;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += stride) {
    const int64_t e = t / Q;
    const int q = (int)(t - e * Q);
    const T* ij = invjac + t * d * d;
    const T wd = weights[q] * jacdet[t];
    T* k = kfac + e * (int64_t)(d * d) * Q;
    for (int a = 0; a < d; ++a)
      for (int c = 0; c < d; ++c)
        k[(int64_t)(a * d + c) * Q + q] = wd * ij[c * d + a];
  }
}

template <typename T>
static int run_stokes(const sfem_stokes_args* a, int mode,
                      hipStream_t stream) {
  StokesParams<T> prm{};
  prm.u = (const T*)a->u; prm.out = (T*)a->out;
  prm.p_in = (const T*)a->p_in; prm.p_out = (T*)a->p_out;
  prm.scale = (const T*)a->scale;
  prm.enc = a->enc; prm.penc = a->penc;
  prm.kfac = (const T*)a->kfac; prm.geo_elem = (const T*)a->geo_elem;
  prm.geo_index = a->geo_index; prm.elem_list = a->elem_list;
  prm.dmat_host = (const T*)a->dmat; prm.weights_host = (const T*)a->weights;
  prm.nodes_host = (const T*)a->nodes; prm.interp_host = (const T*)a->interp;
  prm.num_elements = a->elem_list ? a->num_listed : a->num_elements;
  prm.geo_mode = a->geo_mode;
  prm.node_stride = a->node_stride > 0 ? a->node_stride : a->ndim;
  prm.comp_stride = a->node_stride > 0 ? a->comp_stride : 1;
  if (a->scale_per_node) {
    prm.scale_node_stride = 1;
    prm.scale_comp_stride = 0;
  } else {
    prm.scale_node_stride = prm.node_stride;
    prm.scale_comp_stride = prm.comp_stride;
  }
  prm.shared_order = a->shared_order;
  prm.shared_stride = a->shared_stride;
  prm.dot_out = mode == 0 ? a->dot_out : nullptr;
  if (a->facet_table && (mode == 0 || mode == 1)) {
    StokesFacetParams<T> fprm{};
    fprm.base = prm;
    fprm.tab = a->facet_table;
    fprm.chain_off = a->chain_offsets;
    fprm.chain_elems = a->chain_elems;
    return dispatch_stokes_facet<T>(fprm, a->P, mode, a->num_chains,
                                    a->num_nodes, stream);
  }
  if (a->ndim == 3) return dispatch_stokes<T, 3>(prm, a->P, mode, stream);
  return dispatch_stokes<T, 2>(prm, a->P, mode, stream);
}

static int check_stokes(const char* who, const sfem_stokes_args* a) {
  SFEM_REQUIRE(a, "%s: null args", who);
  SFEM_REQUIRE(a->num_elements >= 0 && (a->ndim == 2 || a->ndim == 3),
               "%s: bad sizes", who);
  // (one Gauss node per direction cannot carry the pressure-space geometry)
  SFEM_REQUIRE(a->P >= 4 && a->P <= SFEM_MAX_P,
               "%s: P=%d outside 4..%d", who, a->P, SFEM_MAX_P);
  SFEM_REQUIRE(a->dtype == SFEM_F32 || a->dtype == SFEM_F64,
               "%s: unknown dtype %d", who, a->dtype);
  if (a->num_elements == 0) return SFEM_OK;
  SFEM_REQUIRE(a->dmat, "%s: null pointer", who);
  if (a->geo_mode == SFEM_GEO_POINT) {
    SFEM_REQUIRE(a->kfac, "%s: per-point geometry needs `kfac`", who);
  } else if (a->geo_mode == SFEM_GEO_AFFINE ||
             a->geo_mode == SFEM_GEO_MULTILINEAR ||
             a->geo_mode == SFEM_GEO_BOX) {
    SFEM_REQUIRE(a->geo_elem && a->weights && a->nodes,
                 "%s: on-the-fly geometry needs geo_elem, weights and nodes",
                 who);
    SFEM_REQUIRE(a->geo_mode != SFEM_GEO_BOX || a->facet_table,
                 "%s: SFEM_GEO_BOX is a facet-table (chain) launch", who);
  } else {
    set_error("%s: unknown geo_mode %d", who, a->geo_mode);
    return SFEM_EINVAL;
  }
  if (a->elem_list)
    SFEM_REQUIRE(a->num_listed >= 0 && a->num_listed <= a->num_elements,
                 "%s: bad element list length", who);
  if (a->facet_table) {
    SFEM_REQUIRE(a->ndim == 3 && stokes_facet_supported_p(a->P),
                 "%s: facet tables are 3D, P = 6..8", who);
    SFEM_REQUIRE(a->node_stride == 1,
                 "%s: facet tables need component-major fields "
                 "(node_stride = 1)", who);
    SFEM_REQUIRE(a->chain_offsets && a->chain_elems && a->num_chains > 0,
                 "%s: facet tables need the chain lists", who);
  }
  return SFEM_OK;
}

}  // namespace sfem

using namespace sfem;

extern "C" {

int sfem_stokes_setup(const void* invjac, const void* jacdet,
                      const void* weights_nd, void* kfac, int64_t num_elements,
                      int ndim, int Q, int dtype, sfem_stream_t stream) {
  SFEM_REQUIRE(num_elements >= 0 && (ndim == 2 || ndim == 3) && Q >= 1,
               "sfem_stokes_setup: bad sizes");
  if (num_elements == 0) return SFEM_OK;
  SFEM_REQUIRE(invjac && jacdet && weights_nd && kfac,
               "sfem_stokes_setup: null pointer");
  const unsigned grid = stream_grid(num_elements * Q, 256);
  if (dtype == SFEM_F64)
    hipLaunchKernelGGL(stokes_setup_kernel<double>, dim3(grid), dim3(256), 0,
                       as_stream(stream), (const double*)invjac,
                       (const double*)jacdet, (const double*)weights_nd,
                       (double*)kfac, num_elements, ndim, Q);
  else if (dtype == SFEM_F32)
    hipLaunchKernelGGL(stokes_setup_kernel<float>, dim3(grid), dim3(256), 0,
                       as_stream(stream), (const float*)invjac,
                       (const float*)jacdet, (const float*)weights_nd,
                       (float*)kfac, num_elements, ndim, Q);
  else {
    set_error("sfem_stokes_setup: unknown dtype %d", dtype);
    return SFEM_EINVAL;
  }
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

int sfem_stokes_convect_local(const sfem_stokes_args* a, sfem_stream_t stream) {
  int rc = check_stokes("sfem_stokes_convect_local", a);
  if (rc) return rc;
  const int64_t work = a->elem_list ? a->num_listed : a->num_elements;
  if (work == 0) return SFEM_OK;
  SFEM_REQUIRE(a->u && a->out, "sfem_stokes_convect_local: null pointer");
  if (a->dtype == SFEM_F64) return run_stokes<double>(a, 2, as_stream(stream));
  return run_stokes<float>(a, 2, as_stream(stream));
}

int sfem_stokes_div(const sfem_stokes_args* a, sfem_stream_t stream) {
  int rc = check_stokes("sfem_stokes_div", a);
  if (rc) return rc;
  SFEM_REQUIRE((a->enc || a->facet_table) && a->interp,
               "sfem_stokes_div: null pointer");
  const int64_t work = a->elem_list ? a->num_listed : a->num_elements;
  if (work == 0) return SFEM_OK;
  SFEM_REQUIRE(a->u && a->p_out, "sfem_stokes_div: null pointer");
  SFEM_REQUIRE(!a->dot_out || a->p_in,
               "sfem_stokes_div: dot_out needs p_in (the vector to dot with)");
  if (a->dtype == SFEM_F64) return run_stokes<double>(a, 0, as_stream(stream));
  return run_stokes<float>(a, 0, as_stream(stream));
}

// zero-fill of the shared-node range of `out` (all components)
static int zero_shared_range(const sfem_stokes_args* a, sfem_stream_t stream) {
  if (a->zero_end <= a->zero_begin) return SFEM_OK;
  const size_t sz = a->dtype == SFEM_F64 ? 8 : 4;
  const int64_t nstr = a->node_stride > 0 ? a->node_stride : a->ndim;
  const int64_t cstr = a->node_stride > 0 ? a->comp_stride : 1;
  return cstr == 1
             ? sfem_zero_strips((char*)a->out + a->zero_begin * nstr * sz,
                                (a->zero_end - a->zero_begin) * nstr, 0, 1,
                                a->dtype, stream)
             : sfem_zero_strips((char*)a->out + a->zero_begin * nstr * sz,
                                (a->zero_end - a->zero_begin) * nstr, cstr,
                                a->ndim, a->dtype, stream);
}

int sfem_stokes_grad_t(const sfem_stokes_args* a, sfem_stream_t stream) {
  int rc = check_stokes("sfem_stokes_grad_t", a);
  if (rc) return rc;
  SFEM_REQUIRE(a->num_elements == 0 ||
                   ((a->enc || a->facet_table) && a->interp),
               "sfem_stokes_grad_t: null pointer");
  SFEM_REQUIRE(a->zero_begin >= 0 && a->zero_end >= a->zero_begin &&
                   a->zero_end <= a->num_nodes,
               "sfem_stokes_grad_t: bad zero range");
  const int64_t work = a->elem_list ? a->num_listed : a->num_elements;
  if (work == 0 && a->zero_end == a->zero_begin) return SFEM_OK;
  SFEM_REQUIRE(a->out && (work == 0 || a->p_in),
               "sfem_stokes_grad_t: null pointer");
  // shared nodes are accumulated with atomics: clear their range first
  rc = zero_shared_range(a, stream);
  if (rc) return rc;
  if (work == 0) return SFEM_OK;
  if (a->dtype == SFEM_F64) return run_stokes<double>(a, 1, as_stream(stream));
  return run_stokes<float>(a, 1, as_stream(stream));
}

int sfem_stokes_e_first(const sfem_stokes_args* a, sfem_stream_t stream) {
  int rc = check_stokes("sfem_stokes_e_first", a);
  if (rc) return rc;
  SFEM_REQUIRE(a->num_elements == 0 || (a->enc && a->interp),
               "sfem_stokes_e_first: null pointer");
  SFEM_REQUIRE(a->zero_begin >= 0 && a->zero_end >= a->zero_begin &&
                   a->zero_end <= a->num_nodes,
               "sfem_stokes_e_first: bad zero range");
  const int64_t work = a->elem_list ? a->num_listed : a->num_elements;
  if (work == 0 && a->zero_end == a->zero_begin) return SFEM_OK;
  SFEM_REQUIRE(a->out && (work == 0 || (a->p_in && a->p_out)),
               "sfem_stokes_e_first: null pointer");
  rc = zero_shared_range(a, stream);
  if (rc) return rc;
  if (work == 0) return SFEM_OK;
  if (a->dtype == SFEM_F64) return run_stokes<double>(a, 3, as_stream(stream));
  return run_stokes<float>(a, 3, as_stream(stream));
}

int sfem_stokes_e_second(const sfem_stokes_args* a, sfem_stream_t stream) {
  int rc = check_stokes("sfem_stokes_e_second", a);
  if (rc) return rc;
  SFEM_REQUIRE(a->enc && a->interp, "sfem_stokes_e_second: null pointer");
  const int64_t work = a->elem_list ? a->num_listed : a->num_elements;
  if (work == 0) return SFEM_OK;
  SFEM_REQUIRE(a->u && a->p_out, "sfem_stokes_e_second: null pointer");
  if (a->dtype == SFEM_F64) return run_stokes<double>(a, 4, as_stream(stream));
  return run_stokes<float>(a, 4, as_stream(stream));
}

}  // extern "C"
