/*
 * sfem.h -- C-ABI of libsfem_hip.so: MI355X (gfx950) kernels for the
 * spectral-element operator-apply + gather-scatter hot path of swirl_fem.
 *
 * The reference (google-research/swirl-fem) is pure Python/JAX and has no FFI
 * layer; its boundary for this path is the Python object API.  Each entry point
 * below names the reference call site(s) it replaces (paths relative to the
 * reference root).  swirl_fem_amd/_ops.py binds them with ctypes; a maintainer
 * of the reference would bind them the same way.
 *
 * Conventions
 *   - all pointers are DEVICE pointers owned by the caller unless marked host;
 *     nothing is allocated or freed inside a call, no call synchronises;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the
 *     legacy default stream);
 *   - `dtype` is SFEM_F32 or SFEM_F64 and applies to every `void*` real array
 *     of the call; index arrays are int32 with -1 (SFEM_SENTINEL) = "missing";
 *   - element-local arrays are (E, n) with n = P^ndim and lexicographic node
 *     order inside an element, axis 0 slowest (reference core/mesh.py:39-46);
 *     quadrature arrays are (E, Q), Q = q^ndim, same ordering;
 *   - vector fields carry their `ncomp` components innermost: (N, ncomp),
 *     (E, n, ncomp), (E, Q, ncomp), gradients (E, Q, ndim, ncomp) with
 *     [j][k] = d u_k / d x_j (reference core/fespace.py:221-225);
 *   - return value: SFEM_OK (0) or a negative error; sfem_last_error() gives a
 *     thread-local message.  Nothing throws across the ABI.
 */
#ifndef SFEM_H_
#define SFEM_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SFEM_ABI_VERSION 7

enum { SFEM_F32 = 0, SFEM_F64 = 1 };
enum {
  SFEM_OK = 0,
  SFEM_EINVAL = -1,       /* bad argument (null pointer, size, dtype, ...)   */
  SFEM_EHIP = -2,         /* a HIP runtime call failed                        */
  SFEM_EUNSUPPORTED = -3  /* shape outside the compiled template range        */
};
#define SFEM_SENTINEL (-1)
#define SFEM_MAX_P 16     /* nodes / quadrature points per direction          */

/* Encoded element index used by the fused operators (sfem_encode_elements):
 * low 30 bits node id (all ones = padding slot: reads 0, writes nothing),
 * two flag bits on top.                                                      */
#define SFEM_IDX_MASK 0x3FFFFFFF
#define SFEM_IDX_PAD 0x3FFFFFFF
#define SFEM_IDX_DIRICHLET 0x80000000u /* row zeroed by the interior mask     */
#define SFEM_IDX_SHARED 0x40000000u /* node belongs to >1 slot: atomic add   */

typedef void* sfem_stream_t;

int sfem_abi_version(void);
const char* sfem_last_error(void);

/* ---------------------------------------------------------------- gather ---
 * out[i] = indices[i] == -1 ? fill : u[indices[i]]          i in [0, count)
 * Replaces gather_scatter.gather (core/gather_scatter.py:121-127) /
 * Mesh.gather (core/mesh.py:155-160).                                        */
int sfem_gather(const void* u, const int32_t* indices, void* out,
                int64_t count, double fill, int dtype, sfem_stream_t stream);

/* out[i, k] = indices[i] == -1 ? 0 : x[indices[i], k]   k in [0, ncomp)
 * Replaces the vmapped gathers Mesh.element_coords (core/mesh.py:170-172) and
 * StokesVelocity.gather (navier_stokes/navier_stokes.py:210-211).            */
int sfem_gather_rows(const void* x, const int32_t* indices, void* out,
                     int64_t count, int ncomp, int dtype, sfem_stream_t stream);

/* --------------------------------------------------------------- scatter ---
 * out[indices[i], k] += u_local[i, k]; entries with index -1 are skipped.
 * `out` (num_nodes, ncomp) is zero-filled by the call first.  Accumulation
 * uses HBM float atomics (order not reproducible).
 * Replaces gather_scatter.scatter (core/gather_scatter.py:130-133) /
 * Mesh.scatter (core/mesh.py:165-168).                                       */
int sfem_scatter_add(const void* u_local, const int32_t* indices, void* out,
                     int64_t count, int64_t num_nodes, int ncomp, int dtype,
                     sfem_stream_t stream);

/* Deterministic direct-stiffness sum through the inverse map (CSR by node):
 * out[v, k] = sum_{s in [offsets[v], offsets[v+1])} u_local[slots[s], k],
 * summed in slot order, so results are bitwise reproducible.                 */
int sfem_scatter_csr(const void* u_local, const int64_t* offsets,
                     const int32_t* slots, void* out, int64_t num_nodes,
                     int ncomp, int dtype, sfem_stream_t stream);

/* -------------------------------------------------------------- exchange ---
 * Unpartitioned QQ^T (periodic images), core/gather_scatter.py:189-261 with
 * axis_name=None:  out = u;  sums[unique[i]] += u[gidx[i]];
 * out[gidx[i]] = sums[unique[i]].   `sums` is caller workspace of num_unique
 * reals (zeroed by the call).  gidx entries of -1 are skipped.               */
int sfem_exchange_local(const void* u, void* out, const int32_t* gidx,
                        const int32_t* unique, int64_t count,
                        int64_t num_nodes, void* sums, int64_t num_unique,
                        int ncomp, int dtype, sfem_stream_t stream);

/* The same QQ^T in place and in ONE launch, from the classes themselves
 * (CSR: class c holds the nodes members[offsets[c] .. offsets[c+1])):
 * every member receives the sum over its class, summed in member order
 * (reproducible, no workspace, no atomics).  `u[k * node_stride +
 * c * comp_stride]` addresses node k, component c: (N, nc) row-major is
 * (nc, 1), component-major strips are (1, N).                               */
int sfem_exchange_classes(void* u, const int32_t* members,
                          const int32_t* offsets, int64_t num_classes,
                          int ncomp, int64_t node_stride, int64_t comp_stride,
                          int dtype, sfem_stream_t stream);

/* Clears `nstrips` strips of `strip_len` reals, `strip_stride` reals apart:
 * the shared-node range of every component ahead of an atomically assembled
 * apply (one launch for short strips, the runtime's fill for long ones).     */
int sfem_zero_strips(void* base, int64_t strip_len, int64_t strip_stride,
                     int nstrips, int dtype, sfem_stream_t stream);

/* Partitioned QQ^T, the pack / unpack halves around the RCCL neighbour
 * exchange that replaces lax.psum (core/gather_scatter.py:247-248):
 *   pack:       buf[i, k] = idx[i] == -1 ? 0 : u[idx[i], k]
 *   unpack_add: u[idx[i], k] += buf[i, k]  (idx unique within one call)      */
int sfem_pack(const void* u, const int32_t* idx, void* buf, int64_t count,
              int ncomp, int dtype, sfem_stream_t stream);
int sfem_unpack_add(const void* buf, const int32_t* idx, void* u,
                    int64_t count, int ncomp, int dtype, sfem_stream_t stream);

/* Single-launch forms for a whole partition interface: `u` is an (N, ncomp)
 * view with element strides (node_stride, comp_stride) -- row-major or
 * component-major -- and `idx` is the concatenation of all per-neighbour lists,
 * so a node may occur several times: the unpack adds atomically.             */
int sfem_pack_strided(const void* u, const int32_t* idx, void* buf,
                      int64_t count, int ncomp, int64_t node_stride,
                      int64_t comp_stride, int dtype, sfem_stream_t stream);
int sfem_unpack_add_atomic(const void* buf, const int32_t* idx, void* u,
                           int64_t count, int ncomp, int64_t node_stride,
                           int64_t comp_stride, int dtype,
                           sfem_stream_t stream);

/* ------------------------------------------------------ geometric factors ---
 * From element node coordinates (E, n, ndim) computes, per quadrature point,
 *   jac[i][j]  = d x_j / d xi_i      (core/fespace.py:338, via I1/G1 factors)
 *   invjac     = jac^-1  (E, Q, ndim, ndim)   (core/fespace.py:345)
 *   jacdet     = det jac (E, Q), signed       (core/fespace.py:346)
 *   quad_coords (E, Q, ndim) or NULL          (core/fespace.py:332-333)
 * interp1 (q, P) = 1D interpolation matrix, grad1 (q, P) = interp1 @ D1, both
 * row-major device arrays of `dtype`.                                        */
int sfem_geom_factors(const void* elem_coords, const void* interp1,
                      const void* grad1, int64_t num_elements, int ndim, int P,
                      int q, void* invjac, void* jacdet, void* quad_coords,
                      int dtype, sfem_stream_t stream);

/* ------------------------------------------------------- basis evaluation ---
 * Sum-factorised evaluation of nodal fields at quadrature points
 * (core/interpolation.py:254-263, :288-292 vmapped by core/fespace.py:178-225):
 *   val  (E, Q, ncomp)        = (I x .. x I) u                  or NULL
 *   grad (E, Q, ndim, ncomp)  = invjac . reference gradient     or NULL
 * invjac == NULL returns the reference-space gradient.  `collocated` != 0
 * skips the interpolation for `val` exactly like interpolation.py:257-258.   */
int sfem_basis_eval(const void* u_local, const void* interp1,
                    const void* grad1, const void* invjac, void* val,
                    void* grad, int64_t num_elements, int ndim, int P, int q,
                    int ncomp, int collocated, int dtype, sfem_stream_t stream);

/* Exact transpose of sfem_basis_eval composed with quadrature: the operator
 * action produced by jax.linear_transpose in FiniteElementSpace.local_covector
 * (core/fespace.py:458-471):
 *   out[e, n, k] = sum_q wdet[e,q] ( val-basis^T c0 + grad-basis^T invjac^T c1 )
 * c0 (E, Q, ncomp) or NULL; c1 (E, Q, ndim, ncomp) or NULL; wdet (E, Q) =
 * jacdet * quadrature weight.                                                */
int sfem_basis_eval_t(const void* c0, const void* c1, const void* interp1,
                      const void* grad1, const void* invjac, const void* wdet,
                      void* out, int64_t num_elements, int ndim, int P, int q,
                      int ncomp, int collocated, int dtype,
                      sfem_stream_t stream);

/* --------------------------------------------------- fused Helmholtz apply ---
 * Collocated (quadrature points == GLL nodes) operator
 *     out = mask * scatter( lambda0 * B_local(g) + lambda1 * A_local(g) ),
 *     g = gather(u)
 * i.e. the mass form u.v, the stiffness form grad u : grad v and their
 * combination H = (beta_k/dt) B + mu A in one pass over the elements:
 * examples/poisson.py:141-154, navier_stokes.py:220-236, :295-307, :431.
 *
 * The per-point symmetric factors G = w detJ (J^-1 J^-T) and W = w detJ come
 * from one of three sources (`geo_mode`):
 *   SFEM_GEO_POINT        stored: ndim(ndim+1)/2 + 1 reals per point in `geo`
 *                         (written by sfem_helmholtz_setup).  Element e uses
 *                         slot geo_index[e] (or e when geo_index is NULL).
 *   SFEM_GEO_MULTILINEAR  computed in registers from the element's multilinear
 *                         map (every refine_premesh mesh): `geo_elem` (E, 24),
 *                         written by sfem_helmholtz_setup_multilinear, plus the
 *                         1D quadrature weights and node values (host arrays).
 *   SFEM_GEO_AFFINE       same data, constant Jacobian: G is a per-element
 *                         constant times the tensor quadrature weight.
 * `elem_list` (device int32, num_listed entries) restricts a launch to some
 * elements, so a mesh mixing the kinds is applied by one call per kind.       */
enum { SFEM_GEO_POINT = 0, SFEM_GEO_AFFINE = 1, SFEM_GEO_MULTILINEAR = 3,
       SFEM_GEO_BOX = 5 /* affine with diagonal J^-1 J^-T (Cartesian boxes):  */
                        /* facet-table applies only.  Helmholtz: needs        */
                        /* `geo_const`; Stokes: `geo_elem` whose Jacobian is  */
                        /* itself diagonal (x_c depends on reference axis c)  */ };

int sfem_helmholtz_setup(const void* invjac, const void* jacdet,
                         const void* weights_nd /* (Q,) */, void* geo,
                         int64_t num_elements, int ndim, int Q, int dtype,
                         sfem_stream_t stream);
/* elem_coords (E, P^ndim, ndim) -> geo_elem (E, 24)                           */
int sfem_helmholtz_setup_multilinear(const void* elem_coords, void* geo_elem,
                                     int64_t num_elements, int ndim, int P,
                                     int dtype, sfem_stream_t stream);

/* enc[i] = node id | flags.  dirichlet (num_nodes,) uint8 or NULL.  The SHARED
 * flag (accumulate instead of store) is set for slots whose node has
 * multiplicity[node] > 1 (multiplicity (num_nodes,) int32 = number of slots
 * referencing the node), or, when slot_shared (count,) uint8 is given, for the
 * slots it marks (coloured assembly: every slot but the first toucher).       */
int sfem_encode_elements(const int32_t* elements, const uint8_t* dirichlet,
                         const int32_t* multiplicity,
                         const uint8_t* slot_shared, int32_t* enc,
                         int64_t count, sfem_stream_t stream);

typedef struct sfem_helmholtz_args {
  const void* u;          /* (N, ncomp); sfem_helmholtz_local: (E, n, ncomp)  */
  void* out;              /* same shape as u                                  */
  const int32_t* enc;     /* (E, n) encoded indices (apply only)              */
  const void* geo;        /* per-point factors or NULL                        */
  const void* geo_elem;   /* (E, 24) or NULL                                  */
  const int32_t* geo_index; /* (E,) slot of element in `geo`, or NULL         */
  const int32_t* elem_list; /* (num_listed,) element ids, or NULL = all       */
  const void* dmat;       /* HOST: (P, P) 1D differentiation matrix           */
  const void* weights;    /* HOST: (P,) 1D quadrature weights, or NULL        */
  const void* nodes;      /* HOST: (P,) 1D node values, or NULL               */
  int64_t num_elements;   /* E                                                */
  int64_t num_listed;     /* length of elem_list                              */
  int64_t num_nodes;      /* N (apply only)                                   */
  int64_t zero_begin;     /* apply: out[zero_begin:zero_end) is cleared first;*/
  int64_t zero_end;       /*   must cover every SHARED / unreferenced node    */
  int32_t ndim;
  int32_t P;
  int32_t ncomp;
  int32_t dtype;          /* of every real array incl. the host ones          */
  int32_t geo_mode;       /* SFEM_GEO_*                                       */
  int32_t colored;        /* apply: != 0 if the listed elements share no node */
                          /*   with each other (one colour class per launch,  */
                          /*   launches stream-ordered): SHARED slots then    */
                          /*   read-modify-write `out` without atomics, the   */
                          /*   sum order is fixed and the result bitwise      */
                          /*   reproducible                                   */
  double lambda0;         /* mass coefficient                                 */
  double lambda1;         /* stiffness coefficient                            */
  int64_t node_stride;    /* layout of u / out in elements: entry (node, k) at*/
  int64_t comp_stride;    /*   node*node_stride + k*comp_stride.  0, 0 = the  */
                          /*   default (N, ncomp) row-major layout; (1, N) =  */
                          /*   component-major storage seen as an (N, ncomp)  */
                          /*   view, which keeps every component contiguous   */
  double* dot_out;        /* apply: NULL, or SFEM_DOT_SLOTS device doubles    */
                          /*   that accumulate partial sums of u . out (the   */
                          /*   p.Ap of CG, cg.py:78, for free in the scatter) */
  const uint16_t* shared_order; /* apply: NULL, or (E, shared_stride): the    */
  int32_t shared_stride;  /*   slots of each element's SHARED, non-Dirichlet  */
                          /*   nodes in ascending node order, padded with     */
                          /*   0xFFFF.  The atomics are then issued in that   */
                          /*   order (values change lanes through LDS): twice */
                          /*   the lanes per 64-byte line, same sums          */
  /* apply, 3D, P = 4..8: cluster assembly (NULL / 0 = off).  One workgroup   */
  /* takes the <= cluster_size elements of a cluster, sums the nodes they     */
  /* share in LDS and touches HBM once per node: plain stores for nodes held  */
  /* by this cluster only, atomics for the cluster surface.                   */
  const int32_t* cluster_elems;   /* (num_clusters, cluster_size) element     */
                                  /*   ids, -1 = empty place                  */
  const int32_t* cluster_offsets; /* (num_clusters + 1,) start of each        */
                                  /*   cluster's table in cluster_nodes       */
  const uint32_t* cluster_nodes;  /* tables, ascending node id per cluster:   */
                                  /*   id | SFEM_IDX_DIRICHLET | SFEM_IDX_    */
                                  /*   SHARED (= also held by an element      */
                                  /*   outside the cluster); at most          */
                                  /*   max_shared entries per cluster (from   */
                                  /*   sfem_helmholtz_cluster_limits; longer  */
                                  /*   tables are truncated by the kernel)    */
  int64_t num_clusters;           /* `enc` must then be in cluster form: a    */
                                  /*   slot whose node is in the table holds  */
                                  /*   SFEM_IDX_SHARED | DIRICHLET bit | its  */
                                  /*   POSITION in the table, other slots     */
                                  /*   id | DIRICHLET bit; elem_list unused   */
  /* apply, 3D, P = 6..12: compact connectivity (NULL = off, `enc` is then    */
  /* required).  (E, 27, 4) int32 from sfem_facet_table_build; every listed   */
  /* element must have qualified there.  `enc`, `shared_order` are not read.  */
  /* Needs node_stride = 1 (scalar or component-major fields).                */
  const int32_t* facet_table;
  const void* geo_const;  /* (E, 8) from sfem_helmholtz_setup_affine: needed  */
                          /*   with facet_table for SFEM_GEO_AFFINE / _BOX    */
  /* facet_table applies of scalar or component-major (node_stride = 1)       */
  /* fields, the latter one launch per component: the elements as chains      */
  /* (NULL / 0 = one workgroup per listed element).  Segment s is the elements */
  /* chain_elems[chain_offsets[s] .. chain_offsets[s+1]); inside a segment    */
  /* the face a = P-1 of every element IS the face a = 0 of the next, node    */
  /* for node: elements[e][(P-1) P^2 + t] == elements[next][t], t < P^2.      */
  /* One wave walks a segment and carries that face in registers (gathered    */
  /* once, summed before it is written: its interior needs no atomic).  The   */
  /* segments must hold every element of the launch exactly once; elem_list   */
  /* is not read.                                                             */
  const int32_t* chain_offsets;   /* (num_chains + 1,)                        */
  const int32_t* chain_elems;     /* (chain_offsets[num_chains],)             */
  int64_t num_chains;
  /* apply with facet_table, scalar fields: layered assembly (0 = off).  `out` */
  /* is then an extended vector of layered_extent reals (see                   */
  /* sfem_cg_update_r_layered) and facet_table is in its LAYERED form          */
  /*   { id0 | flags,  position in `out` | SFEM_IDX_DIRICHLET,  sa,            */
  /*     (si & 0xffff) | (sj << 16) }                                          */
  /* where the position is layer base + id0 of the layer this (element, facet) */
  /* writes: every writer of a facet has a layer of its own (in a chain        */
  /* segment the element that hands a face on does not write it, its successor */
  /* writes the sum).  Slots no element writes must hold zero; the kernel      */
  /* issues plain stores only and clears nothing (zero_begin/zero_end unused). */
  int64_t layered_extent;
  /* with layered_extent and dot_out: 0 = dot_out is SFEM_DOT_SLOTS atomically  */
  /* accumulated slots as above; > 0 = dot_out holds dot_slots doubles, at      */
  /* least one per wave of the launch (workgroups x ceil(P^2 / 64)); wave w     */
  /* STORES its partial sum of u . out at dot_out[w] -- nothing to clear, and   */
  /* the sum over the slots in index order (sfem_cg_scalars_n) is bitwise       */
  /* reproducible                                                               */
  int64_t dot_slots;
} sfem_helmholtz_args;

/* Compact connectivity of refiner-numbered meshes (reference numbering:
 * core/mesh_refiner.py:143-251: the interior nodes of every premesh facet are
 * one contiguous block, read through a cube orientation).  For element e and
 * facet f = 9 cls(a) + 3 cls(i) + cls(j), cls = 0 (index 0) / 1 (interior) /
 * 2 (index P-1), table[e][f] = { id0 | SFEM_IDX_* flags, sa, si, sj } with
 *   elements[e][a, i, j] = id0 + sa (a - 1) + si (i - 1) + sj (j - 1)
 * for every node of the facet (strides of fixed directions are 0).  ok[e] = 1
 * iff all P^3 ids of the element, the Dirichlet flag and the SHARED flag
 * (multiplicity > 1) of every node agree with its table; other elements must
 * be applied through `enc`.  elements (E, P^3), ndim = 3, 2 <= P <= 12.       */
int sfem_facet_table_build(const int32_t* elements, const uint8_t* dirichlet,
                           const int32_t* multiplicity, int32_t* table,
                           uint8_t* ok, int64_t num_elements,
                           int64_t num_nodes, int P, sfem_stream_t stream);

/* Self-test of an assumption of the facet / chain kernels: they read their
 * by-value matrix argument through the kernarg segment at the offset the
 * current code-object ABI places it.  Launches probe kernels that OR a
 * non-zero value into *bad (a zeroed device int32) if the bytes seen there
 * differ from the argument.  The Python layer runs it once per process before
 * the first facet launch and refuses to continue on a mismatch.              */
int sfem_kernarg_selftest(int32_t* bad, sfem_stream_t stream);

/* geo_elem (E, 24) of affine elements -> geo_const (E, 8) =
 * { G00, G01, G02, G11, G12, G22, detJ, box } with G = detJ J^-1 J^-T (no
 * quadrature weight) and box = 1 when |G01|, |G02|, |G12| <= box_tol *
 * max(G00, G11, G22) (the element may then be applied as SFEM_GEO_BOX).      */
int sfem_helmholtz_setup_affine(const void* geo_elem, void* geo_const,
                                int64_t num_elements, double box_tol,
                                int dtype, sfem_stream_t stream);

/* cluster_size and max_shared (table entries per cluster) the cluster kernels
 * of (P, dtype) were compiled for; SFEM_EUNSUPPORTED outside P = 4..8.       */
int sfem_helmholtz_cluster_limits(int P, int dtype, int* cluster_size,
                                  int* max_shared);
#define SFEM_DOT_SLOTS 1024

int sfem_helmholtz_apply(const sfem_helmholtz_args* args, sfem_stream_t stream);

/* Element-local variant (no gather/scatter, no mask): u, out are (E, n, ncomp).
 * StokesVelocity.A_local / B_local (navier_stokes.py:220-236).               */
int sfem_helmholtz_local(const sfem_helmholtz_args* args, sfem_stream_t stream);

/* ------------------------------------------------- fused Stokes operators ---
 * The P_N - P_{N-2} divergence D and pressure gradient D^T of
 * navier_stokes/navier_stokes.py:313-338 as one kernel each (gather, 3 d
 * sum-factorised derivative lines, cofactor geometry, projection onto /
 * interpolation from the P - 2 Gauss pressure nodes, scatter):
 *
 *   sfem_stokes_div:     p_out = pressure.scatter(D_local(velocity.gather(s u)))
 *       D_local(u)_k = sum_q phi_k(x_q) w_q detJ_q div u(x_q)          (:313-320)
 *       `scale` (optional; same layout as u, or one value per node) multiplies u as it is
 *       gathered: E = D Q D^T applies Q = (dt/beta_k) B^-1 (:340-348) for free.
 *   sfem_stokes_grad_t:  out = mask * velocity.scatter(Dt_local(pressure.gather(p)))
 *       Dt_local(p)_{i,c} = sum_q w_q detJ_q p(x_q) d phi_i/d x_c      (:322-338)
 *       DIRICHLET / SHARED bits of `enc` as in sfem_helmholtz_apply; the shared
 *       range [zero_begin, zero_end) of `out` is cleared by the call.  `scale`
 *       (optional, as for div) multiplies every element's contribution before it
 *       is assembled: for a factor that is equal on all copies of a node this
 *       is scale * (D^T p) after QQ^T, and E = D QQ^T (Q . D^T) spares D the
 *       gather of Q.
 *
 * Both integrate on the velocity GLL points (the `quadrature` of :279-282).
 * Geometry: SFEM_GEO_AFFINE / SFEM_GEO_MULTILINEAR evaluate the cofactors of
 * the Jacobian from `geo_elem` (sfem_helmholtz_setup_multilinear) in registers;
 * SFEM_GEO_POINT reads `kfac` (slots, ndim*ndim, Q) from sfem_stokes_setup:
 *   kfac[e][a*ndim + c][q] = w_q detJ_q invjac[e][q][c][a].
 * `interp` (HOST, (P, P-2) row-major) = pressure basis phi_k at the GLL points;
 * `penc` (E, (P-2)^ndim) = pressure node ids (negative = skip) or NULL for
 * e * (P-2)^ndim + k.  P = 4..12, ndim = 2, 3.                                */
typedef struct sfem_stokes_args {
  const void* u;          /* div: (N, ndim) velocity                           */
  void* out;              /* grad_t: (N, ndim) result                          */
  const void* p_in;       /* grad_t: (Np,) pressure                            */
  void* p_out;            /* div: (Np,) result                                 */
  const void* scale;      /* div: optional per-node factor, or NULL: (N, ndim)
                             in the layout of u, or (N,) if scale_per_node   */
  const int32_t* enc;     /* (E, n) encoded velocity indices                   */
  const int32_t* penc;    /* (E, np) pressure node ids or NULL                 */
  const void* kfac;       /* per-point weighted cofactors or NULL              */
  const void* geo_elem;   /* (E, 24) or NULL                                   */
  const int32_t* geo_index; /* (E,) slot of element e in kfac, or NULL = e     */
  const int32_t* elem_list; /* element ids of this launch, or NULL = all       */
  const void* dmat;       /* HOST (P, P)                                       */
  const void* weights;    /* HOST (P,)                                         */
  const void* nodes;      /* HOST (P,)                                         */
  const void* interp;     /* HOST (P, P-2)                                     */
  int64_t num_elements;
  int64_t num_listed;
  int64_t num_nodes;
  int64_t zero_begin, zero_end;   /* grad_t only                               */
  int32_t ndim, P, dtype, geo_mode;
  int64_t node_stride, comp_stride;  /* layout of u / out / scale (0 = (N, ndim)
                                        row-major)                             */
  int32_t scale_per_node; /* scale is one (N,) factor shared by the components */
  const uint16_t* shared_order; /* grad_t / e_first: as in sfem_helmholtz_args, */
  int32_t shared_stride;        /*   or NULL                                    */
  double* dot_out;        /* div: NULL, or SFEM_DOT_SLOTS device doubles that    */
                          /*   accumulate partial sums of p_in . p_out (the p.Ap */
                          /*   of the pressure CG when p_in is its direction)    */
  /* div / grad_t, 3D, P = 6..8, node_stride = 1 (component-major fields):    */
  /* compact connectivity of the velocity mesh and its chains, exactly as in  */
  /* sfem_helmholtz_args (NULL = off; `enc` / `shared_order` are then not     */
  /* read).  Both are needed: a launch without chains passes segments of one  */
  /* element.  The segments must hold every element of the launch once.       */
  /* geo_mode SFEM_GEO_BOX (these launches only): elements whose `geo_elem`   */
  /* has a diagonal constant Jacobian -- component c of div / grad_t then     */
  /* takes the one 1D derivative along axis c.                                */
  const int32_t* facet_table;     /* (E, 27, 4) from sfem_facet_table_build   */
  const int32_t* chain_offsets;   /* (num_chains + 1,)                        */
  const int32_t* chain_elems;
  int64_t num_chains;
} sfem_stokes_args;

int sfem_stokes_setup(const void* invjac, const void* jacdet,
                      const void* weights_nd, void* kfac, int64_t num_elements,
                      int ndim, int Q, int dtype, sfem_stream_t stream);
int sfem_stokes_div(const sfem_stokes_args* args, sfem_stream_t stream);
/* Convection integrand on a collocated grid of P points per direction (the
 * over-integration grid of StokesVelocity.C_local, navier_stokes.py:238-245,
 * after interpolation to it):  u, out (E, P^ndim, ndim) element-local,
 *   out[e,q,c] = w_q detJ_q sum_j u_j(x_q) d u_c/d x_j(x_q).
 * `dmat`, `weights`, `nodes` are those of that grid; enc / penc / interp /
 * p_in / p_out / scale are unused.                                           */
int sfem_stokes_convect_local(const sfem_stokes_args* args,
                              sfem_stream_t stream);
int sfem_stokes_grad_t(const sfem_stokes_args* args, sfem_stream_t stream);
/* E = D Q D^T (StokesSEM.E, navier_stokes.py:340-348) in two halves for a
 * diagonal Q = `scale`.  A velocity node held by one element only, and not part
 * of the periodic / partition exchange (`enc` must flag every other node
 * SHARED), is complete after that element's D^T, so the first half keeps it
 * in registers: scaled by Q it goes straight into the element's D.
 *   sfem_stokes_e_first : p_in -> `out` (SHARED nodes only: zero-filled range +
 *                         atomics; other entries are not written) and
 *                         p_out = D (Q . complete part)
 *   [caller: QQ^T exchange of `out` on the shared nodes]
 *   sfem_stokes_e_second: p_out += D (Q . u restricted to the SHARED nodes)
 * Same argument block as sfem_stokes_div / sfem_stokes_grad_t; `scale`
 * applies in both halves.  At P = 8 in 3D 216 of the 512 nodes of an element
 * never reach memory.                                                        */
int sfem_stokes_e_first(const sfem_stokes_args* args, sfem_stream_t stream);
int sfem_stokes_e_second(const sfem_stokes_args* args, sfem_stream_t stream);

/* ------------------------------------------------------------ CG kernels ---
 * Preconditioned CG of linalg/cg.py:30-97 with device-resident scalars: no
 * host synchronisation inside an iteration (the reference keeps its loop on
 * device with lax.while_loop, cg.py:94-95).  `scalars` is a device array of
 * SFEM_CG_NSCALARS (= 80) doubles -- ALL of them are read and written by the
 * update kernels and the scalar phases (ABI <= 2 callers allocated 16: they
 * must grow the array):
 *   [0] gamma = r.M r   [1] p.Ap   [2] gamma_new   [3] alpha   [4] beta
 *   [5] b.b   [6] atol2 = max(tol^2 b.b, atol^2)   [7] done (0/1)
 *   [8] iterations   [9] an iteration is open (phases 5 / 6)
 *   [11] r.Mr - r.r of the iteration, [12] the mean c (sfem_cg_update_xp_mean)
 *   [10] status, SFEM_CG_STATUS_*: why `done` was raised.  Beyond the
 *        reference's stop rule (cg.py:68-73, which reads a negative or NaN
 *        r.Mr as "converged" and divides by any p.Ap) the solve stops with
 *        BAD_GAMMA when r.Mr is negative or not finite and with BAD_PAP when
 *        p.Ap is zero or not finite (checked before that iteration's updates,
 *        so x is the last good iterate); a negative p.Ap is divided by, as in
 *        the reference.
 * Once `done` is set every kernel below is a no-op, so the host may run ahead
 * and poll [7] asynchronously; unless one of the two breakdowns occurs, the
 * iterate and the iteration count are exactly those of a loop that tests the
 * condition of cg.py:68-73 every iteration.
 *
 * sfem_dot:            *result  = sum a*b          (clears result first)
 * sfem_dot_accumulate: *result += sum a*b
 * sfem_cg_scalars:     phase 2 = init (after b.b and gamma0 are in place),
 *                      phase 0 = after p.Ap, phase 1 = end of iteration,
 *                      phase 3 = p.Ap <- sum of the SFEM_DOT_SLOTS partial
 *                      sums written by sfem_helmholtz_apply (`partials`);
 *                      phase 4 = phase 3 then phase 0 in one launch;
 *                      phase 5 = [close the previous iteration as phase 1
 *                      would, if one is open: scalars[9]] then phase 4, and
 *                      `partials` is cleared at once: ONE scalar launch per
 *                      iteration, placed between the apply and the updates;
 *                      the done flag is then raised one (harmless) apply
 *                      late; phase 6 = close the open iteration now (before
 *                      the host reads [0], [7] or [8]);
 *                      phase 7 = fold the striped r.r (below) into [2] now,
 *                      before the caller corrects or all-reduces it;
 *                      phases 1 and 2 clear `partials` when it is given
 *                      gamma_new is scalars[2] PLUS the SFEM_CG_RR_SLOTS partial
 *                      sums scalars[16..80): sfem_cg_update_r with fuse_rr = 2
 *                      spreads its per-workgroup r.r sums over them (one
 *                      address would serialise 32 k atomics); every consumer
 *                      (update_p, update_xp, the closing phases) adds them up,
 *                      the closing phases clear them
 * sfem_cg_update_xr:   x += alpha p; r -= alpha Ap;  (cg.py:80-81)
 *                      fuse_rr != 0 also accumulates gamma_new += r.r (M = I)
 * sfem_cg_update_p:    p = z + beta p                (cg.py:84-85)
 * sfem_cg_update_r / sfem_cg_update_xp: the same two updates regrouped into
 *                      8 instead of 9 vector passes (bitwise the same result):
 *                      r -= alpha Ap (+ gamma_new += r.r), then, once beta is
 *                      known,  x += alpha p;  p = z + beta p
 * sfem_cg_update_r_mean / sfem_cg_update_xp_mean: the same pair for the
 *                      preconditioner  M r = r - (w . r / total) 1  (the mean
 *                      projection of the pressure solve, navier_stokes.py:
 *                      73-78, w = B 1, total = 1 . B 1) without storing z = M r:
 *                      update_r_mean also sums r.r (striped slots), 1.r and w.r
 *                      (`sums`: SFEM_CG_MEAN_SUMS device doubles, zero before
 *                      the first iteration, owned by the solve); update_xp_mean
 *                      forms c = w.r / total, gamma_new = r.r - c 1.r, beta, and
 *                      p = (r - c) + beta p.  9 vector passes per iteration
 *                      instead of 12.  The closing phases of sfem_cg_scalars
 *                      pick up gamma_new - r.r from scalars[11].               */
#define SFEM_CG_NSCALARS_NAMED 16 /* [0..16): the named scalars above      */
#define SFEM_CG_RR_SLOTS 64       /* [16..80): partial sums of gamma_new     */
#define SFEM_CG_NSCALARS (SFEM_CG_NSCALARS_NAMED + SFEM_CG_RR_SLOTS)
#define SFEM_CG_MEAN_SUMS (4 * SFEM_CG_RR_SLOTS) /* 2 parities x (1.r, w.r) */
#define SFEM_CG_STATUS_RUNNING 0.0
#define SFEM_CG_STATUS_CONVERGED 1.0
#define SFEM_CG_STATUS_MAXITER 2.0
#define SFEM_CG_STATUS_BAD_PAP 3.0
#define SFEM_CG_STATUS_BAD_GAMMA 4.0
int sfem_dot(const void* a, const void* b, int64_t count, double* result,
             int dtype, sfem_stream_t stream);
int sfem_dot_accumulate(const void* a, const void* b, int64_t count,
                        double* result, int dtype, sfem_stream_t stream);
/* *result += scale * sum_i w[i] * sum_k a[idx[i], k] b[idx[i], k]: the interface
 * correction that turns the plain local dot of two partition-consistent
 * vectors into this rank's share of the global inner product (scale = -1,
 * w = 1 - 1/holders).  a, b: (N,) or (N, ncomp) views with the given strides. */
int sfem_dot_indexed(const void* a, const void* b, const int64_t* idx,
                     const double* w, int64_t count, int ncomp,
                     int64_t node_stride, int64_t comp_stride, double scale,
                     double* result, int dtype, sfem_stream_t stream);
/* out = w - (b . w / total) 1: the nullspace projection of the pressure
 * preconditioner (navier_stokes.py:73-78 with b = B 1, total = 1 . B 1).
 * Two launches; `partials`: SFEM_DOT_SLOTS device doubles of workspace (need
 * not be cleared).  `out` may alias `w`.  `dot_result`: NULL, or one device
 * double that accumulates w . out (the r . z of a CG preconditioned by this
 * projection: its separate dot pass disappears).                             */
int sfem_subtract_weighted_mean(const void* w, const void* b, double total,
                                void* out, double* partials, int64_t count,
                                double* dot_result, int dtype,
                                sfem_stream_t stream);
int sfem_cg_scalars(double* scalars, int phase, double maxiter, double tol,
                    double atol, double* partials, sfem_stream_t stream);
int sfem_cg_update_xr(void* x, void* r, const void* p, const void* ap,
                      int64_t count, double* scalars, int fuse_rr, int dtype,
                      sfem_stream_t stream);
int sfem_cg_update_p(void* p, const void* z, int64_t count, double* scalars,
                     int dtype, sfem_stream_t stream);
int sfem_cg_update_r(void* r, const void* ap, int64_t count, double* scalars,
                     int fuse_rr, int dtype, sfem_stream_t stream);
int sfem_cg_update_xp(void* x, void* p, const void* z, int64_t count,
                      double* scalars, int dtype, sfem_stream_t stream);
/* Reproducible inner products.  sfem_cg_scalars_n = sfem_cg_scalars reading
 * `num_partials` STORED partial sums of p.Ap (sfem_helmholtz_args.dot_slots;
 * phases 3, 4, 5: summed in a fixed order, not cleared), plus phase 8:
 * gamma_new (scalars[2]) <- the fixed-order sum of `num_partials` stored
 * per-workgroup sums of r.r.  sfem_cg_update_r_layered with rr_partials != NULL
 * (fuse_rr != 0) stores its workgroups' sums there instead of accumulating them
 * atomically; `*num_rr` receives the number of workgroups (<= rr_capacity, the
 * launch is sized to fit).  A CG iteration built from these and the layered
 * apply is bitwise reproducible from run to run.                              */
/* `partials` must have room for num_partials + SFEM_FOLD_GROUPS doubles: long
 * sums are taken in two fixed-order stages through the scratch behind them.   */
#define SFEM_FOLD_GROUPS 256
int sfem_cg_scalars_n(double* scalars, int phase, double maxiter, double tol,
                      double atol, double* partials, int64_t num_partials,
                      sfem_stream_t stream);
int sfem_cg_update_r_layered_det(void* r, const void* ap_ext, int64_t count,
                                 const int64_t* layer_len,
                                 const int64_t* layer_off, int num_layers,
                                 const uint8_t* layer_masks,
                                 const int64_t* mask_off,
                                 double* scalars, double* rr_partials,
                                 int64_t rr_capacity, int64_t* num_rr,
                                 int dtype, sfem_stream_t stream);
/* Lazy solution update: sfem_cg_update_xp with x touched every m-th iteration
 * only.  `pring`: m direction vectors, slot s at pring + s * ring_stride
 * (elements; ring_stride a multiple of 16 bytes); p_k lives in slot k mod m,
 * k = scalars[8], and  p_{k+1} = z + beta p_k  goes to the next slot.  When
 * (k + 1) mod m = 0 the pending terms are added,
 *   x = (((x + alpha_{k-m+1} p_{k-m+1}) + ...) + alpha_k p_k),
 * in the order and with the roundings of the iteration-by-iteration update
 * (cg.py:80): bitwise the same x, (4 m + 1) / m vector passes per iteration
 * instead of 5.  `lazy`: 1 + SFEM_CG_LAZY_MAX device doubles owned by the
 * solve, zero before the first iteration ([0] = iterations already in x,
 * [1 + s] = alpha of slot s).  sfem_cg_flush_x adds what is still pending
 * after scalars[8] iterations (call it before reading x; the open iteration
 * of the one-scalar-launch scheme must be closed first: sfem_cg_scalars
 * phase 6).                                                                  */
#define SFEM_CG_LAZY_MAX 8
int sfem_cg_update_xp_lazy(void* x, void* pring, int64_t ring_stride,
                           const void* z, int64_t count, double* scalars,
                           double* lazy, int m, int dtype,
                           sfem_stream_t stream);
int sfem_cg_flush_x(void* x, const void* pring, int64_t ring_stride,
                    int64_t count, const double* scalars, double* lazy, int m,
                    int dtype, sfem_stream_t stream);
/* Layered assembly (sfem_helmholtz_args.layered_extent): the operator result
 * is an EXTENDED vector  [ count nodal values | layer 1 | layer 2 | ... ];
 * layer k (k = 0 .. num_layers-1 here) holds further contributions to the
 * nodes [0, layer_len[k]) and starts at element layer_off[k] of the vector
 * (HOST arrays; lengths and offsets multiples of 16 bytes, lengths not
 * increasing, at most SFEM_MAX_LAYERS).  The assembled value of node i is
 *   ext[i] + sum_{k : i < layer_len[k]} ext[layer_off[k] + i],
 * added in layer order -- the direct-stiffness sum of core/gather_scatter.py:
 * 130-133 in a fixed order: no atomics, no cleared range, bitwise reproducible.
 * sfem_cg_update_r_layered = sfem_cg_update_r (cg.py:81) reading Ap that way,
 * where it streams Ap anyway; sfem_fold_layers writes the assembled values
 * back into ext[0 .. count) for every other consumer.                         */
#define SFEM_MAX_LAYERS 15
/* layer_masks / mask_off (optional, NULL = read every layer in full): one byte
 * per SFEM_LAYER_CHUNK nodes of a layer, layer k's bytes starting at
 * layer_masks[mask_off[k]] (device array, host offsets; mask_off[k] < 0 = no
 * mask for that layer); 0 = no element writes into the chunk (it holds zeros
 * and is not read).                                                          */
#define SFEM_LAYER_CHUNK 512
int sfem_cg_update_r_layered(void* r, const void* ap_ext, int64_t count,
                             const int64_t* layer_len,
                             const int64_t* layer_off, int num_layers,
                             const uint8_t* layer_masks,
                             const int64_t* mask_off, double* scalars,
                             int fuse_rr, int dtype, sfem_stream_t stream);
int sfem_fold_layers(void* out_ext, int64_t count, const int64_t* layer_len,
                     const int64_t* layer_off, int num_layers, int dtype,
                     sfem_stream_t stream);
/* The same at `num_idx` DISTINCT nodes idx[] (negative entries skipped), and
 * the folded layer slots are cleared: a partitioned operator makes its
 * interface values whole before it packs them for the neighbours
 * (core/gather_scatter.py:247-248), later consumers of the layers find zeros. */
int sfem_fold_layers_at(void* out_ext, const int64_t* idx, int64_t num_idx,
                        int64_t count, const int64_t* layer_len,
                        const int64_t* layer_off, int num_layers, int dtype,
                        sfem_stream_t stream);
int sfem_cg_update_r_mean(void* r, const void* ap, const void* w,
                          int64_t count, double* scalars, double* sums,
                          int dtype, sfem_stream_t stream);
int sfem_cg_update_xp_mean(void* x, void* p, const void* r, int64_t count,
                           double* scalars, double* sums, double total,
                           int dtype, sfem_stream_t stream);
/* Element-wise fast diagonalisation solve: for every element e
 *   z_e = (S_0 (x) .. (x) S_{d-1}) [ w_e .* (S_0 (x) .. (x) S_{d-1})^T r_e ]
 * r_e / z_e: the Pp^ndim values of element e at r[pel[e][.]] (pel NULL: element
 * e owns the contiguous range [e Pp^d, (e + 1) Pp^d)), lexicographic, axis 0
 * slowest; S (num_cases, Pp, Pp) eigenvector matrices (rows = nodes, columns =
 * modes), cases (ndim, E): which one element e takes along its axis a; w
 * (E, Pp^d) the (pseudo-)inverted eigenvalues.  The local part of the opt-in
 * pressure preconditioner (swirl_fem_amd/navier_stokes/
 * pressure_preconditioner.py) for the reference's hook
 * navier_stokes/navier_stokes.py:354, :449-452.  Pp <= 10.                   */
int sfem_fdm_solve(const void* r, void* z, const int64_t* pel, const void* S,
                   const int32_t* cases, const void* inv_eigenvalues,
                   int64_t num_elements, int ndim, int Pp, int dtype,
                   sfem_stream_t stream);
/* The same solve that also hands back, per element, elem_sum[e] = sum of r
 * over the element (the restriction R_0 r of the coarse level) and, when
 * `weighted_sum` is given, weighted_sum[e] = sum_i weights[node] z[node] (the
 * element's share of the mean projection that closes the preconditioner):
 * two vector passes less per application.                                    */
int sfem_fdm_solve_sums(const void* r, void* z, const int64_t* pel,
                        const void* S, const int32_t* cases,
                        const void* inv_eigenvalues, const void* weights,
                        void* elem_sum, void* weighted_sum,
                        int64_t num_elements, int ndim, int Pp, int dtype,
                        sfem_stream_t stream);
/* z[e n + i] += yc[e] - shift[e / elems_per_member] (element e owns the nodes
 * [e n, (e + 1) n)): coarse correction and mean removal in one pass; `shift`
 * is a device array (one value per ensemble member, 1 without ensembles).    */
int sfem_add_element_constants(void* z, const void* yc, const void* shift,
                               int64_t num_elements, int n,
                               int64_t elems_per_member, int dtype,
                               sfem_stream_t stream);
/* x = q(D^-1 A) D^-1 b: `steps` steps of the Chebyshev iteration for the
 * interval [lmin, lmax] of the Jacobi-scaled sparse matrix A, a FIXED
 * polynomial (no inner products; linear, symmetric positive definite whenever
 * lmax bounds the spectrum): the coarse solve of the same preconditioner.
 * A: n rows of `width` entries, column-major (entry k of row i at
 * [k * n + i]; unused entries: value 0, any valid column); dinv = 1 / diag(A);
 * work: 3 n reals.  One launch per step.                                      */
int sfem_ell_chebyshev(const int32_t* cols, const void* vals, const void* dinv,
                       const void* b, void* x, void* work, int64_t n,
                       int width, int steps, double lmin, double lmax,
                       int dtype, sfem_stream_t stream);
/* --------------------------------------------- CG for an ensemble (vmap) ---
 * B independent CG recurrences in one set of launches: what the reference
 * gets from jax.vmap of its solver step over an ensemble (niles/train.py:232,
 * :262-264).  Member m owns the contiguous range [m len, (m + 1) len) of every
 * vector (B disjoint copies of the mesh seen as one mesh, so the operator
 * kernels serve all members in one launch unchanged) and the
 * SFEM_ENS_NSCALARS doubles at scalars + m SFEM_ENS_NSCALARS, same slots as
 * the single solve above ([0] gamma [1] p.Ap [3] alpha [4] beta [5] b.b
 * [6] atol2 [7] done [8] iterations [9] active in this iteration [10] status).
 * Each member follows linalg/cg.py:60-97 with its own step lengths and stop
 * test; a member that has stopped is a no-op from then on.  Inner products are
 * stored partial sums, `partials`: members x 2 x SFEM_ENS_GROUPS doubles,
 * added in index order by the consumers (no atomics, nothing to clear).
 * One iteration:
 *     Ap = A p;  sfem_ens_dot(p, Ap, which = 0);  sfem_ens_update_r;
 *     z = M r;   sfem_ens_dot(r, z, which = 1);   sfem_ens_close;
 *     sfem_ens_update_xp
 * and before the first: sfem_ens_dot(b, b, 0), sfem_ens_dot(r, z, 1),
 * sfem_ens_init.                                                             */
#define SFEM_ENS_NSCALARS 16
#define SFEM_ENS_GROUPS 32
#define SFEM_ENS_MAX_MEMBERS 4096
int sfem_ens_dot(const void* a, const void* b, int64_t len, int members,
                 double* partials, int which, int dtype, sfem_stream_t stream);
int sfem_ens_init(double* scalars, const double* partials, int members,
                  double maxiter, double tol, double atol,
                  sfem_stream_t stream);
/* r -= alpha Ap, alpha = gamma / p.Ap from the partial sums                  */
int sfem_ens_update_r(void* r, const void* ap, int64_t len, int members,
                      const double* scalars, const double* partials,
                      int dtype, sfem_stream_t stream);
/* closes the iteration of every member: alpha, beta, gamma <- gamma_new,
 * counter, stop test (cg.py:68-73 + the breakdown guards of the single solve) */
int sfem_ens_close(double* scalars, const double* partials, int members,
                   double maxiter, sfem_stream_t stream);
/* x += alpha p;  p = z + beta p  for the members active in this iteration    */
int sfem_ens_update_xp(void* x, void* p, const void* z, int64_t len,
                       int members, const double* scalars, int dtype,
                       sfem_stream_t stream);
/* The same iteration for M r = r - (w . r / total) 1 per member (w: ONE
 * member's weights, len values) without storing z = M r, as
 * sfem_cg_update_r_mean / sfem_cg_update_xp_mean do for the single solve:
 *     Ap = A p;  sfem_ens_dot(p, Ap, 0);  sfem_ens_update_r_mean;
 *     sfem_ens_close_mean;  sfem_ens_update_xp_mean
 * `sums`: members x 3 x SFEM_ENS_GROUPS doubles (r.r, 1.r, w.r); the mean c of
 * the iteration is left in scalars[12].                                      */
int sfem_ens_update_r_mean(void* r, const void* ap, const void* w, int64_t len,
                           int members, const double* scalars,
                           const double* partials, double* sums, int dtype,
                           sfem_stream_t stream);
int sfem_ens_close_mean(double* scalars, const double* partials,
                        const double* sums, double total, int members,
                        double maxiter, sfem_stream_t stream);
int sfem_ens_update_xp_mean(void* x, void* p, const void* r, int64_t len,
                            int members, const double* scalars, int dtype,
                            sfem_stream_t stream);
/* out_m = w_m - (b . w_m / total) 1 for every member m (b: one member's
 * weights, len values; partials: members x SFEM_ENS_GROUPS doubles): the mean
 * projection of the pressure solve (navier_stokes.py:73-78) per member.      */
int sfem_ens_subtract_weighted_mean(const void* w, const void* b, double total,
                                    void* out, double* partials, int64_t len,
                                    int members, int dtype,
                                    sfem_stream_t stream);
/* y = a*x + b*y (plain fused vector update used outside the CG core)         */
int sfem_axpby(double a, const void* x, double b, void* y, int64_t count,
               int dtype, sfem_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif  /* SFEM_H_ */
