// Fused Stokes divergence / pressure gradient on compact (facet) connectivity,
// as chain walks: 3D, one wavefront per element (P = 6..8), component-major
// velocity fields (component c of node n at n + c * comp_stride).
//
// Same operators and reference call sites as sfem_stokes.h
// (navier_stokes/navier_stokes.py:313-338: D_local, Dt_local, D, Dt, and the
// two halves of E = D Q D^T, :340-348); connectivity, LDS layout, shared
// scatter and the chain walk are those of sfem_helmholtz_facet.h.  Counters of
// the index-row kernels at 64^3 elements (profiles/traffic_r03.json):
//   stokes_grad_t  4.3e7 memory-side atomic requests in 1.96 ms -- the rate
//                  at which the chip retires them, three components' worth;
//   stokes_div     7.4 GB fetched for 3.4 GB of input (three components and a
//                  scale gathered slot by slot).
// A wave that walks a chain carries the common face of consecutive elements in
// registers: it is gathered once (div) and summed before it is written, its
// interior without an atomic (grad_t).
#pragma once
#include "sfem_helmholtz_facet.h"
#include "sfem_stokes.h"

namespace sfem {

template <typename T>
struct StokesFacetParams {
  StokesParams<T> base;       // fields, geometry, strides (enc / shared_order unused)
  const int32_t* tab;         // (E, 27, 4) facet table of the velocity mesh
  const int32_t* chain_off;   // (S + 1,) segment bounds in chain_elems
  const int32_t* chain_elems;
};

// The cofactors of a slice depend on nothing that is loaded: without a pin the
// scheduler evaluates all eight slices (72 values) ahead of the LDS passes and
// spills.  `dep`: a value the slice really waits for.
template <typename T, int P, int GM>
__device__ __forceinline__ ElemCof<T, P, 3, GM> cof_pinned(
    const ElemCof<T, P, 3, GM>& geom, T dep) {
  ElemCof<T, P, 3, GM> g = geom;
  if (GM == GEO_POINT) {
    asm volatile("" : "+v"(g.lane_off) : "v"(dep));
  } else {
    asm volatile("" : "+v"(g.wbc) : "v"(dep));
    if (GM == GEO_MULTILINEAR) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
        asm volatile("" : "+v"(g.p1[c]), "+v"(g.p2[c]) : "v"(dep));
    }
  }
  return g;
}

// Kernel arguments of the kernels below: (StokesFacetParams, DMat, IMat).
template <typename T, int P>
struct StokesKernarg {
  using DM = DMat<T, P>;
  using IM = IMat<T, P, P - 2>;
  static constexpr size_t DM_OFF =
      (sizeof(StokesFacetParams<T>) + alignof(DM) - 1) / alignof(DM) * alignof(DM);
  static constexpr size_t IM_OFF =
      (DM_OFF + sizeof(DM) + alignof(IM) - 1) / alignof(IM) * alignof(IM);
  static __device__ __forceinline__ const SFEM_CONSTANT_AS DM* dm() {
#if defined(__HIP_DEVICE_COMPILE__)
    return (const SFEM_CONSTANT_AS DM*)((const SFEM_CONSTANT_AS char*)
                                            __builtin_amdgcn_kernarg_segment_ptr() +
                                        DM_OFF);
#else
    return nullptr;
#endif
  }
  static __device__ __forceinline__ const SFEM_CONSTANT_AS IM* im() {
#if defined(__HIP_DEVICE_COMPILE__)
    return (const SFEM_CONSTANT_AS IM*)((const SFEM_CONSTANT_AS char*)
                                            __builtin_amdgcn_kernarg_segment_ptr() +
                                        IM_OFF);
#else
    return nullptr;
#endif
  }
};

// 2 waves per SIMD (256 registers): at 3 both kernels spill (div 300-970 B per
// lane: 2.59 ms against 1.73 for the index-row kernel; at 2: 1.45 ms)
#ifndef SFEM_STOKES_FACET_MINW
#define SFEM_STOKES_FACET_MINW 2
#endif

// Pressure of element e at the velocity points of the lane's line (tq[a] at
// node (a, i, j)): interpolate the last axis (lines [k0, k1, *]), the middle
// one, then axis 0 in registers.  s0: one LDS copy.
template <typename T, int P, int PP>
__device__ __forceinline__ void stokes_pressure_at_nodes(
    const StokesParams<T>& prm, int64_t e, const FacetWave<P>& w, T* s0,
    T (&tq)[P]) {
  const SFEM_CONSTANT_AS IMat<T, P, PP>* im = StokesKernarg<T, P>::im();
  using L = FacetLayout<P>;
  constexpr int NP = PP * PP * PP;
  const int i = w.i, j = w.j;
  const int32_t* penc0 = prm.penc ? prm.penc + e * NP : nullptr;
  const int64_t pbase = e * NP;
  if (w.ok && i < PP && j < PP) {
    T x[PP], y[P];
#pragma unroll
    for (int k = 0; k < PP; ++k) {
      const int slot = (i * PP + j) * PP + k;
      const int64_t pid = penc0 ? (int64_t)penc0[slot] : pbase + slot;
      x[k] = pid >= 0 ? prm.p_in[pid] : T(0);
    }
    interp_fwd_mem<T, P, PP>(im, x, y);
#pragma unroll
    for (int m = 0; m < P; ++m) s0[L::word(i, j, m)] = y[m];
  }
  facet_sync<P>();
  if (w.ok && i < PP) {
    T x[PP], y[P];
#pragma unroll
    for (int k = 0; k < PP; ++k) x[k] = s0[L::word(i, k, j)];
    interp_fwd_mem<T, P, PP>(im, x, y);
#pragma unroll
    for (int m = 0; m < P; ++m) s0[L::word(i, m, j)] = y[m];
  }
  facet_sync<P>();
  {
    T x[PP];
#pragma unroll
    for (int k = 0; k < PP; ++k) x[k] = w.ok ? s0[L::word(k, i, j)] : T(0);
    interp_fwd_mem<T, P, PP>(im, x, tq);
  }
  facet_sync<P>();
}

// out = mask * velocity.scatter(Dt_local(p)) [* scale], one chain segment per
// one-wave workgroup.
template <typename T, int P, int GM, bool OFF32>
__global__ void __launch_bounds__(64, SFEM_STOKES_FACET_MINW)
stokes_grad_t_chain_kernel(StokesFacetParams<T> fprm, DMat<T, P> dm,
                           IMat<T, P, P - 2> im) {
  using L = FacetLayout<P>;
  constexpr int PP = P - 2;
  const StokesParams<T>& prm = fprm.base;
  __shared__ T lds[2 * L::COPY];
  T* s0 = lds;
  T* s1 = lds + L::COPY;
  uint32_t* codes = reinterpret_cast<uint32_t*>(s1);

  FacetWave<P> w;
  w.init();
  const int32_t k0 = fprm.chain_off[blockIdx.x];
  const int32_t k1 = fprm.chain_off[blockIdx.x + 1];
  uint16_t slots[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) slots[q] = g_facet_slots<P>.s[w.lane][q];
  const bool face_inner = FacetLane<P>::cls(w.i) == 1 &&
                          FacetLane<P>::cls(w.j) == 1;
  const int64_t ks = prm.comp_stride;

  // lane constants of the geometry: in-plane weight, node coordinates
  const DMat<T, P>* kdm = reinterpret_cast<const DMat<T, P>*>(
      kernarg_bytes() + StokesKernarg<T, P>::DM_OFF);
  const T lane_w = kdm->w[w.i] * kdm->w[w.j];
  const T lane_s = kdm->x[w.i], lane_t = kdm->x[w.j];
  const bool has_scale = prm.scale != nullptr;
  const bool scale_node = has_scale && prm.scale_comp_stride == 0;

  FacetLane<P> fl, fn;
  typename FacetLane<P>::Raw traw;
  fl.load(fprm.tab, (int64_t)fprm.chain_elems[k0], w.i, w.j);
  fn = fl;
  if (k0 + 1 < k1)
    FacetLane<P>::issue(traw, fprm.tab, (int64_t)fprm.chain_elems[k0 + 1],
                        w.i, w.j);
  T carry[3] = {T(0), T(0), T(0)};
  for (int32_t k = k0; k < k1; ++k) {
    const bool has_pred = k > k0, has_succ = k + 1 < k1;
    const int64_t e = (int64_t)fprm.chain_elems[k];
    ElemCof<T, P, 3, GM> geom;
    geom.init_lane(prm, e, w.ok ? w.lane : 0, lane_w, lane_s, lane_t);
    // one factor per node (scale_comp_stride = 0, the lumped-mass Q of E):
    // gathered once per element, behind the pressure interpolation
    T sc[P];
    if (scale_node) {
#pragma unroll
      for (int a = 0; a < P; ++a)
        sc[a] = w.ok ? *facet_node<const T, OFF32>(prm.scale, fl.code(a))
                     : T(0);
    }
    T tq[P];
    stokes_pressure_at_nodes<T, P, PP>(prm, e, w, s0, tq);
    if (has_succ) fn.finish(traw, w.i, w.j);
    if (k + 2 < k1)
      FacetLane<P>::issue(traw, fprm.tab, (int64_t)fprm.chain_elems[k + 2],
                          w.i, w.j);
    FacetLane<P> fe = fl;
    if (has_pred && face_inner) fe.t[0] &= ~(uint32_t)SFEM_IDX_SHARED;
    if (has_succ) fe.t[2] |= SFEM_IDX_SHARED | SFEM_IDX_DIRICHLET;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      cof_fence(geom);
      T w0[P], acc[P];
      if (w.ok) {
        T dep = tq[0];
#pragma unroll
        for (int a = 0; a < P; ++a) {
          T K[9];
          {
            const SFEM_CONSTANT_AS DMat<T, P>* km = StokesKernarg<T, P>::dm();
            cof_pinned<T, P, GM>(geom, dep).cof_wx(km->w[a], km->x[a], a, K);
          }
          w0[a] = K[c] * tq[a];
          s0[w.own_w + a * L::A] = K[3 + c] * tq[a];
          s1[w.own_w + a * L::A] = K[6 + c] * tq[a];
          dep = w0[a];
        }
      } else {
#pragma unroll
        for (int a = 0; a < P; ++a) w0[a] = T(0);
      }
      facet_sync<P>();
      if (w.ok) {   // transposed derivative along the last axis, copy 1
        T x[P], y[P];
#pragma unroll
        for (int m = 0; m < P; ++m) x[m] = s1[w.last_w + m];
        line_apply_mem<T, P, true>(StokesKernarg<T, P>::dm(), x, y);
#pragma unroll
        for (int m = 0; m < P; ++m) s1[w.last_w + m] = y[m];
      }
      if (w.ok) {   // ... along the middle axis, copy 0
        T x[P], y[P];
#pragma unroll
        for (int m = 0; m < P; ++m) x[m] = s0[w.mid_w + m * L::B];
        line_apply_mem<T, P, true>(StokesKernarg<T, P>::dm(), x, y);
#pragma unroll
        for (int m = 0; m < P; ++m) s0[w.mid_w + m * L::B] = y[m];
      }
      line_apply_mem<T, P, true>(StokesKernarg<T, P>::dm(), w0, acc);
      facet_sync<P>();
      T* og = prm.out + c * ks;
      if (w.ok) {
#pragma unroll
        for (int a = 0; a < P; ++a)
          acc[a] += s0[w.own_w + a * L::A] + s1[w.own_w + a * L::A];
        // a diagonal factor that is the same on every copy of a node
        // commutes with the assembly (E = D QQ^T (Q . D^T))
        if (scale_node) {
#pragma unroll
          for (int a = 0; a < P; ++a) acc[a] *= sc[a];
        } else if (has_scale) {
          const T* sg = prm.scale + c * prm.scale_comp_stride;
#pragma unroll
          for (int a = 0; a < P; ++a) {
            uint32_t code = fl.code(a);
            asm volatile("" : "+v"(code));
            acc[a] *= *facet_node<const T, OFF32>(sg, code);
          }
        }
      }
      facet_sync<P>();
      if (has_pred) acc[0] += carry[c];
      carry[c] = acc[P - 1];
      if (w.ok) {
#pragma unroll
        for (int a = 0; a < P; ++a)
          if (fe.flags(a) & SFEM_IDX_DIRICHLET) acc[a] = T(0);
      }
      facet_scatter_tail<T, P, OFF32>(fe, slots, acc, og, s0, codes, w.own_w,
                                      w.ok);
      facet_sync<P>();
    }
    fl = fn;
  }
}

// p_out = D_local(velocity.gather(scale * u)), one chain segment per one-wave
// workgroup; SECOND: the second half of the split E (only SHARED slots are
// gathered, the projection is added to what the first half stored).
template <typename T, int P, int GM, bool OFF32>
__global__ void __launch_bounds__(64, SFEM_STOKES_FACET_MINW)
stokes_div_chain_kernel(StokesFacetParams<T> fprm, DMat<T, P> dm,
                        IMat<T, P, P - 2> im) {
  using L = FacetLayout<P>;
  constexpr int PP = P - 2, NP = PP * PP * PP;
  const StokesParams<T>& prm = fprm.base;
  __shared__ T lds[2 * L::COPY];
  T* s0 = lds;
  T* s1 = lds + L::COPY;

  FacetWave<P> w;
  w.init();
  const int i = w.i, j = w.j;
  const int32_t k0 = fprm.chain_off[blockIdx.x];
  const int32_t k1 = fprm.chain_off[blockIdx.x + 1];
  const int64_t ks = prm.comp_stride;

  const DMat<T, P>* kdm = reinterpret_cast<const DMat<T, P>*>(
      kernarg_bytes() + StokesKernarg<T, P>::DM_OFF);
  const T lane_w = kdm->w[w.i] * kdm->w[w.j];
  const T lane_s = kdm->x[w.i], lane_t = kdm->x[w.j];
  const bool has_scale = prm.scale != nullptr;

  FacetLane<P> fl, fn;
  typename FacetLane<P>::Raw traw;
  fl.load(fprm.tab, (int64_t)fprm.chain_elems[k0], w.i, w.j);
  fn = fl;
  if (k0 + 1 < k1)
    FacetLane<P>::issue(traw, fprm.tab, (int64_t)fprm.chain_elems[k0 + 1],
                        w.i, w.j);
  T u_last[3] = {T(0), T(0), T(0)};   // scaled values of the carried face
  double pdot = 0.0;
  for (int32_t k = k0; k < k1; ++k) {
    const bool has_pred = k > k0, has_succ = k + 1 < k1;
    const int64_t e = (int64_t)fprm.chain_elems[k];
    ElemCof<T, P, 3, GM> geom;
    geom.init_lane(prm, e, w.ok ? w.lane : 0, lane_w, lane_s, lane_t);
    T tq[P];
#pragma unroll
    for (int a = 0; a < P; ++a) tq[a] = T(0);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      cof_fence(geom);
      T ua[P], d0[P];
      const T* ug = prm.u + c * ks;
      // (codes pinned per component: the three components' gathers would
      // otherwise all be issued up front, 96 registers of loads in flight)
      uint32_t cd[P];
#pragma unroll
      for (int a = 0; a < P; ++a) {
        cd[a] = fl.code(a);
        asm volatile("" : "+v"(cd[a]));
        ua[a] = w.ok ? *facet_node<const T, OFF32>(ug, cd[a]) : T(0);
      }
      // (a per-node factor is re-read per component: holding it across the
      // three costs 16 registers and measured 1.54 vs 1.21 ms without scale)
      if (has_scale) {
        const T* sg = prm.scale + c * prm.scale_comp_stride;
#pragma unroll
        for (int a = 0; a < P; ++a)
          if (w.ok) ua[a] *= *facet_node<const T, OFF32>(sg, cd[a]);
      }
      if (has_pred) ua[0] = u_last[c];
      u_last[c] = ua[P - 1];
      if (c == 0) {   // the next table travels with this element's first gather
        if (has_succ) fn.finish(traw, w.i, w.j);
        if (k + 2 < k1)
          FacetLane<P>::issue(traw, fprm.tab,
                              (int64_t)fprm.chain_elems[k + 2], w.i, w.j);
      }
      line_apply_mem<T, P, false>(StokesKernarg<T, P>::dm(), ua, d0);
      if (w.ok) {
#pragma unroll
        for (int a = 0; a < P; ++a) {
          s0[w.own_w + a * L::A] = ua[a];
          s1[w.own_w + a * L::A] = ua[a];
        }
      }
      facet_sync<P>();
      if (w.ok) {   // last axis, copy 1
        T x[P], y[P];
#pragma unroll
        for (int m = 0; m < P; ++m) x[m] = s1[w.last_w + m];
        line_apply_mem<T, P, false>(StokesKernarg<T, P>::dm(), x, y);
#pragma unroll
        for (int m = 0; m < P; ++m) s1[w.last_w + m] = y[m];
      }
      if (w.ok) {   // middle axis, copy 0
        T x[P], y[P];
#pragma unroll
        for (int m = 0; m < P; ++m) x[m] = s0[w.mid_w + m * L::B];
        line_apply_mem<T, P, false>(StokesKernarg<T, P>::dm(), x, y);
#pragma unroll
        for (int m = 0; m < P; ++m) s0[w.mid_w + m * L::B] = y[m];
      }
      facet_sync<P>();
      if (w.ok) {
#pragma unroll
        for (int a = 0; a < P; ++a) {
          T K[9];
          const T g1 = s0[w.own_w + a * L::A], g2 = s1[w.own_w + a * L::A];
          {
            const SFEM_CONSTANT_AS DMat<T, P>* km = StokesKernarg<T, P>::dm();
            cof_pinned<T, P, GM>(geom, g2).cof_wx(km->w[a], km->x[a], a, K);
          }
          tq[a] += K[c] * d0[a] + K[3 + c] * g1 + K[6 + c] * g2;
        }
      }
      facet_sync<P>();
    }
    // projection onto the pressure basis: axis 0 in registers, then LDS lines
    {
      T r[PP];
      interp_t_mem<T, P, PP>(StokesKernarg<T, P>::im(), tq, r);
      if (w.ok) {
#pragma unroll
        for (int q = 0; q < PP; ++q) s0[L::word(q, i, j)] = r[q];
      }
    }
    facet_sync<P>();
    if (w.ok && i < PP) {   // line [k0 = i, *, j]
      T x[P], y[PP];
#pragma unroll
      for (int m = 0; m < P; ++m) x[m] = s0[L::word(i, m, j)];
      interp_t_mem<T, P, PP>(StokesKernarg<T, P>::im(), x, y);
#pragma unroll
      for (int q = 0; q < PP; ++q) s0[L::word(i, q, j)] = y[q];
    }
    facet_sync<P>();
    if (w.ok && i < PP && j < PP) {   // line [k0 = i, k1 = j, *]
      T x[P], y[PP];
#pragma unroll
      for (int m = 0; m < P; ++m) x[m] = s0[L::word(i, j, m)];
      interp_t_mem<T, P, PP>(StokesKernarg<T, P>::im(), x, y);
      const int32_t* penc0 = prm.penc ? prm.penc + e * NP : nullptr;
#pragma unroll
      for (int q = 0; q < PP; ++q) {
        const int slot = (i * PP + j) * PP + q;
        const int64_t pid = penc0 ? (int64_t)penc0[slot] : e * NP + slot;
        if (pid >= 0) {
          prm.p_out[pid] = y[q];
          if (prm.dot_out) pdot += (double)y[q] * (double)prm.p_in[pid];
        }
      }
    }
    facet_sync<P>();
    fl = fn;
  }
  if (prm.dot_out) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) pdot += __shfl_down(pdot, off, 64);
    if (w.lane == 0)
      unsafeAtomicAdd(&prm.dot_out[blockIdx.x & (SFEM_DOT_SLOTS - 1)], pdot);
  }
}

// ---------------------------------------------------------------------------
// Axis-aligned box elements (geo_mode SFEM_GEO_BOX): x_c depends on the
// reference coordinate of axis c only, so the weighted cofactor matrix is the
// constant diagonal  K_cc = w_q h_(c+1) h_(c+2)  (h_c = A[4 c], the half edge
// of geo_elem) and component c of D^T p / D u takes ONE 1D derivative -- along
// axis c -- instead of three: axis 0 in registers, axes 1 and 2 one LDS pass
// each.  Same chain walk, tables and scatter as the kernels above.
#ifndef SFEM_STOKES_BOX_MINW
#define SFEM_STOKES_BOX_MINW 4
#endif
#ifndef SFEM_STOKES_BOX_GRAD_MINW
#define SFEM_STOKES_BOX_GRAD_MINW 3
#endif

template <typename T>
__device__ __forceinline__ void stokes_box_cof(const StokesParams<T>& prm,
                                               int64_t e, T (&kd)[3]) {
  const T* A = prm.geo_elem + e * 24;
  const T hx = A[0], hy = A[4], hz = A[8];
  kd[0] = hy * hz;
  kd[1] = hz * hx;
  kd[2] = hx * hy;
}

template <typename T, int P, bool OFF32>
__global__ void __launch_bounds__(64, SFEM_STOKES_BOX_GRAD_MINW)
stokes_grad_t_box_kernel(StokesFacetParams<T> fprm, DMat<T, P> dm,
                         IMat<T, P, P - 2> im) {
  using L = FacetLayout<P>;
  constexpr int PP = P - 2;
  const StokesParams<T>& prm = fprm.base;
  __shared__ T lds[2 * L::COPY];
  T* s0 = lds;
  uint32_t* codes = reinterpret_cast<uint32_t*>(lds + L::COPY);

  FacetWave<P> w;
  w.init();
  const int32_t k0 = fprm.chain_off[blockIdx.x];
  const int32_t k1 = fprm.chain_off[blockIdx.x + 1];
  uint16_t slots[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) slots[q] = g_facet_slots<P>.s[w.lane][q];
  const bool face_inner = FacetLane<P>::cls(w.i) == 1 &&
                          FacetLane<P>::cls(w.j) == 1;
  const int64_t ks = prm.comp_stride;
  const DMat<T, P>* kdm = reinterpret_cast<const DMat<T, P>*>(
      kernarg_bytes() + StokesKernarg<T, P>::DM_OFF);
  const T lane_w = kdm->w[w.i] * kdm->w[w.j];
  const bool has_scale = prm.scale != nullptr;
  const bool scale_node = has_scale && prm.scale_comp_stride == 0;

  FacetLane<P> fl, fn;
  typename FacetLane<P>::Raw traw;
  fl.load(fprm.tab, (int64_t)fprm.chain_elems[k0], w.i, w.j);
  fn = fl;
  if (k0 + 1 < k1)
    FacetLane<P>::issue(traw, fprm.tab, (int64_t)fprm.chain_elems[k0 + 1],
                        w.i, w.j);
  T carry[3] = {T(0), T(0), T(0)};
  for (int32_t k = k0; k < k1; ++k) {
    const bool has_pred = k > k0, has_succ = k + 1 < k1;
    const int64_t e = (int64_t)fprm.chain_elems[k];
    T kd[3];
    stokes_box_cof<T>(prm, e, kd);
    T sc[P];
    if (scale_node) {
#pragma unroll
      for (int a = 0; a < P; ++a)
        sc[a] = w.ok ? *facet_node<const T, OFF32>(prm.scale, fl.code(a))
                     : T(0);
    }
    T tq[P];
    stokes_pressure_at_nodes<T, P, PP>(prm, e, w, s0, tq);
    if (has_succ) fn.finish(traw, w.i, w.j);
    if (k + 2 < k1)
      FacetLane<P>::issue(traw, fprm.tab, (int64_t)fprm.chain_elems[k + 2],
                          w.i, w.j);
    FacetLane<P> fe = fl;
    if (has_pred && face_inner) fe.t[0] &= ~(uint32_t)SFEM_IDX_SHARED;
    if (has_succ) fe.t[2] |= SFEM_IDX_SHARED | SFEM_IDX_DIRICHLET;
    {   // quadrature weight of the node
      const SFEM_CONSTANT_AS DMat<T, P>* km = StokesKernarg<T, P>::dm();
#pragma unroll
      for (int a = 0; a < P; ++a) tq[a] *= lane_w * km->w[a];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      T acc[P];
      if (c == 0) {
        T x[P];
#pragma unroll
        for (int a = 0; a < P; ++a) x[a] = kd[0] * tq[a];
        line_apply_mem<T, P, true>(StokesKernarg<T, P>::dm(), x, acc);
      } else {
        if (w.ok) {
#pragma unroll
          for (int a = 0; a < P; ++a) s0[w.own_w + a * L::A] = kd[c] * tq[a];
        }
        facet_sync<P>();
        if (w.ok) {
          T x[P], y[P];
#pragma unroll
          for (int m = 0; m < P; ++m)
            x[m] = c == 1 ? s0[w.mid_w + m * L::B] : s0[w.last_w + m];
          line_apply_mem<T, P, true>(StokesKernarg<T, P>::dm(), x, y);
#pragma unroll
          for (int m = 0; m < P; ++m) {
            if (c == 1) s0[w.mid_w + m * L::B] = y[m];
            else s0[w.last_w + m] = y[m];
          }
        }
        facet_sync<P>();
#pragma unroll
        for (int a = 0; a < P; ++a)
          acc[a] = w.ok ? s0[w.own_w + a * L::A] : T(0);
        facet_sync<P>();
      }
      T* og = prm.out + c * ks;
      if (w.ok) {
        if (scale_node) {
#pragma unroll
          for (int a = 0; a < P; ++a) acc[a] *= sc[a];
        } else if (has_scale) {
          const T* sg = prm.scale + c * prm.scale_comp_stride;
#pragma unroll
          for (int a = 0; a < P; ++a) {
            uint32_t code = fl.code(a);
            asm volatile("" : "+v"(code));
            acc[a] *= *facet_node<const T, OFF32>(sg, code);
          }
        }
      }
      if (has_pred) acc[0] += carry[c];
      carry[c] = acc[P - 1];
      if (w.ok) {
#pragma unroll
        for (int a = 0; a < P; ++a)
          if (fe.flags(a) & SFEM_IDX_DIRICHLET) acc[a] = T(0);
      }
      facet_scatter_tail<T, P, OFF32>(fe, slots, acc, og, s0, codes, w.own_w,
                                      w.ok);
      facet_sync<P>();
    }
    fl = fn;
  }
}

template <typename T, int P, bool OFF32>
__global__ void __launch_bounds__(64, SFEM_STOKES_BOX_MINW)
stokes_div_box_kernel(StokesFacetParams<T> fprm, DMat<T, P> dm,
                      IMat<T, P, P - 2> im) {
  using L = FacetLayout<P>;
  constexpr int PP = P - 2, NP = PP * PP * PP;
  const StokesParams<T>& prm = fprm.base;
  __shared__ T lds[L::COPY];
  T* s0 = lds;

  FacetWave<P> w;
  w.init();
  const int i = w.i, j = w.j;
  const int32_t k0 = fprm.chain_off[blockIdx.x];
  const int32_t k1 = fprm.chain_off[blockIdx.x + 1];
  const int64_t ks = prm.comp_stride;
  const DMat<T, P>* kdm = reinterpret_cast<const DMat<T, P>*>(
      kernarg_bytes() + StokesKernarg<T, P>::DM_OFF);
  const T lane_w = kdm->w[w.i] * kdm->w[w.j];
  const bool has_scale = prm.scale != nullptr;

  FacetLane<P> fl, fn;
  typename FacetLane<P>::Raw traw;
  fl.load(fprm.tab, (int64_t)fprm.chain_elems[k0], w.i, w.j);
  fn = fl;
  if (k0 + 1 < k1)
    FacetLane<P>::issue(traw, fprm.tab, (int64_t)fprm.chain_elems[k0 + 1],
                        w.i, w.j);
  T u_last[3] = {T(0), T(0), T(0)};   // scaled values of the carried face
  double pdot = 0.0;
  for (int32_t k = k0; k < k1; ++k) {
    const bool has_pred = k > k0, has_succ = k + 1 < k1;
    const int64_t e = (int64_t)fprm.chain_elems[k];
    T kd[3];
    stokes_box_cof<T>(prm, e, kd);
    T tq[P];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      T ua[P];
      const T* ug = prm.u + c * ks;
      uint32_t cd[P];
#pragma unroll
      for (int a = 0; a < P; ++a) {
        cd[a] = fl.code(a);
        asm volatile("" : "+v"(cd[a]));
        ua[a] = w.ok ? *facet_node<const T, OFF32>(ug, cd[a]) : T(0);
      }
      if (has_scale) {
        const T* sg = prm.scale + c * prm.scale_comp_stride;
#pragma unroll
        for (int a = 0; a < P; ++a)
          if (w.ok) ua[a] *= *facet_node<const T, OFF32>(sg, cd[a]);
      }
      if (has_pred) ua[0] = u_last[c];
      u_last[c] = ua[P - 1];
      if (c == 0) {   // the next table travels with this element's first gather
        if (has_succ) fn.finish(traw, w.i, w.j);
        if (k + 2 < k1)
          FacetLane<P>::issue(traw, fprm.tab,
                              (int64_t)fprm.chain_elems[k + 2], w.i, w.j);
        T d0[P];
        line_apply_mem<T, P, false>(StokesKernarg<T, P>::dm(), ua, d0);
#pragma unroll
        for (int a = 0; a < P; ++a) tq[a] = kd[0] * d0[a];
      } else {
        if (w.ok) {
#pragma unroll
          for (int a = 0; a < P; ++a) s0[w.own_w + a * L::A] = ua[a];
        }
        facet_sync<P>();
        if (w.ok) {
          T x[P], y[P];
#pragma unroll
          for (int m = 0; m < P; ++m)
            x[m] = c == 1 ? s0[w.mid_w + m * L::B] : s0[w.last_w + m];
          line_apply_mem<T, P, false>(StokesKernarg<T, P>::dm(), x, y);
#pragma unroll
          for (int m = 0; m < P; ++m) {
            if (c == 1) s0[w.mid_w + m * L::B] = y[m];
            else s0[w.last_w + m] = y[m];
          }
        }
        facet_sync<P>();
        if (w.ok) {
#pragma unroll
          for (int a = 0; a < P; ++a) tq[a] += kd[c] * s0[w.own_w + a * L::A];
        }
        facet_sync<P>();
      }
    }
    {
      const SFEM_CONSTANT_AS DMat<T, P>* km = StokesKernarg<T, P>::dm();
#pragma unroll
      for (int a = 0; a < P; ++a) tq[a] *= lane_w * km->w[a];
    }
    // projection onto the pressure basis: axis 0 in registers, then LDS lines
    {
      T r[PP];
      interp_t_mem<T, P, PP>(StokesKernarg<T, P>::im(), tq, r);
      if (w.ok) {
#pragma unroll
        for (int q = 0; q < PP; ++q) s0[L::word(q, i, j)] = r[q];
      }
    }
    facet_sync<P>();
    if (w.ok && i < PP) {   // line [k0 = i, *, j]
      T x[P], y[PP];
#pragma unroll
      for (int m = 0; m < P; ++m) x[m] = s0[L::word(i, m, j)];
      interp_t_mem<T, P, PP>(StokesKernarg<T, P>::im(), x, y);
#pragma unroll
      for (int q = 0; q < PP; ++q) s0[L::word(i, q, j)] = y[q];
    }
    facet_sync<P>();
    if (w.ok && i < PP && j < PP) {   // line [k0 = i, k1 = j, *]
      T x[P], y[PP];
#pragma unroll
      for (int m = 0; m < P; ++m) x[m] = s0[L::word(i, j, m)];
      interp_t_mem<T, P, PP>(StokesKernarg<T, P>::im(), x, y);
      const int32_t* penc0 = prm.penc ? prm.penc + e * NP : nullptr;
#pragma unroll
      for (int q = 0; q < PP; ++q) {
        const int slot = (i * PP + j) * PP + q;
        const int64_t pid = penc0 ? (int64_t)penc0[slot] : e * NP + slot;
        if (pid >= 0) {
          prm.p_out[pid] = y[q];
          if (prm.dot_out) pdot += (double)y[q] * (double)prm.p_in[pid];
        }
      }
    }
    facet_sync<P>();
    fl = fn;
  }
  if (prm.dot_out) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) pdot += __shfl_down(pdot, off, 64);
    if (w.lane == 0)
      unsafeAtomicAdd(&prm.dot_out[blockIdx.x & (SFEM_DOT_SLOTS - 1)], pdot);
  }
}

// mode 0 = div, 1 = grad_t
template <typename T, int P>
int launch_stokes_facet(const StokesFacetParams<T>& fprm, int mode,
                        int64_t num_chains, int64_t field_reals,
                        hipStream_t stream) {
  constexpr int PP = P - 2;
  const StokesParams<T>& prm = fprm.base;
  if (num_chains > 0x7fffffff) {
    set_error("stokes (facet): too many workgroups (%lld)",
              (long long)num_chains);
    return SFEM_EINVAL;
  }
  const DMat<T, P> dm =
      make_dmat<T, P>(prm.dmat_host, prm.weights_host, prm.nodes_host);
  IMat<T, P, PP> im;
  for (int q = 0; q < P * PP; ++q)
    im.m[q] = prm.interp_host ? prm.interp_host[q] : T(0);
  const dim3 grid((unsigned)num_chains), block(64);
  const char* force64 = getenv("SFEM_FACET_OFF64");
  const bool off32 = (uint64_t)field_reals * sizeof(T) < ((uint64_t)1 << 32) &&
                     !(force64 && force64[0] == '1');
#define SFEM_STOKES_FACET_GO(GMV)                                             \
  do {                                                                        \
    if (mode == 1) {                                                          \
      if (off32)                                                              \
        hipLaunchKernelGGL((stokes_grad_t_chain_kernel<T, P, GMV, true>),     \
                           grid, block, 0, stream, fprm, dm, im);             \
      else                                                                    \
        hipLaunchKernelGGL((stokes_grad_t_chain_kernel<T, P, GMV, false>),    \
                           grid, block, 0, stream, fprm, dm, im);             \
    } else {                                                                  \
      if (off32)                                                              \
        hipLaunchKernelGGL((stokes_div_chain_kernel<T, P, GMV, true>), grid,  \
                           block, 0, stream, fprm, dm, im);                   \
      else                                                                    \
        hipLaunchKernelGGL((stokes_div_chain_kernel<T, P, GMV, false>), grid, \
                           block, 0, stream, fprm, dm, im);                   \
    }                                                                         \
  } while (0)
  switch (prm.geo_mode) {
    case GEO_BOX:
      if (mode == 1) {
        if (off32)
          hipLaunchKernelGGL((stokes_grad_t_box_kernel<T, P, true>), grid,
                             block, 0, stream, fprm, dm, im);
        else
          hipLaunchKernelGGL((stokes_grad_t_box_kernel<T, P, false>), grid,
                             block, 0, stream, fprm, dm, im);
      } else {
        if (off32)
          hipLaunchKernelGGL((stokes_div_box_kernel<T, P, true>), grid, block,
                             0, stream, fprm, dm, im);
        else
          hipLaunchKernelGGL((stokes_div_box_kernel<T, P, false>), grid, block,
                             0, stream, fprm, dm, im);
      }
      break;
    case GEO_POINT: SFEM_STOKES_FACET_GO(GEO_POINT); break;
    case GEO_AFFINE: SFEM_STOKES_FACET_GO(GEO_AFFINE); break;
    default: SFEM_STOKES_FACET_GO(GEO_MULTILINEAR); break;
  }
#undef SFEM_STOKES_FACET_GO
  SFEM_LAUNCH_CHECK();
  return SFEM_OK;
}

template <typename T>
int dispatch_stokes_facet(const StokesFacetParams<T>& fprm, int P, int mode,
                          int64_t num_chains, int64_t field_reals,
                          hipStream_t stream);

inline bool stokes_facet_supported_p(int P) { return P >= 6 && P <= 8; }

#define SFEM_DEFINE_STOKES_FACET_DISPATCH(TYPE)                               \
  template <>                                                                 \
  int dispatch_stokes_facet<TYPE>(const StokesFacetParams<TYPE>& fprm, int P, \
                                  int mode, int64_t num_chains,               \
                                  int64_t field_reals, hipStream_t stream) {  \
    switch (P) {                                                              \
      case 6:                                                                 \
        return launch_stokes_facet<TYPE, 6>(fprm, mode, num_chains,           \
                                            field_reals, stream);             \
      case 7:                                                                 \
        return launch_stokes_facet<TYPE, 7>(fprm, mode, num_chains,           \
                                            field_reals, stream);             \
      case 8:                                                                 \
        return launch_stokes_facet<TYPE, 8>(fprm, mode, num_chains,           \
                                            field_reals, stream);             \
      default:                                                                \
        set_error("stokes (facet): P=%d outside 6..8", P);                    \
        return SFEM_EUNSUPPORTED;                                             \
    }                                                                         \
  }

}  // namespace sfem
