"""Tensor-level wrappers over the C-ABI (`include/sfem.h`).

PyTorch is plumbing here: it owns the HBM allocations and the HIP stream; every
number is produced by `libsfem_hip.so`.  All tensors must live on the GPU.
"""

from __future__ import annotations

import ctypes
import os

import numpy as np
import torch

from swirl_fem_amd import switches
from swirl_fem_amd import _lib

_DT = {torch.float32: _lib.SFEM_F32, torch.float64: _lib.SFEM_F64}


def _dtype_code(t: torch.Tensor) -> int:
  try:
    return _DT[t.dtype]
  except KeyError:
    raise TypeError(f'unsupported dtype {t.dtype}; use float32 or float64')


def _dev(*tensors):
  """Checks that all tensors are contiguous GPU tensors on one device."""
  dev = None
  for t in tensors:
    if t is None:
      continue
    if not t.is_cuda:
      raise RuntimeError(
          'swirl_fem_amd kernels run on MI355X device tensors only; got a '
          f'{t.device} tensor (there is no CPU fallback)')
    if not t.is_contiguous():
      raise ValueError('expected a contiguous tensor')
    if dev is None:
      dev = t.device
    elif t.device != dev:
      raise ValueError(f'tensors on different devices: {dev} vs {t.device}')
  return dev


def _ptr(t):
  return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(dev):
  return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _idx(indices: torch.Tensor) -> torch.Tensor:
  if indices.dtype != torch.int32:
    indices = indices.to(torch.int32)
  return indices.contiguous()


# ---------------------------------------------------------------- gather etc
def gather(u, indices, fill):
  indices = _idx(indices)
  u = u.contiguous()
  dev = _dev(u, indices)
  out = torch.empty(indices.shape, dtype=u.dtype, device=dev)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_gather(
        _ptr(u), _ptr(indices), _ptr(out), indices.numel(), float(fill),
        _dtype_code(u), _stream(dev)), 'sfem_gather')
  return out


def gather_rows(x, indices):
  """x (N, nc), indices (...) -> (..., nc); SENTINEL rows are zero.

  A component-major `x` gives a component-major result (one scalar gather per
  contiguous component strip)."""
  indices = _idx(indices)
  if is_component_major(x):
    from swirl_fem_amd.core import layout
    out = layout.empty_component_major(tuple(indices.shape) + (x.shape[-1],),
                                       x.dtype, x.device)
    for k in range(x.shape[-1]):
      out[..., k].copy_(gather(x[:, k], indices, 0.0))
    return out
  x = x.contiguous()
  dev = _dev(x, indices)
  nc = x.shape[-1]
  out = torch.empty(tuple(indices.shape) + (nc,), dtype=x.dtype, device=dev)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_gather_rows(
        _ptr(x), _ptr(indices), _ptr(out), indices.numel(), nc,
        _dtype_code(x), _stream(dev)), 'sfem_gather_rows')
  return out


def scatter_add(u_local, indices, num_nodes, ncomp=1):
  indices = _idx(indices)
  if ncomp > 1 and is_component_major(u_local):
    from swirl_fem_amd.core import layout
    out = layout.empty_component_major((num_nodes, ncomp), u_local.dtype,
                                       u_local.device)
    for k in range(ncomp):
      out[:, k].copy_(scatter_add(u_local[..., k], indices, num_nodes))
    return out
  u_local = u_local.contiguous()
  dev = _dev(u_local, indices)
  shape = (num_nodes,) if ncomp == 1 and u_local.dim() == indices.dim() else (
      num_nodes, ncomp)
  out = torch.empty(shape, dtype=u_local.dtype, device=dev)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_scatter_add(
        _ptr(u_local), _ptr(indices), _ptr(out), indices.numel(), num_nodes,
        ncomp, _dtype_code(u_local), _stream(dev)), 'sfem_scatter_add')
  return out


def scatter_csr(u_local, offsets, slots, num_nodes, ncomp=1):
  u_local = u_local.contiguous()
  dev = _dev(u_local, offsets, slots)
  shape = (num_nodes,) if ncomp == 1 else (num_nodes, ncomp)
  out = torch.empty(shape, dtype=u_local.dtype, device=dev)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_scatter_csr(
        _ptr(u_local), _ptr(offsets), _ptr(slots), _ptr(out), num_nodes, ncomp,
        _dtype_code(u_local), _stream(dev)), 'sfem_scatter_csr')
  return out


_class_cache = {}


def _exchange_classes(gather_indices, unique_indices, dev):
  """CSR of the periodic classes on `dev`: (members, offsets, num_classes).

  `gather_indices[i]` is a participating node and `unique_indices[i]` its
  class (core/gather_scatter.py:284-315); entries of -1 are padding.  Cached
  per index pair (the arrays of a mesh are static)."""
  key = (id(gather_indices), id(unique_indices), str(dev))
  cached = _class_cache.get(key)
  if (cached is not None and cached[0] is gather_indices and
      cached[1] is unique_indices):
    return cached[2:]
  gi = (gather_indices.detach().cpu().numpy()
        if isinstance(gather_indices, torch.Tensor)
        else np.asarray(gather_indices)).astype(np.int64).reshape(-1)
  ui = np.asarray(unique_indices).astype(np.int64).reshape(-1)
  if gi.shape != ui.shape:
    raise ValueError('gather and unique indices must have the same length')
  keep = gi >= 0
  gi, ui = gi[keep], ui[keep]
  order = np.lexsort((gi, ui))                  # by class, then by node id
  classes, counts = np.unique(ui, return_counts=True)
  offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
  members = torch.as_tensor(gi[order].astype(np.int32), device=dev)
  out = (members, torch.as_tensor(offsets, device=dev), len(classes))
  if len(_class_cache) >= 64:                   # meshes come and go
    _class_cache.pop(next(iter(_class_cache)))
  _class_cache[key] = (gather_indices, unique_indices) + out
  return out


def exchange_local(u, gather_indices, unique_indices, inplace=False):
  """Unpartitioned QQ^T; `unique_indices` is a host array (static).

  One launch for all components, in place on a copy of `u` (or on `u` itself
  with `inplace=True`: only the periodic images change)."""
  cm = is_component_major(u)
  if not cm:
    u = u.contiguous()
  dev = _dev(u.movedim(-1, 0) if cm else u)
  members, offsets, num_classes = _exchange_classes(gather_indices,
                                                    unique_indices, dev)
  out = u if inplace else u.clone()             # clone keeps the dense layout
  ncomp, ns, cs = _node_view(out)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_exchange_classes(
        _ptr(out), _ptr(members), _ptr(offsets), num_classes, ncomp, ns, cs,
        _dtype_code(out), _stream(dev)), 'sfem_exchange_classes')
  return out


def exchange_classes_(u, members, offsets, num_classes):
  """In place: every member position of a class receives the sum over the
  class (`sfem_exchange_classes`; members / offsets: int32 CSR on the device)."""
  if num_classes == 0:
    return u
  cm = is_component_major(u)
  if not (cm or u.is_contiguous()):
    raise ValueError('exchange_classes_: dense field expected')
  dev = _dev(u.movedim(-1, 0) if cm else u, members, offsets)
  ncomp, ns, cs = _node_view(u)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_exchange_classes(
        _ptr(u), _ptr(members), _ptr(offsets), int(num_classes), ncomp, ns, cs,
        _dtype_code(u), _stream(dev)), 'sfem_exchange_classes')
  return u


def exchange_local_atomic(u, gather_indices, unique_indices):
  """The same QQ^T through `sfem_exchange_local` (segment sums by atomics into
  a workspace, then expansion): the C-ABI's out-of-place form."""
  gidx = _idx(gather_indices)
  u = u.contiguous()
  dev = _dev(u, gidx)
  uni = torch.as_tensor(np.ascontiguousarray(unique_indices),
                        dtype=torch.int32, device=dev)
  num_unique = int(np.max(unique_indices)) + 1 if len(unique_indices) else 0
  ncomp = 1 if u.dim() == 1 else u.shape[-1]
  sums = torch.empty((max(num_unique, 1), ncomp), dtype=u.dtype, device=dev)
  out = torch.empty_like(u)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_exchange_local(
        _ptr(u), _ptr(out), _ptr(gidx), _ptr(uni), gidx.numel(), u.shape[0],
        _ptr(sums), num_unique, ncomp, _dtype_code(u), _stream(dev)),
               'sfem_exchange_local')
  return out


def subtract_weighted_mean(w, b, total, partials, out=None, dot_result=None):
  """out = w - (b . w / total) 1 (two launches, no host synchronisation).
  `dot_result = (scalars, slot)`: scalars[slot] += w . out as well."""
  w, b = w.contiguous(), b.contiguous()
  if w.shape != b.shape or w.dtype != b.dtype:
    raise ValueError('subtract_weighted_mean: operands differ in shape / dtype')
  if out is None:
    out = torch.empty_like(w)
  dev = _dev(w, b, out, partials)
  if partials.dtype != torch.float64 or partials.numel() < _lib.SFEM_DOT_SLOTS:
    raise ValueError('partials: SFEM_DOT_SLOTS float64 values')
  with torch.cuda.device(dev):
    dot_ptr = None
    if dot_result is not None:
      scalars, slot = dot_result
      _dev(scalars)
      if scalars.dtype != torch.float64:
        raise ValueError('dot_result: float64 scalars')
      dot_ptr = ctypes.c_void_p(scalars.data_ptr() + 8 * slot)
    _lib.check(_lib.load().sfem_subtract_weighted_mean(
        _ptr(w), _ptr(b), float(total), _ptr(out), _ptr(partials), w.numel(),
        dot_ptr, _dtype_code(w), _stream(dev)), 'sfem_subtract_weighted_mean')
  return out


def pack(u, idx):
  ncomp = 1 if u.dim() == 1 else u.shape[-1]
  u = u.contiguous()
  dev = _dev(u, idx)
  shape = (idx.numel(),) if u.dim() == 1 else (idx.numel(), ncomp)
  buf = torch.empty(shape, dtype=u.dtype, device=dev)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_pack(
        _ptr(u), _ptr(idx), _ptr(buf), idx.numel(), ncomp, _dtype_code(u),
        _stream(dev)), 'sfem_pack')
  return buf


def unpack_add(buf, idx, u):
  """In place: u[idx] += buf."""
  ncomp = 1 if u.dim() == 1 else u.shape[-1]
  dev = _dev(buf, idx, u)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_unpack_add(
        _ptr(buf), _ptr(idx), _ptr(u), idx.numel(), ncomp, _dtype_code(u),
        _stream(dev)), 'sfem_unpack_add')
  return u


def _node_view(u):
  """(ncomp, node_stride, comp_stride) of a dense (N,) / (N, nc) field."""
  if u.dim() == 1:
    if u.stride(0) != 1:
      raise ValueError('nodal vector must be dense')
    return 1, 1, 1
  if u.dim() != 2 or not (u.is_contiguous() or is_component_major(u)):
    raise ValueError('field must be (N,), row-major (N, nc) or component-major')
  return u.shape[1], u.stride(0), u.stride(1)


def pack_strided(u, idx, out=None):
  """buf[i, k] = u[idx[i], k] for any dense field layout (one launch)."""
  ncomp, ns, cs = _node_view(u)
  dev = _dev(u.movedim(-1, 0) if is_component_major(u) else u, idx)
  shape = (idx.numel(),) if u.dim() == 1 else (idx.numel(), ncomp)
  buf = out if out is not None else torch.empty(shape, dtype=u.dtype,
                                                device=dev)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_pack_strided(
        _ptr(u), _ptr(idx), _ptr(buf), idx.numel(), ncomp, ns, cs,
        _dtype_code(u), _stream(dev)), 'sfem_pack_strided')
  return buf


def unpack_add_atomic(buf, idx, u):
  """In place: u[idx[i], k] += buf[i, k]; idx may repeat nodes."""
  ncomp, ns, cs = _node_view(u)
  dev = _dev(buf, idx, u.movedim(-1, 0) if is_component_major(u) else u)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_unpack_add_atomic(
        _ptr(buf), _ptr(idx), _ptr(u), idx.numel(), ncomp, ns, cs,
        _dtype_code(u), _stream(dev)), 'sfem_unpack_add_atomic')
  return u


# ------------------------------------------------------------------ geometry
def geom_factors(elem_coords, interp1, grad1, ndim, P, q, want_quad_coords):
  elem_coords = elem_coords.contiguous()
  dev = _dev(elem_coords, interp1, grad1)
  E = elem_coords.shape[0]
  Q = q ** ndim
  dt = elem_coords.dtype
  invjac = torch.empty((E, Q, ndim, ndim), dtype=dt, device=dev)
  jacdet = torch.empty((E, Q), dtype=dt, device=dev)
  quad = (torch.empty((E, Q, ndim), dtype=dt, device=dev)
          if want_quad_coords else None)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_geom_factors(
        _ptr(elem_coords), _ptr(interp1), _ptr(grad1), E, ndim, P, q,
        _ptr(invjac), _ptr(jacdet), _ptr(quad), _dtype_code(elem_coords),
        _stream(dev)), 'sfem_geom_factors')
  return invjac, jacdet, quad


def basis_eval(u_local, interp1, grad1, invjac, ndim, P, q, collocated,
               want_val, want_grad):
  """u_local (E, n, nc) -> val (E, Q, nc), grad (E, Q, d, nc)."""
  u_local = u_local.contiguous()
  dev = _dev(u_local, interp1, grad1, invjac)
  E, _, nc = u_local.shape
  Q = q ** ndim
  dt = u_local.dtype
  val = torch.empty((E, Q, nc), dtype=dt, device=dev) if want_val else None
  grad = (torch.empty((E, Q, ndim, nc), dtype=dt, device=dev)
          if want_grad else None)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_basis_eval(
        _ptr(u_local), _ptr(interp1), _ptr(grad1), _ptr(invjac), _ptr(val),
        _ptr(grad), E, ndim, P, q, nc, int(collocated), _dtype_code(u_local),
        _stream(dev)), 'sfem_basis_eval')
  return val, grad


def basis_eval_t(c0, c1, interp1, grad1, invjac, wdet, ndim, P, q, nc,
                 collocated):
  """Transpose of `basis_eval` with quadrature: -> (E, n, nc)."""
  c0 = None if c0 is None else c0.contiguous()
  c1 = None if c1 is None else c1.contiguous()
  dev = _dev(c0, c1, interp1, grad1, invjac, wdet)
  E = wdet.shape[0]
  out = torch.empty((E, P ** ndim, nc), dtype=wdet.dtype, device=dev)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_basis_eval_t(
        _ptr(c0), _ptr(c1), _ptr(interp1), _ptr(grad1), _ptr(invjac),
        _ptr(wdet), _ptr(out), E, ndim, P, q, nc, int(collocated),
        _dtype_code(wdet), _stream(dev)), 'sfem_basis_eval_t')
  return out


# ----------------------------------------------------------------- helmholtz
def helmholtz_setup(invjac, jacdet, weights_nd):
  dev = _dev(invjac, jacdet, weights_nd)
  E, Q, ndim, _ = invjac.shape
  ng = ndim * (ndim + 1) // 2
  geo = torch.empty((E, ng + 1, Q), dtype=invjac.dtype, device=dev)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_helmholtz_setup(
        _ptr(invjac), _ptr(jacdet), _ptr(weights_nd), _ptr(geo), E, ndim, Q,
        _dtype_code(invjac), _stream(dev)), 'sfem_helmholtz_setup')
  return geo


def encode_elements(elements, dirichlet_u8, multiplicity, slot_shared=None):
  elements = _idx(elements)
  dev = _dev(elements, dirichlet_u8, multiplicity, slot_shared)
  enc = torch.empty_like(elements)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_encode_elements(
        _ptr(elements), _ptr(dirichlet_u8), _ptr(multiplicity),
        _ptr(slot_shared), _ptr(enc), elements.numel(), _stream(dev)),
        'sfem_encode_elements')
  return enc


def helmholtz_setup_multilinear(elem_coords, ndim, P):
  """elem_coords (E, n, d) -> (E, 24) multilinear-map coefficients."""
  elem_coords = elem_coords.contiguous()
  dev = _dev(elem_coords)
  E = elem_coords.shape[0]
  geo_elem = torch.empty((E, 24), dtype=elem_coords.dtype, device=dev)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_helmholtz_setup_multilinear(
        _ptr(elem_coords), _ptr(geo_elem), E, ndim, P,
        _dtype_code(elem_coords), _stream(dev)),
        'sfem_helmholtz_setup_multilinear')
  return geo_elem


_KERNARG_CHECKED = set()


def kernarg_selftest(dev):
  """Once per process and device: the facet kernels' view of their matrix
  argument through the kernarg segment is what was passed
  (`sfem_kernarg_selftest`); raises otherwise."""
  if dev in _KERNARG_CHECKED:
    return
  bad = torch.zeros(1, dtype=torch.int32, device=dev)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_kernarg_selftest(_ptr(bad), _stream(dev)),
               'sfem_kernarg_selftest')
  if int(bad.item()) != 0:
    raise _lib.SfemError(
        'the kernel-argument layout differs from what the facet kernels '
        f'assume (probe flags {int(bad.item())}): rebuild libsfem_hip.so with '
        'the toolchain it was written for (FacetKernarg::MAT_OFF)')
  _KERNARG_CHECKED.add(dev)


def facet_table(elements, dirichlet_u8, multiplicity, P):
  """Compact connectivity (`sfem_facet_table_build`): `(E, 27, 4)` int32
  table and the `(E,)` bool mask of the elements it describes exactly."""
  elements = _idx(elements)
  dev = _dev(elements, dirichlet_u8, multiplicity)
  kernarg_selftest(dev)      # every facet launch starts from a table
  E = elements.shape[0]
  tab = torch.empty((E, 27, 4), dtype=torch.int32, device=dev)
  ok = torch.empty((E,), dtype=torch.uint8, device=dev)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_facet_table_build(
        _ptr(elements), _ptr(dirichlet_u8), _ptr(multiplicity), _ptr(tab),
        _ptr(ok), E, multiplicity.shape[0], P, _stream(dev)),
        'sfem_facet_table_build')
  return tab, ok != 0


def helmholtz_setup_affine(geo_elem, box_tol):
  """(E, 24) coefficients of affine elements -> (E, 8) constants
  (`sfem_helmholtz_setup_affine`); column 7 flags Cartesian boxes."""
  geo_elem = geo_elem.contiguous()
  dev = _dev(geo_elem)
  E = geo_elem.shape[0]
  out = torch.empty((E, 8), dtype=geo_elem.dtype, device=dev)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_helmholtz_setup_affine(
        _ptr(geo_elem), _ptr(out), E, float(box_tol), _dtype_code(geo_elem),
        _stream(dev)), 'sfem_helmholtz_setup_affine')
  return out


def _host(a, dtype):
  np_dt = np.float64 if dtype == torch.float64 else np.float32
  return None if a is None else np.ascontiguousarray(a, dtype=np_dt)


def _hptr(a):
  return None if a is None else a.ctypes.data


def _dptr(t):
  return None if t is None else t.data_ptr()


def _helmholtz_args(u, out, enc, part, host, ndim, P, num_elements, num_nodes,
                    lambda0, lambda1, zero_range, dot_out=None,
                    layered_extent=0, dot_slots=0):
  """Builds `sfem_helmholtz_args`; `part` = dict(geo_mode, geo, geo_elem,
  geo_index, elem_list), `host` = dict(dmat, weights, nodes) NumPy arrays
  (kept alive by the caller for the duration of the call)."""
  vec = u.dim() != (1 if enc is not None else 2)
  ncomp = u.shape[-1] if vec else 1
  node_stride = comp_stride = 0
  if vec and not u.is_contiguous():
    # component-major storage viewed as (..., ncomp): every component is a
    # contiguous strip
    node_stride, comp_stride = 1, u.stride(-1)
  lst = part.get('elem_list')
  so = part.get('shared_order')
  cl = part.get('cluster') if enc is not None else None
  if cl is not None:
    enc, so = cl.enc, None
  ft = part.get('facet_table') if enc is not None else None
  ch = None
  if layered_extent:
    # same launches, layered table form; the layer plan was made for exactly
    # these chain segments (or for none)
    ft, so = part['layered_table'], None
    ch = part.get('chains') if part.get('layered_chains') else None
  elif ft is not None:
    so = None
    # chains walk scalar fields; a component-major vector field is walked
    # component by component
    if ((not vec or node_stride == 1) and
        switches.get('SFEM_CHAIN') != '0' and
        (not vec or switches.get('SFEM_CHAIN_VECTOR') != '0')):
      ch = part.get('chains')        # (offsets, elems) int32 device tensors
  return _lib.HelmholtzArgs(
      u=u.data_ptr(), out=out.data_ptr(), enc=_dptr(enc),
      geo=_dptr(part.get('geo')), geo_elem=_dptr(part.get('geo_elem')),
      geo_index=_dptr(part.get('geo_index')), elem_list=_dptr(lst),
      dmat=_hptr(host['dmat']), weights=_hptr(host.get('weights')),
      nodes=_hptr(host.get('nodes')), num_elements=num_elements,
      num_listed=0 if lst is None else lst.numel(), num_nodes=num_nodes,
      zero_begin=int(zero_range[0]), zero_end=int(zero_range[1]), ndim=ndim,
      P=P, ncomp=ncomp, dtype=_dtype_code(u), geo_mode=part['geo_mode'],
      colored=int(bool(part.get('colored', False))), lambda0=float(lambda0),
      lambda1=float(lambda1), node_stride=node_stride,
      comp_stride=comp_stride,
      dot_out=_dptr(dot_out),
      shared_order=_dptr(so if enc is not None else None),
      shared_stride=0 if so is None or enc is None else so.shape[1],
      cluster_elems=_dptr(cl.elems if cl is not None else None),
      cluster_offsets=_dptr(cl.offsets if cl is not None else None),
      cluster_nodes=_dptr(cl.nodes if cl is not None else None),
      num_clusters=0 if cl is None else cl.num_clusters,
      facet_table=_dptr(ft),
      geo_const=_dptr(part.get('geo_const') if ft is not None else None),
      chain_offsets=_dptr(ch[0] if ch is not None else None),
      chain_elems=_dptr(ch[1] if ch is not None else None),
      num_chains=0 if ch is None else ch[0].numel() - 1,
      layered_extent=int(layered_extent), dot_slots=int(dot_slots))


_CLUSTER_LIMITS = {}


def helmholtz_cluster_limits(P, dtype):
  """(cluster_size, max_shared) of the cluster kernels for (P, dtype), or None
  (`sfem_helmholtz_cluster_limits`)."""
  key = (P, dtype)
  if key not in _CLUSTER_LIMITS:
    size, kmax = ctypes.c_int32(0), ctypes.c_int32(0)
    code = _lib.SFEM_F64 if dtype == torch.float64 else _lib.SFEM_F32
    rc = _lib.load().sfem_helmholtz_cluster_limits(
        P, code, ctypes.byref(size), ctypes.byref(kmax))
    _CLUSTER_LIMITS[key] = None if rc else (size.value, kmax.value)
  return _CLUSTER_LIMITS[key]


def helmholtz_kernel_name(real, P, ndim, scalar, geo_mode, part, mass,
                          layered=False):
  """Mirror of `launch_helmholtz`'s choice (csrc/sfem_helmholtz.h).  Facet
  kernels: the name up to the addressing-width argument; `layered` (scalar
  fields through `helmholtz_apply_layered`) appends `*, true>`: the last
  template argument of those instantiations."""
  b = lambda v: 'true' if v else 'false'
  if part.get('facet_table') is not None:
    elem = ('sfem::BoxElem<%s, %d, %s>' % (real, P, b(mass)) if geo_mode == 5
            else 'sfem::FacetElem<%s, %d, %d, %s>' % (real, P, geo_mode,
                                                     b(mass)))
    if layered:
      if part.get('chains') is not None and part.get('layered_chains'):
        return 'sfem::helmholtz_chain_kernel<%s, %d, %s, *, true>' % (
            real, P, elem)
      return 'sfem::helmholtz_facet_kernel<%s, %d, %s, true, *, true>' % (
          real, P, elem)
    if (scalar and part.get('chains') is not None and
        switches.get('SFEM_CHAIN') != '0'):
      return 'sfem::helmholtz_chain_kernel<%s, %d, %s, ' % (real, P, elem)
    return 'sfem::helmholtz_facet_kernel<%s, %d, %s, %s, ' % (
        real, P, elem, b(scalar))
  if (real == 'float' and P == 12 and ndim == 3 and scalar and
      geo_mode in (1, 3) and part.get('cluster') is None and
      not part.get('colored') and switches.get('SFEM_MFMA') == '1'):
    return 'sfem::helmholtz_mfma_p12_kernel<%d, %s>' % (geo_mode, b(mass))
  if part.get('cluster') is not None:
    return 'sfem::helmholtz_cluster_kernel<%s, %d, %s, %d, %s>' % (
        real, P, b(scalar), geo_mode, b(mass))
  tpe = P * P if ndim == 3 else P
  can_sort = ndim == 3 and tpe <= 64
  sort = (can_sort and part.get('shared_order') is not None and
          not part.get('colored') and not (geo_mode == 3 and mass))
  b = lambda v: 'true' if v else 'false'
  return 'sfem::helmholtz_kernel<%s, %d, %d, true, %s, %d, %s, %s>' % (
      real, P, ndim, b(scalar), geo_mode, b(sort), b(mass))


def helmholtz_apply(u, out, enc, parts, host, ndim, P, lambda0, lambda1,
                    zero_range, dot_out=None):
  """out <- mask * scatter((l0 B + l1 A)_local(gather(u))).

  `parts`: one dict per geometry kind present in the mesh (see
  `_helmholtz_args`); the shared-node range of `out` is cleared by the first
  launch only.
  """
  dev = _dev(enc)
  _check_vector_layout(u, out)
  host = {k: _host(v, u.dtype) for k, v in host.items()}
  if not parts and zero_range[1] > zero_range[0]:
    out[zero_range[0]:zero_range[1]].zero_()
  with torch.cuda.device(dev):
    for n, part in enumerate(parts):
      args = _helmholtz_args(u, out, part.get('enc', enc), part, host, ndim,
                             P, enc.shape[0], u.shape[0], lambda0, lambda1,
                             zero_range if n == 0 else (0, 0), dot_out)
      _lib.check(_lib.load().sfem_helmholtz_apply(ctypes.byref(args),
                                                  _stream(dev)),
                 'sfem_helmholtz_apply')
  return out


def layered_dot_waves(parts, P, num_elements):
  """Waves each launch of a layered apply starts (its share of a per-wave
  `dot_out`): workgroups (chain segments or elements) x ceil(P^2 / 64)."""
  waves = (P * P + 63) // 64
  out = []
  for part in parts:
    if part.get('layered_chains'):
      groups = part['chains'][0].numel() - 1
    elif 'elem_list' in part:
      groups = part['elem_list'].numel()
    else:
      groups = num_elements
    out.append(groups * waves)
  return out


def helmholtz_apply_layered(u, ext, enc, parts, host, ndim, P, lambda0,
                            lambda1, dot_out=None, per_wave=False):
  """`sfem_helmholtz_apply` with layered assembly: `ext` is the extended
  output [N nodal values | layers] of the operator's layer plan (slots nobody
  writes hold zero), `parts` carry `layered_table`.  Scalar fields; nothing is
  cleared, no atomics are issued.  `per_wave`: `dot_out` has one double per
  wave of all launches (`layered_dot_waves`), stored not accumulated."""
  dev = _dev(enc)
  if u.dim() != 1 or not u.is_contiguous() or not ext.is_contiguous():
    raise ValueError('layered assembly takes contiguous scalar fields')
  if ext.dtype != u.dtype:
    raise ValueError('u and the extended output differ in dtype')
  host = {k: _host(v, u.dtype) for k, v in host.items()}
  waves = (layered_dot_waves(parts, P, enc.shape[0])
           if per_wave and dot_out is not None else None)
  # (the caller's buffer may carry scratch behind the slots)
  if waves is not None and sum(waves) > dot_out.numel():
    raise ValueError(f'{sum(waves)} waves but {dot_out.numel()} dot slots')
  at = 0
  with torch.cuda.device(dev):
    for n, part in enumerate(parts):
      dots = dot_out if waves is None else dot_out[at:at + waves[n]]
      args = _helmholtz_args(u, ext, enc, part, host, ndim, P, enc.shape[0],
                             u.shape[0], lambda0, lambda1, (0, 0), dots,
                             layered_extent=ext.numel(),
                             dot_slots=0 if waves is None else waves[n])
      if waves is not None:
        at += waves[n]
      _lib.check(_lib.load().sfem_helmholtz_apply(ctypes.byref(args),
                                                  _stream(dev)),
                 'sfem_helmholtz_apply')
  return ext


def _layer_arrays(layers):
  """(len, off) int64 ctypes arrays of a layer list [(length, offset), ...]."""
  n = len(layers)
  arr = ctypes.c_int64 * max(n, 1)
  return (arr(*[int(l[0]) for l in layers]) if n else arr(0),
          arr(*[int(l[1]) for l in layers]) if n else arr(0), n)


def _mask_args(masks, n):
  """(device pointer, host offsets) of a `(bytes, offsets)` layer-mask pair."""
  if masks is None:
    return None, None
  data, offs = masks
  arr = ctypes.c_int64 * max(n, 1)
  return _ptr(data), arr(*[int(o) for o in offs])


def cg_update_r_layered(r, ap_ext, layers, scalars, fuse_rr, masks=None):
  """r -= alpha (Ap assembled from its layers) (+ gamma_new += r.r).
  `masks = (uint8 device tensor, per-layer offsets)`: chunks of
  SFEM_LAYER_CHUNK nodes that no element writes are not read."""
  dev = _dev(r, ap_ext, scalars)
  ln, off, n = _layer_arrays(layers)
  mptr, moff = _mask_args(masks, n)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_cg_update_r_layered(
        _ptr(r), _ptr(ap_ext), r.numel(), ln, off, n, mptr, moff,
        _ptr(scalars), int(fuse_rr), _dtype_code(r), _stream(dev)),
        'sfem_cg_update_r_layered')


def cg_update_r_layered_det(r, ap_ext, layers, scalars, rr_partials,
                            masks=None):
  """The same with r.r left as STORED per-workgroup sums in `rr_partials`
  (summed in index order by `cg_scalars_n(..., 8, ...)`): returns how many."""
  dev = _dev(r, ap_ext, scalars, rr_partials)
  ln, off, n = _layer_arrays(layers)
  mptr, moff = _mask_args(masks, n)
  count = ctypes.c_int64(0)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_cg_update_r_layered_det(
        _ptr(r), _ptr(ap_ext), r.numel(), ln, off, n, mptr, moff,
        _ptr(scalars),
        _ptr(rr_partials), rr_partials.numel(), ctypes.byref(count),
        _dtype_code(r), _stream(dev)), 'sfem_cg_update_r_layered_det')
  return count.value


def cg_scalars_n(scalars, phase, maxiter, tol, atol, partials, num_partials):
  """`cg_scalars` over `num_partials` stored partial sums (phases 3, 4, 5, 8)."""
  dev = _dev(scalars, partials)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_cg_scalars_n(
        _ptr(scalars), phase, float(maxiter), float(tol), float(atol),
        _ptr(partials), int(num_partials), _stream(dev)), 'sfem_cg_scalars_n')


def fold_layers_at(ext, idx, count, layers):
  """ext[idx] += its layers there, the folded slots cleared (in place); `idx`:
  distinct node positions (int64 device tensor)."""
  dev = _dev(ext, idx)
  if idx.dtype != torch.int64:
    raise ValueError('fold_layers_at: int64 node positions')
  ln, off, n = _layer_arrays(layers)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_fold_layers_at(
        _ptr(ext), _ptr(idx), idx.numel(), int(count), ln, off, n,
        _dtype_code(ext), _stream(dev)), 'sfem_fold_layers_at')
  return ext


def fdm_solve(r, pel, S, cases, inv_ev, ndim, Pp):
  """z_e = (S (x) ..) [inv_ev_e .* (S (x) ..)^T r_e] for every element
  (`sfem_fdm_solve`); `pel` (E, Pp^d) int64 or None for element-contiguous
  numbering; `cases` (ndim, E) int32."""
  dev = _dev(r, pel, S, cases, inv_ev)
  z = torch.empty_like(r)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_fdm_solve(
        _ptr(r), _ptr(z), _ptr(pel), _ptr(S), _ptr(cases), _ptr(inv_ev),
        cases.shape[1], int(ndim), int(Pp), _dtype_code(r), _stream(dev)),
        'sfem_fdm_solve')
  return z


def fdm_solve_sums(r, pel, S, cases, inv_ev, weights, ndim, Pp):
  """`fdm_solve` that also returns the element sums of r and the elements'
  shares of weights . z (`sfem_fdm_solve_sums`)."""
  dev = _dev(r, pel, S, cases, inv_ev, weights)
  E = cases.shape[1]
  z = torch.empty_like(r)
  elem_sum = torch.empty(E, dtype=r.dtype, device=r.device)
  weighted = torch.empty(E, dtype=r.dtype, device=r.device)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_fdm_solve_sums(
        _ptr(r), _ptr(z), _ptr(pel), _ptr(S), _ptr(cases), _ptr(inv_ev),
        _ptr(weights), _ptr(elem_sum), _ptr(weighted), E, int(ndim), int(Pp),
        _dtype_code(r), _stream(dev)), 'sfem_fdm_solve_sums')
  return z, elem_sum, weighted


def add_element_constants_(z, yc, shift, n, elems_per_member):
  """In place: z[e n + i] += yc[e] - shift[e // elems_per_member]."""
  dev = _dev(z, yc, shift)
  if not z.is_contiguous() or z.numel() != yc.numel() * n:
    raise ValueError('add_element_constants_: z is (E n,) contiguous')
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_add_element_constants(
        _ptr(z), _ptr(yc.contiguous()), _ptr(shift.contiguous()), yc.numel(),
        int(n), int(elems_per_member), _dtype_code(z), _stream(dev)),
        'sfem_add_element_constants')
  return z


def ell_chebyshev(cols, vals, dinv, b, steps, lmin, lmax, work=None):
  """x = Chebyshev polynomial of the Jacobi-scaled ELL matrix applied to b
  (`sfem_ell_chebyshev`); cols / vals (width, n) int32 / real."""
  dev = _dev(cols, vals, dinv, b)
  n = b.numel()
  x = torch.empty_like(b)
  if work is None:
    work = torch.empty(3 * n, dtype=b.dtype, device=b.device)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_ell_chebyshev(
        _ptr(cols), _ptr(vals), _ptr(dinv), _ptr(b), _ptr(x), _ptr(work), n,
        cols.shape[0], int(steps), float(lmin), float(lmax), _dtype_code(b),
        _stream(dev)), 'sfem_ell_chebyshev')
  return x


# ------------------------------------------------------- ensemble CG (vmap)
def _ens_check(members, *vs):
  dev = _dev(*vs)
  n = vs[0].numel()
  if n % members:
    raise ValueError(f'{n} values do not split into {members} members')
  for v in vs:
    if not v.is_contiguous() or v.numel() != n or v.dtype != vs[0].dtype:
      raise ValueError('ensemble vectors: contiguous, same size and dtype '
                       '(member m owns [m len, (m + 1) len))')
  return dev, n // members


def ens_dot(a, b, members, partials, which):
  """partials[m, which, :] = stored partial sums of a_m . b_m."""
  dev, ln = _ens_check(members, a, b)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_ens_dot(
        _ptr(a), _ptr(b), ln, members, _ptr(partials), int(which),
        _dtype_code(a), _stream(dev)), 'sfem_ens_dot')


def ens_init(scalars, partials, members, maxiter, tol, atol):
  dev = _dev(scalars, partials)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_ens_init(
        _ptr(scalars), _ptr(partials), members, float(maxiter), float(tol),
        float(atol), _stream(dev)), 'sfem_ens_init')


def ens_update_r(r, ap, members, scalars, partials):
  dev, ln = _ens_check(members, r, ap)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_ens_update_r(
        _ptr(r), _ptr(ap), ln, members, _ptr(scalars), _ptr(partials),
        _dtype_code(r), _stream(dev)), 'sfem_ens_update_r')


def ens_close(scalars, partials, members, maxiter):
  dev = _dev(scalars, partials)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_ens_close(
        _ptr(scalars), _ptr(partials), members, float(maxiter), _stream(dev)),
        'sfem_ens_close')


def ens_update_xp(x, p, z, members, scalars):
  dev, ln = _ens_check(members, x, p, z)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_ens_update_xp(
        _ptr(x), _ptr(p), _ptr(z), ln, members, _ptr(scalars), _dtype_code(x),
        _stream(dev)), 'sfem_ens_update_xp')


def ens_update_r_mean(r, ap, w, members, scalars, partials, sums):
  dev, ln = _ens_check(members, r, ap)
  if w.numel() != ln or w.dtype != r.dtype or not w.is_contiguous():
    raise ValueError('ens_update_r_mean: weights of one member')
  _dev(w, sums)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_ens_update_r_mean(
        _ptr(r), _ptr(ap), _ptr(w), ln, members, _ptr(scalars), _ptr(partials),
        _ptr(sums), _dtype_code(r), _stream(dev)), 'sfem_ens_update_r_mean')


def ens_close_mean(scalars, partials, sums, total, members, maxiter):
  dev = _dev(scalars, partials, sums)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_ens_close_mean(
        _ptr(scalars), _ptr(partials), _ptr(sums), float(total), members,
        float(maxiter), _stream(dev)), 'sfem_ens_close_mean')


def ens_update_xp_mean(x, p, r, members, scalars):
  dev, ln = _ens_check(members, x, p, r)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_ens_update_xp_mean(
        _ptr(x), _ptr(p), _ptr(r), ln, members, _ptr(scalars), _dtype_code(x),
        _stream(dev)), 'sfem_ens_update_xp_mean')


def ens_subtract_weighted_mean(w, b, total, members, partials, out=None):
  """out_m = w_m - (b . w_m / total) 1 for every member (b: one member)."""
  w, b = w.contiguous(), b.contiguous()
  if out is None:
    out = torch.empty_like(w)
  dev, ln = _ens_check(members, w, out)
  if b.numel() != ln or b.dtype != w.dtype:
    raise ValueError('ens_subtract_weighted_mean: weights of one member')
  _dev(b, partials)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_ens_subtract_weighted_mean(
        _ptr(w), _ptr(b), float(total), _ptr(out), _ptr(partials), ln, members,
        _dtype_code(w), _stream(dev)), 'sfem_ens_subtract_weighted_mean')
  return out


def fold_layers(ext, count, layers):
  """ext[:count] += its layers (in place); returns the view ext[:count]."""
  dev = _dev(ext)
  ln, off, n = _layer_arrays(layers)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_fold_layers(
        _ptr(ext), int(count), ln, off, n, _dtype_code(ext), _stream(dev)),
        'sfem_fold_layers')
  return ext[:count]


def stokes_setup(invjac, jacdet, weights_nd):
  """Weighted cofactor planes (E, d*d, Q) for curved elements."""
  invjac, jacdet = invjac.contiguous(), jacdet.contiguous()
  dev = _dev(invjac, jacdet, weights_nd)
  E, Q, d, _ = invjac.shape
  kfac = torch.empty((E, d * d, Q), dtype=invjac.dtype, device=dev)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_stokes_setup(
        _ptr(invjac), _ptr(jacdet), _ptr(weights_nd), _ptr(kfac), E, d, Q,
        _dtype_code(invjac), _stream(dev)), 'sfem_stokes_setup')
  return kfac


def _stokes_args(vec, enc, penc, part, host, ndim, P, zero_range,
                 shared_order=None, **ptrs):
  node_stride = comp_stride = 0
  if not vec.is_contiguous():
    node_stride, comp_stride = 1, vec.stride(-1)
  lst = part.get('elem_list')
  ft = part.get('facet_table')
  ch = part.get('chains') if ft is not None else None
  if ft is not None:
    ptrs = dict(ptrs, facet_table=_dptr(ft), chain_offsets=_dptr(ch[0]),
                chain_elems=_dptr(ch[1]), num_chains=ch[0].numel() - 1)
    shared_order = None
  return _lib.StokesArgs(
      enc=_dptr(enc), penc=_dptr(penc), kfac=_dptr(part.get('kfac')),
      geo_elem=_dptr(part.get('geo_elem')),
      geo_index=_dptr(part.get('geo_index')), elem_list=_dptr(lst),
      dmat=_hptr(host['dmat']), weights=_hptr(host['weights']),
      nodes=_hptr(host['nodes']), interp=_hptr(host['interp']),
      num_elements=enc.shape[0], num_listed=0 if lst is None else lst.numel(),
      num_nodes=vec.shape[0], zero_begin=int(zero_range[0]),
      zero_end=int(zero_range[1]), ndim=ndim, P=P, dtype=_dtype_code(vec),
      geo_mode=part['geo_mode'], node_stride=node_stride,
      comp_stride=comp_stride, shared_order=_dptr(shared_order),
      shared_stride=0 if shared_order is None else shared_order.shape[1],
      **ptrs)


def _check_field(u, ndim):
  if u.dim() != 2 or u.shape[1] != ndim or not (
      u.is_contiguous() or is_component_major(u)):
    raise ValueError(f'expected a dense (N, {ndim}) velocity field')


def stokes_div(u, p_out, enc, penc, parts, host, ndim, P, scale=None,
               dot_with=None, dot_out=None):
  """p_out <- D(scale * u) (navier_stokes.py:313-333), one launch per
  geometry kind.  `dot_out` (SFEM_DOT_SLOTS doubles) accumulates partial sums
  of `dot_with . p_out`."""
  dev = _dev(enc, p_out)
  _check_field(u, ndim)
  per_node = scale is not None and scale.dim() == 1
  if per_node:
    if scale.shape[0] != u.shape[0] or not scale.is_contiguous():
      raise ValueError('per-node scale must be a contiguous (N,) vector')
  elif scale is not None:
    _check_field(scale, ndim)
    if scale.stride() != u.stride():
      raise ValueError('scale must share the layout of u')
  if scale is not None and scale.dtype != u.dtype:
    raise ValueError('scale must share the dtype of u')
  host = {k: _host(v, u.dtype) for k, v in host.items()}
  with torch.cuda.device(dev):
    for part in parts:
      args = _stokes_args(u, enc, penc, part, host, ndim, P, (0, 0),
                          u=u.data_ptr(), p_out=p_out.data_ptr(),
                          scale=_dptr(scale), scale_per_node=int(per_node),
                          p_in=_dptr(dot_with if dot_out is not None else None),
                          dot_out=_dptr(dot_out))
      _lib.check(_lib.load().sfem_stokes_div(ctypes.byref(args), _stream(dev)),
                 'sfem_stokes_div')
  return p_out


def stokes_grad_t(p, out, enc, penc, parts, host, ndim, P, zero_range,
                  shared_order=None, scale=None):
  """out <- mask * QQ^T-ready (scale * D^T p) (navier_stokes.py:322-338);
  `scale`: (N,) or field-shaped factor applied to every contribution before
  it is assembled (valid for factors equal on all copies of a node)."""
  dev = _dev(enc, p)
  _check_field(out, ndim)
  per_node = _scale_args(scale, out, ndim)
  host = {k: _host(v, out.dtype) for k, v in host.items()}
  with torch.cuda.device(dev):
    for n, part in enumerate(parts):
      args = _stokes_args(out, enc, penc, part, host, ndim, P,
                          zero_range if n == 0 else (0, 0), shared_order,
                          out=out.data_ptr(), p_in=p.data_ptr(),
                          scale=_dptr(scale), scale_per_node=int(per_node))
      _lib.check(_lib.load().sfem_stokes_grad_t(ctypes.byref(args),
                                                _stream(dev)),
                 'sfem_stokes_grad_t')
  return out


def _scale_args(scale, field, ndim):
  per_node = scale is not None and scale.dim() == 1
  if per_node:
    if scale.shape[0] != field.shape[0] or not scale.is_contiguous():
      raise ValueError('per-node scale must be a contiguous (N,) vector')
  elif scale is not None:
    _check_field(scale, ndim)
    if scale.stride() != field.stride():
      raise ValueError('scale must share the layout of the velocity field')
  if scale is not None and scale.dtype != field.dtype:
    raise ValueError('scale must share the dtype of the velocity field')
  return per_node


def stokes_e_first(p, w, p_out, enc, penc, parts, host, ndim, P, zero_range,
                   scale=None, shared_order=None):
  """First half of E = D Q D^T (`sfem_stokes_e_first`): `w` receives D^T p at
  the SHARED nodes only, `p_out` = D(scale * complete part)."""
  dev = _dev(enc, p, p_out)
  _check_field(w, ndim)
  per_node = _scale_args(scale, w, ndim)
  host = {k: _host(v, w.dtype) for k, v in host.items()}
  with torch.cuda.device(dev):
    for n, part in enumerate(parts):
      args = _stokes_args(w, enc, penc, part, host, ndim, P,
                          zero_range if n == 0 else (0, 0), shared_order,
                          out=w.data_ptr(), p_in=p.data_ptr(),
                          p_out=p_out.data_ptr(), scale=_dptr(scale),
                          scale_per_node=int(per_node))
      _lib.check(_lib.load().sfem_stokes_e_first(ctypes.byref(args),
                                                 _stream(dev)),
                 'sfem_stokes_e_first')
  return p_out


def stokes_e_second(w, p_out, enc, penc, parts, host, ndim, P, scale=None):
  """Second half: p_out += D(scale * w restricted to the SHARED nodes)."""
  dev = _dev(enc, p_out)
  _check_field(w, ndim)
  per_node = _scale_args(scale, w, ndim)
  host = {k: _host(v, w.dtype) for k, v in host.items()}
  with torch.cuda.device(dev):
    for part in parts:
      args = _stokes_args(w, enc, penc, part, host, ndim, P, (0, 0),
                          u=w.data_ptr(), p_out=p_out.data_ptr(),
                          scale=_dptr(scale), scale_per_node=int(per_node))
      _lib.check(_lib.load().sfem_stokes_e_second(ctypes.byref(args),
                                                  _stream(dev)),
                 'sfem_stokes_e_second')
  return p_out


def stokes_convect_local(u_local, parts, host, ndim, P):
  """(E, P^d, d) values on a collocated grid -> w detJ (u . grad) u there."""
  u_local = u_local.contiguous()
  dev = _dev(u_local)
  E, n, d = u_local.shape
  if d != ndim or n != P ** ndim:
    raise ValueError(f'expected (E, {P ** ndim}, {ndim}) local values, got '
                     f'{tuple(u_local.shape)}')
  out = torch.empty_like(u_local)
  host = {k: _host(v, u_local.dtype) for k, v in host.items()}
  with torch.cuda.device(dev):
    for part in parts:
      lst = part.get('elem_list')
      args = _lib.StokesArgs(
          u=u_local.data_ptr(), out=out.data_ptr(),
          kfac=_dptr(part.get('kfac')), geo_elem=_dptr(part.get('geo_elem')),
          geo_index=_dptr(part.get('geo_index')), elem_list=_dptr(lst),
          dmat=_hptr(host['dmat']), weights=_hptr(host['weights']),
          nodes=_hptr(host['nodes']), num_elements=E,
          num_listed=0 if lst is None else lst.numel(), ndim=ndim, P=P,
          dtype=_dtype_code(u_local), geo_mode=part['geo_mode'])
      _lib.check(_lib.load().sfem_stokes_convect_local(ctypes.byref(args),
                                                       _stream(dev)),
                 'sfem_stokes_convect_local')
  return out


from swirl_fem_amd.core.layout import is_component_major  # noqa: E402


def _check_vector_layout(u, out):
  for t in (u, out):
    if not t.is_cuda:
      raise RuntimeError(
          'swirl_fem_amd kernels run on MI355X device tensors only; got a '
          f'{t.device} tensor (there is no CPU fallback)')
    if not (t.is_contiguous() or is_component_major(t)):
      raise ValueError('expected a contiguous or component-major tensor')
  if u.stride() != out.stride() or u.shape != out.shape:
    raise ValueError('u and out must share shape and memory layout')


def helmholtz_local(u_local, parts, host, ndim, P, lambda0, lambda1):
  if not (u_local.is_contiguous() or is_component_major(u_local)):
    u_local = u_local.contiguous()
  dev = u_local.device
  host = {k: _host(v, u_local.dtype) for k, v in host.items()}
  out = torch.empty_like(u_local)       # preserves the (dense) layout
  _check_vector_layout(u_local, out)
  with torch.cuda.device(dev):
    for part in parts:
      args = _helmholtz_args(u_local, out, None, part, host, ndim, P,
                             u_local.shape[0], 0, lambda0, lambda1, (0, 0))
      _lib.check(_lib.load().sfem_helmholtz_local(ctypes.byref(args),
                                                  _stream(dev)),
                 'sfem_helmholtz_local')
  return out


# ------------------------------------------------------------------------ CG
def dot(a, b, result, slot, accumulate=False):
  """result[slot] (+)= sum(a * b); result is a float64 device tensor."""
  dev = _dev(a, b, result)
  fn = (_lib.load().sfem_dot_accumulate if accumulate
        else _lib.load().sfem_dot)
  with torch.cuda.device(dev):
    _lib.check(fn(_ptr(a), _ptr(b), a.numel(),
                  ctypes.c_void_p(result.data_ptr() + 8 * slot),
                  _dtype_code(a), _stream(dev)), 'sfem_dot')


def dot_indexed(a, b, idx, w, result, slot, scale):
  """result[slot] += scale * sum_i w[i] <a[idx[i]], b[idx[i]]>."""
  ncomp, ns, cs = _node_view(a)
  if _node_view(b) != (ncomp, ns, cs) or a.dtype != b.dtype:
    raise ValueError('dot_indexed: operands must share layout and dtype')
  flat = lambda t: t.movedim(-1, 0) if is_component_major(t) else t
  dev = _dev(flat(a), flat(b), idx, w, result)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_dot_indexed(
        _ptr(a), _ptr(b), _ptr(idx), _ptr(w), idx.numel(), ncomp, ns, cs,
        float(scale), result.data_ptr() + 8 * slot, _dtype_code(a),
        _stream(dev)), 'sfem_dot_indexed')


def cg_scalars(scalars, phase, maxiter, tol, atol, partials=None):
  dev = _dev(scalars, partials)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_cg_scalars(
        _ptr(scalars), phase, float(maxiter), float(tol), float(atol),
        _ptr(partials), _stream(dev)), 'sfem_cg_scalars')


def cg_update_xr(x, r, p, ap, scalars, fuse_rr):
  dev = _dev(x, r, p, ap, scalars)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_cg_update_xr(
        _ptr(x), _ptr(r), _ptr(p), _ptr(ap), x.numel(), _ptr(scalars),
        int(fuse_rr), _dtype_code(x), _stream(dev)), 'sfem_cg_update_xr')


def cg_update_p(p, z, scalars):
  dev = _dev(p, z, scalars)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_cg_update_p(
        _ptr(p), _ptr(z), p.numel(), _ptr(scalars), _dtype_code(p),
        _stream(dev)), 'sfem_cg_update_p')


def cg_update_r(r, ap, scalars, fuse_rr):
  dev = _dev(r, ap, scalars)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_cg_update_r(
        _ptr(r), _ptr(ap), r.numel(), _ptr(scalars), int(fuse_rr),
        _dtype_code(r), _stream(dev)), 'sfem_cg_update_r')


def cg_update_xp_lazy(x, pring, z, scalars, lazy):
  """p_{k+1} = z + beta p_k into the next slot of the ring `pring` (m, stride);
  x takes its m pending terms every m-th iteration (`sfem_cg_update_xp_lazy`)."""
  dev = _dev(x, pring, z, scalars, lazy)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_cg_update_xp_lazy(
        _ptr(x), _ptr(pring), pring.stride(0), _ptr(z), x.numel(),
        _ptr(scalars), _ptr(lazy), pring.shape[0], _dtype_code(x),
        _stream(dev)), 'sfem_cg_update_xp_lazy')


def cg_flush_x(x, pring, scalars, lazy):
  """Adds the terms of x the lazy update still holds back."""
  dev = _dev(x, pring, scalars, lazy)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_cg_flush_x(
        _ptr(x), _ptr(pring), pring.stride(0), x.numel(), _ptr(scalars),
        _ptr(lazy), pring.shape[0], _dtype_code(x), _stream(dev)),
        'sfem_cg_flush_x')


def cg_update_xp(x, p, z, scalars):
  dev = _dev(x, p, z, scalars)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_cg_update_xp(
        _ptr(x), _ptr(p), _ptr(z), x.numel(), _ptr(scalars), _dtype_code(x),
        _stream(dev)), 'sfem_cg_update_xp')


def cg_update_r_mean(r, ap, w, scalars, sums):
  dev = _dev(r, ap, w, scalars, sums)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_cg_update_r_mean(
        _ptr(r), _ptr(ap), _ptr(w), r.numel(), _ptr(scalars), _ptr(sums),
        _dtype_code(r), _stream(dev)), 'sfem_cg_update_r_mean')


def cg_update_xp_mean(x, p, r, scalars, sums, total):
  dev = _dev(x, p, r, scalars, sums)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_cg_update_xp_mean(
        _ptr(x), _ptr(p), _ptr(r), x.numel(), _ptr(scalars), _ptr(sums),
        float(total), _dtype_code(x), _stream(dev)), 'sfem_cg_update_xp_mean')


def axpby(a, x, b, y):
  """In place: y = a * x + b * y."""
  dev = _dev(x, y)
  with torch.cuda.device(dev):
    _lib.check(_lib.load().sfem_axpby(
        float(a), _ptr(x), float(b), _ptr(y), x.numel(), _dtype_code(x),
        _stream(dev)), 'sfem_axpby')
  return y
