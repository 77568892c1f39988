"""Fused element operators (build-side fast paths behind the reference API).

* `HelmholtzOperator`: the collocated  H = lambda0 * B + lambda1 * A  of the
  reference's callers -- examples/poisson.py:141-154 (A, B),
  navier_stokes/navier_stokes.py:220-236, :295-307, :431 -- as ONE kernel:
  gather, sum-factorised apply, Dirichlet mask and direct-stiffness summation
  (`sfem_helmholtz_apply`).  The geometric factors of affine / multilinear
  elements are evaluated in registers, curved elements read 6 (+1) stored
  factors per point (`classify_geometry`).
* `TwoGridHelmholtzOperator`: the same operator when the quadrature differs
  from the nodes (interpolate -> fused kernel on the quadrature grid ->
  transposed interpolation).
* `StokesDivGrad`: D and D^T of the P_N - P_{N-2} pair (navier_stokes.py:
  313-338), one kernel each.
* `ConvectionOperator`: the over-integrated convection term (:238-245).

Every class has a `supports_*` predicate; callers fall back to the generic
q-function path (`FiniteElementSpace.local_covector`) when it says no.
"""

from __future__ import annotations

import dataclasses
import os

import numpy as np
import torch

from swirl_fem_amd import switches
from swirl_fem_amd import _ops

MAX_FUSED_P = 12


def supports_fused(fespace) -> str | None:
  """None if the fused kernel applies, else the reason it does not."""
  P = fespace.mesh.gridpoints_1d.num_points
  if not fespace.is_collocated:
    return 'quadrature points differ from the nodes'
  if fespace.mesh.ndim not in (2, 3):
    return f'ndim={fespace.mesh.ndim}'
  if not 2 <= P <= MAX_FUSED_P:
    return f'P={P} outside 2..{MAX_FUSED_P}'
  d = fespace.interpolator._differentiation_matrix_1d()
  if not np.allclose(d[::-1, ::-1], -d, rtol=0,
                     atol=1e-12 * max(1.0, np.abs(d).max())):
    return 'node set is not symmetric about 0'
  return None


MULTILINEAR_RTOL = {torch.float64: 1e-12, torch.float32: 2e-6}
AFFINE_RTOL = {torch.float64: 1e-11, torch.float32: 1e-5}


def classify_geometry(fespace):
  """Per element: 0 = curved (stored per-point factors), 1 = affine,
  3 = multilinear image of the reference cube; plus the (E, 24) coefficients.

  An element is multilinear when its nodes coincide with the multilinear
  interpolant of its 2^d corner nodes (every `refine_premesh` mesh); it is
  affine when the bilinear/trilinear coefficients vanish as well.
  """
  mesh = fespace.mesh
  d, P = mesh.ndim, mesh.gridpoints_1d.num_points
  xe = mesh.element_coords()                                   # (E, n, d)
  coef = _ops.helmholtz_setup_multilinear(xe, d, P)            # (E, 24)
  x1 = torch.as_tensor(mesh.gridpoints_1d.node_values, dtype=xe.dtype,
                       device=xe.device)
  grids = torch.meshgrid(*([x1] * d), indexing='ij')
  r = [g.reshape(-1) for g in grids]                           # (n,) each
  if d == 3:
    mono = torch.stack([r[0], r[1], r[2], r[0] * r[1], r[1] * r[2],
                        r[0] * r[2], r[0] * r[1] * r[2]], dim=1)   # (n, 7)
    A = coef[:, :21].reshape(-1, 7, 3)
    lin, nonlin = A[:, :3], A[:, 3:]
  else:
    mono = torch.stack([r[0], r[1], r[0] * r[1]], dim=1)          # (n, 3)
    A = coef[:, :6].reshape(-1, 3, 2)
    lin, nonlin = A[:, :2], A[:, 2:]
  centre = xe.mean(dim=1, keepdim=True) - torch.einsum(
      'nm,emd->ed', mono, A)[:, None, :] / mono.shape[0]
  recon = centre + torch.einsum('nm,emd->end', mono, A)
  size = lin.abs().amax(dim=(1, 2))
  err = (recon - xe).abs().amax(dim=(1, 2))
  # coordinates carry rounding of their own magnitude (matters in fp32)
  noise = 8 * torch.finfo(xe.dtype).eps * xe.abs().amax(dim=(1, 2))
  multi = err <= MULTILINEAR_RTOL[xe.dtype] * size + noise
  affine = multi & (nonlin.abs().amax(dim=(1, 2)) <=
                    AFFINE_RTOL[xe.dtype] * size + noise)
  kind = torch.zeros(xe.shape[0], dtype=torch.int32, device=xe.device)
  kind[multi] = _GEO_MULTILINEAR
  # a handful of affine elements among multilinear ones (a jittered mesh in
  # fp32: 39 of 32768 pass the tolerance) are not worth a launch of their own
  # and an element list for everybody else: the multilinear kernels evaluate
  # them exactly
  n_affine, n_multi = int(affine.sum()), int(multi.sum())
  if n_affine >= 0.02 * max(n_multi, 1) or n_affine == n_multi:
    kind[affine] = _GEO_AFFINE
  return kind, coef


_GEO_POINT, _GEO_AFFINE, _GEO_MULTILINEAR, _GEO_BOX = 0, 1, 3, 5
FACET_P = tuple(range(6, 13))  # orders the facet-table kernels are compiled for
STOKES_FACET_P = (6, 7, 8)     # ... and the facet-table Stokes kernels
BOX_TOL = {torch.float64: 1e-13, torch.float32: 5e-7}


def chain_segment_length(num_elements):
  """Elements per chain segment (= per workgroup of a chain launch): 8 on
  large meshes; shorter when that would leave the 256 CUs x 16 waves of an
  MI355X with fewer than ~4 rounds of workgroups (16^3 elements in chains of
  8 are 512 workgroups: the Taylor-Green step at 16^3 went from 33 to 50 ms).
  `SFEM_CHAIN_LEN` overrides."""
  env = switches.get('SFEM_CHAIN_LEN')
  if env:
    return max(1, int(env))
  return max(1, min(8, num_elements // 16384))


def facet_chains(elements, ids, P, seg_len):
  """Walks the elements `ids` (int64, device) as chains for the chain launches
  of the facet kernels (`sfem_helmholtz_args.chain_offsets`): element y follows
  x when the face a = P-1 of x is the face a = 0 of y, node for node in the
  lane layout, i.e. `elements[x, (P-1) P^2 + t] == elements[y, t]` for all
  t < P^2 (on a refiner mesh: the neighbour across that face, met with the
  same orientation).  Chains are cut into segments of at most `seg_len`
  elements.  Returns (offsets (S + 1,), elems) int32; elements without a
  neighbour of that kind are segments of their own."""
  m = ids.numel()
  dev = ids.device
  n2 = P * P
  first = elements[ids, :n2].to(torch.int64)
  last = elements[ids, (P - 1) * n2:].to(torch.int64)
  gen = torch.Generator(device='cpu').manual_seed(1234)
  wts = torch.randint(1, 1 << 31, (n2,), generator=gen).to(dev)
  hf, hl = (first * wts).sum(1), (last * wts).sum(1)
  hs, order = torch.sort(hf)
  pos = torch.searchsorted(hs, hl).clamp(max=m - 1)
  cand = order[pos]
  me = torch.arange(m, device=dev)
  ok = (hs[pos] == hl) & (cand != me)
  ok &= (first[cand] == last).all(dim=1)
  ok &= (last >= 0).all(dim=1)
  succ = torch.where(ok, cand, torch.full_like(cand, -1))
  # a face has two elements, so successors are distinct; keep it true anyway
  pred = torch.full((m,), -1, dtype=torch.int64, device=dev)
  src = torch.nonzero(succ >= 0).reshape(-1)
  pred[succ[src]] = src
  valid = torch.zeros(m, dtype=torch.bool, device=dev)
  valid[src] = pred[succ[src]] == src
  succ = torch.where(valid, succ, torch.full_like(succ, -1))
  pred.fill_(-1)
  src = torch.nonzero(succ >= 0).reshape(-1)
  pred[succ[src]] = src
  # list ranking by pointer jumping: head and distance to it
  p = torch.where(pred >= 0, pred, me)
  d = (pred >= 0).to(torch.int64)
  for _ in range(max(1, int(m).bit_length())):
    d = d + d[p]
    p = p[p]
  loop = pred[p] >= 0            # closed rings never reach a head
  if bool(loop.any()):           # members of a ring walk alone
    p = torch.where(loop, me, p)
    d = torch.where(loop, torch.zeros_like(d), d)
  key = p * (int(d.max()) + 1 if m else 1) + d
  walk = torch.argsort(key)
  rank = d[walk]
  starts = torch.nonzero(rank % seg_len == 0).reshape(-1)
  offsets = torch.cat([starts, torch.tensor([m], device=dev)])
  return (offsets.to(torch.int32).contiguous(),
          ids[walk].to(torch.int32).contiguous())


def _facet_parts(fespace, parts, mask, multiplicity, coef):
  """`parts` re-expressed on compact connectivity (`sfem_facet_table_build`):
  elements whose index row is 27 affine facet maps (every `refine_premesh`
  mesh, reference core/mesh_refiner.py:143-251) are applied from their
  432-byte table, affine ones with a diagonal metric as boxes; the remaining
  elements keep their index rows.  None if no element qualifies."""
  mesh = fespace.mesh
  E = mesh.num_elements
  tab, ok = _ops.facet_table(mesh.elements, mask, multiplicity,
                             mesh.gridpoints_1d.num_points)
  if not bool(ok.any()):
    return None
  every = torch.arange(E, device=ok.device)
  cst = None
  out = []
  P = mesh.gridpoints_1d.num_points
  seg_len = chain_segment_length(E)

  def add(part, ids, mode, facet):
    if ids.numel() == 0:
      return
    new = {k: v for k, v in part.items() if k != 'elem_list'}
    new['geo_mode'] = mode
    if ids.numel() < E:
      new['elem_list'] = ids.to(torch.int32).contiguous()
    if facet:
      new.pop('shared_order', None)
      new['facet_table'] = tab
      new['geo_const'] = cst
      # chains (one-wave elements): box 0.72 -> 0.58 ms, affine 0.72 -> 0.66,
      # multilinear 0.92 -> 0.87, stored 1.86 -> 1.80 at config 2; elements
      # that span several waves (P >= 9) lose with them (p = 11 fp32 box 1.74
      # vs 1.46 ms)
      # (P >= 9 has chain instantiations for box / affine elements only:
      # FacetElem::CHAINS; the library refuses the others)
      if seg_len > 1 and (P <= 8 or (
          switches.get('SFEM_CHAIN_HI') == '1' and
          mode in (_GEO_BOX, _GEO_AFFINE))):
        new['chains'] = facet_chains(mesh.elements, ids, P, seg_len)
        new['chain_len'] = seg_len
    out.append(new)

  use_box = switches.get('SFEM_BOX') != '0'
  for part in parts:
    ids = every if 'elem_list' not in part else part['elem_list'].long()
    good = ok[ids]
    mode = part['geo_mode']
    if mode == _GEO_MULTILINEAR and P >= 9:
      # 12 nodes per lane plus the multilinear geometry state need 203 VGPRs
      # on the facet kernel (2 waves per SIMD): measured 3.97 vs 2.99 ms for
      # the index-row kernel at p = 11 fp32, which therefore keeps them
      good = torch.zeros_like(good)
    if mode == _GEO_AFFINE:
      if cst is None:
        cst = _ops.helmholtz_setup_affine(coef, BOX_TOL[fespace.dtype])
      box = (cst[ids, 7] != 0) & good if use_box else torch.zeros_like(good)
      add(part, ids[box], _GEO_BOX, True)
      add(part, ids[good & ~box], _GEO_AFFINE, True)
    else:
      add(part, ids[good], mode, True)
    add(part, ids[~good], mode, False)
  return out


@dataclasses.dataclass(eq=False)
class LayerPlan:
  """Layered assembly of an operator's facet launches (`build_layer_plan`).

  `layers[k] = (length, offset)`: layer k + 1 of the extended output
  [N nodal values | layer 1 | layer 2 | ...] covers the nodes [0, length) and
  starts at element `offset`; `extent` = size of the extended vector;
  `parts` = the launches with their `layered_table`; `written` = number of
  slots the launches store per apply (layer 0 included)."""
  layers: list
  extent: int
  parts: list
  written: int
  num_nodes: int
  # (uint8 device tensor, per-layer offsets): byte c of a layer = 1 iff some
  # element writes into its nodes [512 c, 512 (c + 1)); the consumers skip the
  # other chunks (zeros).  `read` = layer values they then read per pass.
  masks: tuple | None = None
  read: int = 0


def build_layer_plan(facet_parts, num_elements, num_nodes, P):
  """Gives every (element, facet) that writes a facet of the mesh a LAYER of
  its own for it, so that the direct-stiffness sum (reference
  core/gather_scatter.py:130-133) needs neither atomics nor a cleared range:
  the kernels store every result plainly, the consumer adds the layers of a
  node in layer order (`sfem_cg_update_r_layered`, `sfem_fold_layers`).

  Writers of a facet: every element that holds it, except -- inside a chain
  segment -- the element that hands its last face (and that face's edges and
  vertices) on to its successor, which stores the sum.  Layer = rank of the
  writer among the writers of the facet (0 = the nodal vector itself).  The
  refiner numbers vertices, then edge, face and element interiors, so the
  nodes with k + 1 or more writers are a prefix [0, n_k): layer k is an array
  of n_k values (config 2: n_1 = 37 % of N, n_2 = n_3 = 5 %, n_4.. = 0.3 %).
  Slots nobody writes (a facet with fewer writers than the layer's other
  facets) stay zero from the allocation.

  Returns None when the launches do not cover every element from a facet table
  (index-row elements accumulate with atomics), when a stride does not fit
  the 16 bits of the layered table, or when a facet has more than
  SFEM_MAX_LAYERS + 1 writers."""
  from swirl_fem_amd import _lib
  if not facet_parts or any('facet_table' not in q for q in facet_parts):
    return None
  tab = facet_parts[0]['facet_table']
  dev = tab.device
  E, N = num_elements, num_nodes
  use_chains = switches.get('SFEM_CHAIN') != '0'
  has_succ = torch.zeros(E, dtype=torch.bool, device=dev)
  covered = torch.zeros(E, dtype=torch.int32, device=dev)
  for q in facet_parts:
    if q['facet_table'] is not tab:
      return None
    if 'elem_list' in q:
      covered[q['elem_list'].long()] += 1
    else:
      covered += 1
    if 'chains' in q and use_chains:
      off, elems = q['chains']
      succ = torch.ones(elems.numel(), dtype=torch.bool, device=dev)
      succ[off[1:].long() - 1] = False           # last element of a segment
      has_succ[elems.long()] = succ
  if not bool((covered == 1).all()):
    return None
  t = tab.to(torch.int64)                                    # (E, 27, 4)
  code = t[..., 0] & 0xFFFFFFFF
  id0, dflag = code & 0x3FFFFFFF, code & 0x80000000
  strides = t[..., 1:]                                       # (E, 27, 3)
  if int(strides[..., 1:].abs().max()) >= 32768:
    return None
  f = torch.arange(27, device=dev)
  inner = torch.stack([f // 9 == 1, (f // 3) % 3 == 1, f % 3 == 1], dim=1)
  span = strides * (P - 3) * inner[None].to(torch.int64)     # (index - 1) max
  lo = id0 + span.clamp(max=0).sum(-1)                       # smallest node id
  hi = id0 + span.clamp(min=0).sum(-1) + 1                   # largest + 1
  count = torch.where(inner, P - 2, 1).prod(dim=1)           # nodes per facet
  writer = torch.ones((E, 27), dtype=torch.bool, device=dev)
  writer[:, 18:] &= ~has_succ[:, None]                       # a-class LAST
  idx = torch.nonzero(writer.reshape(-1)).reshape(-1)
  key, perm = torch.sort(lo.reshape(-1)[idx], stable=True)
  first = torch.ones_like(key, dtype=torch.bool)
  first[1:] = key[1:] != key[:-1]
  pos = torch.arange(key.numel(), device=dev)
  start = torch.cummax(torch.where(first, pos, torch.zeros_like(pos)), 0).values
  layer = torch.zeros(E * 27, dtype=torch.int64, device=dev)
  layer[idx[perm]] = pos - start
  nl = int(layer.max())
  if nl > _lib.SFEM_MAX_LAYERS:
    return None
  hi_w = torch.where(writer, hi, torch.zeros_like(hi)).reshape(-1)
  lens = [int(hi_w[layer == k].max()) for k in range(1, nl + 1)]
  for k in range(nl - 2, -1, -1):
    lens[k] = max(lens[k], lens[k + 1])
  pad = lambda v: (v + 3) // 4 * 4                 # 16 bytes in fp32 and fp64
  lens = [pad(v) for v in lens]
  offs, at = [], pad(N)
  for v in lens:
    offs.append(at)
    at += v
  extent = at
  if extent > 0x3FFFFFFF:
    return None
  base = torch.tensor([0] + offs, dtype=torch.int64, device=dev)
  pos_out = (base[layer].reshape(E, 27) + id0) | dflag
  packed = (strides[..., 1] & 0xFFFF) | ((strides[..., 2] & 0xFFFF) << 16)
  wrap = lambda v: ((v + (1 << 31)) % (1 << 32) - (1 << 31)).to(torch.int32)
  tab2 = torch.stack([t[..., 0].to(torch.int32), wrap(pos_out),
                      strides[..., 0].to(torch.int32), wrap(packed)],
                     dim=-1).contiguous()
  written = int((writer.to(torch.int64) * count[None]).sum())
  parts = [dict(q, layered_table=tab2, layered_chains=use_chains and
                'chains' in q) for q in facet_parts]
  # which chunks of each layer are written at all (only shared facets reach a
  # layer beyond the nodal vector: at most (P - 2)^2 <= 100 nodes, so a block
  # touches the chunks of its two ends and no other)
  chunk = _lib.SFEM_LAYER_CHUNK
  flat_layer, flat_w = layer, writer.reshape(-1)
  lo_f, hi_f = lo.reshape(-1), hi.reshape(-1)
  bytes_, moffs, read = [], [], 0
  for k, ln in enumerate(lens, start=1):
    m = torch.zeros((ln + chunk - 1) // chunk, dtype=torch.uint8, device=dev)
    sel = flat_w & (flat_layer == k)
    m[lo_f[sel] // chunk] = 1
    m[(hi_f[sel] - 1) // chunk] = 1
    moffs.append(sum(b.numel() for b in bytes_))
    bytes_.append(m)
    read += min(int(m.sum()) * chunk, ln)
  masks = (torch.cat(bytes_), moffs) if bytes_ else None
  return LayerPlan(layers=list(zip(lens, offs)), extent=extent, parts=parts,
                   written=written, num_nodes=N, masks=masks, read=read)


def _cluster_limits(fespace):
  """(cluster_size, max_shared) if the cluster kernels cover this space."""
  mesh = fespace.mesh
  if mesh.ndim != 3:
    return None
  return _ops.helmholtz_cluster_limits(mesh.gridpoints_1d.num_points,
                                       fespace.dtype)


def _attach_clusters(fespace, enc, multiplicity, parts):
  """`parts` with a `ClusterPlan` each (`core/clusters.py`): every launch then
  sums the shared nodes of 8 neighbouring elements in LDS."""
  from swirl_fem_amd.core import clusters
  size, kmax = _cluster_limits(fespace)
  ids = [None if 'elem_list' not in p else p['elem_list'].to(torch.int64)
         for p in parts]
  plans = clusters.build_cluster_plan(fespace.mesh, enc, multiplicity, ids,
                                      size, kmax)
  out = []
  for part, plan in zip(parts, plans):
    if plan is not None:
      out.append(dict({k: v for k, v in part.items() if k != 'shared_order'},
                      cluster=plan))
  return out


@dataclasses.dataclass(eq=False)
class HelmholtzOperator:
  fespace: object
  parts: list                    # one launch description per geometry kind
  enc: torch.Tensor              # (E, n) encoded indices
  host: dict                     # dmat (P,P), weights (P,), nodes (P,) NumPy
  zero_range: tuple
  num_affine: int = 0
  num_multilinear: int = 0
  num_curved: int = 0
  # the same launches on compact connectivity (scalar / component-major
  # fields), or None: see `_facet_parts`
  facet_parts: list | None = None
  _vector_parts: list | None = None   # launches of vector fields (`_parts_for`)
  _layer_plan: object = None          # LayerPlan, False = none (`layer_plan`)

  @classmethod
  def create(cls, fespace, dirichlet_mask=None, geometry='auto',
             assembly='auto') -> 'HelmholtzOperator':
    """geometry: 'auto' (per element: affine / multilinear / stored factors),
    'multilinear' (no affine shortcut) or 'stored' (6 factors per point for
    every element, the general-geometry path).

    assembly (direct-stiffness summation of the shared nodes):
    'atomic': one HBM atomic per shared slot, issued in ascending node order
    (the fastest measured: DESIGN 3.1); 'cluster': clusters of 8 elements
    summed in LDS, HBM atomics only on the cluster surfaces (3D, P = 4..8;
    `core/clusters.py`: half the atomic traffic, but 1.04 vs 0.78 ms at
    config 2 -- the waves of a cluster wait for each other); 'colored': one
    launch per conflict-free colour class, no atomics, bitwise reproducible;
    'auto': 'atomic' (or 'cluster' with SFEM_CLUSTER=1 in the environment)."""
    why = supports_fused(fespace)
    if why is not None:
      raise NotImplementedError(f'fused Helmholtz kernel unavailable: {why}')
    if geometry not in ('auto', 'multilinear', 'stored'):
      raise ValueError(f'unknown geometry mode {geometry!r}')
    if assembly not in ('auto', 'cluster', 'atomic', 'colored'):
      raise ValueError(f'unknown assembly mode {assembly!r}')
    requested = assembly
    if assembly == 'auto':
      assembly = ('cluster' if _cluster_limits(fespace) is not None and
                  switches.get('SFEM_CLUSTER') == '1' else 'atomic')
    elif assembly == 'cluster' and _cluster_limits(fespace) is None:
      raise NotImplementedError('cluster assembly needs ndim = 3, P = 4..8')
    mesh = fespace.mesh
    E = mesh.num_elements
    w = torch.as_tensor(fespace.quadrature.weights_nd(mesh.ndim),
                        dtype=fespace.dtype, device=fespace.device)
    if geometry == 'stored':
      kind = torch.zeros(E, dtype=torch.int32, device=fespace.device)
      coef = None
    else:
      kind, coef = classify_geometry(fespace)
      if geometry == 'multilinear':
        kind = torch.where(kind == _GEO_AFFINE,
                           torch.full_like(kind, _GEO_MULTILINEAR), kind)
    counts = {k: int((kind == k).sum()) for k in
              (_GEO_POINT, _GEO_AFFINE, _GEO_MULTILINEAR)}
    parts = []
    for k in (_GEO_AFFINE, _GEO_MULTILINEAR, _GEO_POINT):
      if counts[k] == 0:
        continue
      part = {'geo_mode': k}
      sel = kind == k
      if counts[k] < E:
        part['elem_list'] = torch.nonzero(sel).reshape(-1).to(
            torch.int32).contiguous()
      if k == _GEO_POINT:
        if counts[k] < E:
          part['geo'] = _ops.helmholtz_setup(
              fespace.invjacs[sel].contiguous(),
              fespace.jacdets[sel].contiguous(), w)
          part['geo_index'] = (torch.cumsum(sel, 0) - 1).to(
              torch.int32).contiguous()
        else:
          part['geo'] = _ops.helmholtz_setup(fespace.invjacs, fespace.jacdets,
                                             w)
      else:
        part['geo_elem'] = coef
      parts.append(part)
    plan = mesh.assembly_plan()
    mask = None
    if dirichlet_mask is not None:
      mask = torch.as_tensor(dirichlet_mask, device=fespace.device)
      mask = (mask != 0).to(torch.uint8).contiguous()
    enc = _ops.encode_elements(mesh.elements, mask, plan.multiplicity)
    zero_range = plan.zero_range
    if assembly == 'colored':
      # One launch per (geometry kind, colour class): elements of a class share
      # no node, shared slots read-modify-write `out` in colour order (no
      # atomics, no zero-fill of the shared range, bitwise reproducible).
      colors, num_colors, first = plan.coloring()
      slot_shared = ((mesh.elements >= 0) & ~first).to(torch.uint8).contiguous()
      enc = _ops.encode_elements(mesh.elements, mask, None, slot_shared)
      colored_parts = []
      for part in parts:
        in_part = (torch.ones(E, dtype=torch.bool, device=fespace.device)
                   if 'elem_list' not in part else None)
        if in_part is None:
          in_part = torch.zeros(E, dtype=torch.bool, device=fespace.device)
          in_part[part['elem_list'].to(torch.int64)] = True
        for c in range(num_colors):
          lst = torch.nonzero(in_part & (colors == c)).reshape(-1)
          if lst.numel():
            colored_parts.append(dict(
                part, elem_list=lst.to(torch.int32).contiguous(),
                colored=True))
      parts = colored_parts
      unref = torch.nonzero(plan.multiplicity == 0).reshape(-1)
      zero_range = ((int(unref.min()), int(unref.max()) + 1)
                    if unref.numel() else (0, 0))
    if assembly == 'cluster':
      from swirl_fem_amd.core import clusters
      why = clusters.supports_clusters(mesh, enc)
      if why is None:
        parts = _attach_clusters(fespace, enc, plan.multiplicity, parts)
      elif requested == 'cluster':
        raise NotImplementedError(f'cluster assembly unavailable: {why}')
      else:
        assembly = 'atomic'

    if (assembly == 'atomic' and mesh.ndim == 3 and
        mesh.gridpoints_1d.num_points <= 8 and      # one wave per element
        switches.get('SFEM_SORTED_SCATTER') != '0'):
      # 3D: most slots of an element are shared; issue their atomics in node
      # order (better coalesced, see the kernel)
      so = shared_slot_order(enc)
      if so is not None:
        parts = [dict(part, shared_order=so) for part in parts]
    facet_parts = None
    if (assembly == 'atomic' and mesh.ndim == 3 and
        mesh.gridpoints_1d.num_points in FACET_P and
        switches.get('SFEM_FACET') != '0'):
      facet_parts = _facet_parts(fespace, parts, mask, plan.multiplicity, coef)
    host = {'dmat': fespace.interpolator._differentiation_matrix_1d(),
            'weights': np.asarray(fespace.quadrature.weights),
            'nodes': np.asarray(mesh.gridpoints_1d.node_values)}
    return cls(fespace=fespace, parts=parts, enc=enc, host=host,
               zero_range=zero_range, num_affine=counts[_GEO_AFFINE],
               num_multilinear=counts[_GEO_MULTILINEAR],
               num_curved=counts[_GEO_POINT], facet_parts=facet_parts)

  def split(self, element_mask):
    """Two operators over the elements inside / outside `element_mask` (E,)
    that together equal this one; they share all device data.  Used to apply
    the partition-boundary elements first so that the interface exchange
    overlaps with the interior elements (`distributed/solver.py`)."""
    if any(p.get('colored') for p in self.parts):
      raise NotImplementedError('split() of a coloured operator')
    mask = torch.as_tensor(element_mask, device=self.enc.device).to(torch.bool)
    E = self.enc.shape[0]
    if mask.shape != (E,):
      raise ValueError(f'expected an ({E},) element mask')
    clustered = any(p.get('cluster') is not None for p in self.parts)

    def restrict(part_list, keep):
      parts = []
      for part in part_list:
        part = {k: v for k, v in part.items() if k != 'cluster'}
        if 'elem_list' in part:
          lst = part['elem_list']
          lst = lst[keep[lst.to(torch.int64)]]
        else:
          lst = torch.nonzero(keep).reshape(-1).to(torch.int32)
        if lst.numel():
          new = dict(part, elem_list=lst.contiguous())
          if 'chains' in new:
            new['chains'] = facet_chains(
                self.fespace.mesh.elements, lst.to(torch.int64),
                self.fespace.mesh.gridpoints_1d.num_points, new['chain_len'])
          parts.append(new)
      return parts

    halves = []
    for keep in (mask, ~mask):
      parts = restrict(self.parts, keep)
      if clustered:     # each half clusters its own elements
        parts = _attach_clusters(
            self.fespace, self.enc,
            self.fespace.mesh.assembly_plan().multiplicity, parts)
      facet = (None if self.facet_parts is None
               else restrict(self.facet_parts, keep))
      # (each half covers part of the mesh only: no layer plan)
      halves.append(dataclasses.replace(self, parts=parts, facet_parts=facet,
                                        _vector_parts=None, _layer_plan=False))
    return tuple(halves)

  def apply(self, u, lambda0=0.0, lambda1=1.0, out=None, *, zero=True,
            dot_out=None):
    """u (N,) or (N, nc) -> mask * scatter((l0 B + l1 A)_local(gather(u))).

    `zero=False` skips clearing the shared-node range of `out` (the caller has
    cleared it; used by bench.py to time the kernel alone).  `dot_out`: a
    device tensor of `_lib.SFEM_DOT_SLOTS` doubles that accumulates partial
    sums of `u . out` (CG's p.Ap for free inside the scatter stage).
    """
    mesh = self.fespace.mesh
    if u.shape[0] != mesh.num_nodes:
      raise ValueError(f'expected {mesh.num_nodes} nodal values, got '
                       f'{tuple(u.shape)}')
    u = u.to(self.fespace.dtype)
    if not (u.is_contiguous() or _ops.is_component_major(u)):
      u = u.contiguous()
    if out is None:
      out = torch.empty_like(u)        # same (dense) memory layout as u
    return _ops.helmholtz_apply(
        u, out, self.enc, self._parts_for(u), self.host, mesh.ndim,
        mesh.gridpoints_1d.num_points, lambda0, lambda1,
        self.zero_range if zero else (0, 0), dot_out)

  def layer_plan(self):
    """The `LayerPlan` of this operator's facet launches (made on first use),
    or None: see `build_layer_plan`; `SFEM_LAYERED=0` switches it off."""
    if self._layer_plan is None:
      plan = None
      if (self.facet_parts is not None and
          switches.get('SFEM_LAYERED') != '0'):
        mesh = self.fespace.mesh
        plan = build_layer_plan(self.facet_parts, mesh.num_elements,
                                mesh.num_nodes, mesh.gridpoints_1d.num_points)
      self._layer_plan = plan if plan is not None else False
    return self._layer_plan or None

  def new_extended(self):
    """A zeroed extended output for `apply_layered` (slots that no element
    writes must stay zero: keep one such buffer per consumer)."""
    plan = self.layer_plan()
    return torch.zeros(plan.extent, dtype=self.fespace.dtype,
                       device=self.enc.device)

  def apply_layered(self, u, ext, lambda0=0.0, lambda1=1.0, *, dot_out=None,
                    per_wave=False):
    """`apply` for a scalar field with layered assembly: the unassembled
    contributions go to `ext` (from `new_extended`) as plain stores; the
    assembled value of node i is `ext[i]` plus its layers
    (`_ops.fold_layers(ext, N, plan.layers)`, or inside the consumer:
    `_ops.cg_update_r_layered`).  Dirichlet rows are zero in every layer.
    `per_wave`: `dot_out` has `layered_dot_slots()` doubles and every wave
    stores its share of u . out there (reproducible sum) instead of adding it
    to one of SFEM_DOT_SLOTS slots atomically."""
    plan = self.layer_plan()
    if plan is None:
      raise NotImplementedError('this operator has no layer plan')
    mesh = self.fespace.mesh
    if tuple(u.shape) != (mesh.num_nodes,):
      raise ValueError(f'expected ({mesh.num_nodes},) nodal values, got '
                       f'{tuple(u.shape)}')
    if tuple(ext.shape) != (plan.extent,):
      raise ValueError(f'expected an extended output of {plan.extent} values')
    return _ops.helmholtz_apply_layered(
        u.to(self.fespace.dtype).contiguous(), ext, self.enc, plan.parts,
        self.host, mesh.ndim, mesh.gridpoints_1d.num_points, lambda0, lambda1,
        dot_out, per_wave)

  def layered_dot_slots(self):
    """Waves of one `apply_layered` (size of a per-wave `dot_out`)."""
    mesh = self.fespace.mesh
    return sum(_ops.layered_dot_waves(self.layer_plan().parts,
                                      mesh.gridpoints_1d.num_points,
                                      mesh.num_elements))

  def _parts_for(self, u):
    """Facet-table launches for scalar / component-major fields (the kernels
    address node n of component k at n + k * stride), index rows otherwise."""
    if self.facet_parts is None or not (
        u.dim() == 1 or u.shape[-1] == 1 or _ops.is_component_major(u)):
      return self.parts
    if u.dim() == 1 or u.shape[-1] == 1:
      return self.facet_parts
    # Vector fields walk the chains component by component: the geometry is
    # evaluated once per component, which is nothing for box / affine /
    # multilinear elements (48^3, 3 components: 0.77 / 1.21 ms against 1.47 /
    # 2.63 on index rows) but re-reads the stored factors of curved elements
    # three times (2.57 against 2.10 ms): those stay on the index rows in
    # fp64 and on the one-element facet kernel in fp32 (1.14 against 1.23 /
    # 1.32 ms) -- scripts/sweep_vector_fields.py.
    if self._vector_parts is None:
      if not any(q['geo_mode'] == _GEO_POINT and 'facet_table' in q
                 for q in self.facet_parts):
        self._vector_parts = self.facet_parts
      elif self.fespace.dtype == torch.float64:
        self._vector_parts = (
            [q for q in self.facet_parts if q['geo_mode'] != _GEO_POINT] +
            [q for q in self.parts if q['geo_mode'] == _GEO_POINT])
      else:
        self._vector_parts = [
            {k: v for k, v in q.items() if k != 'chains'}
            if q['geo_mode'] == _GEO_POINT else q for q in self.facet_parts]
    return self._vector_parts

  def apply_local(self, u_local, lambda0=0.0, lambda1=1.0):
    """Element-local action (E, n[, nc]) -> (E, n[, nc]); no gather/scatter."""
    mesh = self.fespace.mesh
    return _ops.helmholtz_local(
        u_local.to(self.fespace.dtype), self.parts, self.host, mesh.ndim,
        mesh.gridpoints_1d.num_points, lambda0, lambda1)

  def linear_operator(self, lambda0=0.0, lambda1=1.0):
    """`u -> apply(u, lambda0, lambda1)` as an object that `cg` recognises:
    it also offers `apply_with_dot(u, partials)` (fused p.Ap)."""
    return FusedLinearOperator(self, lambda0, lambda1)

  def bytes_per_apply(self, lambda0=0.0, ncomp=1, layered=False):
    """Bytes one `apply` HAS to move, launch by launch (what the roofline of
    bench.py divides by): the field in and out once (`2 s N ncomp`), the
    connectivity each launch reads -- 432 bytes per element from a facet
    table (+ 4 per element of a chain list), or `4 n` per element of index
    rows plus `2 S` of sorted shared slots -- and the geometry it reads: 64
    bytes (affine / box constants) or `24 s` (multilinear coefficients) per
    element, `(6 or 7) s n` for elements with stored factors.  `layered`
    (`apply_layered`): the field in once, and every slot the launches store
    (`LayerPlan.written`: the nodal values plus the further layers of shared
    facets) instead of the field out once."""
    mesh = self.fespace.mesh
    E, n = mesh.elements.shape
    s = 8 if self.fespace.dtype == torch.float64 else 4
    total = 2 * s * mesh.num_nodes * ncomp
    parts = self.facet_parts if self.facet_parts is not None else self.parts
    if layered:
      plan = self.layer_plan()
      total = s * (mesh.num_nodes + plan.written)
      parts = plan.parts
    for part in parts:
      count = part['elem_list'].numel() if 'elem_list' in part else E
      mode = part['geo_mode']
      if 'facet_table' in part:
        conn = 27 * 16 + (4 if 'chains' in part and ncomp == 1 else 0)
      else:
        conn = 4 * n + (4 if 'elem_list' in part else 0)
        if part.get('shared_order') is not None:
          conn += 2 * part['shared_order'].shape[1]
      if mode == _GEO_POINT:
        geo = (7 if lambda0 else 6) * s * n
      elif 'facet_table' in part and mode in (_GEO_AFFINE, _GEO_BOX):
        geo = 8 * s
      else:
        geo = 24 * s
      total += count * (conn + geo)
    return total

  def kernel_name(self, lambda0=0.0, lambda1=1.0, ncomp=1, layered=False):
    """Name(s) of the kernel instantiation(s) `apply` (`apply_layered`)
    launches, as they appear in a rocprofv3 kernel trace (one per geometry
    kind present)."""
    mesh = self.fespace.mesh
    real = 'double' if self.fespace.dtype == torch.float64 else 'float'
    P = mesh.gridpoints_1d.num_points
    names = []
    parts = (self.layer_plan().parts if layered else
             self.parts if self.facet_parts is None else self.facet_parts)
    for part in parts:
      gm = part['geo_mode']
      names.append(_ops.helmholtz_kernel_name(
          real, P, mesh.ndim, ncomp == 1, gm, part, lambda0 != 0, layered))
    return ' + '.join(sorted(set(names)))


class FusedLinearOperator:
  """Callable operator with a fused `u . A(u)` for `linalg.cg.cg`."""

  def __init__(self, op, lambda0, lambda1):
    self.op, self.lambda0, self.lambda1 = op, lambda0, lambda1
    self._ext = None

  def __call__(self, u):
    return self.op.apply(u, self.lambda0, self.lambda1)

  def apply_with_dot(self, u, partials):
    """Returns A(u) and accumulates partial sums of u . A(u)."""
    return self.op.apply(u, self.lambda0, self.lambda1, dot_out=partials)

  def layer_plan(self):
    """The operator's layer plan if a Krylov iteration gains from it: the
    apply loses its atomics and the clearing of the shared range, the
    `r -= alpha Ap` that adds the layers up reads 0.5 GB more at config 2
    (`scripts/time_layered.py`, ms per CG iteration atomic -> layered: box
    1.556 -> 1.523, multilinear 1.769 -> 1.751, p = 11 fp32 3.249 -> 3.188;
    elements that stream stored factors 2.710 -> 2.738: those keep the
    atomics unless SFEM_LAYERED=force)."""
    plan = self.op.layer_plan()
    if plan is not None and switches.get('SFEM_LAYERED') != 'force' and any(
        q['geo_mode'] == _GEO_POINT for q in plan.parts):
      return None
    return plan

  def apply_layered_with_dot(self, u, partials, per_wave=False):
    """A(u) in layered form (an extended vector owned by this object, see
    `HelmholtzOperator.apply_layered`) + partial sums of u . A(u): for
    consumers that add the layers up themselves (`linalg.cg`)."""
    if self._ext is None:
      self._ext = self.op.new_extended()
    return self.op.apply_layered(u, self._ext, self.lambda0, self.lambda1,
                                 dot_out=partials, per_wave=per_wave)

  def layered_dot_slots(self):
    return self.op.layered_dot_slots()


# ---------------------------------------------------------------------------
# Fused Stokes divergence / pressure gradient (P_N - P_{N-2})
# ---------------------------------------------------------------------------
def supports_fused_stokes(vspace, pspace) -> str | None:
  """None if `sfem_stokes_div` / `sfem_stokes_grad_t` apply, else the reason."""
  why = supports_fused(vspace)
  if why is not None:
    return why
  P = vspace.mesh.gridpoints_1d.num_points
  if P < 4:
    return f'P={P} < 4'
  if pspace.mesh.gridpoints_1d.num_points != P - 2:
    return 'pressure space is not P - 2 points per direction'
  if pspace.interpolator.evalpoints_1d != vspace.mesh.gridpoints_1d:
    return 'pressure quadrature differs from the velocity nodes'
  if pspace.mesh.num_elements != vspace.mesh.num_elements:
    return 'velocity and pressure meshes have different elements'
  # the kernels take w detJ at the quadrature points from the velocity
  # geometry; the reference takes it from the pressure space (:313-320).  The
  # two agree whenever both meshes refine one premesh; checked on a sample of
  # elements (the full factor arrays are never needed by the fused kernels).
  E = vspace.mesh.num_elements
  sample = torch.unique(torch.linspace(
      0, max(E - 1, 0), steps=min(E, 512), device=vspace.device).round().long())
  jv, jp = _sample_jacdets(vspace, sample), _sample_jacdets(pspace, sample)
  # what rounding alone does to det J: the nodal coordinates carry eps |x|,
  # differentiating them over an element of size h with P points amplifies
  # that by ~ P^2 |x| / h (fp32, 40 elements across a unit cube, P = 8: 1.5e-4)
  ndim = vspace.mesh.ndim
  if jv.numel():
    h = 2.0 * float(jv.abs().amax()) ** (1.0 / ndim)
    xmax = float(vspace.mesh.node_coords.abs().amax())
    P = vspace.mesh.gridpoints_1d.num_points
    rounding = 32 * torch.finfo(jv.dtype).eps * P * P * max(xmax / h, 1.0)
  else:
    rounding = 0.0
  tight = 1e-10 if jv.dtype == torch.float64 else 1e-4
  # ... which excuses only elements that ARE images of one premesh in both
  # spaces (affine / multilinear: exact up to that rounding); elements with
  # nodes of their own (curved) must agree tightly, or a pressure geometry that
  # really differs would slip through on fine fp32 meshes
  kind = _cached_geometry_kind(vspace)[sample]
  tol = torch.where(kind != _GEO_POINT, max(tight, rounding), tight).to(
      jv.dtype)
  if jv.shape != jp.shape or not bool(
      ((jv - jp).abs() <= tol[:, None] * jv.abs().amax()).all()):
    return 'velocity and pressure spaces carry different geometry'
  return None


def _cached_geometry_kind(fespace):
  """`classify_geometry(fespace)[0]`, kept in the space's cache."""
  cache = fespace._cache
  if 'geometry_kind' not in cache:
    cache['geometry_kind'] = classify_geometry(fespace)[0]
  return cache['geometry_kind']


def _sample_jacdets(space, elements):
  """det J at the quadrature points of the listed elements."""
  mesh = space.mesh
  i1, g1 = space._matrices()
  xe = _ops.gather_rows(mesh.node_coords,
                        mesh.elements[elements].contiguous())
  _, jacdets, _ = _ops.geom_factors(
      xe, i1, g1, mesh.ndim, mesh.gridpoints_1d.num_points,
      space.quadrature.num_points, want_quad_coords=False)
  return jacdets


def shared_slot_order(enc):
  """Per element, the slots of its SHARED non-Dirichlet nodes in ascending
  node order: `(E, S)` int16 (bit pattern of uint16, 0xFFFF padded), S = the
  largest count.  The assembled kernels issue their atomics in this order
  (`sfem_helmholtz_args.shared_order`); None when no slot is shared."""
  E, n = enc.shape
  chunks, smax = [], 0
  step = max(1, (1 << 25) // n)
  for lo in range(0, E, step):
    e64 = enc[lo:lo + step].to(torch.int64)
    ids = e64 & 0x3FFFFFFF
    take = ((e64 & (1 << 30)) != 0) & (e64 >= 0) & (ids != 0x3FFFFFFF)
    key = torch.where(take, ids, torch.full_like(ids, 1 << 40))
    order = torch.argsort(key, dim=1, stable=True)
    count = take.sum(dim=1)
    smax = max(smax, int(count.max()) if count.numel() else 0)
    chunks.append((order, count))
  if smax == 0:
    return None
  out = torch.empty((E, smax), dtype=torch.int16, device=enc.device)
  col = torch.arange(smax, device=enc.device)[None, :]
  at = 0
  for order, count in chunks:
    tab = order[:, :smax].to(torch.int32)
    tab = torch.where(col < count[:, None], tab, torch.full_like(tab, 0xFFFF))
    out[at:at + len(tab)] = tab.to(torch.int16)      # 0xFFFF wraps to -1
    at += len(tab)
  return out.contiguous()


@dataclasses.dataclass(eq=False)
class StokesDivGrad:
  """`D` and `D^T` of navier_stokes/navier_stokes.py:313-338 as one kernel
  each (`sfem_stokes_div`, `sfem_stokes_grad_t`)."""
  vspace: object
  pspace: object
  parts: list
  enc: torch.Tensor               # (E, n) encoded velocity indices
  penc: torch.Tensor | None       # (E, np) pressure node ids; None = identity
  host: dict
  zero_range: tuple
  num_pressure_nodes: int
  dirichlet_u8: torch.Tensor | None = None
  shared_order: torch.Tensor | None = None   # see `shared_slot_order`
  _split: tuple | None = None
  # the launches on compact connectivity + chains (component-major fields,
  # 3D, P = 6..8: `csrc/sfem_stokes_facet.h`), or None
  facet_parts: list | None = None
  _div_parts: list | None = None      # what `div` launches, see `_parts_for`
  _lay: object = None                 # see `_layered_plan`
  _lay_scale: tuple | None = None

  @classmethod
  def create(cls, vspace, pspace, dirichlet_mask=None,
             geometry='auto') -> 'StokesDivGrad':
    why = supports_fused_stokes(vspace, pspace)
    if why is not None:
      raise NotImplementedError(f'fused Stokes kernels unavailable: {why}')
    if geometry not in ('auto', 'multilinear', 'stored'):
      raise ValueError(f'unknown geometry mode {geometry!r}')
    mesh = vspace.mesh
    E = mesh.num_elements
    w = torch.as_tensor(vspace.quadrature.weights_nd(mesh.ndim),
                        dtype=vspace.dtype, device=vspace.device)
    if geometry == 'stored':
      kind = torch.zeros(E, dtype=torch.int32, device=vspace.device)
      coef = None
    else:
      kind, coef = classify_geometry(vspace)
      if geometry == 'multilinear':
        kind = torch.where(kind == _GEO_AFFINE,
                           torch.full_like(kind, _GEO_MULTILINEAR), kind)
    parts = []
    for k in (_GEO_AFFINE, _GEO_MULTILINEAR, _GEO_POINT):
      sel = kind == k
      count = int(sel.sum())
      if count == 0:
        continue
      part = {'geo_mode': k}
      if count < E:
        part['elem_list'] = torch.nonzero(sel).reshape(-1).to(
            torch.int32).contiguous()
      if k == _GEO_POINT:
        if count < E:
          part['kfac'] = _ops.stokes_setup(vspace.invjacs[sel].contiguous(),
                                           vspace.jacdets[sel].contiguous(), w)
          part['geo_index'] = (torch.cumsum(sel, 0) - 1).to(
              torch.int32).contiguous()
        else:
          part['kfac'] = _ops.stokes_setup(vspace.invjacs, vspace.jacdets, w)
      else:
        part['geo_elem'] = coef
      parts.append(part)
    plan = mesh.assembly_plan()
    mask = None
    if dirichlet_mask is not None:
      mask = torch.as_tensor(dirichlet_mask, device=vspace.device)
      mask = (mask != 0).to(torch.uint8).contiguous()
    enc = _ops.encode_elements(mesh.elements, mask, plan.multiplicity)
    pel = pspace.mesh.elements
    ident = torch.arange(pel.numel(), device=pel.device,
                         dtype=pel.dtype).reshape(pel.shape)
    penc = None if torch.equal(pel, ident) else pel.to(torch.int32).contiguous()
    host = {'dmat': vspace.interpolator._differentiation_matrix_1d(),
            'weights': np.asarray(vspace.quadrature.weights),
            'nodes': np.asarray(mesh.gridpoints_1d.node_values),
            'interp': pspace.interpolator._interpolation_matrix_1d()}
    facet_parts = None
    if (mesh.ndim == 3 and mesh.gridpoints_1d.num_points in STOKES_FACET_P and
        switches.get('SFEM_FACET') != '0' and
        switches.get('SFEM_STOKES_FACET') != '0'):
      facet_parts = cls._facet_parts(mesh, parts, mask, plan.multiplicity)
    return cls(vspace=vspace, pspace=pspace, parts=parts, enc=enc, penc=penc,
               host=host, zero_range=plan.zero_range,
               num_pressure_nodes=pspace.mesh.num_nodes, dirichlet_u8=mask,
               shared_order=cls._order(mesh, enc), facet_parts=facet_parts)

  @staticmethod
  def _facet_parts(mesh, parts, mask, multiplicity):
    """`parts` on the velocity mesh's facet table, every launch a list of chain
    segments (`facet_chains`); elements the table builder refuses keep their
    index rows.  None if no element qualifies."""
    E = mesh.num_elements
    P = mesh.gridpoints_1d.num_points
    tab, ok = _ops.facet_table(mesh.elements, mask, multiplicity, P)
    if not bool(ok.any()):
      return None
    seg_len = chain_segment_length(E)
    if switches.get('SFEM_CHAIN') == '0':
      seg_len = 1
    every = torch.arange(E, device=ok.device)
    use_box = switches.get('SFEM_BOX') != '0'
    out = []
    for part in parts:
      ids = every if 'elem_list' not in part else part['elem_list'].long()
      good = ok[ids]
      box = torch.zeros_like(good)
      if part['geo_mode'] == _GEO_AFFINE and use_box:
        box = good & _diagonal_jacobian(part['geo_elem'][ids])
      for sel, facet, mode in ((ids[box], True, _GEO_BOX),
                               (ids[good & ~box], True, part['geo_mode']),
                               (ids[~good], False, part['geo_mode'])):
        if sel.numel() == 0:
          continue
        new = {k: v for k, v in part.items() if k != 'elem_list'}
        new['geo_mode'] = mode
        if sel.numel() < E:
          new['elem_list'] = sel.to(torch.int32).contiguous()
        if facet:
          new['facet_table'] = tab
          new['chains'] = facet_chains(mesh.elements, sel, P, seg_len)
        out.append(new)
    return out

  def _parts_for(self, field, div=False):
    """Facet / chain launches for component-major fields.  The divergence
    takes them for box elements only: its general-geometry chain kernels hold
    three gathered components next to nine cofactors per point and spill
    (multilinear, 48^3 elements: 1.98 ms against 0.62 for the index-row kernel;
    affine with the fused dot 0.61 against 0.56), while `grad_t`, whose cost is
    the shared-node atomics, gains a quarter from the chains on every geometry
    (`scripts/time_stokes_jitter.py`).  `SFEM_STOKES_FACET_DIV=all` sends the
    divergence through them regardless (tests)."""
    if self.facet_parts is None or not _ops.is_component_major(field):
      return self.parts
    if not div or switches.get('SFEM_STOKES_FACET_DIV') == 'all':
      return self.facet_parts
    if self._div_parts is None:
      # (fp32: the box divergence measures 0.20 against 0.18 ms on index rows
      # at 40^3, so it keeps them too -- scripts/sweep_stokes_facet_vs_rows.py)
      box_ok = self.vspace.dtype == torch.float64
      def rows(q):      # the same elements from their index rows
        q = {k: v for k, v in q.items() if k not in ('facet_table', 'chains')}
        if q['geo_mode'] == _GEO_BOX:
          q['geo_mode'] = _GEO_AFFINE
        return q
      self._div_parts = [
          q if (q['geo_mode'] == _GEO_BOX and box_ok) or
          'facet_table' not in q else rows(q) for q in self.facet_parts]
    return self._div_parts

  @staticmethod
  def _order(mesh, enc):
    if (mesh.ndim == 3 and mesh.gridpoints_1d.num_points <= 8 and
        switches.get('SFEM_SORTED_SCATTER') != '0'):
      return shared_slot_order(enc)
    return None

  def _split_encoding(self):
    """`enc` with every node of the periodic / partition exchange flagged
    SHARED too, and the node range that then needs clearing: what the split
    `E` (`sfem_stokes_e_first` / `_second`) assumes."""
    if self._split is None:
      mesh = self.vspace.mesh
      mult = mesh.assembly_plan().multiplicity.clone()
      ids = []
      gi = mesh.exchange_gather_indices
      if gi is not None and gi.numel():
        ids.append(gi[gi >= 0].to(torch.int64))
      if mesh.neighbor_plan is not None:
        ids.append(mesh.neighbor_plan.interface_nodes(mult.device).to(
            torch.int64))
      if ids:
        idx = torch.cat(ids)
        mult[idx] = torch.clamp(mult[idx], min=2)
      other = torch.nonzero(mult != 1).reshape(-1)
      rng = (int(other.min()), int(other.max()) + 1) if other.numel() else (0, 0)
      enc = _ops.encode_elements(mesh.elements, self.dirichlet_u8, mult)
      self._split = (enc, rng, self._order(mesh, enc))
    return self._split

  def e_apply(self, p, scale=None, exchange=None):
    """(Np,) -> (Np,):  D [ scale * QQ^T (mask * D^T p) ] = `StokesSEM.E` for a
    diagonal Q, with the element-interior velocity nodes kept in registers
    (`sfem_stokes_e_first` / `sfem_stokes_e_second`).  `exchange(w)` applies
    QQ^T in place to the (N, d) component-major intermediate, of which only
    the shared nodes are defined; None on a mesh without periodic images or
    partitions."""
    mesh = self.vspace.mesh
    if self.penc is not None:
      raise NotImplementedError('split E needs element-local pressure nodes')
    if tuple(p.shape) != (self.num_pressure_nodes,):
      raise ValueError(f'expected ({self.num_pressure_nodes},) pressure, got '
                       f'{tuple(p.shape)}')
    from swirl_fem_amd.core import layout
    p = p.to(self.vspace.dtype).contiguous()
    enc, zero_range, order = self._split_encoding()
    w = layout.empty_component_major((mesh.num_nodes, mesh.ndim), p.dtype,
                                     p.device)
    if scale is not None:
      scale = scale.to(p.dtype)
      scale = (scale.contiguous() if scale.dim() == 1
               else _like_layout(scale.expand_as(w), w))
    out = torch.empty(self.num_pressure_nodes, dtype=p.dtype, device=p.device)
    args = (enc, self.penc, self.parts, self.host, mesh.ndim,
            mesh.gridpoints_1d.num_points)
    _ops.stokes_e_first(p, w, out, *args, zero_range, scale, order)
    if exchange is not None:
      w = exchange(w)
    return _ops.stokes_e_second(w, out, *args, scale)

  # Layered D^T inside E (index-row kernels: 2D, and 3D outside the facet
  # kernels' range).  The scatter of `sfem_stokes_grad_t` waits on the
  # memory-side atomic unit as soon as a launch has enough elements (an
  # ensemble of 8 Kolmogorov flows: 1.17 M requests in 72 us, 16 G/s).  The
  # kernel needs no change to lose them: a second index row per element in
  # which every (element, slot) writer of a shared node has a position of its
  # OWN -- the first writer keeps the node, the others get positions behind
  # the N nodes -- and no SHARED flag, so every slot takes the kernel's
  # plain-store branch (nothing to clear either).  The sums are then formed by
  # the class kernel that applies QQ^T to periodic images anyway
  # (`sfem_exchange_classes`): a class is all positions of all images of a
  # node.  `sfem_stokes_div` reads the first N positions of each component.
  def _layered_plan(self):
    """None, or (enc, n_ext, node_of_pos, members, offsets, num_classes)."""
    if self._lay is None:
      mesh = self.vspace.mesh
      ok = (self.facet_parts is None and mesh.axis_name is None and
            mesh.neighbor_plan is None and
            mesh.elements.numel() <= (1 << 28))
      self._lay = self._build_layered(mesh) if ok else False
    return self._lay or None

  def _build_layered(self, mesh):
    dev = mesh.device
    N = mesh.num_nodes
    el = mesh.elements.to(torch.int64).reshape(-1)
    order = torch.argsort(el, stable=True)
    ids = el[order]
    first = torch.ones_like(ids, dtype=torch.bool)
    first[1:] = ids[1:] != ids[:-1]
    extra = ~first
    pos_sorted = torch.where(first, ids, N + torch.cumsum(extra, 0) - 1)
    pos = torch.empty_like(el)
    pos[order] = pos_sorted
    n_ext = N + int(extra.sum())
    if n_ext >= _lib_idx_mask():
      return False
    node_of_pos = torch.cat([torch.arange(N, device=dev), ids[extra]])
    pos32 = pos.to(torch.int32)
    dirichlet = self.enc.reshape(-1) < 0           # SFEM_IDX_DIRICHLET: bit 31
    enc = torch.where(dirichlet, pos32 | torch.tensor(
        -2 ** 31, dtype=torch.int32, device=dev), pos32).reshape(
            self.enc.shape).contiguous()
    # classes: positions of one node and of its periodic images
    key = torch.arange(N, device=dev)
    gi, ui = mesh.exchange_gather_indices, mesh.exchange_unique_indices
    if gi is not None and gi.numel() and ui is not None:
      g = gi.to(torch.int64)
      u = torch.as_tensor(np.asarray(ui).astype(np.int64), device=dev)
      keep = g >= 0
      key[g[keep]] = N + u[keep]
    pkey = key[node_of_pos]
    korder = torch.argsort(pkey, stable=True)
    _, counts = torch.unique_consecutive(pkey[korder], return_counts=True)
    big = counts > 1
    members = korder[torch.repeat_interleave(big, counts)]
    offsets = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev),
                         torch.cumsum(counts[big], 0)])
    return (enc, n_ext, node_of_pos, members.to(torch.int32).contiguous(),
            offsets.to(torch.int32).contiguous(), int(big.sum()))

  def supports_layered_e(self) -> bool:
    return self._layered_plan() is not None

  def e_layered(self, p, scale=None, dot_with=None, dot_out=None):
    """(Np,) -> (Np,):  D [ QQ^T ( scale * mask * D^T p ) ]  = `StokesSEM.E`
    for a diagonal Q, the direct-stiffness sum of D^T without atomics (see
    `_layered_plan`); periodic images included, one partition."""
    enc, n_ext, node_of_pos, members, offsets, num_classes = \
        self._layered_plan()
    mesh = self.vspace.mesh
    if tuple(p.shape) != (self.num_pressure_nodes,):
      raise ValueError(f'expected ({self.num_pressure_nodes},) pressure, got '
                       f'{tuple(p.shape)}')
    from swirl_fem_amd.core import layout
    p = p.to(self.vspace.dtype).contiguous()
    w = layout.empty_component_major((n_ext, mesh.ndim), p.dtype, p.device)
    if scale is not None:
      if self._lay_scale is None or self._lay_scale[0] is not scale:
        ext = scale.to(p.dtype).index_select(0, node_of_pos)
        ext = (ext.contiguous() if ext.dim() == 1
               else _like_layout(ext.expand_as(w), w))
        self._lay_scale = (scale, ext)
      scale = self._lay_scale[1]
    P = mesh.gridpoints_1d.num_points
    _ops.stokes_grad_t(p, w, enc, self.penc, self.parts, self.host, mesh.ndim,
                       P, (0, 0), None, scale)
    _ops.exchange_classes_(w, members, offsets, num_classes)
    out = (torch.empty if self.penc is None else torch.zeros)(
        self.num_pressure_nodes, dtype=p.dtype, device=p.device)
    if dot_out is not None:
      dot_with = dot_with.to(p.dtype).contiguous()
    return _ops.stokes_div(w, out, self.enc, self.penc, self.parts, self.host,
                           mesh.ndim, P, None, dot_with, dot_out)

  def div(self, u, scale=None, out=None, dot_with=None, dot_out=None):
    """(N, d) -> (Np,):  D (scale * u); `scale` is (N, d) or (N,).
    `dot_out`: SFEM_DOT_SLOTS doubles accumulating `dot_with . result`."""
    mesh = self.vspace.mesh
    if tuple(u.shape) != (mesh.num_nodes, mesh.ndim):
      raise ValueError(f'expected ({mesh.num_nodes}, {mesh.ndim}) velocity, '
                       f'got {tuple(u.shape)}')
    u = u.to(self.vspace.dtype)
    if not (u.is_contiguous() or _ops.is_component_major(u)):
      u = u.contiguous()
    if scale is not None:
      scale = scale.to(u.dtype)
      if scale.dim() == 1:           # one factor per node for all components
        scale = scale.contiguous()
      else:
        scale = _like_layout(scale.expand_as(u), u)
    if out is None:
      # pressure nodes that no element references (none on refiner meshes)
      out = (torch.empty if self.penc is None else torch.zeros)(
          self.num_pressure_nodes, dtype=u.dtype, device=u.device)
    if dot_out is not None:
      dot_with = dot_with.to(u.dtype).contiguous()
      if tuple(dot_with.shape) != (self.num_pressure_nodes,):
        raise ValueError('dot_with must be a pressure vector')
    return _ops.stokes_div(u, out, self.enc, self.penc,
                           self._parts_for(u, div=True),
                           self.host, mesh.ndim, mesh.gridpoints_1d.num_points,
                           scale, dot_with, dot_out)

  def grad_t(self, p, out=None, component_major=False, scale=None):
    """(Np,) -> (N, d):  mask * (scale * D^T p), the factor applied to every
    element's contribution before assembly (it must be equal on all copies
    of a node; then it commutes with the direct-stiffness sum and QQ^T)."""
    mesh = self.vspace.mesh
    if tuple(p.shape) != (self.num_pressure_nodes,):
      raise ValueError(f'expected ({self.num_pressure_nodes},) pressure, got '
                       f'{tuple(p.shape)}')
    p = p.to(self.vspace.dtype).contiguous()
    if out is None:
      shape = (mesh.num_nodes, mesh.ndim)
      if component_major:
        from swirl_fem_amd.core import layout
        out = layout.empty_component_major(shape, p.dtype, p.device)
      else:
        out = torch.empty(shape, dtype=p.dtype, device=p.device)
    if scale is not None:
      scale = scale.to(p.dtype)
      scale = (scale.contiguous() if scale.dim() == 1
               else _like_layout(scale.expand_as(out), out))
    return _ops.stokes_grad_t(p, out, self.enc, self.penc,
                              self._parts_for(out), self.host, mesh.ndim,
                              mesh.gridpoints_1d.num_points, self.zero_range,
                              self.shared_order, scale)


def _lib_idx_mask():
  from swirl_fem_amd import _lib
  return _lib.SFEM_IDX_MASK


def _diagonal_jacobian(coef):
  """(E,) bool: rows of the (E, 24) multilinear coefficients whose map is
  x_c = x0_c + h_c * xi_c (axis-aligned boxes in the element's own axis order):
  the Stokes kernels then need one derivative per component
  (`SFEM_GEO_BOX`)."""
  lin = coef[:, :9].reshape(-1, 3, 3)            # d x_c / d xi_k at [k, c]
  diag = torch.diagonal(lin, dim1=1, dim2=2).abs()
  scale = diag.max(dim=1).values
  tol = BOX_TOL[coef.dtype] * scale
  off = lin.abs() * (1 - torch.eye(3, dtype=coef.dtype, device=coef.device))
  higher = coef[:, 9:21].abs().max(dim=1).values
  return ((off.reshape(-1, 9).max(dim=1).values <= tol) & (higher <= tol) &
          (diag.min(dim=1).values > 0))


def _like_layout(t, ref):
  """`t` (same shape as `ref`) materialised in the memory layout of `ref`."""
  if t.stride() == ref.stride() and (t.is_contiguous() or
                                     _ops.is_component_major(t)):
    return t
  out = torch.empty_like(ref)        # preserves the dense layout of ref
  out.copy_(t)
  return out


# ---------------------------------------------------------------------------
# Helmholtz operator with a quadrature that differs from the nodes
# ---------------------------------------------------------------------------
def supports_two_grid(fespace) -> str | None:
  """None if `TwoGridHelmholtzOperator` applies, else the reason."""
  mesh = fespace.mesh
  q = fespace.quadrature.num_points
  if mesh.ndim not in (2, 3):
    return f'ndim={mesh.ndim}'
  if not 2 <= q <= MAX_FUSED_P:
    return f'q={q} outside 2..{MAX_FUSED_P}'
  d = _quadrature_dmat(fespace)
  if not np.allclose(d[::-1, ::-1], -d, rtol=0,
                     atol=1e-12 * max(1.0, np.abs(d).max())):
    return 'quadrature points are not symmetric about 0'
  return None


def _quadrature_dmat(fespace):
  """Differentiation matrix of the Lagrange basis ON the quadrature points."""
  from swirl_fem_amd.core.interpolation import BarycentricInterpolator
  qn = fespace.quadrature.nodes
  return BarycentricInterpolator(ndim=fespace.mesh.ndim, gridpoints_1d=qn,
                                 evalpoints_1d=qn)._differentiation_matrix_1d()


@dataclasses.dataclass(eq=False)
class TwoGridHelmholtzOperator:
  """`mask * scatter((l0 B + l1 A)_local(gather(u)))` when the quadrature is
  not the node set (the reference's Poisson example integrates with
  `order + (ndim+1)//2` Gauss points, examples/poisson.py:112-114).

  With I the (Q, n) interpolation to the quadrature points,
  `A_local = I^T A_q I` where `A_q` is the *collocated* operator of the
  Lagrange basis on the quadrature points (the gradient of the interpolant is
  exact there because q >= P).  So the element work is: interpolate
  (`sfem_basis_eval`), the fused element-local kernel on the q-grid
  (`sfem_helmholtz_local` with the q-point differentiation matrix, weights and
  nodes; geometry of multilinear elements evaluated in registers), transposed
  interpolation (`sfem_basis_eval_t`).  Replaces the generic chain basis_eval
  (values + physical gradients) -> pointwise form -> basis_eval_t."""
  fespace: object
  parts: list
  host: dict
  mask: torch.Tensor | None        # (N,) 1 inside, 0 on Dirichlet nodes

  @classmethod
  def create(cls, fespace, dirichlet_mask=None,
             geometry='auto') -> 'TwoGridHelmholtzOperator':
    why = supports_two_grid(fespace)
    if why is not None:
      raise NotImplementedError(f'two-grid Helmholtz unavailable: {why}')
    if geometry not in ('auto', 'stored'):
      raise ValueError(f'unknown geometry mode {geometry!r}')
    mesh = fespace.mesh
    E, dev = mesh.num_elements, fespace.device
    w = torch.as_tensor(fespace.quadrature.weights_nd(mesh.ndim),
                        dtype=fespace.dtype, device=dev)
    from swirl_fem_amd.core.interpolation import NodeType
    corners = mesh.gridpoints_1d.node_type in (
        NodeType.GAUSS_LOBATTO_LEGENDRE, NodeType.NEWTON_COTES)
    if geometry == 'stored' or not corners:
      kind = torch.zeros(E, dtype=torch.int32, device=dev)
      coef = None
    else:
      kind, coef = classify_geometry(fespace)
    parts = []
    for k in (_GEO_AFFINE, _GEO_MULTILINEAR, _GEO_POINT):
      sel = kind == k
      count = int(sel.sum())
      if count == 0:
        continue
      part = {'geo_mode': k}
      if count < E:
        part['elem_list'] = torch.nonzero(sel).reshape(-1).to(
            torch.int32).contiguous()
      if k == _GEO_POINT:
        if count < E:
          part['geo'] = _ops.helmholtz_setup(
              fespace.invjacs[sel].contiguous(),
              fespace.jacdets[sel].contiguous(), w)
          part['geo_index'] = (torch.cumsum(sel, 0) - 1).to(
              torch.int32).contiguous()
        else:
          part['geo'] = _ops.helmholtz_setup(fespace.invjacs, fespace.jacdets,
                                             w)
      else:
        part['geo_elem'] = coef
      parts.append(part)
    mask = None
    if dirichlet_mask is not None:
      mask = (torch.as_tensor(dirichlet_mask, device=dev) == 0).to(
          fespace.dtype)
    host = {'dmat': _quadrature_dmat(fespace),
            'weights': np.asarray(fespace.quadrature.weights),
            'nodes': np.asarray(fespace.quadrature.nodes.node_values)}
    return cls(fespace=fespace, parts=parts, host=host, mask=mask)

  def apply_local(self, u_local, lambda0=0.0, lambda1=1.0):
    """(E, n[, nc]) -> (E, n[, nc])."""
    fes = self.fespace
    mesh = fes.mesh
    E, n = mesh.num_elements, mesh.num_nodes_per_element
    scalar = u_local.dim() == 2
    u3 = (u_local[..., None] if scalar else u_local).to(fes.dtype)
    nc = u3.shape[-1]
    q = fes.quadrature.num_points
    uq = u3 if fes.is_collocated else fes._basis(u3, True, False)[0]
    rq = _ops.helmholtz_local(uq.contiguous(), self.parts, self.host,
                              mesh.ndim, q, lambda0, lambda1)
    if fes.is_collocated:
      r3 = rq
    else:
      i1, g1 = fes._matrices()
      ones = fes._cache.get('ones_eq')
      if ones is None:
        ones = fes._cache['ones_eq'] = torch.ones(
            (E, q ** mesh.ndim), dtype=fes.dtype, device=fes.device)
      r3 = _ops.basis_eval_t(rq, None, i1, g1, None, ones, mesh.ndim,
                             mesh.gridpoints_1d.num_points, q, nc, False)
    return r3[..., 0] if scalar else r3

  def apply(self, u, lambda0=0.0, lambda1=1.0):
    """u (N,) or (N, nc) -> mask * scatter(local(gather(u)))."""
    mesh = self.fespace.mesh
    if u.shape[0] != mesh.num_nodes:
      raise ValueError(f'expected {mesh.num_nodes} nodal values, got '
                       f'{tuple(u.shape)}')
    u = u.to(self.fespace.dtype)
    if u.dim() == 1:
      out = mesh.scatter(self.apply_local(mesh.gather(u), lambda0, lambda1))
      return out if self.mask is None else out * self.mask
    loc = self.apply_local(_ops.gather_rows(u.contiguous(), mesh.elements),
                           lambda0, lambda1)
    out = _ops.scatter_add(loc, mesh.elements, mesh.num_nodes, ncomp=u.shape[1])
    return out if self.mask is None else out * self.mask[:, None]

  def linear_operator(self, lambda0=0.0, lambda1=1.0):
    return lambda u: self.apply(u, lambda0, lambda1)


# ---------------------------------------------------------------------------
# Over-integrated convection
# ---------------------------------------------------------------------------
def _grid_geometry_parts(fespace, point_setup, geometry='auto'):
  """One launch description per geometry kind for kernels that work on the
  quadrature grid of `fespace` (`point_setup(invjacs, jacdets, w)` builds the
  stored per-point data of curved elements)."""
  from swirl_fem_amd.core.interpolation import NodeType
  mesh = fespace.mesh
  E, dev = mesh.num_elements, fespace.device
  w = torch.as_tensor(fespace.quadrature.weights_nd(mesh.ndim),
                      dtype=fespace.dtype, device=dev)
  corners = mesh.gridpoints_1d.node_type in (
      NodeType.GAUSS_LOBATTO_LEGENDRE, NodeType.NEWTON_COTES)
  if geometry == 'stored' or not corners:
    kind = torch.zeros(E, dtype=torch.int32, device=dev)
    coef = None
  else:
    kind, coef = classify_geometry(fespace)
  parts = []
  for k in (_GEO_AFFINE, _GEO_MULTILINEAR, _GEO_POINT):
    sel = kind == k
    count = int(sel.sum())
    if count == 0:
      continue
    part = {'geo_mode': k}
    if count < E:
      part['elem_list'] = torch.nonzero(sel).reshape(-1).to(
          torch.int32).contiguous()
    if k == _GEO_POINT:
      if count < E:
        part['kfac'] = point_setup(fespace.invjacs[sel].contiguous(),
                                   fespace.jacdets[sel].contiguous(), w)
        part['geo_index'] = (torch.cumsum(sel, 0) - 1).to(
            torch.int32).contiguous()
      else:
        part['kfac'] = point_setup(fespace.invjacs, fespace.jacdets, w)
    else:
      part['geo_elem'] = coef
    parts.append(part)
  return parts


@dataclasses.dataclass(eq=False)
class ConvectionOperator:
  """`C_local(u)_{i,c} = int phi_i u_j d_j u_c` on the over-integration space
  (navier_stokes.py:238-245) as: interpolate the nodal velocity to the
  quadrature grid (`sfem_basis_eval`), one fused kernel there
  (`sfem_stokes_convect_local`: collocated derivative lines, cofactor
  geometry, contravariant velocity, product), transposed interpolation
  (`sfem_basis_eval_t`).  Nothing of size (E, Q, d, d) is ever stored."""
  fespace: object
  parts: list
  host: dict

  @classmethod
  def create(cls, fespace, geometry='auto') -> 'ConvectionOperator':
    why = supports_two_grid(fespace)
    q = fespace.quadrature.num_points
    if why is None and q < 4:
      why = f'q={q} < 4'
    if why is not None:
      raise NotImplementedError(f'fused convection unavailable: {why}')
    parts = _grid_geometry_parts(fespace, _ops.stokes_setup, geometry)
    host = {'dmat': _quadrature_dmat(fespace),
            'weights': np.asarray(fespace.quadrature.weights),
            'nodes': np.asarray(fespace.quadrature.nodes.node_values)}
    return cls(fespace=fespace, parts=parts, host=host)

  def apply_local(self, u_local):
    """(E, n, d) nodal velocity -> (E, n, d) local convection covector."""
    fes = self.fespace
    mesh = fes.mesh
    q = fes.quadrature.num_points
    u3 = u_local.to(fes.dtype)
    uq = u3 if fes.is_collocated else fes._basis(u3, True, False)[0]
    cq = _ops.stokes_convect_local(uq, self.parts, self.host, mesh.ndim, q)
    if fes.is_collocated:
      return cq
    i1, g1 = fes._matrices()
    ones = fes._cache.get('ones_eq')
    if ones is None:
      ones = fes._cache['ones_eq'] = torch.ones(
          (mesh.num_elements, q ** mesh.ndim), dtype=fes.dtype,
          device=fes.device)
    return _ops.basis_eval_t(cq, None, i1, g1, None, ones, mesh.ndim,
                             mesh.gridpoints_1d.num_points, q, mesh.ndim,
                             False)
