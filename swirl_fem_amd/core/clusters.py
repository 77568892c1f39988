"""Element clusters for the cluster-assembled operator kernels.

The fused operator kernels assemble (direct-stiffness summation, reference
`core/gather_scatter.py:130-133`) the nodes shared by the elements of one
*cluster* in LDS and touch HBM once per node (`csrc/sfem_helmholtz_cluster.h`).
This module builds, once per operator, what those kernels read:

* a grouping of the elements into clusters of up to `cluster_size` elements
  that sit together (recursive coordinate bisection of the element centroids;
  on a structured mesh the leaves are the 2x2x2 blocks),
* per cluster the table of the nodes on the lattice boundary of its elements
  (the only nodes a conforming mesh can share) in ascending node order, each
  flagged DIRICHLET and/or SHARED (= also held by an element outside the
  cluster: that node still needs an atomic),
* the element index rows in cluster form: a lattice-boundary slot holds the
  position of its node in the table instead of the node id.  Which slots those
  are follows from their place in the element, so the kernels test no flags.

Everything is torch index arithmetic on the device; the reference has no
counterpart (XLA's scatter-add hides the assembly).
"""

from __future__ import annotations

import dataclasses

import torch

from swirl_fem_amd import _ops

IDX_MASK = 0x3FFFFFFF
IDX_SHARED = 0x40000000
IDX_DIRICHLET = 0x80000000


@dataclasses.dataclass(eq=False)
class ClusterPlan:
  elems: torch.Tensor        # (C, cluster_size) int32 element ids, -1 = empty
  offsets: torch.Tensor      # (C + 1,) int32
  nodes: torch.Tensor        # (T,) int32 (bit pattern of uint32)
  enc: torch.Tensor          # (E, n) int32 cluster-form index rows
  cluster_size: int
  max_shared: int            # largest table
  num_surface: int           # table entries that still need an atomic
  num_complete: int          # table entries stored plainly

  @property
  def num_clusters(self) -> int:
    return self.elems.shape[0]


def rcb_order(cent: torch.Tensor, leaf: int) -> tuple[torch.Tensor, torch.Tensor]:
  """Recursive coordinate bisection of `cent` (M, d) down to groups of at most
  `leaf` points.  Returns `(perm, group)`: `perm` lists the points group by
  group, `group[k]` is the group of point `perm[k]` (ascending).  A segment of
  L > leaf points is cut across its longest bounding-box axis so that the left
  part is a multiple of `leaf` (all groups are full except one per ragged
  segment)."""
  M, d = cent.shape
  dev = cent.device
  perm = torch.arange(M, device=dev)
  seg = torch.zeros(M, dtype=torch.int64, device=dev)      # of perm[k]
  while True:
    nseg = int(seg.max()) + 1 if M else 0
    length = torch.bincount(seg, minlength=nseg)
    if M == 0 or int(length.max()) <= leaf:
      break
    x = cent[perm]
    lo = torch.full((nseg, d), float('inf'), dtype=cent.dtype, device=dev)
    hi = torch.full((nseg, d), float('-inf'), dtype=cent.dtype, device=dev)
    idx = seg[:, None].expand(M, d)
    lo.scatter_reduce_(0, idx, x, 'amin')
    hi.scatter_reduce_(0, idx, x, 'amax')
    axis = torch.argmax(hi - lo, dim=1)                    # (nseg,)
    key = x.gather(1, axis[seg][:, None])[:, 0]
    # sort by (segment, coordinate along the segment's axis), stable
    o1 = torch.argsort(key, stable=True)
    o2 = torch.argsort(seg[o1], stable=True)
    order = o1[o2]
    perm, seg = perm[order], seg[order]
    start = torch.cumsum(length, 0) - length
    pos = torch.arange(M, device=dev) - start[seg]
    groups = (length + leaf - 1) // leaf
    left = ((groups + 1) // 2) * leaf                      # multiple of leaf
    split = length > leaf
    right = split[seg] & (pos >= left[seg])
    seg = 2 * seg + right.to(torch.int64)
    # renumber densely, keeping the order
    _, seg = torch.unique(seg, return_inverse=True)
  return perm, seg


def corner_centroids(mesh, elem_ids: torch.Tensor) -> torch.Tensor:
  """Mean of the 2^d corner nodes of the listed elements, (M, d)."""
  d, P = mesh.ndim, mesh.gridpoints_1d.num_points
  n = mesh.num_nodes_per_element
  idx = torch.arange(n, device=elem_ids.device).reshape([P] * d)
  ends = torch.tensor([0, P - 1], device=elem_ids.device)
  for ax in range(d):
    idx = idx.index_select(ax, ends)
  cn = mesh.elements[elem_ids][:, idx.reshape(-1)].to(torch.int64)
  return mesh.node_coords[cn].mean(dim=1)


def lattice_boundary_slots(P: int, ndim: int, device) -> torch.Tensor:
  """(P^ndim,) bool: slots with an index 0 or P-1 along some axis."""
  r = torch.arange(P, device=device)
  rim = (r == 0) | (r == P - 1)
  out = torch.zeros([P] * ndim, dtype=torch.bool, device=device)
  for ax in range(ndim):
    shape = [1] * ndim
    shape[ax] = P
    out = out | rim.reshape(shape)
  return out.reshape(-1)


def supports_clusters(mesh, enc: torch.Tensor) -> str | None:
  """None if the mesh can be cluster-assembled, else the reason: every real
  element must be complete (no -1 slots) and only lattice-boundary slots may
  be shared (true of every conforming mesh)."""
  P, d = mesh.gridpoints_1d.num_points, mesh.ndim
  rows = enc.to(torch.int64) & 0xFFFFFFFF
  pad = (rows & IDX_MASK) == IDX_MASK
  real = ~pad.all(dim=1)
  if bool((pad & real[:, None]).any()):
    return 'an element has padding (-1) slots'
  inside = ~lattice_boundary_slots(P, d, enc.device)
  if bool((((rows & IDX_SHARED) != 0) & real[:, None])[:, inside].any()):
    return 'a lattice-interior slot is shared (non-conforming mesh)'
  return None


def build_cluster_plan(mesh, enc: torch.Tensor, multiplicity: torch.Tensor,
                       parts_elem_ids: list, cluster_size: int,
                       max_shared: int, order: str = 'rcb') -> list:
  """One `ClusterPlan` per entry of `parts_elem_ids` (element id lists, int64,
  or None = all elements); the plans share one cluster-form `enc` array.

  `enc`: (E, n) int32 from `sfem_encode_elements` (SHARED = the node has more
  than one slot in the whole mesh), `multiplicity`: (N,) slots per node.
  The mesh must pass `supports_clusters`.
  """
  dev = enc.device
  E, n = enc.shape
  cenc = enc.clone()
  rim = lattice_boundary_slots(mesh.gridpoints_1d.num_points, mesh.ndim, dev)
  plans = []
  for ids in parts_elem_ids:
    if ids is None:
      ids = torch.arange(E, device=dev)
    ids = ids.to(torch.int64)
    # padding elements (all slots -1) take no part
    real = ((enc[ids].to(torch.int64) & IDX_MASK) != IDX_MASK).any(dim=1)
    ids = ids[real]
    M = ids.numel()
    if M == 0:
      plans.append(None)
      continue
    if order == 'rcb':
      perm, group = rcb_order(corner_centroids(mesh, ids).to(torch.float64),
                              cluster_size)
    else:                                  # consecutive elements (testing)
      perm = torch.arange(M, device=dev)
      group = perm // cluster_size
    ids = ids[perm]
    plan = None
    for _ in range(8):
      plan = _tables(enc, multiplicity, ids, group, cluster_size, cenc, rim)
      if plan.max_shared <= max_shared:
        break
      # a cluster whose table does not fit the LDS strip: halve it
      size = plan.offsets[1:] - plan.offsets[:-1]
      big = size.to(torch.int64) > max_shared
      start = torch.cumsum(torch.bincount(group), 0) - torch.bincount(group)
      pos = torch.arange(M, device=dev) - start[group]
      count = torch.bincount(group)[group]
      second = big[group] & (pos >= (count + 1) // 2)
      _, group = torch.unique(2 * group + second.to(torch.int64),
                              return_inverse=True)
    else:
      raise RuntimeError('cluster tables do not fit the kernel limit '
                         f'({plan.max_shared} > {max_shared})')
    plans.append(plan)
  for plan in plans:
    if plan is not None:
      plan.enc = cenc
  return plans


def _tables(enc, multiplicity, ids, group, cluster_size, cenc,
            rim) -> ClusterPlan:
  """Tables and cluster-form rows for elements `ids` grouped by `group`
  (ascending, dense); writes the rows of `ids` into `cenc`.  `rim`: (n,) bool,
  the lattice-boundary slots (all of them go through the table)."""
  dev = enc.device
  M, n = ids.numel(), enc.shape[1]
  C = int(group.max()) + 1
  rows = enc[ids].to(torch.int64) & 0xFFFFFFFF                   # (M, n)
  node = rows & IDX_MASK
  shared = rim[None, :].expand(M, n) & (node != IDX_MASK)
  dirichlet = (rows & IDX_DIRICHLET) != 0
  sel = torch.nonzero(shared.reshape(-1)).reshape(-1)            # slot indices
  slot_cluster = group[sel // n]
  key = (slot_cluster << 31) | node.reshape(-1)[sel]
  uniq, inverse, counts = torch.unique(key, return_inverse=True,
                                       return_counts=True)
  u_cluster, u_node = uniq >> 31, uniq & 0x7FFFFFFF
  size = torch.bincount(u_cluster, minlength=C)
  offsets = torch.zeros(C + 1, dtype=torch.int64, device=dev)
  offsets[1:] = torch.cumsum(size, 0)
  position = torch.arange(uniq.numel(), device=dev) - offsets[u_cluster]
  surface = counts < multiplicity[u_node].to(torch.int64)
  u_dir = torch.zeros(uniq.numel(), dtype=torch.bool, device=dev)
  u_dir[inverse] = dirichlet.reshape(-1)[sel]
  table = (u_node | (surface.to(torch.int64) * IDX_SHARED) |
           (u_dir.to(torch.int64) * IDX_DIRICHLET))
  new_rows = (rows & (IDX_MASK | IDX_DIRICHLET)).reshape(-1).clone()
  new_rows[sel] = (rows.reshape(-1)[sel] & IDX_DIRICHLET) | position[inverse]
  cenc[ids] = _as_int32(new_rows).reshape(M, n)
  # (C, cluster_size) element ids
  start = torch.cumsum(torch.bincount(group, minlength=C), 0) - torch.bincount(
      group, minlength=C)
  pos = torch.arange(M, device=dev) - start[group]
  elems = torch.full((C, cluster_size), -1, dtype=torch.int32, device=dev)
  elems[group, pos] = ids.to(torch.int32)
  return ClusterPlan(
      elems=elems.contiguous(), offsets=offsets.to(torch.int32).contiguous(),
      nodes=_as_int32(table).contiguous(), enc=cenc,
      cluster_size=cluster_size,
      max_shared=int(size.max()) if size.numel() else 0,
      num_surface=int(surface.sum()),
      num_complete=int((~surface).sum()))


def _as_int32(x: torch.Tensor) -> torch.Tensor:
  """uint32 bit patterns held in int64 -> int32 with the same bits."""
  return torch.where(x >= (1 << 31), x - (1 << 32), x).to(torch.int32)


def cluster_limits(P: int, dtype: torch.dtype) -> tuple[int, int] | None:
  """(cluster_size, max_shared) of the compiled cluster kernels, or None when
  there are none for this order."""
  return _ops.helmholtz_cluster_limits(P, dtype)
