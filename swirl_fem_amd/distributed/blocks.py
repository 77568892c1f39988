"""Block partitions of a structured hex mesh, built rank-locally.

The reference partitions by regrouping the *global* element list
(`Premesh.finalize(axis_name)`, core/premesh.py:170-222, with block partitions
from `unit_cube_mesh(partitions=...)`, common/premesh_commons.py:130-138) and
derives a dense `(P, S)` table of all shared DOFs
(core/gather_scatter.py:318-358).  That is O(global mesh) work on one host and
does not scale to 128^3 elements at p = 7 (0.7 G nodes).

Here every rank builds only its own `n^3`-element block -- same generator, same
refiner, same local numbering as an unpartitioned mesh of that block -- and
finds the DOFs it shares with each of its (up to 26) neighbour blocks from the
global GLL lattice coordinates of its surface nodes.  Both sides of a pair sort
the shared DOFs by the same global lattice key, so no setup communication is
needed.  The result is the `NeighborPlan` consumed by
`distributed.comm.neighbor_exchange` (QQ^T over RCCL).
"""

from __future__ import annotations

import dataclasses
import itertools

import numpy as np
import torch

from swirl_fem_amd.common.premesh_commons import box_mesh
from swirl_fem_amd.core.interpolation import Nodes1D
from swirl_fem_amd.core.interpolation import NodeType
from swirl_fem_amd.core.mesh import Mesh
from swirl_fem_amd.core.mesh_refiner import refine_premesh
from swirl_fem_amd.distributed import comm


@dataclasses.dataclass(eq=False)
class BlockPartition:
  mesh: Mesh
  rank: int
  block_grid: tuple
  block_coords: tuple
  num_global_nodes: int
  plan: comm.NeighborPlan
  global_keys: np.ndarray | None = None   # global GLL lattice key per node
  premesh: object = None                  # the block's order-1 premesh

  def reduce_sum_(self, t: torch.Tensor) -> torch.Tensor:
    """All-reduce of CG scalars across the partitions (RCCL)."""
    return comm.all_reduce_sum_(t)


_SETUP_GROUP = {}


def _setup_group():
  """A gloo process group over all ranks for setup-time collectives (created
  once; every rank must reach this call)."""
  dist = comm.dist
  if comm.transport() is not None or dist.get_backend() == 'gloo':
    return None
  if 'group' not in _SETUP_GROUP:
    _SETUP_GROUP['group'] = dist.new_group(backend='gloo')
  return _SETUP_GROUP['group']


def tiled_element_order(n, ndim: int, tile: int) -> np.ndarray:
  """Permutation of the C-ordered elements of an `n[0] x .. x n[d-1]` block
  (`n` an int = cube) that visits them tile by tile (`tile^ndim` elements
  each).  Elements that share faces are then close in the launch order *and*
  (through the refiner's first-touch numbering) in memory, which shortens the
  reuse distance of the gathered nodal values."""
  ns = (n,) * ndim if np.isscalar(n) else tuple(n)
  idx = np.arange(int(np.prod(ns))).reshape(ns)
  out = []
  for t in np.ndindex(*[-(-k // tile) for k in ns]):
    sl = tuple(slice(ti * tile, min((ti + 1) * tile, k))
               for ti, k in zip(t, ns))
    out.append(idx[sl].reshape(-1))
  return np.concatenate(out)


def build_block_partition(n, P: int, block_grid, rank: int, *,
                          device=None, dtype=torch.float64, lo=0.0, hi=1.0,
                          jitter: float = 0.0, tile: int = 0,
                          periodic_dims=()) -> BlockPartition:
  """This rank's block of the `(n*px, n*py, n*pz)`-element mesh on [lo,hi]^3.

  Args:
    n: elements per direction in one block: an int, or one count per
      direction (strong scaling splits a fixed mesh into non-cubic blocks).
    P: GLL nodes per direction (order + 1).
    block_grid: (px, py, pz) blocks; rank = C-order ravel of block coords.
    jitter: optional smooth deformation amplitude (fraction of h), identical on
      all ranks because it is a function of the global coordinates.
    periodic_dims: directions in which the global box is periodic (config 4:
      all three).  Needs at least two blocks along each of them, so that no
      rank holds two images of a node; the neighbour lists then come from the
      router-based discovery (`distributed/discover.py`, a collective call)
      on the periodic lattice keys -- a pair of ranks may meet twice, at the
      cut and through the wrap-around.
  """
  ndim = len(block_grid)
  block_grid = tuple(int(p) for p in block_grid)
  ns = (int(n),) * ndim if np.isscalar(n) else tuple(int(k) for k in n)
  if len(ns) != ndim:
    raise ValueError(f'n has {len(ns)} entries for a {ndim}-d block grid')
  periodic_dims = tuple(int(d) for d in periodic_dims)
  # directions with a single block: both images of a boundary node live on
  # this rank (local periodic links, summed before / copied after the exchange)
  self_periodic = tuple(d for d in periodic_dims if block_grid[d] == 1)
  if jitter and periodic_dims:
    raise NotImplementedError('jitter on a periodic box')
  coords_b = tuple(int(c) for c in np.unravel_index(rank, block_grid))
  pm = box_mesh(ns, (0.0,) * ndim, (1.0,) * ndim, periodic_dims=self_periodic)
  # affine map of the unit block into its slot of the global box
  x = np.array(pm.node_coords)
  for d in range(ndim):
    x[:, d] = lo + (hi - lo) * (coords_b[d] + x[:, d]) / block_grid[d]
  if jitter:
    h = (hi - lo) / max(k * g for k, g in zip(ns, block_grid))
    s = np.ones(len(x))
    for d in range(ndim):
      s = s * np.sin(np.pi * (x[:, d] - lo) / (hi - lo))
    x = x + jitter * h * s[:, None] * np.cos(
        2 * np.pi * x[:, ::-1] / (hi - lo))
  # keep only the faces that lie on the global boundary (the group lists, per
  # axis, the FIRST side then the LAST side; box_mesh turns the two sides of a
  # self-periodic axis into links instead)
  bfaces, at, keep = pm.physical_groups.get('boundary'), 0, []
  for d in range(ndim):
    if d in self_periodic:
      continue
    nf = int(np.prod([ns[a] for a in range(ndim) if a != d]))
    first, last = bfaces[at:at + nf], bfaces[at + nf:at + 2 * nf]
    at += 2 * nf
    if d in periodic_dims:
      continue
    if coords_b[d] == 0:
      keep.append(first)
    if coords_b[d] == block_grid[d] - 1:
      keep.append(last)
  groups = {'boundary': np.concatenate(keep)} if keep else {}
  pm = pm.replace(node_coords=x, physical_groups=groups)
  perm = None
  if tile and tile < max(ns):
    perm = tiled_element_order(ns, ndim, tile)
    pm = pm.replace(elements=pm.elements[perm])
  rp = refine_premesh(pm, Nodes1D.create(P, NodeType.GAUSS_LOBATTO_LEGENDRE))

  # block-local GLL lattice coordinates of every local node
  m = P - 1
  L = np.array(ns, dtype=np.int64) * m        # last lattice index per dim
  ecoord = np.stack(np.meshgrid(*[np.arange(k) for k in ns], indexing='ij'),
                    axis=-1).reshape(-1, ndim)              # (E, d)
  if perm is not None:
    ecoord = ecoord[perm]
  lcoord = np.stack(np.meshgrid(*([np.arange(P)] * ndim), indexing='ij'),
                    axis=-1).reshape(-1, ndim)              # (n_loc, d)
  lat = np.zeros((rp.num_nodes, ndim), dtype=np.int32)
  flat = rp.elements.reshape(-1)
  for d in range(ndim):
    vals = (ecoord[:, None, d] * m + lcoord[None, :, d]).reshape(-1)
    lat[flat, d] = vals
  glob = lat.astype(np.int64) + np.array(coords_b, dtype=np.int64) * L
  gdims = [int(block_grid[d] * L[d] + (0 if d in periodic_dims else 1))
           for d in range(ndim)]
  for d in periodic_dims:
    glob[:, d] %= gdims[d]                    # x = hi is the image of x = lo
  key = np.ravel_multi_index(tuple(glob[:, d] for d in range(ndim)), gdims)

  neighbors, indices = [], []
  arrays = rp.finalize_all()
  local = {}
  if self_periodic:
    # class representative (smallest position) of every node, images of it
    ni = np.asarray(arrays['node_indices']).astype(np.int64)
    gi = np.asarray(arrays['exchange_gather_indices']).astype(np.int32)
    local = dict(local_gather=gi,
                 local_unique=np.asarray(arrays['exchange_unique_indices']),
                 local_rep=ni[gi].astype(np.int32))
    is_rep = ni == np.arange(len(ni))
  if periodic_dims and int(np.prod(block_grid)) > 1:
    from swirl_fem_amd.distributed import discover
    # setup-time routing of a few integers per surface node: always over a
    # host (gloo) group, whatever backend carries the solver's traffic.  Only
    # class representatives take part (one image of a node per rank).
    found = discover.discover_neighbors(
        np.where(is_rep, key, -1) if self_periodic else key,
        group=_setup_group(), device='cpu')
    neighbors, indices = list(found.neighbors), list(found.indices)
  for off in itertools.product((-1, 0, 1), repeat=ndim):
    if periodic_dims:
      break
    if not any(off):
      continue
    nb = tuple(c + o for c, o in zip(coords_b, off))
    if any(c < 0 or c >= g for c, g in zip(nb, block_grid)):
      continue
    sel = np.ones(rp.num_nodes, dtype=bool)
    for d, o in enumerate(off):
      if o == -1:
        sel &= lat[:, d] == 0
      elif o == 1:
        sel &= lat[:, d] == L[d]
    pos = np.nonzero(sel)[0]
    pos = pos[np.argsort(key[pos], kind='stable')]
    neighbors.append(int(np.ravel_multi_index(nb, block_grid)))
    indices.append(pos.astype(np.int32))
  # several offsets never map to the same rank without periodic wrap-around
  assert len(set(neighbors)) == len(neighbors)
  order = np.argsort(neighbors)
  plan = comm.NeighborPlan(rank=rank, neighbors=[neighbors[i] for i in order],
                           indices=[indices[i] for i in order], **local)

  world = int(np.prod(block_grid))
  mesh = Mesh.create(
      gridpoints_1d=rp.gridpoints_1d, device=device, dtype=dtype,
      axis_name='blocks' if world > 1 else None,
      neighbor_plan=plan if world > 1 else None, **{
          k: v for k, v in arrays.items()
          if k in ('node_coords', 'elements', 'node_indices', 'physical_masks')
      },
      exchange_gather_indices=(
          arrays['exchange_gather_indices'] if self_periodic else
          np.concatenate(indices).astype(np.int32)
          if world > 1 and indices else None),
      exchange_unique_indices=(arrays['exchange_unique_indices']
                               if self_periodic else None))
  return BlockPartition(mesh=mesh, rank=rank, block_grid=block_grid,
                        block_coords=coords_b,
                        num_global_nodes=int(np.prod(gdims)), plan=plan,
                        global_keys=key.astype(np.int64), premesh=pm)
