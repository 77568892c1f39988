"""Structured order-1 premeshes on `[a, b]^d`.

`unit_cube_mesh` has the signature and output numbering of the reference
`swirl_fem/common/premesh_commons.py:67-145` (node id = C-order ravel of the
vertex multi-index, elements in C order with axis 0 slowest, element vertices
in lexicographic order, boundary / periodic facets enumerated per axis, FIRST
side then LAST side) but is assembled with array arithmetic so that 128^3
element meshes are built in seconds.
"""

from __future__ import annotations

from collections.abc import Sequence

import numpy as np

from swirl_fem_amd.core.premesh import Premesh


def _cell_vertices(starts, ndim, num_nodes_1d, free_axes):
  """Vertex ids of cells: `starts (m, ndim)` lower multi-indices; the cell
  extends by one node along `free_axes` only.  Returns `(m, 2^len(free))`.
  `num_nodes_1d`: nodes per direction (an int, or one per axis)."""
  dims = (np.full(ndim, num_nodes_1d, dtype=np.int64)
          if np.isscalar(num_nodes_1d) else np.asarray(num_nodes_1d, np.int64))
  strides = np.concatenate([np.cumprod(dims[:0:-1])[::-1],
                            [1]]).astype(np.int64)
  base = starts.astype(np.int64) @ strides
  offs = np.zeros(1, dtype=np.int64)
  for ax in free_axes:  # ascending axis order -> lexicographic corner order
    offs = (offs[:, None] + np.array([0, strides[ax]])[None, :]).reshape(-1)
  return base[:, None] + offs[None, :]


def box_mesh(num_elements: Sequence[int], lo: Sequence[float],
             hi: Sequence[float], periodic_dims: Sequence[int] = ()) -> Premesh:
  """Uniform order-1 mesh of a box with its own element count per direction
  (`unit_cube_mesh` is the equal-count case; same numbering conventions)."""
  ns = tuple(int(k) for k in num_elements)
  ndim = len(ns)
  n1 = tuple(k + 1 for k in ns)
  grids = np.meshgrid(*[np.linspace(lo[d], hi[d], num=n1[d])
                        for d in range(ndim)], indexing='ij')
  node_coords = np.stack(grids, axis=-1).reshape(int(np.prod(n1)), ndim)
  cells = np.stack(np.meshgrid(*[np.arange(k) for k in ns], indexing='ij'),
                   axis=-1).reshape(-1, ndim)
  elements = _cell_vertices(cells, ndim, n1, range(ndim)).astype(np.int32)

  def side_facets(axis, last):
    free = [ax for ax in range(ndim) if ax != axis]
    if free:
      sub = np.stack(np.meshgrid(*[np.arange(ns[ax]) for ax in free],
                                 indexing='ij'), axis=-1).reshape(-1, len(free))
    else:
      sub = np.zeros((1, 0), dtype=np.int64)
    starts = np.zeros((len(sub), ndim), dtype=np.int64)
    starts[:, free] = sub
    starts[:, axis] = ns[axis] if last else 0
    return _cell_vertices(starts, ndim, n1, free).astype(np.int32)

  boundary, links = [], []
  for axis in range(ndim):
    first, last = side_facets(axis, False), side_facets(axis, True)
    if axis in periodic_dims:
      links.append(np.stack([first, last], axis=1))
    else:
      boundary += [first, last]
  physical_groups = {}
  if boundary:
    physical_groups['boundary'] = np.concatenate(boundary).astype(np.int32)
  periodic_links = (np.concatenate(links).astype(np.int32) if links else None)
  return Premesh.create(node_coords=node_coords, elements=elements,
                        periodic_links=periodic_links,
                        physical_groups=physical_groups)


def unit_cube_mesh(num_elements_per_dim: int, ndim: int = 2, a: float = 0.0,
                   b: float = 1.0, periodic_dims: Sequence[int] = (),
                   partitions: np.ndarray | None = None) -> Premesh:
  """Uniform order-1 mesh over `[a, b]^ndim` (see module docstring)."""
  n = num_elements_per_dim
  n1 = n + 1
  coords_1d = np.linspace(a, b, num=n1)
  grids = np.meshgrid(*([coords_1d] * ndim), indexing='ij')
  node_coords = np.stack(grids, axis=-1).reshape(n1 ** ndim, ndim)

  cells = np.stack(np.meshgrid(*([np.arange(n)] * ndim), indexing='ij'),
                   axis=-1).reshape(-1, ndim)
  elements = _cell_vertices(cells, ndim, n1, range(ndim)).astype(np.int32)

  def side_facets(axis, last):
    free = [ax for ax in range(ndim) if ax != axis]
    if free:
      sub = np.stack(np.meshgrid(*([np.arange(n)] * len(free)),
                                 indexing='ij'), axis=-1).reshape(-1, len(free))
    else:
      sub = np.zeros((1, 0), dtype=np.int64)
    starts = np.zeros((len(sub), ndim), dtype=np.int64)
    starts[:, free] = sub
    starts[:, axis] = n if last else 0
    return _cell_vertices(starts, ndim, n1, free).astype(np.int32)

  boundary, links = [], []
  for axis in range(ndim):
    first, last = side_facets(axis, False), side_facets(axis, True)
    if axis in periodic_dims:
      links.append(np.stack([first, last], axis=1))
    else:
      boundary += [first, last]

  physical_groups = {}
  if boundary:
    physical_groups['boundary'] = np.concatenate(boundary).astype(np.int32)
  periodic_links = (np.concatenate(links).astype(np.int32) if links else None)

  if partitions is not None:
    partitions = np.asarray(partitions)
    for axis in range(ndim):
      assert n % partitions.shape[axis] == 0, partitions.shape
      partitions = np.repeat(partitions, repeats=n // partitions.shape[axis],
                             axis=axis)
    partitions = partitions.reshape(len(elements))

  return Premesh.create(node_coords=node_coords, elements=elements,
                        periodic_links=periodic_links,
                        physical_groups=physical_groups, partitions=partitions)
