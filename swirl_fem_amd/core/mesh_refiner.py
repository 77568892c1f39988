"""p-refinement of order-1 premeshes to tensor-product order-p nodes.

`refine_premesh(premesh, gridpoints_1d)` has the contract of the reference
`swirl_fem/core/mesh_refiner.py:35-57` and reproduces its node numbering
exactly (vertices first, then for every facet signature in
`itertools.product(FIRST, LAST, INNER)` order the interior nodes of each
*newly seen* facet, element by element; shared facets are re-used through the
2^k k! orientation tables of `facet_util`, :143-229).

The reference walks a Python dict facet by facet (:198-218), which is
unusable at 64^3 elements x 27 facet types.  Here each facet signature is
processed as one batch: facets are keyed by their sorted vertex tuple, looked
up / first-occurrence-ranked with `np.unique`, and orientation permutations
come from a table lookup, so a 64^3, p=7 refinement is a few seconds of NumPy.
"""

from __future__ import annotations

import numpy as np

from swirl_fem_amd.common import facet_util
from swirl_fem_amd.common.facet_util import FacetDimType
from swirl_fem_amd.core.interpolation import BarycentricInterpolator
from swirl_fem_amd.core.interpolation import Nodes1D
from swirl_fem_amd.core.premesh import Premesh


def refine_premesh(premesh: Premesh, gridpoints_1d: Nodes1D, *,
                   face_orientation: str = 'corrected') -> Premesh:
  """Returns the p-refined premesh with `gridpoints_1d` nodes per direction.

  Node numbering is the reference's (first-touch order, facets before element
  interiors).  `face_orientation` selects how a facet that was already refined
  by an earlier element is read back by a later one:

  * `'corrected'` (default): the refined nodes are permuted by the orientation
    of the later element's vertex ordering relative to the stored one.
  * `'reference'`: the reference's lookup (core/mesh_refiner.py:213-217), which
    builds the key from the *inverse* vertex permutation.  The two agree for
    every orientation that is its own inverse -- all edge orientations, and
    flips / transposes of faces, hence on structured and Gmsh-generated
    meshes -- but for a face seen rotated by +-90 degrees the reference places
    the face-interior nodes of the later element at the wrong positions (the
    element no longer matches the multilinear map of its vertices;
    `tests/test_host_logic.py::test_refiner_random_orientations`).  Kept for
    bit-for-bit comparison with the reference's output.
  """
  if premesh.order != 1:
    raise ValueError(f'Expecting mesh of order 1. Got {premesh.order}.')
  if face_orientation not in ('corrected', 'reference'):
    raise ValueError(f'unknown face_orientation {face_orientation!r}')
  return _BatchRefiner(premesh, gridpoints_1d,
                       face_orientation == 'reference').refine()


def _row_view(a: np.ndarray) -> np.ndarray:
  """Views the rows of a 2D int array as opaque scalars (for np.unique)."""
  a = np.ascontiguousarray(a)
  return a.view(np.dtype((np.void, a.dtype.itemsize * a.shape[1]))).reshape(-1)


class _FacetTable:
  """Known facets of one dimension: sorted-vertex key -> (orientation, start)."""

  def __init__(self, width: int):
    self.keys = np.zeros((0, width), dtype=np.int64)
    self.orient = np.zeros((0, width), dtype=np.int64)
    self.start = np.zeros(0, dtype=np.int64)

  def lookup_or_insert(self, facets: np.ndarray, next_node: int,
                       nodes_per_facet: int, allow_new: bool):
    """Returns (entry id per facet, is_new per facet, number of new facets).

    New facets get node blocks `[next_node + r*m, next_node + (r+1)*m)` where r
    ranks the new facets by first position in `facets`.
    """
    skey = np.sort(facets, axis=1)
    known = len(self.keys)
    both = np.concatenate([self.keys, skey])
    _, first, inv = np.unique(_row_view(both), return_index=True,
                              return_inverse=True)
    first_of = first[inv.reshape(-1)][known:]     # first occurrence in `both`
    pos = np.arange(len(facets)) + known
    is_new = first_of == pos
    num_new = int(is_new.sum())
    if num_new and not allow_new:
      raise ValueError('facet is not a facet of any element of the premesh')
    # entry ids: existing keep theirs; new ones are appended in order
    new_entry = known + np.cumsum(is_new) - 1
    entry_of_pos = np.empty(len(both), dtype=np.int64)
    entry_of_pos[:known] = np.arange(known)
    entry_of_pos[pos[is_new]] = new_entry[is_new]
    entry = entry_of_pos[first_of]
    if num_new:
      self.keys = np.concatenate([self.keys, skey[is_new]])
      self.orient = np.concatenate([self.orient, facets[is_new]])
      self.start = np.concatenate([
          self.start,
          next_node + nodes_per_facet * np.arange(num_new, dtype=np.int64)])
    return entry, is_new, num_new


class _BatchRefiner:

  def __init__(self, premesh: Premesh, gridpoints_1d: Nodes1D,
               reference_keys: bool = False):
    self.premesh = premesh
    self.reference_keys = reference_keys
    self.gridpoints_1d = gridpoints_1d
    self.num_points = gridpoints_1d.num_points
    self.interpolator = BarycentricInterpolator(
        ndim=premesh.ndim, gridpoints_1d=premesh.gridpoints_1d,
        evalpoints_1d=gridpoints_1d)
    self.continuous = gridpoints_1d.is_continuous()
    self.coord_blocks = [np.asarray(premesh.node_coords)] if self.continuous \
        else []
    self.num_nodes = premesh.num_nodes if self.continuous else 0
    self.tables = {k: _FacetTable(2 ** k) for k in range(1, premesh.ndim + 1)}

  # ------------------------------------------------------------------ helpers
  def _append_nodes(self, coords: np.ndarray | None, count: int) -> int:
    start = self.num_nodes
    if coords is not None and count:
      self.coord_blocks.append(coords.reshape(count, self.premesh.ndim))
    self.num_nodes += count
    return start

  def _refine_facets(self, facets: np.ndarray, ndim: int,
                     target_coords: np.ndarray | None = None) -> np.ndarray:
    """Refines `(F, 2^ndim)` facets to `(F, P^ndim)` node ids."""
    p, full_dim = self.num_points, self.premesh.ndim
    nf = len(facets)
    facets_nd = np.asarray(facets).astype(np.int64).reshape([nf] + [2] * ndim)
    target = np.full([nf] + [p] * ndim, -1, dtype=np.int64)
    coords_nd = None
    if target_coords is not None:
      coords_nd = target_coords.reshape([nf] + [p] * ndim + [full_dim])

    for facet_type in facet_util.get_facet_types(ndim):
      src = facet_util.slice_from_facet_type(facet_type, False)
      dst = facet_util.slice_from_facet_type(facet_type, True)
      k = facet_type.count(FacetDimType.INNER)
      curr = facets_nd[(slice(None), *src)].reshape(nf, -1)
      if k == 0:
        target[(slice(None), *dst)] = curr[:, 0]
        continue
      m = (p - 2) ** k
      inner_shape = [nf] + [p - 2] * k
      fcoords = None
      if coords_nd is not None:
        fcoords = coords_nd[(slice(None), *dst, slice(None))].reshape(
            nf, m, full_dim)

      if k == full_dim and fcoords is not None:
        # element interiors: never shared
        start = self._append_nodes(fcoords, nf * m)
        ids = start + np.arange(nf * m, dtype=np.int64)
        target[(slice(None), *dst)] = ids.reshape(inner_shape)
        continue

      table = self.tables[k]
      entry, is_new, num_new = table.lookup_or_insert(
          curr, self.num_nodes, m, allow_new=fcoords is not None)
      if num_new:
        self._append_nodes(fcoords[is_new], num_new * m)
      # orientation of the current vertex ordering w.r.t. the stored one
      orient = table.orient[entry]                       # (nf, 2^k)
      if self.reference_keys:    # position of each stored vertex in `curr`
        key = (curr[:, None, :] == orient[:, :, None]).argmax(axis=2)
      else:                      # position of each vertex of `curr` in `orient`
        key = (orient[:, None, :] == curr[:, :, None]).argmax(axis=2)
      base = 2 ** k
      code = key @ (base ** np.arange(base, dtype=np.int64))
      codes, perms = facet_util.orientation_table(k, p - 2)
      row = np.searchsorted(codes, code)
      if np.any(codes[np.minimum(row, len(codes) - 1)] != code):
        raise ValueError('facet vertex ordering is not a cube orientation')
      ids = table.start[entry][:, None] + perms[row]
      target[(slice(None), *dst)] = ids.reshape(inner_shape)

    return target.reshape(nf, p ** ndim)

  # --------------------------------------------------------------------- main
  def refine(self) -> Premesh:
    pm, ndim, p = self.premesh, self.premesh.ndim, self.num_points
    interp = self.interpolator.interpolation_matrix()          # (P^d, 2^d)
    elem_vertex_coords = np.asarray(pm.node_coords)[np.asarray(pm.elements)]
    target_coords = np.matmul(interp[None], elem_vertex_coords)  # (E, P^d, d)

    if not self.continuous:
      count = pm.num_elements * p ** ndim
      self._append_nodes(target_coords, count)
      elements = np.arange(count, dtype=np.int64).reshape(
          pm.num_elements, p ** ndim)
    else:
      elements = self._refine_facets(pm.elements, ndim, target_coords)

    node_coords = np.concatenate(self.coord_blocks, axis=0)

    physical_groups = {}
    if pm.physical_groups and self.continuous:
      for name, facets in pm.physical_groups.items():
        if not np.asarray(facets).size:
          raise ValueError(f'Got an empty physical group "{name}".')
        physical_groups[name] = self._refine_facets(
            facets, ndim - 1).astype(np.int64)

    periodic_links = None
    if pm.periodic_links is not None and self.continuous:
      links = np.asarray(pm.periodic_links)
      periodic_links = np.stack([
          self._refine_facets(links[:, 0, :], ndim - 1),
          self._refine_facets(links[:, 1, :], ndim - 1)], axis=-2)

    return Premesh.create(
        node_coords=node_coords, elements=elements.astype(np.int32),
        gridpoints_1d=self.gridpoints_1d, physical_groups=physical_groups,
        periodic_links=periodic_links, partitions=pm.partitions)
