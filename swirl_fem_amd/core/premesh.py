"""Staging mesh (host, NumPy) that is refined and then finalised into a `Mesh`.

API follows the reference `swirl_fem/core/premesh.py` (`Premesh` :38-71,
`create` :73-115, `finalize` :141-222).  `finalize()` builds the device `Mesh`
(periodic de-duplication, exchange indices, boundary masks).  For a partitioned
premesh the reference regroups elements per partition, renumbers nodes locally
and `pmap`-places one partition per device (:170-222); here one *process* owns
one partition, so `finalize(axis_name)` returns this rank's `Mesh` (rank taken
from `torch.distributed` or passed explicitly) and `finalize_all(axis_name)`
returns the stacked per-partition arrays, which is what the index-parity tests
compare against the reference's `(P, ...)`-shaped outputs.
"""

from __future__ import annotations

import dataclasses
from collections.abc import Mapping

import numpy as np

from swirl_fem_amd.core import gather_scatter
from swirl_fem_amd.core.interpolation import Nodes1D
from swirl_fem_amd.core.interpolation import NodeType


def _mask(facets: np.ndarray, node_indices: np.ndarray) -> np.ndarray:
  """Boolean mask of which `node_indices` occur in `facets`."""
  return np.isin(node_indices, np.unique(np.asarray(facets).reshape(-1)))


def _default_gridpoints(num_nodes_per_element: int, ndim: int) -> Nodes1D:
  num_points = int(round(np.exp(np.log(num_nodes_per_element) / ndim)))
  return Nodes1D.create(num_points=num_points,
                        node_type=NodeType.NEWTON_COTES)


@dataclasses.dataclass(frozen=True)
class Premesh:
  """Intermediate mesh format: order, nodes, elements, groups, links, parts."""
  order: int
  gridpoints_1d: Nodes1D
  node_coords: np.ndarray
  elements: np.ndarray
  physical_groups: Mapping[str, np.ndarray]
  periodic_links: np.ndarray | None = None
  partitions: np.ndarray | None = None

  @classmethod
  def create(cls, node_coords, elements, order=None, gridpoints_1d=None,
             physical_groups=None, periodic_links=None,
             partitions=None) -> 'Premesh':
    ndim = node_coords.shape[-1]
    n = elements.shape[-1]
    if gridpoints_1d is None:
      gridpoints_1d = _default_gridpoints(n, ndim)
    if n != gridpoints_1d.num_points ** ndim:
      raise ValueError(
          'Expected the number of nodes in each element to be equal '
          f'to the number of gridpoints in {ndim} dimensions. But got '
          f'{n} != {gridpoints_1d.num_points} ** {ndim}.')
    if physical_groups is None:
      physical_groups = {}
    if order is None:
      order = gridpoints_1d.num_points - 1
    return cls(order=order, gridpoints_1d=gridpoints_1d,
               node_coords=node_coords, elements=elements,
               physical_groups=physical_groups, periodic_links=periodic_links,
               partitions=partitions)

  def replace(self, **kw) -> 'Premesh':
    return dataclasses.replace(self, **kw)

  @property
  def ndim(self) -> int:
    return self.node_coords.shape[-1]

  @property
  def num_nodes(self) -> int:
    return self.node_coords.shape[-2]

  @property
  def num_elements(self) -> int:
    return len(self.elements)

  @property
  def num_nodes_per_element(self) -> int:
    return self.elements.shape[-1]

  def is_partitioned(self) -> bool:
    return self.partitions is not None

  # ------------------------------------------------------------------ finalize
  def _finalize_unpartitioned_arrays(self):
    node_indices = gather_scatter.get_unique_node_indices(
        node_indices=np.arange(self.num_nodes, dtype=np.int32),
        periodic_links=self.periodic_links)
    masks = {k: _mask(f, node_indices)
             for k, f in self.physical_groups.items()}
    if self.periodic_links is None:
      # every node is its own class: nothing to exchange (what the general
      # routine returns after sorting all N ids)
      gi, ui = np.zeros(0, np.int32), np.zeros(0, np.int32)
    else:
      gi, ui = gather_scatter.get_exchange_indices(node_indices)
    return dict(node_coords=self.node_coords, elements=self.elements,
                node_indices=node_indices, physical_masks=masks,
                exchange_gather_indices=gi, exchange_unique_indices=ui)

  def finalize_all(self, axis_name: str | None = None) -> dict:
    """Host arrays of the finalised mesh; leading axis P when partitioned."""
    if not self.is_partitioned():
      return self._finalize_unpartitioned_arrays()
    if not axis_name:
      raise ValueError('If partitioned, we need a non-trivial axis_name')

    element_indices = gather_scatter.group_by_partitions(self.partitions)
    # (P, Eloc, n) global node ids; padded elements are all SENTINEL
    pad = element_indices == gather_scatter.SENTINEL
    elements = self.elements[np.where(pad, 0, element_indices)]
    elements = np.where(pad[..., None], gather_scatter.SENTINEL, elements)

    local_nodes, local_elements = gather_scatter.get_local_elements(elements)
    node_indices = gather_scatter.get_unique_node_indices(
        local_nodes, periodic_links=self.periodic_links)
    gi, ui = gather_scatter.get_exchange_indices(node_indices)
    masks = {k: _mask(f, node_indices)
             for k, f in self.physical_groups.items()}
    # Geometry is taken from the *un-deduplicated* global ids (a periodic image
    # keeps its own coordinates; SURVEY 7 "reference quirk").  Padded local
    # nodes get the last node's coordinates like `x[-1]` would.
    node_coords = self.node_coords[local_nodes]
    return dict(node_coords=node_coords, elements=local_elements,
                node_indices=node_indices, physical_masks=masks,
                exchange_gather_indices=gi, exchange_unique_indices=ui,
                global_node_ids=local_nodes, element_indices=element_indices)

  def finalize(self, axis_name: str | None = None, *, rank: int | None = None,
               device=None, dtype=None):
    """Builds the device `Mesh` (this rank's partition when partitioned)."""
    from swirl_fem_amd.core.mesh import Mesh
    arrays = self.finalize_all(axis_name)
    if not self.is_partitioned():
      return Mesh.create(gridpoints_1d=self.gridpoints_1d, device=device,
                         dtype=dtype, **arrays)

    from swirl_fem_amd.distributed import comm
    if rank is None:
      rank = comm.get_rank()
    num_partitions = arrays['node_indices'].shape[0]
    if not 0 <= rank < num_partitions:
      raise ValueError(f'rank {rank} outside the {num_partitions} partitions')
    plan = comm.NeighborPlan.from_gather_indices(
        arrays['exchange_gather_indices'], rank)
    return Mesh.create(
        node_coords=arrays['node_coords'][rank],
        elements=arrays['elements'][rank],
        node_indices=arrays['node_indices'][rank],
        gridpoints_1d=self.gridpoints_1d,
        physical_masks={k: m[rank]
                        for k, m in arrays['physical_masks'].items()},
        exchange_gather_indices=arrays['exchange_gather_indices'][rank],
        exchange_unique_indices=None,
        axis_name=axis_name, neighbor_plan=plan, device=device, dtype=dtype)
