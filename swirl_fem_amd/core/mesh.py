"""Immutable mesh whose arrays live in MI355X HBM.

API follows the reference `swirl_fem/core/mesh.py` (`Mesh` fields :75-88,
`create` :90-133, `gather` :155-160, `scatter` :165-168, `element_coords`
:170-172, `exchange` :174-179) with `torch.Tensor`s in place of `jax.Array`s.

HBM layout: `node_coords (N, d)` real, `elements (E, n)` int32 (lexicographic
node order inside an element, axis 0 slowest), `node_indices (N,)` int32,
`physical_masks[name] (N,)` bool, `exchange_gather_indices (S,)` int32.
Setup products that the kernels need (shared/owned classification of element
slots, CSR inverse map for the deterministic assembly) are built lazily and
cached on the instance (`assembly_plan`).
"""

from __future__ import annotations

import dataclasses
from collections.abc import Mapping

import numpy as np
import torch

from swirl_fem_amd.core import gather_scatter
from swirl_fem_amd.core.interpolation import Nodes1D
from swirl_fem_amd.core.interpolation import NodeType


def default_device() -> torch.device:
  return torch.device('cuda' if torch.cuda.is_available() else 'cpu')


def _as_tensor(x, dtype=None, device=None):
  if x is None:
    return None
  if isinstance(x, torch.Tensor):
    t = x
  else:
    t = torch.from_numpy(np.ascontiguousarray(np.asarray(x)))
  if dtype is not None:
    t = t.to(dtype)
  return t.to(device).contiguous()


@dataclasses.dataclass(frozen=True, eq=False)
class Mesh:
  """An N-dimensional mesh of equal-order tensor-product elements."""
  node_coords: torch.Tensor
  elements: torch.Tensor
  node_indices: torch.Tensor
  order: int
  gridpoints_1d: Nodes1D
  physical_masks: Mapping[str, torch.Tensor] = dataclasses.field(
      default_factory=dict)
  exchange_gather_indices: torch.Tensor | None = None
  exchange_unique_indices: np.ndarray | None = None
  axis_name: str | None = None
  # this rank's neighbour lists when partitioned (build-side addition)
  neighbor_plan: object | None = None
  _cache: dict = dataclasses.field(default_factory=dict, repr=False,
                                   compare=False)

  @classmethod
  def create(cls, node_coords, elements, node_indices=None,
             gridpoints_1d: Nodes1D | None = None, physical_masks=None,
             exchange_gather_indices=None, exchange_unique_indices=None,
             axis_name: str | None = None, *, neighbor_plan=None,
             device=None, dtype=None) -> 'Mesh':
    """Creates a `Mesh`; arrays are placed on `device` (default: the GPU)."""
    device = torch.device(device) if device is not None else default_device()
    ndim = node_coords.shape[-1]
    num_nodes_per_element = elements.shape[-1]
    physical_masks = physical_masks or {}

    if gridpoints_1d is None:
      num_points = int(round(np.exp(np.log(num_nodes_per_element) / ndim)))
      gridpoints_1d = Nodes1D.create(num_points=num_points,
                                     node_type=NodeType.NEWTON_COTES)
    if num_nodes_per_element != gridpoints_1d.num_points ** ndim:
      raise ValueError(
          'Expected the number of nodes in each element of `mesh` to be equal '
          f'to the number of gridpoints in {ndim} dimensions. But got '
          f'{num_nodes_per_element} != {gridpoints_1d.num_points} ** {ndim}.')

    coords = _as_tensor(node_coords, device=device)
    if dtype is not None:
      coords = coords.to(dtype)
    elif not coords.dtype.is_floating_point:
      coords = coords.to(torch.float64)
    if node_indices is None:
      node_indices = torch.arange(len(node_coords), dtype=torch.int32,
                                  device=device)
    if exchange_unique_indices is not None:
      exchange_unique_indices = np.asarray(exchange_unique_indices)
    return cls(
        node_coords=coords,
        elements=_as_tensor(elements, torch.int32, device),
        node_indices=_as_tensor(node_indices, torch.int32, device),
        order=gridpoints_1d.num_points - 1,
        gridpoints_1d=gridpoints_1d,
        physical_masks={k: _as_tensor(v, torch.bool, device)
                        for k, v in physical_masks.items()},
        exchange_gather_indices=_as_tensor(exchange_gather_indices,
                                           torch.int32, device),
        exchange_unique_indices=exchange_unique_indices,
        axis_name=axis_name, neighbor_plan=neighbor_plan)

  def replace(self, **kw) -> 'Mesh':
    kw.setdefault('_cache', {})
    return dataclasses.replace(self, **kw)

  def replicate(self, members: int) -> 'Mesh':
    """`members` disjoint copies of this mesh as ONE mesh: copy b owns the
    nodes [b N, (b + 1) N) and the elements [b E, (b + 1) E), periodic images
    stay inside their copy.  An ensemble of fields on the same mesh is one
    field on this mesh ((B, N, ...) viewed as (B N, ...)), so every operator
    kernel works on all members in a single launch (`StokesSEM.ensemble`)."""
    if members < 1:
      raise ValueError(f'members must be positive; got {members}')
    if self.axis_name is not None or self.neighbor_plan is not None:
      raise NotImplementedError('ensembles of a partitioned mesh')
    if members == 1:
      return self
    N, dev = self.num_nodes, self.device
    off = torch.arange(members, dtype=torch.int32, device=dev) * N
    elements = (self.elements[None] + off[:, None, None]).reshape(
        -1, self.num_nodes_per_element)
    node_indices = (self.node_indices[None] + off[:, None]).reshape(-1)
    gi, ui = self.exchange_gather_indices, self.exchange_unique_indices
    if gi is not None and gi.numel():
      gi = (gi[None] + off[:, None]).reshape(-1)
      if ui is not None:
        ui = np.asarray(ui)
        width = int(ui.max()) + 1 if ui.size else 0
        ui = (ui[None] + width * np.arange(members)[:, None]).reshape(-1)
    return self.replace(
        node_coords=self.node_coords.repeat(members, 1), elements=elements,
        node_indices=node_indices,
        physical_masks={k: v.repeat(members)
                        for k, v in self.physical_masks.items()},
        exchange_gather_indices=gi, exchange_unique_indices=ui)

  # ------------------------------------------------------------- properties
  @property
  def ndim(self) -> int:
    return self.node_coords.shape[-1]

  @property
  def num_nodes(self) -> int:
    return self.node_coords.shape[-2]

  @property
  def num_elements(self) -> int:
    return self.elements.shape[-2]

  @property
  def num_nodes_per_element(self) -> int:
    return self.elements.shape[-1]

  @property
  def device(self) -> torch.device:
    return self.node_coords.device

  @property
  def dtype(self) -> torch.dtype:
    return self.node_coords.dtype

  # -------------------------------------------------------------------- ops
  def gather(self, u: torch.Tensor) -> torch.Tensor:
    """Nodal values `(N,)` -> element-local values `(E, n)`."""
    if tuple(u.shape) != (self.num_nodes,):
      raise ValueError(
          f'Expected `u` to have shape ({self.num_nodes},) but got: '
          f'{tuple(u.shape)}.')
    return gather_scatter.gather(u, indices=self.elements, fill_value=0.)

  def scatter(self, u_local: torch.Tensor) -> torch.Tensor:
    """Element-local values `(E, n)` -> summed nodal values `(N,)`."""
    return gather_scatter.scatter(u_local, indices=self.elements,
                                  num_nodes=self.num_nodes)

  def element_coords(self) -> torch.Tensor:
    """Coordinates of the nodes of each element, `(E, n, d)`."""
    from swirl_fem_amd import _ops
    return _ops.gather_rows(self.node_coords, self.elements)

  def exchange(self, u: torch.Tensor) -> torch.Tensor:
    """QQ^T on nodal values (periodic images / partition-shared nodes)."""
    return gather_scatter.exchange(
        u, gather_indices=self.exchange_gather_indices,
        unique_indices=self.exchange_unique_indices,
        axis_name=self.axis_name, plan=self.neighbor_plan)

  # ------------------------------------------------------- kernel-side plans
  def assembly_plan(self):
    """Slot classification + CSR inverse map used by the fused operators."""
    if 'assembly' not in self._cache:
      from swirl_fem_amd.core import assembly
      self._cache['assembly'] = assembly.AssemblyPlan.build(self)
    return self._cache['assembly']
