"""Setup products for direct-stiffness summation on the device.

`AssemblyPlan` classifies the nodes of a mesh by multiplicity (how many element
slots reference them) and, on demand, builds the inverse map (CSR by node) used
by the deterministic assembly `sfem_scatter_csr`.  These are one-off setup
computations (torch index ops on the device); the reference has no counterpart
because XLA's scatter-add hides them (core/gather_scatter.py:130-133).
"""

from __future__ import annotations

import dataclasses

import torch


@dataclasses.dataclass(eq=False)
class AssemblyPlan:
  multiplicity: torch.Tensor        # (N,) int32
  zero_range: tuple                 # [lo, hi) covering every node with mult != 1
  num_shared: int
  _mesh: object = None
  _csr: tuple | None = None

  @classmethod
  def build(cls, mesh) -> 'AssemblyPlan':
    el = mesh.elements.reshape(-1)
    valid = el[el >= 0].to(torch.int64)
    mult = torch.bincount(valid, minlength=mesh.num_nodes).to(torch.int32)
    other = torch.nonzero(mult != 1).reshape(-1)
    if other.numel():
      rng = (int(other.min()), int(other.max()) + 1)
    else:
      rng = (0, 0)
    num_shared = int((mult > 1).sum())
    return cls(multiplicity=mult, zero_range=rng, num_shared=num_shared,
               _mesh=mesh)

  def csr(self):
    """(offsets (N+1,) int64, slots (nnz,) int32), slots ascending per node."""
    if self._csr is None:
      el = self._mesh.elements.reshape(-1).to(torch.int64)
      n = self._mesh.num_nodes
      key = torch.where(el >= 0, el, torch.full_like(el, n))
      order = torch.argsort(key, stable=True)
      counts = torch.bincount(key, minlength=n + 1)[:n]
      offsets = torch.zeros(n + 1, dtype=torch.int64, device=el.device)
      offsets[1:] = torch.cumsum(counts, 0)
      slots = order[: int(offsets[-1])].to(torch.int32).contiguous()
      self._csr = (offsets.contiguous(), slots)
    return self._csr
